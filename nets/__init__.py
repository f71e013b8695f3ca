"""Drop-in for the reference's `nets` package (`/root/reference/nets/__init__.py:1-8`), MI355X-native underneath.

Same exported names; the body path (`s2g_body_pixel`, `s2g_body_vq`), the face generator and the FGD feature extractor
(`s2g_body_ae`) run on libtalkshow_hip.so.  The one component the scope leaves out — the Habibie et al.
baseline `LS3DCG` — raises `NotImplementedError` when constructed, naming the reason.
"""
from .smplx_face import TrainWrapper as s2g_face
from .smplx_body_vq import TrainWrapper as s2g_body_vq
from .smplx_body_pixel import TrainWrapper as s2g_body_pixel
from .body_ae import TrainWrapper as s2g_body_ae
from .out_of_scope import LS3DCG
from .base import TrainWrapperBaseClass

from .utils import normalize, denormalize
