"""Drop-in for the reference's `nets` package (`/root/reference/nets/__init__.py:1-8`), MI355X-native underneath.

Same exported names; the body path (`s2g_body_pixel`, `s2g_body_vq`) runs on libtalkshow_hip.so.  Components the
hot-path scope leaves out raise `NotImplementedError` when constructed, naming the reason.
"""
from .smplx_face import TrainWrapper as s2g_face
from .smplx_body_vq import TrainWrapper as s2g_body_vq
from .smplx_body_pixel import TrainWrapper as s2g_body_pixel
from .out_of_scope import s2g_body_ae, LS3DCG
from .base import TrainWrapperBaseClass

from .utils import normalize, denormalize
