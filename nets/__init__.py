"""Drop-in for the reference's `nets` package (`/root/reference/nets/__init__.py:1-8`), MI355X-native underneath.

Same exported names; the body path (`s2g_body_pixel`, `s2g_body_vq`), the face generator and the FGD feature extractor
(`s2g_body_ae`) run on libtalkshow_hip.so.  The one component the scope leaves out — the Habibie et al.
baseline `LS3DCG` — raises `NotImplementedError` when constructed, naming the reason.
"""
from importlib import import_module as _import

# exported name -> (module of this package, attribute): the names `nets.init_model` and the reference's scripts look up
_EXPORTS = {
    "s2g_face": ("smplx_face", "TrainWrapper"),
    "s2g_body_vq": ("smplx_body_vq", "TrainWrapper"),
    "s2g_body_pixel": ("smplx_body_pixel", "TrainWrapper"),
    "s2g_body_ae": ("body_ae", "TrainWrapper"),
    "LS3DCG": ("out_of_scope", "LS3DCG"),
    "TrainWrapperBaseClass": ("base", "TrainWrapperBaseClass"),
    "normalize": ("utils", "normalize"),
    "denormalize": ("utils", "denormalize"),
}
for _name, (_mod, _attr) in _EXPORTS.items():
    globals()[_name] = getattr(_import(f"{__name__}.{_mod}"), _attr)
__all__ = sorted(_EXPORTS)
