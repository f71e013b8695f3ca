"""ctypes binding of libtalkshow_hip.so (C ABI: include/talkshow_hip.h; tuning / test aids: include/talkshow_hip_debug.h).

There is NO fallback: if the library has not been built, or no gfx950 device is present when a context is
requested, this module raises.  PyTorch is imported first on purpose — the library must share the HIP runtime
instance that owns the torch tensors whose device pointers it is handed.
"""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  (loads libamdhip64 before ours)

_HERE = os.path.dirname(os.path.abspath(__file__))
# TS_LIB_PATH: another build of the same library (the AddressSanitizer build of `make asan`, an A/B build of tools/ab_libs.sh); it must
# export every symbol of include/*.h like the default one (load() checks)
LIB_PATH = os.environ.get("TS_LIB_PATH") or os.path.join(_HERE, "lib", "libtalkshow_hip.so")

TS_SAMPLE_GREEDY, TS_SAMPLE_UNIFORMS, TS_SAMPLE_PHILOX, TS_TEACHER_FORCED = 0, 1, 2, 3


class TsTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


_vp, _i, _i64, _u64, _fp = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.POINTER(C.c_float)

# name -> (restype, argtypes); every symbol include/*.h declares
SIGNATURES = {
    "ts_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "ts_ctx_destroy": (None, [_vp]),
    "ts_last_error": (C.c_char_p, []),
    "ts_version": (C.c_char_p, []),
    "ts_stream_create": (_i, [_vp, C.POINTER(_vp)]),
    "ts_stream_create_cus": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "ts_debug_skinny_trace": (_i, [C.POINTER(C.c_uint64), _i]),
    "ts_debug_clock_sample": (_i, [_vp, _i, _i, _vp]),
    "ts_debug_conv_bands": (_i, [_i, _i, _i, C.POINTER(_i)]),
    "ts_debug_split_tile": (_i, [_i, _i, _i, _i, C.POINTER(_i)]),
    "ts_debug_tile_weights": (_i, [_vp, _i, _i, C.c_long, _i, _i, _vp]),
    "ts_assemble_full": (_i, [_vp, _vp, _i, _vp, _i, _i, _fp, _vp, _vp]),
    "ts_stream_destroy": (_i, [_vp, _vp]),
    "ts_audioenc_create": (_i, [_vp, C.POINTER(TsTensor), _i, _i, _i, _i, C.POINTER(_vp)]),
    "ts_convnet_destroy": (None, [_vp]),
    "ts_audioenc_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "ts_vqvae_create": (_i, [_vp, C.POINTER(TsTensor), _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ts_vqvae_destroy": (None, [_vp]),
    "ts_vqvae_encode": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "ts_vqvae_decode": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "ts_vqvae_decode_z": (_i, [_vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "ts_vqvae_decode_pair": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "ts_vqvae_forward": (_i, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp]),
    "ts_pixelcnn_create": (_i, [_vp, C.POINTER(TsTensor), _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ts_pixelcnn_destroy": (None, [_vp]),
    "ts_pixelcnn_generate": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _u64, _i64, _vp, _vp, _vp, _vp, _i, _vp]),
    "ts_pixelcnn_v_create": (_i, [_vp, C.POINTER(TsTensor), _i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ts_pixelcnn_v_destroy": (None, [_vp]),
    "ts_pixelcnn_v_generate": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _u64, _i64, _vp, _vp, _vp, _vp, _i, _vp]),
    "ts_pixelcnn_stream_open": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "ts_pixelcnn_stream_step": (_i, [_vp, _vp, _i, _i, _vp, _u64, _i64, _vp, _vp]),
    "ts_pixelcnn_stream_rows": (_i64, [_vp]),
    "ts_pixelcnn_stream_close": (None, [_vp]),
    "ts_face_create": (_i, [_vp, C.POINTER(TsTensor), _i, _i, _i, C.POINTER(_vp)]),
    "ts_face_destroy": (None, [_vp]),
    "ts_face_generate": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "ts_face_set_arith": (_i, [_vp, _i]),
    "ts_mfcc_create": (_i, [_vp, _i, _i, _i, C.POINTER(_vp)]),
    "ts_mfcc_destroy": (None, [_vp]),
    "ts_mfcc_num_frames": (_i, [_vp, C.c_long]),
    "ts_mfcc_forward": (_i, [_vp, _vp, _i, C.c_long, _vp, _vp]),
    "ts_mfcc_resampled_len": (C.c_long, [_vp, C.c_long]),
    "ts_mfcc_resample": (_i, [_vp, _vp, _i, C.c_long, _vp, _vp]),
    "ts_resample_kaiser_len": (C.c_long, [C.c_long, _i, _i]),
    "ts_resample_kaiser": (_i, [_vp, _vp, _i, C.c_long, _i, _i, _vp, _vp]),
    "ts_pixelcnn_graph_stats": (_i, [_vp, _vp, _i, _i, _i, C.POINTER(_i64), C.POINTER(C.c_double)]),
    "ts_body_pixel_infer": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _u64, _i64, _vp, _vp, _vp]),
    "ts_body_vq_infer": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "ts_op_conv1d": (_i, [_vp, _vp, _i, _i, _i, _fp, _fp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "ts_op_conv1d_timed": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, C.POINTER(C.c_float), _vp]),
    "ts_op_conv_taps48_timed": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, C.POINTER(C.c_float), _vp]),
    "ts_op_conv1d_strided_timed": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, C.POINTER(C.c_float), _vp]),
    "ts_debug_pixelcnn_graphs": (_i, [_vp, _vp]),
    "ts_debug_conv_sk_plan": (_i, [_i, _i, _i, _i, C.POINTER(C.c_int)]),
    "ts_debug_conv_sk_run": (_i, [_i, _i, _i, _i, C.POINTER(C.c_int)]),
    "ts_debug_conv_sk_supported": (_i, []),
    "ts_pixelcnn_graph_captures": (C.c_long, [_vp, _vp]),
    "ts_pixelcnn_prepare": (_i, [_vp, _i, _i, _i, _vp]),
    "ts_debug_conv_ring_pick": (_i, [_i, _i, _i]),
    "ts_debug_gate_act": (_i, [_vp, _vp, _vp, C.c_long, _vp]),
    "ts_op_vq_argmin": (_i, [_vp, _vp, _i, _vp, _i, _i, _vp, _vp]),
    "ts_op_linear": (_i, [_vp, _vp, _i, _i, _fp, _fp, _i, _i, _vp, _vp]),
    "ts_op_sample": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "ts_op_sample_philox": (_i, [_vp, _vp, _i, _i, _u64, _i64, C.c_uint32, _vp, _vp]),
    "ts_debug_skinny_chain": (_i, [_vp, _i, _i, _i, _i, _i, C.POINTER(C.c_float)]),
    "ts_smplx_create": (_i, [_vp, _i, _i, _i, _i, _fp, _fp, _fp, _fp, C.POINTER(C.c_int32), _fp, _fp, C.POINTER(C.c_int32), _i,
                             C.POINTER(C.c_int32), _i, C.POINTER(C.c_int32), _fp, _i, C.POINTER(_vp)]),
    "ts_smplx_destroy": (None, [_vp]),
    "ts_smplx_num_joints": (_i, [_vp]),
    "ts_smplx_forward": (_i, [_vp, _vp, _i, _vp, _i, _i, _i64, _vp, _vp, _vp]),
    "ts_eval_feat_stats": (_i, [_vp, _vp, _i64, _i, _vp, _vp]),
    "ts_eval_l1_total": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "ts_eval_body_loss": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "ts_eval_diversity": (_i, [_vp, _vp, _i, _i64, _vp, _vp]),
    "ts_prof_enable": (_i, [_vp, _i]),
    "ts_prof_read": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), _i]),
    "ts_prof_read_n": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), _i]),
}

_lib = None


def load():
    """dlopen the library and declare every prototype.  Raises if it is missing (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libtalkshow_hip: " + load().ts_last_error().decode())


def fptr(a):
    """host float32 numpy array -> POINTER(c_float) (the array must outlive the call)."""
    return a.ctypes.data_as(_fp) if a is not None else None


def dptr(t):
    """torch CUDA tensor -> raw device pointer (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device tensors handed to the C ABI must be contiguous HIP tensors"
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack_state_dict(sd):
    """{name: tensor/ndarray} -> (ts_tensor array, n, keepalive).  Non-float entries are passed with data=NULL."""
    items = list(sd.items())
    arr = (TsTensor * len(items))()
    keep = []
    for k, (name, v) in enumerate(items):
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        bname = name.encode()
        keep.append(bname)
        arr[k].name = bname
        arr[k].ndim = min(v.ndim, 4)
        for d in range(min(v.ndim, 4)):
            arr[k].shape[d] = v.shape[d]
        if v.dtype == np.float32 and v.ndim <= 4:
            v = np.ascontiguousarray(v)
            keep.append(v)
            arr[k].data = fptr(v)
        else:
            arr[k].data = None
    return arr, len(items), keep


def create_streams(n, device_index=None, cus=None):
    """n library-created HIP streams wrapped as torch ExternalStreams (created back to back -> distinct HW queues).

    cus=(first, count) restricts their kernels to that range of compute units (ts_stream_create_cus)."""
    ctx = context(device_index)
    out = []
    for _ in range(n):
        h = _vp()
        if cus is None:
            check(load().ts_stream_create(ctx, C.byref(h)))
        else:
            check(load().ts_stream_create_cus(ctx, int(cus[0]), int(cus[1]), C.byref(h)))
        out.append(torch.cuda.ExternalStream(h.value, device=torch.device("cuda", device_index if device_index is not None
                                                                              else torch.cuda.current_device())))
    return out


_contexts = {}


def context(device_index=None):
    """One ts_ctx per HIP device per process."""
    if not torch.cuda.is_available():
        raise RuntimeError("talkshow_amd needs a HIP device (MI355X / gfx950); torch.cuda.is_available() is False. "
                           "There is no CPU path.")
    if device_index is None:
        device_index = torch.cuda.current_device()
    if device_index not in _contexts:
        lib = load()
        h = _vp()
        check(lib.ts_ctx_create(int(device_index), C.byref(h)))
        _contexts[device_index] = h
    return _contexts[device_index]
