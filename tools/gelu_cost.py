#!/usr/bin/env python3
"""What does the GELU (erf) epilogue cost on the ring engine?  The same GEMM shape through ts_op_conv1d_timed (LeakyReLU epilogue) and
ts_op_conv1d_strided_timed with K = 1, stride = 1 (GELU epilogue), same tile plan; FFN1 of a face batch of 64 and a feature-convolution shape."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from talkshow_amd import _lib  # noqa: E402

lib, ctx = _lib.load(), _lib.context(0)
rng = np.random.default_rng(1)
for name, B, L, Cin, Cout in (("ffn1", 64, 300, 768, 3072), ("qkv", 64, 300, 768, 2304), ("feat-like", 64, 4000, 512, 512)):
    x = torch.from_numpy(rng.standard_normal((B, L, Cin)).astype(np.float32)).cuda()
    npad = (Cout + 127) // 128 * 128
    w = torch.from_numpy((rng.standard_normal((npad, Cin)) / np.sqrt(Cin)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal(npad).astype(np.float32)).cuda()
    out = torch.empty((B, L, Cout), device="cuda")
    res = {}
    for rep in range(2):
        for tile in (37, 35):
            ms = C.c_float()
            _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(x), B, L, Cin, _lib.dptr(w), _lib.dptr(b), Cout, 1, tile, 20, _lib.dptr(out), C.byref(ms), None))
            res[("leaky", tile)] = ms.value * 1e3
            _lib.check(lib.ts_op_conv1d_strided_timed(ctx, _lib.dptr(x), B, L, Cin, _lib.dptr(w), _lib.dptr(b), Cout, 1, 1, tile, 20, _lib.dptr(out), C.byref(ms), None))
            res[("gelu", tile)] = ms.value * 1e3
    print(name, {f"{a}@{t}": round(v, 1) for (a, t), v in res.items()}, "GELU - leaky (us):", {t: round(res[("gelu", t)] - res[("leaky", t)], 1) for t in (37, 35)}, flush=True)
