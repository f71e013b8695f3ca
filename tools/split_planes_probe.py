#!/usr/bin/env python3
"""What would pre-split operands buy the split-bf16 GEMMs?  The face generator's big layer shapes on conv_gemm_f32 (tile 0),
conv_gemm_split x3 with fp32 operands split in the kernel (tile 22) and the same kernel fed plane images (tile 24):
[row][K / 32 chunks][hi: 32 bf16 | lo: 32 bf16], the same pitch as the fp32 tensor.  The tile-24 output must equal tile 22's bit for
bit (same bf16 values, same products, same order)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from talkshow_amd import _lib  # noqa: E402

lib = _lib.load()
ctx = _lib.context(0)


def planes(x):   # (..., K) fp32 -> plane image of the same shape, viewed as fp32
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    sh = x.shape[:-1] + (x.shape[-1] // 32, 1, 32)
    img = torch.cat([hi.reshape(sh), lo.reshape(sh)], dim=-2)          # (..., K/32, 2, 32) bf16
    return img.reshape(x.shape[:-1] + (x.shape[-1] * 2,)).contiguous().view(torch.float32)


B = 64
shapes = [(300, 768, 2304, 1), (300, 768, 768, 1), (300, 768, 3072, 1), (300, 3072, 768, 1), (3999, 512, 512, 3), (1999, 512, 512, 3)]
for (L, Cin, Cout, K) in shapes:
    x = torch.randn(B, L, Cin, device="cuda")
    npad = (Cout + 127) // 128 * 128
    w = torch.randn(npad, K * Cin, device="cuda") / np.sqrt(K * Cin)
    b = torch.randn(npad, device="cuda")
    xs, ws = planes(x), planes(w)
    assert xs.shape == x.shape and ws.shape == w.shape
    flops = 2.0 * B * L * Cout * K * Cin
    outs, row = {}, []
    for tile, xi, wi in ((0, x, w), (22, x, w), (24, xs, ws)):
        out = torch.empty(B, L, Cout, device="cuda")
        ms = C.c_float()
        _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(xi), B, L, Cin, _lib.dptr(wi), _lib.dptr(b), Cout, K, tile, 20,
                                          _lib.dptr(out), C.byref(ms), None))
        outs[tile] = out
        row.append(f"tile {tile}: {ms.value * 1e3:7.1f} us {flops / (ms.value * 1e-3) / 1e12:6.1f} TF")
    same = bool(torch.equal(outs[22], outs[24]))
    err = float((outs[22] - outs[0]).abs().max())
    print(f"B={B} L={L:5d} {Cin:4d}->{Cout:4d} k{K} | " + " | ".join(row) + f" | 24 == 22: {same} | x3 vs fp32 {err:.2e}", flush=True)
