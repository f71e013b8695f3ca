#!/usr/bin/env python3
"""Tile sweep of conv_gemm_f32 on the layer shapes of the VQ encoder/decoder at the BASELINE batch (B=32).
Prints mean launch duration (HIP events on the launch stream) and achieved TFLOP/s per (shape, tile)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from talkshow_amd import _lib  # noqa: E402

lib = _lib.load()
ctx = _lib.context(0)
B = int(os.environ.get("TS_B", "32"))
shapes = [(75, 1024, 1024, 3), (150, 512, 512, 3), (300, 256, 256, 3), (300, 64, 64, 3), (150, 128, 128, 3), (75, 256, 256, 3),
          (75 * 32 // B, 64, 1024, 1), (128, 4096, 4096, 1), (600, 768, 768, 1), (600, 768, 3072, 1), (600, 3072, 768, 1),
          (15999, 512, 512, 3)]
if os.environ.get("TS_TUNE_FEW"):
    shapes = [(75, 1024, 1024, 3), (150, 512, 512, 3), (300, 256, 256, 3), (128, 4096, 4096, 1), (600, 768, 3072, 1)]
names = {0: "auto", 1: "128x128", 2: "64x64", 3: "128x64", 4: "64x128", 5: "64x64k64", 6: "160x128", 7: "96x128", 8: "256x128",
         9: "128x256"}
for (L, Cin, Cout, K) in shapes:
    x = torch.randn(B, L, Cin, device="cuda")
    npad = (Cout + 127) // 128 * 128
    w = torch.randn(npad, K * Cin, device="cuda") / np.sqrt(K * Cin)
    b = torch.randn(npad, device="cuda")
    out = torch.empty(B, L, Cout, device="cuda")
    flops = 2.0 * B * L * Cout * K * Cin
    row = []
    for tile in [int(t) for t in os.environ.get("TS_TILES", "0,1,2,3,4,6,7,8,9").split(",")]:
        ms = C.c_float()
        _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(x), B, L, Cin, _lib.dptr(w), _lib.dptr(b), Cout, K, tile, 20,
                                          _lib.dptr(out), C.byref(ms), None))
        row.append(f"{names[tile]}: {ms.value * 1e3:7.1f} us {flops / (ms.value * 1e-3) / 1e12:6.1f} TF")
    print(f"B={B} L={L:4d} {Cin:4d}->{Cout:4d} k{K} | " + " | ".join(row), flush=True)
