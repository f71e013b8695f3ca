#!/usr/bin/env python3
"""conv_taps48.hip alone at the face generator's shape (BASELINE configs[2]: 64 clips x 300 frames, 16 groups of 48 channels, 128 taps):
mean launch duration over 20 launches, 3 rounds, and the useful TFLOP/s (2 M 768 48 128 flops)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from talkshow_amd import _lib  # noqa: E402

lib = _lib.load()
ctx = _lib.context(0)
for (B, T, G, ntap) in ((64, 300, 16, 128), (128, 300, 16, 128), (8, 300, 16, 128)):
    g = torch.Generator(device="cuda").manual_seed(B + T)
    x = torch.randn(B, T, G * 48, device="cuda", generator=g)
    w = torch.randn(G, 48, ntap * 48, device="cuda", generator=g) / np.sqrt(ntap * 48)
    b = torch.randn(G * 48, device="cuda", generator=g)
    out = torch.empty_like(x)
    ts = []
    for r in range(3):
        ms = C.c_float()
        _lib.check(lib.ts_op_conv_taps48_timed(ctx, _lib.dptr(x), B, T, G, ntap, _lib.dptr(w), _lib.dptr(b), _lib.dptr(x), 20, _lib.dptr(out),
                                               C.byref(ms), None))
        ts.append(ms.value)
    fl = 2.0 * B * T * G * 48 * 48 * ntap
    print(f"B={B} T={T} G={G} taps={ntap}: {np.median(ts) * 1e3:8.1f} us  {fl / (np.median(ts) * 1e-3) / 1e12:6.1f} TF useful (best {fl / (min(ts) * 1e-3) / 1e12:6.1f})", flush=True)
