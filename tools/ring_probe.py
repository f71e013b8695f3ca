#!/usr/bin/env python3
"""conv_gemm_f32: the LDS-DMA ring engine (conv_gemm_ring.hip, tile ids 31..) against the register-staged engine (tile 0 = the
production plan incl. bands, tile 1 = plain 128 x 128 grid) on the face generator's transformer GEMMs, the body's k3 conv layers
at 256 clips, and exact-round shapes (no tail) at three depths for a per-stage / per-tile fit.

Variants are interleaved inside one process, `ROUNDS` rounds each (guide rule 24); per (shape, variant): median and best mean
launch duration (HIP events on the launch stream, 20 launches per measurement) and TFLOP/s; the ring outputs are compared with
tile 1's bit for bit (same MFMA, same k order).  The variants that were measured with this tool and set aside (16-deep stages with 2-4
ring slots, 64 x 128 / 64 x 64 tiles, persistent workgroups, staggered start, whole-row stores through LDS) live in
tools/experiments/conv_ring/ with their numbers in profiles/r05_notes/."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from talkshow_amd import _lib  # noqa: E402

lib = _lib.load()
ctx = _lib.context(0)
ROUNDS = int(os.environ.get("TS_ROUNDS", "3"))
TILES = [int(t) for t in os.environ.get("TS_TILES", "0,1,31,39").split(",")]
NAMES = {0: "prod", 1: "reg128", 4: "reg64x128", 6: "reg160x128", 7: "reg96x128", 31: "ring 4w", 33: "ring 96x128", 39: "ring 8w"}
# (B, L, Cin, Cout, K, tag)
SHAPES = [
    (64, 300, 768, 2304, 1, "qkv"), (64, 300, 768, 768, 1, "out-proj"), (64, 300, 768, 3072, 1, "ffn1"), (64, 300, 3072, 768, 1, "ffn2"),
    (256, 75, 1024, 1024, 3, "vq k3 1024"), (256, 150, 512, 512, 3, "vq k3 512"), (256, 300, 256, 256, 3, "vq k3 256"),
    (128, 128, 1024, 1024, 1, "exact 1024 tiles K=1024"), (128, 128, 2048, 1024, 1, "exact K=2048"), (128, 128, 4096, 1024, 1, "exact K=4096"),
    (64, 128, 1024, 1024, 1, "exact 512 tiles K=1024"), (64, 128, 4096, 1024, 1, "exact 512 tiles K=4096"),
    (32, 128, 1024, 1024, 1, "256 tiles (1 per CU) K=1024"), (32, 128, 4096, 1024, 1, "256 tiles K=4096"),
    (16, 128, 4096, 1024, 1, "128 tiles K=4096"),
]
if os.environ.get("TS_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["TS_SHAPES"].split(",")]

for (B, L, Cin, Cout, K, tag) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(B + L + Cin)
    x = torch.randn(B, L, Cin, device="cuda", generator=g)
    npad = (Cout + 127) // 128 * 128
    w = torch.randn(npad, K * Cin, device="cuda", generator=g) / np.sqrt(K * Cin)
    b = torch.randn(npad, device="cuda", generator=g)
    flops = 2.0 * B * L * Cout * K * Cin
    outs, times = {}, {t: [] for t in TILES}
    for r in range(ROUNDS):
        for tile in TILES:
            out = torch.full((B, L, Cout), float("nan"), device="cuda")
            ms = C.c_float()
            _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(x), B, L, Cin, _lib.dptr(w), _lib.dptr(b), Cout, K, tile, 20,
                                              _lib.dptr(out), C.byref(ms), None))
            times[tile].append(ms.value)
            if r == 0:
                outs[tile] = out
    ref = outs.get(1, outs[TILES[0]])
    row = []
    for tile in TILES:
        same = torch.equal(outs[tile], ref)
        err = float((outs[tile] - ref).abs().max()) if not same else 0.0
        med, best = float(np.median(times[tile])), min(times[tile])
        row.append(f"{NAMES.get(tile, tile)}: {med * 1e3:7.1f} us {flops / (med * 1e-3) / 1e12:6.1f} TF (best {flops / (best * 1e-3) / 1e12:6.1f})"
                   + ("" if same else f" DIFFERS {err:.2e}"))
    print(f"{tag:26s} M={B * L:6d} N={Cout:4d} K={K * Cin:4d} | " + " | ".join(row), flush=True)
