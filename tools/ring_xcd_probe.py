#!/usr/bin/env python3
"""conv_gemm_f32's ring engine with the tiles dealt to the XCDs in operand-sharing blocks (tile ids 35 / 36) against its plain grid
(39 / 33) on the face generator's layers at BASELINE configs[2] (batch 64, 10 s): the six strided feature convolutions — M up to
1 023 936 rows, operands far beyond L2 and the Infinity Cache — and the four GEMMs of an encoder block.  Variants interleaved in one process,
ROUNDS x 20 launches (HIP events on the launch stream); outputs compared bit for bit with tile 39's."""
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
TILES = [int(t) for t in os.environ.get("TS_TILES", "39,35,33,36").split(",")]
NAMES = {39: "plain 128", 35: "dealt 128", 33: "plain 96", 36: "dealt 96", 37: "bands+dealt", 1: "reg 128", 0: "prod"}
# (B, Lin, Cin, Cout, K, stride, tag); stride 0 = the padded stride-1 entry
SHAPES = [
    (64, 31999, 512, 512, 3, 2, "feat conv1"), (64, 15999, 512, 512, 3, 2, "feat conv2"), (64, 7999, 512, 512, 3, 2, "feat conv3"),
    (64, 3999, 512, 512, 3, 2, "feat conv4"), (64, 1999, 512, 512, 2, 2, "feat conv5"), (64, 999, 512, 512, 2, 2, "feat conv6"),
    (64, 300, 768, 2304, 1, 0, "qkv"), (64, 300, 768, 768, 1, 0, "out-proj"), (64, 300, 768, 3072, 1, 0, "ffn1"), (64, 300, 3072, 768, 1, 0, "ffn2"),
    (256, 75, 1024, 1024, 3, 0, "vq k3 1024"), (256, 300, 256, 256, 3, 0, "vq k3 256"),
]
if os.environ.get("TS_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["TS_SHAPES"].split(",")]

for (B, L, Cin, Cout, K, stride, tag) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(B + L + Cin)
    x = torch.randn(B, L, Cin, device="cuda", generator=g)
    npad = (Cout + 127) // 128 * 128
    w = torch.randn(npad, K * Cin, device="cuda", generator=g) / np.sqrt(K * Cin)
    b = torch.randn(npad, device="cuda", generator=g)
    Lout = (L - K) // stride + 1 if stride else L
    flops = 2.0 * B * Lout * Cout * K * Cin
    outs, times = {}, {t: [] for t in TILES}
    for r in range(ROUNDS):
        for tile in TILES:
            out = outs.get(tile)
            if out is None:
                out = outs[tile] = torch.full((B, Lout, Cout), float("nan"), device="cuda")
            ms = C.c_float()
            if stride:
                _lib.check(lib.ts_op_conv1d_strided_timed(ctx, _lib.dptr(x), B, L, Cin, _lib.dptr(w), _lib.dptr(b), Cout, K, stride, tile, 20,
                                                          _lib.dptr(out), C.byref(ms), None))
            else:
                _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(x), B, L, Cin, _lib.dptr(w), _lib.dptr(b), Cout, K, tile, 20,
                                                  _lib.dptr(out), C.byref(ms), None))
            times[tile].append(ms.value)
    ref = outs[TILES[0]]
    row = []
    for tile in TILES:
        same = torch.equal(outs[tile], ref) and not bool(torch.isnan(outs[tile]).any())
        med, best = float(np.median(times[tile])), min(times[tile])
        row.append(f"{NAMES.get(tile, tile)}: {med * 1e3:8.1f} us {flops / (med * 1e-3) / 1e12:6.1f} TF (best {flops / (best * 1e-3) / 1e12:6.1f})"
                   + ("" if same else " DIFFERS"))
    print(f"{tag:12s} M={B * Lout:7d} N={Cout:4d} K={K * Cin:4d} | " + " | ".join(row), flush=True)
    del outs, x
    torch.cuda.empty_cache()
