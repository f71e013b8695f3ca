#!/usr/bin/env python3
"""Where a conv_gemm_f32 tile's time goes: the TRACE build of the LDS-DMA ring engine (tile 41 = variant 31 + clock stamps) on the
transformer GEMM shapes.  Per workgroup: entry -> pointers ready -> first stage landed -> main loop done -> stores issued -> stores
acknowledged (100 MHz wall clock); per launch: when workgroups start / end relative to the first start, how many are in each phase
at a time, and the spread per compute unit."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from talkshow_amd import _lib  # noqa: E402

lib = _lib.load()
ctx = _lib.context(0)
SHAPES = [(64, 300, 768, 768, 1, "out-proj"), (64, 300, 768, 2304, 1, "qkv"), (64, 300, 3072, 768, 1, "ffn2"),
          (64, 128, 1024, 1024, 1, "exact 512 tiles K=1024"), (128, 128, 1024, 1024, 1, "exact 1024 tiles K=1024")]
if os.environ.get("TS_SHAPES"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["TS_SHAPES"].split(",")]
TICK = 0.01   # us per tick of the 100 MHz clock

TILE = int(os.environ.get("TS_TRACE_TILE", "41"))   # 141: the register epilogue
for (B, L, Cin, Cout, K, tag) in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, L, Cin, device="cuda", generator=g)
    npad = (Cout + 127) // 128 * 128
    w = torch.randn(npad, K * Cin, device="cuda", generator=g) / np.sqrt(K * Cin)
    b = torch.randn(npad, device="cuda", generator=g)
    out = torch.empty(B, L, Cout, device="cuda")
    ntile = ((B * L + 127) // 128) * (npad // 128)
    rec = torch.zeros(ntile, 8, dtype=torch.int64, device="cuda")
    ms = C.c_float()
    _lib.check(lib.ts_debug_conv_trace(_lib.dptr(rec), ntile))
    _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(x), B, L, Cin, _lib.dptr(w), _lib.dptr(b), Cout, K, TILE, 3, _lib.dptr(out),
                                      C.byref(ms), None))
    torch.cuda.synchronize()
    _lib.check(lib.ts_debug_conv_trace(None, 0))
    r = rec.cpu().numpy().astype(np.int64)
    assert (r[:, 7] == np.arange(ntile) + 1).all(), "missing records"
    t = (r[:, :6] - r[:, 0].min()) * TICK                  # us since the first workgroup's entry
    d = np.diff(t, axis=1)                                 # ptr setup, first data, main loop, epilogue issue, store ack
    life = t[:, 5] - t[:, 0]
    hw = r[:, 6] & 0xffffffff
    xcc = (r[:, 6] >> 32) & 0xf
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)   # (xcc, se, sh, cu)
    T = K * Cin // 32
    print(f"== {tag}: M={B * L} N={Cout} K={K * Cin}: {ntile} tiles, {T} stages; launch {ms.value * 1e3:.1f} us "
          f"({2.0 * B * L * Cout * K * Cin / ms.value / 1e9:.1f} TF), last end {t[:, 5].max():.1f} us, {len(np.unique(cu))} CUs seen")
    names = ["ptr setup", "first data", "main loop", "epilogue issue", "store ack"]
    for k, n in enumerate(names):
        print(f"   {n:15s} median {np.median(d[:, k]):7.2f}  p10 {np.percentile(d[:, k], 10):7.2f}  p90 {np.percentile(d[:, k], 90):7.2f}  max {d[:, k].max():7.2f} us")
    print(f"   {'lifetime':15s} median {np.median(life):7.2f}  p10 {np.percentile(life, 10):7.2f}  p90 {np.percentile(life, 90):7.2f}; "
          f"main loop per stage {np.median(d[:, 2]) / T * 1e3:.0f} ns (pipe-rate floor with 2 workgroups per CU: {2 * 128 * 128 * 32 * 2 / 256 / 2.4:.0f} ns at 2.4 GHz)")
    # rounds: workgroups ordered by entry time, in groups of 512
    order = np.argsort(t[:, 0])
    for k in range(0, ntile, 512):
        sel = order[k:k + 512]
        print(f"   wgs {k:5d}..{k + len(sel) - 1:5d} by entry: entry {t[sel, 0].min():7.1f}..{t[sel, 0].max():7.1f}  first MFMA {np.median(t[sel, 2]):7.1f}  "
              f"loop done {np.median(t[sel, 3]):7.1f}  end {np.median(t[sel, 5]):7.1f} (p90 {np.percentile(t[sel, 5], 90):7.1f})")
    # how much of the launch is some workgroup of a CU inside its main loop? (per CU: union of [first data, loop done] intervals)
    fr = []
    for c in np.unique(cu):
        sel = np.flatnonzero(cu == c)
        iv = sorted((t[i, 2], t[i, 3]) for i in sel)
        cov, end = 0.0, -1.0
        for a, e in iv:
            if e > end:
                cov += e - max(a, end)
                end = e
        two = sum(t[i, 3] - t[i, 2] for i in sel)
        fr.append((cov, two, len(sel)))
    fr = np.array(fr)
    total = t[:, 5].max()
    print(f"   per CU: some workgroup in its main loop {np.median(fr[:, 0]) / total * 100:.1f} % of the launch (min {fr[:, 0].min() / total * 100:.1f} %), "
          f"sum of main-loop time {np.median(fr[:, 1]) / total:.2f} x the launch, tiles per CU {fr[:, 2].min():.0f}..{fr[:, 2].max():.0f}")
