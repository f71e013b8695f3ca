#!/usr/bin/env python3
"""Per-launch cost of a dependent skinny_gemm chain (hipGraph replay) with ablations; see ts_debug_skinny_chain."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from talkshow_amd import _lib  # noqa: E402

lib = _lib.load()
ctx = _lib.context(0)
names = {0: "full", 1: "no-W-loads", 2: "no-A-loads", 3: "no-loads", 4: "no-mfma", 7: "no-loads-no-mfma", 8: "no-transc", 15: "empty"}
for (M, K, gate) in [(32, 256, 0), (32, 256, 1), (32, 512, 1), (32, 1024, 1), (64, 512, 0), (32, 1536, 1)]:
    row = []
    for dbg in (0, 1, 2, 3, 4, 7, 8, 15):
        us = C.c_float()
        _lib.check(lib.ts_debug_skinny_chain(ctx, M, K, gate, 400, dbg, C.byref(us)))
        row.append(f"{names[dbg]} {us.value:5.2f}")
    print(f"M={M} K={K:4d} N={(2 if gate else 1) * K:4d} {'gate' if gate else 'lin '} | " + " | ".join(row), flush=True)
