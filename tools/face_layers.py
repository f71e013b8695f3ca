#!/usr/bin/env python3
"""Per-launch time of the face generator's conv_gemm layers (BASELINE configs[2], batch 64): HIP-event pairs around every launch of one
instrumented pass (ts_prof + TS_PROF_LOG), grouped by (M, N, K)."""
import ctypes as C
import os
import re
import sys
import io
import collections

os.environ["TS_PROF_LOG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from talkshow_amd import _lib, synth  # noqa: E402

lib = _lib.load()
ctx = _lib.context(0)
m = bench.build_face(0)
B = int(os.environ.get("TS_B", "64"))
wav = torch.from_numpy(synth.wav16(3000, B, 160000)).cuda()
ids = torch.nn.functional.one_hot(torch.arange(B) % 4, 4).float().cuda()
m.run(wav, ids, 300)
torch.cuda.synchronize()
_lib.check(lib.ts_prof_enable(ctx, 1))
m.run(wav, ids, 300)
torch.cuda.synchronize()
ms, n, fl = (C.c_double * 4)(), (C.c_int64 * 4)(), (C.c_double * 4)()
_lib.check(lib.ts_prof_read_n(ctx, 4, ms, n, fl, 1))      # prints the per-launch log lines on stderr
print(f"conv total {ms[0]:.2f} ms in {n[0]} launches = {fl[0] / ms[0] / 1e9:.1f} TF; attention {ms[3]:.2f} ms; other {ms[2]:.2f} ms")
