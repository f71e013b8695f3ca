#!/usr/bin/env python3
"""The face generator alone for profilers: `--passes` calls of BASELINE configs[2] (batch 64 x 10 s @16 kHz -> (64, 300, 103)), fp32."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from talkshow_amd import synth
ap = argparse.ArgumentParser(); ap.add_argument("--passes", type=int, default=3); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--arith", type=int, default=0, help="0 fp32 (default), 3 / 6: the opt-in split-bf16 plan")
a = ap.parse_args()
m = bench.build_face(0)
if a.arith:
    m.set_arith(a.arith)
wav = torch.from_numpy(synth.wav16(3000, a.batch, 160000)).cuda()
ids = torch.nn.functional.one_hot(torch.arange(a.batch) % 4, 4).float().cuda()
m.run(wav, ids, 300); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.passes):
    m.run(wav, ids, 300)
torch.cuda.synchronize()
print(f"face batch {a.batch}: {(time.perf_counter() - t0) / a.passes * 1e3:.2f} ms per call")
