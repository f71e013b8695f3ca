#!/usr/bin/env python3
"""The device audio front-end alone for profilers: `--passes` calls of 256 x 10 s @16 kHz waveforms -> resample 22 kHz -> MFCC(64)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from talkshow_amd import synth
from talkshow_amd.modules import MFCC
ap = argparse.ArgumentParser(); ap.add_argument("--passes", type=int, default=3); ap.add_argument("--clips", type=int, default=256)
a = ap.parse_args()
fe = MFCC(16000, 22000, 30)
wav = torch.from_numpy(synth.wav16(7000, a.clips, 160000)).cuda()
fe(wav); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.passes):
    fe(wav)
torch.cuda.synchronize()
print(f"front-end, {a.clips} clips: {(time.perf_counter() - t0) / a.passes * 1e3:.2f} ms per call")
