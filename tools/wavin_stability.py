#!/usr/bin/env python3
"""Wav-in code stability: do the greedy codes change when the MFCC rows come from the device front-end (fp32 DFT-as-GEMM)
instead of the host twin (numpy, float64 where noted)?  32 synthetic 10 s clips; prints MFCC error, flipped codes, pose delta."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from talkshow_amd import _lib, synth, frontend as fe
from talkshow_amd.modules import MFCC

w, _ = bench.build_models(0)
clips = int(os.environ.get("TS_CLIPS", "32"))
wav = synth.wav16(7000, clips, 160000)
ids = torch.from_numpy(synth.speaker_ids(clips)).cuda()
dev = MFCC(16000, 22000, 30)(torch.from_numpy(wav).cuda())                 # (clips, 300, 64)
twin = np.stack([fe.mfcc(fe.resample_sinc_hann(x[None], 16000, 22000)[0], 22000, hop_length=734).T for x in wav]).astype(np.float32)
d = dev.cpu().numpy()
print("MFCC: max |dev - twin|", np.abs(d - twin).max(), "rms", np.sqrt(np.mean((d - twin) ** 2)), "scale", np.abs(twin).max(), np.abs(twin).mean())
c0, p0 = w.generate_batch(dev, ids, mode=_lib.TS_SAMPLE_GREEDY)
c1, p1 = w.generate_batch(torch.from_numpy(twin).cuda(), ids, mode=_lib.TS_SAMPLE_GREEDY)
c0, c1, p0, p1 = c0.cpu().numpy(), c1.cpu().numpy(), p0.cpu().numpy(), p1.cpu().numpy()
diff = (c0 != c1)
print(f"codes differing: {int(diff.sum())} / {diff.size}; clips with any flip: {int(diff.reshape(clips, -1).any(1).sum())} / {clips}; max pose delta {np.abs(p0 - p1).max():.3e}")
