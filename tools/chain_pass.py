#!/usr/bin/env python3
"""One operating point of the PixelCNN chain for profilers: audio encoder + `passes` greedy generate passes over `--batch`
clips (10 s, 75 code rows) on one stream, nothing else.  Used under rocprofv3 by tools/profile_chain.sh."""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402

import bench  # noqa: E402
from talkshow_amd import _lib, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--passes", type=int, default=2)
ap.add_argument("--convs", action="store_true", help="also run the VQ encode + decode conv stacks of the pass")
a = ap.parse_args()
w, _ = bench.build_models(0)
B, T = a.batch, 300
mf = torch.from_numpy(synth.mfcc_features(1, B, T)).cuda()
ids = torch.from_numpy(synth.speaker_ids(B)).cuda()
gt = torch.from_numpy(synth.gt_poses(2, B, T)).cuda()
codes = torch.empty((B, T // 4, 2), dtype=torch.int64, device="cuda")
lib = _lib.load()
feat = w.audioencoder.forward_nlc(mf)
w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)        # graph capture
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.passes):
    if a.convs:
        _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(gt), B, T, _lib.dptr(codes), None,
                                        _lib.stream_ptr()))
        w.generate_batch(mf, ids, mode=_lib.TS_SAMPLE_GREEDY)
    else:
        w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)
torch.cuda.synchronize()
print(f"batch {B}: {(time.perf_counter() - t0) / a.passes * 1e3:.2f} ms per pass")
