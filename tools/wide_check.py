#!/usr/bin/env python3
"""Wide (64 x 64, LDS-staged) chain kernel vs the split-K kernels: greedy codes of the same clips must be BIT-IDENTICAL,
and the pass time is printed for both.  Each arm runs in a child process (the knobs are read once per process).

    python tools/wide_check.py [--batches 256,192,96,64] [--passes 3]
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(batch, passes):
    import torch
    import bench
    from talkshow_amd import _lib, synth
    w, _ = bench.build_models(0)
    B, T = batch, 300
    mf = torch.from_numpy(synth.mfcc_features(1, B, T)).cuda()
    ids = torch.from_numpy(synth.speaker_ids(B)).cuda()
    feat = w.audioencoder.forward_nlc(mf)
    codes = w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)[0]   # graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(passes):
        codes = w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)[0]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / passes * 1e3
    c = codes.cpu().numpy()
    print(json.dumps({"batch": B, "ms_per_pass": ms, "sha": hashlib.sha256(c.tobytes()).hexdigest(), "head": c[0, :3].tolist()}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="256,192,96,64")
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--child", type=int, default=0)
    a = ap.parse_args()
    if a.child:
        child(a.child, a.passes)
        sys.exit(0)
    bad = 0
    for b in [int(x) for x in a.batches.split(",")]:
        res = {}
        for arm, env in (("splitk", {"TS_SKINNY_WIDE_MIN": "0"}), ("wide", {"TS_SKINNY_WIDE_MIN": "1"}), ("auto", {})):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(b), "--passes", str(a.passes)],
                               env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                print(arm, "FAILED", r.stdout[-500:], r.stderr[-1500:])
                bad += 1
                continue
            res[arm] = json.loads(r.stdout.strip().splitlines()[-1])
        same = len({v["sha"] for v in res.values()}) == 1
        bad += 0 if same else 1
        print(f"batch {b}: " + "  ".join(f"{k} {v['ms_per_pass']:.2f} ms" for k, v in res.items()) + ("  codes identical" if same else "  CODES DIFFER"),
              flush=True)
    sys.exit(1 if bad else 0)
