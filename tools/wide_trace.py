#!/usr/bin/env python3
"""In-kernel timeline of the wide chain kernel (skinny_wide.hip, TS_SKINNY_TRACE=1): wave 0 of every workgroup stamps the
100 MHz wall clock at entry, pointers ready, prologue issued, every stage barrier, loop end, epilogue operands back, end.
Prints per launch kind (workgroup count): dispatch spread, launch duration, period and the median phase lengths.

    TS_B=256 python tools/wide_trace.py
"""
import os, sys
os.environ["TS_SKINNY_TRACE"] = "1"
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from talkshow_amd import _lib, synth

lib = _lib.load()
w, _ = bench.build_models(0)
B, T = int(os.environ.get("TS_B", "256")), 300
dev = torch.device("cuda", 0)
mfcc = torch.from_numpy(synth.mfcc_features(1000, B, T)).to(dev)
ids = torch.from_numpy(synth.speaker_ids(B)).to(dev)
s = _lib.create_streams(1, 0)[0]
CAP, REC = 512 * 4096, 24
buf = (C.c_uint64 * (REC * CAP))()
with torch.cuda.stream(s):
    feat = w.audioencoder.forward_nlc(mfcc)
    for _ in range(2):
        w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)
    torch.cuda.synchronize()
    lib.ts_debug_skinny_trace(buf, CAP)          # reset
    w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)
    torch.cuda.synchronize()
n = lib.ts_debug_skinny_trace(buf, CAP)
r = np.frombuffer(buf, dtype=np.uint64)[: n * REC].reshape(n, REC)
r = r[r[:, 0] > 0]
wide = ((r[:, 5] >> np.uint64(62)) & np.uint64(1)).astype(bool)
print(f"{len(r)} workgroup records, {wide.sum()} from the wide kernel")
t_all = r[:, [0, 4]].astype(np.int64) * 0.01
order = np.argsort(t_all[:, 0], kind="stable")
r, wide, t_all = r[order], wide[order], t_all[order]
# group into launches (one stream: a new launch starts after every end stamp seen so far)
launches, start, cur_end = [], 0, t_all[0, 1]
for i in range(1, len(r)):
    if t_all[i, 0] > cur_end:
        launches.append((start, i)); start = i; cur_end = t_all[i, 1]
    else:
        cur_end = max(cur_end, t_all[i, 1])
launches.append((start, len(r)))
rows = []
for a, b in launches:
    x = r[a:b]
    if not wide[a]:
        rows.append((b - a, 0, t_all[a:b, 0].min(), t_all[a:b, 1].max(), None, None))
        continue
    t = x[:, :16].astype(np.int64) * 0.01
    Q = ((x[:, 5] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64)
    full = Q == Q.max()
    rows.append((b - a, 1, t[:, 0].min(), t[:, 4].max(), (t, full, Q), x[:, 15].astype(np.float64)))
R = rows
starts = np.array([x[2] for x in R]); ends = np.array([x[3] for x in R])
period = np.diff(starts)
kinds = {}
for i, x in enumerate(R):
    kinds.setdefault((x[0], x[1]), []).append(i)
print(" wgs wide launches | spread  duration  period | (wide, biggest-K workgroups) ptrs  prologue  ->st0 | stage deltas ... | loop-end  epi-ops  store | wg life median / max")
for (g, wd), idx in sorted(kinds.items()):
    dur = np.median([ends[i] - starts[i] for i in idx])
    per = np.median([period[i] for i in idx if i < len(period)]) if any(i < len(period) for i in idx) else 0
    line = f"{g:4d} {wd:4d} {len(idx):6d}   |"
    if not wd:
        print(line + f"   -    {dur:6.2f}   {per:6.2f}  |")
        continue
    sp, ph, life, lifemax, clk = [], [], [], [], []
    for i in idx:
        t, full, Q = R[i][4]
        sp.append(t[:, 0].max() - t[:, 0].min())
        tf = t[full]
        nst = int(Q.max()) // 4
        d = [tf[:, 1] - tf[:, 0], tf[:, 14] - tf[:, 1], tf[:, 6] - tf[:, 14]]
        d += [tf[:, 6 + k + 1] - tf[:, 6 + k] for k in range(min(nst, 8) - 1)]
        d += [tf[:, 2] - tf[:, 6 + min(nst, 8) - 1], tf[:, 3] - tf[:, 2], tf[:, 4] - tf[:, 3]]
        ph.append([np.median(v) for v in d])
        life.append(np.median(t[:, 4] - t[:, 0])); lifemax.append((t[:, 4] - t[:, 0]).max())
        clk.append(np.median(R[i][5][full] / np.maximum(tf[:, 4] - tf[:, 0], 0.01)) / 1e3)
    ph = np.median(np.array(ph), axis=0)
    print(line + f" {np.median(sp):5.2f}  {dur:6.2f}   {per:6.2f}  | " + " ".join(f"{v:5.2f}" for v in ph) + f" | {np.median(life):5.2f} / {np.median(lifemax):5.2f} | shader clock {np.median(clk):.2f} GHz")

# who are the stragglers?  workgroup life by problem index and by XCD (workgroup id % 8) for the most common wide launch kind
if os.environ.get("TS_TRACE_DETAIL"):
    common = max((k for k in kinds if k[1]), key=lambda k: len(kinds[k]))
    lifez, lifex, st0 = {}, {}, {}
    for i in kinds[common]:
        a, b = launches[i]
        x = r[a:b]
        t = x[:, :16].astype(np.int64) * 0.01
        zz = ((x[:, 5] >> np.uint64(48)) & np.uint64(0x3fff)).astype(np.int64)
        # records are sorted by entry time, not by workgroup id: recover the id from the record's position in the launch block is not possible here,
        # so the XCD view uses the entry order (dispatch order == id order within a launch)
        order_in = np.argsort(np.argsort(t[:, 0], kind="stable"), kind="stable")
        for k in range(len(x)):
            lifez.setdefault(int(zz[k]), []).append(t[k, 4] - t[k, 0])
            lifex.setdefault(int(order_in[k]) % 8, []).append(t[k, 4] - t[k, 0])
            st0.setdefault(int(zz[k]), []).append(t[k, 6] - t[k, 0])
    print(f"launch kind {common}: workgroup life by problem: " + "  ".join(f"z{z}: med {np.median(v):.2f} p95 {np.percentile(v, 95):.2f} max {np.max(v):.2f} (first data {np.median(st0[z]):.2f})" for z, v in sorted(lifez.items())))
    print("   by dispatch slot % 8: " + "  ".join(f"{k}: {np.median(v):.2f}/{np.percentile(v, 95):.2f}" for k, v in sorted(lifex.items())))
