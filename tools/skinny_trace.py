#!/usr/bin/env python3
"""Where do the microseconds of one PixelCNN chain stage go?  (TS_SKINNY_TRACE=1 instrumentation build path.)

Every workgroup of every chain launch stamps the 100 MHz device wall clock at: kernel entry, descriptor loaded, MFMAs done
(all operand loads back), partial sums reduced (after the barrier), end (output stored).  Records are grouped into
launches (one stream: launches do not overlap), giving per launch: dispatch spread (first -> last workgroup entry),
duration (first entry -> last end) and the gap to the next launch."""
import os, sys
os.environ["TS_SKINNY_TRACE"] = "1"
import ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from talkshow_amd import _lib, synth

lib = _lib.load()
w, _ = bench.build_models(0)
B, T, H = int(os.environ.get("TS_B", "32")), 300, 75
dev = torch.device("cuda", 0)
mfcc = torch.from_numpy(synth.mfcc_features(1000, B, T)).to(dev)
ids = torch.from_numpy(synth.speaker_ids(B)).to(dev)
s = _lib.create_streams(1, 0)[0]
CAP = 512 * 4096
REC = 24
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
n = len(r)
meta = r[:, 5]
t = r[:, :5].astype(np.int64) * 0.01               # microseconds
dead = (meta >> np.uint64(63)).astype(bool)
nwg = (meta & np.uint64(0xffffff)).astype(np.int64)
order = np.argsort(t[:, 0], kind="stable")
t, dead, nwg = t[order], dead[order], nwg[order]
# group into launches: a new launch starts when an entry stamp is later than every end stamp seen so far
launches, start, cur_end = [], 0, t[0, 4]
for i in range(1, n):
    if t[i, 0] > cur_end:
        launches.append((start, i)); start = i; cur_end = t[i, 4]
    else:
        cur_end = max(cur_end, t[i, 4])
launches.append((start, n))
print(f"{n} workgroup records in {len(launches)} launches")
rows = []
for a, b in launches:
    tt, dd = t[a:b], dead[a:b]
    live = tt[~dd] if (~dd).any() else tt
    rows.append((b - a, int(dd.sum()), tt[:, 0].max() - tt[:, 0].min(), tt[:, 4].max() - tt[:, 0].min(),
                 np.median(live[:, 4] - live[:, 0]), (live[:, 4] - live[:, 0]).max(), tt[:, 0].min(), tt[:, 4].max(),
                 np.median(live[:, 1] - live[:, 0]), np.median(live[:, 2] - live[:, 1]), np.median(live[:, 3] - live[:, 2]), np.median(live[:, 4] - live[:, 3])))
R = np.array(rows)
gap = R[1:, 6] - R[:-1, 7]
period = np.diff(R[:, 6])
print(f"median period {np.median(period):.2f} us, mean {period.mean():.2f} us; gap last-end -> next first-entry median {np.median(gap):.2f} mean {gap.mean():.2f}")
print("  wgs  dead  launches | dispatch spread | duration | wg median / max in-kernel | desc  loads+mfma  reduce  store | period")
for g in np.unique(R[:, 0]):
    m = R[:, 0] == g
    mm = m[:-1]
    x = R[m]
    print(f"  {int(g):4d} {int(np.median(x[:,1])):4d} {m.sum():6d}    | {np.median(x[:,2]):6.2f}          | {np.median(x[:,3]):6.2f}   | {np.median(x[:,4]):5.2f} / {np.median(x[:,5]):5.2f}"
          f"             | {np.median(x[:,8]):.2f}  {np.median(x[:,9]):.2f}  {np.median(x[:,10]):.2f}  {np.median(x[:,11]):.2f} | {np.median(period[mm]) if mm.any() else 0:.2f}")

# per-wave view (workgroups of 8 waves): offsets from the workgroup's entry stamp
wt = r[order][:, 6:22].astype(np.int64) * 0.01
ent = t[:, 0:1]
ld = wt[:, 0::2] - ent
mf = wt[:, 1::2] - ent
cntv = ((meta[order] >> np.uint64(24)) & np.uint64(0xff)).astype(np.int64)
zv = (meta[order] >> np.uint64(48)).astype(np.int64) & 0x7fff
for c in np.unique(cntv):
    m = (cntv == c) & (wt[:, 14] > 0)
    if not m.any():
        continue
    print(f"cnt={c} ({m.sum()} workgroups): per wave median [operands back | MFMAs done] us after entry")
    print("   loads:", " ".join(f"{np.median(ld[m, k]):5.2f}" for k in range(8)))
    print("   mfma :", " ".join(f"{np.median(mf[m, k]):5.2f}" for k in range(8)))
