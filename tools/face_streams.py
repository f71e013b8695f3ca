#!/usr/bin/env python3
"""Face generator (BASELINE configs[2]): batches of 64 clips back to back on ONE stream against the same batches on TWO / THREE streams
(each stream its own work set of the one weight copy: `ts_face_generate` keeps scratch per stream), and the batch of 64 cut into two
half batches on two streams.  A GEMM's partly filled last round and the launch-to-launch gap of one stream's dependent launches are
filled by the other stream's workgroups — the face generator's analogue of the body path's passes in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from talkshow_amd import _lib, synth

m = bench.build_face(0)
T, N = 300, 160000
def data(B, seed):
    return (torch.from_numpy(synth.wav16(seed, B, N)).cuda(), torch.nn.functional.one_hot(torch.arange(B) % 4, 4).float().cuda())
def measure(B, nstreams, rounds=4):
    streams = _lib.create_streams(nstreams, 0)
    sets = [data(B, 3000 + i) for i in range(nstreams)]
    outs = [None] * nstreams
    def go():
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs[i] = m.run(sets[i][0], sets[i][1], T)
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / rounds
    return dt, outs
ref_wav, ref_ids = data(64, 3000)
ref = m.run(ref_wav, ref_ids, T); torch.cuda.synchronize()
CASES = ((64, 1), (64, 2), (64, 3), (32, 2), (32, 4), (64, 1)) if not os.environ.get("TS_CASES") else tuple(tuple(int(v) for v in c.split("x")) for c in os.environ["TS_CASES"].split(","))
for B, ns in CASES:
    dt, outs = measure(B, ns)
    clips = B * ns
    same = torch.equal(outs[0], ref[:B]) if B <= 64 else None
    print(f"batch {B:3d} x {ns} stream(s): {dt * 1e3:7.2f} ms per round = {dt * 1e3 * 64 / clips:6.2f} ms per 64 clips = {clips * T / dt / 1e3:7.1f} k frames/s; "
          f"stream 0's rows == the one-stream batch: {same}", flush=True)
