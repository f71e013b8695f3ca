"""The thread contract of include/talkshow_hip.h, exercised: ONE HOST THREAD PER STREAM AT A TIME —
several host threads may call into the SAME handles concurrently as long as each drives its own stream.

Three host threads, one library stream each, run the body wrapper (greedy and Philox decode, VQ encode) and the face generator on
different clips at the same time, several rounds, started together behind a barrier; every result must equal, bit for bit, the same
call made serially on the default stream.  (ctypes releases the GIL for the duration of a C call, so the threads really are inside
the library together: weight handles shared, one scratch arena + graph cache per stream, per-stream maps behind a mutex.)
"""
import threading

import numpy as np
import pytest
import torch

from talkshow_amd import synth

pytestmark = pytest.mark.gpu


def test_three_host_threads_one_stream_each():
    import bench
    from talkshow_amd import _lib
    lib = _lib.load()
    w, _ = bench.build_models(0)
    face = bench.build_face(0)
    NT, ROUNDS = 3, 4
    T = 300
    jobs = []
    for t in range(NT):
        B = (32, 17, 64)[t]                                       # different shapes per thread: different graphs, tiles, arenas
        jobs.append(dict(B=B, mf=torch.from_numpy(synth.mfcc_features(9000 + t, B, T)).cuda(),
                         ids=torch.from_numpy(synth.speaker_ids(B)).cuda(),
                         gt=torch.from_numpy(synth.gt_poses(9100 + t, B, T)).cuda(),
                         wav=torch.from_numpy(synth.wav16(9200 + t, 2 + t, 32000)).cuda(),
                         fid=torch.eye(4, device="cuda")[torch.arange(2 + t) % 4].contiguous()))

    def work(j, r):
        """everything one thread does in round r (on the CURRENT stream of the calling thread)"""
        B = j["B"]
        c_g, p_g = w.generate_batch(j["mf"], j["ids"], mode=_lib.TS_SAMPLE_GREEDY)
        c_s, p_s = w.generate_batch(j["mf"], j["ids"], mode=_lib.TS_SAMPLE_PHILOX, seed=77 + r, clip_index0=5 * r)
        enc = torch.empty((B, T // 4, 2), dtype=torch.int64, device="cuda")
        _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(j["gt"]), B, T, _lib.dptr(enc), None, _lib.stream_ptr()))
        f = face.run(j["wav"], j["fid"], 60)
        return [c_g, p_g, c_s, p_s, enc, f]

    # serial reference on the default stream
    want = [[[x.cpu().numpy() for x in work(j, r)] for r in range(ROUNDS)] for j in jobs]
    torch.cuda.synchronize()
    streams = _lib.create_streams(NT, 0)
    gate = threading.Barrier(NT)
    got, errors = [[None] * ROUNDS for _ in range(NT)], []

    def thread(t):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(streams[t]):
                for r in range(ROUNDS):
                    gate.wait(timeout=120)                          # all three enter the library together, every round
                    out = work(jobs[t], r)
                    streams[t].synchronize()
                    got[t][r] = [x.cpu().numpy() for x in out]
        except Exception as e:                                      # noqa: BLE001
            errors.append((t, repr(e)))
            gate.abort()

    th = [threading.Thread(target=thread, args=(t,)) for t in range(NT)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=600)
    assert not errors, errors
    assert not any(x.is_alive() for x in th)
    names = ("greedy codes", "greedy poses", "philox codes", "philox poses", "vq-encode codes", "face rows")
    for t in range(NT):
        for r in range(ROUNDS):
            for name, a, b in zip(names, got[t][r], want[t][r]):
                assert np.array_equal(a, b), f"thread {t} round {r}: {name} differ from the serial run"
    # the rounds really drew different random streams (the seeds reached the sampler of the right call)
    assert not np.array_equal(want[0][0][2], want[0][1][2])
    for s in streams:
        _lib.check(lib.ts_stream_destroy(_lib.context(0), s.cuda_stream))
