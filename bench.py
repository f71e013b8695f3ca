#!/usr/bin/env python3
"""Throughput bench of the speech -> SMPL-X body hot path on MI355X (contract: see the task statement / DESIGN.md §4).

One step = BASELINE.json configs[1]: one batch of 32 synthetic 10 s clips through
    VQ-VAE encode of 300 GT frames (body + hand)  ->  audio encoder -> PixelCNN greedy decode (75 x 2 codes)
    ->  VQ decode to (32, 300, 129) SMPL-X pose parameters,
inputs resident in HBM before the timed region, fp32, seeded random-init weights of the reference architecture.
value = generated frames / s over all ranks (32 * 300 frames per step per rank).

    python bench.py                          # N=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Clips are independent, so ranks shard them with no data-path collective ("weak" scaling: 32 clips per rank per
step); the only exchange is one all-gather of the generated pose sequences after the timed region's last step
(inside the timed region, so it is paid for).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# independent batches are pipelined over several HIP streams; give each its own hardware queue
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FRAMES_PER_CLIP = 300
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 2.4 GHz
PEAK_HBM_GBS = 8000.0


def build_models(device_index, seed=0):
    import tempfile
    from nets.init_model import init_model
    from talkshow_amd import synth
    from talkshow_amd.config import load_JsonConfig
    cfg = load_JsonConfig(os.path.join(REPO, "config", "body_pixel.json"))
    tmp = tempfile.mkdtemp(prefix="ts_bench_")
    sd_body = synth.vqvae_state_dict(seed=seed, in_dim=39)
    sd_hand = synth.vqvae_state_dict(seed=seed, in_dim=90, salt=1)
    cfg.Model.vq_path = os.path.join(tmp, "vq.pth")
    torch.save({"generator": {"g_body": synth.to_torch(sd_body), "g_hand": synth.to_torch(sd_hand)}}, cfg.Model.vq_path)
    args = argparse.Namespace(gpu=device_index, infer=True)
    sd_pix, sd_aud = synth.pixelcnn_state_dict(seed=seed), synth.audioencoder_state_dict(seed=seed)
    w = init_model("s2g_body_pixel", args, cfg)
    w.load_state_dict({"generator": synth.to_torch(sd_pix), "audioencoder": synth.to_torch(sd_aud)})
    return w, dict(audio=sd_aud, pix=sd_pix, body=sd_body, hand=sd_hand)


def cpu_baseline(sds, seed):
    """The oracle (= the reference's algorithm incl. its full-grid recompute per position) on the host cores, one clip."""
    from oracle import talkshow_oracle as O
    from talkshow_amd import synth
    mf, ids = synth.mfcc_features(seed, 1, FRAMES_PER_CLIP), synth.speaker_ids(1)
    gt = synth.gt_poses(seed, 1, FRAMES_PER_CLIP)
    t0 = time.perf_counter()
    O.vqvae_encode(gt[..., :39], sds["body"])
    O.vqvae_encode(gt[..., 39:], sds["hand"])
    O.body_pixel_infer(mf, ids, sds["audio"], sds["pix"], sds["body"], sds["hand"])
    dt = time.perf_counter() - t0
    return {"value": FRAMES_PER_CLIP / dt, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"1 clip (10 s, 300 frames), VQ encode + greedy full-grid PixelCNN + VQ decode, numpy/BLAS fp32, {dt:.1f} s"}



def face_block(local):
    """BASELINE configs[2] as extra information: face generator, batch 64 x 10 s @16 kHz -> (64,300,103), fp32."""
    from talkshow_amd import _lib, synth
    from talkshow_amd.modules import FaceGenerator
    lib = _lib.load()
    ctx = _lib.context(local)
    m = FaceGenerator().cuda()
    m.load_state_dict(synth.to_torch(synth.face_state_dict(seed=0)))
    B, N, T = 64, 160000, 300
    wav = torch.from_numpy(synth.wav16(3000, B, N)).cuda()
    ids = torch.nn.functional.one_hot(torch.arange(B) % 4, 4).float().cuda()
    m.run(wav, ids, T)
    torch.cuda.synchronize()
    n0, f0 = (C.c_int64 * 3)(), (C.c_double * 3)()
    t0 = time.perf_counter()
    K = 3
    for _ in range(K):
        m.run(wav, ids, T)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    # instrumented pass: HIP event pairs around every launch
    _lib.check(lib.ts_prof_enable(ctx, 1))
    m.run(wav, ids, T)
    torch.cuda.synchronize()
    msf, nf, flf = (C.c_double * 3)(), (C.c_int64 * 3)(), (C.c_double * 3)()
    _lib.check(lib.ts_prof_read(ctx, msf, nf, flf, 1))
    _lib.check(lib.ts_prof_enable(ctx, 0))
    ach = flf[0] / (msf[0] * 1e-3) / 1e12
    return {"workload": "BASELINE configs[2]: face generator, batch=64 x 10 s @16 kHz, 103 params @30 fps",
            "frames_per_s": B * T / dt, "ms_per_batch": dt * 1e3,
            "conv_gemm_f32": {"launches": nf[0], "ms": msf[0], "achieved_TFLOPs": ach, "frac_of_fp32_mfma_peak": ach / PEAK_FP32_MFMA_TFLOPS},
            "other_kernels_ms": msf[2]}


def diversity_block(w, _lib, mfcc1):
    """BASELINE configs[3] as extra information: num_samples=12 stochastic decodes of ONE 10 s clip (Philox seed 2024)."""
    B = 12
    mf = mfcc1[:1].repeat(B, 1, 1).contiguous()
    ids = torch.zeros(B, dtype=torch.int64, device=mf.device)
    codes, _ = w.generate_batch(mf, ids, mode=_lib.TS_SAMPLE_PHILOX, seed=2024)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 3
    for _ in range(K):
        codes, poses = w.generate_batch(mf, ids, mode=_lib.TS_SAMPLE_PHILOX, seed=2024)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    distinct = len({c.cpu().numpy().tobytes() for c in codes})
    return {"workload": "BASELINE configs[3]: 12 stochastic samples of one 10 s clip (Philox4x32-10, seed 2024), audio encoder -> PixelCNN -> VQ decode",
            "frames_per_s": B * FRAMES_PER_CLIP / dt, "ms_per_call": dt * 1e3, "distinct_samples": distinct}


def roofline_block(w, lib, _lib, stream, mfcc, ids, B, H, local, step):
    """Roofline of the dominant kernel + per-family breakdown.

    Dominant kernel (≈85 % of a batch's device time): skinny_gemm_f32, the per-position GEMM of the PixelCNN chain.
    Its launches are replayed from one hipGraph per batch, so the live measurement is: HIP events recorded on the launch
    stream around one replay (5 repeats, median) / the number of skinny launches inside (ts_pixelcnn_graph_stats, which
    also gives the algorithmic flops 2*M*N*K summed over those launches).  The 150 sampler launches inside the same
    replay are <2 % of it.  conv_gemm_f32 (VQ encoder/decoder, audio encoder) is measured with HIP event pairs around
    every launch on the launch stream (ts_prof_*).
    """
    res = {}
    with torch.cuda.stream(stream):
        feat = w.audioencoder.forward_nlc(mfcc)
        w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)
        times = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            w.generator.run(ids, feat, mode=_lib.TS_SAMPLE_GREEDY)
            e1.record(stream)
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
        n, fl = C.c_int64(), C.c_double()
        _lib.check(lib.ts_pixelcnn_graph_stats(w.generator.handle(), C.c_void_p(stream.cuda_stream), B, H,
                                               _lib.TS_SAMPLE_GREEDY, C.byref(n), C.byref(fl)))
    ms = sorted(times)[len(times) // 2]
    # algorithmic work of the incremental PixelCNN: 34,734,080 MAC per code row per clip, valid taps only;
    # the launches execute ~25 % more (composed horizontal maps), which is NOT counted as achieved work
    alg = 2.0 * 34734080.0 * H * B
    ach = alg / (ms * 1e-3) / 1e12
    traffic = None
    pmc = os.path.join(REPO, "profiles", "r01_pmc_summary.json")
    if os.path.exists(pmc):
        traffic = json.load(open(pmc)).get("skinny_gemm_f32", {}).get("hbm_bytes_per_launch")
    res["roofline"] = {"kernel": "skinny_gemm_f32 (PixelCNN per-position GEMM chain)", "bound": "mfma", "achieved": ach,
                       "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
                       "traffic": traffic, "launches_per_batch": n.value, "avg_launch_us": ms * 1e3 / n.value,
                       "algorithmic_flops_per_launch": alg / n.value, "executed_flops_per_launch": fl.value / n.value,
                       "chain_ms_per_batch": ms}
    # per-family pass: event pair around every launch (eager launches, so the chain is slower here than in production)
    ctx = _lib.context(local)
    _lib.check(lib.ts_prof_enable(ctx, 1))
    step(0)
    torch.cuda.synchronize()
    msf, nf, flf = (C.c_double * 3)(), (C.c_int64 * 3)(), (C.c_double * 3)()
    _lib.check(lib.ts_prof_read(ctx, msf, nf, flf, 1))
    _lib.check(lib.ts_prof_enable(ctx, 0))
    ach_c = flf[0] / (msf[0] * 1e-3) / 1e12
    res["roofline_conv_gemm"] = {"kernel": "conv_gemm_f32 (VQ encoder/decoder + audio encoder layers)", "bound": "mfma",
                                 "achieved": ach_c, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": ach_c / PEAK_FP32_MFMA_TFLOPS, "launches_per_batch": nf[0],
                                 "avg_launch_us": msf[0] * 1e3 / nf[0], "ms_per_batch": msf[0]}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("TS_BENCH_STREAMS", "4")),
                    help="independent steps (batches of 32 clips) in flight at once, one HIP stream each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-face", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    torch.cuda.set_device(local)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from talkshow_amd import _lib, synth
    from talkshow_amd.parallel import gather_sequences
    lib = _lib.load()
    w, sds = build_models(local)
    B, T = a.batch, FRAMES_PER_CLIP
    dev = torch.device("cuda", local)
    # a few distinct resident input batches, cycled; rank r owns global clips [r*B, (r+1)*B) of each step
    NB = 3
    mfcc = [torch.from_numpy(synth.mfcc_features(1000 + 10 * rank + k, B, T)).to(dev) for k in range(NB)]
    gt = [torch.from_numpy(synth.gt_poses(2000 + 10 * rank + k, B, T)).to(dev) for k in range(NB)]
    ids = torch.from_numpy(synth.speaker_ids(B)).to(dev)
    H = T // 4
    S = max(1, a.streams)
    streams = _lib.create_streams(S, local)
    gt_codes = [torch.empty((B, H, 2), dtype=torch.int64, device=dev) for _ in range(S)]

    def step(k):
        # one complete pass over one batch of 32 clips, enqueued on stream k % S (the library keeps one scratch arena
        # per stream; weights are shared)
        with torch.cuda.stream(streams[k % S]):
            s = _lib.stream_ptr()
            # VQ-VAE encode half of configs[1] (VQVAE.encode of the 300 GT frames, body and hand)
            _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(gt[k % NB]), B, T,
                                            _lib.dptr(gt_codes[k % S]), None, s))
            # audio encoder -> PixelCNN greedy -> VQ decode
            return w.generate_batch(mfcc[k % NB], ids, mode=_lib.TS_SAMPLE_GREEDY, clip_index0=rank * B)

    def barrier():
        if world > 1:
            dist.barrier()

    torch.cuda.synchronize()
    for k in range(max(a.warmup, S)):      # at least one pass per stream: graph capture + scratch allocation are warm-up
        step(k)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        codes, poses = step(k)
    torch.cuda.synchronize()
    if world > 1:
        all_poses = gather_sequences(poses)            # the one exchange: (N*B, 300, 129) on every rank
        torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    frames = world * a.steps * B * FRAMES_PER_CLIP
    out = {
        "metric": "generated SMPL-X frames/sec (10 s @ 30 fps clips), whole job",
        "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (seeded MFCC-scale features / poses, random-init weights of the reference architecture)",
        "config": {"workload": "BASELINE configs[1]: batch=32 x 10 s clips, body+hand VQ-VAE encode -> PixelCNN greedy decode -> VQ decode, 30 fps",
                   "batch_per_gpu": B, "frames_per_clip": FRAMES_PER_CLIP,
                   "parallelism": f"clip-sharded x{world}, {S} batches in flight per GPU (one HIP stream each)"},
        "per_gpu_frames_per_s": frames / dt / world,
        "streams": S,
    }
    # whole path against the fp32 MFMA roof: algorithmic work of configs[1] (64.25 MFLOP per generated frame)
    ach = 64.25e6 * frames / dt / world / 1e12
    out["whole_path"] = {"algorithmic_TFLOPs_per_gpu": ach, "frac_of_fp32_mfma_peak": ach / PEAK_FP32_MFMA_TFLOPS}
    # latency of ONE isolated batch (a single stream, nothing else in flight)
    lat = []
    for k in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(0)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - t1)
    out["batch_latency_ms"] = sorted(lat)[1] * 1e3

    if rank == 0 and not a.no_roofline:
        out.update(roofline_block(w, lib, _lib, streams[0], mfcc[0], ids, B, H, local, step))
    if rank == 0 and not a.no_face:
        try:
            out["diversity"] = diversity_block(w, _lib, mfcc[0])
        except Exception as e:
            out["diversity"] = {"error": repr(e)}
        try:
            out["face"] = face_block(local)
        except Exception as e:                       # the face line is extra information; never lose the main line
            out["face"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sds, 1000)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()          # rank 0's extra measurement legs are over: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
