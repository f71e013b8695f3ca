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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
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
    gt_codes = torch.empty((B, H, 2), dtype=torch.int64, device=dev)
    lat_b = torch.empty((B, H), dtype=torch.int64, device=dev)
    lat_h = torch.empty((B, H), dtype=torch.int64, device=dev)
    gt_body = [g[..., :39].contiguous() for g in gt]
    gt_hand = [g[..., 39:].contiguous() for g in gt]

    def step(k):
        s = _lib.stream_ptr()
        # VQ-VAE encode half of configs[1] (VQVAE.encode of the 300 GT frames, body and hand)
        _lib.check(lib.ts_vqvae_encode(w.g_body.handle(), _lib.dptr(gt_body[k % NB]), B, T, None, _lib.dptr(lat_b), None, s))
        _lib.check(lib.ts_vqvae_encode(w.g_hand.handle(), _lib.dptr(gt_hand[k % NB]), B, T, None, _lib.dptr(lat_h), None, s))
        # audio encoder -> PixelCNN greedy -> VQ decode
        return w.generate_batch(mfcc[k % NB], ids, mode=_lib.TS_SAMPLE_GREEDY, clip_index0=rank * B)

    def barrier():
        if world > 1:
            dist.barrier()

    for k in range(a.warmup):
        step(k)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        codes, poses = step(k)
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
                   "batch_per_gpu": B, "frames_per_clip": FRAMES_PER_CLIP, "parallelism": f"clip-sharded x{world}"},
        "per_gpu_frames_per_s": frames / dt / world,
    }

    if rank == 0 and not a.no_roofline:
        # instrumented pass (HIP events around every launch, on the launch stream): per-kernel-family device time
        ctx = _lib.context(local)
        _lib.check(lib.ts_prof_enable(ctx, 1))
        nprof = 2
        for k in range(nprof):
            step(k)
        ms = (C.c_double * 3)()
        n = (C.c_int64 * 3)()
        fl = (C.c_double * 3)()
        _lib.check(lib.ts_prof_read(ctx, ms, n, fl, 1))
        _lib.check(lib.ts_prof_enable(ctx, 0))
        fam = ["conv_gemm_f32", "skinny_gemm_f32", "vq/sample/glue"]
        per = {fam[i]: {"ms_per_step": ms[i] / nprof, "launches_per_step": n[i] / nprof,
                        "avg_launch_us": (ms[i] / n[i] * 1e3) if n[i] else None,
                        "tflops": (fl[i] / (ms[i] * 1e-3) / 1e12) if ms[i] > 0 and fl[i] > 0 else None} for i in range(3)}
        dom = max(range(2), key=lambda i: ms[i])
        ach = fl[dom] / (ms[dom] * 1e-3) / 1e12
        out["roofline"] = {"kernel": fam[dom], "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                           "avg_launch_us": per[fam[dom]]["avg_launch_us"], "launches_per_step": per[fam[dom]]["launches_per_step"]}
        out["kernel_families"] = per
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sds, 1000)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
