#!/usr/bin/env python3
"""Throughput bench of the speech -> SMPL-X body hot path on MI355X (contract: see the task statement / DESIGN.md §4).

One step = BASELINE.json configs[1]: one batch of 32 synthetic 10 s clips through
    VQ-VAE encode of 300 GT frames (body + hand)  ->  audio encoder -> PixelCNN greedy decode (75 x 2 codes)
    ->  VQ decode to (32, 300, 129) SMPL-X pose parameters,
inputs resident in HBM before the timed region, fp32, seeded random-init weights of the reference architecture.
value = generated frames / s over all ranks (32 * 300 frames per step per rank).

How the steps are executed (serving-style, `--coalesce G --streams S`): the K steps of a run are K queued batches; the
engine stacks up to G of them per pass (spread evenly over a multiple of S passes) and each pass goes through the path as
one pass of up to 32*G clips (the autoregressive chain is latency-bound below ~64 clips per stage: one pass over 256 clips
streams each stage's weights once instead of 8 times), on S HIP streams so that the conv stacks of one pass overlap the
chain of another.  A clip's result does not depend on how
its batch was grouped (bit-identical; tests/test_gpu_parity.py::test_golden_clips_inside_baseline_batches).  Every
step's batch is completely processed inside the timed region.  The strict one-batch-at-a-time figure and the round-1
mode (4 independent batches on 4 streams, no coalescing) are measured beside it (`modes`).

    python bench.py                          # N=1, configs[1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W
    ... bench.py --config whole_body         # BASELINE configs[4]: 128 clips per rank, body_pixel + face -> 265-d rows

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

# independent groups are pipelined over several HIP streams; give each its own hardware queue
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
if os.environ.get("TS_BENCH_WATCHDOG"):     # debugging aid: dump every thread's Python stack and exit after N seconds
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ["TS_BENCH_WATCHDOG"]), exit=True)

FRAMES_PER_CLIP = 300
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_{32x32x2,16x16x4}_f32, 256 CUs x 2.4 GHz
PEAK_HBM_GBS = 8000.0
ALG_MAC_PER_ROW_PER_CLIP = 34734080.0   # incremental PixelCNN, SURVEY.md §8d (valid taps only)
ALG_FLOP_PER_FRAME = 64.25e6            # configs[1] whole path, SURVEY.md §8d


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


def build_face(device_index, seed=0):
    from talkshow_amd import synth
    from talkshow_amd.modules import FaceGenerator
    m = FaceGenerator().to(torch.device("cuda", device_index))
    m.load_state_dict(synth.to_torch(synth.face_state_dict(seed=seed)))
    return m


def reference_greedy(pix, label, aud):
    """The greedy harness of SURVEY.md §0.3 around the REFERENCE's `GatedPixelCNN.forward` (`gated_pixelcnn_v2.py:130-150`): one
    full-grid forward per code position, argmax of that position's logits (the reference's own `generate` draws from torch's
    multinomial, `:167-176`; same cost per position)."""
    B, H = aud.shape[0], aud.shape[2]
    x = torch.zeros((B, H, 2), dtype=torch.int64)
    with torch.no_grad():
        for i in range(H):
            for j in range(2):
                x[:, i, j] = torch.argmax(pix(x, label, aud)[:, :, i, j], dim=-1)
    return x


def cpu_model():
    try:
        return [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:                                        # noqa: BLE001
        return "unknown"


def cpu_baseline_reference(sds, seed, budget_s=25.0, dims=None, frames=None):
    """kind "reference": the reference's OWN nn.Modules — `nets/spg/gated_pixelcnn_v2.py::GatedPixelCNN`, `nets/spg/vqvae_1d.py::VQVAE /
    AudioEncoder`, lifted into oracle/_ref as hash-verified code objects where /root/reference exists (oracle/build_ref_callers.py; the
    tree itself cannot travel) — on this box's host cores, on a bounded sample of configs[1]: VQVAE.encode of the GT poses (body + hand),
    AudioEncoder, the greedy harness around GatedPixelCNN.forward (2 H full-grid forwards), VQVAE.decode of both code rows.
    -> the cpu_baseline block, or None where oracle/_ref holds no modules (the caller falls back to the torch port)."""
    import contextlib
    import io
    from oracle import build_ref_callers as BRC
    from talkshow_amd import synth
    try:
        R = BRC.load_reference_modules()
    except BRC.RefCallersError as e:
        print(f"[bench] reference modules not loadable: {e}", file=sys.stderr)
        return None
    if R is None:
        return None
    d = dict(input_dim=2048, dim=256, n_layers=15, num_embeddings=2048, num_hiddens=1024)
    d.update(dims or {})
    T_ = FRAMES_PER_CLIP if frames is None else frames
    H = T_ // 4
    tt = synth.to_torch
    with contextlib.redirect_stdout(io.StringIO()):      # the constructor prints one line per layer (gated_pixelcnn_v2.py:6-13)
        pix = R.GatedPixelCNN(d["input_dim"], d["dim"], d["n_layers"], 4, True, True)
    pix.load_state_dict(tt(sds["pix"]), strict=True)
    ae = R.AudioEncoder(64, 256, 2, 256)
    ae.load_state_dict(tt(sds["audio"]), strict=True)
    vb = R.VQVAE(39, 64, d["num_embeddings"], d["num_hiddens"], 2, 512)
    vb.load_state_dict(tt(sds["body"]), strict=True)
    vh = R.VQVAE(90, 64, d["num_embeddings"], d["num_hiddens"], 2, 512)
    vh.load_state_dict(tt(sds["hand"]), strict=True)
    for m in (pix, ae, vb, vh):
        m.eval()
    threads = min(32, os.cpu_count() or 1)               # fixed pool, as for the port: these layers stop scaling well before 256 threads
    torch.set_num_threads(threads)
    with torch.no_grad():
        x0, aud0, lab0 = torch.zeros((4, H, 2), dtype=torch.int64), torch.zeros((4, 256, H, 2)), torch.zeros(4, dtype=torch.int64)
        pix(x0, lab0, aud0)
        t0 = time.perf_counter()
        pix(x0, lab0, aud0)
        per_fwd4 = time.perf_counter() - t0
    clips = int(max(1, min(32, budget_s // max(per_fwd4 / 4 * 2 * H * 1.15, 1e-3))))
    mf, ids = synth.mfcc_features(seed, clips, T_), synth.speaker_ids(clips)
    gt = torch.from_numpy(synth.gt_poses(seed, clips, T_))
    t0 = time.perf_counter()
    with torch.no_grad():
        vb.encode(gt_poses=gt[..., :39].contiguous())                                       # the encode half of configs[1]
        vh.encode(gt_poses=gt[..., 39:].contiguous())
        feat = ae(torch.from_numpy(mf).transpose(1, 2), frame_num=0)                        # smplx_body_pixel.py:274
        codes = reference_greedy(pix, torch.from_numpy(ids), feat.unsqueeze(-1).repeat(1, 1, 1, 2))
        body, _ = vb.decode(b=clips, w=H, latents=codes[..., 0])                            # :282-285
        hand, _ = vh.decode(b=clips, w=H, latents=codes[..., 1])
        poses = torch.cat([body, hand], dim=1).transpose(1, 2)
    dt = time.perf_counter() - t0
    assert tuple(poses.shape) == (clips, 4 * H, 129)
    return {"value": clips * T_ / dt, "unit": "frames/s", "cores": threads, "kind": "reference", "cpu": cpu_model(), "host_cpus": os.cpu_count(),
            "sample": f"{clips} clip(s) ({T_} frames each) of configs[1] through the reference's own nn.Modules (oracle/_ref: nets/spg/gated_pixelcnn_v2.py, "
                      f"vqvae_1d.py, vqvae_modules.py compiled where /root/reference exists): VQVAE.encode x2 + AudioEncoder + greedy harness around "
                      f"GatedPixelCNN.forward ({2 * H} full-grid forwards) + VQVAE.decode x2, fp32, torch {torch.__version__}, {threads} threads of "
                      f"{os.cpu_count()} host CPUs, one complete pass, {dt:.1f} s"}


def cpu_baseline(sds, seed, budget_s=25.0):
    """The reference's algorithm (full-grid recompute per code position) on this box's host cores.

    kind "reference" (`cpu_baseline_reference`) where oracle/_ref holds the reference's own modules — what this function returns
    then (TS_BENCH_CPU_PORT=1 forces the port, the figure of rounds 1-5); else:

    kind "port": oracle/torch_port.py — the reference's forward()s restated on the torch CPU ops its nn.Modules dispatch
    to (pinned to the reference goldens, tests/test_oracle_golden.py); /root/reference itself cannot travel to the GPU
    box.  Bounded sample: as many clips of the configs[1] workload (VQ encode + greedy generate + VQ decode) as fit
    ~`budget_s` seconds, sized from a probe of ONE full-grid forward, run once, complete (nothing extrapolated).
    Threads: torch's intra-op pool is fixed at 32 (comparable run to run; a 256-thread pool on these layer sizes only adds
    synchronisation).  `reference_build_box`: the reference's OWN modules timed in the build container
    (tools/time_reference_cpu.py)."""
    from oracle import torch_port as TP
    from talkshow_amd import synth
    if os.environ.get("TS_BENCH_CPU_PORT", "0") != "1":
        ref_leg = cpu_baseline_reference(sds, seed, budget_s)
        if ref_leg is not None:
            return ref_leg
    H = FRAMES_PER_CLIP // 4
    # one full-grid PixelCNN forward at 4 clips sizes the sample.  The intra-op pool is FIXED at 32 threads (or the host's CPU
    # count if smaller): these layers stop scaling well before a 256-thread host is full (a 256-thread pool made the pass 50x
    # slower), and a per-box probe of the pool size made the number swing 310-610 frames/s between boxes
    with torch.no_grad():
        sp = TP._t(sds["pix"])
        x0 = torch.zeros((4, H, 2), dtype=torch.int64)
        aud0 = torch.zeros((4, 256, H, 2))
        lab0 = torch.zeros(4, dtype=torch.int64)
        threads = min(32, os.cpu_count() or 1)
        torch.set_num_threads(threads)
        TP.pixelcnn_forward(x0, lab0, aud0, sp, 15)
        t0 = time.perf_counter()
        TP.pixelcnn_forward(x0, lab0, aud0, sp, 15)
        per_fwd4 = time.perf_counter() - t0
    est_per_clip = per_fwd4 / 4 * 2 * H * 1.15            # + VQ encode / decode
    clips = int(max(1, min(32, budget_s // max(est_per_clip, 1e-3))))
    mf, ids = synth.mfcc_features(seed, clips, FRAMES_PER_CLIP), synth.speaker_ids(clips)
    gt = synth.gt_poses(seed, clips, FRAMES_PER_CLIP)
    t0 = time.perf_counter()
    TP.vq_encode_pair(gt, sds["body"], sds["hand"])
    TP.body_pixel_infer(mf, ids, sds["audio"], sds["pix"], sds["body"], sds["hand"])
    dt = time.perf_counter() - t0
    out = {"value": clips * FRAMES_PER_CLIP / dt, "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{clips} clip(s) (10 s, 300 frames each) of configs[1]: VQ encode + greedy full-grid PixelCNN generate "
                     f"(150 forwards) + VQ decode, torch CPU ops fp32 (oracle/torch_port.py), {threads} threads of "
                     f"{os.cpu_count()} host CPUs, one complete pass, {dt:.1f} s"}
    ref = os.path.join(REPO, "profiles", "r02_reference_cpu_buildbox.json")
    if os.path.exists(ref):
        r = json.load(open(ref))
        out["reference_build_box"] = {
            "what": "the reference's own nn.Modules (imported from /root/reference) timed in the build container by "
                    "tools/time_reference_cpu.py; not this box",
            "cpu": r.get("cpu"), "cores": r.get("cores"),
            "body_frames_per_s": r["body"]["frames_per_s"], "body_batch": r["body"]["batch"],
            "face_frames_per_s": r["face"]["frames_per_s"], "face_batch": r["face"]["batch"]}
    return out


def face_block(local):
    """BASELINE configs[2] as extra information: face generator, batch 64 x 10 s @16 kHz -> (64,300,103), fp32."""
    from talkshow_amd import _lib, synth
    lib = _lib.load()
    ctx = _lib.context(local)
    m = build_face(local)
    B, N, T = 64, 160000, 300
    wav = torch.from_numpy(synth.wav16(3000, B, N)).cuda()
    ids = torch.nn.functional.one_hot(torch.arange(B) % 4, 4).float().cuda()
    m.run(wav, ids, T)
    torch.cuda.synchronize()
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
    msf, nf, flf = (C.c_double * 4)(), (C.c_int64 * 4)(), (C.c_double * 4)()
    _lib.check(lib.ts_prof_read_n(ctx, 4, msf, nf, flf, 1))
    _lib.check(lib.ts_prof_enable(ctx, 0))
    ach = flf[0] / (msf[0] * 1e-3) / 1e12
    ach_a = flf[3] / (msf[3] * 1e-3) / 1e12 if msf[3] > 0 else 0.0
    out = {"workload": "BASELINE configs[2]: face generator, batch=64 x 10 s @16 kHz, 103 params @30 fps",
           "frames_per_s": B * T / dt, "ms_per_batch": dt * 1e3,
           "conv_gemm_f32": {"launches": nf[0], "ms": msf[0], "achieved_TFLOPs": ach, "frac_of_fp32_mfma_peak": ach / PEAK_FP32_MFMA_TFLOPS},
           "attention_fused": {"launches": nf[3], "ms": msf[3], "achieved_TFLOPs": ach_a, "frac_of_fp32_mfma_peak": ach_a / PEAK_FP32_MFMA_TFLOPS,
                               "what": "QK^T -> online soft-max -> PV per (clip, head) in one kernel (csrc/face.hip::attention_kernel), 12 layers"},
           "other_kernels_ms": msf[2]}
    # two batches of 64 in flight on two streams (one weight copy, scratch per stream): what a serving loop gains from filling one
    # batch's partly filled GEMM rounds and launch gaps with the other's workgroups; each batch's rows are the one-stream rows
    try:
        streams = _lib.create_streams(2, local)
        sets = [(wav, ids), (torch.from_numpy(synth.wav16(3001, B, N)).cuda(), ids)]
        res = [None, None]

        def both():
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    res[i] = m.run(sets[i][0], sets[i][1], T)
        ref0 = m.run(wav, ids, T)
        both()
        torch.cuda.synchronize()
        same = bool(torch.equal(res[0], ref0))
        t0 = time.perf_counter()
        for _ in range(K):
            both()
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / K / 2
        out["two_batches_in_flight"] = {"ms_per_batch": dt2 * 1e3, "frames_per_s": B * T / dt2, "rows_equal_one_stream": same}
    except Exception as e:                                   # extra information; never lose the block
        out["two_batches_in_flight"] = {"error": repr(e)}
    # OPT-IN split-bf16 plans beside the fp32 line (never the headline; dtype of the main line stays f32): speed on the same batch,
    # error of the same two reference-golden clips the parity tests use (tests/golden/face_10s.npz), embedded in a batch of 64
    try:
        g = np.load(os.path.join(REPO, "tests", "golden", "face_10s.npz"))
        seed, gb, gn = (int(v) for v in g["wav_seed"])
        gwav = torch.from_numpy(synth.wav16(seed, gb, gn)).cuda()
        gids = torch.from_numpy(g["ids"]).cuda()
        m.load_state_dict(synth.to_torch(synth.face_state_dict(seed=7)))      # the golden's weights (timing does not depend on them)
        split = {"fp32_max_abs_err_vs_reference_golden": float((m.run(gwav, gids, T).cpu() - torch.from_numpy(g["out"])).abs().max())}
        for products in (6, 3):
            m.set_arith(products)
            m.run(wav, ids, T)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                m.run(wav, ids, T)
            torch.cuda.synchronize()
            dts = (time.perf_counter() - t0) / K
            err = float((m.run(gwav, gids, T).cpu() - torch.from_numpy(g["out"])).abs().max())
            split[f"bf16x{products}"] = {"terms": products, "frames_per_s": B * T / dts, "ms_per_batch": dts * 1e3,
                                         "speedup_vs_fp32": dt / dts, "max_abs_err_vs_reference_golden": err}
        m.set_arith(0)
        out["split_bf16"] = split
    except Exception as e:
        out["split_bf16"] = {"error": repr(e)}
    return out


def frontend_block(w, _lib, clips):
    """The audio front-end on the device (a4 / f1), as extra information: `clips` synthetic 10 s 16 kHz waveforms ->
    sinc-Hann resample to 22 kHz -> MFCC(64) (`ts_mfcc_forward`: fused framing + real FFT + power kernel, mel / DCT GEMMs), alone and in front of
    one body pass (wav in -> poses out).  The headline's timed region starts from resident MFCC features, as the reference's
    hot path does after `get_mfcc_ta`; this block shows what wav-in costs on top."""
    from talkshow_amd import synth
    from talkshow_amd.modules import MFCC
    fe = MFCC(16000, 22000, 30)
    wav = torch.from_numpy(synth.wav16(7000, clips, 160000)).cuda()
    ids = torch.from_numpy(synth.speaker_ids(clips)).cuda()
    feat = fe(wav)
    assert tuple(feat.shape) == (clips, FRAMES_PER_CLIP, 64), feat.shape
    w.generate_batch(feat, ids, mode=_lib.TS_SAMPLE_GREEDY)
    t_fe = timed(lambda: fe(wav))
    t_all = timed(lambda: w.generate_batch(fe(wav), ids, mode=_lib.TS_SAMPLE_GREEDY))
    t_body = timed(lambda: w.generate_batch(feat, ids, mode=_lib.TS_SAMPLE_GREEDY))
    return {"workload": f"{clips} x 10 s @16 kHz waveforms resident in HBM -> resample 22 kHz -> MFCC(64, n_fft 2048, hop 734) "
                        f"[-> audio encoder -> PixelCNN greedy -> VQ decode]",
            "frontend_ms": t_fe * 1e3, "frontend_ms_per_32_clips": t_fe * 1e3 * 32 / clips,
            "wav_to_poses_ms": t_all * 1e3, "features_to_poses_ms": t_body * 1e3,
            "frontend_share_of_wav_to_poses": t_fe / t_all,
            "stability": wav_in_stability(w, _lib, wav[:32].cpu().numpy(), ids[:32])}


def wav_in_stability(w, _lib, wav16, ids):
    """Outside any timed region: do the greedy codes depend on WHICH arithmetic produced the MFCC rows?  The same resampled
    waveforms go through the device MFCC (fp32 FFT in LDS, `ts_mfcc_forward`) and through the float64 host twin
    (`frontend.mfcc_float64`); both feature sets then run the same greedy body pass.  Reported: the largest MFCC difference,
    codes that differ (of clips x 150), clips with any difference, the largest pose difference."""
    from talkshow_amd import frontend as FE
    from talkshow_amd.modules import MFCC
    n = len(wav16)
    x22 = np.stack([FE.resample_sinc_hann(x[None], 16000, 22000)[0] for x in wav16])          # one resampler for both arms
    dev = MFCC(22000, 22000, 30)(torch.from_numpy(x22).cuda())
    twin = np.stack([FE.mfcc_float64(x, 22000, hop_length=734).T for x in x22])                # (n, 300, 64) float64
    err = float(np.abs(dev.cpu().numpy().astype(np.float64) - twin).max())
    c0, p0 = w.generate_batch(dev, ids, mode=_lib.TS_SAMPLE_GREEDY)
    c1, p1 = w.generate_batch(torch.from_numpy(twin.astype(np.float32)).cuda(), ids, mode=_lib.TS_SAMPLE_GREEDY)
    diff = (c0 != c1).cpu().numpy()
    return {"clips": n, "mfcc_max_abs_err_vs_float64": err, "mfcc_max_abs": float(np.abs(twin).max()),
            "codes_differing": int(diff.sum()), "codes_total": int(diff.size), "clips_with_a_difference": int(diff.reshape(n, -1).any(1).sum()),
            "max_pose_delta": float((p0 - p1).abs().max().item())}


def diversity_block(w, _lib, mfcc1):
    """BASELINE configs[3] as extra information: num_samples=12 stochastic decodes of ONE 10 s clip (Philox seed 2024)."""
    B = 12
    mf = mfcc1[:1].repeat(B, 1, 1).contiguous()
    ids = torch.zeros(B, dtype=torch.int64, device=mf.device)
    w.generator.prepare(B, mf.shape[1] // 4, _lib.TS_SAMPLE_PHILOX)
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


def chain_roofline(w, lib, _lib, stream, mfcc, ids, H, pmc_key):
    """Dominant kernel: skinny_gemm_f32, the per-position GEMM of the PixelCNN chain, at M = mfcc.shape[0] clips per stage.

    Its launches are replayed from one hipGraph per pass, so the live measurement is: HIP events recorded on the launch
    stream around one replay (5 repeats, median) / the number of skinny launches inside (ts_pixelcnn_graph_stats, which
    also gives the executed flops 2*M*N*K summed over those launches).  The sampler launches inside the same replay are
    <2 % of it.  `achieved` uses the ALGORITHMIC flops (34,734,080 MAC per code row per clip); the launches
    execute ~10 % more (composed horizontal maps), which is not counted as achieved work."""
    M = int(mfcc.shape[0])
    with torch.cuda.stream(stream):
        feat = w.audioencoder.forward_nlc(mfcc)
        w.generator.prepare(M, H, _lib.TS_SAMPLE_GREEDY)       # the whole-call graph (one replay per call), not the chunk graphs of a first sighting
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
        _lib.check(lib.ts_pixelcnn_graph_stats(w.generator.handle(), C.c_void_p(stream.cuda_stream), M, H,
                                               _lib.TS_SAMPLE_GREEDY, C.byref(n), C.byref(fl)))
    ms = sorted(times)[len(times) // 2]
    alg = 2.0 * ALG_MAC_PER_ROW_PER_CLIP * H * M
    ach = alg / (ms * 1e-3) / 1e12
    # HBM-side bytes per launch come from rocprofv3 PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md), which cannot
    # run inside this process: the recorded value of the committed summary is quoted, with its source, never passed off as live
    traffic = traffic_source = None
    for name in ("r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json"):
        pmc = os.path.join(REPO, "profiles", name)
        if os.path.exists(pmc):
            rec = json.load(open(pmc)).get(pmc_key, {})
            if rec.get("hbm_bytes_per_launch") is not None:
                traffic = rec["hbm_bytes_per_launch"]
                traffic_source = f"profiles/{name} [{pmc_key}] (rocprofv3 --pmc passes of tools/profile_{name[:3]}.sh, recorded at commit {rec.get('commit', 'see git log of the file')}; not measured by this run)"
                break
    return {"kernel": "skinny_gemm_f32 (PixelCNN per-position GEMM chain)", "clips_per_stage": M, "bound": "mfma",
            "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_MFMA_TFLOPS,
            "traffic": traffic, "traffic_source": traffic_source, "launches_per_pass": n.value, "avg_launch_us": ms * 1e3 / n.value,
            "algorithmic_flops_per_launch": alg / n.value, "executed_flops_per_launch": fl.value / n.value,
            "chain_ms_per_pass": ms, "chain_ms_per_32_clips": ms * 32.0 / M}


def conv_roofline(lib, _lib, local, run_pass):
    """conv_gemm_f32 (VQ encoder/decoder, audio encoder): HIP event pairs around every launch on the launch stream
    (ts_prof_*; eager launches), one pass of the operating point."""
    ctx = _lib.context(local)
    _lib.check(lib.ts_prof_enable(ctx, 1))
    run_pass()
    torch.cuda.synchronize()
    msf, nf, flf = (C.c_double * 3)(), (C.c_int64 * 3)(), (C.c_double * 3)()
    _lib.check(lib.ts_prof_read(ctx, msf, nf, flf, 1))
    _lib.check(lib.ts_prof_enable(ctx, 0))
    ach_c = flf[0] / (msf[0] * 1e-3) / 1e12
    return {"kernel": "conv_gemm_f32 (VQ encoder/decoder + audio encoder layers)", "bound": "mfma",
            "achieved": ach_c, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": ach_c / PEAK_FP32_MFMA_TFLOPS, "launches_per_pass": nf[0],
            "avg_launch_us": msf[0] * 1e3 / nf[0], "ms_per_pass": msf[0]}


def whole_body_block(w, _lib, local, batch_body):
    """BASELINE configs[4] at ONE rank, as extra information in the default line: 128 clips per step, whole body =
    body_pixel greedy (one coalesced pass) + face generator (2 batches of 64) -> (128, 300, 265) rows assembled on the GPU; fp32
    (the parity path) and, beside it, the opt-in bf16x3 plan of the face GEMMs.  The multi-rank form of the same step is
    `bench.py --config whole_body` (one all-gather of the rows per step)."""
    import types
    from talkshow_amd import parallel, synth
    face = types.SimpleNamespace(generator=build_face(local))
    dev = torch.device("cuda", local)
    n, T = 128, FRAMES_PER_CLIP
    mfcc = torch.from_numpy(synth.mfcc_features(4000, n, T)).to(dev)
    ids = torch.from_numpy(synth.speaker_ids(n)).to(dev)
    wav = torch.from_numpy(synth.wav16(5000, n, 160000)).to(dev)
    fid = torch.nn.functional.one_hot(torch.arange(n) % 4, 4).float().to(dev)

    def step(overlap=True):
        return parallel.whole_body_local(w, face, mfcc, ids, wav, fid, mode=_lib.TS_SAMPLE_GREEDY, clip_index0=0,
                                         batch_body=batch_body, batch_face=64, overlap=overlap)
    rows = step()
    rows_serial = step(overlap=False)
    torch.cuda.synchronize()
    assert tuple(rows.shape) == (n, T, 265) and bool(torch.isfinite(rows).all()) and torch.equal(rows, rows_serial)
    t32 = timed(step)
    t32_serial = timed(lambda: step(overlap=False))
    out = {"workload": f"BASELINE configs[4] at one rank: {n} synthetic 10 s clips per step, body_pixel greedy (one pass of {min(n, batch_body)} "
                       "clips) + face generator (batches of 64) -> (128, 300, 265) rows; no exchange at N = 1",
           "fp32": {"ms_per_step": t32 * 1e3, "frames_per_s": n * T / t32,
                    "one_stream_ms_per_step": t32_serial * 1e3, "what": "body path on a side stream under the face generator's GEMMs (bit-identical rows)"}}
    try:
        face.generator.set_arith(3)
        rows3 = step()
        torch.cuda.synchronize()
        t3 = timed(step)
        out["face_bf16x3_opt_in"] = {"ms_per_step": t3 * 1e3, "frames_per_s": n * T / t3,
                                     "max_abs_delta_vs_fp32_rows": float((rows3 - rows).abs().max().item())}
    finally:
        face.generator.set_arith(0)
    return out


class Engine:
    """configs[1] executor: submit() a 32-clip batch per step; groups of G batches run as one pass on alternating streams."""

    def __init__(self, w, lib, _lib, streams, B, T, G, mfcc, gt, ids, rank, enc_streams=None, pcie=False, wav=None):
        self.w, self.lib, self._lib = w, lib, _lib
        S = len(streams)
        self.B, self.T, self.H, self.G, self.S = B, T, T // 4, G, S
        self.mfcc, self.gt, self.rank = mfcc, gt, rank
        self.dev = mfcc[0].device
        self.streams = streams
        self.enc_streams = enc_streams     # optional: the VQ-encode half of a pass on its own stream(s) (it feeds nothing downstream)
        self.ids_rep = ids.repeat(G).contiguous()
        self.gt_codes = [torch.empty((B * G, self.H, 2), dtype=torch.int64, device=self.dev) for _ in range(S)]
        self.last = self.last_group = None
        self.pcie = pcie
        self.wav = wav                     # wav-in mode: resident (B, 160000) 16 kHz batches; the device front-end is part of every pass
        if wav is not None:
            from talkshow_amd.modules import MFCC
            self.fe = MFCC(16000, 22000, 30)
        if pcie:
            # PCIe-inclusive mode: the batches live in pinned host memory; per stream one device staging set + pinned result buffers
            pin = lambda t: t.cpu().pin_memory()
            self.h_mfcc, self.h_gt = [pin(t) for t in mfcc], [pin(t) for t in gt]
            n = B * G
            self.d_mfcc = [torch.empty((n, T, 64), dtype=torch.float32, device=self.dev) for _ in range(S)]
            self.d_gt = [torch.empty((n, T, gt[0].shape[-1]), dtype=torch.float32, device=self.dev) for _ in range(S)]
            self.h_poses = [torch.empty((n, T, gt[0].shape[-1]), dtype=torch.float32).pin_memory() for _ in range(S)]
            self.h_codes = [torch.empty((2, n, self.H, 2), dtype=torch.int64).pin_memory() for _ in range(S)]

    def run_group(self, ks, stream_index):
        NB, B, T, lib, _lib, w = len(self.mfcc), self.B, self.T, self.lib, self._lib, self.w
        n = B * len(ks)
        self.last_group = (list(ks), stream_index)
        if self.pcie:
            return self.run_group_pcie(ks, stream_index)

        def encode():
            # VQ-VAE encode half of configs[1] (VQVAE.encode of the 300 GT frames, body and hand)
            gtc = self.gt[ks[0] % NB] if len(ks) == 1 else torch.cat([self.gt[k % NB] for k in ks], 0)
            _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(gtc), n, T,
                                            _lib.dptr(self.gt_codes[stream_index][:n]), None, _lib.stream_ptr()))
        if self.enc_streams:
            with torch.cuda.stream(self.enc_streams[stream_index % len(self.enc_streams)]):
                encode()
        with torch.cuda.stream(self.streams[stream_index]):
            if not self.enc_streams:
                encode()
            # stacking the resident batches is part of the pass (device copies on the pass's stream)
            if self.wav is not None:   # 16 kHz samples -> sinc-Hann resample to 22 kHz -> MFCC(64) on this pass's stream (get_mfcc_ta, utils.py:148-231)
                mfc = self.fe(self.wav[ks[0] % NB] if len(ks) == 1 else torch.cat([self.wav[k % NB] for k in ks], 0))
            else:
                mfc = self.mfcc[ks[0] % NB] if len(ks) == 1 else torch.cat([self.mfcc[k % NB] for k in ks], 0)
            # audio encoder -> PixelCNN greedy -> VQ decode
            self.last = w.generate_batch(mfc, self.ids_rep[:n], mode=_lib.TS_SAMPLE_GREEDY, clip_index0=self.rank * B)
        return self.last

    def run_group_pcie(self, ks, si):
        """run_group with the pass's inputs copied in from pinned host memory and its results copied out to pinned host memory,
        all on the pass's stream (the copies of one pass overlap the compute of the passes on the other streams)."""
        NB, B, T, lib, _lib, w = len(self.mfcc), self.B, self.T, self.lib, self._lib, self.w
        n = B * len(ks)
        with torch.cuda.stream(self.streams[si]):
            for j, k in enumerate(ks):
                self.d_mfcc[si][j * B:(j + 1) * B].copy_(self.h_mfcc[k % NB], non_blocking=True)
                self.d_gt[si][j * B:(j + 1) * B].copy_(self.h_gt[k % NB], non_blocking=True)
            _lib.check(lib.ts_body_vq_infer(w.g_body.handle(), w.g_hand.handle(), _lib.dptr(self.d_gt[si][:n]), n, T,
                                            _lib.dptr(self.gt_codes[si][:n]), None, _lib.stream_ptr()))
            self.last = w.generate_batch(self.d_mfcc[si][:n], self.ids_rep[:n], mode=_lib.TS_SAMPLE_GREEDY, clip_index0=self.rank * B)
            self.h_poses[si][:n].copy_(self.last[1], non_blocking=True)
            self.h_codes[si][0, :n].copy_(self.last[0], non_blocking=True)
            self.h_codes[si][1, :n].copy_(self.gt_codes[si][:n], non_blocking=True)
        return self.last

    def pcie_bytes_per_step(self):
        B, T, C = self.B, self.T, self.gt[0].shape[-1]
        return B * T * 64 * 4 + 2 * B * T * C * 4 + 2 * B * self.H * 2 * 8

    def plan(self, steps):
        """How `steps` queued batches are grouped into passes: full passes of G batches, then the remainder.  (Spreading
        them evenly instead — 20 steps as 7, 7, 6 rather than 8, 8, 4 — measured 5 % SLOWER: a chain pass costs almost the
        same for 192, 224 or 256 clips, so partly filled passes waste it; round 3 on one box, `tools/plan_env_ab.sh`: 8,8,4 1.57–1.58 M,
        8,6,6 1.51, 6,6,8 1.51, 7,7,6 1.48 M frames/s.)"""
        forced = os.environ.get("TS_BENCH_PLAN")           # A/B aid: "8,6,6" (must add up to the step count)
        if forced:
            p = [int(x) for x in forced.split(",")]
            if sum(p) == steps and all(0 < x <= self.G for x in p):
                return p
        full, rest = divmod(steps, self.G)
        return [self.G] * full + ([rest] if rest else [])

    def run_steps(self, steps):
        k = 0
        self.outputs = []                  # every pass's (clips, 300, 129) poses, in step order: what the N > 1 exchange carries
        for gi, size in enumerate(self.plan(steps)):
            self.outputs.append(self.run_group(list(range(k, k + size)), gi % self.S)[1])
            k += size
        return self.last

    def all_rows(self):
        """(steps * 32, 300, 129): this rank's generated sequences of the last run_steps, in step order."""
        return self.outputs[0] if len(self.outputs) == 1 else torch.cat(self.outputs, 0)

    def warm(self, steps):
        """graph capture + scratch allocation for every (pass size, stream) the timed steps will use.  The PixelCNN's whole-call graph of
        a shape is captured and pinned explicitly (`ts_pixelcnn_prepare`): left to itself the library would run a shape on chunk
        graphs until its third sighting and capture the whole-call graph THERE — inside the timed region.  `captures()`
        before / after the timed steps shows that nothing was captured in between."""
        for g in sorted(set(self.plan(steps)), reverse=True):          # largest first: a buffer growth drops the graphs captured so far
            for si in range(self.S):
                with torch.cuda.stream(self.streams[si]):
                    self.w.generator.prepare(self.B * g, self.H, self._lib.TS_SAMPLE_GREEDY)
                self.run_group(list(range(g)), si)
        torch.cuda.synchronize()

    def captures(self):
        """hipGraphs captured so far on this engine's streams"""
        n = 0
        for st in self.streams:
            with torch.cuda.stream(st):
                n += self.w.generator.graph_captures()
        return n


def timed(fn, reps=3, warm=3):
    """median wall time of `reps` synchronised calls after `warm` untimed ones (three: a PixelCNN shape gets its whole-call graph on its
    third sighting on a stream, so the timed calls are replays whatever ran before)"""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def main_whole_body(a, world, rank, local, dist):
    """BASELINE configs[4]: 1 024 clips over 8 ranks = 128 clips per rank per step (weak scaling), whole body:
    body_pixel (batches of 32, coalesced) + face (batches of 64) -> (n_local, 300, 265) rows assembled on the GPU, ONE
    RCCL all-gather of the rows per step inside the timed region."""
    from talkshow_amd import _lib, synth
    from talkshow_amd import parallel
    import types
    coll = world > 1 or os.environ.get("TS_BENCH_FORCE_COLLECTIVES", "0") == "1"   # (world 1 over RCCL: tools/rccl_smoke.sh)
    w, sds = build_models(local)
    face = types.SimpleNamespace(generator=build_face(local))
    dev = torch.device("cuda", local)
    n_local = a.clips_per_rank
    N = n_local * world
    T = FRAMES_PER_CLIP
    # every rank materialises only its own block of the global inputs (seeded by global clip index range)
    lo, hi = parallel.shard_range(N, rank, world)
    mfcc = torch.from_numpy(synth.mfcc_features(4000 + rank, n_local, T)).to(dev)
    ids = torch.from_numpy(synth.speaker_ids(n_local)).to(dev)
    wav = torch.from_numpy(synth.wav16(5000 + rank, n_local, 160000)).to(dev)
    fid = torch.nn.functional.one_hot(torch.arange(n_local) % 4, 4).float().to(dev)

    def step():
        return parallel.whole_body_local(w, face, mfcc, ids, wav, fid, mode=_lib.TS_SAMPLE_GREEDY, clip_index0=lo,
                                         batch_body=a.batch * a.coalesce, batch_face=64)

    def barrier():
        if coll:
            dist.barrier()

    for _ in range(max(3, a.warmup)):      # >= 3: the body pass's whole-call graph is captured on a shape's third sighting
        rows = step()
    if coll:
        parallel.gather_sequences(rows, N)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        rows = step()
        allrows = parallel.gather_sequences(rows, N) if coll else rows
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if coll:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    frames = world * a.steps * n_local * FRAMES_PER_CLIP
    out = {
        "metric": "generated SMPL-X frames/sec (10 s @ 30 fps clips), whole job", "value": frames / dt, "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (seeded MFCC-scale features, 16 kHz noise clips, random-init weights of the reference architecture)",
        "config": {"workload": f"BASELINE configs[4]: {N} synthetic 10 s clips sharded over {world} rank(s) ({n_local} per rank per step), "
                               "whole body = body_pixel greedy + face generator -> (N,300,265) rows, one RCCL all-gather of the rows per step",
                   "clips_per_rank": n_local, "body_batch": a.batch, "coalesce": a.coalesce, "face_batch": 64,
                   "parallelism": f"clip-sharded x{world}, full weight replica per rank"},
        "per_gpu_frames_per_s": frames / dt / world, "gathered_shape": list(allrows.shape),
    }
    if rank == 0:
        print(json.dumps(out), flush=True)


class BodyJob:
    """configs[1] on this rank's GPU: what `run_contract` drives (tests/test_bench_control_flow.py drives the same function under
    gloo with a device-free stand-in of this class)."""

    def __init__(self, a, world, rank, local):
        from talkshow_amd import _lib, synth
        self.a, self.world, self.rank, self.local = a, world, rank, local
        self._lib, self.lib = _lib, _lib.load()
        self.w, self.sds = build_models(local)
        B, T = a.batch, FRAMES_PER_CLIP
        self.B, self.T, self.H = B, T, T // 4
        self.dev = torch.device("cuda", local)
        # a few distinct resident input batches, cycled; rank r owns global clips [r*B, (r+1)*B) of each step
        self.NB = 3
        self.mfcc = [torch.from_numpy(synth.mfcc_features(1000 + 10 * rank + k, B, T)).to(self.dev) for k in range(self.NB)]
        self.gt = [torch.from_numpy(synth.gt_poses(2000 + 10 * rank + k, B, T)).to(self.dev) for k in range(self.NB)]
        self.ids = torch.from_numpy(synth.speaker_ids(B)).to(self.dev)
        self.G, self.S = max(1, a.coalesce), max(1, a.streams)
        # one pool of library streams, created back to back (distinct hardware queues); every execution mode draws from it
        self.pool = _lib.create_streams(max(self.S, 1 if a.no_modes else 4), local)
        enc_pool = _lib.create_streams(a.enc_streams, local) if a.enc_streams > 0 else None
        self.eng = Engine(self.w, self.lib, _lib, self.pool[:self.S], B, T, self.G, self.mfcc, self.gt, self.ids, rank,
                          enc_streams=enc_pool)

    def warm(self, steps):
        torch.cuda.synchronize()
        self.eng.warm(steps)

    def run_steps(self, k):
        return self.eng.run_steps(k)

    def region_begin(self):
        self._cap0 = self.eng.captures()

    def region_end(self):
        self.captures_in_regions = getattr(self, "captures_in_regions", []) + [self.eng.captures() - self._cap0]

    def sync(self):
        torch.cuda.synchronize()

    def scalar(self, x):
        return torch.tensor([x], dtype=torch.float64, device=self.dev)

    def gather(self):
        """the ONE exchange of the job (north_star): every generated sequence of every step, (steps * 32, 300, 129) per rank,
        all-gathered to (N * steps * 32, 300, 129) on every rank"""
        from talkshow_amd.parallel import gather_sequences
        rows = self.eng.all_rows()
        allp = gather_sequences(rows)
        return {"gather_bytes_per_rank": int(rows.numel() * 4), "gathered_shape": list(allp.shape)}

    def frames_per_step(self):
        return self.B * FRAMES_PER_CLIP

    def describe(self):
        B, G, S = self.B, self.G, self.S
        return {"workload": "BASELINE configs[1]: batch=32 x 10 s clips, body+hand VQ-VAE encode -> PixelCNN greedy decode -> VQ decode, 30 fps",
                "batch_per_gpu": B, "frames_per_clip": FRAMES_PER_CLIP, "coalesce": G, "streams": S,
                "batches_per_pass": self.eng.plan(self.a.steps),
                "parallelism": f"clip-sharded x{self.world}; per GPU the queued batches run as passes of up to {G} batches "
                               f"({B * G} clips), {S} passes in flight (one HIP stream each)"}

    def selfcheck(self):
        """Outside the timed region: the LAST timed pass's outputs against the same clips run alone, one 32-clip batch on one
        stream (the strict mode) — generated codes, poses and the VQ-encode codes must be bit-equal (a clip's result does not
        depend on how its batch was grouped), so a skipped or mis-ordered launch inside the timed region cannot pass as a
        better number.  Raises on a mismatch: a wrong result must not leave a bench line behind."""
        eng, B, _lib = self.eng, self.B, self._lib
        caps = getattr(self, "captures_in_regions", [])
        if any(caps):
            raise RuntimeError(f"bench: hipGraphs were captured INSIDE the timed regions ({caps}): warm() did not cover a pass shape")
        ks, si = eng.last_group
        codes, poses = eng.last
        gtc = eng.gt_codes[si][:B * len(ks)]
        checked = []
        for j in sorted({0, len(ks) - 1}):
            k = ks[j]
            with torch.cuda.stream(self.pool[0]):
                c1, p1 = self.w.generate_batch(self.mfcc[k % self.NB], self.ids, mode=_lib.TS_SAMPLE_GREEDY, clip_index0=self.rank * B)
                g1 = torch.empty((B, self.H, 2), dtype=torch.int64, device=self.dev)
                _lib.check(self.lib.ts_body_vq_infer(self.w.g_body.handle(), self.w.g_hand.handle(), _lib.dptr(self.gt[k % self.NB]), B,
                                                     self.T, _lib.dptr(g1), None, _lib.stream_ptr()))
            torch.cuda.synchronize()
            sl = slice(j * B, (j + 1) * B)
            same = (torch.equal(codes[sl], c1), torch.equal(poses[sl], p1), torch.equal(gtc[sl], g1))
            if not all(same):
                raise RuntimeError(f"bench selfcheck FAILED for step {k} (batch {j} of the last pass of {len(ks)}): generated codes equal "
                                   f"{same[0]}, poses equal {same[1]}, VQ-encode codes equal {same[2]}")
            if not (torch.isfinite(p1).all() and int(c1.min()) >= 0 and int(c1.max()) < 2048 and int(g1.min()) >= 0 and int(g1.max()) < 2048):
                raise RuntimeError("bench selfcheck FAILED: outputs out of range")
            checked.append(int(k))
        return {"selfcheck": "ok", "graph_captures_in_timed_regions": caps, "selfcheck_what": f"steps {checked} of the last timed pass ({len(ks)} batches on stream {si}) re-run alone as "
                "one 32-clip batch: generated codes, poses and VQ-encode codes bit-equal"}

    def extras(self, out):
        """rank 0's measurement legs beside the headline; no process group is alive while these run"""
        a, w, lib, _lib, eng, B, T, G, S = self.a, self.w, self.lib, self._lib, self.eng, self.B, self.T, self.G, self.S
        mfcc, gt, ids, rank, pool, NB, H = self.mfcc, self.gt, self.ids, self.rank, self.pool, self.NB, self.H
        # the same workload under the other execution modes, for comparison (not the headline); N = 1 only: at N > 1 rank 0 keeps
        # to the roofline legs so that the job ends soon after the other ranks have left
        if not a.no_modes and self.world == 1:
            modes = {}
            one = Engine(w, lib, _lib, pool[:1], B, T, 1, mfcc, gt, ids, rank)
            one.warm(1)
            lat = timed(lambda: one.run_steps(1))
            modes["one_batch_in_flight"] = {"what": "strict: one 32-clip batch at a time, one stream (= latency of a batch)",
                                            "ms_per_step": lat * 1e3, "frames_per_s": B * FRAMES_PER_CLIP / lat}
            out["batch_latency_ms"] = lat * 1e3
            # the same ONE batch with its VQ-encode half (which feeds nothing downstream) on a second stream, under the latency-bound chain
            one2 = Engine(w, lib, _lib, pool[:1], B, T, 1, mfcc, gt, ids, rank, enc_streams=pool[1:2])
            one2.warm(1)
            lat2 = timed(lambda: one2.run_steps(1))
            modes["one_batch_in_flight_encode_on_a_side_stream"] = {
                "what": "strict: one 32-clip batch at a time; its VQ encode runs on a second stream beside audio encoder -> chain -> decode",
                "ms_per_step": lat2 * 1e3, "frames_per_s": B * FRAMES_PER_CLIP / lat2}
            r01 = Engine(w, lib, _lib, pool[:4], B, T, 1, mfcc, gt, ids, rank)
            r01.warm(1)
            t4 = timed(lambda: r01.run_steps(16)) / 16
            modes["four_streams_no_coalescing"] = {"what": "round-1 mode: 4 independent 32-clip batches on 4 HIP streams",
                                                   "ms_per_step": t4 * 1e3, "frames_per_s": B * FRAMES_PER_CLIP / t4}
            tg = timed(lambda: eng.run_steps(G * S)) / (G * S)
            modes["coalesced"] = {"what": f"headline mode re-measured: passes of {B * G} clips on {S} streams",
                                  "ms_per_step": tg * 1e3, "frames_per_s": B * FRAMES_PER_CLIP / tg,
                                  "latency_of_a_pass_ms": timed(lambda: eng.run_group(list(range(G)), 0)) * 1e3}
            # BASELINE's literal wording — "synthetic 16 kHz audio" in, poses out: the same passes with the device front-end (resample to
            # 22 kHz + MFCC) inside every pass, the waveforms resident in HBM
            try:
                from talkshow_amd import synth
                wv = [torch.from_numpy(synth.wav16(7100 + 10 * rank + k, B, 160000)).to(self.dev) for k in range(NB)]
                wi = Engine(w, lib, _lib, pool[:S], B, T, G, mfcc, gt, ids, rank, wav=wv)
                wi.warm(G * S)
                tw = timed(lambda: wi.run_steps(G * S)) / (G * S)
                modes["wav_in"] = {"what": f"coalesced mode from 16 kHz waveforms resident in HBM: resample to 22 kHz + MFCC(64) on the pass's stream, "
                                           "then the same pass (VQ encode of the GT poses, audio encoder -> PixelCNN greedy -> VQ decode)",
                                   "ms_per_step": tw * 1e3, "frames_per_s": B * FRAMES_PER_CLIP / tw, "vs_features_resident": tg / tw}
                del wi, wv
            except Exception as e:
                modes["wav_in"] = {"error": repr(e)}
            # the same passes with every step's inputs arriving from, and its outputs leaving to, pinned HOST memory inside the
            # timed region (the reference hands numpy arrays in and out: `value` is the resident-input figure, this is the other)
            try:
                pc = Engine(w, lib, _lib, pool[:S], B, T, G, mfcc, gt, ids, rank, pcie=True)
                pc.warm(G * S)
                tp = timed(lambda: pc.run_steps(G * S)) / (G * S)
                modes["pcie_inclusive"] = {"what": f"coalesced mode with per-pass H2D of the MFCC + GT-pose batches and D2H of poses + both code "
                                                   f"grids from / to pinned host memory on the pass's stream ({pc.pcie_bytes_per_step() / 1e6:.1f} MB per step)",
                                           "ms_per_step": tp * 1e3, "frames_per_s": B * FRAMES_PER_CLIP / tp, "vs_resident": tg / tp}
            except Exception as e:
                modes["pcie_inclusive"] = {"error": repr(e)}
            out["modes"] = modes

        if not a.no_roofline:
            big_mf = torch.cat([mfcc[k % NB] for k in range(G)], 0) if G > 1 else mfcc[0]
            out["roofline"] = chain_roofline(w, lib, _lib, eng.streams[0], big_mf, eng.ids_rep, H, f"skinny_gemm_f32_M{B * G}")
            if G > 1:
                out["roofline_one_batch"] = chain_roofline(w, lib, _lib, eng.streams[0], mfcc[0], ids, H, f"skinny_gemm_f32_M{B}")
            out["roofline_conv_gemm"] = conv_roofline(lib, _lib, self.local, lambda: eng.run_group(list(range(G)), 0))
        if not a.no_face and self.world == 1:
            for key, fn in (("diversity", lambda: diversity_block(w, _lib, mfcc[0])), ("frontend", lambda: frontend_block(w, _lib, B * G)),
                            ("face", lambda: face_block(self.local)), ("whole_body", lambda: whole_body_block(w, _lib, self.local, B * G))):
                try:
                    out[key] = fn()
                except Exception as e:                   # extra information; never lose the main line
                    out[key] = {"error": repr(e)}
        if self.world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(self.sds, 1000)


def run_contract(job, dist, world, rank, steps, warmup, clock=time.perf_counter, collectives=None, repeats=1, cpu_clock=time.process_time):
    """The bench contract's control flow, device-agnostic: W untimed warm-up steps, then EXACTLY `steps` steps bracketed by
    barrier + synchronize on both sides, the job's one exchange inside the bracket (N > 1), MAX over ranks.  Every collective
    of the job sits in here; the caller destroys the process group before rank 0 starts its extra measurement legs, so no rank
    ever waits in a collective for them.  Returns (seconds, compute seconds on this rank, exchange info or None).
    `repeats` > 1 times that same bracketed region `repeats` times back to back in the same warm state (each one EXACTLY `steps`
    steps, each with its own barriers and MAX over ranks) and returns the MEDIAN region as `seconds`; every region's time and this
    rank's host CPU seconds inside it (whole region, and the part until `run_steps` returned = queueing) land in `job.contract_runs` =
    {"runs_ms": [...], "host_cpu_s": [...], "host_enqueue_cpu_s": [...], "host_enqueue_wall_s": [...], "median_index": i}
    (one 0.12 s shot cannot adjudicate a 2 % change).
    `collectives=True` takes the N > 1 code path at world 1 too (TS_BENCH_FORCE_COLLECTIVES=1: the RCCL plumbing check that one
    metered GPU allows — process group on the `nccl` backend, barriers, the all-gather, the MAX all-reduce)."""
    coll = world > 1 if collectives is None else collectives

    def barrier():
        if coll:
            dist.barrier()

    job.warm(steps)
    job.run_steps(max(warmup, 1))            # W untimed steps (passes of sizes the warm-up above has seen or captures now)
    job.sync()
    if coll:
        job.gather()                         # the exchange once untimed: RCCL sets up its rings / buffers on first use
        job.sync()
    runs = []
    begin, end = getattr(job, "region_begin", None), getattr(job, "region_end", None)
    for _ in range(max(1, int(repeats))):
        if begin:
            begin()                          # (outside the bracket) e.g. the job notes how many hipGraphs exist
        barrier()
        c0 = cpu_clock()
        t0 = clock()
        job.run_steps(steps)
        cpu_enq, t_enq = cpu_clock() - c0, clock() - t0      # the launching thread is back: everything is queued (the sync below busy-waits)
        job.sync()
        t_compute = clock() - t0
        info = None
        if coll:
            info = job.gather()
            job.sync()
            info.update({"ranks_seen": world, "gather_ms": (clock() - t0 - t_compute) * 1e3, "compute_ms": t_compute * 1e3,
                         "note": "rank 0's clock; the headline takes the max over ranks of compute + gather"})
        barrier()
        dt = clock() - t0
        cpu = cpu_clock() - c0
        if coll:
            tmax = job.scalar(dt)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        runs.append((dt, t_compute, info, cpu, cpu_enq, t_enq))
        if end:
            end()
    order = sorted(range(len(runs)), key=lambda i: runs[i][0])
    mid = order[len(order) // 2]             # the median region (the upper one of an even count); every rank picks the same index
    job.contract_runs = {"runs_ms": [r[0] * 1e3 for r in runs], "host_cpu_s": [r[3] for r in runs], "host_enqueue_cpu_s": [r[4] for r in runs],
                         "host_enqueue_wall_s": [r[5] for r in runs], "median_index": mid}
    return runs[mid][0], runs[mid][1], runs[mid][2]


def finish(job, dist, world, rank, steps, warmup, dt, info, emit=print, collectives=None):
    """After the timed region: every rank checks what it timed (local work), the ranks leave the process group TOGETHER, and only
    then does rank 0 run its extra legs and print the one JSON line.  Ranks != 0 run nothing else.  A rank whose check fails does
    not leave the others waiting in a barrier: the failure is all-reduced first, every rank leaves the group, every rank raises."""
    coll = world > 1 if collectives is None else collectives
    failure = None
    try:
        check = job.selfcheck()
    except Exception as e:                      # noqa: BLE001 — re-raised below, after the group has been left together
        failure, check = e, None
    if coll:
        bad = job.scalar(1.0 if failure is not None else 0.0)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        dist.barrier()
        dist.destroy_process_group()
        if failure is None and float(bad.item()) > 0:
            failure = RuntimeError("selfcheck failed on another rank")
    if failure is not None:
        raise failure
    if rank != 0:
        return None
    frames = world * steps * job.frames_per_step()
    out = {
        "metric": "generated SMPL-X frames/sec (10 s @ 30 fps clips), whole job",
        "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (seeded MFCC-scale features / poses, random-init weights of the reference architecture)",
        "config": job.describe(),
        "per_gpu_frames_per_s": frames / dt / world,
        "rccl": info,      # N > 1: the job's one all-gather, timed apart from the compute (null at N = 1); unmeasured on hardware until SCALE runs
    }
    cr = getattr(job, "contract_runs", None)
    if cr:
        # `value` / `ms_per_step` are the MEDIAN of these timed regions (each exactly `steps` steps, same warm state); host_cpu_s =
        # time.process_time() of this rank's process over the median region (the launching thread + the runtime's helper threads)
        out["runs_ms"] = cr["runs_ms"]
        out["runs_spread"] = (max(cr["runs_ms"]) - min(cr["runs_ms"])) / cr["runs_ms"][cr["median_index"]]
        out["host_cpu_s"] = cr["host_cpu_s"][cr["median_index"]]
        out["host_cpu_per_wall"] = out["host_cpu_s"] / dt
        # ... of which until run_steps() returned, i.e. queueing the region's work (launches, graph replays, device copies); the rest is the
        # runtime's busy-wait inside the closing synchronize.  This is the figure eight ranks on one host compete with.
        out["host_enqueue_cpu_s"] = cr["host_enqueue_cpu_s"][cr["median_index"]]
        out["host_enqueue_wall_s"] = cr["host_enqueue_wall_s"][cr["median_index"]]
    if getattr(job, "host_affinity", None) is not None:
        out["host_affinity"] = job.host_affinity
    out.update(check)
    # whole path against the fp32 MFMA roof: algorithmic work of configs[1] (64.25 MFLOP per generated frame)
    ach = ALG_FLOP_PER_FRAME * frames / dt / world / 1e12
    out["whole_path"] = {"algorithmic_TFLOPs_per_gpu": ach, "frac_of_fp32_mfma_peak": ach / PEAK_FP32_MFMA_TFLOPS}
    job.extras(out)
    emit(json.dumps(out))
    return out


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_ranks(n, script, script_args, env=None, timeout=None):
    """`python bench.py --gpus N` WITHOUT a launcher (WORLD_SIZE unset): start the N ranks ourselves through
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` — the command the
    driver uses — and hand back its exit code (the assert that stood here ended a launcher-less SCALE run before
    it touched a GPU).  Rank 0's JSON line goes to this process's stdout unchanged."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(script_args)
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes fails without it on this driver
    e["TS_BENCH_SPAWNED"] = "1"
    return subprocess.run(cmd, env=e, timeout=timeout).returncode


def needs_spawn(environ, gpus):
    """More than one GPU asked for and no launcher's environment: this process is not a rank, it has to start the ranks."""
    return gpus > 1 and "WORLD_SIZE" not in environ and "RANK" not in environ


def pin_to_gpu_numa(local):
    """Best effort: keep this rank's host threads on the CPUs of the NUMA node its GPU hangs off (8 ranks x (launch thread + ROCr
    helper threads) otherwise wander over both sockets).  -> description for the JSON line, or the reason nothing was done."""
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return {"pinned": False, "why": f"{bdf}: numa_node = {node} (single-node host or not exposed)"}
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"pinned": False, "why": f"node {node} has no CPU this process may use"}
        os.sched_setaffinity(0, cpus)
        return {"pinned": True, "gpu": bdf, "numa_node": node, "cpus": len(cpus)}
    except Exception as e:                                   # noqa: BLE001 — affinity is an optimisation, never a reason to fail
        return {"pinned": False, "why": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--coalesce", type=int, default=int(os.environ.get("TS_BENCH_COALESCE", "8")),
                    help="submitted batches stacked into one pass (clips per chain stage = batch * coalesce)")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("TS_BENCH_STREAMS", "3")),
                    help="groups in flight at once, one HIP stream each")
    ap.add_argument("--enc-streams", type=int, default=int(os.environ.get("TS_BENCH_ENC_STREAMS", "0")),
                    help="run the VQ-encode half of each pass on this many extra streams (0 = on the pass's own stream)")
    ap.add_argument("--config", default="body", choices=["body", "whole_body"])
    ap.add_argument("--clips-per-rank", type=int, default=128, help="whole_body: clips per rank per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-face", action="store_true")
    ap.add_argument("--no-modes", action="store_true")
    ap.add_argument("--repeats", type=int, default=int(os.environ.get("TS_BENCH_REPEATS", "3")),
                    help="timed regions of exactly --steps steps each, back to back in the same warm state; the line reports their median")
    a = ap.parse_args()

    if needs_spawn(os.environ, a.gpus):
        sys.exit(spawn_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    torch.cuda.set_device(local)
    affinity = pin_to_gpu_numa(local) if (world > 1 or os.environ.get("TS_BENCH_PIN", "0") == "1") else {"pinned": False, "why": "one rank"}
    import torch.distributed as dist
    # TS_BENCH_FORCE_COLLECTIVES=1 (tools/rccl_smoke.sh): the N > 1 code path — RCCL process group, barriers, the all-gather of every
    # step's rows, the MAX all-reduce — at world 1, which is all one metered GPU allows; the line then carries an `rccl` block
    coll = world > 1 or os.environ.get("TS_BENCH_FORCE_COLLECTIVES", "0") == "1"
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if a.config == "whole_body":
        main_whole_body(a, world, rank, local, dist)
        if coll:
            dist.barrier()
            dist.destroy_process_group()
        return

    job = BodyJob(a, world, rank, local)
    job.host_affinity = affinity
    dt, _, info = run_contract(job, dist, world, rank, a.steps, a.warmup, collectives=coll, repeats=a.repeats)
    finish(job, dist, world, rank, a.steps, a.warmup, dt, info, emit=lambda line: print(line, flush=True), collectives=coll)


if __name__ == "__main__":
    main()
