#!/usr/bin/env python3
"""Time the REFERENCE's own modules on this container's host cores.

Runs only where /root/reference exists (the build container; not the GPU box).  The reference `nets` package is imported
through the shim of tests/golden/make_golden.py (absent third-party modules stubbed, nothing written to the reference
tree), loaded with the same seeded synthetic checkpoints the parity goldens use, and timed exactly as a caller would
use it:

  body  nets.s2g_body_pixel(args, config).infer_on_audio(wav, id=tensor([k]), fps=30, B=32)   -> (32, 300, 129)
        (`get_mfcc_ta` patched to return resident (300, 64) features: torchaudio is absent, and the front-end is outside
        the timed hot path on the GPU side as well).  This is the reference algorithm as shipped: full-grid recompute per
        code position, softmax + multinomial.
  face  nets.s2g_face(args, config).generate(wav (B,1,160000), 300), B = 8                     -> (8, 300, 103)

1 warm-up + 3 timed runs each, median.  Output: one JSON document on stdout / --out (committed under profiles/ and
printed by bench.py as cpu_baseline.reference_build_box).
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--body-batch", type=int, default=32)
    ap.add_argument("--face-batch", type=int, default=8)
    ap.add_argument("--runs", type=int, default=3)
    a = ap.parse_args()

    import make_golden as MG
    sys.path.insert(0, REPO)
    from talkshow_amd import synth
    sys.path.remove(REPO)
    nets = MG.import_reference()
    import tempfile
    import torch
    from trainer.config import Object
    torch.set_num_threads(os.cpu_count())
    T = synth.to_torch
    res = {"where": "build container (no GPU)", "cpu": cpu_model(), "cores": os.cpu_count(),
           "torch_threads": torch.get_num_threads(), "torch": torch.__version__,
           "weights": "seeded synthetic checkpoints of talkshow_amd/synth.py (seed 0), reference architecture",
           "protocol": f"1 warm-up + {a.runs} runs, median, time.perf_counter"}

    # ---- body: the reference wrapper, batch 32 x 10 s ----
    tmp = tempfile.mkdtemp(prefix="ts_refcpu_")
    vq_path = os.path.join(tmp, "vq.pth")
    torch.save({"generator": {"g_body": T(synth.vqvae_state_dict(seed=0, in_dim=39)),
                              "g_hand": T(synth.vqvae_state_dict(seed=0, in_dim=90, salt=1))}}, vq_path)
    cfg = json.load(open(os.path.join(MG.REF, "config/body_pixel.json")))
    cfg["Model"]["vq_path"] = vq_path
    targs = argparse.Namespace(gpu="cpu", infer=True)
    w = MG.quiet(nets.s2g_body_pixel, targs, Object(cfg))
    w.load_state_dict({"generator": T(synth.pixelcnn_state_dict(seed=0)), "audioencoder": T(synth.audioencoder_state_dict(seed=0))})
    feat = synth.mfcc_features(1000, 1, 300)[0]                       # (300, 64): what get_mfcc_ta returns for a 10 s clip
    import nets.smplx_body_pixel as sbp
    sbp.get_mfcc_ta = lambda *args, **kw: feat
    B = a.body_batch

    def body_run():
        t0 = time.perf_counter()
        out = w.infer_on_audio("synthetic.wav", id=torch.tensor([0]), fps=30, B=B)
        dt = time.perf_counter() - t0
        assert out.shape == (B, 300, 129), out.shape
        return dt
    body_run()
    ts = sorted(body_run() for _ in range(a.runs))
    dt = ts[len(ts) // 2]
    res["body"] = {"entry": "nets.s2g_body_pixel.TrainWrapper.infer_on_audio (smplx_body_pixel.py:232-289), multinomial sampling",
                   "batch": B, "frames": B * 300, "seconds_median": dt, "seconds_all": ts, "frames_per_s": B * 300 / dt,
                   "note": "audio encoder + full-grid GatedPixelCNN.generate + two VQ decoders; the VQ-encode half of "
                           "BASELINE configs[1] is not part of this entry point"}
    print(json.dumps(res["body"]), file=sys.stderr, flush=True)

    # ---- face: the reference wrapper's batched tensor entry ----
    fcfg = json.load(open(os.path.join(MG.REF, "config/face.json")))
    wf = MG.quiet(nets.s2g_face, targs, Object(fcfg))
    wf.load_state_dict({"generator": T(synth.face_state_dict(seed=0))})
    Bf = a.face_batch
    wav = torch.from_numpy(synth.wav16(3000, Bf, 160000))[:, None, :]

    def face_run():
        t0 = time.perf_counter()
        out = wf.generate(wav, 300)
        dt = time.perf_counter() - t0
        assert tuple(out.shape) == (Bf, 300, 103)
        return dt
    face_run()
    ts = sorted(face_run() for _ in range(a.runs))
    dt = ts[len(ts) // 2]
    res["face"] = {"entry": "nets.s2g_face.TrainWrapper.generate (smplx_face.py:221-238)", "batch": Bf, "frames": Bf * 300,
                   "seconds_median": dt, "seconds_all": ts, "frames_per_s": Bf * 300 / dt}
    s = json.dumps(res, indent=1)
    print(s)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(os.path.join(REPO, a.out))), exist_ok=True)
        open(os.path.join(REPO, a.out), "w").write(s + "\n")


if __name__ == "__main__":
    main()
