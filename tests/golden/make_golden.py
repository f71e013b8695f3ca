#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container, where /root/reference exists (it does not on the GPU box):

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (`/root/reference`, pure PyTorch) is imported through the shim of SURVEY.md
Appendix C (third-party modules that are absent here and unused on the hot path are stubbed;
nothing is written to the reference tree).  Weights are the seeded synthetic checkpoints of
`talkshow_amd/synth.py`, loaded with `strict=True`, so the key names / shapes those builders
emit are checked against the reference here.  Inputs are seeded too; inputs AND outputs are
stored so the CPU oracle (`oracle/`) and the HIP path can both be compared on identical data.

Greedy decode: the reference has none (`gated_pixelcnn_v2.py:173-176` always samples), so — as
SURVEY.md §0.3 prescribes — the harness below drives the reference `GatedPixelCNN.forward`
position by position and takes `torch.argmax(logits[:, :, i, j], -1)`.
"""
import argparse
import contextlib
import importlib.machinery
import io
import json
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def import_reference():
    sys.dont_write_bytecode = True
    import torch  # noqa: F401
    import transformers  # noqa: F401  (must precede the stubs)
    from transformers import Wav2Vec2Config, Wav2Vec2Processor  # noqa: F401

    def stub(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__path__ = []
        sys.modules[name] = m
        return m

    for n in ["torchvision", "torchvision.datasets", "torchvision.transforms", "torchaudio",
              "torchaudio.functional", "torchaudio.transforms", "torchaudio.sox_effects", "librosa",
              "python_speech_features", "textgrid", "smplx"]:
        stub(n)
    sys.modules["torchaudio.sox_effects"].apply_effects_tensor = None
    os.chdir(REF)
    # the repo root also holds a package called `nets` (our drop-in); the reference's must win here
    sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
    sys.path.insert(0, REF)
    import nets
    assert nets.__file__.startswith(REF), nets.__file__
    from nets.spg import wav2vec as w2
    w2.Wav2Vec2Model.from_pretrained = classmethod(
        lambda cls, *a, **k: cls(Wav2Vec2Config(attn_implementation="eager")))
    return nets


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def greedy_reference(pix, label, aud, H):
    """Greedy harness around the reference forward; returns codes and per-step logits."""
    import torch
    B = aud.shape[0]
    x = torch.zeros((B, H, 2), dtype=torch.int64)
    step_logits = torch.zeros((B, H, 2, pix.embedding.weight.shape[0]))
    with torch.no_grad():
        for i in range(H):
            for j in range(2):
                logits = pix(x, label, aud)
                step_logits[:, i, j] = logits[:, :, i, j]
                x[:, i, j] = torch.argmax(logits[:, :, i, j], dim=-1)
    return x, step_logits


def margins(step_logits):
    top2 = np.sort(step_logits.reshape(-1, step_logits.shape[-1]), axis=-1)[:, -2:]
    return top2[:, 1] - top2[:, 0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()

    sys.path.insert(0, REPO)
    from talkshow_amd import synth
    from talkshow_amd import frontend as fe          # the host twin of the third-party front-end (real-audio cases only)
    sys.path.remove(REPO)
    nets = import_reference()
    import torch
    from nets.spg.gated_pixelcnn_v2 import GatedPixelCNN
    from nets.spg.vqvae_1d import VQVAE, AudioEncoder

    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    T = synth.to_torch
    meta = {"torch": torch.__version__, "numpy": np.__version__, "cases": {}}

    def save(name, **arrs):
        if args.only and args.only != name:
            return
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
        meta["cases"][name] = {k: list(np.asarray(v).shape) for k, v in arrs.items()}
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")

    def want(name):
        return not args.only or args.only == name

    # ---- 1. VQ-VAE, small and full architecture: encode -> argmin -> gather -> decode -----------
    for name, kw, B, Tn in [
        ("vq_small", dict(in_dim=39, num_embeddings=128, num_hiddens=128), 3, 24),
        ("vq_full_body", dict(in_dim=39, num_embeddings=2048, num_hiddens=1024), 2, 40),
        ("vq_full_hand", dict(in_dim=90, num_embeddings=2048, num_hiddens=1024, salt=1), 2, 40),
    ]:
        if not want(name):
            continue
        sd = synth.vqvae_state_dict(seed=7, **kw)
        net = VQVAE(kw["in_dim"], 64, kw["num_embeddings"], kw["num_hiddens"], 2, 512)
        net.load_state_dict(T(sd), strict=True)
        net.eval()
        poses = synth.gt_poses(11, B, Tn, dim=kw["in_dim"])
        with torch.no_grad():
            z = net.encoder(torch.from_numpy(poses).transpose(1, 2))        # (B,64,H)
            e, idx = net.vq_layer(z)                                        # eval branch
            e2, recon = net(gt_poses=torch.from_numpy(poses))               # VQVAE.forward eval
            dec, _ = net.decode(b=B, w=idx.shape[1], latents=idx)
        assert torch.equal(e, e2) and torch.equal(dec, recon)
        print(name, "z std", float(z.std()), "recon std", float(recon.std()), "uniq codes", idx.unique().numel())
        save(name, poses=poses, z=z.numpy(), idx=idx.numpy(), quantized=e.numpy(), recon=recon.numpy(),
             cfg=np.asarray([kw["in_dim"], 64, kw["num_embeddings"], kw["num_hiddens"], 2, kw.get("salt", 0), 7]))

    # ---- 2. audio encoder ------------------------------------------------------------------------
    if want("audioenc_full"):
        sd = synth.audioencoder_state_dict(seed=7)
        net = AudioEncoder(64, 256, 2, 256)
        net.load_state_dict(T(sd), strict=True)
        net.eval()
        mf = synth.mfcc_features(12, 2, 60)
        with torch.no_grad():
            out = net(torch.from_numpy(mf).transpose(1, 2))
        print("audioenc out std", float(out.std()))
        save("audioenc_full", mfcc=mf, out=out.numpy())

    # ---- 3. PixelCNN greedy, small and full architecture -----------------------------------------
    for name, kw, B, H in [
        ("pix_small", dict(input_dim=128, dim=64, n_layers=4), 3, 10),
        ("pix_full", dict(input_dim=2048, dim=256, n_layers=15), 2, 12),
    ]:
        if not want(name):
            continue
        sd = synth.pixelcnn_state_dict(seed=7, **kw)
        pix = quiet(GatedPixelCNN, kw["input_dim"], kw["dim"], kw["n_layers"], 4, True, True)
        pix.load_state_dict(T(sd), strict=True)
        pix.eval()
        rng = np.random.default_rng(13)
        aud = rng.standard_normal((B, H, 256)).astype(np.float32)           # (B,H,256) row features
        label = synth.speaker_ids(B)
        aud_t = torch.from_numpy(aud).permute(0, 2, 1).unsqueeze(-1).repeat(1, 1, 1, 2)
        codes, step_logits = greedy_reference(pix, torch.from_numpy(label), aud_t, H)
        with torch.no_grad():
            full_logits = pix(codes, torch.from_numpy(label), aud_t)        # teacher-forced full grid
        m = margins(step_logits.numpy())
        print(name, "logit std", float(step_logits.std()), "margin min/median", float(m.min()), float(np.median(m)),
              "uniq", codes.unique().numel())
        save(name, aud=aud, label=label, codes=codes.numpy(), step_logits=step_logits.numpy().astype(np.float32),
             full_logits=full_logits.permute(0, 2, 3, 1).numpy(), margin=m,
             cfg=np.asarray([kw["input_dim"], kw["dim"], kw["n_layers"], 4, 7]))

    # ---- 3b. the other constructor variants of GatedPixelCNN (`gated_pixelcnn_v2.py:90-128`): audio=False and / or bh_model=False
    # (single vertical stack, columns never mix; grid widths 2 and 4).  No shipped config uses them; same greedy harness.
    if want("pix_variants"):
        out = {}
        for tag, audio, bh, Wd in (("noaud_bh", False, True, 2), ("aud_v", True, False, 2), ("noaud_v", False, False, 4)):
            kw = dict(input_dim=128, dim=64, n_layers=4)
            sd = synth.pixelcnn_state_dict(seed=11, audio=audio, bh_model=bh, **kw)
            pix = quiet(GatedPixelCNN, kw["input_dim"], kw["dim"], kw["n_layers"], 4, audio, bh)
            pix.load_state_dict(T(sd), strict=True)                       # pins the variant's key set / shapes
            pix.eval()
            B, H = 3, 8
            rng = np.random.default_rng(17)
            aud = rng.standard_normal((B, H, 256)).astype(np.float32)
            label = synth.speaker_ids(B)
            aud_t = torch.from_numpy(aud).permute(0, 2, 1).unsqueeze(-1).repeat(1, 1, 1, Wd) if audio else None
            x = torch.zeros((B, H, Wd), dtype=torch.int64)
            step_logits = torch.zeros((B, H, Wd, kw["input_dim"]))
            with torch.no_grad():
                for i in range(H):
                    for j in range(Wd):
                        lg = pix(x, torch.from_numpy(label), aud_t) if audio else pix(x, torch.from_numpy(label))
                        step_logits[:, i, j] = lg[:, :, i, j]
                        x[:, i, j] = torch.argmax(lg[:, :, i, j], dim=-1)
                full = pix(x, torch.from_numpy(label), aud_t) if audio else pix(x, torch.from_numpy(label))
            m = margins(step_logits.numpy())
            print("pix_variants", tag, "margin min/median", float(m.min()), float(np.median(m)), "uniq", x.unique().numel())
            out.update({f"{tag}_codes": x.numpy(), f"{tag}_step_logits": step_logits.numpy(), f"{tag}_full_logits": full.permute(0, 2, 3, 1).numpy()})
        save("pix_variants", aud=aud, label=label, cfg=np.asarray([128, 64, 4, 4, 11]), **out)

    # ---- 4. end-to-end through the reference WRAPPERS (ckpt plumbing included) --------------------
    if want("body_e2e_full") or want("body_vq_e2e_full"):
        tmp = tempfile.mkdtemp(prefix="ts_golden_")
        vq_path = os.path.join(tmp, "vq.pth")
        body_sd = synth.vqvae_state_dict(seed=7, in_dim=39)
        hand_sd = synth.vqvae_state_dict(seed=7, in_dim=90, salt=1)
        torch.save({"generator": {"g_body": T(body_sd), "g_hand": T(hand_sd)}}, vq_path)
        cfg = json.load(open(os.path.join(REF, "config/body_pixel.json")))
        cfg["Model"]["vq_path"] = vq_path
        from trainer.config import Object
        config = Object(cfg)
        targs = argparse.Namespace(gpu="cpu", infer=True)

        if want("body_e2e_full"):
            B, Tn = 2, 300
            w = quiet(nets.s2g_body_pixel, targs, config)
            ckpt = {"generator": T(synth.pixelcnn_state_dict(seed=7)),
                    "audioencoder": T(synth.audioencoder_state_dict(seed=7))}
            # exactly what scripts/demo.py:58-59 does with ckpt['generator']
            w.load_state_dict({"generator": {("module." + k): v for k, v in ckpt["generator"].items()},
                               "audioencoder": ckpt["audioencoder"]})
            mf = synth.mfcc_features(21, B, Tn)
            ids = synth.speaker_ids(B)
            w.generator.eval(); w.g_body.eval(); w.g_hand.eval(); w.audioencoder.eval()
            with torch.no_grad():
                feat = w.audioencoder(torch.from_numpy(mf).transpose(1, 2), frame_num=0)   # (B,256,H)
                aud = feat.unsqueeze(-1).repeat(1, 1, 1, 2)
                H = aud.shape[2]
                codes, step_logits = greedy_reference(w.generator, torch.from_numpy(ids), aud, H)
                body, _ = w.g_body.decode(b=B, w=H, latents=codes[..., 0])
                hand, _ = w.g_hand.decode(b=B, w=H, latents=codes[..., 1])
                poses = torch.cat([body, hand], dim=1).transpose(1, 2)
            m = margins(step_logits.numpy())
            print("body_e2e_full: H", H, "margin min/median", float(m.min()), float(np.median(m)),
                  "pose std", float(poses.std()), "uniq codes", codes.unique().numel())
            save("body_e2e_full", mfcc=mf, ids=ids, aud_feat=feat.permute(0, 2, 1).numpy(), codes=codes.numpy(),
                 poses=poses.numpy(), margin=m)

        if want("body_vq_e2e_full"):
            B, Tn = 2, 300
            vcfg = json.load(open(os.path.join(REF, "config/body_vq.json")))
            vconfig = Object(vcfg)
            w = quiet(nets.s2g_body_vq, targs, vconfig)
            w.load_state_dict({"g_body": T(body_sd), "g_hand": T(hand_sd)})
            from data_utils.lower_body import c_index_3d
            full = np.zeros((B, 165, Tn), np.float32)
            p129 = synth.gt_poses(22, B, Tn)
            full[:, c_index_3d, :] = p129.transpose(0, 2, 1)
            out = quiet(w.infer_on_audio, torch.zeros(B, 64, Tn), initial_pose=torch.from_numpy(full),
                        id=torch.tensor([0]), fps=30)
            with torch.no_grad():
                _, lat_b = w.g_body.encode(gt_poses=torch.from_numpy(p129[..., :39]))
                _, lat_h = w.g_hand.encode(gt_poses=torch.from_numpy(p129[..., 39:]))
            print("body_vq_e2e_full: out", out.shape, "std", float(out.std()))
            save("body_vq_e2e_full", poses129=p129, out=out, codes=np.stack([lat_b.numpy(), lat_h.numpy()], -1),
                 c_index=np.asarray(c_index_3d))

    # ---- 4b. BASELINE operating point: the greedy harness on a whole batch of 32 ten-second clips (4 800 decisions) ------------
    # (every batch test embedded the same two body_e2e_full clips.)  ~2-4 min on 8 cores.
    if args.only == "body_e2e_b32" or (not args.only and os.environ.get("TS_GOLDEN_BIG")):
        tmp = tempfile.mkdtemp(prefix="ts_golden_")
        vq_path = os.path.join(tmp, "vq.pth")
        torch.save({"generator": {"g_body": T(synth.vqvae_state_dict(seed=7, in_dim=39)),
                                  "g_hand": T(synth.vqvae_state_dict(seed=7, in_dim=90, salt=1))}}, vq_path)
        cfg = json.load(open(os.path.join(REF, "config/body_pixel.json")))
        cfg["Model"]["vq_path"] = vq_path
        from trainer.config import Object
        w = quiet(nets.s2g_body_pixel, argparse.Namespace(gpu="cpu", infer=True), Object(cfg))
        w.load_state_dict({"generator": T(synth.pixelcnn_state_dict(seed=7)), "audioencoder": T(synth.audioencoder_state_dict(seed=7))})
        B, Tn, mf_seed = 32, 300, 23
        mf, ids = synth.mfcc_features(mf_seed, B, Tn), synth.speaker_ids(B)
        w.generator.eval(); w.g_body.eval(); w.g_hand.eval(); w.audioencoder.eval()
        with torch.no_grad():
            feat = w.audioencoder(torch.from_numpy(mf).transpose(1, 2), frame_num=0)
            aud = feat.unsqueeze(-1).repeat(1, 1, 1, 2)
            H = aud.shape[2]
            codes, step_logits = greedy_reference(w.generator, torch.from_numpy(ids), aud, H)
            keep = np.asarray([0, 9, 18, 31])
            body, _ = w.g_body.decode(b=len(keep), w=H, latents=codes[keep][..., 0])
            hand, _ = w.g_hand.decode(b=len(keep), w=H, latents=codes[keep][..., 1])
            poses = torch.cat([body, hand], dim=1).transpose(1, 2)
        m = margins(step_logits.numpy()).reshape(B, H, 2)
        print("body_e2e_b32: margin min/median", float(m.min()), float(np.median(m)), "uniq codes", codes.unique().numel(),
              "decisions under 1e-3:", int((m < 1e-3).sum()))
        save("body_e2e_b32", mfcc_seed=np.asarray([mf_seed, B, Tn]), ids=ids, codes=codes.numpy().astype(np.int16),
             margin=m.astype(np.float32), pose_clips=keep, poses=poses.numpy())

    # ---- 4c. the VQ-encode half of configs[1] at its operating point: 32 clips x 300 GT frames through the reference's
    # VQVAE.encode (vqvae_1d.py:196-199 -> vqvae_modules.py:274-286,311-319), body and hand, with a codebook that the encoder's
    # outputs actually SPREAD over: the default synthetic codebook attracts 9 / 30 of 2 048 entries, so the
    # entries are re-drawn around the per-channel mean / spread of z (`synth.vqvae_state_dict(codebook=(mu, sigma))`, stored here)
    if want("vq_encode_b32"):
        B, Tn, gt_seed = 32, 300, 24
        p129 = synth.gt_poses(gt_seed, B, Tn)
        out = {}
        for part, in_dim, salt, sl in (("body", 39, 0, slice(0, 39)), ("hand", 90, 1, slice(39, 129))):
            x = torch.from_numpy(np.ascontiguousarray(p129[..., sl]))
            net = VQVAE(in_dim, 64, 2048, 1024, 2, 512)
            net.load_state_dict(T(synth.vqvae_state_dict(seed=7, in_dim=in_dim, salt=salt)), strict=True)
            net.eval()
            with torch.no_grad():
                z = net.encoder(x.transpose(1, 2)).permute(0, 2, 1).reshape(-1, 64).numpy().astype(np.float64)
            mu, sigma = z.mean(0).astype(np.float32), z.std(0).astype(np.float32)
            sd = synth.vqvae_state_dict(seed=7, in_dim=in_dim, salt=salt, codebook=(mu, sigma))
            net.load_state_dict(T(sd), strict=True)
            with torch.no_grad():
                e, lat = net.encode(gt_poses=x)                                  # what bench.py's encode half computes
                zt = net.encoder(x.transpose(1, 2)).permute(0, 2, 1).reshape(-1, 64)
                d = ((zt ** 2).sum(1, keepdim=True) + (net.vq_layer.embeddings ** 2).sum(1)
                     - 2.0 * zt @ net.vq_layer.embeddings.t())                  # get_code_indices' own expression
                top2 = torch.topk(d, 2, dim=1, largest=False).values
                assert torch.equal(d.argmin(1).view(B, -1), lat)
                rec, _ = net.decode(b=2, w=lat.shape[1], latents=lat[:2])
            mg = (top2[:, 1] - top2[:, 0]).view(B, -1).numpy()
            print(f"vq_encode_b32 {part}: distinct codes {lat.unique().numel()} margin min/median {mg.min():.2e} {np.median(mg):.3f}")
            out.update({f"mu_{part}": mu, f"sigma_{part}": sigma, f"codes_{part}": lat.numpy().astype(np.int16),
                        f"margin_{part}": mg.astype(np.float32), f"recon2_{part}": rec.numpy()})
        save("vq_encode_b32", gt_seed=np.asarray([gt_seed, B, Tn]), **out)

    # ---- 4d. convert_to_6d=true through the reference wrappers s2g_body_vq (78 + 180 modelled dims, c_index_6d over 330-wide rows,
    # `smplx_body_vq.py:50-53`) and s2g_body_ae (`body_ae.py:50-53`): no shipped config uses it, the wrappers support it
    if want("wrappers_6d"):
        from trainer.config import Object
        from data_utils.lower_body import c_index_6d
        targs = argparse.Namespace(gpu="cpu", infer=True)
        B, Tn = 2, 24
        vcfg = json.load(open(os.path.join(REF, "config/body_vq.json")))
        vcfg["Data"]["pose"]["convert_to_6d"] = True
        w = quiet(nets.s2g_body_vq, targs, Object(vcfg))
        sd_b = synth.vqvae_state_dict(seed=9, in_dim=78)
        sd_h = synth.vqvae_state_dict(seed=9, in_dim=180, salt=1)
        w.load_state_dict({"g_body": T(sd_b), "g_hand": T(sd_h)})
        p258 = synth.gt_poses(26, B, Tn, dim=258)
        full = np.zeros((B, 330, Tn), np.float32)
        full[:, c_index_6d, :] = p258.transpose(0, 2, 1)
        out = quiet(w.infer_on_audio, torch.zeros(B, 64, Tn), initial_pose=torch.from_numpy(full), id=torch.tensor([0]), fps=30)
        with torch.no_grad():
            _, lat_b = w.g_body.encode(gt_poses=torch.from_numpy(np.ascontiguousarray(p258[..., :78])))
            _, lat_h = w.g_hand.encode(gt_poses=torch.from_numpy(np.ascontiguousarray(p258[..., 78:])))
        cfg = json.load(open(os.path.join(REF, "config/body_pixel.json")))
        cfg["Data"]["pose"]["convert_to_6d"] = True
        a = quiet(nets.s2g_body_ae, targs, Object(cfg))
        a.load_state_dict({"g": T(synth.ae_state_dict(seed=9, in_dim=258))})
        wide = np.zeros((B, Tn, 330), np.float32)
        wide[:, :, c_index_6d] = p258
        a.g.eval()
        with torch.no_grad():
            feat, x258 = a.extract(torch.from_numpy(wide))
        assert torch.equal(x258, torch.from_numpy(p258))
        print("wrappers_6d: vq out", out.shape, "std", float(out.std()), "ae feat", tuple(feat.shape), "std", float(feat.std()))
        save("wrappers_6d", poses258=p258, vq_out=out, vq_codes=np.stack([lat_b.numpy(), lat_h.numpy()], -1), ae_feat=feat.numpy(),
             c_index=np.asarray(c_index_6d))

    # ---- 5. face generator over the installed transformers wav2vec2 (reference: s2g_face.Generator + wav2vec.py) ----
    if want("face_full"):
        fcfg = json.load(open(os.path.join(REF, "config/face.json")))
        from trainer.config import Object
        fconfig = Object(fcfg)
        targs = argparse.Namespace(gpu="cpu", infer=True)
        w = quiet(nets.s2g_face, targs, fconfig)
        fsd = synth.face_state_dict(seed=7)
        w.load_state_dict({"generator": T(fsd)})            # TrainWrapperBaseClass.load_state_dict, strict
        B, N = 2, 32000                                      # 2 s of 16 kHz audio -> 60 frames
        wav = synth.wav16(31, B, N)
        frame = N * 30 // 16000
        w.generator.eval()
        with torch.no_grad():
            ids = torch.zeros(B, 4); ids[1, 2] = 1.0         # the all-zero id of smplx_face.py:206 and a one-hot
            out = w.generator(torch.from_numpy(wav)[:, None, :], None, ids, time_steps=frame)[0]
            hs = w.generator.audio_encoder(torch.from_numpy(wav), frame_num=frame).last_hidden_state
            gen = w.generate(torch.from_numpy(wav)[:, None, :], frame)      # smplx_face.py:221-238 (zero id)
        print("face_full: out", tuple(out.shape), "std", float(out.std()), "hidden std", float(hs.std()))
        save("face_full", wav=wav, ids=ids.numpy(), out=out.numpy(), hidden=hs.numpy(), generate_zero_id=gen.numpy())

    # ---- 5a. the face wrapper with convert_to_6d=true: Generator(identity=False) — no id channels, 6-wide jaw head (`smplx_face.py:37-45`,
    # `s2g_face.py:107-113`); 2 s clips like face_full
    if want("face_6d"):
        fcfg = json.load(open(os.path.join(REF, "config/face.json")))
        fcfg["Data"]["pose"]["convert_to_6d"] = True
        from trainer.config import Object
        w = quiet(nets.s2g_face, argparse.Namespace(gpu="cpu", infer=True), Object(fcfg))
        assert w.generator.identity is False and w.each_dim[0] == 6
        w.load_state_dict({"generator": T(synth.face_state_dict(seed=8, identity=False, jaw_dim=6))})     # strict: pins the key set
        B, N = 2, 32000
        wav = synth.wav16(35, B, N)
        w.generator.eval()
        with torch.no_grad():
            out = w.infer_on_audio(torch.from_numpy(wav)[:, None, :], id=torch.tensor([1, 2]))
            gen = w.generate(torch.from_numpy(wav)[:, None, :], N * 30 // 16000)
        assert np.array_equal(out, gen.numpy())                                  # the id is not looked at
        print("face_6d: out", out.shape, "std", float(out.std()))
        save("face_6d", wav_seed=np.asarray([35, B, N]), out=out)

    # ---- 5b. the face generator at BASELINE length: two full 10 s clips (160 000 samples -> 499 conv frames -> 300 output frames)
    if want("face_10s"):
        fcfg = json.load(open(os.path.join(REF, "config/face.json")))
        from trainer.config import Object
        w = quiet(nets.s2g_face, argparse.Namespace(gpu="cpu", infer=True), Object(fcfg))
        w.load_state_dict({"generator": T(synth.face_state_dict(seed=7))})
        B, N = 2, 160000
        wav = synth.wav16(33, B, N)                          # regenerated from the seed by the tests (not stored: 1.3 MB)
        w.generator.eval()
        with torch.no_grad():
            ids = torch.zeros(B, 4); ids[0, 1] = 1.0
            out = w.generator(torch.from_numpy(wav)[:, None, :], None, ids, time_steps=300)[0]
        print("face_10s: out", tuple(out.shape), "std", float(out.std()))
        save("face_10s", wav_seed=np.asarray([33, B, N]), ids=ids.numpy(), out=out.numpy())

    # ---- 5c. FGD feature extractor: vqvae_1d.AE through the reference wrapper nets.s2g_body_ae (body_ae.py:145-152) -------
    if want("ae_full"):
        cfg = json.load(open(os.path.join(REF, "config/body_pixel.json")))
        from trainer.config import Object
        w = quiet(nets.s2g_body_ae, argparse.Namespace(gpu="cpu", infer=True), Object(cfg))
        sd = synth.ae_state_dict(seed=7)
        w.load_state_dict({"g": T(sd)})                      # strict: pins the key names / shapes of ae_state_dict
        B, Tn = 2, 40
        p129 = synth.gt_poses(51, B, Tn)
        from data_utils.lower_body import c_index_3d
        wide = np.zeros((B, Tn, 165), np.float32)
        wide[:, :, c_index_3d] = p129
        w.g.eval()
        with torch.no_grad():
            feat, x129 = w.extract(torch.from_numpy(wide))                   # 165-wide rows -> c_index gather -> encode
            feat2, _ = w.extract(torch.from_numpy(p129))                     # already 129 wide
            z, recon = w.g(gt_poses=torch.from_numpy(p129))                  # AE.forward, eval branch
        assert torch.equal(feat, feat2) and torch.equal(x129, torch.from_numpy(p129))
        print("ae_full: feat", tuple(feat.shape), "std", float(feat.std()), "recon std", float(recon.std()))
        save("ae_full", poses129=p129, feat=feat.numpy(), z=z.numpy(), recon=recon.numpy())

    # ---- 5d. evaluation metrics: the reference's own evaluation/FGD.py, evaluation/metrics.py and test_body.body_loss ------
    if want("eval_metrics"):
        import ast
        from evaluation.FGD import EmbeddingSpaceEvaluator
        from evaluation import metrics as RM
        rng = np.random.default_rng(61)

        class StubAE:                                         # extract() = identity on ready-made feature rows
            def extract(self, x):
                return x, x
        ev = EmbeddingSpaceEvaluator(StubAE(), None, "cpu")
        real, gen = [], []
        for clip in range(5):
            H = 20 + 3 * clip
            r = (rng.standard_normal((1, H, 64)) * 0.5 + 0.1).astype(np.float32)
            g_ = (r + 0.2 * rng.standard_normal((2, H, 64))).astype(np.float32)
            ev.push_samples(torch.from_numpy(g_), torch.from_numpy(r))
            real.append(r); gen.append(g_)
        fgd, feat_dist = ev.get_scores()
        # body_loss is defined inside scripts/test_body.py, which cannot be imported (CUDA, dataset, smplx): take the
        # function's own source out of the file and run it with the reference's LVD
        src = open(os.path.join(REF, "scripts/test_body.py")).read()
        fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "body_loss"][0]
        ns = {"LVD": RM.LVD, "torch": torch}
        exec(compile(ast.Module([fn], []), "test_body.body_loss", "exec"), ns)
        T_, J_ = 50, 55
        gt_j = rng.standard_normal((T_, J_, 3)).astype(np.float32).cumsum(0) * 0.01
        pr_j = (gt_j[None] + 0.05 * rng.standard_normal((3, T_, J_, 3))).astype(np.float32)
        bl = ns["body_loss"](torch.from_numpy(gt_j), torch.from_numpy(pr_j))
        kps = rng.standard_normal((4, 30, 129)).astype(np.float32)
        div = RM.diversity(kps)
        lvd_single = RM.LVD(torch.from_numpy(gt_j), torch.from_numpy(pr_j[0]))
        # beat metrics: 2 clips of joints + audio onset times
        for clip in range(2):
            jt = (rng.standard_normal((60, 55, 3)) * 0.3).astype(np.float32)
            ev.push_joints(torch.from_numpy(jt[None].copy()), torch.from_numpy(jt.copy() * 0.9))
            ev.push_aud(torch.from_numpy(np.sort(rng.uniform(0.1, 1.9, 7)).reshape(-1, 1)))
        joints_real = [j.numpy().copy() for j in ev.real_joints_list]
        joints_gen = [j.numpy().copy() for j in ev.generated_joints_list]
        beats = [a.numpy().copy() for a in ev.audio_beat_list]
        maac = ev.get_MAAC().numpy()
        bc = ev.get_BCscore()
        print("eval_metrics: fgd", float(fgd), "feat_dist", float(feat_dist), "body_loss", {k: float(v) for k, v in bl.items()},
              "diversity", float(div), "bc", float(bc))
        save("eval_metrics", real=np.concatenate([r.reshape(-1) for r in real]), gen=np.concatenate([g_.reshape(-1) for g_ in gen]),
             clip_rows=np.asarray([r.shape[1] for r in real]), fgd=np.float64(fgd), feat_dist=np.float64(feat_dist),
             gt_joints=gt_j, pr_joints=pr_j, lvd=np.float64(bl["LVD"]), error=np.float64(bl["error"]),
             diverse=np.float64(bl["diverse"]), lvd_single=np.float64(lvd_single), kps=kps, diversity=np.float64(div),
             joints_real=np.stack(joints_real), joints_gen=np.stack(joints_gen), beats=np.stack(beats), maac=maac,
             bc=np.float64(bc))

    # ---- 5e. symmetric LVD (metrics.py:36-65): the reference's own function; it calls .cuda() on two masks, which is patched to
    # a no-op here (no GPU in the build container) — everything else is the reference's arithmetic, its `~mask.long()` included
    if want("lvd_symmetric"):
        from evaluation import metrics as RM
        rng = np.random.default_rng(67)
        T_, J_ = 40, 22
        gt_j = rng.standard_normal((T_, J_, 3)).astype(np.float32).cumsum(0) * 0.02
        pr_j = (gt_j[None] + 0.05 * rng.standard_normal((3, T_ + 4, J_, 3))[:, :T_]).astype(np.float32)
        pr_long = np.concatenate([pr_j, pr_j[:, -5:]], 1)                   # longer than gt: the reference cuts to gt's length
        keep = torch.Tensor.cuda
        torch.Tensor.cuda = lambda self, *a, **k: self
        try:
            sym = RM.LVD(torch.from_numpy(gt_j), torch.from_numpy(pr_j), symmetrical=True, weight=False)
            sym_long = RM.LVD(torch.from_numpy(gt_j), torch.from_numpy(pr_long), symmetrical=True, weight=False)
        finally:
            torch.Tensor.cuda = keep
        plain = RM.LVD(torch.from_numpy(gt_j), torch.from_numpy(pr_j), symmetrical=False, weight=False)
        print("lvd_symmetric:", float(sym), float(sym_long), "plain", float(plain))
        save("lvd_symmetric", gt_joints=gt_j, pr_joints=pr_j, pr_long=pr_long, lvd_sym=np.float64(sym), lvd_sym_long=np.float64(sym_long),
             lvd_plain=np.float64(plain))

    # ---- 6. output assembly: demo.py:207-229 (length alignment + concat) and lower_body.part2full ------------------
    if want("assemble_full"):
        from data_utils.lower_body import part2full
        rng = np.random.default_rng(41)
        body = rng.standard_normal((2, 12, 129)).astype(np.float32)
        outs = {}
        for tag, Tf in (("longer_face", 15), ("shorter_face", 9)):
            face = rng.standard_normal((2, Tf, 103)).astype(np.float32)
            res = {False: [], True: []}
            for b in range(2):
                pred_face = torch.from_numpy(face[b])
                pred_jaw, pred_exp = pred_face[:, :3], pred_face[:, 3:]          # demo.py:188-190
                pred = torch.from_numpy(body[b])
                if pred.shape[0] < pred_face.shape[0]:                            # demo.py:207-211
                    repeat_frame = pred[-1].unsqueeze(dim=0).repeat(pred_face.shape[0] - pred.shape[0], 1)
                    pred = torch.cat([pred, repeat_frame], dim=0)
                else:
                    pred = pred[:pred_face.shape[0], :]
                pred = torch.cat([pred_jaw, pred, pred_exp], dim=-1)              # demo.py:225
                for stand in (False, True):
                    res[stand].append(part2full(pred, stand).numpy())             # demo.py:228
            outs["face_" + tag] = face
            outs["full_" + tag] = np.stack(res[False])
            outs["full_stand_" + tag] = np.stack(res[True])
        print("assemble_full:", {k: v.shape for k, v in outs.items()})
        save("assemble_full", body=body, **outs)

    # ---- 7. REAL AUDIO: the reference's own demo recordings (demo_audio/{style,1st-page,french}.wav, the inputs of scripts/demo.py)
    # through the reference's modules downstream of the third-party front-end (every other golden uses iid
    # N(0, 20^2) feature rows / white noise).  Body: the float64 twin's MFCC rows (talkshow_amd/frontend.py::mfcc_float64 — the
    # front-end is third-party code, its pin stays separate) cast to float32 are the INPUT; the reference's AudioEncoder, the greedy
    # harness around GatedPixelCNN.forward and both VQVAE.decode calls (smplx_body_pixel.py:272-285) produce the expected values.
    # Each recording runs under all four speaker ids (B = 4: 1 944 greedy decisions over the three).  Stored: rows, audio-encoder output,
    # codes and top-2 margins of the four, poses of one (the id listed in RECS).  H = 75 / 96 / 72; about a minute on 8 cores.
    AUDIO = os.path.join(HERE, "audio")
    RECS = (("style.wav", 2), ("1st-page.wav", 0), ("french.wav", 3))                # (recording, speaker id)
    if args.only == "real_audio_body" or (not args.only and os.environ.get("TS_GOLDEN_BIG")):
        tmp = tempfile.mkdtemp(prefix="ts_golden_")
        vq_path = os.path.join(tmp, "vq.pth")
        torch.save({"generator": {"g_body": T(synth.vqvae_state_dict(seed=7, in_dim=39)),
                                  "g_hand": T(synth.vqvae_state_dict(seed=7, in_dim=90, salt=1))}}, vq_path)
        cfg = json.load(open(os.path.join(REF, "config/body_pixel.json")))
        cfg["Model"]["vq_path"] = vq_path
        from trainer.config import Object
        w = quiet(nets.s2g_body_pixel, argparse.Namespace(gpu="cpu", infer=True), Object(cfg))
        w.load_state_dict({"generator": T(synth.pixelcnn_state_dict(seed=7)), "audioencoder": T(synth.audioencoder_state_dict(seed=7))})
        w.generator.eval(); w.g_body.eval(); w.g_hand.eval(); w.audioencoder.eval()
        out = {}
        for name, spk in RECS:
            tag = name[:-4].replace("-", "_")
            wave = fe._load_mono_resampled(os.path.join(REF, "demo_audio", name), 22000)
            rows = fe.mfcc_float64(wave, 22000, hop_length=734).T.astype(np.float32)      # (T, 64): what get_mfcc_ta hands the wrapper
            ids = np.arange(4, dtype=np.int64)                                            # the recording under every speaker id: B = 4
            with torch.no_grad():
                feat = w.audioencoder(torch.from_numpy(rows[None]).transpose(1, 2), frame_num=0)    # smplx_body_pixel.py:274
                aud = feat.unsqueeze(-1).repeat(4, 1, 1, 2)                                   # `aud_feat[np.newaxis].repeat(B)` (:249)
                H = aud.shape[2]
                codes, step_logits = greedy_reference(w.generator, torch.from_numpy(ids), aud, H)
                body, _ = w.g_body.decode(b=4, w=H, latents=codes[..., 0])
                hand, _ = w.g_hand.decode(b=4, w=H, latents=codes[..., 1])
                poses = torch.cat([body, hand], dim=1).transpose(1, 2)
            m = margins(step_logits.numpy()).reshape(4, H, 2)
            print(f"real_audio_body {name}: rows {rows.shape} |max| {np.abs(rows).max():.0f}  H {H}  aud_feat |max| {float(feat.abs().max()):.2f} "
                  f"margin min/median {m.min():.2e} {np.median(m):.3f}  decisions under 1e-3: {int((m < 1e-3).sum())} of {m.size}  "
                  f"uniq codes {codes.unique().numel()}  pose std {float(poses.std()):.3f}")
            out.update({f"{tag}_rows": rows, f"{tag}_aud_feat": feat.permute(0, 2, 1).numpy()[0], f"{tag}_codes": codes.numpy().astype(np.int16),
                        f"{tag}_margin": m.astype(np.float32), f"{tag}_pose_id": np.asarray(spk), f"{tag}_poses": poses.numpy()[spk]})
        save("real_audio_body", **out)

    # ---- 7a. a SECOND set of weights on a recording (seed 11 instead of 7: other logits, other near-ties): french.wav under all four ids
    if args.only == "real_audio_body_w11" or (not args.only and os.environ.get("TS_GOLDEN_BIG")):
        tmp = tempfile.mkdtemp(prefix="ts_golden_")
        vq_path = os.path.join(tmp, "vq.pth")
        torch.save({"generator": {"g_body": T(synth.vqvae_state_dict(seed=11, in_dim=39)),
                                  "g_hand": T(synth.vqvae_state_dict(seed=11, in_dim=90, salt=1))}}, vq_path)
        cfg = json.load(open(os.path.join(REF, "config/body_pixel.json")))
        cfg["Model"]["vq_path"] = vq_path
        from trainer.config import Object
        w = quiet(nets.s2g_body_pixel, argparse.Namespace(gpu="cpu", infer=True), Object(cfg))
        w.load_state_dict({"generator": T(synth.pixelcnn_state_dict(seed=11)), "audioencoder": T(synth.audioencoder_state_dict(seed=11))})
        w.generator.eval(); w.g_body.eval(); w.g_hand.eval(); w.audioencoder.eval()
        wave = fe._load_mono_resampled(os.path.join(REF, "demo_audio", "french.wav"), 22000)
        rows = fe.mfcc_float64(wave, 22000, hop_length=734).T.astype(np.float32)
        ids = np.arange(4, dtype=np.int64)
        with torch.no_grad():
            feat = w.audioencoder(torch.from_numpy(rows[None]).transpose(1, 2), frame_num=0)
            aud = feat.unsqueeze(-1).repeat(4, 1, 1, 2)
            H = aud.shape[2]
            codes, step_logits = greedy_reference(w.generator, torch.from_numpy(ids), aud, H)
            body, _ = w.g_body.decode(b=4, w=H, latents=codes[..., 0])
            hand, _ = w.g_hand.decode(b=4, w=H, latents=codes[..., 1])
            poses = torch.cat([body, hand], dim=1).transpose(1, 2)
        m = margins(step_logits.numpy()).reshape(4, H, 2)
        print(f"real_audio_body_w11 french.wav: H {H} margin min/median {m.min():.2e} {np.median(m):.3f} under 1e-3: {int((m < 1e-3).sum())} of {m.size} uniq codes {codes.unique().numel()}")
        save("real_audio_body_w11", weight_seed=np.asarray(11), codes=codes.numpy().astype(np.int16), margin=m.astype(np.float32), poses_id1=poses.numpy()[1])

    # ---- 7b. the face on the recordings: the reference wrapper's `generate(wav[None, None], frame)` (smplx_face.py:221-238: zero id)
    # and `infer_on_audio`-style one-hot id through `generator(...)`, frame = N * 30 // 16000 (smplx_face.py:203).  1st-page.wav is
    # native 16 kHz mono: int16 / 32768, no third-party step at all.  style.wav / french.wav: the 16 kHz samples the host twin of
    # librosa's kaiser_best produces (tests/golden/audio/<name>.wav16.npy, written by fetch_reference_audio.py; their sha256 is stored
    # here) are the input — the resampler's own pin stays separate.  Hidden state: every 6th frame of the lerped wav2vec2 output.
    if args.only == "real_audio_face" or (not args.only and os.environ.get("TS_GOLDEN_BIG")):
        import hashlib
        fcfg = json.load(open(os.path.join(REF, "config/face.json")))
        from trainer.config import Object
        w = quiet(nets.s2g_face, argparse.Namespace(gpu="cpu", infer=True), Object(fcfg))
        w.load_state_dict({"generator": T(synth.face_state_dict(seed=7))})
        w.generator.eval()
        out = {}
        for name, spk in RECS:
            tag = name[:-4].replace("-", "_")
            wav = fe.get_wav16(os.path.join(REF, "demo_audio", name), host=True)[:, 0]            # (N,) float32 at 16 kHz
            side = os.path.join(AUDIO, name + ".wav16.npy")
            if os.path.exists(side):
                assert np.array_equal(np.load(side), wav), f"{side} is not what the host twin produces here"
            frame = wav.shape[0] * 30 // 16000
            x = torch.from_numpy(wav)[None, None, :]
            with torch.no_grad():
                gen = w.generate(x, frame)                                                      # zero id
                idv = torch.zeros(1, 4); idv[0, spk] = 1.0
                hot = w.generator(x, None, idv, time_steps=frame)[0]                            # one-hot id (infer_on_audio's branch)
                hs = w.generator.audio_encoder(torch.from_numpy(wav)[None], frame_num=frame).last_hidden_state
            print(f"real_audio_face {name}: N {wav.shape[0]} frame {frame} out std {float(gen.std()):.3f} hidden std {float(hs.std()):.3f} "
                  f"|wav| max {np.abs(wav).max():.3f} rms {np.sqrt((wav.astype(np.float64) ** 2).mean()):.4f}")
            out.update({f"{tag}_n": np.asarray([wav.shape[0], frame, spk]), f"{tag}_wav16_sha256": np.asarray(hashlib.sha256(wav.tobytes()).hexdigest()),
                        f"{tag}_out_zero_id": gen.numpy()[0], f"{tag}_out_one_hot": hot.numpy()[0], f"{tag}_hidden_6": hs.numpy()[0, ::6]})
        save("real_audio_face", **out)

    meta_path = os.path.join(HERE, "golden_meta.json")
    old = json.load(open(meta_path)) if os.path.exists(meta_path) else {"cases": {}}
    old["cases"].update(meta["cases"])
    old.update({k: v for k, v in meta.items() if k != "cases"})
    json.dump(old, open(meta_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
