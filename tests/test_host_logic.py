"""CPU-side tests: the C-ABI library loads and exports every declared symbol, the wrappers keep the reference's
checkpoint / call-surface contract, and nothing silently falls back to a CPU implementation."""
import argparse
import json
import os
import re

import numpy as np
import pytest
import torch

from talkshow_amd import synth

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from talkshow_amd import _lib
    import glob
    headers = sorted(glob.glob(os.path.join(REPO, "include", "*.h")))           # the drop-in ABI and the debug / tuning header
    assert [os.path.basename(h) for h in headers] == ["talkshow_hip.h", "talkshow_hip_debug.h"]
    declared = set()
    for h in headers:
        declared |= set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", open(h).read()))
    declared -= {"ts_tensor"}
    public = set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", open(headers[0]).read()))
    assert not [n for n in public if n.startswith("ts_debug_")], "debug entry points belong in talkshow_hip_debug.h"
    lib = _lib.load()                       # raises if the .so is missing: there is no fallback
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype"
    assert lib.ts_version().decode().startswith("talkshow_hip")
    # no compute without a GPU: asking for a context must fail loudly, not fall back
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="HIP device"):
            _lib.context(0)


def test_debug_entry_points_have_no_product_callers():
    """`include/talkshow_hip_debug.h` is for tools/ and tests/: nothing under nets/, evaluation/ or talkshow_amd/ (its ctypes prototype
    table and the stream helper that forwards an explicit `cus=` request apart) calls an entry point declared there, and the library
    reads its TS_* test levers in ONE place (api.cpp, at the first ts_ctx_create), never on a launch path."""
    dbg = set(re.findall(r"\b(ts_[a-z0-9_]+)\s*\(", open(os.path.join(REPO, "include", "talkshow_hip_debug.h")).read()))
    assert {"ts_debug_skinny_trace", "ts_op_conv1d_timed", "ts_stream_create_cus"} <= dbg
    for root in ("nets", "evaluation", "talkshow_amd"):
        for dp, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if not f.endswith(".py") or f == "_lib.py":
                    continue
                src = open(os.path.join(dp, f)).read()
                used = [n for n in dbg if n in src]
                assert not used, f"{dp}/{f} calls debug entry points {used}"
    csrc = os.path.join(REPO, "talkshow_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cpp", ".h")) and f != "api.cpp":
            code = "\n".join(l.split("//")[0] for l in open(os.path.join(csrc, f)).read().splitlines())
            assert "getenv" not in code, f"{f} reads the environment outside ts::knobs()"


def test_no_oracle_import_in_product_code():
    """The oracle is test infrastructure: nothing under talkshow_amd/ or nets/ may import it."""
    for root in ("talkshow_amd", "nets"):
        for dp, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dp, f)).read()
                    assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f"{dp}/{f} imports the oracle"


def _config(tmp_path, which):
    from talkshow_amd.config import Object
    cfg = json.load(open(os.path.join(REPO, "config", which + ".json")))
    if "vq_path" in cfg["Model"]:
        p = str(tmp_path / "vq.pth")
        small = dict(num_embeddings=2048, num_hiddens=1024)
        torch.save({"generator": {"g_body": synth.to_torch(synth.vqvae_state_dict(seed=1, in_dim=39, **small)),
                                  "g_hand": synth.to_torch(synth.vqvae_state_dict(seed=1, in_dim=90, salt=1, **small))}}, p)
        cfg["Model"]["vq_path"] = p
    return Object(cfg)


def test_init_model_surface(tmp_path):
    import nets
    from nets.init_model import init_model
    for name in ("s2g_face", "s2g_body_vq", "s2g_body_pixel", "s2g_body_ae", "LS3DCG", "TrainWrapperBaseClass",
                 "normalize", "denormalize"):
        assert hasattr(nets, name)
    with pytest.raises(ValueError):
        init_model("nope", None, None)
    args = argparse.Namespace(gpu="cpu", infer=True)
    w = init_model("s2g_body_pixel", args, _config(tmp_path, "body_pixel"))
    assert w.each_dim == [0, 39, 90, 100] and len(w.c_index) == 129 and w.num_classes == 4
    for attr in ("generator", "g_body", "g_hand", "audioencoder", "device"):
        assert hasattr(w, attr)
    # VQ checkpoint was loaded by the constructor (smplx_body_pixel.py:59-62)
    ref = synth.vqvae_state_dict(seed=1, in_dim=39)
    np.testing.assert_array_equal(w.g_body.state_dict()["decoder.project.weight"].numpy(), ref["decoder.project.weight"])
    # checkpoint round trip incl. DataParallel 'module.' prefixes and optimiser entries (smplx_body_pixel.py:115-142)
    gen = synth.to_torch(synth.pixelcnn_state_dict(seed=2))
    aud = synth.to_torch(synth.audioencoder_state_dict(seed=2))
    w.load_state_dict({"generator": {"module." + k: v for k, v in gen.items()}, "audioencoder": aud,
                       "generator_optim": {"state": {}}, "audioencoder_optim": None, "discriminator": None})
    sd = w.state_dict()
    assert list(sd["generator"].keys()) == list(gen.keys())
    np.testing.assert_array_equal(sd["audioencoder"]["project.conv.weight"].numpy(), aud["project.conv.weight"].numpy())
    assert float(sd["generator"]["layers.0.horiz_stack.weight"][..., -1].abs().max()) == 0.0     # mask 'A' applied
    np.testing.assert_array_equal(sd["generator"]["layers.1.horiz_stack.weight"].numpy(),
                                  gen["layers.1.horiz_stack.weight"].numpy())
    with pytest.raises(RuntimeError, match="size mismatch|missing"):
        w.generator.load_state_dict({k: v[..., :1] if v.ndim == 4 else v for k, v in gen.items()})
    w.generator.eval(); w.g_body.eval(); w.g_hand.eval(); w.audioencoder.eval()
    # no CPU path: inference without a HIP device fails loudly
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="HIP device|no CPU path"):
            w.infer_on_audio(synth.mfcc_features(0, 1, 60)[0], id=torch.tensor([0]), fps=30, B=1)
    with pytest.raises(AssertionError):
        argsT = argparse.Namespace(gpu="cpu", infer=False)
        init_model("s2g_body_pixel", argsT, _config(tmp_path, "body_pixel")).infer_on_audio(np.zeros((60, 64), np.float32))
    with pytest.raises(FileNotFoundError):          # wav paths go through the host front-end now
        w.infer_on_audio("no_such_file.wav", id=torch.tensor([0]))


def test_body_vq_wrapper_surface(tmp_path):
    from nets.init_model import init_model
    args = argparse.Namespace(gpu="cpu", infer=True)
    w = init_model("s2g_body_vq", args, _config(tmp_path, "body_vq"))
    b, h = synth.to_torch(synth.vqvae_state_dict(seed=4, in_dim=39)), synth.to_torch(synth.vqvae_state_dict(seed=4, in_dim=90, salt=1))
    w.load_state_dict({"g_body": b, "g_hand": h})
    sd = w.state_dict()
    assert set(sd) >= {"g_body", "g_hand"} and list(sd["g_hand"].keys()) == list(h.keys())
    with pytest.raises(ValueError):
        w.infer_on_audio(torch.zeros(1, 64, 60))


def test_out_of_scope_names_say_so():
    import nets
    with pytest.raises(NotImplementedError):
        nets.LS3DCG(None, None)


def test_body_ae_wrapper_surface(tmp_path):
    """nets.s2g_body_ae (FGD feature extractor): constructor, checkpoint keys, buffers vs parameters, CPU device refusal."""
    import argparse
    import json
    from nets.init_model import init_model
    from talkshow_amd import synth
    from talkshow_amd.config import Object
    cfg = Object(json.load(open(os.path.join(REPO, "config", "body_pixel.json"))))
    w = init_model("s2g_body_ae", argparse.Namespace(gpu="cpu", infer=True), cfg)
    sd = synth.to_torch(synth.ae_state_dict(seed=2))
    w.load_state_dict({"g": sd})
    out = w.state_dict()
    assert set(out) == {"g", "g_optim", "discriminator", "discriminator_optim"} and list(out["g"]) == list(sd)
    assert w.each_dim == [0, 39, 90, 100] and w.full_dim == 129      # expression=true in the shipped config
    n_params = sum(p.numel() for p in w.parameters())
    n_buffers = sum(v.numel() for k, v in sd.items() if k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_params == sum(v.numel() for v in sd.values()) - n_buffers
    with pytest.raises(RuntimeError, match="HIP device"):
        w.extract(torch.zeros(1, 8, 129))


def test_pose_index_matches_reference_layout():
    from talkshow_amd.pose_index import c_index_3d
    g = np.load(os.path.join(REPO, "tests", "golden", "body_vq_e2e_full.npz"))
    np.testing.assert_array_equal(c_index_3d, g["c_index"])      # c_index_3d as the reference computes it


def _tiled_index(m, k, W):
    """kernels.h, SkinnyParams::w_tiled: float index of element (m, k) of a [rows][W] array stored as 16 x 16 fragments."""
    return (((m >> 4) * (W >> 4) + (k >> 4)) << 8) + (((m & 15) + 16 * ((k & 15) >> 2)) << 2) + (k & 3)


@pytest.mark.parametrize("N,K,ldw", [(16, 16, 16), (40, 64, 80), (512, 256, 512)])
def test_tiled_weight_layout_linear(N, K, ldw):
    """Host-side tiling of a chain weight matrix (no GPU): every element lands where the documented fragment formula says,
    rows past N read as zero."""
    import ctypes as C
    from talkshow_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(N + K)
    W = rng.standard_normal((N, ldw)).astype(np.float32)
    nt = (N + 15) // 16
    out = np.full(nt * (K // 16) * 256, np.nan, np.float32)
    _lib.check(lib.ts_debug_tile_weights(W.ctypes.data_as(C.c_void_p), N, K, ldw, 0, 0, out.ctypes.data_as(C.c_void_p)))
    n, k = np.meshgrid(np.arange(nt * 16), np.arange(K), indexing="ij")
    want = np.where(n < N, W[np.minimum(n, N - 1), k], 0.0).astype(np.float32)
    assert np.array_equal(out[_tiled_index(n, k, K)], want)
    assert not np.isnan(out).any()                      # the formula is a bijection onto the buffer


def test_tiled_weight_layout_gate():
    """Gate tiles: tile t holds 8 'tanh' channels followed by their 8 'sigmoid' partners (columns c and c + gateD of a
    2*gateD group), the pairing the epilogue's lane-xor-8 shuffle relies on."""
    import ctypes as C
    from talkshow_amd import _lib
    lib = _lib.load()
    gateD, groups, K = 24, 2, 32
    N = 2 * gateD * groups
    W = np.random.default_rng(3).standard_normal((N, K)).astype(np.float32)
    out = np.empty((N // 16) * (K // 16) * 256, np.float32)
    _lib.check(lib.ts_debug_tile_weights(W.ctypes.data_as(C.c_void_p), N, K, K, 1, gateD, out.ctypes.data_as(C.c_void_p)))
    tiles_per_group = gateD // 8
    for t in range(N // 16):
        group, ch0 = divmod(t, tiles_per_group)
        for li in range(16):
            col = group * 2 * gateD + (li >> 3) * gateD + ch0 * 8 + (li & 7)
            got = out[_tiled_index(t * 16 + li, np.arange(K), K)]
            assert np.array_equal(got, W[col]), (t, li)
    with pytest.raises(RuntimeError, match="bad argument"):
        _lib.check(lib.ts_debug_tile_weights(W.ctypes.data_as(C.c_void_p), N, 24, K, 0, 0, out.ctypes.data_as(C.c_void_p)))


def test_conv_band_plan_covers_every_row_once():
    """conv_gemm_f32 launches layers of more than one round of 512 resident workgroups in two bands: 128 x 128 tiles for whole
    rounds, 64 x 128 tiles for the rows that are left (host logic, no GPU).  The bands must tile the rows exactly once, the
    first band must be a whole number of rounds short of at most one row block, and layers that fit one round or come to
    whole rounds keep the plain grid."""
    import ctypes as C
    from talkshow_amd import _lib
    lib = _lib.load()
    out = (C.c_int * 4)()
    banded = 0
    for M in (300, 4096, 19200, 19227, 38400, 76800, 65536, 100000):
        for N in (64, 256, 500, 512, 1024, 2048):
            for groups in (1, 2, 4):
                r = lib.ts_debug_conv_bands(M, N, groups, out)
                assert r in (0, 1)
                MT, NT = -(-M // 128), -(-N // 128) * groups
                if r == 0:
                    continue
                banded += 1
                mt_big, mt_small, first_small, total = list(out)
                assert MT * NT > 512 and (MT * NT) % 512 != 0
                assert 0 < mt_big < MT and mt_small > 0
                assert first_small == mt_big * NT and total == first_small + mt_small * NT
                rounds = (MT * NT) // 512
                assert rounds * 512 - NT < first_small <= rounds * 512            # whole rounds, short of less than one row block
                left = M - mt_big * 128
                assert 0 < left <= mt_small * 64 < left + 64                       # the small band ends within its last row block
    assert banded > 20
    # the bench's paired 1024-channel layers: 19 200 rows x 1 024 columns x 2 problems = 2 400 tiles = 4.69 rounds
    assert lib.ts_debug_conv_bands(19200, 1024, 2, out) == 1 and list(out) == [128, 44, 2048, 2048 + 44 * 16]
    assert lib.ts_debug_conv_bands(4096, 4096, 1, out) == 0                        # 1 024 tiles: two whole rounds
    assert lib.ts_debug_conv_bands(2400, 1024, 2, out) == 0                        # 304 tiles: one round
    assert lib.ts_debug_conv_bands(0, 64, 1, out) == -1


def test_split_gemm_tile_order_covers_every_tile_once():
    """`conv_gemm_split` (the opt-in bf16x3 face plan) and `conv_gemm_f32`'s ring engine run on a 1-D grid whose workgroups pick their tile through `split_tile_of`
    (`csrc/kernels.h`, the same function on the device and here): ids go round-robin over the 8 XCDs, XCD x takes the x-th eighth of a
    tile list ordered by column groups.  Host logic, no GPU: every tile exactly once, padding workgroups get none, and the 64 workgroups
    an XCD holds at a time (64 consecutive list positions) touch few distinct row and column tiles — that is the point of the order."""
    import ctypes as C
    from talkshow_amd import _lib
    lib = _lib.load()
    out = (C.c_int * 2)()
    for MT, NT, gw in ((150, 6, 8), (150, 18, 8), (150, 24, 8), (8000, 4, 8), (2, 1, 8), (37, 11, 4), (5, 5, 8), (150, 24, 6),
                       (8000, 4, 4), (200, 6, 6), (151, 4, 4), (2000, 4, 4)):   # conv_gemm_f32's ring engine: width min(NT, 8) (conv_gemm_ring.hip)
        total, per = MT * NT, -(-MT * NT // 8)
        seen, by_xcd = set(), {x: [] for x in range(8)}
        for bid in range(per * 8):
            r = lib.ts_debug_split_tile(bid, MT, NT, gw, out)
            assert r in (0, 1)
            if r:
                t = (out[0], out[1])
                assert 0 <= t[0] < MT and 0 <= t[1] < NT and t not in seen
                seen.add(t)
                by_xcd[bid & 7].append(t)
        assert len(seen) == total                                   # a bijection onto the tile grid
        assert lib.ts_debug_split_tile(per * 8 - 1, MT, NT, gw, out) == (1 if total % 8 == 0 else 0)
        if total >= 8 * 64:
            for x, tiles in by_xcd.items():                          # what one XCD holds at a time shares operands:
                for a in range(0, len(tiles) - 63, 64):               # 64 tiles out of a compact block (a plain grid order: ~150 x 4)
                    rows, cols = {t[0] for t in tiles[a:a + 64]}, {t[1] for t in tiles[a:a + 64]}
                    assert len(rows) * len(cols) <= 256 and len(cols) <= 2 * gw, (MT, NT, gw, x, a, len(rows), len(cols))
    assert lib.ts_debug_split_tile(0, 0, 4, 8, out) == -1


def test_bench_pass_plan():
    """bench.py groups the queued 32-clip steps into chain passes: full passes of G batches, then the remainder — the
    driver's `--steps 20` at G = 8 is 8 + 8 + 4, every step is run exactly once."""
    import bench
    eng = bench.Engine.__new__(bench.Engine)
    for G, steps, want in ((8, 20, [8, 8, 4]), (8, 48, [8] * 6), (16, 20, [16, 4]), (8, 5, [5]), (1, 3, [1, 1, 1])):
        eng.G = G
        assert eng.plan(steps) == want and sum(eng.plan(steps)) == steps
    assert bench.FRAMES_PER_CLIP == 300


def test_index_range_check_has_no_stale_hits():
    """the check used to cache on (data_ptr, numel, _version) — tensors built inside a call reuse addresses with
    _version 0, so an out-of-range label could hit a stale entry.  Host-origin indices are now checked on the host every
    time; the device-tensor cache holds a weak reference to the tensor object."""
    from talkshow_amd.modules import _check_index_range
    for _ in range(300):
        _check_index_range(torch.tensor([3]).repeat(32), 4, "label")          # same allocation pattern, in range
        with pytest.raises(IndexError):
            _check_index_range(torch.tensor([4]).repeat(32), 4, "label")      # ... and out of range: never accepted
    _check_index_range([0, 1, 2, 3], 4, "label")
    _check_index_range(np.array([], dtype=np.int64), 4, "label")
    _check_index_range(2, 4, "label")
    for bad in ([0, -1], np.array([5]), 7, torch.tensor([[1, 9]])):
        with pytest.raises(IndexError):
            _check_index_range(bad, 4, "label")


def test_pixelcnn_audio_map_columns_must_be_copies():
    """the reference convolves the whole (B, aud_dim, H, W) audio map; the C entries take one audio row per code row (its only
    caller repeats a row over the columns, `smplx_body_pixel.py:274`).  A map whose columns differ is refused, not cut to column 0."""
    import torch
    from talkshow_amd.modules import GatedPixelCNN
    rows = torch.randn(2, 8, 5)
    same = rows[..., None].repeat(1, 1, 1, 4)
    assert torch.equal(GatedPixelCNN._audio_rows(same), rows.transpose(1, 2))
    assert GatedPixelCNN._audio_rows(None) is None
    other = same.clone()
    other[1, 3, 2, 3] += 1.0
    with pytest.raises(NotImplementedError, match="columns differ"):
        GatedPixelCNN._audio_rows(other)


def test_conv_ring_tile_choice_by_tile_count():
    """The ring engine's tile plan for a layer (host-only): rounds of 512 resident workgroups, a last round at most half full costs half a
    round; 128-row tiles, 96-row tiles (5 % more per flop) or bands (128-row tiles for the whole rounds + 64-row tiles, 7 % more per flop, for
    the rest; reported as 64).  The wav2vec2 blocks at batch 64 x 300 frames: out-proj / FFN2 (N = 768: 900 tiles of 128 rows = 1.76 rounds
    -> 2; 1 200 of 96 rows -> 2.5 x 0.75) take 96-row tiles; FFN1 (7.03 rounds -> 6.98 + half a short one) and the paired body + hand layers of
    a 256-clip pass (2 400 tiles = 4.69 rounds -> 4 + 1.5 short ones) take bands; QKV (5.27 -> 5.5), exact multiples of a round and the long
    feature convolutions (bands within 2 %) keep 128."""
    from talkshow_amd import _lib
    lib = _lib.load()
    assert lib.ts_debug_conv_ring_pick(19200, 768, 1) == 96
    assert lib.ts_debug_conv_ring_pick(19200, 3072, 1) == 64
    assert lib.ts_debug_conv_ring_pick(255936, 512, 1) == 128 and lib.ts_debug_conv_ring_pick(511936, 512, 1) == 128
    assert lib.ts_debug_conv_ring_pick(1023936, 512, 1) == 128 and lib.ts_debug_conv_ring_pick(63936, 512, 1) == 128
    assert lib.ts_debug_conv_ring_pick(19200, 2304, 1) == 128
    assert lib.ts_debug_conv_ring_pick(16384, 1024, 1) == 128          # 1 024 tiles: two whole rounds
    assert lib.ts_debug_conv_ring_pick(8192, 1024, 1) == 128
    for M, N, G in ((19200, 1024, 2), (38400, 512, 2), (76800, 256, 2), (19200, 512, 4)):   # the paired layers of the VQ stacks at 256 clips
        assert lib.ts_debug_conv_ring_pick(M, N, G) == 64
    assert lib.ts_debug_conv_ring_pick(0, 64, 1) == -1


def test_stream_k_band_plan_and_runs():
    """Host side of the ring engine's stream-K plan (csrc/conv_gemm_ring.hip), no GPU needed: which layers get a band,
    and that the band's runs (a) cover every (tile, stage) iteration exactly once, (b) are equal to within one stage inside an XCD, (c) never
    split a tile ACROSS XCDs (the pieces of a tile meet in one L2: the XCDs' L2s are not coherent with each other), (d) give every band
    workgroup at most one partial per slot, and that the owner search the kernel uses to find a tile's pieces agrees with the cut."""
    import ctypes as C
    from talkshow_amd import _lib
    lib = _lib.load()
    o6, o4 = (C.c_int * 6)(), (C.c_int * 4)()
    # the wav2vec2 block GEMMs of a face batch of 64 (M = 19 200): out-proj / FFN2 (900 tiles), QKV (2 700), FFN1 (3 600 = 14 x 256 + 16)
    assert lib.ts_debug_conv_sk_plan(19200, 768, 768, 1, o6) == 1 and list(o6)[:5] == [128, 22, 768, 256, 24] and o6[5] == 3   # K = 768: the band's fixed cost eats the gain (measured)
    assert lib.ts_debug_conv_sk_plan(19200, 768, 3072, 1, o6) == 1 and o6[4] == 96 and o6[5] == 8
    assert lib.ts_debug_conv_sk_plan(19200, 2304, 768, 1, o6) == 1 and list(o6)[:4] == [142, 8, 2560, 256]
    assert lib.ts_debug_conv_sk_plan(19200, 3072, 768, 1, o6) == 1 and o6[0] * 24 + o6[1] * 24 == 3600 and o6[1] * 24 >= 256   # one more unit in the band
    assert lib.ts_debug_conv_sk_plan(32768, 1024, 512, 1, o6) == 0            # 2 048 tiles: whole units, nothing to balance
    assert lib.ts_debug_conv_sk_plan(2400, 1024, 512, 1, o6) == 0             # 152 tiles: under one unit
    assert lib.ts_debug_conv_sk_plan(19200, 768, 768, 3, o6) == 0             # 256 % groups
    for Ts, stages, W in ((132, 24, 256), (132, 96, 256), (144, 24, 256), (288, 24, 256), (33, 48, 128), (47, 7, 64), (232, 48, 256)):
        cover = [0] * (Ts * stages)
        per_tile_xcd = {}
        for q in range(W):
            assert lib.ts_debug_conv_sk_run(Ts, stages, W, q, o4) == 0
            it0, it1, c, r = list(o4)
            assert c == q % 8 and (it0 == it1 or r == q // 8)
            lo_t, hi_t = c * Ts // 8, (c + 1) * Ts // 8
            assert lo_t * stages <= it0 <= it1 <= hi_t * stages
            per_xcd = (hi_t - lo_t) * stages / (W // 8)
            assert abs((it1 - it0) - per_xcd) < 1.0 + 1e-9
            for x in range(it0, it1):
                cover[x] += 1
                per_tile_xcd.setdefault(x // stages, set()).add(c)
            pieces = {}
            for x in range(it0, it1):
                pieces.setdefault(x // stages, []).append(x)
            partial = [t for t, xs in pieces.items() if len(xs) < stages]
            assert len(partial) <= 2 and all(t in (min(pieces), max(pieces)) for t in partial)
        assert set(cover) == {1}, (Ts, stages, W)
        assert all(len(v) == 1 for v in per_tile_xcd.values())
