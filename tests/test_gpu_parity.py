"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors.

Run on a real MI355X with `pytest -m gpu`.  Tolerances: code indices / argmin / argmax / draws are bit-exact;
floats are compared with the absolute tolerance written next to each assert (1e-4 on poses is the bar BASELINE.json
states; intermediate activations are O(1) and agree to ~1e-5, summation-order noise of fp32).
"""
import argparse
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from conftest import assert_close_measured
from oracle import talkshow_oracle as O
from talkshow_amd import synth

pytestmark = pytest.mark.gpu

# PixelCNN logits (|logit| ~ 9-30) against the reference's: the bound is 2x the largest error MEASURED on the MI355X over every
# logits comparison of this file (profiles/r05_notes/measured_errors.jsonl: 9.75e-5 on the full-size network, <= 1.3e-5 on the
# small ones); the smallest top-2 margin that decides a code in the batch-32 golden is 3.4e-4, and the bound stays under it.
LOGIT_ATOL = 2e-4

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    from talkshow_amd import _lib
    assert torch.cuda.is_available(), "gpu tests need a HIP device"
    lib = _lib.load()
    ctx = _lib.context(0)
    return _lib, lib, ctx


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ----------------------------------------------------------------------------------------------- operators
@pytest.mark.parametrize("B,L,Cin,Cout,K,stride,tr,act", [
    (2, 75, 64, 64, 3, 1, 0, 1),      # k3 with LeakyReLU, the stack layer
    (3, 37, 39, 256, 3, 1, 0, 0),     # ragged: odd length, Cin not a multiple of 32 (first VQ encoder layer)
    (1, 1, 32, 32, 3, 1, 0, 2),       # single frame: every halo tap is padding
    (2, 300, 256, 39, 1, 1, 0, 0),    # pointwise projection to 39 pose dims (N tail)
    (2, 150, 64, 128, 4, 2, 0, 1),    # stride-2 down
    (2, 33, 64, 128, 4, 2, 0, 0),     # stride-2 down, odd length
    (2, 75, 128, 64, 4, 2, 1, 1),     # transposed up
    (32, 75, 1024, 1024, 3, 1, 0, 1), # the dominant layer at BASELINE batch size
])
def test_op_conv1d(hip, B, L, Cin, Cout, K, stride, tr, act):
    _lib, lib, ctx = hip
    rng = np.random.default_rng(B * 1000 + L + Cin)
    x = rng.standard_normal((B, L, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K) if tr else (Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    pad = 1 if K in (3, 4) else 0
    xc = np.ascontiguousarray(x.transpose(0, 2, 1))
    ref = O.conv_transpose1d(xc, w, b, 2, 1) if tr else O.conv1d(xc, w, b, stride, pad)
    if act == 1:
        ref = O.leaky_relu(ref)
    elif act == 2:
        ref = O.relu(ref)
    ref = ref.transpose(0, 2, 1)
    xd = dev(x)
    out = torch.empty(ref.shape, dtype=torch.float32, device="cuda")
    _lib.check(lib.ts_op_conv1d(ctx, _lib.dptr(xd), B, L, Cin, _lib.fptr(w), _lib.fptr(b), Cout, K, stride, pad, tr, act,
                                _lib.dptr(out), None))
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5, rtol=1e-5)


def test_conv_tile_shapes_agree(hip):
    """Every tile shape of conv_gemm_f32 (64x64 ... 160x128) and both engines — global -> VGPR -> LDS staging (conv_gemm.hip) and the
    LDS-DMA ring with 4 / 8 waves per 128 x 128 tile and its 96 x 128 tile (conv_gemm_ring.hip: tiles 31 / 39 / 33, and 35 / 36 = 39 / 33 with the tiles
    dealt to the XCDs on a 1-D grid — what production launches; halo taps read a zero buffer, rows beyond M too) —
    walk K in the same order, so the outputs must be bit-identical; M = 225 and N = 200 are ragged against every tile height / width."""
    _lib, lib, ctx = hip
    rng = np.random.default_rng(77)
    B, L, Cin, Cout, K = 3, 75, 64, 200, 3
    x = rng.standard_normal((B, L, Cin)).astype(np.float32)
    npad = (Cout + 127) // 128 * 128
    w = np.zeros((npad, K * Cin), np.float32)
    w[:Cout] = rng.standard_normal((Cout, K * Cin)).astype(np.float32) / np.sqrt(K * Cin)
    b = np.zeros(npad, np.float32)
    b[:Cout] = rng.standard_normal(Cout).astype(np.float32)
    xp = np.pad(x, ((0, 0), (1, 1), (0, 0)))
    ref = sum(xp[:, k:k + L, :] @ w[:Cout, k * Cin:(k + 1) * Cin].T for k in range(K)) + b[:Cout]
    ref = np.where(ref >= 0, ref, 0.2 * ref).astype(np.float32)
    xd, wd, bd = dev(x), dev(w), dev(b)
    outs = {}
    for tile in (1, 2, 3, 4, 5, 6, 7, 31, 33, 39, 35, 36, 37, 0):
        out = torch.full((B, L, Cout), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(xd), B, L, Cin, _lib.dptr(wd), _lib.dptr(bd), Cout, K, tile, 1,
                                          _lib.dptr(out), None, None))
        torch.cuda.synchronize()
        outs[tile] = out.cpu().numpy()
    np.testing.assert_allclose(outs[2], ref, atol=2e-5, rtol=1e-5)
    for tile, o in outs.items():
        assert np.array_equal(o, outs[2]), f"tile {tile} differs from the 64x64 tile"


def test_conv_stream_k_band(hip):
    """The ring engine's stream-K plan (tile id 38; csrc/conv_gemm_ring.hip: whole tiles for the whole units of 256, the rows after them as one
    list of (tile, stage) iterations cut into equal runs, split tiles summed by the last arriver in k order): against a float64 restatement and
    the dealt whole-tile plan (tile 35) on a wav2vec2-block shape (one segment), a 3-tap convolution (the seek walks taps) and a shape whose
    band takes one more unit (runs longer than a tile); deterministic — the same bits run after run, with ANOTHER input launched in between
    (a partial read stale from the previous launch cannot pass as the right value); red zones around the output intact."""
    _lib, lib, ctx = hip
    assert lib.ts_debug_conv_sk_supported() == 1, "MI355X: workgroup ids of equal residue mod 8 must share an XCD (probed by ts_ctx_create)"
    rng = np.random.default_rng(79)
    for B, L, Cin, Cout, K in ((64, 300, 768, 768, 1), (64, 300, 512, 384, 3), (64, 300, 256, 3072, 1)):
        o6 = (C.c_int * 6)()
        assert lib.ts_debug_conv_sk_plan(B * L, Cout, K * Cin, 1, o6) == 1, "this shape must have a stream-K plan"
        npad = (Cout + 127) // 128 * 128
        w = np.zeros((npad, K * Cin), np.float32)
        w[:Cout] = rng.standard_normal((Cout, K * Cin)).astype(np.float32) / np.sqrt(K * Cin)
        b = np.zeros(npad, np.float32)
        b[:Cout] = rng.standard_normal(Cout).astype(np.float32)
        xs = [rng.standard_normal((B, L, Cin)).astype(np.float32) for _ in range(2)]
        wd, bd = dev(w), dev(b)
        guard = 4096
        results = []
        for rep in range(3):
            for k, x in enumerate(xs):
                xd = dev(x)
                buf = torch.full((B * L * Cout + 2 * guard,), float("nan"), dtype=torch.float32, device="cuda")
                out = buf[guard:guard + B * L * Cout].view(B, L, Cout)
                _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(xd), B, L, Cin, _lib.dptr(wd), _lib.dptr(bd), Cout, K, 38, 1, _lib.dptr(out), None, None))
                torch.cuda.synchronize()
                assert bool(torch.isnan(buf[:guard]).all()) and bool(torch.isnan(buf[guard + B * L * Cout:]).all()), "a store landed outside the output"
                if rep == 0:
                    ref35 = torch.empty((B, L, Cout), dtype=torch.float32, device="cuda")
                    _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(xd), B, L, Cin, _lib.dptr(wd), _lib.dptr(bd), Cout, K, 35, 1, _lib.dptr(ref35), None, None))
                    torch.cuda.synchronize()
                    xp = np.pad(x, ((0, 0), (K // 2, K // 2), (0, 0))).astype(np.float64)
                    rows = slice(B * L - 3000, B * L)                      # the band's rows are the last ones: restate those (and the first) in float64
                    pre = sum(xp[:, t:t + L, :].reshape(B * L, Cin)[rows] @ w[:Cout, t * Cin:(t + 1) * Cin].T.astype(np.float64) for t in range(K)) + b[:Cout]
                    want = np.where(pre >= 0, pre, 0.2 * pre)
                    got = out.reshape(B * L, Cout)[rows].cpu().numpy()
                    assert_close_measured(f"conv_stream_k.{Cin}x{Cout}x{K}.vs_float64", got, want, 2e-5)
                    d = float((out - ref35).abs().max())
                    print(f"\nstream-K vs dealt whole tiles ({B * L} x {Cout} x {K * Cin}): max |diff| {d:.2e}")
                    assert d <= 2e-5 and bool((out != ref35).any()), "the band must differ from the whole-tile plan in rounding only (and it must exist)"
                    results.append(out.clone())
                else:
                    assert torch.equal(out, results[k]), f"input {k}, repetition {rep}: stream-K bits changed between runs"


def test_strided_conv_tiles_agree(hip):
    """The wav2vec2 feature convolutions' shape (k = 3, stride 2, no padding, GELU; HF Wav2Vec2FeatureEncoder under
    nets/spg/wav2vec.py) through every engine / tile order that can carry it: bit-identical, and equal to a float64 restatement."""
    import math
    _lib, lib, ctx = hip
    rng = np.random.default_rng(79)
    B, L, Cin, Cout, K, stride = 3, 1201, 64, 200, 3, 2
    Lout = (L - K) // stride + 1
    x = rng.standard_normal((B, L, Cin)).astype(np.float32)
    npad = (Cout + 127) // 128 * 128
    w = np.zeros((npad, K * Cin), np.float32)
    w[:Cout] = rng.standard_normal((Cout, K * Cin)).astype(np.float32) / np.sqrt(K * Cin)
    b = np.zeros(npad, np.float32)
    b[:Cout] = rng.standard_normal(Cout).astype(np.float32)
    pre = sum(x[:, k:k + stride * (Lout - 1) + 1:stride, :].astype(np.float64) @ w[:Cout, k * Cin:(k + 1) * Cin].T.astype(np.float64)
              for k in range(K)) + b[:Cout]
    ref = 0.5 * pre * (1.0 + np.vectorize(math.erf)(pre / math.sqrt(2.0)))
    xd, wd, bd = dev(x), dev(w), dev(b)
    outs = {}
    for tile in (2, 1, 0, 39, 35, 33, 36):
        out = torch.full((B, Lout, Cout), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(lib.ts_op_conv1d_strided_timed(ctx, _lib.dptr(xd), B, L, Cin, _lib.dptr(wd), _lib.dptr(bd), Cout, K, stride, tile, 1,
                                                  _lib.dptr(out), None, None))
        torch.cuda.synchronize()
        outs[tile] = out.cpu().numpy()
    assert_close_measured("strided_conv_gelu", outs[2], ref, 2e-5)
    for tile, o in outs.items():
        assert np.array_equal(o, outs[2]), f"tile {tile} differs from the 64x64 tile"


@pytest.mark.parametrize("B,T,G,ntap,with_res", [(2, 75, 2, 16, True), (3, 131, 16, 128, True), (1, 5, 1, 128, False), (2, 300, 3, 7, True)])
def test_grouped_taps48_conv(hip, B, T, G, ntap, with_res):
    """conv_taps48.hip — the wav2vec2 positional convolution (HF Wav2Vec2PositionalConvEmbedding: Conv1d(768, 768, 128, padding 64, groups 16),
    last frame dropped, GELU; + the encoder's residual) unpadded on the 16 x 16 MFMA — against a float64 restatement: clips shorter than the
    kernel (every tap partly in the zero padding), row counts ragged against the 128-row tile, odd tap counts."""
    import math
    _lib, lib, ctx = hip
    rng = np.random.default_rng(80 + T)
    C = G * 48
    x = rng.standard_normal((B, T, C)).astype(np.float32)
    w = (rng.standard_normal((G, 48, ntap, 48)) / np.sqrt(ntap * 48)).astype(np.float32)     # [group][out][tap][in]
    bias = rng.standard_normal(C).astype(np.float32)
    res = rng.standard_normal((B, T, C)).astype(np.float32) if with_res else None
    d0 = -(ntap // 2)
    xp = np.zeros((B, T + 2 * ntap, C), np.float64)
    xp[:, ntap:ntap + T] = x
    pre = np.zeros((B, T, C), np.float64)
    for g in range(G):
        for k in range(ntap):
            pre[:, :, g * 48:(g + 1) * 48] += xp[:, ntap + d0 + k:ntap + d0 + k + T, g * 48:(g + 1) * 48] @ w[g, :, k, :].T.astype(np.float64)
    pre += bias
    ref = 0.5 * pre * (1.0 + np.vectorize(math.erf)(pre / math.sqrt(2.0)))
    if with_res:
        ref = ref + res
    out = torch.full((B, T, C), float("nan"), dtype=torch.float32, device="cuda")
    xd, wd, bd = dev(x), dev(w.reshape(G, 48, ntap * 48)), dev(bias)
    rd = dev(res) if with_res else None
    _lib.check(lib.ts_op_conv_taps48_timed(ctx, _lib.dptr(xd), B, T, G, ntap, _lib.dptr(wd), _lib.dptr(bd), _lib.dptr(rd) if with_res else None, 1,
                                           _lib.dptr(out), None, None))
    torch.cuda.synchronize()
    assert_close_measured(f"conv_taps48.B{B}T{T}G{G}K{ntap}", out.cpu().numpy(), ref, 2e-5)


def test_conv_banded_launch_matches_plain_tiles(hip):
    """Layers of more than one round of 512 workgroups are launched in two bands (128 x 128 tiles for the whole rounds, 64 x 128
    for the rows that are left: conv_gemm.hip `plan_bands`).  Every tile shape walks K in the same order, so the banded launch
    (tile 0 = auto) must be bit-identical to the plain 128 x 128 and 64 x 64 grids; M = 19 227 is ragged against both heights
    and N = 500 against the tile width (600 big tiles = one round of 512 = 128 row blocks, 23 row blocks left)."""
    _lib, lib, ctx = hip
    rng = np.random.default_rng(78)
    B, L, Cin, Cout, K = 3, 6409, 64, 500, 3
    x = rng.standard_normal((B, L, Cin)).astype(np.float32)
    npad = (Cout + 127) // 128 * 128
    w = np.zeros((npad, K * Cin), np.float32)
    w[:Cout] = rng.standard_normal((Cout, K * Cin)).astype(np.float32) / np.sqrt(K * Cin)
    b = np.zeros(npad, np.float32)
    b[:Cout] = rng.standard_normal(Cout).astype(np.float32)
    xd, wd, bd = dev(x), dev(w), dev(b)
    outs = {}
    for tile in (0, 1, 2, 31, 33, 39, 35, 36, 37):   # 31 / 39 / 33: the ring engine on a plain grid of 128 x 128 / 96 x 128 tiles; 35 / 36: tiles dealt to the XCDs (604 / 804 tiles: padding workgroups idle); 37: bands + dealt tiles
        out = torch.full((B, L, Cout), float("nan"), dtype=torch.float32, device="cuda")
        _lib.check(lib.ts_op_conv1d_timed(ctx, _lib.dptr(xd), B, L, Cin, _lib.dptr(wd), _lib.dptr(bd), Cout, K, tile, 1,
                                          _lib.dptr(out), None, None))
        torch.cuda.synchronize()
        outs[tile] = out.cpu().numpy()
    assert np.isfinite(outs[0]).all()
    for tile in (1, 2, 31, 33, 39, 35, 36, 37):
        assert np.array_equal(outs[0], outs[tile]), f"tile {tile} differs from the banded launch"
    rows = rng.integers(0, L, 64)
    xp = np.pad(x, ((0, 0), (1, 1), (0, 0)))
    ref = sum(xp[:, rows + k, :] @ w[:Cout, k * Cin:(k + 1) * Cin].T for k in range(K)) + b[:Cout]
    ref = np.where(ref >= 0, ref, 0.2 * ref).astype(np.float32)
    np.testing.assert_allclose(outs[0][:, rows, :], ref, atol=2e-5, rtol=1e-5)


def test_gate_activation_accuracy(hip):
    """The chain kernels' gate — tanh(v) * sigmoid(p) on v_exp_f32 / v_rcp_f32 (`kernels.h::gate_act`; reference `GatedActivation`,
    `gated_pixelcnn_v2.py:16-22`: torch.tanh * torch.sigmoid) — against float64 over the whole input range, tiny |v| and saturation
    included: the ABSOLUTE error, which is what the next layer's O(1) sums see, stays under 3e-7; the relative error of
    the tanh factor near v = 0 (the formula cancels there) is recorded."""
    _lib, lib, _ = hip
    rng = np.random.default_rng(3)
    v = np.concatenate([rng.uniform(-12, 12, 200000), rng.uniform(-1, 1, 100000) * 10.0 ** rng.uniform(-6, 0, 100000),
                        np.float64([0.0, -0.0, 1e-8, -1e-8, 30.0, -30.0, 88.0, -88.0, 1e4, -1e4])]).astype(np.float32)
    p = np.concatenate([rng.uniform(-12, 12, 300000), np.float64([0.0, 5.0, -5.0, 30.0, -30.0, 88.0, -88.0, 1e4, -1e4, 0.5])]).astype(np.float32)
    vd, pd = dev(v), dev(p)
    out = torch.empty_like(vd)
    _lib.check(lib.ts_debug_gate_act(_lib.dptr(vd), _lib.dptr(pd), _lib.dptr(out), v.size, None))
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float64)
    v64, p64 = v.astype(np.float64), p.astype(np.float64)
    ref = np.tanh(v64) / (1.0 + np.exp(-np.clip(p64, -700, 700)))
    assert np.isfinite(got).all()
    assert_close_measured("gate_act.abs", got, ref, 3e-7)
    small = (np.abs(v64) > 0) & (np.abs(v64) < 1e-2) & (p64 > -2)
    rel = np.abs(got[small] - ref[small]) / np.abs(ref[small])
    print(f"gate_act: relative error of tanh(v) * sigmoid(p) for |v| < 1e-2: median {np.median(rel):.1e}, max {rel.max():.1e} "
          f"(max |v| * rel = {float((np.abs(v64[small]) * rel).max()):.1e})")
    assert float((np.abs(v64[small]) * rel).max()) < 3e-7            # the relative error IS the absolute one over |tanh v| ~ |v|
    assert got[-10] == 0.0 and got[-9] == 0.0                          # tanh(+-0) * sigmoid = 0 exactly


def test_clock_sampler_reports_a_plausible_shader_clock(hip):
    """ts_debug_clock_sample (tools/conv_clock.py): shader cycles per 100 MHz wall-clock window on an idle device must come out
    between 1 and 2.6 GHz (MI355X: 2.4 GHz nominal), window lengths at the requested 200 us."""
    _lib, lib, ctx = hip
    n, win = 20, 200
    buf = torch.zeros(3 * n, dtype=torch.int64, device="cuda")
    _lib.check(lib.ts_debug_clock_sample(_lib.dptr(buf), n, win, _lib.stream_ptr()))
    torch.cuda.synchronize()
    r = buf.cpu().numpy().reshape(n, 3).astype(np.float64)
    ghz = r[:, 2] / (r[:, 1] * 10.0)
    assert np.all(r[:, 1] >= win * 100) and np.all(r[:, 1] < 2 * win * 100)
    assert np.all(ghz > 1.0) and np.all(ghz < 2.6), ghz


def _fma32(a, b, c):
    """float32 fma emulated through float64 (the product is exact there; the second rounding differs from a true fma only in
    ~2^-29 of the cases)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


# dim 64: the LDS-staged form (csrc/vq.hip::vq_argmin_lds_kernel: 8 rows per workgroup below 16 384 rows, 32 from there on; a codebook
# that is not a multiple of the 64-code tile; rows that do not fill the last workgroup); dim 32: the per-thread form
@pytest.mark.parametrize("M,ncode,dim", [(1, 128, 64), (13, 2048, 64), (2400, 2048, 64), (19200, 2048, 64), (16397, 200, 64), (45, 96, 32)])
def test_op_vq_argmin(hip, M, ncode, dim):
    """Index work is bit-exact: (a) on operands whose every product and partial sum is exactly representable in fp32 (multiples
    of 1/64 in [-4, 4]) the distances are exact whatever the summation order, so the indices must equal numpy's float64 argmin,
    exact ties -> lowest index, no exceptions; (b) on random fp32 operands the indices must equal the argmin of the distances
    recomputed IN THE KERNEL'S ORDER (|x|^2 and |e|^2 as sequential fma chains, the dot product as a sequential fma chain,
    (|x|^2 + |e|^2) - 2 dot), again without exceptions."""
    _lib, lib, ctx = hip
    rng = np.random.default_rng(M)

    def run(x, cb):
        idx = torch.empty(M, dtype=torch.int64, device="cuda")
        xd, cbd = dev(x), dev(cb)           # keep the device tensors alive across the call
        _lib.check(lib.ts_op_vq_argmin(ctx, _lib.dptr(xd), M, _lib.dptr(cbd), ncode, dim, _lib.dptr(idx), None))
        return idx.cpu().numpy()

    # (a) exactly representable arithmetic, with a block of duplicated codes (exact ties everywhere)
    xq = (rng.integers(-256, 257, (M, dim)) / 64.0).astype(np.float32)
    cq = (rng.integers(-256, 257, (ncode, dim)) / 64.0).astype(np.float32)
    cq[ncode // 2:ncode // 2 + 16] = cq[5:21]
    xq[0] = cq[9]
    d64 = (xq.astype(np.float64) ** 2).sum(1, keepdims=True) + (cq.astype(np.float64) ** 2).sum(1)[None] - 2.0 * xq.astype(np.float64) @ cq.astype(np.float64).T
    np.testing.assert_array_equal(run(xq, cq), d64.argmin(1))
    # (b) random operands, distances in the kernel's order
    x = rng.standard_normal((M, dim)).astype(np.float32)
    cb = rng.standard_normal((ncode, dim)).astype(np.float32)
    cb[7] = cb[3]                       # an exact tie between codes 3 and 7 ...
    x[0] = cb[3]                        # ... which row 0 hits: lowest index must win
    xsq, csq = np.zeros(M, np.float32), np.zeros(ncode, np.float32)
    dot = np.zeros((M, ncode), np.float32)
    for c in range(dim):
        xsq = _fma32(x[:, c], x[:, c], xsq)
        csq = _fma32(cb[:, c], cb[:, c], csq)
        dot = _fma32(x[:, c:c + 1], cb[None, :, c], dot)
    d = (xsq[:, None] + csq[None, :]) - np.float32(2.0) * dot
    got = run(x, cb)
    assert got[0] == 3
    np.testing.assert_array_equal(got, d.argmin(1))


@pytest.mark.parametrize("M,K,N,relu", [(32, 256, 512, 0), (64, 512, 256, 1), (3, 64, 128, 0), (32, 512, 2048, 0),
                                        (40, 1536, 96, 0)])
def test_op_linear(hip, M, K, N, relu):
    _lib, lib, ctx = hip
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    out = torch.empty((M, N), dtype=torch.float32, device="cuda")
    xd = dev(x)
    _lib.check(lib.ts_op_linear(ctx, _lib.dptr(xd), M, K, _lib.fptr(w), _lib.fptr(b), N, relu, _lib.dptr(out), None))
    ref = x @ w.T + b
    if relu:
        ref = np.maximum(ref, 0)
    np.testing.assert_allclose(out.cpu().numpy(), ref, atol=2e-5, rtol=1e-5)


def test_op_sample(hip):
    _lib, lib, ctx = hip
    rng = np.random.default_rng(5)
    B, V = 32, 2048
    logits = (rng.standard_normal((B, V)) * 3).astype(np.float32)
    logits[1, 100] = logits[1, 900] = logits[1].max() + 1.0          # tie: lowest index wins
    ld = dev(logits)
    idx = torch.empty(B, dtype=torch.int64, device="cuda")
    _lib.check(lib.ts_op_sample(ctx, _lib.dptr(ld), B, V, _lib.TS_SAMPLE_GREEDY, None, _lib.dptr(idx), None))
    got = idx.cpu().numpy()
    np.testing.assert_array_equal(got, np.argmax(logits, -1))
    assert got[1] == 100
    u = rng.random(B).astype(np.float32)
    u[0], u[2] = 0.0, np.float32(1.0 - 2 ** -24)
    ud = dev(u)
    _lib.check(lib.ts_op_sample(ctx, _lib.dptr(ld), B, V, _lib.TS_SAMPLE_UNIFORMS, _lib.dptr(ud), _lib.dptr(idx), None))
    ref = O.sample_inverse_cdf(logits, u)
    got = idx.cpu().numpy()
    # exact: the oracle restates the kernel's summation order AND its exponential (det_expf: fp32 multiplies / adds only)
    np.testing.assert_array_equal(got, ref)
    assert got[0] == 0 or logits[0, :got[0]].max() < logits[0].max() - 86.0      # u = 0 -> the first class with any mass


# ----------------------------------------------------------------------------------------------- modules vs golden
def _vq_module(cfg):
    from talkshow_amd.modules import VQVAE
    in_dim, emb, n_emb, hid, layers, salt, seed = [int(v) for v in cfg]
    m = VQVAE(in_dim, emb, n_emb, hid, layers).cuda()
    m.load_state_dict(synth.to_torch(synth.vqvae_state_dict(seed=seed, in_dim=in_dim, embedding_dim=emb,
                                                            num_embeddings=n_emb, num_hiddens=hid,
                                                            num_residual_layers=layers, salt=salt)))
    return m


@pytest.mark.parametrize("name", ["vq_small", "vq_full_body", "vq_full_hand"])
def test_vqvae_golden(hip, golden, name):
    g = golden(name)
    m = _vq_module(g["cfg"])
    z, q, lat = m.encode_nlc(g["poses"], want_z=True)
    np.testing.assert_allclose(z.cpu().numpy().transpose(0, 2, 1), g["z"], atol=2e-5, rtol=0)
    np.testing.assert_array_equal(lat.cpu().numpy(), g["idx"])
    np.testing.assert_array_equal(q.cpu().numpy().transpose(0, 2, 1), g["quantized"])
    recon = m.decode_nlc(lat)
    np.testing.assert_allclose(recon.cpu().numpy().transpose(0, 2, 1), g["recon"], atol=1e-4, rtol=0)
    # reference call shapes
    e, x_recon = m(torch.from_numpy(g["poses"]))
    np.testing.assert_allclose(x_recon.cpu().numpy(), g["recon"], atol=1e-4, rtol=0)
    rec2, none = m.decode(b=lat.shape[0], w=lat.shape[1], latents=lat)
    assert none is None
    np.testing.assert_array_equal(rec2.cpu().numpy(), x_recon.cpu().numpy())
    # VQVAE.decode(e=...) (`vqvae_1d.py:201-203`): the decoder on given continuous latents — here the quantised ones, so the
    # result is the reference's reconstruction again (the latents= path gathers rows of a table with aft_vq_conv pre-applied,
    # this one multiplies: same values to rounding)
    rec3, none = m.decode(b=lat.shape[0], w=lat.shape[1], e=torch.from_numpy(g["quantized"]))
    assert none is None and rec3.shape == rec2.shape
    np.testing.assert_allclose(rec3.cpu().numpy(), g["recon"], atol=1e-4, rtol=0)


def test_audioenc_golden(hip, golden):
    from talkshow_amd.modules import AudioEncoder
    g = golden("audioenc_full")
    m = AudioEncoder(64, 256, 2, 256).cuda()
    m.load_state_dict(synth.to_torch(synth.audioencoder_state_dict(seed=7)))
    out = m(torch.from_numpy(g["mfcc"]).transpose(1, 2))
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], atol=2e-5, rtol=0)


def _pix_module(cfg):
    from talkshow_amd.modules import GatedPixelCNN
    input_dim, dim, n_layers, n_cls, seed = [int(v) for v in cfg]
    m = GatedPixelCNN(input_dim, dim, n_layers, n_cls, True, True).cuda()
    m.load_state_dict(synth.to_torch(synth.pixelcnn_state_dict(seed=seed, input_dim=input_dim, dim=dim,
                                                               n_layers=n_layers, n_classes=n_cls)))
    return m


@pytest.mark.parametrize("name", ["pix_small", "pix_full"])
def test_pixelcnn_golden(hip, golden, name):
    from talkshow_amd import _lib
    g = golden(name)
    m = _pix_module(g["cfg"])
    # 1. teacher forced: logits of every position, no error propagation (GatedPixelCNN.forward)
    _, logits = m.run(g["label"], g["aud"], mode=_lib.TS_TEACHER_FORCED, codes=g["codes"], want_logits=True)
    assert_close_measured(f"{name}.teacher_forced_logits", logits.cpu().numpy(), g["full_logits"], LOGIT_ATOL)
    # 2. free-running greedy decode: bit-exact code indices
    codes, step_logits = m.run(g["label"], g["aud"], mode=_lib.TS_SAMPLE_GREEDY, want_logits=True)
    assert_close_measured(f"{name}.step_logits", step_logits.cpu().numpy(), g["step_logits"], LOGIT_ATOL)
    np.testing.assert_array_equal(codes.cpu().numpy(), g["codes"])
    # 3. reference call shape of forward: (B,H,2) codes, (B,256,H,2) audio -> (B,V,H,2)
    aud4 = torch.from_numpy(g["aud"]).cuda().transpose(1, 2).unsqueeze(-1).repeat(1, 1, 1, 2)
    full = m(torch.from_numpy(g["codes"]).cuda(), torch.from_numpy(g["label"]).cuda(), aud4)
    np.testing.assert_array_equal(full.permute(0, 2, 3, 1).cpu().numpy(), logits.cpu().numpy())


@pytest.mark.parametrize("tag,audio,bh", [("noaud_bh", False, True), ("aud_v", True, False), ("noaud_v", False, False)])
def test_pixelcnn_constructor_variants(hip, golden, tag, audio, bh):
    """GatedPixelCNN(audio=False and / or bh_model=False) (`gated_pixelcnn_v2.py:37-42,80-85,137-150`) against the reference's own
    module: teacher-forced logits of every position, free-running greedy codes bit-exact, the reference call shapes of forward /
    generate, injected-uniform sampling against the oracle, and the continuity prefix.  bh_model=False is `ts_pixelcnn_v_*` (single
    vertical stack; the 4-column grid of the third case draws a row's four codes together)."""
    from talkshow_amd import _lib
    from talkshow_amd.modules import GatedPixelCNN
    g = golden("pix_variants")
    input_dim, dim, n_layers, n_cls, seed = [int(v) for v in g["cfg"]]
    sd = synth.pixelcnn_state_dict(seed=seed, input_dim=input_dim, dim=dim, n_layers=n_layers, n_classes=n_cls, audio=audio, bh_model=bh)
    m = GatedPixelCNN(input_dim, dim, n_layers, n_cls, audio, bh).cuda()
    m.load_state_dict(synth.to_torch(sd))
    ref = g[tag + "_codes"]
    B, H, W = ref.shape
    aud = g["aud"] if audio else None
    _, logits = m.run(g["label"], aud, mode=_lib.TS_TEACHER_FORCED, codes=ref, want_logits=True, shape=(B, H, W))
    assert_close_measured(f"pix_variants.{tag}.teacher_forced_logits", logits.cpu().numpy(), g[tag + "_full_logits"], LOGIT_ATOL)
    codes, step = m.run(g["label"], aud, mode=_lib.TS_SAMPLE_GREEDY, want_logits=True, shape=(B, H, W))
    assert_close_measured(f"pix_variants.{tag}.step_logits", step.cpu().numpy(), g[tag + "_step_logits"], LOGIT_ATOL)
    np.testing.assert_array_equal(codes.cpu().numpy(), ref)
    # reference call shapes: forward(x, label[, aud (B,256,H,W)]) -> (B,V,H,W); generate(label, shape, batch_size[, aud_feat])
    aud4 = torch.from_numpy(g["aud"]).cuda().transpose(1, 2).unsqueeze(-1).repeat(1, 1, 1, W) if audio else None
    full = m(torch.from_numpy(ref).cuda(), torch.from_numpy(g["label"]).cuda(), aud4) if audio else m(torch.from_numpy(ref).cuda(), torch.from_numpy(g["label"]).cuda())
    np.testing.assert_array_equal(full.permute(0, 2, 3, 1).cpu().numpy(), logits.cpu().numpy())
    gen = m.generate(torch.from_numpy(g["label"]).cuda(), shape=(H, W), batch_size=B, aud_feat=aud4, mode=_lib.TS_SAMPLE_GREEDY)
    np.testing.assert_array_equal(gen.cpu().numpy(), ref)
    # stochastic decode on injected uniforms == the oracle's inverse-CDF generate; device Philox == oracle Philox (position r * W + j)
    u = np.zeros((B, H, W), np.float32)
    for b in range(B):
        for r in range(H):
            for j in range(W):
                u[b, r, j] = O.philox_uniform(77, 5 + b, r * W + j)
    aud_o = np.repeat(g["aud"].transpose(0, 2, 1)[:, :, :, None], W, axis=3) if audio else None
    want = O.pixelcnn_generate(g["label"], aud_o, sd, n_layers, H, uniforms=u, audio=audio, bh_model=bh, W=W)
    got_u, _ = m.run(g["label"], aud, mode=_lib.TS_SAMPLE_UNIFORMS, uniforms=u, shape=(B, H, W))
    got_p, _ = m.run(g["label"], aud, mode=_lib.TS_SAMPLE_PHILOX, seed=77, clip_index0=5, shape=(B, H, W))
    np.testing.assert_array_equal(got_u.cpu().numpy(), want)
    np.testing.assert_array_equal(got_p.cpu().numpy(), want)
    # continuity prefix: the greedy tail behind the greedy head == the single greedy run
    H0 = 3
    tail, _ = m.run(g["label"], g["aud"][:, H0:] if audio else None, mode=_lib.TS_SAMPLE_GREEDY, pre_codes=ref[:, :H0],
                    pre_aud=g["aud"][:, :H0] if audio else None, shape=(B, H - H0, W))
    np.testing.assert_array_equal(tail.cpu().numpy(), ref[:, H0:])


def test_pixelcnn_sampling_and_prefix(hip, golden):
    """Stochastic decode with injected uniforms == oracle's inverse-CDF generate; continuity prefix reproduces the tail."""
    from talkshow_amd import _lib
    g = golden("pix_small")
    input_dim, dim, n_layers, n_cls, seed = [int(v) for v in g["cfg"]]
    m = _pix_module(g["cfg"])
    sd = synth.pixelcnn_state_dict(seed=seed, input_dim=input_dim, dim=dim, n_layers=n_layers, n_classes=n_cls)
    B, H = g["codes"].shape[:2]
    aud4 = np.repeat(g["aud"].transpose(0, 2, 1)[:, :, :, None], 2, axis=3)
    u = O.philox_uniforms(1234, 10, B, H)
    ref = O.pixelcnn_generate(g["label"], aud4, sd, n_layers, H, uniforms=u)
    got_u, _ = m.run(g["label"], g["aud"], mode=_lib.TS_SAMPLE_UNIFORMS, uniforms=u)
    got_p, _ = m.run(g["label"], g["aud"], mode=_lib.TS_SAMPLE_PHILOX, seed=1234, clip_index0=10)
    np.testing.assert_array_equal(got_u.cpu().numpy(), got_p.cpu().numpy())     # device Philox == oracle Philox
    np.testing.assert_array_equal(got_u.cpu().numpy(), ref)
    # shard invariance: clip 1 alone, addressed as global clip 11, draws what it drew inside the batch
    solo, _ = m.run(g["label"][1:2], g["aud"][1:2], mode=_lib.TS_SAMPLE_PHILOX, seed=1234, clip_index0=11)
    np.testing.assert_array_equal(solo.cpu().numpy()[0], got_p.cpu().numpy()[1])
    # continuity prefix (gated_pixelcnn_v2.py:158-165): greedy tail given the greedy head == the single greedy run
    H0 = 4
    tail, _ = m.run(g["label"], g["aud"][:, H0:], mode=_lib.TS_SAMPLE_GREEDY, pre_codes=g["codes"][:, :H0],
                    pre_aud=g["aud"][:, :H0])
    np.testing.assert_array_equal(tail.cpu().numpy(), g["codes"][:, H0:])
    # ... and the SAMPLED tail given the sampled head == the single sampled run: the Philox position of a code is its absolute
    # (row, column), prefix rows included (the prefix path used to restart at position 0 and replay chunk 0's numbers)
    tail_p, _ = m.run(g["label"], g["aud"][:, H0:], mode=_lib.TS_SAMPLE_PHILOX, seed=1234, clip_index0=10,
                      pre_codes=got_p[:, :H0], pre_aud=g["aud"][:, :H0])
    np.testing.assert_array_equal(tail_p.cpu().numpy(), got_p.cpu().numpy()[:, H0:])


def test_wrapper_sampling_defaults(hip, golden, tmp_path):
    """like the reference, sampling entry points draw from torch's generator when no seed is given (two calls differ,
    `torch.manual_seed` reproduces them), and `infer_on_audio(continuity=True, uniforms=...)` hands the uniforms on to both parts."""
    from nets.init_model import init_model
    import nets.smplx_body_pixel as bp
    g = golden("body_e2e_full")
    w = init_model("s2g_body_pixel", argparse.Namespace(gpu=0, infer=True), _config(tmp_path))
    w.load_state_dict({"generator": synth.to_torch(synth.pixelcnn_state_dict(seed=7)),
                       "audioencoder": synth.to_torch(synth.audioencoder_state_dict(seed=7))})
    torch.manual_seed(5)
    a, _ = w.generate_batch(g["mfcc"], g["ids"])
    b, _ = w.generate_batch(g["mfcc"], g["ids"])
    torch.manual_seed(5)
    a2, _ = w.generate_batch(g["mfcc"], g["ids"])
    assert not torch.equal(a, b) and torch.equal(a, a2)
    # continuity with injected uniforms: the two parts consume the two halves of the (B, H, 2) array
    import pytest as _pt
    mp = _pt.MonkeyPatch()
    try:
        gap = 60
        mp.setattr(bp, "get_mfcc_sepa", lambda *a_, **k_: (g["mfcc"][0].copy(), gap))
        u = O.philox_uniforms(9, 0, 1, 75)
        out = w.infer_on_audio("clip.wav", id=torch.tensor([1]), fps=30, continuity=True, uniforms=u)
        out2 = w.infer_on_audio("clip.wav", id=torch.tensor([1]), fps=30, continuity=True, uniforms=u)
        assert out.shape == (1, 300, 129) and np.array_equal(out, out2)
    finally:
        mp.undo()


def test_pixelcnn_streaming_equals_one_call(hip, golden):
    """ts_pixelcnn_stream_*: ragged chunks behind a persistent row cache reproduce the one-shot result bit for bit —
    greedy vs the reference golden, stochastic (Philox position = absolute row) vs the one-shot call."""
    from talkshow_amd import _lib
    g = golden("pix_small")
    m = _pix_module(g["cfg"])
    B, H = g["codes"].shape[:2]
    for chunks in ((3, 1, 4, 2), (1,) * H, (H,), (5, 5)):
        st = m.open_stream(g["label"], B, max(chunks))
        got, at = [], 0
        for hc in chunks:
            got.append(st.step(g["aud"][:, at:at + hc], mode=_lib.TS_SAMPLE_GREEDY))
            at += hc
        assert st.rows == H
        st.close()
        np.testing.assert_array_equal(torch.cat(got, 1).cpu().numpy(), g["codes"])
    ref, _ = m.run(g["label"], g["aud"], mode=_lib.TS_SAMPLE_PHILOX, seed=77, clip_index0=5)
    st = m.open_stream(g["label"], B, 4)
    got = [st.step(g["aud"][:, a:a + 4], mode=_lib.TS_SAMPLE_PHILOX, seed=77, clip_index0=5) for a in range(0, H, 4)]
    np.testing.assert_array_equal(torch.cat(got, 1).cpu().numpy(), ref.cpu().numpy())
    u = O.philox_uniforms(5, 0, B, H)
    ref_u, _ = m.run(g["label"], g["aud"], mode=_lib.TS_SAMPLE_UNIFORMS, uniforms=u)
    st2 = m.open_stream(g["label"], B, 3)
    got = [st2.step(g["aud"][:, a:a + 3], mode=_lib.TS_SAMPLE_UNIFORMS, uniforms=u[:, a:a + 3]) for a in range(0, H, 3)]
    np.testing.assert_array_equal(torch.cat(got, 1).cpu().numpy(), ref_u.cpu().numpy())
    with pytest.raises(RuntimeError, match="max_chunk_rows"):
        st2.step(g["aud"][:, :4], mode=_lib.TS_SAMPLE_GREEDY)


def test_streaming_60s_clip_cost_independent_of_history(hip):
    """A 60 s clip (450 code rows, full-size network) generated in 2 s chunks (15 rows) == the same clip in ONE call, and a
    late chunk costs what an early one costs (the reference's prefix form re-runs the whole history for every chunk)."""
    import time
    from talkshow_amd import _lib
    from talkshow_amd.modules import GatedPixelCNN
    m = GatedPixelCNN(2048, 256, 15, 4, True, True).cuda()
    m.load_state_dict(synth.to_torch(synth.pixelcnn_state_dict(seed=7)))
    B, H, HC = 2, 450, 15
    rng = np.random.default_rng(60)
    aud = torch.from_numpy(rng.standard_normal((B, H, 256)).astype(np.float32)).cuda()
    label = synth.speaker_ids(B)
    one, _ = m.run(label, aud, mode=_lib.TS_SAMPLE_GREEDY)
    st = m.open_stream(label, B, HC)
    got, cost = [], []
    for a in range(0, H, HC):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got.append(st.step(aud[:, a:a + HC], mode=_lib.TS_SAMPLE_GREEDY))
        torch.cuda.synchronize()
        cost.append(time.perf_counter() - t0)
    assert torch.equal(torch.cat(got, 1), one)
    early, late = np.median(cost[6:12]), np.median(cost[-6:])     # past the graph captures of the first phases
    assert late < 1.25 * early, (early, late)
    st.close()


# ----------------------------------------------------------------------------------------------- wrappers (nets.*)
def _config(tmp_path, which="body_pixel"):
    vq_path = str(tmp_path / "vq.pth")
    torch.save({"generator": {"g_body": synth.to_torch(synth.vqvae_state_dict(seed=7, in_dim=39)),
                              "g_hand": synth.to_torch(synth.vqvae_state_dict(seed=7, in_dim=90, salt=1))}}, vq_path)
    from talkshow_amd.config import Object
    cfg = json.load(open(os.path.join(REPO, "config", which + ".json")))
    if "vq_path" in cfg["Model"]:
        cfg["Model"]["vq_path"] = vq_path
    return Object(cfg)


def test_wrapper_body_pixel_e2e(hip, golden, tmp_path):
    """nets.init_model -> s2g_body_pixel -> load_state_dict (module.-prefixed) -> infer_on_audio, vs reference goldens."""
    from nets.init_model import init_model
    g = golden("body_e2e_full")
    args = argparse.Namespace(gpu=0, infer=True)
    w = init_model("s2g_body_pixel", args, _config(tmp_path))
    gen = synth.to_torch(synth.pixelcnn_state_dict(seed=7))
    w.load_state_dict({"generator": {("module." + k): v for k, v in gen.items()},
                       "audioencoder": synth.to_torch(synth.audioencoder_state_dict(seed=7)),
                       "generator_optim": {"state": {}, "param_groups": []}})
    codes, poses = w.generate_batch(g["mfcc"], g["ids"], mode=0)
    feat = w.audioencoder.forward_nlc(torch.from_numpy(g["mfcc"]).cuda())
    np.testing.assert_allclose(feat.cpu().numpy(), g["aud_feat"], atol=2e-5, rtol=0)
    np.testing.assert_array_equal(codes.cpu().numpy(), g["codes"])                       # bit-exact greedy codes
    np.testing.assert_allclose(poses.cpu().numpy(), g["poses"], atol=1e-4, rtol=0)       # 1e-4 on pose floats
    # the reference entry point: one clip's features repeated B times, speaker id tensor of shape (1,)
    out = w.infer_on_audio(g["mfcc"][0], id=torch.tensor([int(g["ids"][0])]).cuda(), fps=30, B=2, greedy=True,
                           txgfile=None, smooth=False)
    assert out.shape == (2, 300, 129) and out.dtype == np.float32
    np.testing.assert_allclose(out[0], g["poses"][0], atol=1e-4, rtol=0)
    np.testing.assert_array_equal(out[0], out[1])
    # state_dict round trip keeps the reference key scheme (mask-A taps zeroed as the reference leaves them)
    sd = w.state_dict()
    assert set(sd) >= {"generator", "audioencoder"} and len(sd["generator"]) == 146
    assert float(sd["generator"]["layers.0.vert_stack.weight"][:, :, -1].abs().max()) == 0.0


def test_wrapper_continuity_and_id_broadcast(hip, tmp_path):
    """`infer_on_audio(continuity=True)` through the wrapper (`smplx_body_pixel.py:245-269,291-304`) against the oracle's
    restatement of the same two-chunk procedure (greedy), `id=None` with B > 1 (one label broadcast over the batch, as
    nn.Embedding does), and the IndexError for a label outside [0, 4)."""
    from nets.init_model import init_model
    args = argparse.Namespace(gpu=0, infer=True)
    w = init_model("s2g_body_pixel", args, _config(tmp_path))
    sd_p, sd_a = synth.pixelcnn_state_dict(seed=7), synth.audioencoder_state_dict(seed=7)
    w.load_state_dict({"generator": synth.to_torch(sd_p), "audioencoder": synth.to_torch(sd_a)})
    sd_b, sd_h = synth.vqvae_state_dict(seed=7, in_dim=39), synth.vqvae_state_dict(seed=7, in_dim=90, salt=1)
    T = 88                                                     # 2 s head (60 frames -> 15 code rows) + 28 frames (7 rows)
    feat = synth.mfcc_features(123, 1, T)[0]                   # (T, 64): what get_mfcc_sepa returns for a wav
    out = w.infer_on_audio(feat, id=torch.tensor([2]).cuda(), fps=30, sr=22000, B=1, continuity=True, greedy=True)
    gap = 1 + 44000 // 734
    ref, ref_codes = O.body_pixel_infer_continuity(feat[None], gap, np.asarray([2]), sd_a, sd_p, sd_b, sd_h)
    assert out.shape == ref.shape == (1, 4 * (15 + 7), 129)
    np.testing.assert_allclose(out, ref, atol=1e-4, rtol=0)
    # id=None, B=3: label 0 for every clip of the batch (the reference passes a (1,) tensor to nn.Embedding)
    mf = synth.mfcc_features(124, 1, 40)[0]
    o3 = w.infer_on_audio(mf, id=None, fps=30, B=3, greedy=True)
    o1 = w.infer_on_audio(mf, id=torch.tensor([0]).cuda(), fps=30, B=1, greedy=True)
    assert o3.shape == (3, 40, 129)
    for b in range(3):
        np.testing.assert_array_equal(o3[b], o1[0])
    with pytest.raises(IndexError):
        w.infer_on_audio(mf, id=torch.tensor([4]).cuda(), fps=30, B=1, greedy=True)
    with pytest.raises(IndexError):
        w.generate_batch(np.repeat(mf[None], 2, 0), np.asarray([0, -1]), mode=0)
    with pytest.raises(ValueError):
        w.generate_batch(np.repeat(mf[None], 3, 0), np.asarray([0, 1]), mode=0)


def test_wrapper_body_vq_e2e(hip, golden, tmp_path):
    from nets.init_model import init_model
    g = golden("body_vq_e2e_full")
    args = argparse.Namespace(gpu=0, infer=True)
    w = init_model("s2g_body_vq", args, _config(tmp_path, "body_vq"))
    w.load_state_dict({"g_body": synth.to_torch(synth.vqvae_state_dict(seed=7, in_dim=39)),
                       "g_hand": synth.to_torch(synth.vqvae_state_dict(seed=7, in_dim=90, salt=1))})
    B, T = g["poses129"].shape[:2]
    full = np.zeros((B, 165, T), np.float32)
    full[:, g["c_index"], :] = g["poses129"].transpose(0, 2, 1)
    out = w.infer_on_audio(torch.zeros(B, 64, T), initial_pose=torch.from_numpy(full), id=torch.tensor([0]), fps=30)
    assert out.shape == g["out"].shape
    np.testing.assert_allclose(out, g["out"], atol=1e-4, rtol=0)
    codes, _ = w.reconstruct_batch(g["poses129"])
    np.testing.assert_array_equal(codes.cpu().numpy(), g["codes"])


def test_wrapper_body_vq_single_network(hip, tmp_path):
    """`composition=false` (`smplx_body_vq.py:45-46,277`): ONE VQ-VAE over all 129 dims — small widths, vs the oracle."""
    from nets.init_model import init_model
    cfg = _config(tmp_path, "body_vq")
    cfg.Model.composition = False
    w = init_model("s2g_body_vq", argparse.Namespace(gpu=0, infer=True), cfg)
    sd = synth.vqvae_state_dict(seed=11, in_dim=129)
    w.load_state_dict({"g": synth.to_torch(sd)})
    assert set(w.state_dict()) == {"g", "g_optim", "discriminator", "discriminator_optim"}
    B, T = 2, 24
    p129 = synth.gt_poses(23, B, T)
    full = np.zeros((B, 165, T), np.float32)
    full[:, w.c_index, :] = p129.transpose(0, 2, 1)
    out = w.infer_on_audio(None, initial_pose=torch.from_numpy(full), fps=30)
    _, recon, _ = O.vqvae_forward(p129, sd)                                     # (B,129,T)
    ref = np.concatenate(list(recon.transpose(0, 2, 1)), axis=1)               # np.concatenate(output, axis=1)
    assert out.shape == ref.shape == (T, B * 129)
    np.testing.assert_allclose(out, ref, atol=1e-4, rtol=0)


def test_wrapper_body_ae_extract(hip, golden, tmp_path):
    """nets.init_model('s2g_body_ae') -> load_state_dict({'g': ...}) -> extract(): the FGD feature extractor
    (`body_ae.py:145-152`) vs the golden produced by the reference wrapper; AE.forward's reconstruction too."""
    from nets.init_model import init_model
    g = golden("ae_full")
    w = init_model("s2g_body_ae", argparse.Namespace(gpu=0, infer=True), _config(tmp_path))
    w.load_state_dict({"g": synth.to_torch(synth.ae_state_dict(seed=7))})
    B, T = g["poses129"].shape[:2]
    wide = np.zeros((B, T, 165), np.float32)
    wide[:, :, w.c_index] = g["poses129"]
    feat, x129 = w.extract(torch.from_numpy(wide))
    assert feat.shape == (B, T // 4, 64) and x129.shape == (B, T, 129)
    np.testing.assert_allclose(feat.cpu().numpy(), g["feat"], atol=2e-5, rtol=0)
    np.testing.assert_array_equal(x129.cpu().numpy(), g["poses129"])
    z, recon = w.g(torch.from_numpy(g["poses129"]))
    np.testing.assert_allclose(z.cpu().numpy(), g["z"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(recon.cpu().numpy(), g["recon"], atol=1e-4, rtol=0)
    assert len(w.state_dict()["g"]) == 210                                      # the reference AE's key count


def test_wrappers_convert_to_6d_vs_reference(hip, golden, tmp_path):
    """`convert_to_6d=true` through `nets.s2g_body_vq` (`smplx_body_vq.py:50-53`: 78 + 180 modelled dims gathered by `c_index_6d` from
    330-wide rows) and `nets.s2g_body_ae` (`body_ae.py:50-53`) against what the reference's own wrappers returned
    (tests/golden/make_golden.py --only wrappers_6d).  No shipped config uses the branch; the drop-in supports it as the reference does."""
    from nets.init_model import init_model
    from talkshow_amd.config import Object
    from talkshow_amd.pose_index import c_index_6d
    g = golden("wrappers_6d")
    assert np.array_equal(g["c_index"], c_index_6d)
    args = argparse.Namespace(gpu=0, infer=True)
    vcfg = json.load(open(os.path.join(REPO, "config", "body_vq.json")))
    vcfg["Data"]["pose"]["convert_to_6d"] = True
    w = init_model("s2g_body_vq", args, Object(vcfg))
    assert w.each_dim[1:3] == [78, 180] and len(w.c_index) == 258
    w.load_state_dict({"g_body": synth.to_torch(synth.vqvae_state_dict(seed=9, in_dim=78)),
                       "g_hand": synth.to_torch(synth.vqvae_state_dict(seed=9, in_dim=180, salt=1))})
    B, T = g["poses258"].shape[:2]
    full = np.zeros((B, 330, T), np.float32)
    full[:, c_index_6d, :] = g["poses258"].transpose(0, 2, 1)
    out = w.infer_on_audio(torch.zeros(B, 64, T), initial_pose=torch.from_numpy(full), id=torch.tensor([0]), fps=30)
    assert out.shape == g["vq_out"].shape == (T, B * 258)
    np.testing.assert_allclose(out, g["vq_out"], atol=1e-4, rtol=0)
    codes, _ = w.reconstruct_batch(g["poses258"])
    np.testing.assert_array_equal(codes.cpu().numpy(), g["vq_codes"])
    cfg = json.load(open(os.path.join(REPO, "config", "body_pixel.json")))
    cfg["Data"]["pose"]["convert_to_6d"] = True
    cfg["Model"]["vq_path"] = str(tmp_path / "unused.pth")
    a = init_model("s2g_body_ae", args, Object(cfg))
    a.load_state_dict({"g": synth.to_torch(synth.ae_state_dict(seed=9, in_dim=258))})
    wide = np.zeros((B, T, 330), np.float32)
    wide[:, :, c_index_6d] = g["poses258"]
    feat, x258 = a.extract(torch.from_numpy(wide))
    np.testing.assert_array_equal(x258.cpu().numpy(), g["poses258"])
    np.testing.assert_allclose(feat.cpu().numpy(), g["ae_feat"], atol=2e-5, rtol=0)


# ----------------------------------------------------------------------------------------------- full-size properties
def test_full_size_properties(hip, tmp_path):
    """BASELINE batch (32 clips x 10 s): determinism, batch-composition independence, encode(decode) idempotence."""
    from nets.init_model import init_model
    from talkshow_amd import _lib
    args = argparse.Namespace(gpu=0, infer=True)
    w = init_model("s2g_body_pixel", args, _config(tmp_path))
    w.load_state_dict({"generator": synth.to_torch(synth.pixelcnn_state_dict(seed=7)),
                       "audioencoder": synth.to_torch(synth.audioencoder_state_dict(seed=7))})
    B, T = 32, 300
    mf, ids = synth.mfcc_features(31, B, T), synth.speaker_ids(B)
    c1, p1 = w.generate_batch(mf, ids, mode=_lib.TS_SAMPLE_GREEDY)
    c2, p2 = w.generate_batch(mf, ids, mode=_lib.TS_SAMPLE_GREEDY)
    assert torch.equal(c1, c2) and torch.equal(p1, p2)                                   # run-to-run determinism
    assert c1.shape == (B, 75, 2) and p1.shape == (B, 300, 129) and torch.isfinite(p1).all()
    assert int(c1.min()) >= 0 and int(c1.max()) < 2048 and c1.unique().numel() > 100
    c3, p3 = w.generate_batch(mf[5:9], ids[5:9], mode=_lib.TS_SAMPLE_GREEDY)             # a clip's result does not
    assert torch.equal(c3, c1[5:9]) and torch.equal(p3, p1[5:9])                         # depend on its batch
    # stochastic: Philox subsequence = global clip index -> sharding invariant
    s_all, _ = w.generate_batch(mf, ids, mode=_lib.TS_SAMPLE_PHILOX, seed=99, clip_index0=0)
    s_sub, _ = w.generate_batch(mf[16:], ids[16:], mode=_lib.TS_SAMPLE_PHILOX, seed=99, clip_index0=16)
    assert torch.equal(s_all[16:], s_sub) and not torch.equal(s_all, c1)
    # VQ round trip: decode(codes) re-encoded and decoded again is a fixed point of quantise o decode o encode
    # only where the encoder maps back onto the same codes; the size-independent property we can assert for random
    # weights is idempotence of decode: same codes -> same poses, and decode is per-clip.
    body = w.g_body.decode_nlc(c1[..., 0].contiguous())
    np.testing.assert_array_equal(body.cpu().numpy(), p1[..., :39].cpu().numpy())


def test_wrapper_body_pixel_convert_to_6d(hip, tmp_path):
    """`convert_to_6d=true` (`smplx_body_pixel.py:48-52`): 6-D rotations double the modelled widths (78 body + 180 hand dims) and
    the code predictor becomes GatedPixelCNN(2048, dim 512, 10 layers).  No shipped config uses it and the reference holds no
    golden for it: the wrapper's result on short clips is checked against the oracle (full-grid numpy restatement, pinned to
    the reference on the 3-d shapes) built from the same state dicts."""
    from nets.init_model import init_model
    from talkshow_amd import _lib
    from talkshow_amd.config import Object
    from talkshow_amd.pose_index import c_index_6d
    sd_b = synth.vqvae_state_dict(seed=9, in_dim=78)
    sd_h = synth.vqvae_state_dict(seed=9, in_dim=180, salt=1)
    vq_path = str(tmp_path / "vq6d.pth")
    torch.save({"generator": {"g_body": synth.to_torch(sd_b), "g_hand": synth.to_torch(sd_h)}}, vq_path)
    cfg = json.load(open(os.path.join(REPO, "config", "body_pixel.json")))
    cfg["Model"]["vq_path"] = vq_path
    cfg["Data"]["pose"]["convert_to_6d"] = True
    w = init_model("s2g_body_pixel", argparse.Namespace(gpu=0, infer=True), Object(cfg))
    assert w.each_dim[1:3] == [78, 180] and w.generator.dim == 512 and w.generator.n_layers == 10 and len(w.c_index) == 258
    assert np.array_equal(w.c_index, c_index_6d)
    sd_p = synth.pixelcnn_state_dict(seed=9, dim=512, n_layers=10)
    sd_a = synth.audioencoder_state_dict(seed=9)
    w.load_state_dict({"generator": synth.to_torch(sd_p), "audioencoder": synth.to_torch(sd_a)})
    B, T = 3, 24                                                     # 6 code rows: the oracle recomputes the whole grid per position
    mf, ids = synth.mfcc_features(19, B, T), synth.speaker_ids(B)
    codes, poses = w.generate_batch(mf, ids, mode=_lib.TS_SAMPLE_GREEDY)
    rc, rp, _ = O.body_pixel_infer(mf, ids, sd_a, sd_p, sd_b, sd_h, n_layers=10)
    assert poses.shape == (B, T, 258)
    np.testing.assert_array_equal(codes.cpu().numpy(), rc)
    np.testing.assert_allclose(poses.cpu().numpy(), rp, atol=1e-4, rtol=0)


def test_golden_clips_inside_baseline_batches(hip, golden, tmp_path):
    """The two reference-golden clips (body_e2e_full, B=2) embedded at arbitrary slots of a BASELINE batch of 32, of a
    coalesced chain of 128 clips (4 batches in one launch sequence, 64 x 32 split-K tiles), of the bench's 256-clip pass (the
    64 x 64 full-K wide kernel, csrc/skinny_wide.hip) and of a 320-clip pass (ragged row tiles, several tiles per
    workgroup): their codes must equal the golden bit for bit and their poses stay within 1e-4 — every operating point is
    pinned to the reference directly, not only through self-consistency."""
    from nets.init_model import init_model
    from talkshow_amd import _lib
    g = golden("body_e2e_full")
    args = argparse.Namespace(gpu=0, infer=True)
    w = init_model("s2g_body_pixel", args, _config(tmp_path))
    w.load_state_dict({"generator": synth.to_torch(synth.pixelcnn_state_dict(seed=7)),
                       "audioencoder": synth.to_torch(synth.audioencoder_state_dict(seed=7))})
    for B, slots in ((32, (3, 29)), (128, (70, 127)), (256, (64, 255)), (320, (77, 319))):
        mf, ids = synth.mfcc_features(90 + B, B, 300), synth.speaker_ids(B)
        for k, s_ in enumerate(slots):
            mf[s_], ids[s_] = g["mfcc"][k], g["ids"][k]
        codes, poses = w.generate_batch(mf, ids, mode=_lib.TS_SAMPLE_GREEDY)
        codes, poses = codes.cpu().numpy(), poses.cpu().numpy()
        for k, s_ in enumerate(slots):
            np.testing.assert_array_equal(codes[s_], g["codes"][k])
            np.testing.assert_allclose(poses[s_], g["poses"][k], atol=1e-4, rtol=0)
        if B >= 128:     # coalescing is invisible: each 32-clip batch alone gives the same bits
            c32, p32 = w.generate_batch(mf[64:96], ids[64:96], mode=_lib.TS_SAMPLE_GREEDY)
            np.testing.assert_array_equal(c32.cpu().numpy(), codes[64:96])
            np.testing.assert_array_equal(p32.cpu().numpy(), poses[64:96])


# ----------------------------------------------------------------------------------------------- face generator
def test_face_golden(hip, golden):
    """wav2vec2 encoder + LN conv heads vs the reference goldens (transformers module run by make_golden.py)."""
    from talkshow_amd.modules import FaceGenerator
    g = golden("face_full")
    m = FaceGenerator().cuda()
    m.load_state_dict(synth.to_torch(synth.face_state_dict(seed=7)))
    frame = g["out"].shape[1]
    out, hid = m.run(g["wav"], g["ids"], frame, want_hidden=True)
    np.testing.assert_allclose(hid.cpu().numpy(), g["hidden"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], atol=1e-4, rtol=0)      # 1e-4 on jaw / expression floats
    # reference call shape: (B,1,N) in, tuple out; legacy weight-norm key names load to the same result
    m2 = FaceGenerator().cuda()
    sd = synth.to_torch(synth.face_state_dict(seed=7, legacy_weight_norm_keys=True))
    m2.load_state_dict({"module." + k: v for k, v in sd.items()})
    out2, none = m2(torch.from_numpy(g["wav"])[:, None, :], None, torch.from_numpy(g["ids"]), time_steps=frame)
    assert none is None
    np.testing.assert_array_equal(out2.cpu().numpy(), out.cpu().numpy())


def test_face_10s_golden_inside_batch_64(hip, golden):
    """BASELINE configs[2] shape: two reference-golden 10 s clips (160 000 samples) embedded in a batch of 64; their
    rows must match the reference's output within 1e-4 whatever the rest of the batch holds."""
    from talkshow_amd.modules import FaceGenerator
    g = golden("face_10s")
    seed, B0, N = [int(v) for v in g["wav_seed"]]
    gw = synth.wav16(seed, B0, N)
    m = FaceGenerator().cuda()
    m.load_state_dict(synth.to_torch(synth.face_state_dict(seed=7)))
    B = 64
    wav = synth.wav16(500, B, N)
    ids = np.eye(4, dtype=np.float32)[np.arange(B) % 4]
    slots = (5, 63)
    for k, s_ in enumerate(slots):
        wav[s_], ids[s_] = gw[k], g["ids"][k]
    out = m.run(wav, ids, 300).cpu().numpy()
    assert out.shape == (B, 300, 103) and np.isfinite(out).all()
    for k, s_ in enumerate(slots):
        np.testing.assert_allclose(out[s_], g["out"][k], atol=1e-4, rtol=0)


def test_wrapper_face(hip, golden, tmp_path):
    from nets.init_model import init_model
    from oracle import face_oracle as FO
    g = golden("face_full")
    args = argparse.Namespace(gpu=0, infer=True)
    w = init_model("s2g_face", args, _config(tmp_path, "face"))
    w.load_state_dict({"generator": synth.to_torch(synth.face_state_dict(seed=7))})
    gen = w.generate(torch.from_numpy(g["wav"])[:, None, :], g["out"].shape[1])
    np.testing.assert_allclose(gen.cpu().numpy(), g["generate_zero_id"], atol=1e-4, rtol=0)
    # infer_on_audio on raw samples of one clip with a speaker id; vs the oracle (edge: odd sample count, ragged frames)
    wav = synth.wav16(77, 1, 21337)
    out = w.infer_on_audio(wav[0], id=torch.tensor([3]))
    frame = 21337 * 30 // 16000
    assert out.shape == (1, frame, 103)
    ref = FO.face_generator(wav, np.eye(4, dtype=np.float32)[[3]], synth.face_state_dict(seed=7), frame)
    np.testing.assert_allclose(out, ref, atol=1e-4, rtol=0)
    assert w.each_dim == [3, 72, 90, 100]


def test_wrapper_face_convert_to_6d_vs_reference(hip, golden):
    """`nets.s2g_face` with `convert_to_6d=true` builds `Generator(identity=False)` (`smplx_face.py:37-45`): no id channels, identity
    residual in the first LN-conv, a 6-wide jaw head -> (B, T, 106).  Against the reference wrapper's own output (face_6d golden)."""
    from nets.init_model import init_model
    from talkshow_amd.config import Object
    g = golden("face_6d")
    seed, B, N = (int(v) for v in g["wav_seed"])
    cfg = json.load(open(os.path.join(REPO, "config", "face.json")))
    cfg["Data"]["pose"]["convert_to_6d"] = True
    w = init_model("s2g_face", argparse.Namespace(gpu=0, infer=True), Object(cfg))
    assert w.generator.identity is False and w.each_dim[0] == 6
    w.load_state_dict({"generator": synth.to_torch(synth.face_state_dict(seed=8, identity=False, jaw_dim=6))})
    wav = torch.from_numpy(synth.wav16(seed, B, N))[:, None, :]
    out = w.infer_on_audio(wav, id=torch.tensor([1, 2]))
    assert out.shape == g["out"].shape == (B, N * 30 // 16000, 106)
    np.testing.assert_allclose(out, g["out"], atol=1e-4, rtol=0)
    gen = w.generate(wav, N * 30 // 16000)
    np.testing.assert_array_equal(gen.cpu().numpy(), out)


@pytest.mark.parametrize("products,tol_hidden,tol_out", [(6, 1e-4, 1e-4), (3, 1e-4, 1e-4)])
def test_face_split_bf16_plan_vs_reference_golden(hip, golden, products, tol_hidden, tol_out):
    """OPT-IN split-bf16 arithmetic (`ts_face_set_arith`, csrc/conv_gemm_split.hip) on the BASELINE-length reference goldens
    (two 10 s clips, hidden state and output produced by the reference wrapper over transformers): the same 1e-4 bar as the
    fp32 path, on both.  The fp32 plan stays the default and is untouched: switching back reproduces its bits."""
    from talkshow_amd.modules import FaceGenerator
    g, gf = golden("face_10s"), golden("face_full")
    seed, B, N = (int(v) for v in g["wav_seed"])
    wav = synth.wav16(seed, B, N)
    m = FaceGenerator().cuda()
    m.load_state_dict(synth.to_torch(synth.face_state_dict(seed=7)))
    base, base_hid = m.run(wav, g["ids"], 300, want_hidden=True)
    out, hid = m.set_arith(products).run(wav, g["ids"], 300, want_hidden=True)
    err_o, err_h = float(np.abs(out.cpu().numpy() - g["out"]).max()), float((hid - base_hid).abs().max())
    print(f"split-bf16 x{products}: out vs reference golden {err_o:.2e}, hidden vs fp32 path {err_h:.2e}")
    assert err_o <= tol_out
    assert err_h <= tol_hidden
    # the 2 s goldens carry the reference's hidden state itself
    frame = gf["out"].shape[1]
    o2, h2 = m.run(gf["wav"], gf["ids"], frame, want_hidden=True)
    np.testing.assert_allclose(h2.cpu().numpy(), gf["hidden"], atol=tol_hidden, rtol=0)
    np.testing.assert_allclose(o2.cpu().numpy(), gf["out"], atol=tol_out, rtol=0)
    again, _ = m.set_arith(0).run(wav, g["ids"], 300, want_hidden=True)
    assert torch.equal(again, base)


def test_face_split_bf16_plan_at_baseline_shapes(hip, golden):
    """The OPT-IN bf16x3 plan where configs[2] runs it: the two reference-golden 10 s clips inside a batch of 64 (output against the
    reference golden, hidden state against the pinned oracle's `wav2vec2_forward`: the 10 s golden holds no hidden state), and the
    60 s clip of `test_face_one_minute_clip_vs_oracle` (attention over 1 800 frames).  Same 1e-4 bar as the fp32 plan; the measured
    errors are recorded (TS_MEASURED_LOG)."""
    from oracle import face_oracle as FO
    from talkshow_amd.modules import FaceGenerator
    g = golden("face_10s")
    seed, B0, N = [int(v) for v in g["wav_seed"]]
    gw = synth.wav16(seed, B0, N)
    sd = synth.face_state_dict(seed=7)
    m = FaceGenerator().cuda()
    m.load_state_dict(synth.to_torch(sd))
    B, slots = 64, (5, 63)
    wav = synth.wav16(500, B, N)
    ids = np.eye(4, dtype=np.float32)[np.arange(B) % 4]
    for k, s_ in enumerate(slots):
        wav[s_], ids[s_] = gw[k], g["ids"][k]
    out, hid = m.set_arith(3).run(wav, ids, 300, want_hidden=True)
    out, hid = out.cpu().numpy(), hid.cpu().numpy()
    ref_h = FO.wav2vec2_forward(gw, sd, 300)
    for k, s_ in enumerate(slots):
        assert_close_measured(f"face_bf16x3.10s_in_b64.out[{k}]", out[s_], g["out"][k], 1e-4)
        assert_close_measured(f"face_bf16x3.10s_in_b64.hidden[{k}]", hid[s_], ref_h[k], 1e-4)
    sd5 = synth.face_state_dict(seed=5)
    m5 = FaceGenerator().cuda()
    m5.load_state_dict(synth.to_torch(sd5))
    N, frames = 16000 * 60, 30 * 60
    wav = synth.wav16(43 + 60, 1, N)
    ids = np.eye(4, dtype=np.float32)[[2]]
    out, hid = m5.set_arith(3).run(wav, ids, frames, want_hidden=True)
    assert_close_measured("face_bf16x3.60s.out", out.cpu().numpy(), FO.face_generator(wav, ids, sd5, frames), 1e-4)
    assert_close_measured("face_bf16x3.60s.hidden", hid.cpu().numpy(), FO.wav2vec2_forward(wav, sd5, frames), 1e-4)


def test_face_one_minute_clip_vs_oracle(hip):
    """A 60 s clip in ONE call (960 000 samples -> 1 800 frames: attention rows of 1 824 entries, beyond the 512 the softmax
    kernel keeps in registers) against the CPU oracle — the reference's `infer_on_audio` takes a wav of any length
    (`smplx_face.py:169-218`) — and through the wrapper; a 20 s clip (600 frames) rides along as the first size past the old cap."""
    from oracle import face_oracle as FO
    from talkshow_amd.modules import FaceGenerator
    sd = synth.face_state_dict(seed=5)
    m = FaceGenerator().cuda()
    m.load_state_dict(synth.to_torch(sd))
    for seconds in (20, 60):
        N, frames = 16000 * seconds, 30 * seconds
        wav = synth.wav16(43 + seconds, 1, N)
        ids = np.eye(4, dtype=np.float32)[[2]]
        out = m.run(wav, ids, frames).cpu().numpy()
        ref = FO.face_generator(wav, ids, sd, frames)
        assert out.shape == ref.shape == (1, frames, 103)
        np.testing.assert_allclose(out, ref, atol=1e-4, rtol=0)


def test_face_full_length_vs_oracle(hip):
    """A full 10 s clip (160 000 samples -> 499 conv frames -> 300 output frames) against the CPU oracle, plus batch
    independence and run-to-run determinism at B=3."""
    from oracle import face_oracle as FO
    from talkshow_amd.modules import FaceGenerator
    sd = synth.face_state_dict(seed=5)
    m = FaceGenerator().cuda()
    m.load_state_dict(synth.to_torch(sd))
    wav = synth.wav16(41, 3, 160000)
    ids = np.eye(4, dtype=np.float32)[[1, 0, 3]]
    out = m.run(wav, ids, 300)
    out2 = m.run(wav, ids, 300)
    assert torch.equal(out, out2) and torch.isfinite(out).all()
    solo = m.run(wav[1:2], ids[1:2], 300)
    assert torch.equal(solo[0], out[1])
    ref = FO.face_generator(wav[:1], ids[:1], sd, 300)
    np.testing.assert_allclose(out[:1].cpu().numpy(), ref, atol=1e-4, rtol=0)


# ----------------------------------------------------------------------------------------------- ragged / odd sizes
def test_ragged_lengths_and_batches(hip):
    """Edge cases the reference's shapes allow: clip lengths that are not multiples of 4 (odd length entering the second
    stride-2 conv), a single clip, and a batch that does not fill / overflows one 32-row MFMA tile — all vs the oracle."""
    from talkshow_amd import _lib
    from talkshow_amd.modules import AudioEncoder, GatedPixelCNN, VQVAE
    dims = dict(input_dim=128, dim=64, n_layers=3)
    sd_v = synth.vqvae_state_dict(seed=9, in_dim=39, num_embeddings=128, num_hiddens=128)
    vq = VQVAE(39, 64, 128, 128, 2).cuda(); vq.load_state_dict(synth.to_torch(sd_v))
    for B, T in [(1, 78), (3, 31), (2, 4)]:
        poses = synth.gt_poses(50 + T, B, T, dim=39)
        z, q, lat = vq.encode_nlc(poses, want_z=True)
        zr, er, ir = O.vqvae_encode(poses, sd_v)
        np.testing.assert_allclose(z.cpu().numpy().transpose(0, 2, 1), zr, atol=2e-5, rtol=0)
        np.testing.assert_array_equal(lat.cpu().numpy(), ir)
        rec = vq.decode_nlc(lat)
        assert rec.shape == (B, 4 * (T // 4), 39)
        np.testing.assert_allclose(rec.cpu().numpy().transpose(0, 2, 1), O.vqvae_decode(ir, sd_v), atol=1e-4, rtol=0)
    sd_a = synth.audioencoder_state_dict(seed=9)
    ae = AudioEncoder(64, 256, 2).cuda(); ae.load_state_dict(synth.to_torch(sd_a))
    mf = synth.mfcc_features(60, 2, 79)
    np.testing.assert_allclose(ae.forward_nlc(mf).cpu().numpy().transpose(0, 2, 1),
                               O.audio_encoder(np.ascontiguousarray(mf.transpose(0, 2, 1)), sd_a), atol=2e-5, rtol=0)
    sd_p = synth.pixelcnn_state_dict(seed=9, **dims)
    px = GatedPixelCNN(dims["input_dim"], dims["dim"], dims["n_layers"], 4, True, True).cuda()
    px.load_state_dict(synth.to_torch(sd_p))
    for B, H in [(1, 1), (33, 4), (5, 2)]:          # H = 1: no row above at all; B = 33: two MFMA row tiles
        rng = np.random.default_rng(B * 10 + H)
        aud = rng.standard_normal((B, H, 256)).astype(np.float32)
        label = synth.speaker_ids(B)
        codes, _ = px.run(label, aud, mode=_lib.TS_SAMPLE_GREEDY)
        ref = O.pixelcnn_generate(label, np.repeat(aud.transpose(0, 2, 1)[:, :, :, None], 2, axis=3), sd_p, dims["n_layers"], H)
        np.testing.assert_array_equal(codes.cpu().numpy(), ref)


def test_single_layer_pixelcnn(hip):
    """n_layers = 1: no audio fusion, no composed stage (the launch plan's degenerate branch)."""
    from talkshow_amd import _lib
    from talkshow_amd.modules import GatedPixelCNN
    sd = synth.pixelcnn_state_dict(seed=4, input_dim=64, dim=32, n_layers=1)
    px = GatedPixelCNN(64, 32, 1, 4, True, True).cuda(); px.load_state_dict(synth.to_torch(sd))
    rng = np.random.default_rng(1)
    aud = rng.standard_normal((2, 5, 256)).astype(np.float32)
    label = synth.speaker_ids(2)
    codes, logits = px.run(label, aud, mode=_lib.TS_SAMPLE_GREEDY, want_logits=True)
    ref, rl = O.pixelcnn_generate(label, np.repeat(aud.transpose(0, 2, 1)[:, :, :, None], 2, axis=3), sd, 1, 5, return_logits=True)
    assert_close_measured("single_layer.step_logits", logits.cpu().numpy(), rl, LOGIT_ATOL)
    np.testing.assert_array_equal(codes.cpu().numpy(), ref)


def test_error_behaviour(hip):
    """Inputs the path cannot process fail loudly through ts_last_error (no silent fallback, no crash)."""
    from talkshow_amd.modules import AudioEncoder, FaceGenerator, VQVAE
    ae = AudioEncoder(64, 256, 2).cuda()
    with pytest.raises(RuntimeError, match="too short"):
        ae.forward_nlc(np.zeros((1, 3, 64), np.float32))                       # fewer than 4 frames: no code row at all
    vq = VQVAE(39, 64, 128, 128, 2).cuda()
    with pytest.raises(RuntimeError, match="too short"):
        vq.encode_nlc(np.zeros((2, 2, 39), np.float32))
    with pytest.raises(RuntimeError, match="size mismatch|missing"):
        vq.load_state_dict({"encoder.project.conv.weight": torch.zeros(1)})
    fg = FaceGenerator(n_layers=1).cuda()
    fg.load_state_dict(synth.to_torch(synth.face_state_dict(seed=1, n_layers=1)))
    with pytest.raises(RuntimeError, match="too short"):
        fg.run(np.zeros((1, 300), np.float32), np.zeros((1, 4), np.float32), 1)   # below the 400-sample receptive field
    out = fg.run(synth.wav16(1, 1, 4000), np.zeros((1, 4), np.float32), 7)
    assert out.shape == (1, 7, 103) and torch.isfinite(out).all()


def test_device_mfcc_vs_host_restatement(hip, tmp_path):
    """ts_mfcc_* (polyphase resample + fused STFT kernel (real FFT in LDS) + mel + dB/top_db + DCT on the GPU) against the numpy restatement of
    the same torchaudio definitions (talkshow_amd/frontend.py; both unpinned against torchaudio itself)."""
    from scipy.io import wavfile
    from talkshow_amd import frontend as fe
    from talkshow_amd.modules import MFCC
    rng = np.random.default_rng(3)
    t = np.arange(160000) / 16000.0
    wav = (0.2 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.05 * rng.standard_normal(t.size)).astype(np.float32)
    wav2 = (0.1 * rng.standard_normal(t.size)).astype(np.float32)
    dev = MFCC(16000, 22000, 30)(np.stack([wav, wav2])).cpu().numpy()
    for i, w_ in enumerate((wav, wav2)):
        x22 = fe.resample_sinc_hann(w_[None], 16000, 22000)[0]
        ref = fe.mfcc(x22, 22000, hop_length=734).T
        assert dev[i].shape == ref.shape == (300 if len(x22) // 734 + 1 == 300 else len(x22) // 734 + 1, 64)
        # coefficients are O(10..200); the device's fp32 radix-4 FFT vs pocketfft differ by rounding only: the device rows sit within
        # 2e-4 of the float64 twin (test_wav_in_code_stability), the float32 twins within 2e-4 of it too; 2e-3 is the bound
        np.testing.assert_allclose(dev[i], ref, atol=2e-3, rtol=2e-4)
    # no resampling branch + the wav-file entry point used by infer_on_audio
    x = (0.3 * rng.standard_normal(22000 * 2)).astype(np.float32)
    got22 = MFCC(22000, 22000, 30)(x)[0].cpu().numpy()
    np.testing.assert_allclose(got22, fe.mfcc(x, 22000).T, atol=2e-3, rtol=2e-4)
    # ... and against the pipeline assembled from installed third-party code (torch.stft, transformers.audio_utils, scipy):
    # pins the device STFT-as-GEMM / mel / dB / DCT stages independently of this repo's own numpy twin
    from conftest import third_party_mfcc
    np.testing.assert_allclose(got22, third_party_mfcc(x), atol=2e-3, rtol=2e-4)
    x22 = fe.resample_sinc_hann(wav[None], 16000, 22000)[0]                  # resampler itself: unpinned, shared
    np.testing.assert_allclose(dev[0], third_party_mfcc(x22), atol=2e-3, rtol=2e-4)
    p = str(tmp_path / "a.wav")
    wavfile.write(p, 16000, np.stack([wav, wav2], 1))                       # stereo float wav
    a = fe.get_mfcc_ta(p, sr=22000, fps=30)                                    # device path
    b = fe.get_mfcc_ta(p, sr=22000, fps=30, host=True)                         # numpy path
    assert a.shape == b.shape and a.shape[1] == 64
    np.testing.assert_allclose(a, b, atol=2e-3, rtol=2e-4)


def test_wav_in_code_stability(hip, tmp_path):
    """device MFCC rows (fp32 FFT) vs the float64 host twin on the same resampled waveforms — the MFCC
    difference stays under 2e-3 (coefficients are O(10..200)) and NOT ONE of the 32 x 150 greedy codes changes, for noise clips
    (the bench's synthetic audio) and for speech-like clips (amplitude-modulated harmonics + noise, 50 dB of dynamic range)."""
    import bench
    _lib, lib, ctx = hip
    w, _ = bench.build_models(0)
    ids = torch.from_numpy(synth.speaker_ids(32)).cuda()
    rng = np.random.default_rng(12)
    t = np.arange(160000) / 16000.0
    speechy = []
    for k in range(32):
        f0 = 90.0 + 6.0 * k
        env = np.clip(np.sin(2 * np.pi * (2.0 + 0.1 * k) * t + k), 0, None) ** 2
        x = sum(np.sin(2 * np.pi * f0 * h * t + h) / h for h in range(1, 12)) * env * 0.08 + 0.003 * rng.standard_normal(t.size)
        speechy.append(x.astype(np.float32))
    for name, wav in (("noise", synth.wav16(7000, 32, 160000)), ("speech-like", np.stack(speechy))):
        s = bench.wav_in_stability(w, _lib, wav, ids)
        assert s["mfcc_max_abs_err_vs_float64"] <= 2e-3, (name, s)
        assert s["codes_differing"] == 0 and s["max_pose_delta"] == 0.0, (name, s)


@pytest.mark.parametrize("env", [{"TS_SKINNY_V": "0"}, {"TS_NO_GRAPH": "1"}, {"TS_SKINNY_NT": "32"},
                                 {"TS_SKINNY_SHAPE": "22"}, {"TS_SKINNY_SHAPE": "42"},
                                 {"TS_SKINNY_TILED": "0", "TS_WITH_CLIPS": "1"}, {"TS_PIX_DEFER_P": "0", "TS_WITH_CLIPS": "1"},
                                 {"TS_PIX_DEFER_P": "1", "TS_WITH_CLIPS": "1"}, {"TS_SKINNY_WIDE_MIN": "0", "TS_WITH_CLIPS": "1"},
                                 {"TS_SKINNY_WIDE_MIN": "1", "TS_SKINNY_WIDE_PAIR": "0", "TS_WITH_CLIPS": "1"}, {"TS_VQ_LDS": "0", "TS_CONV_RING": "0", "TS_WITH_VQ": "1"},
                                 {"TS_CONV_DEAL": "0", "TS_CONV_TAPS48": "0", "TS_W2V_MOMENTS": "0", "TS_WITH_VQ": "1"}, {"TS_CONV_RING_PAIRED": "0", "TS_WITH_VQ": "1", "TS_WITH_CLIPS": "1"},
                                 {"TS_CONV_SK": "0", "TS_WITH_FACE64": "1"}, {"TS_CONV_SK": "2", "TS_WITH_FACE64": "1"}],
                         ids=["generic_skinny_kernel", "eager_launches", "skinny_32col_kernel", "tile_32x32", "tile_64x32",
                              "row_major_operands", "projections_in_column0", "projections_in_column1",
                              "split_k_kernels_only", "wide_kernel_everywhere_column_major", "per_thread_vq_search_and_register_staged_conv",
                              "ring_conv_on_a_plain_grid_and_padded_positional_conv", "paired_layers_on_the_register_staged_banded_launch",
                              "face_gemms_without_the_stream_k_band", "face_gemms_with_the_stream_k_band_wherever_a_plan_exists"])
def test_alternate_kernel_paths(hip, env):
    """The PixelCNN chain has a fast descriptor-driven kernel + hipGraph replay and generic fallbacks (other shapes, eager
    launches, the 32-column kernel), row-major instead of tiled operands, two placements of the next-row projections, and
    the 64 x 64 wide kernel that coalesced passes use for launches of >= 160 workgroups (forced off / forced onto every
    launch of >= 64 clips here, small head / column-1 launches included); the codebook search has an LDS-staged and a per-thread
    form, conv_gemm_f32 a register-staged and an LDS-DMA engine, the latter with its tiles on a plain grid or dealt to the XCDs, the positional conv a padded and an unpadded kernel (the last entries force the older of each).  The knobs are read once per process, so the
    golden-vector tests are re-run in a child process with each path forced: all must stay bit-exact on the codes."""
    import subprocess
    import sys
    child_env = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q",
                        "-k", "pixelcnn_golden or pixelcnn_sampling_and_prefix or single_layer_pixelcnn or op_linear"
                        + (" or golden_clips" if env.get("TS_WITH_CLIPS") else "")     # BASELINE-size batches: the coalesced tile shapes
                        + (" or op_vq_argmin or vqvae_golden or wrapper_body_vq or face_golden" if env.get("TS_WITH_VQ") else "")
                        # (the bitwise batch-independence check of face_full_length_vs_oracle holds with the band off and under the cost model at
                        # those sizes, not with the band forced onto every layer that has a plan: which tiles are split depends on M)
                        + (" or face_golden or face_10s_golden_inside_batch_64" + (" or face_full_length_vs_oracle" if env.get("TS_CONV_SK") != "2" else "")
                           if env.get("TS_WITH_FACE64") else "")],
                       env=child_env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_assemble_full(hip, golden):
    """Device output assembly (ts_assemble_full) vs the reference's demo.py concat + part2full: pure copies, bit-exact."""
    from talkshow_amd.pose_index import assemble_full
    g = golden("assemble_full")
    for tag in ("longer_face", "shorter_face"):
        for stand, key in ((False, "full_"), (True, "full_stand_")):
            out = assemble_full(g["body"], g["face_" + tag], stand=stand).cpu().numpy()
            assert np.array_equal(out, g[key + tag]), (tag, stand)
    with pytest.raises(ValueError):
        assemble_full(np.zeros((2, 4, 128), np.float32), np.zeros((2, 4, 103), np.float32))
    # BASELINE configs[4] shape: (32, 300, 129) + (32, 300, 103) -> (32, 300, 265), against the oracle
    rng = np.random.default_rng(5)
    body, face = rng.standard_normal((32, 300, 129)).astype(np.float32), rng.standard_normal((32, 300, 103)).astype(np.float32)
    from talkshow_amd.pose_index import lower_pose_block
    assert np.array_equal(assemble_full(body, face).cpu().numpy(), O.assemble_full(body, face, lower_pose_block(False)))


def test_whole_body_sharded_single_rank(hip, tmp_path):
    """configs[4] driver at world size 1: body batches + face batches + device assembly == the pieces run one by one."""
    import types
    import bench
    from talkshow_amd import parallel
    from talkshow_amd.modules import FaceGenerator
    from talkshow_amd.pose_index import lower_pose_block
    _lib, lib, ctx = hip
    w, _ = bench.build_models(0)
    m = FaceGenerator().cuda()
    m.load_state_dict(synth.to_torch(synth.face_state_dict(seed=0)))
    N, T, S = 5, 60, 32000                                   # 5 clips of 2 s: 60 body frames, 60 face frames
    mfcc = dev(synth.mfcc_features(70, N, T))
    ids = torch.from_numpy(synth.speaker_ids(N)).cuda()
    wav = dev(synth.wav16(71, N, S))
    fid = torch.nn.functional.one_hot(torch.arange(N) % 4, 4).float().cuda()
    rows, (a, b) = parallel.whole_body_sharded(w, types.SimpleNamespace(generator=m), mfcc, ids, wav, fid,
                                               mode=_lib.TS_SAMPLE_GREEDY, batch_body=2, batch_face=3)
    assert (a, b) == (0, N) and rows.shape == (N, 60, 265)
    poses = w.generate_batch(mfcc, ids, mode=_lib.TS_SAMPLE_GREEDY)[1].cpu().numpy()
    face = m.run(wav, fid, 60).cpu().numpy()
    ref = O.assemble_full(poses, face, lower_pose_block(False))
    np.testing.assert_allclose(rows.cpu().numpy(), ref, atol=1e-5)


# ----------------------------------------------------------------------------------------------- evaluation on the device
def test_evaluation_reductions_vs_reference_values(hip, golden):
    """ts_eval_* (csrc/eval.hip) through the drop-in `evaluation` package against values the reference's own
    evaluation/FGD.py, evaluation/metrics.py and test_body.body_loss produced (golden `eval_metrics`)."""
    from evaluation.FGD import EmbeddingSpaceEvaluator
    from evaluation import metrics as M
    from talkshow_amd import evaluation as E
    g = golden("eval_metrics")

    class StubAE:                                               # the golden was made on ready-made feature rows
        def extract(self, x):
            return x.cuda(), x
    ev = EmbeddingSpaceEvaluator(StubAE(), None, "cuda")
    at_r = at_g = 0
    for H in g["clip_rows"]:
        H = int(H)
        r = g["real"][at_r:at_r + H * 64].reshape(1, H, 64); at_r += H * 64
        q = g["gen"][at_g:at_g + 2 * H * 64].reshape(2, H, 64); at_g += 2 * H * 64
        ev.push_samples(torch.from_numpy(q), torch.from_numpy(r))
    fgd, feat_dist = ev.get_scores()
    np.testing.assert_allclose(fgd, g["fgd"], rtol=1e-5)        # reference: float32 np.mean, float64 np.cov
    np.testing.assert_allclose(feat_dist, g["feat_dist"], rtol=1e-6)
    bl = E.body_loss(g["gt_joints"], g["pr_joints"])
    for k, ref in (("LVD", "lvd"), ("error", "error"), ("diverse", "diverse")):
        np.testing.assert_allclose(bl[k], g[ref], rtol=2e-5)    # reference: float32 torch arithmetic
    lv = M.LVD(torch.from_numpy(g["gt_joints"]), torch.from_numpy(g["pr_joints"][0]))
    assert torch.is_tensor(lv) and lv.dim() == 0                      # the reference returns a tensor: its callers add them up and call .item()
    np.testing.assert_allclose(lv.item(), g["lvd_single"], rtol=2e-5)
    gs = golden("lvd_symmetric")                                      # the reference's symmetrical=True value (its `~mask.long()` included)
    for key, pr in (("lvd_sym", gs["pr_joints"]), ("lvd_sym_long", gs["pr_long"])):
        np.testing.assert_allclose(M.LVD(torch.from_numpy(gs["gt_joints"]), torch.from_numpy(pr), symmetrical=True).item(), gs[key], rtol=2e-5)
    # one sample (3-D pr): the reference ignores `symmetrical` there and needs equal lengths (`metrics.py:80-94`)
    one = torch.from_numpy(gs["pr_joints"][0])
    assert M.LVD(torch.from_numpy(gs["gt_joints"]), one, symmetrical=True).item() == M.LVD(torch.from_numpy(gs["gt_joints"]), one).item()
    with pytest.raises(RuntimeError):
        M.LVD(torch.from_numpy(gs["gt_joints"]), torch.from_numpy(gs["pr_long"][0]))
    np.testing.assert_allclose(M.LVD(torch.from_numpy(gs["gt_joints"]), torch.from_numpy(gs["pr_joints"])).item(), gs["lvd_plain"], rtol=2e-5)
    with pytest.raises(NotImplementedError):
        M.LVD(torch.from_numpy(gs["gt_joints"]), torch.from_numpy(gs["pr_joints"]), weight=True)
    np.testing.assert_allclose(M.diversity(g["kps"]), g["diversity"], rtol=1e-5)
    # run-to-run determinism of the two-stage reductions (no float atomics)
    assert E.body_loss(g["gt_joints"], g["pr_joints"]) == bl and M.diversity(g["kps"]) == M.diversity(g["kps"])


def test_feature_stats_large_and_ragged(hip):
    """Running float64 moments over many pushes of ragged sizes == numpy float64 mean / cov of all rows together."""
    from oracle import eval_oracle as EO
    from talkshow_amd import evaluation as E
    rng = np.random.default_rng(8)
    st, rows = E.FeatureStats(64), []
    for n in (1, 127, 128, 129, 5000, 33):
        x = (rng.standard_normal((n, 64)) * rng.uniform(0.1, 3.0, 64) + rng.standard_normal(64)).astype(np.float32)
        st.push(x)
        rows.append(x)
    allr = np.vstack(rows).astype(np.float64)
    mu, sig = st.mean_cov()
    np.testing.assert_allclose(mu, allr.mean(0), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sig, np.cov(allr, rowvar=False), rtol=1e-8, atol=1e-10)
    with pytest.raises(RuntimeError, match="32, 64 or 128"):
        E.FeatureStats(48).push(np.zeros((4, 48), np.float32))


@pytest.mark.parametrize("sr_in", [16000, 24000, 44100, 8000])
def test_device_sinc_hann_resampler_vs_twin(hip, sr_in):
    """`ts_mfcc_resample` (torchaudio's sinc-Hann polyphase FIR; `data_utils/utils.py:150-152`) against the numpy twin for the sample
    rates of the reference's demo wavs (16 k, 24 k -> the LDS-staged kernel), 44.1 k (220 x 467-tap table: the plain kernel) and an
    up-sampling ratio; two clips of different content, a length that is not a multiple of the block."""
    from talkshow_amd import frontend as fe
    from talkshow_amd.modules import MFCC
    rng = np.random.default_rng(sr_in)
    n = sr_in * 2 + 37
    x = (0.3 * rng.standard_normal((2, n))).astype(np.float32)
    got = MFCC(sr_in, 22000, 30).resample(x).cpu().numpy()
    ref = fe.resample_sinc_hann(x, sr_in, 22000)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=2e-6, rtol=0)


def test_device_kaiser_resampler_and_sepa(hip, tmp_path):
    """ts_resample_kaiser (face path: librosa.load(sr=16000)'s resampler) and the device get_mfcc_sepa against their numpy
    twins in talkshow_amd/frontend.py (both restate third-party definitions; the resamplers are unpinned against
    librosa / torchaudio themselves)."""
    from scipy.io import wavfile
    from talkshow_amd import frontend as fe
    from talkshow_amd.modules import resample_kaiser_device
    rng = np.random.default_rng(12)
    for sr_in, n in ((22050, 22050 + 7), (8000, 4000), (44100, 30000), (48000, 48000)):
        x = (0.3 * rng.standard_normal((2, n))).astype(np.float32)
        got = resample_kaiser_device(x, sr_in, 16000).cpu().numpy()
        for b in range(2):
            ref = fe.resample_kaiser_best(x[b], sr_in, 16000)
            assert got[b].shape == ref.shape
            np.testing.assert_allclose(got[b], ref, atol=2e-6, rtol=0)
    # wav file at 22.05 kHz through the face front-end: device path == host path
    p = str(tmp_path / "f.wav")
    wavfile.write(p, 22050, (0.2 * rng.standard_normal((22050, 2))).astype(np.float32))
    np.testing.assert_allclose(fe.get_wav16(p), fe.get_wav16(p, host=True), atol=2e-6, rtol=0)
    # get_mfcc_sepa (continuity front-end): device == host restatement, same split frame
    p2 = str(tmp_path / "b.wav")
    t = np.arange(16000 * 5) / 16000.0
    wavfile.write(p2, 16000, (0.2 * np.sin(2 * np.pi * 200 * t) + 0.05 * rng.standard_normal(t.size)).astype(np.float32))
    fd, gd = fe.get_mfcc_sepa(p2, sr=22000, fps=30)
    fh, gh = fe.get_mfcc_sepa(p2, sr=22000, fps=30, host=True)
    assert gd == gh == 1 + 44000 // 734 and fd.shape == fh.shape
    np.testing.assert_allclose(fd, fh, atol=0.05, rtol=2e-4)


# ----------------------------------------------------------------------------------------------- SMPL-X joints / vertices
@pytest.mark.parametrize("V,with_vertices", [(700, True), (10475, False)])
def test_smplx_layer_vs_oracle(hip, V, with_vertices):
    """Batched SMPL-X forward on the device (csrc/smplx.*) vs the float64 numpy restatement of smplx's published LBS
    (oracle/smplx_oracle.py) on synthetic model parameters of the real model's structure; V = 10475 is the real mesh size."""
    from oracle import smplx_oracle as SO
    from talkshow_amd.smplx_lbs import SMPLXLayer
    model = SO.synthetic_model(seed=3, V=V)
    layer = SMPLXLayer(model, with_vertices=with_vertices)
    assert layer.num_joints == 55 + 21 + 51
    rng = np.random.default_rng(V)
    B, T = 2, 9
    rows = (rng.standard_normal((B, T, 265)) * 0.35).astype(np.float32)
    rows[0, 0, 3:9] = 0.0                                      # zero eye poses: Rodrigues of the zero vector
    rows[..., 165:] *= 3.0                                     # expression coefficients are O(1)
    betas = (rng.standard_normal(300) * 0.8).astype(np.float32)
    ref_j, ref_v = SO.smplx_forward(model, betas, rows.reshape(-1, 265))
    if with_vertices:
        joints, verts = layer.vertices(betas, rows)
        assert verts.shape == (B, T, V, 3)
        np.testing.assert_allclose(verts.cpu().numpy().reshape(-1, V, 3), ref_v, atol=1e-4, rtol=0)
    else:
        joints = layer.joints(betas, rows)
        with pytest.raises(RuntimeError, match="with_vertices"):
            layer.vertices(betas, rows)
    assert joints.shape == (B, T, 127, 3)
    np.testing.assert_allclose(joints.cpu().numpy().reshape(-1, 127, 3), ref_j, atol=1e-4, rtol=0)     # metres
    # one betas row per pose row + run-to-run determinism
    bb = np.repeat(betas[None], B * T, 0)
    j2 = layer.joints(bb, rows)
    assert torch.equal(j2, joints)
