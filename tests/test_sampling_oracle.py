"""CPU pins of the stochastic-decode oracle (BASELINE configs[3]): the Philox4x32-10 restatement against Random123's published
known-answer vectors, the fp32-only exponential the sampler and the oracle share against float64 exp, and the inverse-CDF draw
against a direct float64 CDF."""
import numpy as np

from oracle import talkshow_oracle as O


def test_philox4x32_10_known_answers():
    """Random123 `kat_vectors`, philox4x32 with 10 rounds: (counter, key) -> output."""
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, want in kat:
        assert O.philox4x32_10(ctr, key) == want, (ctr, key)
    # the sampler's uniform is word 0 of counter (position, clip_lo, clip_hi, 0) under key (seed_lo, seed_hi), top 24 bits
    seed, clip, pos = 0x299f31d0a4093822, 0x13198a2e85a308d3, 0x243f6a88
    w0 = O.philox4x32_10([pos, clip & 0xffffffff, clip >> 32, 0], [seed & 0xffffffff, seed >> 32])[0]
    assert O.philox_uniform(seed, clip, pos) == np.float32((w0 >> 8) / 16777216.0)
    u = O.philox_uniforms(7, 3, 2, 5)
    assert u.shape == (2, 5, 2) and u[1, 4, 1] == O.philox_uniform(7, 4, 9) and (u >= 0).all() and (u < 1).all()


def test_det_expf_is_an_accurate_exp():
    """`det_expf` (fp32 multiplies / adds only; the HIP sampler's `det_expf` is the same sequence of operations) stays within
    2 ulp of exp on the sampler's domain (l - max <= 0), is exactly 1 at 0, monotone at the cut-off and 0 below -86."""
    rng = np.random.default_rng(0)
    x = -np.abs(rng.standard_normal(400000) * 20).astype(np.float32)
    x[:4] = [0.0, -86.0, np.nextafter(np.float32(-86.0), np.float32(-100)), -1e-8]
    e = O.det_expf(x)
    assert e.dtype == np.float32 and e[0] == 1.0 and e[1] > 0 and e[2] == 0.0 and e[3] == 1.0
    keep = x >= -86.0
    ref = np.exp(x[keep].astype(np.float64))
    assert (np.abs(e[keep] - ref) / ref).max() < 2 * 2.0 ** -24
    assert (e[~keep] == 0).all()


def test_inverse_cdf_draw_matches_a_float64_cdf():
    """`sample_inverse_cdf` (the kernel's chunked fp32 summation) picks the index a float64 CDF picks, except where u * total
    falls within fp32 summation noise of a boundary; u = 0 -> first index with mass, u -> 1 -> last index with mass."""
    rng = np.random.default_rng(1)
    B, V = 64, 2048
    logits = (rng.standard_normal((B, V)) * 3).astype(np.float32)
    u = rng.random(B).astype(np.float32)
    u[0], u[1] = 0.0, np.float32(1.0 - 2 ** -24)
    got = O.sample_inverse_cdf(logits, u)
    p = np.exp(logits.astype(np.float64) - logits.max(1, keepdims=True))
    cdf = np.cumsum(p, 1)
    want = np.array([np.searchsorted(cdf[b], np.float64(u[b]) * cdf[b, -1], side="right") for b in range(B)])
    near = np.abs(got - want) <= 1
    assert near.all() and (got != want).sum() <= 2
    assert got[0] == 0 and got[1] >= V - 8
    # ragged vocabulary (V not a multiple of the 256 chunks) and a degenerate row (one logit carries all the mass)
    lg = np.full((1, 300), -200.0, np.float32)
    lg[0, 123] = 5.0
    assert O.sample_inverse_cdf(lg, np.asarray([0.7], np.float32))[0] == 123
