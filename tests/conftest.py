import os
import sys

# The CPU suite is BLAS-heavy (the oracle's full-grid PixelCNN recompute).  On an 8-core box OpenBLAS and torch's OpenMP pool default
# to 8 threads EACH; 4 run the suite in the same wall time at half the CPU time (measured: 32 s / 1 m 51 s of CPU against 32 s / 3 m 26 s for
# the two heaviest tests) and keep it from degrading by an order of magnitude when the host's cores are contended.  Override by
# exporting the variables.
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "4")

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    try:   # numpy may have been imported (by a plugin) before the variables above were set
        import threadpoolctl
        threadpoolctl.threadpool_limits(int(os.environ["OPENBLAS_NUM_THREADS"]))
    except Exception:
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    config.addinivalue_line("markers", "slow: tens of seconds on CPU")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


def assert_close_measured(name, got, ref, atol):
    """assert_allclose(atol, rtol=0) that also records what it measured: prints `name: max |err|` (visible with -s or on failure)
    and, when TS_MEASURED_LOG names a file, appends one JSON line to it — the bounds written in the tests are 2x those records."""
    got, ref = np.asarray(got), np.asarray(ref)
    err = float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()) if got.size else 0.0
    print(f"\n[measured] {name}: max |err| = {err:.3e} (bound {atol:.1e})")
    log = os.environ.get("TS_MEASURED_LOG")
    if log:
        import json
        with open(log, "a") as f:
            f.write(json.dumps({"name": name, "max_abs_err": err, "bound": atol}) + "\n")
    assert got.shape == ref.shape, f"{name}: shape {got.shape} vs {ref.shape}"
    assert err <= atol, f"{name}: max |err| {err:.3e} > {atol:.1e}"


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def third_party_mfcc(x, sample_rate=22000, n_fft=2048, hop=734, n_mels=256, n_mfcc=64):
    """MFCC assembled from INSTALLED third-party implementations of torchaudio's definitions (torchaudio itself is absent):
    torch.stft (periodic Hann, center, reflect, power 2) -> transformers.audio_utils.mel_filter_bank (htk, norm None) ->
    power_to_db(db_range=80, per clip) -> scipy DCT-II ortho.  x (N,) float32 -> (T, n_mfcc) float64."""
    import torch
    from scipy.fft import dct
    from transformers import audio_utils as au
    spec = torch.stft(torch.from_numpy(np.asarray(x, np.float32)), n_fft=n_fft, hop_length=hop, win_length=n_fft,
                      window=torch.hann_window(n_fft, periodic=True), center=True, pad_mode="reflect", normalized=False,
                      onesided=True, return_complex=True)
    power = spec.abs().pow(2.0).T.numpy().astype(np.float64)
    mel = power @ au.mel_filter_bank(n_fft // 2 + 1, n_mels, 0.0, float(sample_rate // 2), sample_rate, None, "htk")
    db = au.power_to_db(mel, reference=1.0, min_value=1e-10, db_range=80.0)
    return dct(db, type=2, norm="ortho", axis=1)[:, :n_mfcc]
