"""Audio front-end boundary (`data_utils/utils.py:148-231`, `get_mfcc_ta` / `get_mfcc_sepa`).

The reference computes 64-d MFCCs with torchaudio on the CPU before the hot path starts ("next"
row f1).  Until the on-device front-end lands, `aud_fn` may be

  * a `(T, 64)` float array / tensor of MFCC features (what `get_mfcc_ta` returns), or
  * a path to a `.npy` file holding such an array;

a `.wav` path raises with this explanation (torchaudio / librosa are not available in the target image).
"""
import os

import numpy as np
import torch


def get_mfcc_ta(aud_fn, sr=22000, fps=30, smlpx=True, type='mfcc', am=None, am_sr=None, encoder_choice='mfcc'):
    if isinstance(aud_fn, torch.Tensor):
        feat = aud_fn.detach().cpu().numpy()
    elif isinstance(aud_fn, np.ndarray):
        feat = aud_fn
    elif isinstance(aud_fn, (str, os.PathLike)) and str(aud_fn).endswith(".npy"):
        feat = np.load(aud_fn)
    else:
        raise NotImplementedError(
            f"audio front-end: cannot turn {aud_fn!r} into MFCC features here. The wav -> 22 kHz -> MFCC(64) front-end "
            "(torchaudio in the reference, data_utils/utils.py:148-231) is the next scope row (SURVEY.md §8f-1); pass the "
            "(T, 64) MFCC array, or a .npy file of it, as `aud_fn`.")
    feat = np.asarray(feat, dtype=np.float32)
    if feat.ndim != 2 or feat.shape[1] != 64:
        raise ValueError(f"MFCC features must have shape (T, 64), got {feat.shape}")
    return feat


def get_mfcc_sepa(aud_fn, sr=22000, fps=30):
    """`data_utils/utils.py:234-263`: features plus the frame index of the 2 s split used by continuity mode."""
    feat = get_mfcc_ta(aud_fn, sr=sr, fps=fps)
    gap = 2 * fps
    return feat, gap


def get_wav16(aud_fn):
    """Face front-end: `get_mfcc_ta(..., encoder_choice='faceformer')` = `librosa.load(aud_fn, sr=16000)` reshaped to
    (N, 1), no normalisation (`data_utils/utils.py:194-198`).  Accepted: arrays / tensors of samples, `.npy`, and PCM or
    float `.wav` files that are ALREADY at 16 kHz (mono = mean of channels, int PCM scaled to [-1, 1) as librosa does);
    other sample rates need the resampler of the next scope row and raise."""
    if isinstance(aud_fn, torch.Tensor):
        x = aud_fn.detach().cpu().numpy()
    elif isinstance(aud_fn, np.ndarray):
        x = aud_fn
    elif str(aud_fn).endswith(".npy"):
        x = np.load(aud_fn)
    elif str(aud_fn).endswith(".wav"):
        from scipy.io import wavfile
        sr, x = wavfile.read(aud_fn)
        if sr != 16000:
            raise NotImplementedError(f"{aud_fn}: sample rate {sr} != 16000; resampling is the next scope row (SURVEY.md §8f-1)")
        if np.issubdtype(x.dtype, np.integer):
            x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
        if x.ndim == 2:
            x = x.mean(axis=1)
    else:
        raise NotImplementedError(f"audio front-end: cannot read {aud_fn!r}")
    return np.asarray(x, dtype=np.float32).reshape(-1, 1)
