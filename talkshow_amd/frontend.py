"""Audio front-end boundary (`data_utils/utils.py:148-263`: `get_mfcc_ta` / `get_mfcc_sepa`).

In the reference this runs on the CPU before the hot path starts: `torchaudio.load` -> `Resample(sr_0 -> 22 kHz)` ->
mono -> `torchaudio.transforms.MFCC(n_mfcc=64, n_fft=2048, n_mels=256, hop=734|1467, mel_scale='htk')` for the body, and
`librosa.load(sr=16000)` raw samples for the face.  torchaudio / librosa are third-party code that is NOT under
/root/reference and not installed in the target image, so this module restates their published algorithms in
numpy/scipy on the host (same place in the pipeline as the reference's CPU front-end; it is not part of the HIP hot
path and no kernel parity claim depends on it):

  * Resample: torchaudio's `sinc_interp_hann` polyphase kernel (lowpass_filter_width=6, rolloff=0.99, ratio reduced
    by gcd), applied per channel BEFORE the mono mean (`utils.py:150-154`);
  * MFCC: periodic Hann window of n_fft, `center=True` reflect padding, power spectrum, HTK mel filterbank
    (f_min 0, f_max sr/2, norm None), `10*log10(clamp(x, 1e-10))`, clamp at (per-clip max - 80 dB), orthonormal
    DCT-II (256 -> 64).

Parity: torchaudio / librosa cannot be run here and the reference ships no vectors, so the stages are
pinned against the installed third-party implementations of the same definitions (tests/test_frontend.py): STFT / power
<-> `torch.stft` (what torchaudio's Spectrogram calls); HTK mel filterbank and dB / top_db <-> `transformers.audio_utils`;
DCT-II <-> `scipy.fft.dct`; whole MFCC <-> the pipeline assembled from those.  STILL UNPINNED: the sinc-Hann resampler
(and librosa's kaiser resampler of the face path), checked on closed-form properties only.  Everything downstream
(the hot path) is pinned on feature arrays, which is also what `aud_fn` may be directly: a `(T, 64)` float array /
tensor, or a `.npy` file of one.
"""
import math
import os

import numpy as np
import torch

F32 = np.float32


def load_wav(path):
    """`torchaudio.load`: float32 in [-1, 1), shape (channels, N), native sample rate (PCM / float .wav files)."""
    from scipy.io import wavfile
    sr, x = wavfile.read(path)
    if np.issubdtype(x.dtype, np.integer):
        if x.dtype == np.uint8:
            x = (x.astype(F32) - 128.0) / 128.0
        else:
            x = x.astype(F32) / float(np.iinfo(x.dtype).max + 1)
    x = np.asarray(x, dtype=F32)
    if x.ndim == 1:
        x = x[None, :]
    else:
        x = np.ascontiguousarray(x.T)
    return x, int(sr)


def resample_sinc_hann(x, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.transforms.Resample (default `sinc_interp_hann`): x (C, N) -> (C, ceil(new*N/orig)) float32."""
    if orig_freq == new_freq:
        return x
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base_freq))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base_freq
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kern = np.where(t == 0, 1.0, np.sin(t) / np.where(t == 0, 1.0, t)) * window * scale        # (new, 2*width+orig)
    kern = kern.astype(F32)
    C, N = x.shape
    xp = np.pad(x, ((0, 0), (width, width + orig)))
    nwin = (xp.shape[1] - kern.shape[1]) // orig + 1
    win = np.lib.stride_tricks.sliding_window_view(xp, kern.shape[1], axis=1)[:, ::orig][:, :nwin]   # (C, nwin, K)
    y = np.einsum("cwk,nk->cwn", win, kern, optimize=True).reshape(C, -1)                          # phases interleave
    target = int(math.ceil(new * N / orig))
    return np.ascontiguousarray(y[:, :target], dtype=F32)


_KAISER_BEST = {}


def kaiser_best_table():
    """resampy's published 'kaiser_best' interpolation filter (the resampler behind `librosa.load(sr=...)` in the reference's
    pinned librosa 0.9.2): right half of a sinc with 64 zero crossings sampled 512 times per crossing, rolloff
    0.9475937167399596, tapered by a Kaiser window with beta 14.769656459379492.  -> (half_window, deltas, num_table)."""
    if not _KAISER_BEST:
        from scipy.signal.windows import kaiser
        num_zeros, num_table, rolloff, beta = 64, 512, 0.9475937167399596, 14.769656459379492
        n = num_table * num_zeros
        sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
        win = (kaiser(2 * n + 1, beta)[n:] * sinc_win).astype(F32)
        delta = np.zeros_like(win)
        delta[:-1] = np.diff(win)
        _KAISER_BEST["t"] = (win, delta, num_table)
    return _KAISER_BEST["t"]


def resample_kaiser_best(x, orig_sr, target_sr):
    """`librosa.resample(x, orig_sr, target_sr, res_type='kaiser_best', fix=True)` on a mono signal (N,) -> float32
    (ceil(N * target / orig),): resampy's band-limited interpolation (table lookup + linear interpolation between table
    entries, left and right wings), then librosa's fix_length.  PARITY UNPINNED (librosa / resampy are not installed)."""
    x = np.asarray(x, dtype=F32)
    if orig_sr == target_sr:
        return x
    win, delta, num_table = kaiser_best_table()
    ratio = float(target_sr) / float(orig_sr)
    n_out = int(x.shape[0] * ratio)
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    nwin, n_orig = win.shape[0], x.shape[0]
    t_reg = np.arange(n_out, dtype=np.float64) / ratio
    n = t_reg.astype(np.int64)
    y = np.zeros(n_out, np.float64)
    xd, wd, dd = x.astype(np.float64), win.astype(np.float64), delta.astype(np.float64)
    if ratio < 1:                       # resampy: the filter is scaled by the ratio when decimating (unit DC gain)
        wd, dd = wd * ratio, dd * ratio
    for wing in (0, 1):
        frac = scale * (t_reg - n)
        if wing:
            frac = scale - frac
        index_frac = frac * num_table
        offset = index_frac.astype(np.int64)
        eta = index_frac - offset
        taps = (nwin - offset) // index_step
        limit = np.minimum(n + 1, taps) if wing == 0 else np.minimum(n_orig - n - 1, taps)
        for i in range(int(limit.max()) if limit.size else 0):
            live = i < limit
            k = np.where(live, offset + i * index_step, 0)
            src = np.where(live, n - i if wing == 0 else n + i + 1, 0)
            y += np.where(live, (wd[k] + eta * dd[k]) * xd[src], 0.0)
    target = int(math.ceil(x.shape[0] * ratio))
    out = np.zeros(target, F32)
    out[:min(target, n_out)] = y[:target].astype(F32)
    return out


def _hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + f / 700.0)


def _mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (m / 2595.0) - 1.0)


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk'): (n_freqs, n_mels) triangular filters."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_pts = np.linspace(_hz_to_mel_htk(f_min), _hz_to_mel_htk(f_max), n_mels + 2)
    f_pts = _mel_to_hz_htk(m_pts)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(F32)


def create_dct(n_mfcc, n_mels):
    """torchaudio.functional.create_dct(norm='ortho'): (n_mels, n_mfcc) DCT-II matrix."""
    n = np.arange(n_mels, dtype=np.float64)
    k = np.arange(n_mfcc, dtype=np.float64)[:, None]
    dct = np.cos(math.pi / n_mels * (n + 0.5) * k)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / n_mels)
    return dct.T.astype(F32)


def power_spectrogram(wave, n_fft=2048, hop_length=734, win_length=None):
    """torchaudio.transforms.Spectrogram(n_fft, win_length, hop_length, power=2) == |torch.stft(..., window=hann_window(win_length,
    periodic=True), center=True, pad_mode='reflect', onesided=True)|^2 on a mono waveform (N,) -> (T, n_fft//2+1),
    T = N // hop + 1; a window shorter than n_fft sits centred in the frame (torch.stft pads it with zeros on both sides).
    This stage IS pinned: tests/test_frontend.py compares it with torch.stft (installed), the function torchaudio's Spectrogram calls."""
    from scipy import fft as sfft
    x = np.pad(np.asarray(wave, dtype=F32), (n_fft // 2, n_fft // 2), mode="reflect")
    T = 1 + (x.shape[0] - n_fft) // hop_length
    frames = np.lib.stride_tricks.sliding_window_view(x, n_fft)[::hop_length][:T]
    wl = n_fft if win_length is None else int(win_length)
    window = np.zeros(n_fft, F32)
    left = (n_fft - wl) // 2
    window[left:left + wl] = (0.5 - 0.5 * np.cos(2.0 * math.pi * np.arange(wl) / wl)).astype(F32)   # periodic Hann
    spec = sfft.rfft(frames * window[None, :], axis=1)                                             # float32 -> complex64
    return (spec.real.astype(F32) ** 2 + spec.imag.astype(F32) ** 2).astype(F32)                  # (T, 1025)


def mfcc_from_power(power, sample_rate, n_mfcc=64, n_fft=2048, n_mels=256, top_db=80.0):
    """MelScale(htk, norm None) -> AmplitudeToDB('power', top_db=80, per-clip max) -> DCT-II ortho: (T, n_freq) -> (n_mfcc, T)."""
    mel = power @ melscale_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate)   # (T, n_mels)
    db = (10.0 * np.log10(np.maximum(mel, F32(1e-10)))).astype(F32)                                # ref = 1 -> no offset
    db = np.maximum(db, db.max() - F32(top_db))                                                    # per-clip max
    return np.ascontiguousarray((db @ create_dct(n_mfcc, n_mels)).T, dtype=F32)


def mel_spectrogram(wave, sample_rate, n_fft=2048, hop_length=734, n_mels=256, win_length=None):
    """torchaudio.transforms.MelSpectrogram(sample_rate, n_fft, win_length, hop_length, n_mels) with its defaults (f_min 0, f_max
    sample_rate // 2, power 2, HTK scale, no filter normalisation) on a mono waveform (N,) -> (n_mels, T)  [`utils.py:178-191`]."""
    power = power_spectrogram(wave, n_fft, hop_length, win_length)
    return np.ascontiguousarray((power @ melscale_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate)).T, dtype=F32)


def audio_chunking(audio, frame_rate=30, chunk_size=16000):
    """`utils.py:133-145`: (1, N) samples -> (chunks, chunk_size): one window of `chunk_size` samples per video frame, centred on it
    (the signal zero-padded by half a window minus half a frame on both sides)."""
    audio = np.asarray(audio, dtype=F32).reshape(1, -1)
    spf = chunk_size // frame_rate
    pad = (chunk_size - spf) // 2
    x = np.pad(audio, ((0, 0), (pad, pad)))
    anchors = range(chunk_size // 2, x.shape[-1] - chunk_size // 2, spf)
    return np.concatenate([x[:, i - chunk_size // 2:i + chunk_size // 2] for i in anchors], axis=0)


def mfcc(wave, sample_rate, n_mfcc=64, n_fft=2048, hop_length=734, n_mels=256, top_db=80.0):
    """torchaudio.transforms.MFCC(sample_rate, n_mfcc, melkwargs={n_fft, n_mels, hop_length, mel_scale='htk'}) on a
    mono waveform (N,) -> (n_mfcc, T) with T = N // hop + 1 (`center=True`)."""
    return mfcc_from_power(power_spectrogram(wave, n_fft, hop_length), sample_rate, n_mfcc, n_fft, n_mels, top_db)


def mfcc_float64(wave, sample_rate, n_mfcc=64, n_fft=2048, hop_length=734, n_mels=256, top_db=80.0):
    """`mfcc` above carried out in float64 from the (float32) samples on: the yardstick for what fp32 arithmetic costs in the
    device front-end (bench.py `frontend.stability`, tests/test_gpu_parity.py::test_wav_in_code_stability).  -> (n_mfcc, T) float64."""
    x = np.pad(np.asarray(wave, dtype=np.float64), (n_fft // 2, n_fft // 2), mode="reflect")
    T = 1 + (x.shape[0] - n_fft) // hop_length
    frames = np.lib.stride_tricks.sliding_window_view(x, n_fft)[::hop_length][:T]
    window = 0.5 - 0.5 * np.cos(2.0 * math.pi * np.arange(n_fft) / n_fft)
    spec = np.fft.rfft(frames * window[None, :], axis=1)
    power = spec.real ** 2 + spec.imag ** 2
    mel = power @ melscale_fbanks(n_fft // 2 + 1, 0.0, float(sample_rate // 2), n_mels, sample_rate).astype(np.float64)
    db = 10.0 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - top_db)
    return np.ascontiguousarray((db @ create_dct(n_mfcc, n_mels).astype(np.float64)).T)


# ---- onset times (`utils.py:200-201`: librosa.onset.onset_detect(y, sr=16000, units='time'); the beat-consistency score of
# scripts/test_body.py:173 reads them) -----------------------------------------------------------------------------------------
def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3.0)
    log = 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (math.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((math.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3.0))


def slaney_mel_filters(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney') as published: triangles on the Slaney mel scale (linear
    below 1 kHz, logarithmic above), each scaled by 2 / (its band edges' distance in Hz).  -> (n_mels, n_fft // 2 + 1)."""
    fmax = sr / 2.0 if fmax is None else fmax
    fft_f = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    mel_f = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    lower, upper = -ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    return (w * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]).astype(F32)


def onset_strength(wave, sr, n_fft=2048, hop_length=512, n_mels=128):
    """librosa.onset.onset_strength(y, sr) with its defaults as published: mel power spectrogram (Slaney filters up to sr / 2, Hann
    window, centred frames) in dB (top_db 80), first difference along time, half-wave rectified, mean over the bands, shifted by
    lag + n_fft // (2 hop) frames so that a frame's value describes the frame it is reported at.  -> (T,)"""
    power = power_spectrogram(wave, n_fft, hop_length)                              # (T, n_fft // 2 + 1), reflect-padded frames
    mel = power @ slaney_mel_filters(sr, n_fft, n_mels).T                           # (T, n_mels)
    db = 10.0 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - 80.0)
    env = np.maximum(0.0, db[1:] - db[:-1]).mean(axis=1)
    pad = 1 + n_fft // (2 * hop_length)
    return np.concatenate([np.zeros(pad), env])[:power.shape[0]]


def peak_pick(x, pre_max, post_max, pre_avg, post_avg, delta, wait):
    """librosa.util.peak_pick as published: x[n] is a peak if it is the maximum of x[n - pre_max : n + post_max], at least
    mean(x[n - pre_avg : n + post_avg]) + delta, and more than `wait` samples after the previous peak."""
    x = np.asarray(x, dtype=np.float64)
    n_all, peaks, last = x.shape[0], [], -np.inf
    for n in range(n_all):
        if x[n] < x[max(n - pre_max, 0):min(n + post_max, n_all)].max():
            continue
        if x[n] < x[max(n - pre_avg, 0):min(n + post_avg, n_all)].mean() + delta:
            continue
        if n > last + wait:
            peaks.append(n)
            last = n
    return np.asarray(peaks, dtype=np.int64)


_onset_warned = False


def onset_times(wave, sr=16000, hop_length=512):
    """Onset times in seconds of a mono waveform: `librosa.onset.onset_detect(y=wave, sr=sr, units='time')` — by librosa ITSELF when
    it is importable (the reference's pin is librosa~=0.9.2; this image has none), else by the restatement above of its published
    algorithm (onset strength -> normalised to [0, 1] -> peak picking with the 30 ms / 100 ms / 0.07 defaults).  PARITY UNPINNED for
    the restatement: no librosa here to compare with and no fixture in the reference; it exists so that the reference's evaluation
    loop (scripts/test_body.py:173, beat consistency) runs without the package."""
    wave = np.asarray(wave, dtype=F32).reshape(-1)
    try:
        import librosa
        return np.asarray(librosa.onset.onset_detect(y=wave, sr=sr, units='time'), dtype=np.float64)
    except ImportError:
        global _onset_warned
        if not _onset_warned:               # once per process: what comes back is NOT librosa's output
            import warnings
            warnings.warn("talkshow_amd.frontend.onset_times: librosa is not installed; onset times come from a restatement of "
                          "librosa.onset.onset_detect whose parity with librosa ~= 0.9.2 is UNPINNED (peak_pick edge handling, STFT pad mode) — "
                          "beat-consistency numbers computed from them may differ from the reference's", RuntimeWarning, stacklevel=2)
            _onset_warned = True
    env = onset_strength(wave, sr, hop_length=hop_length)
    env = env - env.min()
    if not env.any():
        return np.zeros(0, dtype=np.float64)
    env = env / (env.max() + np.finfo(np.float64).tiny)
    peaks = peak_pick(env, pre_max=int(0.03 * sr // hop_length), post_max=int(0.00 * sr // hop_length + 1), pre_avg=int(0.10 * sr // hop_length),
                      post_avg=int(0.10 * sr // hop_length + 1), delta=0.07, wait=int(0.03 * sr // hop_length))
    return peaks.astype(np.float64) * hop_length / sr


def _hop(fps):
    if fps == 15:
        return 1467
    if fps == 30:
        return 734
    raise ValueError(f"fps must be 15 or 30 (data_utils/utils.py:157-160), got {fps}")


def _features_from_any(aud_fn):
    if isinstance(aud_fn, torch.Tensor):
        return aud_fn.detach().cpu().numpy()
    if isinstance(aud_fn, np.ndarray):
        return aud_fn
    if isinstance(aud_fn, (str, os.PathLike)) and str(aud_fn).endswith(".npy"):
        return np.load(aud_fn)
    return None


def _load_mono_resampled(aud_fn, sr):
    audio, sr_0 = load_wav(aud_fn)
    if sr != sr_0:
        audio = resample_sinc_hann(audio, sr_0, sr)
    if audio.shape[0] > 1:
        audio = audio.mean(axis=0, keepdims=True, dtype=F32)
    return audio[0]


_device_mfcc = {}


def _mfcc_on_device(wave_mono, sr_in, sr, fps):
    """MI355X path of the front-end (ts_mfcc_*): resample + MFCC on the GPU when one is present."""
    from .modules import MFCC
    key = (int(sr_in), int(sr), int(fps), torch.cuda.current_device())
    if key not in _device_mfcc:
        _device_mfcc[key] = MFCC(sr_in, sr, fps)
    return _device_mfcc[key](wave_mono)[0].cpu().numpy()


def get_mfcc_ta(aud_fn, eps=1e-6, fps=15, smlpx=False, sr=16000, n_mfcc=64, win_size=None, type='mfcc', am=None, am_sr=None,
                encoder_choice='mfcc', host=None):
    """`get_mfcc_ta` (`utils.py:148-231`), body branch: -> (T, 64) float32 features.

    wav files: resample + MFCC run on the GPU (ts_mfcc_forward) when a HIP device is present, else on the host in numpy
    (`host=True` forces the numpy path, which is also the checker of the device path in tests/).  With a processor handed in
    (`am is not None`) the reference switches on `encoder_choice` (`utils.py:193-202`): 'faceformer' -> `get_wav16`,
    'meshtalk' -> scaled samples, 'onset' -> onset times in seconds (K, 1) (`onset_times`: librosa itself when it is installed, else a
    restatement), anything else -> the MFCC features as without `am`."""
    if am is not None and encoder_choice in ('faceformer', 'meshtalk', 'onset'):
        # the reference's `am is not None` branch (`utils.py:193-202`): librosa.load(sr=16000), then a switch on encoder_choice
        if encoder_choice == 'faceformer':                      # raw 16 kHz samples (N, 1): the face generator's input
            return get_wav16(aud_fn, host=host)
        if encoder_choice == 'meshtalk':                        # `0.01 * speech_array / np.mean(np.abs(speech_array))`, shape (N,)
            x = get_wav16(aud_fn, host=host)[:, 0]
            return (F32(0.01) * x / np.mean(np.abs(x))).astype(F32)
        return onset_times(get_wav16(aud_fn, host=host)[:, 0], 16000).reshape(-1, 1)   # `utils.py:200-201`, used by test_body.py:173
    feat = _features_from_any(aud_fn)
    if feat is None and type in ('mel', 'mel_mul'):
        # the two other feature types of `utils.py:178-191` (no shipped config asks for them: they run on the host, in numpy)
        wave = _load_mono_resampled(aud_fn, sr)
        if type == 'mel':                                               # (T, 256) mel power spectrogram
            return mel_spectrogram(wave, sr, hop_length=_hop(fps)).T.copy()
        wave = F32(0.01) * wave / np.mean(np.abs(wave), dtype=F32)      # 'mel_mul': one 1 s window per video frame, 10 ms hop, log
        chunks = audio_chunking(wave, frame_rate=fps, chunk_size=sr)
        mels = np.stack([mel_spectrogram(c, sr, hop_length=int(sr / 100), win_length=int(sr / 20)) for c in chunks])
        return np.log(np.maximum(mels, F32(1e-10))).astype(F32)         # (chunks, 256, 101)
    if feat is None:
        if type != 'mfcc':
            raise NotImplementedError(f"get_mfcc_ta: unknown feature type {type!r} (the reference knows 'mfcc', 'mel', 'mel_mul')")
        use_host = host if host is not None else not torch.cuda.is_available()
        if use_host:
            feat = mfcc(_load_mono_resampled(aud_fn, sr), sr, hop_length=_hop(fps)).T
        else:
            audio, sr_0 = load_wav(aud_fn)
            _hop(fps)
            feat = _mfcc_on_device(audio.mean(axis=0, dtype=F32), sr_0, sr, fps)
    feat = np.asarray(feat, dtype=np.float32)
    if feat.ndim != 2 or feat.shape[1] != 64:
        raise ValueError(f"MFCC features must have shape (T, 64), got {feat.shape}")
    return feat


def get_mfcc_sepa(aud_fn, fps=15, sr=16000, host=None):
    """`get_mfcc_sepa` (`utils.py:234-263`): MFCCs of the first 2 s and of the rest, concatenated, + the split frame."""
    feat = _features_from_any(aud_fn)
    if feat is not None:
        feat = np.asarray(feat, dtype=np.float32)
        return feat, 1 + (2 * sr) // _hop(fps)      # MFCC length of the first 2 s (center=True): the wav path's split
    use_host = host if host is not None else not torch.cuda.is_available()
    hop = _hop(fps)
    if use_host:
        wave = _load_mono_resampled(aud_fn, sr)
        f0 = mfcc(wave[:sr * 2], sr, hop_length=hop).T
        f1 = mfcc(wave[sr * 2:], sr, hop_length=hop).T
    else:   # resample the whole clip on the GPU, then the MFCC of the first 2 s and of the rest (two ts_mfcc_forward calls)
        from .modules import MFCC
        audio, sr_0 = load_wav(aud_fn)
        res = MFCC(sr_0, sr, fps)
        x = torch.stack([res.resample(torch.from_numpy(ch)[None])[0] for ch in audio]).mean(0)     # per channel, then mono
        plain = MFCC(sr, sr, fps)
        f0 = plain(x[:sr * 2])[0].cpu().numpy()
        f1 = plain(x[sr * 2:])[0].cpu().numpy()
    return np.concatenate((f0, f1), axis=0), f0.shape[0]


def get_wav16(aud_fn, host=None):
    """Face front-end: `get_mfcc_ta(..., encoder_choice='faceformer')` = `librosa.load(aud_fn, sr=16000)` reshaped to
    (N, 1), no normalisation (`data_utils/utils.py:194-198`).  Accepted: arrays / tensors of samples (taken to be 16 kHz
    already), `.npy`, and PCM or float `.wav` files of any sample rate: mono = mean of channels, int PCM scaled to [-1, 1),
    then librosa's default `kaiser_best` resampling to 16 kHz — on the GPU (`ts_resample_kaiser`) when one is present,
    else / with `host=True` in numpy (`resample_kaiser_best`, the device kernel's checker)."""
    if isinstance(aud_fn, torch.Tensor):
        x = aud_fn.detach().cpu().numpy()
    elif isinstance(aud_fn, np.ndarray):
        x = aud_fn
    elif str(aud_fn).endswith(".npy"):
        x = np.load(aud_fn)
    elif str(aud_fn).endswith(".wav"):
        audio, sr = load_wav(aud_fn)                        # (channels, N) float32
        x = audio.mean(axis=0, dtype=F32) if audio.shape[0] > 1 else audio[0]      # librosa.load: mono first ...
        if sr != 16000:                                                              # ... then resample (kaiser_best)
            if host is None:
                host = not torch.cuda.is_available()
            if host:
                x = resample_kaiser_best(x, sr, 16000)
            else:
                from .modules import resample_kaiser_device
                x = resample_kaiser_device(x[None], sr, 16000)[0].cpu().numpy()
    else:
        raise NotImplementedError(f"audio front-end: cannot read {aud_fn!r}")
    return np.asarray(x, dtype=np.float32).reshape(-1, 1)
