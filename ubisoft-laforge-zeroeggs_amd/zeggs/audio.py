"""Audio front-end: wav -> [n_frames, 81] features, computed by the HIP mel kernel.

Mirrors the call surface the reference uses on this path: `preprocess_audio(audio, anim_fs, anim_length,
params, feature_type)` of ZEGGS/data_pipeline.py:33 (params = the `audio_conf` block of
data_pipeline_conf.json) and the frame-count rules.  Host side keeps only table construction (mel
filterbank) and the optional BS.1770 loudness gain.
"""
import ctypes as C

import numpy as np
import torch

from . import ops


class MelDims(C.Structure):
    _fields_ = [("n_fft", C.c_int), ("hop", C.c_int), ("n_mels", C.c_int), ("fs", C.c_int), ("fps", C.c_float),
                ("min_clip", C.c_float), ("pre_emph", C.c_double), ("flags", C.c_int)]


RESAMPLE_FLAGS = {"linear": 0, "nearest": 4, "cubic": 8}


def mel_flags(centered=True, normalize_range=True, resample_method="linear"):
    """ZeggsMelDims.flags: bit 0 = NOT centered, bit 1 = NOT normalize_range, bits 2-3 = resample_method (the three kinds
    scipy.interpolate.griddata takes for 1-D points, reference data_pipeline.py:65-70)"""
    if resample_method not in RESAMPLE_FLAGS:
        raise ValueError(f"Unknown interpolation method {resample_method!r} for 1 dimensional data")      # (griddata's own message)
    return (0 if centered else 1) | (0 if normalize_range else 2) | RESAMPLE_FLAGS[resample_method]


def n_anim_frames(n_samples, fs=16000, fps=60.0):
    """reference generate.py:170 (banker's rounding of Python round)"""
    return int(round(fps * (n_samples / fs)))


def _slaney_hz_to_mel(f):
    f = np.atleast_1d(np.asarray(f, dtype=np.float64))
    lin = f / (200.0 / 3)
    log_part = 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log_part, lin)


def _slaney_mel_to_hz(m):
    m = np.atleast_1d(np.asarray(m, dtype=np.float64))
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3) * m)


def mel_filterbank(n_fft, fs, n_mels, fmin, fmax, normalize=True):
    """Triangular Slaney filterbank [n_mels, n_fft//2+1] float64 (reference spectrograms.py:386-443)."""
    nbins = n_fft // 2 + 1
    bin_hz = np.linspace(0.0, fs / 2.0, nbins)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin)[0], _slaney_hz_to_mel(fmax)[0], n_mels + 2))
    width = np.diff(edges)
    dist = edges[:, None] - bin_hz[None, :]
    rising = -dist[:-2] / width[:-1, None]
    falling = dist[2:] / width[1:, None]
    fb = np.clip(np.minimum(rising, falling), 0.0, None)
    if normalize:
        fb = fb * (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return np.ascontiguousarray(fb)


_FB_CACHE = {}


def mel_tables(n_fft, fs, n_mels, fmin, fmax, min_clip, normalize_mel_bins=True, real_amplitude=True, device="cuda"):
    """-> (filterbank on `device`, the min_clip the kernel is given).  The kernel divides amplitudes and the clip floor by n_fft
    (audio_conf.real_amplitude, spectrograms.py:266-267, 81-88); real_amplitude = false is the same arithmetic with the filterbank
    and the floor multiplied by n_fft on the host (exact for the float64 table)."""
    dev = torch.device(device)
    key = (n_fft, fs, n_mels, fmin, fmax, normalize_mel_bins, bool(real_amplitude), str(dev))
    if key not in _FB_CACHE:
        fb = mel_filterbank(n_fft, fs, n_mels, fmin, fmax, normalize_mel_bins)
        _FB_CACHE[key] = torch.as_tensor(fb if real_amplitude else fb * float(n_fft)).to(dev)
    return _FB_CACHE[key], (float(min_clip) if real_amplitude else float(min_clip) * float(n_fft))


def mel_features(wav, n_frames, n_fft=800, hop=200, n_mels=80, fs=16000, fps=60.0, fmin=20.0, fmax=7600.0,
                 min_clip=1e-5, normalize_mel_bins=True, device="cuda", pre_emph=0.0, real_amplitude=True, centered=True,
                 normalize_range=True, resample_method="linear"):
    """wav: float array/tensor [n] -> torch float32 [n_frames, n_mels + 1] on `device` (HIP kernel)."""
    dev = torch.device(device)
    fb, min_clip = mel_tables(n_fft, fs, n_mels, fmin, fmax, min_clip, normalize_mel_bins, real_amplitude, dev)
    w = torch.as_tensor(np.asarray(wav, dtype=np.float32) if not torch.is_tensor(wav) else wav,
                        dtype=torch.float32).to(dev).contiguous()
    d = MelDims(n_fft, hop, n_mels, fs, float(fps), float(min_clip), float(pre_emph), mel_flags(centered, normalize_range, resample_method))
    L = ops.lib()
    L.zeggs_mel_workspace_bytes.restype = C.c_size_t
    ws = torch.empty(int(L.zeggs_mel_workspace_bytes(C.byref(d), C.c_long(w.numel()))), dtype=torch.uint8, device=dev)
    out = torch.empty(n_frames, n_mels + 1, device=dev, dtype=torch.float32)
    rc = L.zeggs_mel_features(C.byref(d), C.c_void_p(w.data_ptr()), C.c_long(w.numel()), C.c_void_p(fb.data_ptr()),
                              int(n_frames), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                              C.c_size_t(ws.numel()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError("zeggs_mel_features: " + L.zeggs_last_error().decode())
    return out


def stft_frame_count(n_samples, n_fft=800, hop=200):
    d = MelDims(n_fft, hop, 80, 16000, 60.0, 1e-5, 0.0, 0)
    L = ops.lib()
    L.zeggs_mel_stft_frames.restype = C.c_long
    return int(L.zeggs_mel_stft_frames(C.byref(d), C.c_long(n_samples)))


# ----------------------------------------------------------------------------- BS.1770 loudness (host, O(n))
def integrated_loudness(x, rate):
    """ITU-R BS.1770-4 integrated loudness (mono/multi-channel) as pyloudnorm==0.1.0 computes it (the reference's
    dependency, ZEGGS/data_pipeline.py:34-39): K-weighting biquads re-derived for `rate`, 400 ms blocks with 75 %
    overlap (block bounds = truncated floating-point products, kept literally), absolute (-70) and relative (-10 LU)
    gates.  Checked against the restatement of pyloudnorm's published source in oracle/loudness.py; parity with
    pyloudnorm itself is UNPINNED (package absent, no golden vectors) -- see DESIGN.md."""
    from scipy import signal
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x[:, None]
    if x.shape[0] < 0.4 * rate:
        raise ValueError("Audio must have length greater than the block size.")       # pyloudnorm.util.valid_audio

    def biquad(kind, G, Q, fc):
        A = 10 ** (G / 40.0)
        w0 = 2.0 * np.pi * fc / rate
        alpha = np.sin(w0) / (2.0 * Q)
        if kind == "high_shelf":
            b0 = A * ((A + 1) + (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha)
            b1 = -2 * A * ((A - 1) + (A + 1) * np.cos(w0))
            b2 = A * ((A + 1) + (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha)
            a0 = (A + 1) - (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha
            a1 = 2 * ((A - 1) - (A + 1) * np.cos(w0))
            a2 = (A + 1) - (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha
        else:  # high pass
            b0, b1, b2 = (1 + np.cos(w0)) / 2, -(1 + np.cos(w0)), (1 + np.cos(w0)) / 2
            a0, a1, a2 = 1 + alpha, -2 * np.cos(w0), 1 - alpha
        return np.array([b0, b1, b2]) / a0, np.array([a0, a1, a2]) / a0

    y = x
    for kind, G, Q, fc in (("high_shelf", 4.0, 1 / np.sqrt(2), 1500.0), ("high_pass", 0.0, 0.5, 38.0)):
        b, a = biquad(kind, G, Q, fc)
        y = signal.lfilter(b, a, y, axis=0)
    T_g, overlap = 0.4, 0.75
    step = 1.0 - overlap
    T = x.shape[0] / rate
    nblocks = int(np.round((T - T_g) / (T_g * step)) + 1)
    gains = [1.0, 1.0, 1.0, 1.41, 1.41][:x.shape[1]]
    z = np.zeros((x.shape[1], max(nblocks, 0)))
    for j in range(nblocks):
        lo, hi = int(T_g * (j * step) * rate), int(T_g * (j * step + 1) * rate)
        z[:, j] = np.sum(np.square(y[lo:hi]), axis=0) / (T_g * rate)
    with np.errstate(divide="ignore"):
        l_blocks = -0.691 + 10.0 * np.log10(np.sum(np.array(gains)[:, None] * z, axis=0))
    keep = l_blocks >= -70.0
    if not np.any(keep):
        return -np.inf
    z_avg = np.mean(z[:, keep], axis=1)
    gamma_r = -0.691 + 10.0 * np.log10(np.sum(gains * z_avg)) - 10.0
    keep = (l_blocks > gamma_r) & (l_blocks > -70.0)
    z_avg = np.nan_to_num(np.mean(z[:, keep], axis=1)) if np.any(keep) else np.zeros(x.shape[1])
    with np.errstate(divide="ignore"):
        return float(-0.691 + 10.0 * np.log10(np.sum(gains * z_avg)))


def normalize_loudness(x, rate, target=-20.0):
    lufs = integrated_loudness(x, rate)
    if not np.isfinite(lufs):
        # pyloudnorm would return gain = inf and NaN samples for digital silence; refuse instead of emitting NaN features
        raise ValueError(f"integrated loudness is not finite ({lufs}): the signal is silent under the -70 LUFS gate")
    return np.asarray(x) * (10.0 ** ((target - lufs) / 20.0))


def _kweighting(rate):
    """pyloudnorm's K-weighting biquads for `rate`: [(b, a), (b, a)] with a[0] == 1 (oracle/loudness.py: IIRfilter)."""
    out = []
    for kind, G, Q, fc in (("high_shelf", 4.0, 1 / np.sqrt(2), 1500.0), ("high_pass", 0.0, 0.5, 38.0)):
        A = 10 ** (G / 40.0)
        w0 = 2.0 * np.pi * (fc / rate)
        alpha = np.sin(w0) / (2.0 * Q)
        if kind == "high_shelf":
            b0 = A * ((A + 1) + (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha)
            b1 = -2 * A * ((A - 1) + (A + 1) * np.cos(w0))
            b2 = A * ((A + 1) + (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha)
            a0 = (A + 1) - (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha
            a1 = 2 * ((A - 1) - (A + 1) * np.cos(w0))
            a2 = (A + 1) - (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha
        else:
            b0, b1, b2 = (1 + np.cos(w0)) / 2, -(1 + np.cos(w0)), (1 + np.cos(w0)) / 2
            a0, a1, a2 = 1 + alpha, -2 * np.cos(w0), 1 - alpha
        out.append((np.array([b0, b1, b2]) / a0, np.array([a0, a1, a2]) / a0))
    return out


def gating_blocks(n, rate):
    """[lo, hi) sample bounds of the 400 ms gating blocks with 75 % overlap of an n-sample signal (pyloudnorm 0.1.0 meter.py:
    `l = int(T_g * (j * step) * rate)`, `u = int(T_g * (j * step + 1) * rate)`): the same float64 operations in the same order on
    the whole index vector (18 000 blocks for 30 minutes: 34 ms as a Python loop, 0.3 ms so), truncated like int()."""
    T_g, step = 0.4, 1.0 - 0.75
    nblocks = int(np.round(((n / rate - T_g) / (T_g * step))) + 1)
    j = np.arange(nblocks)
    lo = np.trunc(T_g * (j * step) * rate).astype(np.int64)
    hi = np.minimum(np.trunc(T_g * (j * step + 1) * rate).astype(np.int64), n)
    return lo, hi


def normalize_loudness_device(wav, rate, target=-20.0, device="cuda", chunk=4096):
    """Loudness normalisation of a MONO signal on the device (zeggs_loudness_gain): chunk-parallel K-weighting filters,
    gating-block energies, gates and gain; returns (normalised float32 tensor on `device`, LUFS).  The only host work is
    the integer gating-block table (pyloudnorm's truncated floating-point bounds, computed with its expression)."""
    dev = torch.device(device)
    f32 = torch.is_tensor(wav) and wav.dtype == torch.float32 or (not torch.is_tensor(wav) and np.asarray(wav).dtype == np.float32)
    w = torch.as_tensor(np.asarray(wav, dtype=np.float32) if not torch.is_tensor(wav) else wav, dtype=torch.float32).to(dev).contiguous()
    n = w.numel()
    if w.dim() != 1:
        raise ValueError("normalize_loudness_device: mono signals only")
    if n < 0.4 * rate:
        raise ValueError("Audio must have length greater than the block size.")       # pyloudnorm.util.valid_audio
    lo, hi = gating_blocks(n, rate)
    nblocks = len(lo)
    coef, trans = [], []
    for b, a in _kweighting(rate):
        coef += [b[0], b[1], b[2], a[1], a[2]]
        trans += list(np.linalg.matrix_power(np.array([[-a[1], 1.0], [-a[2], 0.0]]), int(chunk)).ravel())
    coef, trans = (C.c_double * 10)(*coef), (C.c_double * 8)(*trans)
    L = ops.lib()
    L.zeggs_loudness_workspace_bytes.restype = C.c_size_t
    ws = torch.empty(int(L.zeggs_loudness_workspace_bytes(C.c_long(n), int(nblocks), C.c_long(chunk))), dtype=torch.uint8, device=dev)
    lo_d, hi_d = torch.as_tensor(lo).to(dev), torch.as_tensor(hi).to(dev)
    res = torch.empty(2, dtype=torch.float64, device=dev)
    gain = torch.empty(1, dtype=torch.float32, device=dev)
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    rc = L.zeggs_loudness_gain(ptr(w), C.c_long(n), int(rate), C.c_double(target), coef, trans, C.c_long(chunk), ptr(lo_d),
                               ptr(hi_d), int(nblocks), int(bool(f32)), ptr(res), ptr(gain), ptr(ws), C.c_size_t(ws.numel()),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError("zeggs_loudness_gain: " + L.zeggs_last_error().decode())
    lufs = float(res[0])
    if not np.isfinite(lufs):
        raise ValueError(f"integrated loudness is not finite ({lufs}): the signal is silent under the -70 LUFS gate")
    return ops.scale_copy(w, gain), lufs


def preprocess_audio(audio_data, anim_fs, anim_length, params, feature_type, device="cuda"):
    """Drop-in for reference data_pipeline.preprocess_audio (same arguments, returns float32 ndarray
    [anim_length, 81]); `params` may be a dict or an attribute-style config."""
    return preprocess_audio_device(audio_data, anim_fs, anim_length, params, feature_type, device).cpu().numpy()


def preprocess_audio_device(audio_data, anim_fs, anim_length, params, feature_type, device="cuda"):
    """preprocess_audio with the feature table left on the device (what generate_gesture() feeds the speech encoder)."""
    g = (lambda k: params[k]) if isinstance(params, dict) else (lambda k: getattr(params, k))
    if g("normalize_loudness"):
        # the device pre-pass measures float32 samples (what a 16-bit WAV read gives); float64 input keeps the host path,
        # which filters and gates in float64 exactly as pyloudnorm does on such input
        is32 = (audio_data.dtype == torch.float32) if torch.is_tensor(audio_data) else (np.asarray(audio_data).dtype == np.float32)
        if np.ndim(audio_data) == 1 and torch.device(device).type == "cuda" and is32:
            audio_data, _ = normalize_loudness_device(audio_data, g("sampling_rate"), -20.0, device)   # no host pass
        else:
            audio_data = normalize_loudness(audio_data, g("sampling_rate"), -20.0)
    feat = mel_features(audio_data, anim_length, n_fft=g("filter_length"), hop=g("hop_length"),
                        n_mels=g("n_mel_channels"), fs=g("sampling_rate"), fps=float(anim_fs), fmin=g("mel_fmin"),
                        fmax=g("mel_fmax"), min_clip=g("min_clipping"), normalize_mel_bins=g("normalize_mel_bins"),
                        device=device, pre_emph=float(g("pre_emph_coeff")) if g("pre_emphasis") else 0.0,
                        real_amplitude=bool(g("real_amplitude")), centered=bool(g("centered")),
                        normalize_range=bool(g("normalize_range")), resample_method=g("resample_method"))
    cols = []
    if "mel_spec" in feature_type:
        cols.append(feat[:, :-1])
    if "energy" in feature_type:
        cols.append(feat[:, -1:])
    return torch.cat(cols, dim=1)
