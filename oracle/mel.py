"""Oracle restatement of the audio front-end (numpy float64, vectorised).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Reference (paths relative to /root/reference):
  ZEGGS/audio/spectrograms.py:216-269  extract_spectrogram (Hann 800, reflect pad 400/400, rfft, |.|/n_fft)
  ZEGGS/audio/spectrograms.py:386-503  Slaney mel filterbank (_hz_to_mel/_mel_to_hz)
  ZEGGS/audio/spectrograms.py:57-131   clip at min_amplitude/n_fft, 20 log10, map to [0, 1]
  ZEGGS/data_pipeline.py:28-84         preprocess_audio: 10**(x/20) -> ln -> linear resample -> + energy
  ZEGGS/generate.py:170                n_frames = int(round(60 * len / 16000))
Loudness normalisation (pyloudnorm, data_pipeline.py:34-39) is NOT restated
here: parity unpinned for that stage (dependency absent), golden vectors use
normalize_loudness=false.
"""
import numpy as np


def hann_symmetric(n):
    """scipy.signal.hann(n) == windows.hann(n, sym=True)"""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / (n - 1))


def stft_frame_count(n_samples, n_fft=800, hop=200, centered=True):
    """spectrograms.py:233-246 (integer rule, bit-exact)."""
    n = max(n_samples, n_fft) + (2 * (n_fft // 2) if centered else 0)
    if n % hop == 0:
        return int((n - n_fft) // hop)
    return 1 + int((n - n_fft) // hop)


def n_anim_frames(n_samples, fs=16000, fps=60.0):
    """generate.py:170 (Python round = banker's rounding)."""
    return int(round(fps * (n_samples / fs)))


def hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(n_fft=800, fs=16000, n_mels=80, fmin=20.0, fmax=7600.0, normalize=True):
    """spectrograms.py:386-443 -> [n_mels, n_fft//2+1] float64"""
    nb = 1 + n_fft // 2
    fft_freqs = np.linspace(0, fs / 2.0, nb, endpoint=True)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fft_freqs)
    w = np.zeros((n_mels, nb))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    if normalize:
        w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


def mel_spectrogram(wav, n_fft=800, hop=200, fs=16000, n_mels=80, fmin=20.0, fmax=7600.0,
                    min_clip=1e-5, normalize_mel_bins=True, pre_emph=0.0, real_amplitude=True, centered=True, normalize_range=True):
    """extract_mel_spectrogram_for_tts (spectrograms.py:8-54) with the shipped
    conf (centered, real_amplitude, normalize_mel_bins, normalize_range, no
    pre-emphasis): returns [n_mels, M] in [0, ~1]."""
    x = np.asarray(wav, dtype=np.float64)
    if pre_emph:                                                   # signal_manipulation.py:4-12: lfilter([1, -c], [1], x)
        x = np.concatenate([x[:1], x[1:] - pre_emph * x[:-1]])
    if len(x) < n_fft:
        x = np.pad(x, (0, n_fft - len(x)))
    if centered:
        x = np.pad(x, (n_fft // 2, n_fft // 2), mode="reflect")
    M = stft_frame_count(len(wav), n_fft, hop, centered)
    idx = np.arange(M)[:, None] * hop + np.arange(n_fft)[None, :]
    frames = x[idx] * hann_symmetric(n_fft)[None, :]
    amp = np.abs(np.fft.rfft(frames, axis=1)).T / (n_fft if real_amplitude else 1)            # [401, M]
    mel = mel_filterbank(n_fft, fs, n_mels, fmin, fmax, normalize_mel_bins) @ amp      # [80, M]
    amin = min_clip / (n_fft if real_amplitude else 1)              # spectrograms.py:81-88
    mel = np.clip(np.abs(mel), amin, None)
    db = 20.0 * np.log10(mel)
    rng = -20.0 * np.log10(amin)
    return (db + rng) / rng if normalize_range else db


def _notaknot_second_derivatives(y):
    """Second derivatives S_i of the cubic spline through (i, y_i), i = 0..M-1, with not-a-knot ends -- what
    scipy.interpolate.interp1d(kind="cubic") = make_interp_spline(k=3, bc_type=None) interpolates with (reference
    data_pipeline.py:65-79 through griddata / interp1d).  y: [M, C] float64.  Interior equations
    S_{i-1} + 4 S_i + S_{i+1} = 6 (y_{i-1} - 2 y_i + y_{i+1}); the end conditions S_0 - 2 S_1 + S_2 = 0 (and its mirror)
    reduce the first and last equation to S_1 = d_1, S_{M-2} = d_{M-2}; the rest is eliminated by the Thomas recurrence."""
    M = len(y)
    if M < 4:
        raise ValueError("cubic interpolation needs at least 4 points")
    d = np.zeros_like(y)
    d[1:-1] = (y[:-2] - y[1:-1]) - (y[1:-1] - y[2:])
    S = np.zeros_like(y)
    S[1], S[M - 2] = d[1], d[M - 2]
    lo, hi = 2, M - 3
    if hi >= lo:
        rhs = 6.0 * d[lo:hi + 1].copy()
        rhs[0] -= S[1]
        rhs[-1] -= S[M - 2]
        n = hi - lo + 1
        cp = np.zeros(n)
        dp = np.zeros_like(rhs)
        cp[0] = 0.25
        dp[0] = rhs[0] / 4.0
        for i in range(1, n):
            den = 4.0 - cp[i - 1]
            cp[i] = 1.0 / den
            dp[i] = (rhs[i] - dp[i - 1]) / den
        S[hi] = dp[-1]
        for i in range(n - 2, -1, -1):
            S[lo + i] = dp[i] - cp[i] * S[lo + i + 1]
    S[0] = 2.0 * S[1] - S[2]
    S[M - 1] = 2.0 * S[M - 2] - S[M - 3]
    return S


def _resample(y, t, method, extrapolate):
    """y [M, C] float64 on the integer grid -> values at t; outside [0, M-1]: NaN (griddata) or, with `extrapolate`
    (interp1d(fill_value="extrapolate")), the end pieces continued.  "nearest" always clamps (griddata switches its fill value to
    "extrapolate" for that method), halves go DOWN (interp1d searches its bounds x_i + 1/2 from the left)."""
    M = len(y)
    if method == "nearest":
        return y[np.clip(np.ceil(t - 0.5).astype(np.int64), 0, M - 1)]
    lo = np.clip(np.ceil(t).astype(np.int64) - 1, 0, M - 2)
    u = (t - lo)[:, None]
    if method == "linear":
        out = (y[lo + 1] - y[lo]) * u + y[lo]
    elif method == "cubic":
        S = _notaknot_second_derivatives(y)
        v = 1.0 - u
        out = y[lo] * v + y[lo + 1] * u + ((v ** 3 - v) * S[lo] + (u ** 3 - u) * S[lo + 1]) / 6.0
    else:
        raise ValueError(f"Unknown interpolation method {method!r} for 1 dimensional data")
    if not extrapolate:
        out = out.copy()
        out[(t < 0) | (t > M - 1)] = np.nan
    return out


def preprocess_audio(wav, n_frames, fs=16000, hop=200, fps=60.0, resample_method="linear", **kw):
    """data_pipeline.py:33-84 with feature_type = [mel_spec, energy], normalize_loudness = false; resample_method as
    audio_conf.resample_method ("linear" in every shipped configuration).  -> [n_frames, 81] float32"""
    mel = mel_spectrogram(wav, hop=hop, fs=fs, **kw).T              # [M, 80]
    mel = np.log(10.0 ** (mel / 20.0))
    t = ((fs / hop) / fps) * np.arange(n_frames)
    mel_i = _resample(mel, t, resample_method, extrapolate=False)      # griddata: NaN outside the hull [0, M-1]
    energy = np.linalg.norm(np.exp(mel), axis=1)                     # [M]
    en_i = _resample(energy[:, None], t, resample_method, extrapolate=True)[:, 0]
    return np.concatenate([mel_i.astype(np.float32), en_i.astype(np.float32)[:, None]], axis=1)
