"""Oracle restatement of the loudness stage of preprocess_audio (ZEGGS/data_pipeline.py:34-39):

    meter = pyln.Meter(rate); loudness = meter.integrated_loudness(x); x = pyln.normalize.loudness(x, loudness, -20.0)

TEST INFRASTRUCTURE -- see oracle/__init__.py.

The arithmetic lives in a THIRD-PARTY dependency that is absent from /root/reference and from this image:
`pyloudnorm==0.1.0` (ZEGGS requirements.txt).  This file restates that release's published algorithm
(pyloudnorm/meter.py `Meter.integrated_loudness`, pyloudnorm/iirfilter.py `IIRfilter`, pyloudnorm/normalize.py
`loudness`, pyloudnorm/util.py `valid_audio`), i.e. ITU-R BS.1770-4 as pyloudnorm implements it -- note that its
K-weighting biquads are NOT the recommendation's 48 kHz coefficient table but are re-derived per sample rate:

  stage 1  high shelf  G = +4 dB, Q = 1/sqrt(2), fc = 1500 Hz      A = 10^(G/40), w0 = 2 pi fc / rate,
  stage 2  high pass   G =  0 dB, Q = 0.5,       fc = 38 Hz        alpha = sin(w0) / (2 Q)
  (coefficient formulas below), each applied with scipy.signal.lfilter (direct form II transposed, float64),
  passband gain 1;
  gating blocks of T_g = 0.4 s with 75 % overlap: numBlocks = int(round((T - T_g) / (T_g * step)) + 1),
  block j spans samples [int(T_g * (j * step) * rate), int(T_g * (j * step + 1) * rate))  -- floating-point
  products truncated by int(), reproduced here literally;
  z[i, j] = sum(y_i[l:u]^2) / (T_g * rate);  l_j = -0.691 + 10 log10(sum_i G_i z[i, j]),  G = [1, 1, 1, 1.41, 1.41];
  absolute gate l_j >= -70;  relative gate Gamma_r = -0.691 + 10 log10(sum_i G_i mean_{J_abs} z[i, j]) - 10;
  result -0.691 + 10 log10(sum_i G_i mean_{l_j > Gamma_r and l_j > -70} z[i, j])  (NaN means -> 0 -> -inf).
  normalize.loudness: gain = 10^((target - measured) / 20), output = gain * data (a warning, not an error, if it clips).
  valid_audio: raises ValueError when the signal is not longer than one gating block.

PARITY UNPINNED: no pyloudnorm here, no golden vectors in the reference; the restatement is checked against
known answers of the standard only (a 997 Hz full-scale sine at 48 kHz measures -3.01 LKFS; level shifts move the
result dB for dB; stereo adds 3.01 dB) -- see tests/test_loudness_cpu.py and DESIGN.md.
"""
import warnings

import numpy as np
from scipy import signal


class IIRfilter:
    """pyloudnorm/iirfilter.py"""

    def __init__(self, G, Q, fc, rate, filter_type, passband_gain=1.0):
        self.G, self.Q, self.fc, self.rate = G, Q, fc, rate
        self.filter_type, self.passband_gain = filter_type, passband_gain
        self.b, self.a = self.generate_coefficients()

    def generate_coefficients(self):
        A = 10 ** (self.G / 40.0)
        w0 = 2.0 * np.pi * (self.fc / self.rate)
        alpha = np.sin(w0) / (2.0 * self.Q)
        if self.filter_type == "high_shelf":
            b0 = A * ((A + 1) + (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha)
            b1 = -2 * A * ((A - 1) + (A + 1) * np.cos(w0))
            b2 = A * ((A + 1) + (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha)
            a0 = (A + 1) - (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha
            a1 = 2 * ((A - 1) - (A + 1) * np.cos(w0))
            a2 = (A + 1) - (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha
        elif self.filter_type == "high_pass":
            b0 = (1 + np.cos(w0)) / 2
            b1 = -(1 + np.cos(w0))
            b2 = (1 + np.cos(w0)) / 2
            a0 = 1 + alpha
            a1 = -2 * np.cos(w0)
            a2 = 1 - alpha
        else:
            raise ValueError("Invalid filter type", self.filter_type)
        return np.array([b0, b1, b2]) / a0, np.array([a0, a1, a2]) / a0

    def apply_filter(self, data):
        return self.passband_gain * signal.lfilter(self.b, self.a, data)


def valid_audio(data, rate, block_size):
    """pyloudnorm/util.py"""
    if not isinstance(data, np.ndarray):
        raise ValueError("Data must be of type numpy.ndarray.")
    if not np.issubdtype(data.dtype, np.floating):
        raise ValueError("Data must be floating point.")
    if data.ndim == 2 and data.shape[1] > 5:
        raise ValueError("Audio must have five channels or less.")
    if data.shape[0] < block_size * rate:
        raise ValueError("Audio must have length greater than the block size.")
    return True


class Meter:
    """pyloudnorm/meter.py (filter_class "K-weighting")"""

    def __init__(self, rate, block_size=0.400):
        self.rate, self.block_size = rate, block_size
        self._filters = {"high_shelf": IIRfilter(4.0, 1 / np.sqrt(2), 1500.0, rate, "high_shelf"),
                         "high_pass": IIRfilter(0.0, 0.5, 38.0, rate, "high_pass")}

    def integrated_loudness(self, data):
        input_data = data.copy()
        valid_audio(input_data, self.rate, self.block_size)
        if input_data.ndim == 1:
            input_data = np.reshape(input_data, (input_data.shape[0], 1))
        numChannels, numSamples = input_data.shape[1], input_data.shape[0]
        for _, filter_stage in self._filters.items():
            for ch in range(numChannels):
                input_data[:, ch] = filter_stage.apply_filter(input_data[:, ch])
        G = [1.0, 1.0, 1.0, 1.41, 1.41]
        T_g = self.block_size
        Gamma_a = -70.0
        overlap = 0.75
        step = 1.0 - overlap
        T = numSamples / self.rate
        numBlocks = int(np.round(((T - T_g) / (T_g * step))) + 1)
        j_range = np.arange(0, numBlocks)
        z = np.zeros(shape=(numChannels, numBlocks))
        for i in range(numChannels):
            for j in j_range:
                lo = int(T_g * (j * step) * self.rate)
                hi = int(T_g * (j * step + 1) * self.rate)
                z[i, j] = (1.0 / (T_g * self.rate)) * np.sum(np.square(input_data[lo:hi, i]))
        with np.errstate(divide="ignore"):
            l = [-0.691 + 10.0 * np.log10(np.sum([G[i] * z[i, j] for i in range(numChannels)])) for j in j_range]  # noqa: E741
        J_g = [j for j, l_j in enumerate(l) if l_j >= Gamma_a]
        with np.errstate(divide="ignore", invalid="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            z_avg_gated = [np.mean([z[i, j] for j in J_g]) for i in range(numChannels)]
            Gamma_r = -0.691 + 10.0 * np.log10(np.sum([G[i] * z_avg_gated[i] for i in range(numChannels)])) - 10.0
            J_g = [j for j, l_j in enumerate(l) if (l_j > Gamma_r and l_j > Gamma_a)]
            z_avg_gated = np.nan_to_num(np.array([np.mean([z[i, j] for j in J_g]) for i in range(numChannels)]))
            LUFS = -0.691 + 10.0 * np.log10(np.sum([G[i] * z_avg_gated[i] for i in range(numChannels)]))
        return LUFS


def normalize_loudness(data, input_loudness, target_loudness):
    """pyloudnorm/normalize.py: loudness()"""
    gain = np.power(10.0, (target_loudness - input_loudness) / 20.0)
    output = gain * data
    if np.max(np.abs(output)) >= 1.0:
        warnings.warn("Possible clipped samples in output.")
    return output


def preprocess_loudness(audio_data, rate, target=-20.0):
    """the three lines of ZEGGS/data_pipeline.py:36-39"""
    meter = Meter(rate)
    return normalize_loudness(audio_data, meter.integrated_loudness(audio_data), target)
