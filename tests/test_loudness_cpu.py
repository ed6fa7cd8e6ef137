"""CPU tests of the loudness stage (ZEGGS/data_pipeline.py:34-39 -> pyloudnorm==0.1.0, absent here): the product
implementation (zeggs/audio.py, host + device versions share the block / gate rule) against the oracle restatement
of pyloudnorm's published algorithm on noise, speech-like and silence-containing signals, and the oracle itself
against known answers of BS.1770.  PARITY UNPINNED against pyloudnorm itself (oracle/loudness.py header)."""
import numpy as np
import pytest

from oracle import loudness as olo
from zeggs import audio, synth


def _signals():
    rng = np.random.default_rng(3)
    fs = 16000
    speech = synth.synth_wav(5 * fs + 123, seed=4).astype(np.float64) / 32768.0
    noise = 0.05 * rng.standard_normal(3 * fs)
    gap = np.concatenate([speech[:2 * fs], np.zeros(3 * fs), 0.3 * speech[2 * fs:4 * fs], np.zeros(2 * fs + 77)])
    quiet_tail = np.concatenate([noise, 1e-5 * rng.standard_normal(4 * fs)])     # blocks under the absolute gate
    short = speech[:int(0.4 * fs) + 1]                                            # just over one gating block
    return fs, dict(speech=speech, noise=noise, gap=gap, quiet_tail=quiet_tail, short=short)


def test_product_loudness_matches_pyloudnorm_restatement():
    fs, sigs = _signals()
    for name, x in sigs.items():
        ref = olo.Meter(fs).integrated_loudness(x)
        got = audio.integrated_loudness(x, fs)
        assert abs(got - ref) < 1e-9, (name, got, ref)
        y_ref = olo.preprocess_loudness(x, fs, -20.0)
        y = audio.normalize_loudness(x, fs, -20.0)
        np.testing.assert_allclose(y, y_ref, rtol=1e-10, atol=0, err_msg=name)
    for rate in (44100, 48000):                      # other sample rates re-derive the biquads
        x = 0.1 * np.random.default_rng(rate).standard_normal(2 * rate)
        assert abs(audio.integrated_loudness(x, rate) - olo.Meter(rate).integrated_loudness(x)) < 1e-9
    st = np.stack([sigs["speech"], 0.5 * sigs["speech"]], axis=1)
    assert abs(audio.integrated_loudness(st, fs) - olo.Meter(fs).integrated_loudness(st)) < 1e-9


def test_oracle_loudness_known_answers():
    fs = 48000
    t = np.arange(fs * 5) / fs
    x = np.sin(2 * np.pi * 997.0 * t)
    m = olo.Meter(fs)
    l0 = m.integrated_loudness(x)
    assert abs(l0 - (-3.01)) < 0.05                                   # BS.1770: 997 Hz full scale = -3.01 LKFS
    assert abs(m.integrated_loudness(0.1 * x) - (l0 - 20.0)) < 1e-9   # dB for dB
    assert abs(m.integrated_loudness(np.stack([x, x], 1)) - (l0 + 10 * np.log10(2.0))) < 1e-9
    y = olo.preprocess_loudness(x, fs, -20.0)
    assert abs(m.integrated_loudness(y) - (-20.0)) < 1e-9


def test_loudness_edge_cases_follow_pyloudnorm():
    fs = 16000
    with pytest.raises(ValueError):                                    # pyloudnorm.util.valid_audio
        olo.Meter(fs).integrated_loudness(np.zeros(int(0.4 * fs) - 1))
    with pytest.raises(ValueError):
        audio.normalize_loudness(np.zeros(int(0.4 * fs) - 1), fs, -20.0)
    with pytest.raises(ValueError, match="finite"):                   # digital silence has no loudness: refuse the inf gain
        audio.normalize_loudness(np.zeros(2 * fs), fs, -20.0)
    assert olo.Meter(fs).integrated_loudness(np.zeros(2 * fs)) == -np.inf
