"""CPU tests of the host-side engine logic (integer index rules are bit-exact vs the reference)."""
import numpy as np
import torch

from zeggs import engine, synth


def test_device_dataset_index_rules_vs_reference(golden_dir):
    g = np.load(golden_dir / "dataset.npz")
    window, n_total = int(g["window"]), int(g["n_total"])
    stats = synth.make_stats()
    data = synth.make_processed(3, 1, 40, seed=2, stats=stats)
    assert len(data["Y_root_pos"]) == n_total
    ds = engine.DeviceDataset(data, window, torch.device("cpu"))
    np.testing.assert_array_equal(ds.win_start, g["R0"])
    np.testing.assert_array_equal(ds.win_sample, g["S"])
    for q in g["queries"]:
        ex_len, idx, nrows = int(q[0]), int(q[1]), int(q[2])
        rows = ds.example_rows(np.array([idx]), ex_len)[0]
        assert len(rows) == nrows == ex_len
        np.testing.assert_array_equal(rows, q[3:3 + nrows])


def test_flatten_parameters_keeps_state_dict_and_views():
    import helpers
    se, de, st = helpers.build_nets()
    before = {k: v.clone() for k, v in de.state_dict().items()}
    params, flat_p, flat_g, flat_gx = engine.flatten_parameters([se, de, st])
    assert flat_gx.numel() % 4 == 0 and flat_gx.numel() >= flat_g.numel() + 4 and flat_gx.data_ptr() == flat_g.data_ptr()
    assert flat_p.numel() == 25543147 == sum(p.numel() for p in params)
    for k, v in de.state_dict().items():
        assert torch.equal(v, before[k])
    flat_p.zero_()
    assert all(float(p.abs().sum()) == 0 for p in params)          # parameters are views of the flat buffer
    flat_g.fill_(2.0)
    assert all(float(p.grad.min()) == 2.0 for p in params)


def test_mel_filterbank_matches_oracle():
    from oracle import mel as omel
    from zeggs import audio
    np.testing.assert_allclose(audio.mel_filterbank(800, 16000, 80, 20.0, 7600.0), omel.mel_filterbank(), atol=1e-15)
    assert (audio.mel_filterbank(800, 16000, 80, 20.0, 7600.0) > 0).sum() == 742
    assert audio.n_anim_frames(16123) == omel.n_anim_frames(16123)


def test_bs1770_loudness_analytic_case():
    """BS.1770: a 997 Hz full-scale sine measures -3.01 LKFS; normalising to -20 LUFS applies the matching gain."""
    from zeggs import audio
    fs = 48000
    t = np.arange(fs * 5) / fs
    x = np.sin(2 * np.pi * 997.0 * t)
    assert abs(audio.integrated_loudness(x, fs) - (-3.01)) < 0.05
    y = audio.normalize_loudness(x, fs, -20.0)
    assert abs(audio.integrated_loudness(y, fs) - (-20.0)) < 1e-6
