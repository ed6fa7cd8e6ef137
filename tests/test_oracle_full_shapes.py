"""CPU tests: the oracle restatement reproduces the reference at the BENCHMARKED shapes and with the reference's
REAL normalisation statistics (fixtures: oracle/make_golden_full.py).  The training-iteration cases run the
oracle's autograd at B=32 x 256 frames: about a minute on 8 cores."""
import numpy as np
import torch

import helpers
from oracle import loss as oloss
from oracle import mel as omel
from oracle import nets as onets
from zeggs import synth

NAMES = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "ltxy", "lvel", "lvrt")


def _rollout(de_sd, W, speech, style, s, dtype=torch.float32):
    c = lambda t: t.to(dtype)  # noqa: E731
    return onets.decoder_rollout(de_sd, c(W["Y_root_pos"][:, 0]), c(W["Y_root_rot"][:, 0]), c(W["Y_root_vel"][:, 0]),
                                 c(W["Y_root_vrt"][:, 0]), c(W["Y_lpos"][:, 0]), c(W["Y_ltxy"][:, 0]),
                                 c(W["Y_lvel"][:, 0]), c(W["Y_lvrt"][:, 0]), c(W["Y_gaze_pos"]), c(speech), c(style),
                                 s["in_mean"], s["in_std"], s["out_mean"], s["out_std"], synth.DT)


def test_real_stats_fixture_has_the_reference_dynamic_range():
    st = helpers.real_stats("v1")
    assert int((st["anim_output_std"] == 0).sum()) == 364 and float(st["anim_input_std"].max()) > 40
    assert st["anim_input_mean"].shape == (synth.POSE_IN,) and st["anim_output_std"].shape == (synth.POSE_OUT,)


def test_oracle_decoder_b32_t256_vs_reference(golden_dir):
    gd = np.load(golden_dir / "full_dec32.npz")
    _, de, _ = helpers.build_nets()
    for k, v in de.state_dict().items():
        np.testing.assert_allclose(helpers.fingerprint(v), gd[f"fp_decoder.{k}"], rtol=1e-12)
    B, T = int(gd["B"]), int(gd["T"])
    W, speech, style = helpers.full_decoder_inputs(helpers.real_stats("v1"), B, T, int(gd["seed"]))
    helpers.assert_inputs_match(gd, W, speech, style)
    with torch.no_grad():
        O = _rollout(helpers.sd(de), W, speech, style, helpers.real_stats_tensors("v1"))
    pose = helpers.pack_pose(*O[2:]).numpy()
    np.testing.assert_allclose(pose[:, gd["frames"]], gd["pose_frames"], atol=1e-4)
    np.testing.assert_allclose(O[5].numpy().reshape(B, T, -1)[::8, ::4], gd["ltxy_rows"], atol=1e-4)
    np.testing.assert_allclose(O[1].numpy(), gd["root_rot"], atol=1e-4)
    np.testing.assert_allclose(O[0].numpy(), gd["root_pos"], atol=1e-3)      # integrates 255 steps


def test_oracle_rollout_1800_frames_fp64_vs_reference(golden_dir):
    gd = np.load(golden_dir / "full_rollout.npz")
    _, de, _ = helpers.build_nets()
    T = int(gd["T"])
    W, speech, style = helpers.full_decoder_inputs(helpers.real_stats("v1"), 1, T, int(gd["seed"]))
    helpers.assert_inputs_match(gd, W, speech, style)
    with torch.no_grad():
        O = _rollout(helpers.sd(de, torch.float64), W, speech, style,
                     helpers.real_stats_tensors("v1", torch.float64), torch.float64)
    pose = helpers.pack_pose(*O[2:]).numpy()[0]
    np.testing.assert_allclose(pose[::10], gd["pose_every10"], atol=1e-9)
    np.testing.assert_allclose(O[0].numpy()[0], gd["root_pos"], atol=1e-8)
    np.testing.assert_allclose(O[1].numpy()[0], gd["root_rot"], atol=1e-9)


def test_oracle_style_encoder_len512_vs_reference(golden_dir):
    gd = np.load(golden_dir / "full_style512.npz")
    _, _, st = helpers.build_nets()
    s = helpers.real_stats_tensors("v1")
    stats = helpers.real_stats("v1")
    B, L = int(gd["B"]), int(gd["L"])
    clips = [synth.make_clip_stats(L, seed=int(gd["seed"]) + b, stats=stats) for b in range(B)]
    ex = torch.as_tensor(np.stack([np.concatenate(
        [c["Y_root_vel"], c["Y_root_vrt"], c["Y_lpos"].reshape(L, -1), c["Y_ltxy"].reshape(L, -1),
         c["Y_lvel"].reshape(L, -1), c["Y_lvrt"].reshape(L, -1), np.zeros((L, 3), np.float32)], axis=1) for c in clips]))
    np.testing.assert_allclose(helpers.checksum(ex.numpy()), gd["ex_check"], rtol=1e-9)
    with torch.no_grad():
        z, mu, lv = onets.style_encoder(helpers.sd(st), (ex - s["in_mean"]) / s["in_std"], torch.as_tensor(gd["eps"]))
    np.testing.assert_allclose(mu.numpy(), gd["mu"], atol=1e-5)
    np.testing.assert_allclose(lv.numpy(), gd["logvar"], atol=1e-5)
    np.testing.assert_allclose(z.numpy(), gd["z"], atol=2e-5)


def test_oracle_mel_10s_vs_reference(golden_dir):
    gd = np.load(golden_dir / "full_mel10.npz")
    wav = synth.synth_wav(int(gd["n_samples"]), seed=0).astype(np.float32) / 32768.0
    np.testing.assert_allclose(helpers.checksum(wav), gd["wav_check"], rtol=1e-9)
    assert omel.n_anim_frames(len(wav)) == int(gd["nframes"]) == 600           # integers: bit-exact
    feat = omel.preprocess_audio(wav, int(gd["nframes"]))
    assert feat.shape == gd["feat"].shape
    np.testing.assert_allclose(feat, gd["feat"], atol=2e-6)


def oracle_full_iteration(gd, v, dtype=torch.float32):
    """One training iteration of the oracle on the recorded window indices (shared with the GPU test's fp64 check)."""
    from oracle import dataset as ods
    data = helpers.full_dataset(gd, v)
    s = helpers.real_stats_tensors(v, dtype)
    B, T, Lx = int(gd["B"]), int(gd["window"]), int(gd["example_length"])
    label = "eps" not in gd.files
    style_size = int(gd["nlabels"]) if label else 64
    se, de, st = helpers.build_nets(style_size=style_size)
    nets = [se, de] + ([] if label else [st])
    ws = [helpers.sd(m, dtype) for m in nets]
    for w in ws:
        for t in w.values():
            t.requires_grad_(True)
    starts, samples = ods.build_windows(data["ranges_train"], T)
    idx = gd["idx"]
    r0 = starts[idx]
    f = lambda k: torch.as_tensor(np.stack([np.asarray(data[k])[a:a + T] for a in r0])).to(dtype)  # noqa: E731
    Wt = {k: f(k) for k in ("X_audio_features", "Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos", "Y_ltxy",
                            "Y_lvel", "Y_lvrt", "Y_gaze_pos")}
    speech = onets.speech_encoder(ws[0], (Wt["X_audio_features"] - s["a_mean"]) / s["a_std"])
    mu = lv = None
    if label:
        lab = np.asarray(data["ranges_train_labels"])[samples[idx]]
        z = torch.as_tensor(np.eye(style_size)[lab]).to(dtype)
    else:
        n = len(data["Y_root_vel"])
        rows = np.stack([ods.example_rows(*ods.example_range(int(a), T, *data["ranges_train"][samples[i]], Lx, n), Lx)
                         for a, i in zip(r0, idx)])
        pose = np.concatenate([np.asarray(data[k]).reshape(n, -1) for k in
                               ("Y_root_vel", "Y_root_vrt", "Y_lpos", "Y_ltxy", "Y_lvel", "Y_lvrt")] +
                              [np.zeros((n, 3), np.float32)], axis=1)
        ex = torch.as_tensor(pose[rows]).to(dtype)
        z, mu, lv = onets.style_encoder(ws[2], (ex - s["in_mean"]) / s["in_std"], torch.as_tensor(gd["eps"]).to(dtype))
    O = onets.decoder_rollout(ws[1], Wt["Y_root_pos"][:, 0], Wt["Y_root_rot"][:, 0], Wt["Y_root_vel"][:, 0],
                              Wt["Y_root_vrt"][:, 0], Wt["Y_lpos"][:, 0], Wt["Y_ltxy"][:, 0], Wt["Y_lvel"][:, 0],
                              Wt["Y_lvrt"][:, 0], Wt["Y_gaze_pos"], speech, z.unsqueeze(1).repeat(1, T, 1),
                              s["in_mean"], s["in_std"], s["out_mean"], s["out_std"], synth.DT)
    loss, terms = oloss.training_loss(O, [Wt[k] for k in ("Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos",
                                                          "Y_ltxy", "Y_lvel", "Y_lvrt")], Wt["Y_gaze_pos"],
                                      synth.PARENTS, synth.DT, mu, lv, iteration=0)
    loss.backward()
    return loss, terms, ws


def _check_iteration(gd, v):
    loss, terms, ws = oracle_full_iteration(gd, v)
    np.testing.assert_allclose(float(loss), gd["loss"][0], rtol=5e-6)
    np.testing.assert_allclose(terms.detach().numpy(), gd["terms"][0], rtol=5e-5, atol=1e-7)
    plist = [t for w in ws for t in w.values()]
    off = 0
    for i, p in enumerate(plist):
        idx = helpers.sample_idx(p.numel())
        ref = gd["grad_samples"][off:off + len(idx)]
        # 5e-4 of the TENSOR's largest gradient entry (as the GPU test does): after 255 BPTT steps fp32 accumulation noise is
        # relative to the large entries of a tensor, not to each sampled one
        # (the tensor's own max |g| enters the scale only after its |g| SUM has been pinned to the reference's: an inflated
        #  gradient cannot widen its own tolerance)
        np.testing.assert_allclose(helpers.fingerprint(p.grad)[1], gd["grad_fp"][i][1], rtol=2e-3, err_msg=f"param {i} |g| sum")
        scale = max(1e-7, float(np.abs(ref).max()), float(p.grad.abs().max()))
        np.testing.assert_allclose(p.grad.flatten()[idx].numpy(), ref, atol=5e-4 * scale + 1e-9, err_msg=f"param {i}")
        off += len(idx)
    assert off == len(gd["grad_samples"])


def test_oracle_train_iteration_b32_t256_vs_reference(golden_dir):
    _check_iteration(np.load(golden_dir / "full_train32.npz"), "v1")


def test_oracle_train_iteration_v2_label_b64_vs_reference(golden_dir):
    _check_iteration(np.load(golden_dir / "full_trainv2.npz"), "v2")


def test_oracle_train_iteration_v2_label_b64_t256_vs_reference(golden_dir):
    """configs[3] at the shape bench.py times (B=64 x 256, label conditioning)."""
    _check_iteration(np.load(golden_dir / "full_trainv2_256.npz"), "v2")


def test_oracle_style_encoder_7200_frame_exemplar_vs_reference(golden_dir):
    """configs[4]: a whole 2-minute exemplar through the attention block (ZEGGS/generate.py:190-262)."""
    gd = np.load(golden_dir / "full_style7200.npz")
    _, _, st = helpers.build_nets()
    s = helpers.real_stats_tensors("v1")
    ex = torch.as_tensor(helpers.exemplar_rows(helpers.real_stats("v1"), int(gd["L"]), int(gd["seed"]))[None])
    np.testing.assert_allclose(helpers.checksum(ex.numpy()), gd["ex_check"], rtol=1e-9)
    with torch.no_grad():
        z, mu, lv = onets.style_encoder(helpers.sd(st), (ex - s["in_mean"]) / s["in_std"], torch.as_tensor(gd["eps"]))
    np.testing.assert_allclose(mu.numpy(), gd["mu"], atol=1e-5)
    np.testing.assert_allclose(lv.numpy(), gd["logvar"], atol=1e-5)
    np.testing.assert_allclose(z.numpy(), gd["z"], atol=2e-5)


def test_oracle_speech_encoder_with_the_shipped_trained_weights(golden_dir):
    """The reference's one shipped trained artefact (data/outputs/v1/saved_models/speech_encoder.pt) on the 10 s clip."""
    gd = np.load(golden_dir / "full_speech_trained.npz")
    s = helpers.real_stats_tensors("v1")
    feat = torch.as_tensor(np.load(golden_dir / "full_mel10.npz")["feat"])[None]
    x = (feat - s["a_mean"]) / s["a_std"]
    np.testing.assert_allclose(helpers.checksum(x.numpy()), gd["x_check"], rtol=1e-9)
    w = {k[2:]: torch.as_tensor(gd[k]) for k in gd.files if k.startswith("w.")}
    assert sum(v.numel() for v in w.values()) == 136448
    with torch.no_grad():
        y = onets.speech_encoder(w, x)
        y64 = onets.speech_encoder({k: v.double() for k, v in w.items()}, x.double())
    np.testing.assert_allclose(y.numpy(), gd["out"], atol=2e-5)
    np.testing.assert_allclose(y64.numpy(), gd["out64"], atol=1e-9)


def test_oracle_long_rollout_prefix_vs_reference_fp64(golden_dir):
    """configs[4] (108 000 free-running frames, recorded once from the reference in fp64): the oracle reproduces the first
    stored samples (frames 500 and 1000) -- the full run is ~10 CPU-minutes and is the GPU test's yardstick."""
    gd = np.load(golden_dir / "full_rollout108k.npz")
    _, de, _ = helpers.build_nets()
    T = int(gd["T"])
    W, speech, style = helpers.long_decoder_inputs(helpers.real_stats("v1"), T, int(gd["seed"]))
    helpers.assert_inputs_match(gd, W, speech, style)
    n = 1001
    Wn = dict(W, Y_gaze_pos=W["Y_gaze_pos"][:, :n])
    with torch.no_grad():
        O = _rollout(helpers.sd(de, torch.float64), Wn, speech[:, :n], style[:, :n],
                     helpers.real_stats_tensors("v1", torch.float64), torch.float64)
    pose = helpers.pack_pose(*O[2:]).numpy()[0]
    np.testing.assert_allclose(pose[::500], gd["pose_every500"][:3], atol=1e-9)
    np.testing.assert_allclose(O[0].numpy()[0][::100], gd["root_pos_every100"][:11], atol=1e-8)
    np.testing.assert_allclose(O[1].numpy()[0][::100], gd["root_rot_every100"][:11], atol=1e-9)
