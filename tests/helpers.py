"""Shared test helpers: seeded nets (reference construction order) and golden loading."""
import numpy as np
import torch

from zeggs import modules, synth

SEED = 1234


def build_nets(style_size=64, use_vae=True, seed=SEED):
    """torch.manual_seed(seed); SpeechEncoder, Decoder, StyleEncoder -- the
    reference's construction order (train.py:118-139), so weights are
    bit-identical to the reference's random init."""
    torch.manual_seed(seed)
    se = modules.SpeechEncoder(synth.N_AUDIO, 64, 64)
    de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, 64, style_size, 1024, 2)
    st = modules.StyleEncoder(synth.POSE_IN, 512, 64, type="attn", use_vae=use_vae)
    return se, de, st


def fingerprint(t):
    a = t.detach().double().flatten()
    return np.array([float(a.sum()), float(a.abs().sum())])


def sample_idx(numel):
    return np.unique((np.arange(97, dtype=np.int64) * 7919 + 13) % numel)


def sd(module, dtype=None):
    return {k: (v.detach().to(dtype) if dtype else v.detach()) for k, v in module.state_dict().items()}


def stats_tensors(dtype=torch.float32, device="cpu"):
    s = synth.make_stats()
    t = lambda k: torch.as_tensor(np.asarray(s[k]), dtype=dtype, device=device)  # noqa: E731
    return dict(a_mean=t("audio_input_mean"), a_std=t("audio_input_std"), in_mean=t("anim_input_mean"),
                in_std=t("anim_input_std"), out_mean=t("anim_output_mean"), out_std=t("anim_output_std"))


# ----------------------------------------------------------------------------- full-shape fixtures (oracle/make_golden_full.py)
GOLDEN = __import__("pathlib").Path(__file__).resolve().parent / "golden"
STAT_KEYS = ("audio_input_mean", "audio_input_std", "anim_input_mean", "anim_input_std", "anim_output_mean",
             "anim_output_std")


def real_stats(v="v1"):
    """The reference's own normalisation statistics (data/processed_v*/stats.npz), stored as a fixture."""
    s = np.load(GOLDEN / f"real_stats_{v}.npz")
    return {k: np.asarray(s[k]) for k in STAT_KEYS}


def real_stats_tensors(v="v1", dtype=torch.float32, device="cpu"):
    s = real_stats(v)
    t = lambda k: torch.as_tensor(np.asarray(s[k]), dtype=dtype, device=device)  # noqa: E731
    return dict(a_mean=t("audio_input_mean"), a_std=t("audio_input_std"), in_mean=t("anim_input_mean"),
                in_std=t("anim_input_std"), out_mean=t("anim_output_mean"), out_std=t("anim_output_std"))


def checksum(a):
    a = np.asarray(a, np.float64).ravel()
    return np.array([a.sum(), np.abs(a).sum()])


def full_decoder_inputs(st, B, T, seed):
    """The seeded decoder inputs of oracle/make_golden_full.py:decoder_inputs (same generator, same seeds)."""
    clips = [synth.make_clip_stats(T, seed=seed + b, stats=st) for b in range(B)]
    W = {k: torch.as_tensor(np.stack([c[k] for c in clips])) for k in clips[0]}
    rng = np.random.default_rng(seed + 7)
    speech = torch.as_tensor(rng.standard_normal((B, T, 64)).astype(np.float32) * 0.5)
    style = torch.as_tensor(np.repeat(rng.standard_normal((B, 1, 64)).astype(np.float32) * 0.5, T, axis=1))
    return W, speech, style


def assert_inputs_match(gd, W, speech, style):
    got = np.stack([checksum(W[k]) for k in sorted(W)] + [checksum(speech), checksum(style)])
    np.testing.assert_allclose(got, gd["in_check"], rtol=1e-9, err_msg="synthetic input generator drifted")


def pack_pose(vel, vrt, lpos, ltxy, lvel, lvrt):
    B, T = vel.shape[:2]
    return torch.cat([vel.reshape(B, T, -1), vrt.reshape(B, T, -1), lpos.reshape(B, T, -1), ltxy.reshape(B, T, -1),
                      lvel.reshape(B, T, -1), lvrt.reshape(B, T, -1)], dim=-1)


def full_dataset(gd, v):
    """The synthetic processed_data (dict) that oracle/make_golden_full.py:record_train_iteration trained on."""
    st = real_stats(v)
    return synth.make_processed(int(gd["n_train"]), 1, int(gd["nframes"]), int(gd["data_seed"]), int(gd["nlabels"]), st,
                                synth.make_clip_stats)


def long_decoder_inputs(st, T, seed):
    """The B=1 inputs of oracle/make_golden_full.py:long_decoder_inputs (a T-frame free-running decode: only the first
    pose, the gaze target and the speech / style encodings exist)."""
    c = synth.make_clip_stats(8, seed=seed, stats=st)
    W = {k: torch.as_tensor(v[None, :1]) for k, v in c.items() if k != "Y_gaze_pos"}
    rng = np.random.default_rng(seed + 7)
    gaze = np.array([[10.0, 150.0, 100.0]]) + synth._smooth(rng, T, 3, 2.0, k=241)
    W["Y_gaze_pos"] = torch.as_tensor(gaze[None].astype(np.float32))
    env = 0.5 + 0.25 * synth._smooth(rng, T, 1, 1.0, k=121)
    speech = torch.as_tensor((rng.standard_normal((1, T, 64)) * env[None]).astype(np.float32))
    style = torch.as_tensor(np.repeat(rng.standard_normal((1, 1, 64)).astype(np.float32) * 0.5, T, axis=1))
    return W, speech, style


def exemplar_rows(st, L, seed):
    """[L, 1134] un-normalised style-exemplar rows of a seeded clip (gaze slot zero, dataset.py:194)."""
    c = synth.make_clip_stats(L, seed=seed, stats=st)
    return np.concatenate([c["Y_root_vel"], c["Y_root_vrt"], c["Y_lpos"].reshape(L, -1), c["Y_ltxy"].reshape(L, -1),
                           c["Y_lvel"].reshape(L, -1), c["Y_lvrt"].reshape(L, -1), np.zeros((L, 3), np.float32)], axis=1)


def variants_batch_inputs(gd):
    """Inputs of tests/golden/variants_batch.npz that the fixture does not store (oracle/make_golden.py gold_variants_batch):
    first pose, gaze targets and exemplar rebuilt from the same deterministic synth clips, checked against the stored checksums;
    the weights of the differentiated scalar from the stored seed."""
    T, L = (int(v) for v in gd["clip_T_L"])
    B = gd["in_speech"].shape[0]
    stats = synth.make_stats()
    clips = [synth.make_clip(T + L, seed=170 + b, stats=stats) for b in range(B)]
    W = {k: torch.as_tensor(np.stack([c[k][:T] for c in clips])) for k in clips[0]}
    ex = []
    for c in clips:
        ex.append(np.concatenate([c["Y_root_vel"][T:T + L], c["Y_root_vrt"][T:T + L],
                                  c["Y_lpos"][T:T + L].reshape(L, -1), c["Y_ltxy"][T:T + L].reshape(L, -1),
                                  c["Y_lvel"][T:T + L].reshape(L, -1), c["Y_lvrt"][T:T + L].reshape(L, -1),
                                  np.zeros((L, 3), np.float32)], axis=1))
    example = torch.as_tensor(np.stack(ex))
    np.testing.assert_allclose(fingerprint(example), gd["sum_example"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(fingerprint(W["Y_gaze_pos"]), gd["sum_gaze"], rtol=1e-12, atol=0)
    first = [W[k][:, 0] for k in ("Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos", "Y_ltxy", "Y_lvel", "Y_lvrt")]
    names = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "ltxy", "lvel", "lvrt")
    gen = torch.Generator().manual_seed(int(gd["weight_seed"]))
    wts = [torch.randn(tuple(gd["O_" + n].shape), generator=gen) for n in names]
    wz, wm, wl = (torch.randn(B, 64, generator=gen) for _ in range(3))
    return first, W["Y_gaze_pos"], example, wts, (wz, wm, wl)


def assert_grad_samples(gd, tag, named_grads, tol):
    """parameter gradients against the reference's stored samples: |g - g_ref| <= tol * max |g_ref| at the sampled indices"""
    for k, gr in named_grads:
        idx = torch.as_tensor(gd[f"gidx_{tag}.{k}"])
        ref = torch.as_tensor(gd[f"gsamp_{tag}.{k}"]).double()
        got = gr.detach().cpu().flatten()[idx].double()
        scale = max(float(gd[f"gmax_{tag}.{k}"]), 1e-12)
        err = float((got - ref).abs().max()) / scale
        assert err < tol, (tag, k, err)


def width512_inputs(gd):
    """inputs of tests/golden/width512.npz (oracle/make_golden.py gold_width512) rebuilt from the deterministic synth clips"""
    T = int(gd["clip_T"])
    B = gd["in_speech"].shape[0]
    stats = synth.make_stats()
    clips = [synth.make_clip(T, seed=270 + b, stats=stats) for b in range(B)]
    W = {k: torch.as_tensor(np.stack([c[k][:T] for c in clips])) for k in clips[0]}
    np.testing.assert_allclose(fingerprint(W["Y_gaze_pos"]), gd["sum_gaze"], rtol=1e-12, atol=0)
    first = [W[k][:, 0] for k in ("Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos", "Y_ltxy", "Y_lvel", "Y_lvrt")]
    names = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "ltxy", "lvel", "lvrt")
    gen = torch.Generator().manual_seed(int(gd["weight_seed"]))
    wts = [torch.randn(tuple(gd["O_" + n].shape), generator=gen) for n in names]
    return first, W["Y_gaze_pos"], wts


# ----------------------------------------------------------------------------- iteration 1 of train_iter.npz (round 5)
POSE_SPLIT = (3, 3, 225, 450, 225, 225)       # root_vel, root_vrt, lpos, ltxy, lvel, lvrt in the packed pose row


def unpack_pose(pose):
    """[B, T, 1131] packed decoder output -> (root_vel, root_vrt, lpos, ltxy, lvel, lvrt) in the reference's shapes."""
    B, T = pose.shape[:2]
    vel, vrt, lpos, ltxy, lvel, lvrt = torch.split(pose, POSE_SPLIT, dim=-1)
    return vel, vrt, lpos.reshape(B, T, 75, 3), ltxy.reshape(B, T, 75, 2, 3), lvel.reshape(B, T, 75, 3), lvrt.reshape(B, T, 75, 3)


def grads_at_forward_point(g, it, state_dicts, O_point, kl_iteration=None):
    """The float64 arbiter for an iteration whose loss gradient is ill-conditioned in the FORWARD POINT (train_iter.npz,
    iteration 1: one root joint's predicted x / y axes are 0.86 degrees from antiparallel, so d loss / d output ~ 1 / |x cross y| and
    an output deviation of 5e-6 -- float32 forward rounding -- moves the length of the WHOLE gradient by 0.5 %: measured on the
    reference itself, oracle/make_golden.py: gold_train_iter_perturb and DESIGN.md section 4).  An fp32 implementation cannot be
    held to the fp64 gradient at the fp64 forward point closer than the reference's own fp32 run is; it CAN be held to

        G* = J64(theta)^T . grad_O loss64(O_point)          (+ the KL path through mu, logvar)

    i.e. the float64 network Jacobian applied to the float64 loss gradient evaluated AT THE IMPLEMENTATION'S OWN OUTPUTS O_point.
    state_dicts: (speech, decoder, style) float32 state dicts = the weights the implementation ran the iteration with;
    O_point: its 8 decoder outputs (root_pos, root_rot, root_vel, root_vrt, lpos, ltxy, lvel, lvrt), any float dtype.
    Returns (list of float64 gradient tensors in the reference optimizer's parameter order, float64 outputs of the oracle,
    float64 loss gradient w.r.t. the outputs at O_point)."""
    from oracle import loss as oloss
    from oracle import nets as onets
    from zeggs import synth
    dt = torch.float64
    s = {k: v.to(dt) for k, v in stats_tensors().items()}
    ws = [{k: v.detach().cpu().to(dt).clone().requires_grad_(True) for k, v in w.items()} for w in state_dicts]
    b = [torch.as_tensor(g[f"it{it}_batch{j}"]).to(dt) for j in range(11)]
    audio, rpos, rrot, rvel, rvrt, lpos, ltxy, lvel, lvrt, gaze, wstyle = b
    speech = onets.speech_encoder(ws[0], (audio - s["a_mean"]) / s["a_std"])
    z, mu, logvar = onets.style_encoder(ws[2], (wstyle - s["in_mean"]) / s["in_std"], torch.as_tensor(g[f"it{it}_eps"]).to(dt))
    T = audio.shape[1]
    O64 = onets.decoder_rollout(ws[1], rpos[:, 0], rrot[:, 0], rvel[:, 0], rvrt[:, 0], lpos[:, 0], ltxy[:, 0], lvel[:, 0],
                                lvrt[:, 0], gaze, speech, z.unsqueeze(1).repeat(1, T, 1), s["in_mean"], s["in_std"],
                                s["out_mean"], s["out_std"], synth.DT)
    Oe = [o.detach().cpu().to(dt).reshape(r.shape).clone().requires_grad_(True) for o, r in zip(O_point, O64)]
    loss_e, _ = oloss.training_loss(Oe, (rpos, rrot, rvel, rvrt, lpos, ltxy, lvel, lvrt), gaze, synth.PARENTS, synth.DT, mu,
                                    logvar, iteration=it if kl_iteration is None else kl_iteration)
    g_e = torch.autograd.grad(loss_e, Oe, retain_graph=True)
    (loss_e + sum((o * ge.detach()).sum() for o, ge in zip(O64, g_e))).backward()
    grads = [v.grad if v.grad is not None else torch.zeros_like(v) for w in (ws[0], ws[1], ws[2]) for v in w.values()]
    return grads, [o.detach() for o in O64], [x.detach() for x in g_e]
