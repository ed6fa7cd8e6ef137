"""Generate tests/golden/*.npz by running the UNMODIFIED reference in this container.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Run from the repo root:

    python -m oracle.make_golden

Needs /root/reference (present only in the build container); the produced
fixtures are small, committed, and are what the GPU box / CI read.  Networks are
the full-size configs_v1 nets, random-init with torch.manual_seed(1234) in the
reference's construction order (train.py:118-139); the fixtures store weight
fingerprints instead of the 25.5 M weights, the tests re-create the weights
from the same seed and verify the fingerprints first.
"""
import json
import random
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ubisoft-laforge-zeroeggs_amd"))
from oracle import ref_shims  # noqa: E402
from zeggs import synth  # noqa: E402

GOLD = ROOT / "tests" / "golden"
SEED = 1234
NET_OPT = {
    "decoder": {"nhidden": 1024, "num_rnn_layers": 2, "rnn_cond": "normal"},
    "speech_encoder": {"nhidden": 64, "speech_encoding_size": 64},
    "style_encoder": {"nhidden": 512, "style_encoding_size": 64, "example_length": 12,
                      "type": "attn", "use_vae": True},
}


def sample_idx(numel):
    """Fixed pseudo-random sample of flat indices used for gradient/weight fingerprints."""
    return np.unique((np.arange(97, dtype=np.int64) * 7919 + 13) % numel)


def fingerprint(t):
    a = t.detach().double().flatten()
    return np.array([float(a.sum()), float(a.abs().sum())])


def build_ref_nets(ref):
    """Reference construction order and seed (train.py:40-41,118-139)."""
    torch.manual_seed(SEED)
    se = ref.modules.SpeechEncoder(synth.N_AUDIO, 64, 64)
    de = ref.modules.Decoder(pose_input_size=synth.POSE_IN, pose_output_size=synth.POSE_OUT,
                             speech_encoding_size=64, style_encoding_size=64, hidden_size=1024,
                             num_rnn_layers=2)
    st = ref.modules.StyleEncoder(synth.POSE_IN, 512, 64, type="attn", use_vae=True)
    return se, de, st


def gold_nets(ref):
    se, de, st = build_ref_nets(ref)
    se.eval(), de.eval(), st.eval()
    stats = synth.make_stats()
    B, T, L = 2, 6, 10
    clips = [synth.make_clip(T + L, seed=50 + b, stats=stats) for b in range(B)]
    tt = lambda k, sl: torch.as_tensor(np.stack([c[k][sl] for c in clips]))  # noqa: E731
    W = {k: tt(k, slice(0, T)) for k in clips[0]}
    in_mean, in_std = torch.as_tensor(stats["anim_input_mean"]), torch.as_tensor(stats["anim_input_std"])
    out_mean, out_std = torch.as_tensor(stats["anim_output_mean"]), torch.as_tensor(stats["anim_output_std"])
    a_mean, a_std = torch.as_tensor(stats["audio_input_mean"]), torch.as_tensor(stats["audio_input_std"])
    ex = []
    for c in clips:
        n = L
        ex.append(np.concatenate([c["Y_root_vel"][T:T + n], c["Y_root_vrt"][T:T + n],
                                  c["Y_lpos"][T:T + n].reshape(n, -1), c["Y_ltxy"][T:T + n].reshape(n, -1),
                                  c["Y_lvel"][T:T + n].reshape(n, -1), c["Y_lvrt"][T:T + n].reshape(n, -1),
                                  np.zeros((n, 3), np.float32)], axis=1))
    example = torch.as_tensor(np.stack(ex))
    eps = torch.as_tensor(np.random.default_rng(3).standard_normal((B, 64)).astype(np.float32))
    parents = torch.LongTensor(synth.PARENTS)
    out = {}
    with torch.no_grad():
        audio_n = (W["X_audio_features"] - a_mean) / a_std
        speech = se(audio_n)
        orig = torch.randn_like
        torch.randn_like = lambda x, *a, **k: eps.to(x.dtype)
        try:
            z, mu, logvar = st((example - in_mean) / in_std, 1.3)
        finally:
            torch.randn_like = orig
        O = de(W["Y_root_pos"][:, 0], W["Y_root_rot"][:, 0], W["Y_root_vel"][:, 0], W["Y_root_vrt"][:, 0],
               W["Y_lpos"][:, 0], W["Y_ltxy"][:, 0], W["Y_lvel"][:, 0], W["Y_lvrt"][:, 0],
               W["Y_gaze_pos"], speech, z.unsqueeze(1).repeat(1, T, 1), parents,
               in_mean, in_std, out_mean, out_std, synth.DT)
    out.update({"in_" + k: v.numpy() for k, v in W.items()})
    out.update(in_example=example.numpy(), in_eps=eps.numpy(), temperature=np.float32(1.3),
               speech=speech.numpy(), style_z=z.numpy(), style_mu=mu.numpy(), style_logvar=logvar.numpy())
    names = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "ltxy", "lvel", "lvrt")
    out.update({"O_" + n: o.numpy() for n, o in zip(names, O)})
    for tag, net in (("speech", se), ("decoder", de), ("style", st)):
        for k, v in net.state_dict().items():
            out[f"fp_{tag}.{k}"] = fingerprint(v)
    np.savez_compressed(GOLD / "nets.npz", **out)
    print("nets.npz", {k: v.shape for k, v in out.items() if k.startswith("O_")})


def gold_variants(ref):
    """Option-surface variants of the nets on a tiny case (full-size layers): rnn_cond="film" decoder rollout and
    style_encoder type="gru" (+VAE), reference forward outputs; weights seeded 4321 in this construction order."""
    torch.manual_seed(4321)
    de = ref.modules.Decoder(pose_input_size=synth.POSE_IN, pose_output_size=synth.POSE_OUT, speech_encoding_size=64,
                             style_encoding_size=64, hidden_size=1024, num_rnn_layers=2, rnn_cond="film")
    st = ref.modules.StyleEncoder(synth.POSE_IN, 512, 64, type="gru", use_vae=True)
    de.eval(), st.eval()
    stats = synth.make_stats()
    B, T, L = 2, 5, 9
    clips = [synth.make_clip(T + L, seed=70 + b, stats=stats) for b in range(B)]
    W = {k: torch.as_tensor(np.stack([c[k][:T] for c in clips])) for k in clips[0]}
    in_mean, in_std = torch.as_tensor(stats["anim_input_mean"]), torch.as_tensor(stats["anim_input_std"])
    out_mean, out_std = torch.as_tensor(stats["anim_output_mean"]), torch.as_tensor(stats["anim_output_std"])
    ex = []
    for c in clips:
        ex.append(np.concatenate([c["Y_root_vel"][T:T + L], c["Y_root_vrt"][T:T + L],
                                  c["Y_lpos"][T:T + L].reshape(L, -1), c["Y_ltxy"][T:T + L].reshape(L, -1),
                                  c["Y_lvel"][T:T + L].reshape(L, -1), c["Y_lvrt"][T:T + L].reshape(L, -1),
                                  np.zeros((L, 3), np.float32)], axis=1))
    example = torch.as_tensor(np.stack(ex))
    rng = np.random.default_rng(8)
    eps = torch.as_tensor(rng.standard_normal((B, 64)).astype(np.float32))
    speech = torch.as_tensor(rng.standard_normal((B, T, 64)).astype(np.float32) * 0.5)
    style = torch.as_tensor(rng.standard_normal((B, T, 64)).astype(np.float32) * 0.5)
    with torch.no_grad():
        orig = torch.randn_like
        torch.randn_like = lambda x, *a, **k: eps.to(x.dtype)
        try:
            z, mu, logvar = st((example - in_mean) / in_std, 1.0)
        finally:
            torch.randn_like = orig
        O = de(W["Y_root_pos"][:, 0], W["Y_root_rot"][:, 0], W["Y_root_vel"][:, 0], W["Y_root_vrt"][:, 0],
               W["Y_lpos"][:, 0], W["Y_ltxy"][:, 0], W["Y_lvel"][:, 0], W["Y_lvrt"][:, 0], W["Y_gaze_pos"], speech,
               style, torch.LongTensor(synth.PARENTS), in_mean, in_std, out_mean, out_std, synth.DT)
    out = {"in_" + k: v.numpy() for k, v in W.items()}
    out.update(in_example=example.numpy(), in_eps=eps.numpy(), in_speech=speech.numpy(), in_style=style.numpy(),
               gru_z=z.numpy(), gru_mu=mu.numpy(), gru_logvar=logvar.numpy())
    names = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "ltxy", "lvel", "lvrt")
    out.update({"O_" + n: o.numpy() for n, o in zip(names, O)})
    for tag, net in (("decoder", de), ("style", st)):
        for k, v in net.state_dict().items():
            out[f"fp_{tag}.{k}"] = fingerprint(v)
    np.savez_compressed(GOLD / "variants.npz", **out)
    print("variants.npz", out["O_lpos"].shape, out["gru_z"].shape)


def gold_train_iter(ref):
    """Two full reference train() iterations, B=2, window=8 (dropout patched to
    identity, VAE eps injected) -> batches, losses, gradient / weight samples."""
    import torch.nn.functional as F
    tmp = Path(tempfile.mkdtemp(prefix="zeggs_gold_"))
    window, B = 8, 2
    npz, jsn = synth.write_dataset(tmp / "data", n_train=1, n_valid=1, nframes=window + 4, seed=5)
    rec = dict(batches=[], eps=[], loss=[], terms=[], grads=[], weights=[])

    class RecDL(torch.utils.data.DataLoader):
        def __iter__(self):
            for b in super().__iter__():
                rec["batches"].append([t.clone() for t in b])
                yield b

    class RecWriter:
        def __init__(self, *a, **k): pass
        def add_hparams(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_scalars(self, tag, d, it):
            rec["terms"].append([float(v) for v in d.values()])

    eps_rng = np.random.default_rng(11)

    def fake_randn_like(x, *a, **k):
        e = torch.as_tensor(eps_rng.standard_normal(tuple(x.shape)).astype(np.float32))
        if x.shape[0] == B:
            rec["eps"].append(e.clone())
        return e

    orig_step = ref.optimizers.RAdam.step

    def rec_step(self, closure=None):
        ps = [p for g in self.param_groups for p in g["params"]]
        rec["grads"].append([(fingerprint(p.grad), p.grad.flatten()[sample_idx(p.numel())].clone()) for p in ps])
        r = orig_step(self, closure)
        rec["weights"].append([p.detach().flatten()[sample_idx(p.numel())].clone() for p in ps])
        return r

    orig_backward = torch.Tensor.backward

    def rec_backward(self, *a, **k):
        rec["loss"].append(float(self.detach()))
        return orig_backward(self, *a, **k)

    saved = (ref.train.DataLoader, ref.train.SummaryWriter, torch.randn_like, F.dropout)
    ref.train.DataLoader, ref.train.SummaryWriter = RecDL, RecWriter
    torch.randn_like = fake_randn_like
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x
    ref.optimizers.RAdam.step = rec_step
    ref.train.RAdam.step = rec_step
    torch.Tensor.backward = rec_backward
    random.seed(0)
    train_opt = dict(niterations=0.001, batchsize=B, window=window, change_pace=True, learning_rate=1e-4,
                     learning_rate_decay=0.995, eps=1e-5, resume=False, use_gpu=False, thread_count=1,
                     seed=SEED, use_tensorboard=True, style_encoding_type="example",
                     generate_samples_step=10 ** 9, use_script=False)
    try:
        (tmp / "models").mkdir()
        (tmp / "logs").mkdir()
        ref.train.train(tmp / "models", tmp / "logs", npz, jsn, train_opt, NET_OPT)
    finally:
        ref.train.DataLoader, ref.train.SummaryWriter, torch.randn_like, F.dropout = saved
        ref.optimizers.RAdam.step = orig_step
        torch.Tensor.backward = orig_backward
    out = dict(window=np.int64(window), batch=np.int64(B), loss=np.array(rec["loss"]),
               terms=np.array(rec["terms"]))
    data = np.load(npz)
    out.update({"data_" + k: data[k] for k in data.files})
    for it, b in enumerate(rec["batches"]):
        for j, t in enumerate(b):
            out[f"it{it}_batch{j}"] = t.numpy()
        out[f"it{it}_eps"] = rec["eps"][it].numpy()
        out[f"it{it}_grad_fp"] = np.stack([g[0] for g in rec["grads"][it]])
        out[f"it{it}_grad_samples"] = np.concatenate([g[1].numpy() for g in rec["grads"][it]])
        out[f"it{it}_weight_samples"] = np.concatenate([w.numpy() for w in rec["weights"][it]])
    np.savez_compressed(GOLD / "train_iter.npz", **out)
    print("train_iter.npz: iterations", len(rec["loss"]), "loss", rec["loss"])


def gold_mel(ref):
    conf = json.load(open("/root/reference/data/processed_v1/data_pipeline_conf.json"))
    conf["audio_conf"]["normalize_loudness"] = False      # pyloudnorm absent: parity unpinned there
    ac = ref.DictConfig(conf["audio_conf"])
    out = {}
    for tag, n in (("a", 16000), ("b", 16123), ("c", 24400)):
        wav = synth.synth_wav(n, seed=n).astype(np.float32) / 32768.0
        nfr = int(round(60.0 * (n / 16000)))
        feat = ref.data_pipeline.preprocess_audio(wav, 60, nfr, ac, feature_type=conf["audio_feature_type"])
        mel, _ = ref.data_pipeline.extract_mel_spectrogram_for_tts(
            wav_signal=wav, fs=16000, n_fft=800, step_size=200, n_mels=80, mel_fmin=20, mel_fmax=7600,
            min_amplitude=1e-5, pre_emphasis=False, pre_emph_coeff=0.97, dynamic_range=None,
            real_amplitude=True, centered=True, normalize_mel_bins=True, normalize_range=True, logger=None)
        out[f"{tag}_wav"], out[f"{tag}_feat"], out[f"{tag}_mel"] = wav, feat, mel
        out[f"{tag}_nframes"] = np.int64(nfr)
    np.savez_compressed(GOLD / "mel.npz", **out)
    print("mel.npz", {k: v.shape for k, v in out.items() if k.endswith("feat")})
    # audio_conf.normalize_mel_bins = false (spectrograms.py:431-440: the filterbank's rows keep unit peaks), same signals
    conf["audio_conf"]["normalize_mel_bins"] = False
    ac = ref.DictConfig(conf["audio_conf"])
    out2 = {}
    for tag in "ab":
        wav, nfr = out[f"{tag}_wav"], int(out[f"{tag}_nframes"])
        out2[f"{tag}_wav"], out2[f"{tag}_nframes"] = wav, np.int64(nfr)
        out2[f"{tag}_feat"] = ref.data_pipeline.preprocess_audio(wav, 60, nfr, ac, feature_type=conf["audio_feature_type"])
    np.savez_compressed(GOLD / "mel_nonorm.npz", **out2)
    print("mel_nonorm.npz")
    # audio_conf.pre_emphasis = true (spectrograms.py:35) and audio_conf.real_amplitude = false (:266-267, :81-88), one at a time and together
    out3 = {}
    for name, pe, ra in (("pre", True, True), ("raw", False, False), ("preraw", True, False)):
        conf["audio_conf"].update(normalize_mel_bins=True, pre_emphasis=pe, real_amplitude=ra)
        ac = ref.DictConfig(conf["audio_conf"])
        for tag in "ab":
            wav, nfr = out[f"{tag}_wav"], int(out[f"{tag}_nframes"])
            out3[f"{tag}_wav"], out3[f"{tag}_nframes"] = wav, np.int64(nfr)
            out3[f"{tag}_feat_{name}"] = ref.data_pipeline.preprocess_audio(wav, 60, nfr, ac, feature_type=conf["audio_feature_type"])
    # audio_conf.centered = false (no reflect padding: other frame count, spectrograms.py:237-245) and normalize_range = false (:123-129)
    for name, ce, nr in (("unc", False, True), ("rawdb", True, False), ("uncrawdb", False, False)):
        conf["audio_conf"].update(normalize_mel_bins=True, pre_emphasis=False, real_amplitude=True, centered=ce, normalize_range=nr)
        ac = ref.DictConfig(conf["audio_conf"])
        for tag in "ab":
            wav, nfr = out[f"{tag}_wav"], int(out[f"{tag}_nframes"])
            out3[f"{tag}_feat_{name}"] = ref.data_pipeline.preprocess_audio(wav, 60, nfr, ac, feature_type=conf["audio_feature_type"])
    out3["pre_emph_coeff"] = np.float64(conf["audio_conf"]["pre_emph_coeff"])
    np.savez_compressed(GOLD / "mel_options.npz", **out3)
    print("mel_options.npz")
    # audio_conf.resample_method = "nearest" / "cubic" (data_pipeline.py:65-79: griddata / interp1d kinds), centered and not (the
    # uncentered table is shorter than the animation: NaN rows / extrapolated energy at the end)
    out4 = {}
    for name, rm, ce in (("nearest", "nearest", True), ("cubic", "cubic", True), ("unc_nearest", "nearest", False), ("unc_cubic", "cubic", False)):
        conf["audio_conf"].update(normalize_mel_bins=True, pre_emphasis=False, real_amplitude=True, centered=ce, normalize_range=True,
                                  resample_method=rm)
        ac = ref.DictConfig(conf["audio_conf"])
        for tag in "abc":
            wav, nfr = out[f"{tag}_wav"], int(out[f"{tag}_nframes"])
            out4[f"{tag}_wav"], out4[f"{tag}_nframes"] = wav, np.int64(nfr)
            out4[f"{tag}_feat_{name}"] = ref.data_pipeline.preprocess_audio(wav, 60, nfr, ac, feature_type=conf["audio_feature_type"])
    np.savez_compressed(GOLD / "mel_resample.npz", **out4)
    print("mel_resample.npz", {k: int(np.isnan(v).sum()) for k, v in out4.items() if "feat" in k})


def gold_anim_orders(ref):
    """BVH channel orders other than the rigs' "zyx": preprocess_animation of the reference on clips whose rotation channels are
    declared (and meant) in other orders -- quat.from_euler takes any (anim/quat.py:154-163) -- and the euler channels utils.write_bvh /
    quat.to_euler produce for order "xzy" (the second order quat.to_euler implements, anim/quat.py:120-125)."""
    out = {}
    names16 = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "lrot", "ltxy", "lvel", "lvrt", "cpos", "crot", "ctxy", "cvel",
               "cvrt", "gaze_pos", "gaze_dir")
    for order in ("xyz", "yzx", "xzy"):
        clip = synth.make_bvh_clip(24, seed=31)
        clip["order"] = order
        feats = ref.data_pipeline.preprocess_animation(dict(clip))
        for n, v in zip(names16, feats):
            out[f"{order}_{n}"] = np.asarray(v)
    # write side: a short synthetic decoder output -> the reference's euler channels in order "xzy"
    rng = np.random.default_rng(7)
    T, J = 12, 75
    lrot = rng.standard_normal((T, J, 4))
    lrot /= np.linalg.norm(lrot, axis=-1, keepdims=True)
    out["w_lrot"] = lrot
    out["w_euler_xzy"] = np.degrees(ref.quat.to_euler(lrot, order="xzy"))
    out["w_euler_zyx"] = np.degrees(ref.quat.to_euler(lrot, order="zyx"))
    np.savez_compressed(GOLD / "anim_orders.npz", **out)
    print("anim_orders.npz")


def gold_dataset(ref):
    """Window table and style-example row sources from the reference SGDataset;
    Y_root_vel[f] = (f, f, f) encodes the source frame of every returned row."""
    tmp = Path(tempfile.mkdtemp(prefix="zeggs_gold_ds_"))
    npz, jsn = synth.write_dataset(tmp, n_train=3, n_valid=1, nframes=40, seed=2)
    d = dict(np.load(npz))
    n = len(d["Y_root_vel"])
    d["Y_root_vel"] = np.repeat(np.arange(n, dtype=np.float32)[:, None], 3, axis=1)
    np.savez(npz, **d)
    window = 8
    out = dict(ranges_train=d["ranges_train"], window=np.int64(window), n_total=np.int64(n))
    ds = ref.dataset.SGDataset(jsn, npz, window, "example", 12)
    out["R0"] = ds.R[:, 0].numpy()
    out["S"] = ds.S.numpy()
    q = []
    for ex_len in (8, 12, 20, 30):
        ds.example_window_length = ex_len
        for idx in (0, 1, 15, 31, 32, 40, 63, len(ds) - 1):
            item = ds[idx]
            rows = item[10][:, 0].numpy().astype(np.int64)
            rows = np.pad(rows, (0, 30 - len(rows)), constant_values=-1)
            q.append(np.concatenate([[ex_len, idx, item[10].shape[0]], rows]))
    out["queries"] = np.array(q, dtype=np.int64)
    out["split_10_3"] = np.array(ref.helpers.split_by_ratio(10, [0.5, 0.25, 0.25]))
    out["split_601"] = np.array(ref.helpers.split_by_ratio(601, [0.3, 0.7]))
    np.savez_compressed(GOLD / "dataset.npz", **out)
    print("dataset.npz windows", len(ds))


def gold_radam(ref):
    torch.manual_seed(3)
    p = torch.nn.Parameter(torch.randn(257))
    opt = ref.optimizers.RAdam([p], lr=1e-2, eps=1e-5)
    gs, ps = [], [p.detach().clone().numpy()]
    for _ in range(9):
        g = torch.randn(257)
        p.grad = g.clone()
        opt.step()
        gs.append(g.numpy())
        ps.append(p.detach().clone().numpy())
    np.savez_compressed(GOLD / "radam.npz", grads=np.stack(gs), params=np.stack(ps), lr=1e-2, eps=1e-5)
    print("radam.npz")
    # the weight_decay branch (optimizers.py:88-95), same gradients: decay before the update, in both step branches
    torch.manual_seed(3)
    p = torch.nn.Parameter(torch.randn(257))
    opt = ref.optimizers.RAdam([p], lr=1e-2, eps=1e-5, weight_decay=0.05)
    ps = [p.detach().clone().numpy()]
    for g in gs:
        p.grad = torch.as_tensor(g).clone()
        opt.step()
        ps.append(p.detach().clone().numpy())
    np.savez_compressed(GOLD / "radam_wd.npz", grads=np.stack(gs), params=np.stack(ps), lr=1e-2, eps=1e-5, weight_decay=0.05)
    print("radam_wd.npz")


def gold_generate(ref):
    """Reference generate_gesture() end to end (CPU): 2 s synthetic wav, synthetic 75-joint exemplar BVH written by
    the reference's bvh.save, random-init nets (seed 1234) pickled as whole modules.  temperature = 1e8 makes the
    VAE sample deterministic (z = mu + eps * 1e-8 * std).  Also pins the host plumbing (BVH parse, exemplar features)."""
    import scipy.io.wavfile as wavfile
    tmp = Path(tempfile.mkdtemp(prefix="zeggs_gold_gen_"))
    net, data, res = tmp / "net", tmp / "data", tmp / "res"
    net.mkdir(), data.mkdir()
    se, de, st = build_ref_nets(ref)
    torch.save(se, net / "speech_encoder.pt"), torch.save(de, net / "decoder.pt"), torch.save(st, net / "style_encoder.pt")
    np.savez(data / "stats.npz", **synth.make_stats())
    json.dump(synth.data_definition(), open(data / "data_definition.json", "w"))
    conf = json.load(open("/root/reference/data/processed_v1/data_pipeline_conf.json"))
    conf["audio_conf"]["normalize_loudness"] = False
    json.dump(conf, open(data / "data_pipeline_conf.json", "w"))
    wav = synth.synth_wav(32000, seed=9)
    wavfile.write(tmp / "a.wav", 16000, wav)
    clip = synth.make_bvh_clip(40, seed=4)
    ref.bvh.save(str(tmp / "ex.bvh"), clip)
    orig_load = torch.load
    torch.load = ref.torch_load
    cap = {}                                   # decoder outputs as they enter the BVH conversion (generate.py:389-405)
    orig_ortho, orig_write = ref.generate.xform_orthogonalize_from_xy, ref.generate.write_bvh

    def ortho(xy):
        cap["dec_ltxy"] = xy.detach().cpu().numpy()[0]
        return orig_ortho(xy)

    def write(fn, rp, rr, lp, lr, **kw):
        cap.update(dec_root_pos=np.array(rp), dec_root_rot=np.array(rr), dec_lpos=np.array(lp), dec_lrot=np.array(lr))
        return orig_write(fn, rp, rr, lp, lr, **kw)

    ref.generate.xform_orthogonalize_from_xy, ref.generate.write_bvh = ortho, write
    try:
        enc = ref.generate.generate_gesture(tmp / "a.wav", [(tmp / "ex.bvh", None)], net, data, res,
                                            style_encoding_type="example", blend_type="add", blend_ratio=[1.0],
                                            file_name="out", first_pose=tmp / "ex.bvh", temperature=1e8, seed=1234,
                                            use_gpu=False, use_script=False)
    finally:
        torch.load = orig_load
        ref.generate.xform_orthogonalize_from_xy, ref.generate.write_bvh = orig_ortho, orig_write
    out = ref.bvh.load(str(res / "out.bvh"))
    loaded = ref.bvh.load(str(tmp / "ex.bvh"))
    feats = ref.data_pipeline.preprocess_animation(ref.bvh.load(str(tmp / "ex.bvh")))
    names = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "lrot", "ltxy", "lvel", "lvrt", "cpos", "crot",
             "ctxy", "cvel", "cvrt", "gaze_pos", "gaze_dir")
    g = dict(wav=wav, exemplar_bvh=np.frombuffer(open(tmp / "ex.bvh", "rb").read(), dtype=np.uint8),
             out_rotations=out["rotations"], out_positions=out["positions"], out_frametime=out["frametime"],
             encoding=enc.numpy(), ex_rotations=loaded["rotations"], ex_positions=loaded["positions"],
             ex_parents=loaded["parents"], ex_offsets=loaded["offsets"])
    g.update({"feat_" + n: np.asarray(f) for n, f in zip(names, feats)})
    g.update(cap)
    np.savez_compressed(GOLD / "generate.npz", **g)
    print("generate.npz frames", out["rotations"].shape)


def gold_generate_branches(ref):
    """The branches of the reference's generate_gesture() that generate.npz does not reach (generate.py:84,158,264-354):
    two styles "stitch" (integer frame splits, helpers.py:26-37) and "add" with blend_ratio [0.3, 0.7], label strings
    (label-conditioned decoder, style width = nlabels), a pre-computed ndarray embedding, first_pose=None (the first pose is
    the LAST style exemplar's frame 0, here a (start, end)-trimmed one), audio_file=None (encodings only).  Same files, seeds
    and temperature = 1e8 convention as gold_generate()."""
    import scipy.io.wavfile as wavfile
    tmp = Path(tempfile.mkdtemp(prefix="zeggs_gold_genb_"))
    net, netl, data, res = tmp / "net", tmp / "net_label", tmp / "data", tmp / "res"
    net.mkdir(), netl.mkdir(), data.mkdir()
    se, de, st = build_ref_nets(ref)
    torch.save(se, net / "speech_encoder.pt"), torch.save(de, net / "decoder.pt"), torch.save(st, net / "style_encoder.pt")
    nlabels = len(synth.data_definition()["label_names"])
    torch.manual_seed(SEED)                      # label mode: no style encoder, decoder style width = nlabels (train.py:99-131)
    sel = ref.modules.SpeechEncoder(synth.N_AUDIO, 64, 64)
    del_ = ref.modules.Decoder(pose_input_size=synth.POSE_IN, pose_output_size=synth.POSE_OUT, speech_encoding_size=64,
                               style_encoding_size=nlabels, hidden_size=1024, num_rnn_layers=2)
    torch.save(sel, netl / "speech_encoder.pt"), torch.save(del_, netl / "decoder.pt")
    np.savez(data / "stats.npz", **synth.make_stats())
    json.dump(synth.data_definition(), open(data / "data_definition.json", "w"))
    conf = json.load(open("/root/reference/data/processed_v1/data_pipeline_conf.json"))
    conf["audio_conf"]["normalize_loudness"] = False
    json.dump(conf, open(data / "data_pipeline_conf.json", "w"))
    wav = synth.synth_wav(32000 + 4000, seed=19)           # 2.25 s -> 135 frames: 0.3 * 135 = 40.5 truncates to 40
    wavfile.write(tmp / "a.wav", 16000, wav)
    ref.bvh.save(str(tmp / "exa.bvh"), synth.make_bvh_clip(40, seed=4))
    ref.bvh.save(str(tmp / "exb.bvh"), synth.make_bvh_clip(52, seed=6))
    A, Bx, WAV = tmp / "exa.bvh", tmp / "exb.bvh", tmp / "a.wav"
    emb = np.random.default_rng(23).standard_normal(64).astype(np.float32) * 0.5
    label = synth.data_definition()["label_names"][5]
    orig_load = torch.load
    torch.load = ref.torch_load
    g = dict(wav=wav, exa_bvh=np.frombuffer(open(A, "rb").read(), dtype=np.uint8),
             exb_bvh=np.frombuffer(open(Bx, "rb").read(), dtype=np.uint8), embedding=emb, label=np.array(label),
             blend_ratio=np.array([0.3, 0.7]), trim=np.array([7, 45], dtype=np.int64))
    common = dict(temperature=1e8, seed=1234, use_gpu=False, use_script=False)

    def run(tag, audio, styles, netdir, **kw):
        enc = ref.generate.generate_gesture(audio, styles, netdir, data, res if audio is not None else None,
                                            file_name=tag if audio is not None else None, **kw, **common)
        if isinstance(enc, list):
            for i, e in enumerate(enc):
                g[f"{tag}_encoding{i}"] = e.numpy()
        else:
            g[f"{tag}_encoding"] = enc.numpy()
        if audio is not None:
            out = ref.bvh.load(str(res / (tag + ".bvh")))
            g[f"{tag}_rotations"] = out["rotations"].astype(np.float32)
            g[f"{tag}_root_positions"] = out["positions"][:, 0].astype(np.float32)

    try:
        two = [(A, None), (Bx, None)]
        run("stitch", WAV, two, net, style_encoding_type="example", blend_type="stitch", blend_ratio=[0.3, 0.7],
            first_pose=A)
        run("add", WAV, two, net, style_encoding_type="example", blend_type="add", blend_ratio=[0.3, 0.7], first_pose=A)
        run("label", WAV, [label], netl, style_encoding_type="label", blend_type="add", blend_ratio=[1.0], first_pose=Bx)
        run("ndarray", WAV, [(emb, "given")], net, style_encoding_type="example", blend_type="add", blend_ratio=[1.0],
            first_pose=Bx)
        # first_pose=None: pose 0 = frame 0 of the LAST exemplar after its (start, end) trim (generate.py:196-203,313-354)
        run("nofirst", WAV, [(A, None), (Bx, (7, 45))], net, style_encoding_type="example", blend_type="add",
            blend_ratio=[0.6, 0.4], first_pose=None)
        # audio_file=None: embeddings only; "stitch" returns the LIST of per-style encodings (generate.py:282-284)
        run("noaudio_stitch", None, two, net, style_encoding_type="example", blend_type="stitch", blend_ratio=[0.3, 0.7])
        run("noaudio_add", None, two, net, style_encoding_type="example", blend_type="add", blend_ratio=[0.3, 0.7])
        run("noaudio_trim", None, [(Bx, (7, 45))], net, style_encoding_type="example", blend_type="add", blend_ratio=[1.0])
    finally:
        torch.load = orig_load
    g["split_135"] = np.array(ref.helpers.split_by_ratio(135, [0.3, 0.7]))
    np.savez_compressed(GOLD / "generate_branches.npz", **g)
    print("generate_branches.npz", {k: v.shape for k, v in g.items() if k.endswith("encoding") or "encoding" in k})


def gold_train_iter_fp64(ref):
    """The two iterations of train_iter.npz once more through the UNMODIFIED reference train(), in float64 (modules built
    from the same fp32 seed, then .double(); the recorded batches and VAE noise replayed): the arbiter for gradient
    comparisons.  Needed because iteration 1 of that fixture is ill-conditioned: the reference's own fp32 gradients are
    0.5-0.8 % away from its fp64 gradients there (2.5e-5 at iteration 0) -- recorded as `it*_ref32_vs_ref64`."""
    import torch.nn.functional as F
    gd = np.load(GOLD / "train_iter.npz")
    window, B, n_it = int(gd["window"]), int(gd["batch"]), len(gd["loss"])
    tmp = Path(tempfile.mkdtemp(prefix="zeggs_gold64_"))
    (tmp / "data").mkdir(), (tmp / "models").mkdir(), (tmp / "logs").mkdir()
    npz, jsn = tmp / "data" / "processed_data.npz", tmp / "data" / "data_definition.json"
    np.savez(npz, **{k[5:]: gd[k] for k in gd.files if k.startswith("data_")})
    json.dump(synth.data_definition(), open(jsn, "w"))
    rec = dict(grads=[], loss=[])
    state = dict(it=0)

    class ReplayDL:
        def __init__(self, ds, **kw):
            pass

        def __iter__(self):
            for it in range(n_it):
                yield [torch.as_tensor(gd[f"it{it}_batch{j}"]).double() for j in range(11)]

    def fake_randn_like(x, *a, **k):
        if tuple(x.shape) == (B, 64):
            return torch.as_tensor(gd[f"it{min(state['it'], n_it - 1)}_eps"]).to(x.dtype)
        return torch.zeros_like(x)

    def double_module(cls):
        def make(*a, **k):
            m = cls(*a, **k).double()
            m.register_forward_pre_hook(lambda mod, args: tuple(x.double() if torch.is_tensor(x) and x.is_floating_point()
                                                                else x for x in args))
            return m
        return make

    orig_step = ref.optimizers.RAdam.step

    def rec_step(self, closure=None):
        ps = [p for g in self.param_groups for p in g["params"]]
        rec["grads"].append(np.concatenate([p.grad.flatten()[sample_idx(p.numel())].numpy() for p in ps]))
        state["it"] += 1
        return orig_step(self, closure)

    orig_backward = torch.Tensor.backward

    def rec_backward(self, *a, **k):
        rec["loss"].append(float(self.detach()))
        return orig_backward(self, *a, **k)

    rt = ref.train
    saved = (rt.DataLoader, torch.randn_like, F.dropout, rt.SpeechEncoder, rt.Decoder, rt.StyleEncoder, torch.save)
    rt.DataLoader, torch.randn_like = ReplayDL, fake_randn_like
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x
    rt.SpeechEncoder, rt.Decoder, rt.StyleEncoder = (double_module(c) for c in saved[3:6])
    torch.save = lambda *a, **k: None                  # (the pre-hooks are not picklable; no checkpoint is needed here)
    ref.optimizers.RAdam.step = rt.RAdam.step = rec_step
    torch.Tensor.backward = rec_backward
    random.seed(0)
    train_opt = dict(niterations=0.001, batchsize=B, window=window, change_pace=True, learning_rate=1e-4,
                     learning_rate_decay=0.995, eps=1e-5, resume=False, use_gpu=False, thread_count=1, seed=SEED,
                     use_tensorboard=False, style_encoding_type="example", generate_samples_step=10 ** 9, use_script=False)
    try:
        rt.train(tmp / "models", tmp / "logs", npz, jsn, train_opt, NET_OPT)
    finally:
        (rt.DataLoader, torch.randn_like, F.dropout, rt.SpeechEncoder, rt.Decoder, rt.StyleEncoder, torch.save) = saved
        ref.optimizers.RAdam.step = rt.RAdam.step = orig_step
        torch.Tensor.backward = orig_backward
    out = dict(loss64=np.array(rec["loss"]))
    for it in range(n_it):
        g64, g32 = rec["grads"][it], gd[f"it{it}_grad_samples"].astype(np.float64)
        out[f"it{it}_grad_samples64"] = g64
        out[f"it{it}_ref32_vs_ref64"] = np.float64(np.abs(g32 - g64).max() / np.abs(g64).max())
        print(f"train_iter_fp64: iteration {it}: loss {rec['loss'][it]:.8f} (fp32 {gd['loss'][it]:.8f}), reference fp32 vs fp64 "
              f"gradients {out[f'it{it}_ref32_vs_ref64']:.2e} of the largest entry")
    np.savez_compressed(GOLD / "train_iter_fp64.npz", **out)


def gold_train_iter_perturb(ref, n_seeds=10, rel=1e-7):
    """is iteration 1 of train_iter.npz ill-conditioned in the REFERENCE itself?  The two recorded
    iterations are replayed through the unmodified reference train() in float32, `n_seeds` times, with every floating-point
    batch tensor of ITERATION 1 multiplied by (1 + rel * N(0, 1)) (iteration 0 untouched, so the weights iteration 1 starts
    from are the recorded ones bit for bit).  Per tensor and seed: length ratio |g32| / |g64| and 1 - cosine against the float64
    replay (train_iter_fp64.npz); also the same two numbers for a float64 replay of the perturbed inputs (what the
    perturbation itself does to the true gradient).  -> tests/golden/train_iter_perturb.npz: the measured spread the
    reference-side test derives its iteration-1 bound from."""
    import torch.nn.functional as F
    gd = np.load(GOLD / "train_iter.npz")
    g64 = np.load(GOLD / "train_iter_fp64.npz")
    window, B, n_it = int(gd["window"]), int(gd["batch"]), len(gd["loss"])
    se, de, st = build_ref_nets(ref)
    sizes = [len(sample_idx(p.numel())) for m in (se, de, st) for p in m.parameters()]
    ref64 = g64["it1_grad_samples64"]

    def replay(seed, double):
        tmp = Path(tempfile.mkdtemp(prefix="zeggs_goldp_"))
        (tmp / "data").mkdir(), (tmp / "models").mkdir(), (tmp / "logs").mkdir()
        npz, jsn = tmp / "data" / "processed_data.npz", tmp / "data" / "data_definition.json"
        np.savez(npz, **{k[5:]: gd[k] for k in gd.files if k.startswith("data_")})
        json.dump(synth.data_definition(), open(jsn, "w"))
        rec, state = dict(grads=[]), dict(it=0)
        rng = np.random.default_rng(1000 + seed) if seed is not None else None

        class ReplayDL:
            def __init__(self, ds, **kw):
                pass

            def __iter__(self):
                for it in range(n_it):
                    b = []
                    for j in range(11):
                        a = gd[f"it{it}_batch{j}"]
                        if it == 1 and rng is not None and a.dtype.kind == "f":
                            a = (a.astype(np.float64) * (1.0 + rel * rng.standard_normal(a.shape))).astype(a.dtype)
                        t = torch.as_tensor(a)
                        b.append(t.double() if double and t.is_floating_point() else t)
                    yield b

        def fake_randn_like(x, *a, **k):
            if tuple(x.shape) == (B, 64):
                return torch.as_tensor(gd[f"it{min(state['it'], n_it - 1)}_eps"]).to(x.dtype)
            return torch.zeros_like(x)

        def double_module(cls):
            def make(*a, **k):
                m = cls(*a, **k).double()
                m.register_forward_pre_hook(lambda mod, args: tuple(x.double() if torch.is_tensor(x) and x.is_floating_point()
                                                                    else x for x in args))
                return m
            return make

        orig_step = ref.optimizers.RAdam.step

        def rec_step(self, closure=None):
            ps = [p for g in self.param_groups for p in g["params"]]
            rec["grads"].append(np.concatenate([p.grad.flatten()[sample_idx(p.numel())].double().numpy() for p in ps]))
            state["it"] += 1
            return orig_step(self, closure)

        rt = ref.train
        saved = (rt.DataLoader, torch.randn_like, F.dropout, rt.SpeechEncoder, rt.Decoder, rt.StyleEncoder, torch.save)
        rt.DataLoader, torch.randn_like = ReplayDL, fake_randn_like
        F.dropout = lambda x, p=0.5, training=True, inplace=False: x
        if double:
            rt.SpeechEncoder, rt.Decoder, rt.StyleEncoder = (double_module(c) for c in saved[3:6])
        torch.save = lambda *a, **k: None
        ref.optimizers.RAdam.step = rt.RAdam.step = rec_step
        random.seed(0)
        train_opt = dict(niterations=0.001, batchsize=B, window=window, change_pace=True, learning_rate=1e-4,
                         learning_rate_decay=0.995, eps=1e-5, resume=False, use_gpu=False, thread_count=1, seed=SEED,
                         use_tensorboard=False, style_encoding_type="example", generate_samples_step=10 ** 9,
                         use_script=False)
        try:
            rt.train(tmp / "models", tmp / "logs", npz, jsn, train_opt, NET_OPT)
        finally:
            (rt.DataLoader, torch.randn_like, F.dropout, rt.SpeechEncoder, rt.Decoder, rt.StyleEncoder, torch.save) = saved
            ref.optimizers.RAdam.step = rt.RAdam.step = orig_step
        return rec["grads"]

    def per_tensor(g, r):
        ratio, cosd, off = [], [], 0
        for n in sizes:
            a, b = g[off:off + n], r[off:off + n]
            ratio.append(np.linalg.norm(a) / max(1e-300, np.linalg.norm(b)))
            cosd.append(1.0 - float(np.dot(a, b) / max(1e-300, np.linalg.norm(a) * np.linalg.norm(b))))
            off += n
        return np.array(ratio), np.array(cosd)

    base32 = replay(None, False)
    assert np.array_equal(base32[1].astype(np.float32), gd["it1_grad_samples"]), "the unperturbed fp32 replay must reproduce the fixture"
    out = dict(rel=np.float64(rel))
    out["base_ratio"], out["base_cosd"] = per_tensor(base32[1], ref64)
    R32, C32, R64, C64 = [], [], [], []
    for sd in range(n_seeds):
        r, c = per_tensor(replay(sd, False)[1], ref64)
        R32.append(r), C32.append(c)
        r, c = per_tensor(replay(sd, True)[1], ref64)
        R64.append(r), C64.append(c)
        print(f"train_iter_perturb seed {sd}: fp32 length ratio {R32[-1].min():.5f} .. {R32[-1].max():.5f} (1-cos <= {C32[-1].max():.1e}); "
              f"fp64 on the same perturbed inputs {R64[-1].min():.7f} .. {R64[-1].max():.7f} (1-cos <= {C64[-1].max():.1e})", flush=True)
    out.update(ratio32=np.array(R32), cosd32=np.array(C32), ratio64=np.array(R64), cosd64=np.array(C64))
    np.savez_compressed(GOLD / "train_iter_perturb.npz", **out)
    print("train_iter_perturb.npz: fp32 ratio over seeds and tensors", out["ratio32"].min(), out["ratio32"].max(),
          "base", out["base_ratio"].min(), out["base_ratio"].max())


def gold_variants_batch(ref):
    """The same two nets (seed 4321) at a batch the stage kernels split into two 16-row blocks, with a style that changes every
    frame: reference forward outputs AND the reference's own autograd gradients (fp32, as the reference trains) of a seeded
    weighted sum -- input gradients in full, every parameter gradient as 512 evenly spaced samples + its max |g|.
    Pins the round-4 stage-kernel path of rnn_cond="film" / type="gru" to the reference beyond the tiny variants.npz case."""
    torch.manual_seed(4321)
    de = ref.modules.Decoder(pose_input_size=synth.POSE_IN, pose_output_size=synth.POSE_OUT, speech_encoding_size=64,
                             style_encoding_size=64, hidden_size=1024, num_rnn_layers=2, rnn_cond="film")
    st = ref.modules.StyleEncoder(synth.POSE_IN, 512, 64, type="gru", use_vae=True)
    de.train(), st.train()          # (neither net has dropout: train() only matters for autograd bookkeeping)
    stats = synth.make_stats()
    B, T, L = 19, 12, 33
    clips = [synth.make_clip(T + L, seed=170 + b, stats=stats) for b in range(B)]
    W = {k: torch.as_tensor(np.stack([c[k][:T] for c in clips])) for k in clips[0]}
    in_mean, in_std = torch.as_tensor(stats["anim_input_mean"]), torch.as_tensor(stats["anim_input_std"])
    out_mean, out_std = torch.as_tensor(stats["anim_output_mean"]), torch.as_tensor(stats["anim_output_std"])
    ex = []
    for c in clips:
        ex.append(np.concatenate([c["Y_root_vel"][T:T + L], c["Y_root_vrt"][T:T + L],
                                  c["Y_lpos"][T:T + L].reshape(L, -1), c["Y_ltxy"][T:T + L].reshape(L, -1),
                                  c["Y_lvel"][T:T + L].reshape(L, -1), c["Y_lvrt"][T:T + L].reshape(L, -1),
                                  np.zeros((L, 3), np.float32)], axis=1))
    example = torch.as_tensor(np.stack(ex))
    rng = np.random.default_rng(18)
    eps = torch.as_tensor(rng.standard_normal((B, 64)).astype(np.float32))
    speech = torch.as_tensor(rng.standard_normal((B, T, 64)).astype(np.float32) * 0.5).requires_grad_(True)
    style = torch.as_tensor(rng.standard_normal((B, T, 64)).astype(np.float32) * 0.5).requires_grad_(True)
    names = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "ltxy", "lvel", "lvrt")
    orig = torch.randn_like
    torch.randn_like = lambda x, *a, **k: eps.to(x.dtype)
    try:
        z, mu, logvar = st((example - in_mean) / in_std, 1.0)
    finally:
        torch.randn_like = orig
    O = de(W["Y_root_pos"][:, 0], W["Y_root_rot"][:, 0], W["Y_root_vel"][:, 0], W["Y_root_vrt"][:, 0],
           W["Y_lpos"][:, 0], W["Y_ltxy"][:, 0], W["Y_lvel"][:, 0], W["Y_lvrt"][:, 0], W["Y_gaze_pos"], speech,
           style, torch.LongTensor(synth.PARENTS), in_mean, in_std, out_mean, out_std, synth.DT)
    # the weights of the scalar that is differentiated are re-created from this seed by the tests (not stored: 2 MB)
    gen = torch.Generator().manual_seed(1804)
    wts = [torch.randn(tuple(o.shape), generator=gen) for o in O]
    wz, wm, wl = (torch.randn(B, 64, generator=gen) for _ in range(3))
    (sum((o * w).sum() for o, w in zip(O, wts)) + (z * wz).sum() + (mu * wm).sum() + (logvar * wl).sum()).backward()
    # (first pose, gaze targets and the exemplar are NOT stored: the tests rebuild them from synth.make_clip(T + L, seed=170 + b),
    #  the same deterministic NumPy generator; their checksums are)
    out = {"clip_T_L": np.array([T, L]), "sum_example": fingerprint(example), "sum_gaze": fingerprint(W["Y_gaze_pos"])}
    out.update(in_eps=eps.numpy(),
               in_speech=speech.detach().numpy(), in_style=style.detach().numpy(), gru_z=z.detach().numpy(),
               gru_mu=mu.detach().numpy(), gru_logvar=logvar.detach().numpy(), weight_seed=np.int64(1804),
               d_speech=speech.grad.numpy(), d_style=style.grad.numpy())
    out.update({"O_" + n: o.detach().numpy() for n, o in zip(names, O)})
    for tag, net in (("decoder", de), ("style", st)):
        for k, p in net.named_parameters():
            g = p.grad.detach().flatten()
            idx = np.unique(np.linspace(0, g.numel() - 1, 512).astype(np.int64))
            out[f"gidx_{tag}.{k}"] = idx
            out[f"gsamp_{tag}.{k}"] = g[torch.as_tensor(idx)].numpy()
            out[f"gmax_{tag}.{k}"] = np.float32(g.abs().max())
    np.savez_compressed(GOLD / "variants_batch.npz", **out)
    print("variants_batch.npz", out["O_lpos"].shape, (GOLD / "variants_batch.npz").stat().st_size)


def gold_width512(ref):
    """decoder.nhidden = 512 (ZEGGS/train.py:129 honours the option; the shipped configs use 1024): the reference's
    RecurrentDecoderNormal at the other width, B = 18, T = 8 -- outputs in full, input gradients in full, 512 samples of every
    parameter gradient of the reference's autograd (weights of the scalar from the stored seed).  Seed 5512."""
    torch.manual_seed(5512)
    de = ref.modules.Decoder(pose_input_size=synth.POSE_IN, pose_output_size=synth.POSE_OUT, speech_encoding_size=64,
                             style_encoding_size=64, hidden_size=512, num_rnn_layers=2)
    de.train()
    stats = synth.make_stats()
    B, T = 18, 8
    clips = [synth.make_clip(T, seed=270 + b, stats=stats) for b in range(B)]
    W = {k: torch.as_tensor(np.stack([c[k][:T] for c in clips])) for k in clips[0]}
    in_mean, in_std = torch.as_tensor(stats["anim_input_mean"]), torch.as_tensor(stats["anim_input_std"])
    out_mean, out_std = torch.as_tensor(stats["anim_output_mean"]), torch.as_tensor(stats["anim_output_std"])
    rng = np.random.default_rng(28)
    speech = torch.as_tensor(rng.standard_normal((B, T, 64)).astype(np.float32) * 0.5).requires_grad_(True)
    style = torch.as_tensor(rng.standard_normal((B, T, 64)).astype(np.float32) * 0.5).requires_grad_(True)
    names = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "ltxy", "lvel", "lvrt")
    O = de(W["Y_root_pos"][:, 0], W["Y_root_rot"][:, 0], W["Y_root_vel"][:, 0], W["Y_root_vrt"][:, 0],
           W["Y_lpos"][:, 0], W["Y_ltxy"][:, 0], W["Y_lvel"][:, 0], W["Y_lvrt"][:, 0], W["Y_gaze_pos"], speech,
           style, torch.LongTensor(synth.PARENTS), in_mean, in_std, out_mean, out_std, synth.DT)
    gen = torch.Generator().manual_seed(1805)
    wts = [torch.randn(tuple(o.shape), generator=gen) for o in O]
    sum((o * w).sum() for o, w in zip(O, wts)).backward()
    out = {"clip_T": np.int64(T), "sum_gaze": fingerprint(W["Y_gaze_pos"]), "weight_seed": np.int64(1805),
           "in_speech": speech.detach().numpy(), "in_style": style.detach().numpy(),
           "d_speech": speech.grad.numpy(), "d_style": style.grad.numpy()}
    out.update({"O_" + n: o.detach().numpy() for n, o in zip(names, O)})
    for k, v in de.state_dict().items():
        out[f"fp_decoder.{k}"] = fingerprint(v)
    for k, p in de.named_parameters():
        g = p.grad.detach().flatten()
        idx = np.unique(np.linspace(0, g.numel() - 1, 512).astype(np.int64))
        out[f"gidx_decoder.{k}"] = idx
        out[f"gsamp_decoder.{k}"] = g[torch.as_tensor(idx)].numpy()
        out[f"gmax_decoder.{k}"] = np.float32(g.abs().max())
    np.savez_compressed(GOLD / "width512.npz", **out)
    print("width512.npz", out["O_lpos"].shape, (GOLD / "width512.npz").stat().st_size)


def main():
    assert ref_shims.available(), "/root/reference is required to (re)generate golden vectors"
    GOLD.mkdir(parents=True, exist_ok=True)
    ref = ref_shims.load()
    torch.set_num_threads(1)
    which = sys.argv[1:] or ["nets", "train", "mel", "dataset", "radam", "generate", "variants", "generate_branches", "anim_orders"]
    if "variants" in which:
        gold_variants(ref)
    if "variants" in which or "variants_batch" in which:
        gold_variants_batch(ref)
    if "nets" in which or "width512" in which:
        gold_width512(ref)
    if "nets" in which:
        gold_nets(ref)
    if "mel" in which:
        gold_mel(ref)
    if "dataset" in which:
        gold_dataset(ref)
    if "radam" in which:
        gold_radam(ref)
    if "anim_orders" in which:
        gold_anim_orders(ref)
    if "generate" in which:
        gold_generate(ref)
    if "generate_branches" in which:
        gold_generate_branches(ref)
    if "train" in which:
        gold_train_iter(ref)
    if "train" in which or "train_fp64" in which:
        gold_train_iter_fp64(ref)
    if "train_perturb" in which:
        gold_train_iter_perturb(ref)


if __name__ == "__main__":
    main()
