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
