"""Drop-in network classes for the ZeroEGGS hot path, executed by the HIP library.

Mirrors the public surface of the reference's `ZEGGS/modules.py` (same class
names, constructor signatures, forward signatures and `state_dict` keys --
SURVEY.md section 8(b)) so trained weights move both ways, but every forward /
backward runs in hand-written gfx950 kernels through the C ABI declared in
include/zeggs_hip.h (see `zeggs.ops`).  There is NO CPU / eager fallback: a
forward on a module raises if the HIP library cannot be loaded.

Parameter creation order and initialisers match the reference so that seeded
construction (`torch.manual_seed(s)`; SpeechEncoder, Decoder, StyleEncoder in
that order, reference train.py:118-139) gives bit-identical initial weights.
"""
import math

import torch
import torch.nn as nn

from . import ops


# ----------------------------------------------------------------------------
# parameter containers (layout == reference state_dict)
# ----------------------------------------------------------------------------
class _RecurrentDecoderNormal(nn.Module):
    """Parameters of reference RecurrentDecoderNormal (modules.py:165-177)."""

    def __init__(self, pose_input_size, speech_size, style_size, output_size, hidden_size, num_rnn_layers):
        super().__init__()
        all_in = pose_input_size + speech_size + style_size
        self.layer0 = nn.Linear(all_in, hidden_size)
        self.layer1 = nn.GRU(all_in + hidden_size, hidden_size, num_rnn_layers, batch_first=True)
        self.layer2 = nn.Linear(hidden_size, output_size)


class _RecurrentDecoderFiLM(nn.Module):
    """Parameters of reference RecurrentDecoderFiLM (modules.py:188-211): the style modulates the two hidden layers
    (feature-wise affine, gamma/beta predicted from the style) instead of entering the step input."""

    def __init__(self, pose_input_size, speech_size, style_size, output_size, hidden_size, num_rnn_layers):
        super().__init__()
        self.hidden_size = hidden_size
        self.gammas_predictor = LinearNorm(style_size, hidden_size * 2, w_init_gain="linear")
        self.betas_predictor = LinearNorm(style_size, hidden_size * 2, w_init_gain="linear")
        self.layer0 = nn.Linear(pose_input_size + speech_size, hidden_size)
        self.layer1 = nn.GRU(pose_input_size + speech_size + hidden_size, hidden_size, num_rnn_layers,
                             batch_first=True, dropout=0.0)
        self.layer2 = nn.Linear(hidden_size, hidden_size)
        self.layer3 = nn.Linear(hidden_size, output_size)


class CellStateEncoder(nn.Module):
    """Parameters of reference CellStateEncoder (modules.py:230-236)."""

    def __init__(self, input_size, hidden_size, num_rnn_layers):
        super().__init__()
        self.num_rnn_layers = num_rnn_layers
        self.layer0 = nn.Linear(input_size, hidden_size)
        self.layer1 = nn.Linear(hidden_size, hidden_size)
        self.layer2 = nn.Linear(hidden_size, hidden_size * num_rnn_layers)


class ConvNorm1D(nn.Module):
    """Conv1d over [B, L, C] with Xavier-uniform weight (reference modules.py:615-641)."""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, padding=None, dilation=1,
                 bias=True, w_init_gain="linear"):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, stride=stride,
                              padding=padding, dilation=dilation, bias=bias)
        nn.init.xavier_uniform_(self.conv.weight, gain=nn.init.calculate_gain(w_init_gain))


class _MultiHeadAttention(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.multi_head_attention = nn.MultiheadAttention(hidden_size, 4, 0.1)
        self.dropout = nn.Dropout(0.1)
        self.layer_norm = nn.LayerNorm(hidden_size)


class _PositionWiseConvFF(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.convs = nn.Sequential(
            ConvNorm1D(hidden_size, hidden_size, kernel_size=3, padding=1, w_init_gain="relu"),
            nn.ReLU(),
            ConvNorm1D(hidden_size, hidden_size, kernel_size=3, padding=1, w_init_gain="linear"),
            nn.Dropout(0.1),
        )
        self.layer_norm = nn.LayerNorm(hidden_size)


class _FFTBlock(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.attention = _MultiHeadAttention(hidden_size)
        self.feed_forward = _PositionWiseConvFF(hidden_size)


class StyleEncoderAttn(nn.Module):
    """Parameters of reference StyleEncoderAttn (modules.py:353-389).  The
    sinusoidal table (reference PositionalEncoding, :450-459) is built on demand
    for the exemplar length (ops.positional_table), so no 20000x128 attribute is kept."""

    def __init__(self, input_size, hidden_size, style_embedding_size):
        super().__init__()
        self.embed_dim = style_embedding_size
        self.convs = nn.Sequential(
            ConvNorm1D(input_size, hidden_size, kernel_size=3, padding=1, w_init_gain="relu"),
            nn.ReLU(), nn.LayerNorm(hidden_size), nn.Dropout(0.2),
            ConvNorm1D(hidden_size, style_embedding_size, kernel_size=3, padding=1, w_init_gain="relu"),
            nn.ReLU(), nn.LayerNorm(style_embedding_size), nn.Dropout(0.2),
        )
        self.blocks = nn.ModuleList([_FFTBlock(style_embedding_size)])


class StyleEncoderGRU(nn.Module):
    """Parameters of reference StyleEncoderGRU (modules.py:307-337)."""

    def __init__(self, input_size, hidden_size, style_embedding_size):
        super().__init__()
        self.convs = nn.Sequential(
            ConvNorm1D(input_size, hidden_size, kernel_size=3, stride=1, padding=1, dilation=1, w_init_gain="relu"),
            nn.ReLU(),
            ConvNorm1D(hidden_size, hidden_size, kernel_size=3, stride=1, padding=1, dilation=1, w_init_gain="relu"),
            nn.ReLU(),
        )
        self.rnn_layer = nn.GRU(hidden_size, hidden_size, 1, batch_first=True, bidirectional=True)
        self.projection_layer = LinearNorm(hidden_size * 2, style_embedding_size, w_init_gain="linear")


# ----------------------------------------------------------------------------
# public modules
# ----------------------------------------------------------------------------
class SpeechEncoder(nn.Module):
    """conv1d(k=1)+ELU -> conv1d(k=31, replicate)+ELU -> Linear+ELU, dropout .2
    after the two convs in training (reference modules.py:249-272)."""

    def __init__(self, input_size, hidden_size, output_size):
        super().__init__()
        self.layer0 = nn.Conv1d(input_size, hidden_size, kernel_size=1, padding="same", padding_mode="replicate")
        self.drop0 = nn.Dropout(p=0.2)
        self.layer1 = nn.Conv1d(hidden_size, output_size, kernel_size=31, padding="same", padding_mode="replicate")
        self.drop1 = nn.Dropout(p=0.2)
        self.layer2 = nn.Linear(output_size, output_size)

    def forward(self, x):
        p = 0.2 if self.training else 0.0
        return ops.speech_encoder(x, self.layer0.weight, self.layer0.bias, self.layer1.weight,
                                  self.layer1.bias, self.layer2.weight, self.layer2.bias, p)


class StyleEncoder(nn.Module):
    """Style encoder + VAE re-parameterisation (reference modules.py:278-304).
    As in the reference, eval mode still samples eps; `eps` may be injected for
    testing (otherwise drawn from the library's counter-hash normal stream)."""

    def __init__(self, input_size, hidden_size, style_embedding_size, type="attn", use_vae=False):
        super().__init__()
        self.use_vae = use_vae
        self.style_embedding_size = style_embedding_size
        output_size = 2 * style_embedding_size if use_vae else style_embedding_size
        if type == "gru":
            self.encoder = StyleEncoderGRU(input_size, hidden_size, output_size)
        elif type == "attn":
            self.encoder = StyleEncoderAttn(input_size, hidden_size, output_size)
        else:
            raise ValueError(f"unknown style encoder type {type!r}")

    def forward(self, input, temprature: float = 1.0, eps=None):
        if isinstance(self.encoder, StyleEncoderGRU):
            out = ops.style_encoder_gru(input, self.encoder)
        else:
            out = ops.style_encoder_attn(input, self.encoder, self.training)
        if not self.use_vae:
            return out, None, None
        S = self.style_embedding_size
        if eps is None:
            eps = ops.randn((out.shape[0], S), out.device)      # library counter-hash stream (ops.manual_seed)
        return ops.vae_reparam(out, eps, float(temprature), S)


class Decoder(nn.Module):
    """Autoregressive GRU gesture decoder (reference modules.py:11-162)."""

    def __init__(self, pose_input_size, pose_output_size, speech_encoding_size, style_encoding_size,
                 hidden_size, num_rnn_layers, rnn_cond="normal"):
        super().__init__()
        if num_rnn_layers != 2:
            raise NotImplementedError("the reference hard-codes 2 GRU layers (train.py:130)")
        if rnn_cond == "normal":
            cls = _RecurrentDecoderNormal
        elif rnn_cond == "film":          # fragment-packed stage kernels, 4 launches per step (the persistent sweeps are "normal" only)
            cls = _RecurrentDecoderFiLM
        else:
            raise ValueError(f"unknown rnn_cond {rnn_cond!r}")
        self.recurrent_decoder = cls(pose_input_size, speech_encoding_size, style_encoding_size, pose_output_size,
                                     hidden_size, num_rnn_layers)
        self.cell_state_encoder = CellStateEncoder(pose_input_size + style_encoding_size,
                                                   hidden_size, num_rnn_layers)

    def forward(self, Z_root_pos, Z_root_rot, Z_root_vel, Z_root_vrt, Z_lpos, Z_ltxy, Z_lvel, Z_lvrt,
                Z_gaze_pos, speech_encoding, style_encoding, parents, anim_input_mean, anim_input_std,
                anim_output_mean, anim_output_std, dt: float):
        return ops.decoder_rollout(self, Z_root_pos, Z_root_rot, Z_root_vel, Z_root_vrt, Z_lpos, Z_ltxy,
                                   Z_lvel, Z_lvrt, Z_gaze_pos, speech_encoding, style_encoding,
                                   anim_input_mean, anim_input_std, anim_output_mean, anim_output_std,
                                   float(dt))


# names under which the reference pickles its sub-modules (`torch.save(module)` stores the class path
# `modules.<Class>`); zeggs.compat maps that module name here so reference checkpoints load into this engine
RecurrentDecoderNormal = _RecurrentDecoderNormal
RecurrentDecoderFiLM = _RecurrentDecoderFiLM
FFTBlock = _FFTBlock
MultiHeadAttention = _MultiHeadAttention
PositionWiseConvFF = _PositionWiseConvFF


class PositionalEncoding(nn.Module):
    """Placeholder for un-pickling reference checkpoints (their 20000x128 table attribute is ignored: the engine
    builds the same sinusoidal table on demand, see ops.positional_table)."""

    def __init__(self, embed_dim=128, max_len=20000, timestep=10000.0):
        super().__init__()
        self.embed_dim = embed_dim


class LinearNorm(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, w_init_gain="linear"):
        super().__init__()
        self.linear_layer = nn.Linear(in_dim, out_dim, bias=bias)
        nn.init.xavier_uniform_(self.linear_layer.weight, gain=nn.init.calculate_gain(w_init_gain))


# ----------------------------------------------------------------------------
# free functions of the reference module namespace (ZEGGS/modules.py:673-813) -- what `from modules import ...` in the
# reference's train.py:20-24 / generate.py resolves to when this module stands in for it (INTEGRATION.md route 2).
# Same names, argument order and return values; the arithmetic runs in csrc/funcs.hip (forward + backward).
# ----------------------------------------------------------------------------
def normalize(x, eps: float = 1e-8):
    """x / (||x|| + eps) over the last dimension (reference modules.py:673-675)"""
    return ops.normalize_vec(x, eps)


def vectorize_input(Z_root_pos, Z_root_rot, Z_root_vel, Z_root_vrt, Z_lpos, Z_ltxy, Z_lvel, Z_lvrt, Z_gaze_pos, parents,
                    anim_input_mean, anim_input_std):
    """The decoder's normalised autoregressive input [B, 1134] (reference modules.py:677-713; `parents` is unused there too)"""
    return ops.vectorize_input(Z_root_pos, Z_root_rot, Z_root_vel, Z_root_vrt, Z_lpos, Z_ltxy, Z_lvel, Z_lvrt, Z_gaze_pos,
                               anim_input_mean, anim_input_std)


def devectorize_output(predicted, Z_root_pos, Z_root_rot, batchsize: int, njoints: int, dt: float, anim_output_mean,
                       anim_output_std):
    """De-normalise one decoder output and integrate the root (reference modules.py:716-742) ->
    (P_root_pos, P_root_rot, P_root_vel, P_root_vrt, P_lpos, P_ltxy, P_lvel, P_lvrt)"""
    pose, rpos, rrot = ops.devectorize_output(predicted, Z_root_pos, Z_root_rot, njoints, dt, anim_output_mean,
                                              anim_output_std)
    J, B = njoints, batchsize
    return (rpos, rrot, pose[:, 0:3], pose[:, 3:6], pose[:, 6:6 + 3 * J].reshape(B, J, 3),
            pose[:, 6 + 3 * J:6 + 9 * J].reshape(B, J, 2, 3), pose[:, 6 + 9 * J:6 + 12 * J].reshape(B, J, 3),
            pose[:, 6 + 12 * J:6 + 15 * J].reshape(B, J, 3))


def generalized_logistic_function(x, center=0.0, B=1.0, A=0.0, K=1.0, C=1.0, Q=1.0, nu=1.0):
    return A + (K - A) / (C + Q * math.exp(-B * (x - center))) ** (1 / nu)


def kl_div_weight(iteration):
    """KL annealing weight of reference compute_KL_div (modules.py:773-788)."""
    return min(generalized_logistic_function(iteration, center=7500, B=0.005), 2e-1)


def compute_KL_div(mu, logvar, iteration):
    """-> (kl_div device scalar, kl_div_weight float) (reference modules.py:764-789)"""
    return ops.kl_div(mu, logvar), kl_div_weight(iteration)


def get_mask_from_lengths(lengths):
    """bool mask [B, max(lengths)] (reference modules.py:802-813)"""
    return ops.mask_from_lengths(lengths)
