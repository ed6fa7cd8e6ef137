"""Oracle restatement of the ZeroEGGS networks (torch CPU, dtype-generic).

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Pure functions over a
`state_dict`-style mapping of weights; works in float32 or float64, and is
differentiable through torch autograd (used to produce reference gradients).

Reference being restated (paths relative to /root/reference):
  ZEGGS/modules.py:249-272   SpeechEncoder
  ZEGGS/modules.py:278-304   StyleEncoder (VAE re-parameterisation)
  ZEGGS/modules.py:346-420   StyleEncoderAttn (+ :445-481 pos-enc, :484-612 FFT block)
  ZEGGS/modules.py:230-243   CellStateEncoder
  ZEGGS/modules.py:165-185   RecurrentDecoderNormal (nn.GRU gate order r,z,n)
  ZEGGS/modules.py:47-162    Decoder.forward (autoregressive rollout)
  ZEGGS/modules.py:677-742   vectorize_input / devectorize_output
  ZEGGS/anim/tquat.py        quaternion helpers
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# quaternion helpers (ZEGGS/anim/tquat.py:6-108); quaternions are (w, x, y, z)
# ----------------------------------------------------------------------------
def _cross(a, b):
    ax, ay, az = a[..., 0], a[..., 1], a[..., 2]
    bx, by, bz = b[..., 0], b[..., 1], b[..., 2]
    return torch.stack([ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx], dim=-1)


def quat_mul(x, y):
    """tquat.py:6-15"""
    x0, x1, x2, x3 = x[..., 0], x[..., 1], x[..., 2], x[..., 3]
    y0, y1, y2, y3 = y[..., 0], y[..., 1], y[..., 2], y[..., 3]
    return torch.stack([
        y0 * x0 - y1 * x1 - y2 * x2 - y3 * x3,
        y0 * x1 + y1 * x0 - y2 * x3 + y3 * x2,
        y0 * x2 + y1 * x3 + y2 * x0 - y3 * x1,
        y0 * x3 - y1 * x2 + y2 * x1 + y3 * x0], dim=-1)


def quat_mul_vec(q, v):
    """tquat.py:18-20: t = 2 q_v x v ; v + w t + q_v x t"""
    t = 2.0 * _cross(q[..., 1:], v)
    return v + q[..., 0:1] * t + _cross(q[..., 1:], t)


def quat_inv(q):
    """tquat.py:23-24"""
    return torch.cat([q[..., 0:1], -q[..., 1:]], dim=-1)


def quat_inv_mul_vec(q, v):
    """tquat.py:31-32"""
    return quat_mul_vec(quat_inv(q), v)


def quat_normalize(x, eps=1e-5):
    """tquat.py:50-51: x / (|x| + eps)"""
    return x / (torch.sqrt(torch.sum(x * x, dim=-1, keepdim=True)) + eps)


def quat_exp(x, eps=1e-5):
    """tquat.py:94-99"""
    half = torch.sqrt(torch.sum(x * x, dim=-1, keepdim=True))
    small = quat_normalize(torch.cat([torch.ones_like(half), x], dim=-1))
    big = torch.cat([torch.cos(half), x * torch.sinc(half / math.pi)], dim=-1)
    return torch.where(half < eps, small, big)


def quat_from_helical(x, eps=1e-5):
    """tquat.py:105-107"""
    return quat_exp(x / 2.0, eps)


def quat_to_xform(q):
    """tquat.py:54-69 -> [..., 3, 3]"""
    qw, qx, qy, qz = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    x2, y2, z2 = qx + qx, qy + qy, qz + qz
    xx, yy, wx = qx * x2, qy * y2, qw * x2
    xy, yz, wy = qx * y2, qy * z2, qw * y2
    xz, zz, wz = qx * z2, qz * z2, qw * z2
    r0 = torch.stack([1.0 - (yy + zz), xy - wz, xz + wy], dim=-1)
    r1 = torch.stack([xy + wz, 1.0 - (xx + zz), yz - wx], dim=-1)
    r2 = torch.stack([xz - wy, yz + wx, 1.0 - (xx + yy)], dim=-1)
    return torch.stack([r0, r1, r2], dim=-2)


# ----------------------------------------------------------------------------
# Speech encoder (modules.py:249-272), eval mode (dropout = identity)
# ----------------------------------------------------------------------------
def speech_encoder(w, x, prefix=""):
    """x: [B, T, F] already normalised ((a - mean)/std, train.py:232-234)."""
    p = prefix
    h = x.transpose(1, 2)                                        # [B, F, T]
    h = F.elu(F.conv1d(h, w[p + "layer0.weight"], w[p + "layer0.bias"]))
    k = w[p + "layer1.weight"].shape[-1]
    pad = (k - 1) // 2
    h = F.pad(h, (pad, k - 1 - pad), mode="replicate")           # padding="same", replicate
    h = F.elu(F.conv1d(h, w[p + "layer1.weight"], w[p + "layer1.bias"]))
    h = h.transpose(1, 2)                                        # [B, T, C]
    return F.elu(F.linear(h, w[p + "layer2.weight"], w[p + "layer2.bias"]))


# ----------------------------------------------------------------------------
# Style encoder, "attn" variant (modules.py:346-420, 445-612)
# ----------------------------------------------------------------------------
def positional_table(length, dim, dtype=torch.float32, timestep=10000.0):
    """modules.py:450-459: the table is built in float32 in the reference."""
    pos = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, dim, 2).float() * (-math.log(timestep) / dim))
    tab = torch.zeros(length, dim, dtype=torch.float32)
    tab[:, 0::2] = torch.sin(pos * div)
    tab[:, 1::2] = torch.cos(pos * div)
    return tab.to(dtype)


def _conv3(x, wt, b):
    """ConvNorm1D (modules.py:615-651): [B, L, Cin] -> [B, L, Cout], zero pad 1."""
    return F.conv1d(x.transpose(1, 2), wt, b, padding=1).transpose(1, 2)


def _mha(x, in_w, in_b, out_w, out_b, nheads=4):
    """nn.MultiheadAttention(E, 4) self-attention, no mask (modules.py:529-550).
    Packed in_proj rows ordered q, k, v; scores scaled by 1/sqrt(head_dim)."""
    B, L, E = x.shape
    hd = E // nheads
    qkv = F.linear(x, in_w, in_b)                                # [B, L, 3E]
    q, k, v = qkv[..., :E], qkv[..., E:2 * E], qkv[..., 2 * E:]
    q = q.reshape(B, L, nheads, hd).transpose(1, 2)              # [B, h, L, hd]
    k = k.reshape(B, L, nheads, hd).transpose(1, 2)
    v = v.reshape(B, L, nheads, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, L, E)
    return F.linear(o, out_w, out_b)


def style_encoder_attn(w, x, prefix="encoder."):
    """x: [B, L, 1134] normalised -> [B, Eout]   (eval mode)."""
    p = prefix
    h = F.relu(_conv3(x, w[p + "convs.0.conv.weight"], w[p + "convs.0.conv.bias"]))
    h = F.layer_norm(h, h.shape[-1:], w[p + "convs.2.weight"], w[p + "convs.2.bias"], 1e-5)
    h = F.relu(_conv3(h, w[p + "convs.4.conv.weight"], w[p + "convs.4.conv.bias"]))
    h = F.layer_norm(h, h.shape[-1:], w[p + "convs.6.weight"], w[p + "convs.6.bias"], 1e-5)
    L, E = h.shape[1], h.shape[2]
    h = h + positional_table(L, E, h.dtype)[None]                # modules.py:403,410
    b = p + "blocks.0."
    a = _mha(h, w[b + "attention.multi_head_attention.in_proj_weight"],
             w[b + "attention.multi_head_attention.in_proj_bias"],
             w[b + "attention.multi_head_attention.out_proj.weight"],
             w[b + "attention.multi_head_attention.out_proj.bias"])
    a = F.layer_norm(a + h, (E,), w[b + "attention.layer_norm.weight"],
                     w[b + "attention.layer_norm.bias"], 1e-5)   # modules.py:555
    f = F.relu(_conv3(a, w[b + "feed_forward.convs.0.conv.weight"],
                      w[b + "feed_forward.convs.0.conv.bias"]))
    f = _conv3(f, w[b + "feed_forward.convs.2.conv.weight"], w[b + "feed_forward.convs.2.conv.bias"])
    f = F.layer_norm(f + a, (E,), w[b + "feed_forward.layer_norm.weight"],
                     w[b + "feed_forward.layer_norm.bias"], 1e-5)  # modules.py:603
    return f.sum(dim=1) / L                                      # modules.py:416


def style_encoder_gru(w, x, prefix="encoder."):
    """StyleEncoderGRU.forward (modules.py:307-343): two conv(3)+ReLU, one bidirectional GRU layer, projection of
    the LAST time step (forward direction: final state; reverse direction: its first step, which saw x[-1] only)."""
    p = prefix
    h = torch.relu(_conv3(x, w[p + "convs.0.conv.weight"], w[p + "convs.0.conv.bias"]))
    h = torch.relu(_conv3(h, w[p + "convs.2.conv.weight"], w[p + "convs.2.conv.bias"]))
    B, L, Hh = h.shape
    hf = torch.zeros(B, Hh, dtype=h.dtype)
    for t in range(L):
        hf = gru_cell(h[:, t], hf, w[p + "rnn_layer.weight_ih_l0"], w[p + "rnn_layer.weight_hh_l0"],
                      w[p + "rnn_layer.bias_ih_l0"], w[p + "rnn_layer.bias_hh_l0"])
    hb = gru_cell(h[:, L - 1], torch.zeros(B, Hh, dtype=h.dtype), w[p + "rnn_layer.weight_ih_l0_reverse"],
                  w[p + "rnn_layer.weight_hh_l0_reverse"], w[p + "rnn_layer.bias_ih_l0_reverse"],
                  w[p + "rnn_layer.bias_hh_l0_reverse"])
    return F.linear(torch.cat([hf, hb], dim=-1), w[p + "projection_layer.linear_layer.weight"],
                    w[p + "projection_layer.linear_layer.bias"])


def style_encoder(w, x, eps, temperature=1.0, S=64):
    """StyleEncoder.forward with use_vae (modules.py:289-302); `eps` is the
    injected N(0,1) sample that the reference draws with randn_like."""
    out = style_encoder_gru(w, x) if "encoder.rnn_layer.weight_ih_l0" in w else style_encoder_attn(w, x)   # type
    mu, logvar = out[:, :S], out[:, S:]
    std = torch.exp(0.5 * logvar) / temperature
    return mu + eps * std, mu, logvar


# ----------------------------------------------------------------------------
# Decoder (modules.py:11-243, 677-742)
# ----------------------------------------------------------------------------
def vectorize_input(root_pos, root_rot, root_vel, root_vrt, lpos, ltxy, lvel, lvrt, gaze_pos,
                    in_mean, in_std):
    """modules.py:677-713 (gaze direction is NOT normalised there)."""
    B = lpos.shape[0]
    gaze_dir = quat_inv_mul_vec(root_rot, gaze_pos - root_pos)
    v = torch.cat([root_vel.reshape(B, -1), root_vrt.reshape(B, -1), lpos.reshape(B, -1),
                   ltxy.reshape(B, -1), lvel.reshape(B, -1), lvrt.reshape(B, -1),
                   gaze_dir.reshape(B, -1)], dim=1)
    return (v - in_mean) / in_std


def devectorize_output(pred, root_pos, root_rot, J, dt, out_mean, out_std):
    """modules.py:716-742"""
    B = pred.shape[0]
    p = pred * out_std + out_mean
    vel, vrt = p[:, 0:3], p[:, 3:6]
    lpos = p[:, 6:6 + 3 * J].reshape(B, J, 3)
    ltxy = p[:, 6 + 3 * J:6 + 9 * J].reshape(B, J, 2, 3)
    lvel = p[:, 6 + 9 * J:6 + 12 * J].reshape(B, J, 3)
    lvrt = p[:, 6 + 12 * J:6 + 15 * J].reshape(B, J, 3)
    new_pos = quat_mul_vec(root_rot, vel * dt) + root_pos
    new_rot = quat_mul(quat_from_helical(quat_mul_vec(root_rot, vrt * dt)), root_rot)
    return new_pos, new_rot, vel, vrt, lpos, ltxy, lvel, lvrt


def cell_state_encoder(w, pose, style, nlayers=2, prefix="cell_state_encoder."):
    """modules.py:238-243 -> [nlayers, B, H]"""
    p = prefix
    h = F.elu(F.linear(torch.cat([pose, style], dim=-1), w[p + "layer0.weight"], w[p + "layer0.bias"]))
    h = F.elu(F.linear(h, w[p + "layer1.weight"], w[p + "layer1.bias"]))
    o = F.linear(h, w[p + "layer2.weight"], w[p + "layer2.bias"])
    return o.reshape(o.shape[0], nlayers, -1).transpose(0, 1).contiguous()


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """One nn.GRU layer step; gate order r, z, n (modules.py:173-183)."""
    H = h.shape[-1]
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1.0 - z) * n + z * h


def recurrent_step(w, pose, speech, style, state, prefix="recurrent_decoder."):
    """RecurrentDecoderNormal.forward (modules.py:179-185); state [2, B, H]."""
    p = prefix
    x = torch.cat([pose, speech, style], dim=-1)
    hid = F.elu(F.linear(x, w[p + "layer0.weight"], w[p + "layer0.bias"]))
    g_in = torch.cat([hid, x], dim=-1)
    h0 = gru_cell(g_in, state[0], w[p + "layer1.weight_ih_l0"], w[p + "layer1.weight_hh_l0"],
                  w[p + "layer1.bias_ih_l0"], w[p + "layer1.bias_hh_l0"])
    h1 = gru_cell(h0, state[1], w[p + "layer1.weight_ih_l1"], w[p + "layer1.weight_hh_l1"],
                  w[p + "layer1.bias_ih_l1"], w[p + "layer1.bias_hh_l1"])
    out = F.linear(h1, w[p + "layer2.weight"], w[p + "layer2.bias"])
    return out, torch.stack([h0, h1], dim=0)


def recurrent_step_film(w, pose, speech, style, state, prefix="recurrent_decoder."):
    """RecurrentDecoderFiLM.forward (modules.py:213-227): gamma/beta from the style modulate both hidden layers."""
    p = prefix
    H = state.shape[-1]
    gam = F.linear(style, w[p + "gammas_predictor.linear_layer.weight"], w[p + "gammas_predictor.linear_layer.bias"]) + 1
    bet = F.linear(style, w[p + "betas_predictor.linear_layer.weight"], w[p + "betas_predictor.linear_layer.bias"])
    x = torch.cat([pose, speech], dim=-1)
    hid = F.elu(F.linear(x, w[p + "layer0.weight"], w[p + "layer0.bias"]))
    hid = hid * gam[:, :H] + bet[:, :H]
    h0 = gru_cell(torch.cat([hid, x], dim=-1), state[0], w[p + "layer1.weight_ih_l0"], w[p + "layer1.weight_hh_l0"],
                  w[p + "layer1.bias_ih_l0"], w[p + "layer1.bias_hh_l0"])
    h1 = gru_cell(h0, state[1], w[p + "layer1.weight_ih_l1"], w[p + "layer1.weight_hh_l1"],
                  w[p + "layer1.bias_ih_l1"], w[p + "layer1.bias_hh_l1"])
    h2 = F.elu(F.linear(h1, w[p + "layer2.weight"], w[p + "layer2.bias"]))
    h2 = h2 * gam[:, H:] + bet[:, H:]
    out = F.linear(h2, w[p + "layer3.weight"], w[p + "layer3.bias"])
    return out, torch.stack([h0, h1], dim=0)


def decoder_rollout(w, Z_root_pos, Z_root_rot, Z_root_vel, Z_root_vrt, Z_lpos, Z_ltxy, Z_lvel, Z_lvrt,
                    gaze_pos, speech, style, in_mean, in_std, out_mean, out_std, dt,
                    return_raw=False):
    """Decoder.forward (modules.py:47-162).  speech/style/gaze are [B, T, *];
    returns the 8 [B, T, ...] tensors, frame 0 = the given first pose."""
    T = speech.shape[1]
    J = Z_lpos.shape[1]
    cur = (Z_root_pos, Z_root_rot, Z_root_vel, Z_root_vrt, Z_lpos, Z_ltxy, Z_lvel, Z_lvrt)
    outs = [[c] for c in cur]
    state = cell_state_encoder(
        w, vectorize_input(*cur, gaze_pos[:, 0], in_mean, in_std), style[:, 0])
    raws = []
    for i in range(1, T):
        pose = vectorize_input(*cur, gaze_pos[:, i], in_mean, in_std)
        step = recurrent_step_film if "recurrent_decoder.layer3.weight" in w else recurrent_step   # rnn_cond
        pred, state = step(w, pose, speech[:, i], style[:, i], state)
        raws.append(pred)
        cur = devectorize_output(pred, cur[0], cur[1], J, dt, out_mean, out_std)
        for o, c in zip(outs, cur):
            o.append(c)
    res = tuple(torch.stack(o, dim=1) for o in outs)
    if return_raw:
        return res, (torch.stack(raws, dim=1) if raws else None), state
    return res
