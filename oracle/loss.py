"""Oracle restatement of the training loss (torch CPU, differentiable).

TEST INFRASTRUCTURE -- see oracle/__init__.py.

Reference: ZEGGS/train.py:276-421 (FK + 17 weighted L1 terms + KL, sum / 18),
ZEGGS/anim/txform.py:10-34 (xform_fk_vel, xform_orthogonalize_from_xy),
ZEGGS/modules.py:673 (normalize), :745-789 (KL weight / KL divergence).
"""
import math

import torch

from .nets import _cross, quat_inv_mul_vec, quat_mul_vec, quat_to_xform

LOSS_NAMES = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "lrot", "lvel", "lvrt",
              "cpos", "crot", "cvel", "cvrt", "ldvl", "ldvt", "cdvl", "cdvt", "gaze", "kl")
LOSS_WEIGHTS = (0.1, 10.0, 0.1, 5.0, 15.0, 15.0, 10.0, 7.0,
                0.1, 3.0, 0.06, 1.25, 7.0, 8.0, 0.06, 1.25, 10.0)


def _norm(x):
    return torch.sqrt(torch.sum(x * x, dim=-1, keepdim=True))


def orthogonalize_from_xy(xy, eps=1e-10):
    """txform.py:23-34: xy [..., 2, 3] -> rotation matrices [..., 3, 3] whose
    COLUMNS are the normalised x, y, z axes."""
    x = xy[..., 0, :]
    z = _cross(x, xy[..., 1, :])
    y = _cross(z, x)
    rows = torch.stack([x / (_norm(x) + eps), y / (_norm(y) + eps), z / (_norm(z) + eps)], dim=-2)
    return rows.transpose(-1, -2)


def fk_vel(lmat, lpos, lvrt, lvel, parents):
    """txform.py:10-20; joints axis is -3 for lmat, -2 for vectors."""
    gr, gp, gt, gv = [lmat[..., 0, :, :]], [lpos[..., 0, :]], [lvrt[..., 0, :]], [lvel[..., 0, :]]

    def mv(m, v):
        return torch.matmul(m, v[..., None])[..., 0]

    for i in range(1, len(parents)):
        p = int(parents[i])
        rp = mv(gr[p], lpos[..., i, :])
        gp.append(gp[p] + rp)
        gr.append(torch.matmul(gr[p], lmat[..., i, :, :]))
        gt.append(gt[p] + mv(gr[p], lvrt[..., i, :]))
        gv.append(gv[p] + mv(gr[p], lvel[..., i, :]) + _cross(gt[p], rp))
    return (torch.stack(gr, dim=-3), torch.stack(gp, dim=-2),
            torch.stack(gt, dim=-2), torch.stack(gv, dim=-2))


def kl_weight(iteration):
    """modules.py:745-761,773-788: min(logistic(0.005 (it - 7500)), 0.2)"""
    v = 1.0 / (1.0 + math.exp(-0.005 * (iteration - 7500)))
    return min(v, 0.2)


def _world(root_pos, root_rot, root_vel, root_vrt, lpos, ltxy, lvel, lvrt):
    """train.py:277-322 for one side (O_ or W_)."""
    lmat = orthogonalize_from_xy(ltxy)
    # root velocities rotated by the PREVIOUS frame's root rotation (frame 0 by itself)
    prev_rot = torch.cat([root_rot[:, 0:1], root_rot[:, :-1]], dim=1)
    rvel = quat_mul_vec(prev_rot, root_vel)
    rvrt = quat_mul_vec(prev_rot, root_vrt)
    r_lpos0 = quat_mul_vec(root_rot, lpos[:, :, 0])
    lpos0 = r_lpos0 + root_pos
    lmat0 = torch.matmul(quat_to_xform(root_rot), lmat[:, :, 0])
    lvel0 = rvel + quat_mul_vec(root_rot, lvel[:, :, 0]) + _cross(rvrt, r_lpos0)
    lvrt0 = rvrt + quat_mul_vec(root_rot, lvrt[:, :, 0])
    lpos = torch.cat([lpos0.unsqueeze(2), lpos[:, :, 1:]], dim=2)
    lmat = torch.cat([lmat0.unsqueeze(2), lmat[:, :, 1:]], dim=2)
    lvel = torch.cat([lvel0.unsqueeze(2), lvel[:, :, 1:]], dim=2)
    lvrt = torch.cat([lvrt0.unsqueeze(2), lvrt[:, :, 1:]], dim=2)
    return rvel, rvrt, lpos, lmat, lvel, lvrt


def training_loss(O, W, gaze_pos, parents, dt, mu=None, logvar=None, iteration=0):
    """O, W: 8-tuples (root_pos, root_rot, root_vel, root_vrt, lpos, ltxy, lvel, lvrt)
    of [B, T, ...] tensors (prediction / ground truth).  Returns (loss, terms[18])."""
    O_rvel, O_rvrt, O_lpos, O_lmat, O_lvel, O_lvrt = _world(*O)
    W_rvel, W_rvrt, W_lpos, W_lmat, W_lvel, W_lvrt = _world(*W)
    O_cmat, O_cpos, O_cvrt, O_cvel = fk_vel(O_lmat, O_lpos, O_lvrt, O_lvel, parents)
    W_cmat, W_cpos, W_cvrt, W_cvel = fk_vel(W_lmat, W_lpos, W_lvrt, W_lvel, parents)
    O_rmat, W_rmat = quat_to_xform(O[1]), quat_to_xform(W[1])

    def normalize(x, eps=1e-8):                                   # modules.py:673
        return x / (_norm(x) + eps)

    W_gaze = quat_inv_mul_vec(W[1], normalize(gaze_pos - W[0]))    # train.py:336
    O_gaze = quat_inv_mul_vec(O[1], normalize(gaze_pos - O[0]))    # train.py:337

    def l1(w, a, b):
        return torch.mean(torch.abs(w * (a - b)))

    def dl1(w, a, b):                                              # train.py:355-393
        return torch.mean(torch.abs(w * ((a[:, 1:] - a[:, :-1]) / dt - (b[:, 1:] - b[:, :-1]) / dt)))

    wt = LOSS_WEIGHTS
    terms = [
        l1(wt[0], O[0], W[0]), l1(wt[1], O_rmat, W_rmat), l1(wt[2], O_rvel, W_rvel),
        l1(wt[3], O_rvrt, W_rvrt),
        l1(wt[4], O_lpos, W_lpos), l1(wt[5], O[5], W[5]), l1(wt[6], O_lvel, W_lvel),
        l1(wt[7], O_lvrt, W_lvrt),
        l1(wt[8], O_cpos, W_cpos), l1(wt[9], O_cmat, W_cmat), l1(wt[10], O_cvel, W_cvel),
        l1(wt[11], O_cvrt, W_cvrt),
        dl1(wt[12], O_lpos, W_lpos), dl1(wt[13], O[5], W[5]), dl1(wt[14], O_cpos, W_cpos),
        dl1(wt[15], O_cmat, W_cmat),
        l1(wt[16], O_gaze, W_gaze),
    ]
    if mu is not None and logvar is not None:                      # train.py:397-400
        kl = torch.mean(-0.5 * torch.mean(1 + logvar - mu.pow(2) - logvar.exp(), dim=1))
        terms.append(kl_weight(iteration) * kl)
    else:
        terms.append(torch.zeros((), dtype=O[0].dtype))
    loss = sum(terms) / 18.0
    return loss, torch.stack([t.detach() for t in terms])
