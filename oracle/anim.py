"""CPU restatement (NumPy, float64) of the animation pre-/post-processing either side of the decoder.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product path runs csrc/anim.hip through
zeggs.anim.preprocess_animation / zeggs.anim.write_bvh; this file is the checker those kernels are compared with.

Follows the reference:
  ZEGGS/anim/quat.py            quaternion conventions (w first): mul, inv, mul_vec, abs, log/to_helical, between,
                                from_euler, to_euler, unroll, from_xform (:111-206), fk, fk_vel
  ZEGGS/data_pipeline.py:90-228 preprocess_animation
  ZEGGS/anim/txform.py:23-34    xform_orthogonalize_from_xy
  ZEGGS/utils.py:47-87          write_bvh (root re-basing, root folded into joint 0, euler channels)
Pinned: tests/test_oracle_golden.py checks preprocess_animation and the BVH channels against vectors produced by
the unmodified reference (tests/golden/generate.npz, written by oracle/make_golden.py).
"""
import numpy as np

_AXES = {"x": np.array([1.0, 0.0, 0.0]), "y": np.array([0.0, 1.0, 0.0]), "z": np.array([0.0, 0.0, 1.0])}


# ----------------------------------------------------------------------------- quaternions (w, x, y, z)
def q_mul(a, b):
    aw, av = a[..., :1], a[..., 1:]
    bw, bv = b[..., :1], b[..., 1:]
    return np.concatenate([aw * bw - np.sum(av * bv, axis=-1, keepdims=True),
                           aw * bv + bw * av + np.cross(av, bv)], axis=-1)


def q_inv(q):
    return q * np.array([1.0, -1.0, -1.0, -1.0], dtype=q.dtype)


def q_mul_vec(q, v):
    t = 2.0 * np.cross(q[..., 1:], v)
    return v + q[..., :1] * t + np.cross(q[..., 1:], t)


def q_abs(q):
    return np.where(q[..., :1] > 0.0, q, -q)


def q_normalize(q, eps=0.0):
    return q / (np.linalg.norm(q, axis=-1, keepdims=True) + eps)


def q_log(q, eps=1e-5):
    n = np.linalg.norm(q[..., 1:], axis=-1, keepdims=True)
    scale = np.where(n < eps, 1.0, np.arctan2(n, q[..., :1]) / np.where(n < eps, 1.0, n))
    return scale * q[..., 1:]


def q_to_helical(q, eps=1e-5):
    return 2.0 * q_log(q, eps)


def q_between(a, b):
    """rotation taking direction a to direction b (un-normalised), reference quat.between"""
    w = np.sqrt(np.sum(a * a, axis=-1) * np.sum(b * b, axis=-1)) + np.sum(a * b, axis=-1)
    return np.concatenate([w[..., None], np.cross(a, b)], axis=-1)


def q_from_angle_axis(angle, axis):
    h = 0.5 * angle[..., None]
    return np.concatenate([np.cos(h), np.sin(h) * axis], axis=-1)


def q_from_euler(e, order="zyx"):
    """e in radians, intrinsic order as in the reference (quat.from_euler): q0 * (q1 * q2)"""
    qs = [q_from_angle_axis(e[..., i], _AXES[order[i]]) for i in range(3)]
    return q_mul(qs[0], q_mul(qs[1], qs[2]))


def q_to_euler(q, order="zyx"):
    """quat.to_euler (ZEGGS/anim/quat.py:111-127): the two orders the reference implements"""
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    if order == "xzy":
        return np.stack([np.arctan2(2.0 * (x * w - y * z), -x * x + y * y - z * z + w * w),
                         np.arctan2(2.0 * (y * w - x * z), x * x - y * y - z * z + w * w),
                         np.arcsin(np.clip(2.0 * (x * y + z * w), -1.0, 1.0))], axis=-1)
    if order != "zyx":
        raise NotImplementedError("Cannot convert to ordering %s" % order)
    return np.stack([np.arctan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z)),
                     np.arcsin(np.clip(2.0 * (w * y - z * x), -1.0, 1.0)),
                     np.arctan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))], axis=-1)


def q_unroll(q):
    """make consecutive frames sign-continuous"""
    out = q.copy()
    for i in range(1, len(out)):
        flip = np.sum(out[i] * out[i - 1], axis=-1) < 0.0
        out[i][flip] = -out[i][flip]
    return out


def q_from_xform(m, eps=1e-10):
    """rotation matrices [..., 3, 3] -> quaternions (branch choice as reference quat.from_xform)"""
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    tr = m00 + m11 + m22
    sw = 0.5 / np.sqrt(np.maximum(tr + 1.0, eps))
    sx = 2.0 * np.sqrt(np.maximum(1.0 + m00 - m11 - m22, eps))
    sy = 2.0 * np.sqrt(np.maximum(1.0 + m11 - m00 - m22, eps))
    sz = 2.0 * np.sqrt(np.maximum(1.0 + m22 - m00 - m11, eps))
    a, b, c = m[..., 2, 1] - m[..., 1, 2], m[..., 0, 2] - m[..., 2, 0], m[..., 1, 0] - m[..., 0, 1]
    p, q, r = m[..., 0, 1] + m[..., 1, 0], m[..., 0, 2] + m[..., 2, 0], m[..., 1, 2] + m[..., 2, 1]
    cand = [np.stack([0.25 / sw, sw * a, sw * b, sw * c], axis=-1),
            np.stack([a / sx, 0.25 * sx, p / sx, q / sx], axis=-1),
            np.stack([b / sy, p / sy, 0.25 * sy, r / sy], axis=-1),
            np.stack([c / sz, q / sz, r / sz, 0.25 * sz], axis=-1)]
    x_big = (m00 > m11) & (m00 > m22)
    y_big = ~x_big & (m11 > m22)
    case = np.where(tr > 0, 0, np.where(x_big, 1, np.where(y_big, 2, 3)))[..., None]
    return np.select([case == 0, case == 1, case == 2, case == 3], cand)


def q_fk(lrot, lpos, parents):
    grot, gpos = [lrot[..., 0, :]], [lpos[..., 0, :]]
    for i in range(1, len(parents)):
        p = parents[i]
        gpos.append(q_mul_vec(grot[p], lpos[..., i, :]) + gpos[p])
        grot.append(q_mul(grot[p], lrot[..., i, :]))
    return np.stack(grot, axis=-2), np.stack(gpos, axis=-2)


def q_fk_vel(lrot, lpos, lvrt, lvel, parents):
    gr, gp, gt, gv = [lrot[..., 0, :]], [lpos[..., 0, :]], [lvrt[..., 0, :]], [lvel[..., 0, :]]
    for i in range(1, len(parents)):
        p = parents[i]
        rp = q_mul_vec(gr[p], lpos[..., i, :])
        gp.append(rp + gp[p])
        gr.append(q_mul(gr[p], lrot[..., i, :]))
        gt.append(gt[p] + q_mul_vec(gr[p], lvrt[..., i, :]))
        gv.append(gv[p] + q_mul_vec(gr[p], lvel[..., i, :]) + np.cross(gt[p], rp))
    return np.stack(gr, axis=-2), np.stack(gp, axis=-2), np.stack(gt, axis=-2), np.stack(gv, axis=-2)


# ----------------------------------------------------------------------------- features of an exemplar clip
def _finite_diff_first(x):
    """frame 0 by linear extrapolation of the next differences (data_pipeline.py:150-156)"""
    x[0] = x[1] - (x[3] - x[2])
    return x


def preprocess_animation(anim):
    """BVH dict -> the 16 feature arrays of reference data_pipeline.preprocess_animation (same order)."""
    names, parents, dt = anim["names"], anim["parents"], anim["frametime"]
    n = len(anim["rotations"])
    lrot = q_unroll(q_from_euler(np.radians(anim["rotations"].astype(np.float64)), anim["order"]))
    lpos = anim["positions"].astype(np.float64).copy()
    grot, gpos = q_fk(lrot, lpos, parents)
    fwd = np.array([[0.0, 0.0, 1.0]])
    root_pos = gpos[:, names.index("Spine2")] * np.array([1.0, 0.0, 1.0])
    root_fwd = q_mul_vec(grot[:, names.index("Hips")], fwd)
    root_fwd[:, 1] = 0.0
    root_fwd /= np.linalg.norm(root_fwd, axis=-1, keepdims=True)
    root_rot = q_normalize(q_between(np.repeat(fwd, n, axis=0), root_fwd))
    look = q_mul_vec(grot[:, names.index("Head")], fwd[0])
    look[:, 1] = 0.0
    look /= np.linalg.norm(look, axis=-1, keepdims=True)
    gaze_pos = np.repeat(np.median(root_pos + 100.0 * look, axis=0)[None], n, axis=0)
    inv_root = q_inv(root_rot)
    gaze_dir = q_mul_vec(inv_root, gaze_pos - root_pos)
    lrot[:, 0] = q_mul(inv_root, lrot[:, 0])
    lpos[:, 0] = q_mul_vec(inv_root, lpos[:, 0] - root_pos)

    lvel = np.zeros_like(lpos)
    lvel[1:] = (lpos[1:] - lpos[:-1]) / dt
    _finite_diff_first(lvel)
    lvrt = np.zeros_like(lpos)
    lvrt[1:] = q_to_helical(q_abs(q_mul(lrot[1:], q_inv(lrot[:-1])))) / dt
    _finite_diff_first(lvrt)
    root_vrt = np.zeros_like(root_pos)
    root_vrt[1:] = q_to_helical(q_abs(q_mul(root_rot[1:], q_inv(root_rot[:-1])))) / dt
    _finite_diff_first(root_vrt)
    root_vrt[1:] = q_mul_vec(inv_root[:-1], root_vrt[1:])
    root_vrt[0] = q_mul_vec(inv_root[0], root_vrt[0])
    root_vel = np.zeros_like(root_pos)
    root_vel[1:] = (root_pos[1:] - root_pos[:-1]) / dt
    _finite_diff_first(root_vel)
    root_vel[1:] = q_mul_vec(inv_root[:-1], root_vel[1:])
    root_vel[0] = q_mul_vec(inv_root[0], root_vel[0])

    crot, cpos, cvrt, cvel = q_fk_vel(lrot, lpos, lvrt, lvel, parents)
    ex, ey = np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])
    ltxy = np.stack([q_mul_vec(lrot, ex), q_mul_vec(lrot, ey)], axis=-2).astype(np.float32)
    ctxy = np.stack([q_mul_vec(crot, ex), q_mul_vec(crot, ey)], axis=-2).astype(np.float32)
    return (root_pos, root_rot, root_vel, root_vrt, lpos, lrot, ltxy, lvel, lvrt, cpos, crot, ctxy, cvel, cvrt,
            gaze_pos, gaze_dir)


def xform_from_xy(xy, eps=1e-10):
    """two-axis rows [..., 2, 3] -> rotation matrices (columns = axes), reference txform.py:23-34 in NumPy"""
    x = xy[..., 0, :]
    z = np.cross(x, xy[..., 1, :])
    y = np.cross(z, x)
    rows = np.stack([v / (np.linalg.norm(v, axis=-1, keepdims=True) + eps) for v in (x, y, z)], axis=-2)
    return np.swapaxes(rows, -1, -2)


def bvh_channels(root_pos, root_rot, lpos, ltxy, start_position=None, start_rotation=None, order="zyx"):
    """decoder output -> (positions [T,J,3], euler degrees [T,J,3]) as written by reference utils.write_bvh after
    generate.py:389 turned the two-axis encodings into quaternions."""
    lrot = q_from_xform(xform_from_xy(np.asarray(ltxy, np.float64)))
    root_pos, root_rot = np.asarray(root_pos, np.float64), np.asarray(root_rot, np.float64)
    if start_position is not None and start_rotation is not None:
        p0, r0 = root_pos[0:1].copy(), root_rot[0:1].copy()
        root_pos = q_mul_vec(q_inv(r0), root_pos - p0)
        root_rot = q_mul(q_inv(r0), root_rot)
        sr = np.asarray(start_rotation, np.float64)[None]
        root_pos = q_mul_vec(sr, root_pos) + np.asarray(start_position, np.float64)[None]
        root_rot = q_mul(sr, root_rot)
    lpos = np.array(lpos, np.float64)
    lpos[:, 0] = q_mul_vec(root_rot, lpos[:, 0]) + root_pos
    lrot[:, 0] = q_mul(root_rot, lrot[:, 0])
    return lpos, np.degrees(q_to_euler(lrot, order=order))
