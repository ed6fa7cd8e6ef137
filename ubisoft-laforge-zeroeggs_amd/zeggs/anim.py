"""Animation plumbing around the decoder: BVH text I/O on the host, feature extraction and BVH-channel
conversion on the device (csrc/anim.hip, float64).

Semantics follow the reference so that `generate_gesture()` consumes and produces the same files:
  ZEGGS/anim/bvh.py (BVH reader/writer), ZEGGS/data_pipeline.py:90-228 (preprocess_animation),
  ZEGGS/generate.py:389 + ZEGGS/utils.py:47-87 (two-axis rotations -> quaternions -> euler channels, write_bvh).
The NumPy restatement of the device kernels lives in oracle/anim.py (test infrastructure).
"""
import ctypes as C
import re

import numpy as np
import torch

from . import ops

_CHAN = {"Xrotation": "x", "Yrotation": "y", "Zrotation": "z"}
_CHAN_INV = {v: k for k, v in _CHAN.items()}


# ----------------------------------------------------------------------------- BVH
def bvh_load(filename):
    """-> dict(rotations [F,J,3] degrees, positions [F,J,3], offsets [J,3], parents [J], names, order, frametime)"""
    names, offsets, parents, chans = [], [], [], []
    stack, order, end_site = [], None, False
    with open(filename, "rb") as f:
        raw = f.read()
    # the MOTION section starts at a line that holds nothing but the keyword (a joint called LOCOMOTION_root is not it: ADVICE r4)
    msec = re.search(rb"(?m)^[ \t]*MOTION[ \t]*\r?$", raw)
    if msec is None:
        raise ValueError(f"{filename}: no MOTION section (a line holding only the keyword MOTION)")
    mpos, mend = msec.start(), msec.end()
    lines = iter(raw[:mpos].decode().splitlines() + ["MOTION"])
    for line in lines:
        tok = line.split()
        if not tok:
            continue
        if tok[0] in ("ROOT", "JOINT"):
            names.append(tok[1])
            offsets.append([0.0, 0.0, 0.0])
            parents.append(stack[-1] if stack else -1)
            chans.append(0)
            stack.append(len(names) - 1)
        elif tok[0] == "End":
            end_site = True
        elif tok[0] == "}":
            if end_site:
                end_site = False
            else:
                stack.pop()
        elif tok[0] == "OFFSET" and not end_site:
            offsets[stack[-1]] = [float(v) for v in tok[1:4]]
        elif tok[0] == "CHANNELS":
            n = int(tok[1])
            chans[stack[-1]] = n
            rot = [c for c in tok[2:2 + n] if c in _CHAN]
            if order is None and len(rot) == 3:
                order = "".join(_CHAN[c] for c in rot)
        elif tok[0] == "MOTION":
            break
    # MOTION block: "Frames: N", "Frame Time: dt", then N rows of numbers -- parsed by the library's host helper (strtod on a few
    # threads: 7 200 rows x 228 numbers in ~8 ms, numpy.loadtxt 76 ms), numpy.loadtxt when it declines (ragged rows: loadtxt
    # raises the error the caller expects)
    off = mend
    probe = raw[off:off + 256]
    m1 = re.match(rb"\s*Frames:\s+(\d+)\s*?\n", probe)
    if m1 is None:
        raise ValueError(f"{filename}: the MOTION section does not start with 'Frames: <count>'")
    nframes = int(m1.group(1))
    m2 = re.match(rb"\s*Frame Time:\s+([\d\.eE\-\+]+)[^\n]*\n?", probe[m1.end():])
    if m2 is None:
        raise ValueError(f"{filename}: no 'Frame Time: <seconds>' line after 'Frames:'")
    frametime = float(m2.group(1))
    off += m1.end() + m2.end()                          # the rows start here; `raw` (a bytes object) ends in a NUL
    ncols = sum(chans)
    data = None
    if nframes > 0 and ncols > 0 and off < len(raw):
        out = np.empty((nframes, ncols), dtype=np.float64)
        base = C.cast(C.c_char_p(raw), C.c_void_p).value
        if ops.lib().zeggs_parse_table_text(C.c_void_p(base + off), C.c_size_t(len(raw) - off), out.ctypes.data_as(C.c_void_p),
                                            C.c_long(nframes), int(ncols)) == 0:
            data = out
    body = None
    if data is None:
        rows = [ln for ln in raw[off:].decode().splitlines() if ln.strip()][:nframes]
        data = np.loadtxt(rows, dtype=np.float64, ndmin=2) if rows else np.zeros((0, 0))     # C parser
    J = len(names)
    offsets = np.asarray(offsets, dtype=np.float32)
    positions = np.repeat(offsets[None], len(data), axis=0)
    rotations = np.zeros((len(data), J, 3), dtype=np.float32)
    col = 0
    for j in range(J):
        if chans[j] == 6:
            positions[:, j] = data[:, col:col + 3]
            rotations[:, j] = data[:, col + 3:col + 6]
        elif chans[j] == 3:
            rotations[:, j] = data[:, col:col + 3]
        else:
            raise ValueError(f"unsupported CHANNELS {chans[j]} on joint {names[j]}")
        col += chans[j]
    return dict(rotations=rotations, positions=positions, offsets=offsets, parents=np.asarray(parents, np.int32),
                names=names, order=order or "zyx", frametime=frametime)


def bvh_header(offsets, parents, names=None, order="zyx", nframes=0, frametime=1.0 / 60.0):
    """-> (text of the HIERARCHY block + the MOTION head, joint order of the motion rows) as reference bvh.save writes them
    (root with 6 channels, every other joint with 3, translations=False)."""
    offsets = np.asarray(offsets)
    parents = list(parents)
    names = names or [f"joint_{i}" for i in range(len(parents))]
    rot_ch = " ".join(_CHAN_INV[c] for c in order)
    children = {i: [j for j, p in enumerate(parents) if p == i] for i in range(len(parents))}
    seq, out = [], ["HIERARCHY"]

    def emit(i, depth):
        seq.append(i)
        ind = "\t" * depth
        out.append(f"{ind}{'ROOT' if depth == 0 else 'JOINT'} {names[i]}")
        out.append(ind + "{")
        out.append("%s\tOFFSET %f %f %f" % ((ind,) + tuple(offsets[i])))
        if depth == 0:
            out.append(f"{ind}\tCHANNELS 6 Xposition Yposition Zposition {rot_ch} ")
        else:
            out.append(f"{ind}\tCHANNELS 3 {rot_ch}")
        if children[i]:
            for c in children[i]:
                emit(c, depth + 1)
        else:
            out.extend([f"{ind}\tEnd Site", ind + "\t{", "%s\t\tOFFSET %f %f %f" % (ind, 0.0, 0.0, 0.0), ind + "\t}"])
        out.append(ind + "}")

    emit(0, 0)
    out += ["MOTION", "Frames: %i" % nframes, "Frame Time: %f" % frametime]
    return "\n".join(out) + "\n", seq


def bvh_save(filename, data):
    """Writes root with 6 channels and every other joint with 3 (reference bvh.save, translations=False)."""
    rots, poss = np.asarray(data["rotations"]), np.asarray(data["positions"])
    head, seq = bvh_header(data["offsets"], data["parents"], data.get("names"), data.get("order", "zyx"), len(rots),
                           data.get("frametime", 1.0 / 60.0))
    # motion block: one row per frame = root position + the rotations in hierarchy order
    table = np.concatenate([np.asarray(poss[:, 0], np.float64).reshape(len(rots), 3)] +
                           [np.asarray(rots[:, j], np.float64).reshape(len(rots), 3) for j in seq], axis=1)
    with open(filename, "w") as fh:
        fh.write(head)
    # the motion block is formatted by the library's host helper (same correctly-rounded "%f" as numpy.savetxt gives, ten
    # times faster: a 30-minute clip is 24.6 M numbers)
    table = np.ascontiguousarray(table, dtype=np.float64)
    rc = ops.lib().zeggs_write_table_text(str(filename).encode(), 1, table.ctypes.data_as(C.c_void_p), C.c_long(table.shape[0]),
                                          int(table.shape[1]))
    if rc != 0:      # as np.savetxt / open() would: an OSError, which generate_gesture() reports and survives (generate.py:407)
        raise OSError("zeggs_write_table_text: " + ops.lib().zeggs_last_error().decode())


def format_rows(table):
    """host float64 [rows, cols] (C-contiguous; e.g. a view of a pinned staging buffer) -> bytes of the BVH motion rows
    (zeggs_format_table_text: releases the GIL, so several host threads format row blocks side by side)"""
    rows, cols = table.shape
    cap = rows * (cols * 24 + 1) + 400
    buf = C.create_string_buffer(cap)
    n = C.c_size_t(0)
    ptr = table.ctypes.data_as(C.c_void_p) if isinstance(table, np.ndarray) else C.c_void_p(table.data_ptr())
    rc = ops.lib().zeggs_format_table_text(ptr, C.c_long(rows), int(cols), buf, C.c_size_t(cap), C.byref(n))
    if rc != 0:
        raise RuntimeError("zeggs_format_table_text: " + ops.lib().zeggs_last_error().decode())
    return buf.raw[:n.value]


# ----------------------------------------------------------------------------- device kernels (csrc/anim.hip)
class AnimDims(C.Structure):       # mirrors ZeggsAnimDims
    _fields_ = [("N", C.c_int), ("J", C.c_int), ("hips", C.c_int), ("spine2", C.c_int), ("head", C.c_int),
                ("dt", C.c_double), ("order", C.c_int)]


def order_code(order):
    """'zyx' -> the packed channel order of ZeggsAnimDims / ZeggsBvhDims (axis of channel i in bits 2i..2i+1)"""
    if sorted(order) != ["x", "y", "z"]:
        raise ValueError(f"not a rotation channel order: {order!r}")
    return sum("xyz".index(c) << (2 * i) for i, c in enumerate(order))


ANIM_OUT = (("root_pos", 3, torch.float64), ("root_rot", 4, torch.float64), ("root_vel", 3, torch.float64),
            ("root_vrt", 3, torch.float64), ("lpos", -3, torch.float64), ("lrot", -4, torch.float64),
            ("lvel", -3, torch.float64), ("lvrt", -3, torch.float64), ("ltxy", -6, torch.float32),
            ("cpos", -3, torch.float64), ("crot", -4, torch.float64), ("cvel", -3, torch.float64),
            ("cvrt", -3, torch.float64), ("ctxy", -6, torch.float32), ("gaze_pos", 3, torch.float64),
            ("gaze_dir", 3, torch.float64))
AnimOut = type("AnimOut", (C.Structure,), {"_fields_": [(n, C.c_void_p) for n, _, _ in ANIM_OUT]})   # ZeggsAnimOut


class BvhDims(C.Structure):        # mirrors ZeggsBvhDims
    _fields_ = [("T", C.c_int), ("J", C.c_int), ("rebase", C.c_int), ("start_pos", C.c_double * 3),
                ("start_rot", C.c_double * 4), ("order", C.c_int)]


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def preprocess_animation(anim, device=None):
    """BVH dict -> the 16 feature arrays of reference data_pipeline.preprocess_animation, same order:
    (root_pos, root_rot, root_vel, root_vrt, lpos, lrot, ltxy, lvel, lvrt, cpos, crot, ctxy, cvel, cvrt, gaze_pos,
    gaze_dir) as DEVICE tensors (float64; ltxy / ctxy float32)."""
    device = torch.device(device or "cuda")
    names = list(anim["names"])
    rot = torch.as_tensor(np.ascontiguousarray(anim["rotations"], dtype=np.float64), device=device)
    pos = torch.as_tensor(np.ascontiguousarray(anim["positions"], dtype=np.float64), device=device)
    parents = torch.as_tensor(np.ascontiguousarray(anim["parents"], dtype=np.int32), device=device)
    N, J = rot.shape[0], rot.shape[1]
    d = AnimDims(N, J, names.index("Hips"), names.index("Spine2"), names.index("Head"), float(anim["frametime"]),
                 order_code(anim["order"]))      # any channel order (quat.from_euler takes any)
    L = ops.lib()
    L.zeggs_anim_features_workspace_bytes.restype = C.c_size_t
    ws = torch.empty(int(L.zeggs_anim_features_workspace_bytes(C.byref(d))), dtype=torch.uint8, device=device)
    out, ptr = {}, AnimOut()
    for n, w, dt in ANIM_OUT:
        shape = (N, w) if w > 0 else ((N, J, 2, 3) if w == -6 else (N, J, -w))
        out[n] = torch.empty(shape, dtype=dt, device=device)
        setattr(ptr, n, out[n].data_ptr())
    rc = L.zeggs_anim_features(C.byref(d), C.c_void_p(parents.data_ptr()), C.c_void_p(rot.data_ptr()),
                               C.c_void_p(pos.data_ptr()), C.byref(ptr), C.c_void_p(ws.data_ptr()),
                               C.c_size_t(ws.numel()), _stream())
    if rc != 0:
        raise RuntimeError("zeggs_anim_features: " + L.zeggs_last_error().decode())
    order = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "lrot", "ltxy", "lvel", "lvrt", "cpos", "crot",
             "ctxy", "cvel", "cvrt", "gaze_pos", "gaze_dir")
    return tuple(out[k] for k in order)


def _to_euler_order(order):
    if order not in ("zyx", "xzy"):      # what quat.to_euler implements (ZEGGS/anim/quat.py:111-127); the reference raises the same
        raise NotImplementedError("Cannot convert to ordering %s" % order)
    return order_code(order)


def bvh_channels(root_pos, root_rot, lpos, ltxy, start_position=None, start_rotation=None, order="zyx"):
    """decoder output (device float32: [T,3], [T,4], [T,J,3], [T,J,2,3]) -> (positions, euler degrees) float64
    device tensors [T,J,3], channel order `order`, root folded into joint 0 (reference generate.py:389, utils.py:47-87)."""
    T, J = lpos.shape[0], lpos.shape[1]
    d = BvhDims(T, J, 0)
    d.order = _to_euler_order(order)
    if start_position is not None and start_rotation is not None:
        d.rebase = 1
        d.start_pos[:] = [float(v) for v in start_position]
        d.start_rot[:] = [float(v) for v in start_rotation]
    f = lambda t: t.detach().to(torch.float32).contiguous()  # noqa: E731
    root_pos, root_rot, lpos, ltxy = f(root_pos), f(root_rot), f(lpos), f(ltxy)
    positions = torch.empty(T, J, 3, dtype=torch.float64, device=lpos.device)
    euler = torch.empty(T, J, 3, dtype=torch.float64, device=lpos.device)
    L = ops.lib()
    rc = L.zeggs_pose_to_bvh(C.byref(d), C.c_void_p(root_pos.data_ptr()), C.c_void_p(root_rot.data_ptr()),
                             C.c_void_p(lpos.data_ptr()), C.c_void_p(ltxy.data_ptr()),
                             C.c_void_p(positions.data_ptr()), C.c_void_p(euler.data_ptr()), _stream())
    if rc != 0:
        raise RuntimeError("zeggs_pose_to_bvh: " + L.zeggs_last_error().decode())
    return positions, euler


def write_bvh(filename, root_pos, root_rot, lpos, ltxy, parents, names, order, dt, start_position=None,
              start_rotation=None):
    """reference utils.write_bvh, fed with the decoder's two-axis rotations (the quaternion / euler conversion of
    generate.py:389 happens on the device)."""
    positions, euler = bvh_channels(root_pos, root_rot, lpos, ltxy, start_position, start_rotation, order)
    write_bvh_channels(filename, positions, euler, parents, names, order, dt)


def write_bvh_channels(filename, positions, euler, parents, names, order, dt):
    """the file half of write_bvh: device channel tables (bvh_channels) -> BVH text"""
    positions = positions.cpu().numpy()
    bvh_save(filename, dict(order=order, offsets=positions[0], names=names, frametime=dt, parents=parents,
                            positions=positions, rotations=euler.cpu().numpy()))
