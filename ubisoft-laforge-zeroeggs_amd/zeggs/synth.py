"""Synthetic ZeroEGGS-shaped data (no dataset ships with the reference: the
`data/Zeggs_data.z*` files are LFS pointers), used by bench.py, the tests and
the golden-vector generator.

Produces exactly the on-disk formats the reference consumes:
  processed_data.npz  keys as written by ZEGGS/data_pipeline.py:650-671
  data_definition.json keys as written by ZEGGS/data_pipeline.py:686-691
  stats.npz            keys as read by ZEGGS/generate.py:108-127
The skeleton (75 joints, parent table) is the one declared in
data/processed_v1/data_definition.json of the reference.
"""
import json
from pathlib import Path

import numpy as np

# parent table / joint names of the ZeroEGGS skeleton (data_definition.json)
PARENTS = [-1, 0, 1, 2, 3, 4, 5, 6, 7, 4, 9, 10, 11, 12, 13, 14, 15, 12, 17, 18, 19, 12, 21, 22,
           23, 12, 25, 26, 27, 12, 29, 30, 31, 12, 11, 4, 35, 36, 37, 38, 39, 40, 41, 38, 43, 44,
           45, 38, 47, 48, 49, 38, 51, 52, 53, 38, 55, 56, 57, 38, 37, 0, 61, 62, 63, 64, 63, 62,
           0, 68, 69, 70, 71, 70, 69]


def _names():
    n = ["Hips", "Spine", "Spine1", "Spine2", "Spine3", "Neck", "Neck1", "Head", "HeadEnd"]
    fingers = ["Thumb", "Index", "Middle", "Ring", "Pinky"]
    for side in ("Right", "Left"):
        n += [side + "Shoulder", side + "Arm", side + "ForeArm", side + "Hand"]
        for f in fingers:
            n += [f"{side}Hand{f}{i}" for i in range(1, 5)]
        n += [side + "ForeArmEnd", side + "ArmEnd"]
    for side in ("Right", "Left"):
        n += [side + "UpLeg", side + "Leg", side + "Foot", side + "ToeBase", side + "ToeBaseEnd",
              side + "LegEnd", side + "UpLegEnd"]
    return n


BONE_NAMES = _names()
assert len(BONE_NAMES) == len(PARENTS) == 75
LABEL_NAMES = ["Laughing", "Agreement", "Sad", "Distracted", "Speech", "Happy", "Angry", "Still",
               "Scared", "Flirty", "Disagreement", "Tired", "Sneaky", "Old", "Threatening",
               "Neutral", "Relaxed", "Pensive", "Sarcastic"]
DT = 0.016667
NJ = 75
POSE_IN = 6 + 15 * NJ + 3      # 1134
POSE_OUT = 6 + 15 * NJ         # 1131
N_AUDIO = 81


def _smooth(rng, n, d, scale, k=31):
    x = rng.standard_normal((n + k, d))
    ker = np.hanning(k)
    ker /= ker.sum()
    y = np.stack([np.convolve(x[:, i], ker, mode="valid")[:n] for i in range(d)], axis=1)
    return (y * scale * np.sqrt(k / 2.0)).astype(np.float64)


def _quat_exp(v):
    """helical/2 -> unit quaternion (w, x, y, z)"""
    h = np.linalg.norm(v, axis=-1, keepdims=True)
    s = np.where(h < 1e-8, 1.0, np.sin(h) / np.maximum(h, 1e-8))
    return np.concatenate([np.cos(h), v * s], axis=-1)


def _quat_mul_vec(q, v):
    t = 2.0 * np.cross(q[..., 1:], v)
    return v + q[..., 0:1] * t + np.cross(q[..., 1:], t)


def make_stats(seed=7):
    """Normalisation statistics shaped like the reference's stats.npz.  As in
    the real file, anim_output_std contains exact zeros (constant channels)."""
    rng = np.random.default_rng(seed)
    audio_mean = np.concatenate([rng.uniform(0.02, 0.1, 80), [9.1]]).astype(np.float32)
    in_mean = rng.normal(0.0, 1.0, POSE_IN).astype(np.float32)
    in_std = rng.uniform(0.5, 2.0, POSE_IN).astype(np.float32)
    out_mean = in_mean[:POSE_OUT].copy()
    out_std = rng.uniform(0.3, 1.5, POSE_OUT).astype(np.float32)
    out_std[rng.choice(POSE_OUT, 40, replace=False)] = 0.0
    return dict(audio_input_mean=audio_mean, audio_input_std=np.float32(1.0132982),
                anim_input_mean=in_mean, anim_input_std=in_std,
                anim_output_mean=out_mean, anim_output_std=out_std)


def make_clip(nframes, seed=0, stats=None):
    """One synthetic 60-fps clip: dict of float32 arrays with the Y_*/X_* keys."""
    rng = np.random.default_rng(seed)
    stats = stats or make_stats()
    n = nframes
    yaw = _smooth(rng, n, 1, 0.4)[:, 0]
    root_rot = np.stack([np.cos(yaw / 2), np.zeros(n), np.sin(yaw / 2), np.zeros(n)], axis=1)
    root_pos = np.cumsum(_smooth(rng, n, 3, 0.5) * np.array([1.0, 0.0, 1.0]) * DT * 30, axis=0)
    root_vel = np.zeros((n, 3))
    root_vel[1:] = (root_pos[1:] - root_pos[:-1]) / DT
    root_vel[0] = root_vel[1]
    inv = root_rot * np.array([1, -1, -1, -1.0])
    root_vel[1:] = _quat_mul_vec(inv[:-1], root_vel[1:])
    root_vrt = _smooth(rng, n, 3, 0.2) * np.array([0.0, 1.0, 0.0])
    hel = _smooth(rng, n * NJ, 3, 0.35).reshape(n, NJ, 3)
    lrot = _quat_exp(hel / 2.0)
    ex = np.zeros((n, NJ, 3)); ex[..., 0] = 1.0
    ey = np.zeros((n, NJ, 3)); ey[..., 1] = 1.0
    ltxy = np.stack([_quat_mul_vec(lrot, ex), _quat_mul_vec(lrot, ey)], axis=2)
    offsets = rng.normal(0, 8.0, (1, NJ, 3))
    lpos = offsets + _smooth(rng, n, NJ * 3, 0.05).reshape(n, NJ, 3)
    lvel = np.zeros_like(lpos)
    lvel[1:] = (lpos[1:] - lpos[:-1]) / DT
    lvel[0] = lvel[1]
    lvrt = np.zeros_like(lpos)
    lvrt[1:] = (hel[1:] - hel[:-1]) / DT
    lvrt[0] = lvrt[1]
    gaze = np.tile(np.array([[10.0, 150.0, 100.0]]), (n, 1)) + rng.normal(0, 1.0, (1, 3))
    audio = stats["audio_input_mean"][None] + rng.standard_normal((n, N_AUDIO)) * \
        np.concatenate([np.full(80, 0.02), [1.0]])[None]
    f = np.float32
    return dict(X_audio_features=audio.astype(f), Y_root_pos=root_pos.astype(f),
                Y_root_rot=root_rot.astype(f), Y_root_vel=root_vel.astype(f),
                Y_root_vrt=root_vrt.astype(f), Y_lpos=lpos.astype(f), Y_ltxy=ltxy.astype(f),
                Y_lvel=lvel.astype(f), Y_lvrt=lvrt.astype(f), Y_gaze_pos=gaze.astype(f))


def make_clip_stats(nframes, seed, stats, amp=0.8):
    """A clip drawn AROUND GIVEN normalisation statistics (e.g. the reference's real data/processed_v*/stats.npz):
    every pose channel is mean + std * smooth noise, so that normalised inputs are O(1) whatever the dynamic
    range of the statistics is, and channels with anim_output_std == 0 are exactly constant -- as in the real
    dataset.  Root trajectory / gaze / audio as in make_clip."""
    rng = np.random.default_rng(seed)
    n = nframes
    om = np.asarray(stats["anim_output_mean"], np.float64)
    osd = np.asarray(stats["anim_output_std"], np.float64)
    pose = om[None] + osd[None] * _smooth(rng, n, POSE_OUT, amp)
    yaw = _smooth(rng, n, 1, 0.4)[:, 0]
    root_rot = np.stack([np.cos(yaw / 2), np.zeros(n), np.sin(yaw / 2), np.zeros(n)], axis=1)
    root_pos = np.cumsum(_smooth(rng, n, 3, 0.5) * np.array([1.0, 0.0, 1.0]) * DT * 30, axis=0)
    gaze = np.tile(np.array([[10.0, 150.0, 100.0]]), (n, 1)) + rng.normal(0, 1.0, (1, 3)) + _smooth(rng, n, 3, 2.0)
    am = np.asarray(stats["audio_input_mean"], np.float64)
    audio = am[None] + rng.standard_normal((n, N_AUDIO)) * np.concatenate([np.full(80, 0.02), [1.0]])[None]
    f = np.float32
    o = [0, 3, 6, 6 + 3 * NJ, 6 + 9 * NJ, 6 + 12 * NJ, POSE_OUT]
    return dict(X_audio_features=audio.astype(f), Y_root_pos=root_pos.astype(f), Y_root_rot=root_rot.astype(f),
                Y_root_vel=pose[:, o[0]:o[1]].astype(f), Y_root_vrt=pose[:, o[1]:o[2]].astype(f),
                Y_lpos=pose[:, o[2]:o[3]].reshape(n, NJ, 3).astype(f),
                Y_ltxy=pose[:, o[3]:o[4]].reshape(n, NJ, 2, 3).astype(f),
                Y_lvel=pose[:, o[4]:o[5]].reshape(n, NJ, 3).astype(f),
                Y_lvrt=pose[:, o[5]:o[6]].reshape(n, NJ, 3).astype(f), Y_gaze_pos=gaze.astype(f))


def make_processed(n_train, n_valid, nframes, seed=0, nlabels=19, stats=None, clip_fn=None):
    """Concatenate clips into the processed_data.npz layout (dict of arrays)."""
    stats = stats or make_stats()
    clip_fn = clip_fn or (lambda n, seed, stats: make_clip(n, seed=seed, stats=stats))
    clips = [clip_fn(nframes, seed=seed * 1000 + i, stats=stats) for i in range(n_train + n_valid)]
    data = {k: np.concatenate([c[k] for c in clips], axis=0) for k in clips[0]}
    bounds = np.arange(n_train + n_valid + 1) * nframes
    rng = np.random.default_rng(seed + 99)
    data["ranges_train"] = np.stack([bounds[:n_train], bounds[1:n_train + 1]], axis=1).astype(np.int32)
    data["ranges_valid"] = np.stack([bounds[n_train:-1], bounds[n_train + 1:]], axis=1).astype(np.int32)
    data["ranges_train_labels"] = rng.integers(0, nlabels, n_train).astype(np.int32)
    data["ranges_valid_labels"] = rng.integers(0, nlabels, n_valid).astype(np.int32)
    data.update(stats)
    return data


def data_definition(nlabels=19):
    return dict(dt=DT, label_names=LABEL_NAMES[:nlabels], parents=PARENTS, bone_names=BONE_NAMES)


def write_dataset(directory, n_train=2, n_valid=1, nframes=600, seed=0, nlabels=19, stats=None, clip_fn=None):
    """Write processed_data.npz + data_definition.json + stats.npz into `directory`."""
    d = Path(directory)
    d.mkdir(parents=True, exist_ok=True)
    stats = stats or make_stats()
    data = make_processed(n_train, n_valid, nframes, seed, nlabels, stats, clip_fn)
    np.savez(d / "processed_data.npz", **data)
    np.savez(d / "stats.npz", **stats)
    with open(d / "data_definition.json", "w") as f:
        json.dump(data_definition(nlabels), f)
    return d / "processed_data.npz", d / "data_definition.json"


def synth_wav(n_samples, seed=0, fs=16000):
    """Band-limited noise with a 4 Hz syllabic envelope, int16 mono."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(n_samples)
    spec = np.fft.rfft(x)
    fr = np.fft.rfftfreq(n_samples, 1.0 / fs)
    spec[(fr < 100) | (fr > 4000)] = 0
    x = np.fft.irfft(spec, n_samples)
    env = 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * np.arange(n_samples) / fs)
    x = x / np.max(np.abs(x)) * env * 0.6
    return (x * 32767).astype(np.int16)


def make_bvh_clip(nframes, seed=0):
    """A synthetic 75-joint, 60-fps BVH animation dict (zyx Euler degrees) for exemplar / first-pose inputs."""
    rng = np.random.default_rng(seed)
    offsets = rng.normal(0, 6.0, (NJ, 3)).astype(np.float32)
    offsets[:, 1] = np.abs(offsets[:, 1])
    offsets[0] = [0.0, 90.0, 0.0]
    rot = (_smooth(rng, nframes, NJ * 3, 12.0).reshape(nframes, NJ, 3)).astype(np.float32)
    pos = np.repeat(offsets[None], nframes, axis=0)
    pos[:, 0] += (_smooth(rng, nframes, 3, 3.0) * np.array([1.0, 0.1, 1.0])).astype(np.float32)
    return dict(rotations=rot, positions=pos, offsets=offsets, parents=np.asarray(PARENTS, np.int32),
                names=list(BONE_NAMES), order="zyx", frametime=1.0 / 60.0)
