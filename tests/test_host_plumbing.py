"""CPU tests of the host-side plumbing of generate_gesture() against fixtures recorded from the reference:
BVH parsing of a file written by the reference's bvh.save, BVH writing round trip; and of the NumPy oracle of the
animation kernels (exemplar feature extraction, decoder output -> BVH channels) against the same fixtures."""
import numpy as np

from oracle import anim as oanim
from zeggs import anim, generate, synth


def _exemplar(golden_dir, tmp_path):
    g = np.load(golden_dir / "generate.npz")
    p = tmp_path / "ex.bvh"
    p.write_bytes(g["exemplar_bvh"].tobytes())
    return g, p


def test_bvh_load_matches_reference(golden_dir, tmp_path):
    g, p = _exemplar(golden_dir, tmp_path)
    clip = anim.bvh_load(p)
    np.testing.assert_array_equal(clip["parents"], g["ex_parents"])          # integer skeleton table: bit-exact
    np.testing.assert_allclose(clip["rotations"], g["ex_rotations"], atol=0)
    np.testing.assert_allclose(clip["positions"], g["ex_positions"], atol=0)
    np.testing.assert_allclose(clip["offsets"], g["ex_offsets"], atol=0)
    assert clip["order"] == "zyx" and abs(clip["frametime"] - 1 / 60) < 1e-6 and clip["names"] == synth.BONE_NAMES


def test_preprocess_animation_matches_reference(golden_dir, tmp_path):
    g, p = _exemplar(golden_dir, tmp_path)
    feats = oanim.preprocess_animation(anim.bvh_load(p))
    names = ("root_pos", "root_rot", "root_vel", "root_vrt", "lpos", "lrot", "ltxy", "lvel", "lvrt", "cpos", "crot",
             "ctxy", "cvel", "cvrt", "gaze_pos", "gaze_dir")
    for n, f in zip(names, feats):
        np.testing.assert_allclose(np.asarray(f), g["feat_" + n], atol=2e-4, rtol=1e-4, err_msg=n)   # reference: float32 path


def test_bvh_save_load_round_trip(tmp_path):
    clip = synth.make_bvh_clip(7, seed=1)
    anim.bvh_save(tmp_path / "a.bvh", clip)
    back = anim.bvh_load(tmp_path / "a.bvh")
    np.testing.assert_allclose(back["rotations"], clip["rotations"], atol=1e-5)
    np.testing.assert_allclose(back["positions"][:, 0], clip["positions"][:, 0], atol=1e-5)
    np.testing.assert_array_equal(back["parents"], clip["parents"])


def test_quaternion_matrix_round_trip_and_split():
    rng = np.random.default_rng(0)
    q = oanim.q_normalize(rng.standard_normal((50, 4)))
    ex, ey = np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
    xy = np.stack([oanim.q_mul_vec(q, ex), oanim.q_mul_vec(q, ey)], axis=-2)
    q2 = oanim.q_from_xform(oanim.xform_from_xy(xy))
    assert np.abs(np.abs(np.sum(q * q2, axis=-1)) - 1.0).max() < 1e-9
    assert generate.split_by_ratio(601, [0.3, 0.7]) == [[0, 180], [180, 601]]


def test_bvh_channels_oracle_matches_reference(golden_dir):
    """decoder outputs captured inside the reference's generate_gesture -> the channels it wrote to out.bvh
    (text with 6 decimals: compared as rotations, 1e-5)"""
    g = np.load(golden_dir / "generate.npz")
    pos, eul = oanim.bvh_channels(g["dec_root_pos"], g["dec_root_rot"], g["dec_lpos"], g["dec_ltxy"],
                                  start_position=np.array([0, 0, 0]), start_rotation=np.array([1, 0, 0, 0]))
    np.testing.assert_allclose(pos[:, 0], g["out_positions"][:, 0], atol=2e-5)
    qa = oanim.q_from_euler(np.radians(eul))
    qb = oanim.q_from_euler(np.radians(g["out_rotations"].astype(np.float64)))
    assert np.abs(np.abs(np.sum(qa * qb, axis=-1)) - 1.0).max() < 1e-9
    # the quaternions the reference derived from the two-axis encodings (generate.py:389), float32
    q = oanim.q_from_xform(oanim.xform_from_xy(g["dec_ltxy"].astype(np.float64)))
    assert np.abs(np.abs(np.sum(q * g["dec_lrot"], axis=-1)) - 1.0).max() < 1e-5


def test_state_dict_checkpoint_round_trip(tmp_path):
    """pickle-free checkpoints (safetensors + arch.json) rebuild every variant with identical weights"""
    import torch
    from zeggs import compat, modules
    torch.manual_seed(3)
    for rnn_cond, typ in (("normal", "attn"), ("film", "gru")):
        se = modules.SpeechEncoder(81, 64, 64)
        de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, 64, 64, 128, 2, rnn_cond=rnn_cond)
        st = modules.StyleEncoder(synth.POSE_IN, 96, 64, type=typ, use_vae=True)
        compat.save_state(tmp_path / rnn_cond, se, de, st, meta={"iteration": 7})
        se2, de2, st2, meta = compat.load_state(tmp_path / rnn_cond)
        assert meta["iteration"] == 7
        for a, b in ((se, se2), (de, de2), (st, st2)):
            sa, sb = a.state_dict(), b.state_dict()
            assert sa.keys() == sb.keys()
            assert all(torch.equal(sa[k], sb[k]) for k in sa)
    se2, de2, st2, _ = (lambda d: (compat.save_state(d, se, de, None), compat.load_state(d))[1])(tmp_path / "label")
    assert st2 is None
