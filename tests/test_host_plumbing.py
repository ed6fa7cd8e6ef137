"""CPU tests of the host-side plumbing of generate_gesture() against fixtures recorded from the reference:
BVH parsing of a file written by the reference's bvh.save, BVH writing round trip; and of the NumPy oracle of the
animation kernels (exemplar feature extraction, decoder output -> BVH channels) against the same fixtures."""
import numpy as np
import pytest

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


def test_cli_option_surface_and_csv_batch_mode(tmp_path, monkeypatch):
    """`python -m zeggs.cli` keeps the reference scripts' options (main.py -o/-n; generate.py -o -p -se -s -a -n -fp -t -r -g -f -c)
    and the CSV batch columns; the entry points it calls are the drop-in train() / generate_gesture() (stubbed here: no GPU)."""
    import json
    from zeggs import cli
    import zeggs.generate as zg
    import zeggs.train as zt
    calls = []
    monkeypatch.setattr(zt, "train", lambda **k: calls.append(("train", k)))
    monkeypatch.setattr(zg, "generate_gesture", lambda **k: calls.append(("gen", k)))
    opts = {"train_opt": {"resume": False}, "net_opt": {"x": 1},
            "paths": {"base_path": str(tmp_path), "path_processed_data": "data/processed_v1", "output_dir": None, "models_dir": None}}
    of = tmp_path / "o.json"
    of.write_text(json.dumps(opts))
    assert cli.main(["train", "-o", str(of), "-n", "run1"]) == 0
    kind, k = calls[-1]
    assert kind == "train" and k["path_processed_data"].name == "processed_data.npz" and k["models_dir"].name == "saved_models"
    written = json.loads((k["logs_dir"].parent / "options.json").read_text())
    assert written["name"] == "run1" and written["paths"]["models_dir"] == str(k["models_dir"])
    gen_opts = k["logs_dir"].parent / "options.json"
    assert cli.main(["generate", "-o", str(gen_opts), "-s", "ex.bvh", "-a", "a.wav", "-f", "10", "200", "-t", "0.5", "-r", "7",
                     "-n", "out", "-fp", "fp.bvh", "-g"]) == 0
    kind, k = calls[-1]
    assert kind == "gen" and k["styles"] == [(cli.Path("ex.bvh"), [10, 200])] and k["temperature"] == 0.5 and k["seed"] == 7
    assert k["file_name"] == "out" and k["first_pose"] == cli.Path("fp.bvh") and k["results_path"].name == "results"
    csvf = tmp_path / "p.csv"
    csvf.write_text("base_path,audio,style,file_name,temperature,seed,use_gpu,frames,first_pose,generate\n"
                    f"{tmp_path},a1.wav,s1.bvh,o1,1.0,1234,True,5 50,s1.bvh,True\n"
                    f"{tmp_path},a2.wav,s2.bvh,o2,0.8,99,True,,s2.bvh,False\n"
                    f"{tmp_path},a3.wav,s3.bvh,o3,0.8,99,True,,,True\n")
    n0 = len(calls)
    assert cli.main(["generate", "-o", str(gen_opts), "-c", str(csvf)]) == 0
    new = [k for kind, k in calls[n0:]]
    assert [k["file_name"] for k in new] == ["o1", "o3"] and new[0]["styles"][0][1] == [5, 50] and new[1]["styles"][0][1] is None
    assert new[1]["first_pose"] is None and new[1]["temperature"] == 0.8 and new[1]["seed"] == 99


def test_generate_branch_fixture_integer_rules(golden_dir):
    """The integer rules of the generate_gesture() branches recorded in generate_branches.npz (round 4): "stitch" split frames
    (helpers.split_by_ratio: truncation, last end = length) bit-exact, and the shapes the reference returns for every branch."""
    g = np.load(golden_dir / "generate_branches.npz")
    ratio = [float(r) for r in g["blend_ratio"]]
    assert generate.split_by_ratio(135, ratio) == g["split_135"].tolist() == [[0, 40], [40, 135]]
    assert generate.split_by_ratio(10, [0.5, 0.25, 0.25]) == [[0, 5], [5, 7], [7, 10]]
    enc = g["stitch_encoding"]
    assert enc.shape == (1, 135, 64)
    s0 = int(g["split_135"][0][1])
    assert np.array_equal(enc[0, 0], enc[0, s0 - 1]) and np.array_equal(enc[0, s0], enc[0, -1]) and not np.array_equal(enc[0, 0], enc[0, s0])
    assert g["add_encoding"].shape == (1, 135, 64) and g["label_encoding"].shape == (1, 135, 19)
    assert g["label_encoding"][0, 0].sum() == 1.0 and g["label_encoding"][0, 0, synth.LABEL_NAMES.index(str(g["label"]))] == 1.0
    assert g["noaudio_stitch_encoding0"].shape == g["noaudio_stitch_encoding1"].shape == g["noaudio_add_encoding"].shape == (1, 64)
    # "add" without audio = the blend of the two per-style encodings "stitch" returns as a list (generate.py:299-308)
    blend = ratio[0] * g["noaudio_stitch_encoding0"] + ratio[1] * g["noaudio_stitch_encoding1"]
    np.testing.assert_allclose(g["noaudio_add_encoding"], blend, atol=1e-6)
    np.testing.assert_allclose(g["ndarray_encoding"][0, 7], g["embedding"], atol=0)
    for tag in ("stitch", "add", "label", "ndarray", "nofirst"):
        assert g[f"{tag}_rotations"].shape == (135, 75, 3) and g[f"{tag}_root_positions"].shape == (135, 3)


def test_format_table_text_equals_the_file_writer(tmp_path):
    """the host helper behind the streaming BVH writer (zeggs_format_table_text, single-threaded, row blocks) produces the bytes
    zeggs_write_table_text writes (= the reference's "%f " rows), for any split of the table into blocks"""
    import ctypes as C
    from zeggs import ops
    rng = np.random.default_rng(3)
    table = np.ascontiguousarray(rng.standard_normal((257, 228)) * np.array([1e3, 1.0, 1e-4] * 76), dtype=np.float64)
    table[5, 7] = -0.0
    table[6, 8] = 123456789.125
    path = tmp_path / "t.txt"
    assert ops.lib().zeggs_write_table_text(str(path).encode(), 0, table.ctypes.data_as(C.c_void_p), C.c_long(257), 228) == 0
    whole = path.read_bytes()
    assert whole.decode().splitlines()[0] == "".join("%f " % v for v in table[0])
    pieces = b"".join(anim.format_rows(table[a:b]) for a, b in ((0, 1), (1, 100), (100, 100), (100, 257)))
    assert pieces == whole
    head, seq = anim.bvh_header(np.zeros((75, 3)), synth.PARENTS, synth.BONE_NAMES, "zyx", 257, synth.DT)
    assert sorted(seq) == list(range(75)) and seq[0] == 0 and head.endswith("Frames: 257\nFrame Time: %f\n" % synth.DT)


def test_parse_table_text_is_exact_and_declines_malformed_tables(tmp_path):
    """the host helper behind bvh_load's MOTION block (zeggs_parse_table_text: plain decimals through one exact division,
    everything else through strtod, rows dealt to threads): every value == float(token); blank lines are skipped; a ragged or
    short table is DECLINED (-1) so that bvh_load falls back to numpy.loadtxt and its error"""
    import ctypes as C
    from zeggs import ops
    rng = np.random.default_rng(5)
    toks = ["0.1", "-0.000001", "123456.654321", "9007199254740991", "9007199254740993", "1e-7", "-3.5e3",
            "0.3333333333333333333", "179.999999", "-0", "42", "+7.25", "1e400", "0.000000", "-179.123456789012345678"]
    toks += ["%f" % v for v in rng.standard_normal(3000 - len(toks)) * 10.0 ** rng.integers(-3, 4, 3000 - len(toks))]
    rows, cols = 300, 10
    lines = [" ".join(toks[r * cols:(r + 1) * cols]) + " " for r in range(rows)]
    text = ("\n".join(lines[:100]) + "\n\n  \n" + "\n".join(lines[100:]) + "\n").encode()

    def parse(buf, r, c):
        out = np.full((r, c), np.nan)
        rc = ops.lib().zeggs_parse_table_text(buf + b"\0", C.c_size_t(len(buf)), out.ctypes.data_as(C.c_void_p), C.c_long(r), int(c))
        return rc, out

    rc, out = parse(text, rows, cols)
    assert rc == 0
    want = np.array([float(t) for t in toks]).reshape(rows, cols)
    assert np.array_equal(out, want) and np.array_equal(np.signbit(out), np.signbit(want))
    assert parse(text, rows + 1, cols)[0] == -1 and parse(text, rows, cols + 1)[0] == -1          # wrong shape
    assert parse(text.replace(b"42", b"4x2"), rows, cols)[0] == -1                                 # not a number
    # end to end: a BVH with blank lines inside the MOTION block loads as before
    clip = synth.make_bvh_clip(9, seed=2)
    anim.bvh_save(tmp_path / "a.bvh", clip)
    raw = (tmp_path / "a.bvh").read_text().split("\n")
    raw.insert(len(raw) - 4, "")
    (tmp_path / "b.bvh").write_text("\n".join(raw))
    a, b = anim.bvh_load(tmp_path / "a.bvh"), anim.bvh_load(tmp_path / "b.bvh")
    assert np.array_equal(a["rotations"], b["rotations"]) and np.array_equal(a["positions"], b["positions"])


@pytest.mark.parametrize("n,rate", [(6400, 16000), (160000, 16000), (123457, 16000), (28_800_000, 16000), (441000, 44100)])
def test_loudness_gating_blocks_equal_the_reference_loop(n, rate):
    """the vectorised gating-block table of the loudness pre-pass = pyloudnorm's per-block Python expressions, integer for integer"""
    from zeggs import audio
    lo, hi = audio.gating_blocks(n, rate)
    T_g, step = 0.4, 1.0 - 0.75
    nblocks = int(np.round(((n / rate - T_g) / (T_g * step))) + 1)
    j = range(0, nblocks) if nblocks < 2000 else list(range(0, 500)) + list(range(nblocks - 500, nblocks)) + list(range(500, nblocks, 97))
    assert len(lo) == nblocks == len(hi)
    for jj in j:
        assert lo[jj] == int(T_g * (jj * step) * rate)
        assert hi[jj] == min(int(T_g * (jj * step + 1) * rate), n)


def test_wav_reader_scaling_is_the_exact_division(tmp_path):
    """16-bit PCM -> float32 / 32768 in one in-place pass: bit-identical to the division (the scale is a power of two)"""
    import scipy.io.wavfile as wavfile
    rng = np.random.default_rng(5)
    x = rng.integers(-32768, 32768, 50000).astype(np.int16)
    x[:4] = [-32768, 32767, 0, -1]
    wavfile.write(tmp_path / "a.wav", 16000, x)
    fs, y = generate.read_wav_mono16k(tmp_path / "a.wav")
    assert fs == 16000 and y.dtype == np.float32
    assert np.array_equal(y, x.astype(np.float32) / 32768.0)
    wavfile.write(tmp_path / "b.wav", 8000, x)
    with pytest.raises(ValueError):
        generate.read_wav_mono16k(tmp_path / "b.wav")


def test_bvh_load_motion_section_is_found_by_its_own_line(tmp_path):
    """the MOTION section starts at a LINE that holds only the keyword -- a joint called LOCOMOTION_root (the bytes
    'MOTION' inside a name) must not cut the header short; a file without the section, or without its Frames / Frame Time lines,
    raises a ValueError that says so (not an AttributeError on a failed regular expression)."""
    import pytest
    names = ["LOCOMOTION_root", "Spine_MOTION", "Head"]
    parents = np.array([-1, 0, 1])
    offsets = np.array([[0.0, 90.0, 0.0], [0.0, 10.0, 0.0], [0.0, 20.0, 1.5]], np.float32)
    rng = np.random.default_rng(2)
    F = 5
    pos = np.repeat(offsets[None], F, 0).copy()
    pos[:, 0] += rng.standard_normal((F, 3)).astype(np.float32)
    rot = (30 * rng.standard_normal((F, 3, 3))).astype(np.float32)
    anim.bvh_save(tmp_path / "m.bvh", dict(rotations=rot, positions=pos, offsets=offsets, parents=parents, names=names,
                                           order="zyx", frametime=1 / 60))
    clip = anim.bvh_load(tmp_path / "m.bvh")
    assert list(clip["names"]) == names and clip["rotations"].shape == (F, 3, 3)
    np.testing.assert_allclose(clip["rotations"], rot, atol=2e-4)
    np.testing.assert_allclose(clip["positions"][:, 0], pos[:, 0], atol=2e-4)
    text = (tmp_path / "m.bvh").read_text()
    head = text[:text.index("\nMOTION")]
    (tmp_path / "nomotion.bvh").write_text(head + "\n")
    with pytest.raises(ValueError, match="MOTION"):
        anim.bvh_load(tmp_path / "nomotion.bvh")
    (tmp_path / "noframes.bvh").write_text(head + "\nMOTION\nFrame Time: 0.016\n0 0 0\n")
    with pytest.raises(ValueError, match="Frames"):
        anim.bvh_load(tmp_path / "noframes.bvh")
    (tmp_path / "notime.bvh").write_text(head + "\nMOTION\nFrames: 1\n0 0 0\n")
    with pytest.raises(ValueError, match="Frame Time"):
        anim.bvh_load(tmp_path / "notime.bvh")
