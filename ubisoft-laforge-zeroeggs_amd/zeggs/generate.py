"""generate_gesture(): the reference's inference entry point on the HIP engine.

Same signature, file layout and return value as ZEGGS/generate.py:22-411.  Host side: file I/O, style blending.
Device side (HIP kernels): exemplar feature extraction, BVH channel conversion, mel front-end, speech encoder, style encoder + VAE,
autoregressive decoder rollout (no-grad ring-buffer path).  `use_script` is accepted and ignored (TorchScript cannot
wrap the C-ABI calls; every shipped config sets it to false).
"""
import json
import pathlib
from pathlib import Path
from shutil import copyfile

import numpy as np
import torch
from scipy.io import wavfile

from . import anim, audio, compat

PROFILE = None      # bench.py sets this to a dict: generate_gesture() then adds the wall-clock ms of each stage to it (device
                    # stages bracketed by a synchronisation -- measurement only, the default path never synchronises for it)


class _stage:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if PROFILE is not None:
            import time
            torch.cuda.synchronize()
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if PROFILE is not None:
            import time
            torch.cuda.synchronize()
            PROFILE[self.name] = PROFILE.get(self.name, 0.0) + (time.perf_counter() - self.t0) * 1e3


def split_by_ratio(length, ratio):
    """integer frame splits of the "stitch" blend (reference helpers.py:26-37: truncation, last end = length)"""
    assert sum(ratio) == 1.0
    out, prev = [], 0
    for r in ratio:
        s, e = int(prev), int(prev + r * length)
        out.append([s, e])
        prev = e
    out[-1][-1] = length
    return out


def read_wav_mono16k(path):
    """16 kHz WAV -> float32 in [-1, 1) (the reference rescales int PCM the same way; other formats would need SoX)"""
    fs, x = wavfile.read(str(path))
    if fs != 16000:
        raise ValueError(f"{path}: expected a 16 kHz wav (got {fs} Hz); resample offline")
    if x.ndim > 1:
        x = x.mean(axis=1)
    if np.issubdtype(x.dtype, np.integer):
        scale = float(np.iinfo(x.dtype).max + 1)          # a power of two: multiplying by its reciprocal is the same division, exactly
        x = x.astype(np.float32)
        np.multiply(x, np.float32(1.0 / scale), out=x)    # (one pass over the 115 MB of a 30-minute clip instead of three)
    return fs, x if x.dtype == np.float32 else x.astype(np.float32)


def _example_features(p):
    """feature rows [F, 1134] of an exemplar clip (gaze slot zero), as in generate.py:229-248 (device tensors)"""
    root_pos, root_rot, root_vel, root_vrt, lpos, lrot, ltxy, lvel, lvrt = p[:9]
    n = len(root_vel)
    cols = [root_vel, root_vrt, lpos, ltxy, lvel, lvrt]
    return torch.cat([c.reshape(n, -1).to(torch.float32) for c in cols] +
                     [torch.zeros(n, 3, dtype=torch.float32, device=root_vel.device)], dim=1)


STREAM_MIN_FRAMES = 20000     # clips longer than this are decoded chunk by chunk with the BVH text written meanwhile
STREAM_CHUNK = 8192
STREAM_BLOCK = 512            # rows per formatting task (host threads)


def _decode_to_bvh_streaming(decoder, pose0, rpos0, rrot0, gaze_row, speech, style, stats, dt, path, parents, names,
                             chunk=None, block=None, threads=None):
    """Long clips (configs[4]: 108 000 frames): the B = 1 rollout runs as a sequence of persistent launches of `chunk` frames,
    each resumed from the state of the one before (zeggs_decoder_fwd_state_ex); behind every chunk its frames are converted to
    the rows of the BVH motion block ON THE DEVICE (zeggs_pose_to_bvh_table), downloaded on a copy stream into pinned memory
    and formatted by host threads (zeggs_format_table_text) WHILE the next chunks are being decoded; this thread writes the
    text blocks in order.  Same file as the one-launch path (decoder -> bvh_channels -> write_bvh_channels) up to the fp32
    re-association at the chunk boundaries (joint rotations a few hundredths of a degree apart after 700 free-running frames):
    tests/test_gpu_parity.py::test_generate_gesture_streaming_writer_equals_one_launch.  The decoder's frames are not kept
    (488 MB for 30 minutes).  Give-ups of the persistent kernel are collected in one status word that is looked at ONCE, after
    the last chunk: then everything is redone on the stage launches."""
    import ctypes as C
    import time as _t0
    from concurrent.futures import ThreadPoolExecutor
    from . import ops
    t_enter = _t0.perf_counter()
    chunk, block = chunk or STREAM_CHUNK, block or STREAM_BLOCK
    dev = speech.device
    T = speech.shape[1]
    J = len(parents)
    in_mean, in_std, out_mean, out_std = stats
    L = ops.lib()
    cols = 3 + 3 * J
    # frame 0 is the given first pose: its joint positions are the OFFSETs of the file (utils.write_bvh: offsets = positions[0])
    sl = lambda a, b: pose0[:, a:b]  # noqa: E731
    pos0, _ = anim.bvh_channels(rpos0, rrot0, sl(6, 6 + 3 * J).reshape(1, J, 3), sl(6 + 3 * J, 6 + 9 * J).reshape(1, J, 2, 3),
                                np.array([0, 0, 0]), np.array([1, 0, 0, 0]))
    head, seq = anim.bvh_header(pos0[0].cpu().numpy(), parents, names, "zyx", T, dt)
    seq_dev = torch.as_tensor(np.asarray(seq, np.int32), device=dev)
    d = anim.BvhDims(0, J, 1)
    d.start_pos[:] = [0.0, 0.0, 0.0]
    d.start_rot[:] = [1.0, 0.0, 0.0, 0.0]
    ref_pos, ref_rot = rpos0.to(torch.float32).contiguous(), rrot0.to(torch.float32).contiguous()
    copy_stream = torch.cuda.Stream(device=dev)
    status = ops.new_status(dev)
    main = torch.cuda.current_stream()

    # the last chunk is what nothing hides: its rows are converted, downloaded, formatted (36 us of host time per row: 18 ms for
    # a 512-row task) and written AFTER the decode has ended -- so it is short and its formatting tasks are small
    workers = threads or min(16, (__import__("os").cpu_count() or 4))
    tail_rows = max(1, min(chunk // 16, 512))
    RING = 4                                             # pinned staging buffers in flight (page-locking 15 MB costs ~5 ms: not per chunk)
    ring = _pinned_ring(RING, min(chunk, T) + 1, cols)

    marks = []

    def run(pool, fh):
        """decode chunk by chunk; chunk c's text is written while chunks c + 1 .. c + RING - 1 are queued on the device"""
        pending = []                                     # per chunk in flight: its formatting futures, in row order
        state = (pose0, rpos0, rrot0, None)
        k, c = 0, 0                                      # last frame produced so far, chunk index

        def drain(keep):
            while len(pending) > keep:
                for f in pending.pop(0):
                    fh.write(f.result())
        while True:
            n = min(chunk, T - 1 - k)                    # new frames of this chunk
            if n == T - 1 - k and n > tail_rows:         # the LAST chunk is short: what follows the end of the decode is only its
                n -= tail_rows                           # conversion, download and formatting (the pipeline's tail)
            sp, sty = speech[:, k:k + n + 1], style[:, k:k + n + 1]
            gz = gaze_row.expand(n + 1, 3)[None]
            if n > 0:
                pose, rpos, rrot, h = ops.decoder_chunk(decoder, state[0], state[1], state[2], gz, sp, sty, in_mean, in_std,
                                                        out_mean, out_std, dt, h_in=state[3], status=status)
                state = (pose[:, -1], rpos[:, -1], rrot[:, -1], h)
            else:                                        # a one-frame clip
                pose, rpos, rrot = pose0[:, None], rpos0[:, None], rrot0[:, None]
            lo = 0 if k == 0 else 1                      # (frame 0 of a later chunk was written with the chunk before)
            rows = n + 1 - lo
            table = torch.empty(rows, cols, dtype=torch.float64, device=dev)
            d.T = rows
            # (named, not temporaries: a tensor freed between two argument expressions hands its block to the next one)
            P = pose[0, lo:]
            a_rpos, a_rrot = rpos[0, lo:].contiguous(), rrot[0, lo:].contiguous()
            a_lpos, a_ltxy = P[:, 6:6 + 3 * J].contiguous(), P[:, 6 + 3 * J:6 + 9 * J].contiguous()
            rc = L.zeggs_pose_to_bvh_table(C.byref(d), C.c_void_p(a_rpos.data_ptr()), C.c_void_p(a_rrot.data_ptr()),
                                           C.c_void_p(a_lpos.data_ptr()), C.c_void_p(a_ltxy.data_ptr()),
                                           C.c_void_p(ref_pos.data_ptr()), C.c_void_p(ref_rot.data_ptr()),
                                           C.c_void_p(seq_dev.data_ptr()), C.c_void_p(table.data_ptr()),
                                           C.c_void_p(main.cuda_stream))
            del a_rpos, a_rrot, a_lpos, a_ltxy
            if rc != 0:
                raise RuntimeError("zeggs_pose_to_bvh_table: " + L.zeggs_last_error().decode())
            done = torch.cuda.Event()
            done.record(main)
            drain(RING - 1)                              # the staging buffer of chunk c - RING has been formatted and written
            host = ring[c % RING][:rows]
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done)
                host.copy_(table, non_blocking=True)
                table.record_stream(copy_stream)
                ev = torch.cuda.Event()
                ev.record(copy_stream)

            def fmt(host=host, ev=ev, r0=0, r1=0):
                ev.synchronize()
                return anim.format_rows(host[r0:r1].numpy())
            last = n == 0 or k + n >= T - 1
            blk = block if not last else max(16, -(-rows // (2 * workers)))
            pending.append([pool.submit(fmt, r0=r0, r1=min(r0 + blk, rows)) for r0 in range(0, rows, blk)])
            k += n
            c += 1
            if n == 0 or k >= T - 1:
                break
        marks.append(_t0.perf_counter())                 # everything is enqueued
        drain(0)
        marks.append(_t0.perf_counter())                 # ... decoded, converted, downloaded, formatted and written

    try:      # (the ring goes back to the pool whatever happens in between: ADVICE r5)
        import time as _t
        t_setup = _t.perf_counter()
        with ThreadPoolExecutor(max_workers=workers) as pool:
            with open(path, "wb") as fh:
                fh.write(head.encode())
                run(pool, fh)
                t_run = _t.perf_counter()
            t_close = _t.perf_counter()
            if PROFILE is not None:
                PROFILE["  streaming_writer_breakdown_ms"] = {
                    "setup": round((t_setup - t_enter) * 1e3, 2), "enqueue_all_chunks": round((marks[0] - t_setup) * 1e3, 2),
                    "last_chunks_decoded_formatted_written": round((marks[1] - marks[0]) * 1e3, 2), "file_close": round((t_close - t_run) * 1e3, 2)}
            if ops._persistent_live(0) and int(status[0].item()):     # (every chunk has been downloaded by now: no extra wait)
                ops._warn_gave_up(int(status[0].item()), "the whole rollout")
                # redo on the stage launches -- for THIS call only: the process-wide switch is restored afterwards (a
                # give-up here used to turn the persistent decode off for every later caller without a word)
                was = ops._OPTIONS.get("persistent", 1)
                ops.set_option("persistent", 0)
                try:
                    ops.fill_(status.view(torch.float32))
                    with open(path, "wb") as fh:
                        fh.write(head.encode())
                        run(pool, fh)
                finally:
                    ops.set_option("persistent", was)
    finally:
        _pinned_release(ring)


_PINNED = {}                      # (count, cols) -> list of free rings: a ring is CHECKED OUT for the duration of one call
_PINNED_LOCK = __import__("threading").Lock()


def _pinned_ring(count, rows, cols):
    """`count` page-locked float64 staging buffers [rows, cols].  Page-locking is the expensive part (4 x 15 MB for the default
    chunk), so rings are pooled for the life of the process -- but a ring belongs to ONE call at a time (two
    generate_gesture() calls on two threads shared the buffers and corrupted each other's rows): take one with this function,
    give it back with _pinned_release()."""
    key = (count, cols)
    with _PINNED_LOCK:
        free = _PINNED.setdefault(key, [])
        for i, ring in enumerate(free):
            if ring[0].shape[0] >= rows:
                return free.pop(i)
    return [torch.empty(rows, cols, dtype=torch.float64).pin_memory() for _ in range(count)]


def _pinned_release(ring):
    """Back to the pool -- which keeps ONE free ring per (count, cols), the largest: a smaller one can serve nobody the larger
    cannot, and page-locked memory that only accumulates with varying clip / chunk sizes is memory the system cannot page."""
    with _PINNED_LOCK:
        free = _PINNED.setdefault((len(ring), ring[0].shape[1]), [])
        free.append(ring)
        free.sort(key=lambda r: -r[0].shape[0])
        del free[1:]


def generate_gesture(audio_file, styles, network_path, data_path, results_path, style_encoding_type="example",
                     blend_type="add", blend_ratio=[0.5, 0.5], file_name=None, first_pose=None, temperature=1.0,
                     seed=1234, use_gpu=True, use_script=False):
    network_path, data_path = Path(network_path), Path(data_path)
    if results_path is not None:
        results_path = Path(results_path)
        results_path.mkdir(exist_ok=True)
    assert (audio_file is None) == (results_path is None)
    if not (use_gpu and torch.cuda.is_available()):
        raise RuntimeError("the ZeroEGGS MI355X engine has no CPU path (use_gpu=False / no GPU visible)")
    np.random.seed(seed)
    torch.manual_seed(seed)
    from . import ops
    ops.manual_seed(seed)       # the VAE noise comes from the library's counter-hash stream: the same seed gives the same clip,
                                # as `torch.manual_seed(seed)` does for the reference's randn_like (generate.py:86-87)
    device = torch.device("cuda")

    with open(data_path / "data_pipeline_conf.json") as f:
        pipe_conf = json.load(f)
    with open(data_path / "data_definition.json") as f:
        details = json.load(f)
    nlabels, label_names, bone_names = len(details["label_names"]), details["label_names"], details["bone_names"]
    parents, dt = np.asarray(details["parents"]), details["dt"]
    stat = np.load(data_path / "stats.npz")
    tt = lambda k: torch.as_tensor(np.asarray(stat[k]), dtype=torch.float32, device=device)  # noqa: E731
    audio_mean, audio_std = tt("audio_input_mean"), tt("audio_input_std")
    in_mean, in_std, out_mean, out_std = (tt("anim_input_mean"), tt("anim_input_std"), tt("anim_output_mean"),
                                          tt("anim_output_std"))
    if (network_path / "decoder.pt").exists():      # the reference's whole-module pickles
        speech_net = compat.load_module(network_path / "speech_encoder.pt", device).to(device).eval()
        decoder = compat.load_module(network_path / "decoder.pt", device).to(device).eval()
        style_net = None
        if style_encoding_type == "example":
            style_net = compat.load_module(network_path / "style_encoder.pt", device).to(device).eval()
    else:                                           # pickle-free twin written by zeggs.train (safetensors + arch.json)
        speech_net, decoder, style_net, _ = compat.load_state(network_path, device)
        speech_net, decoder = speech_net.eval(), decoder.eval()
        style_net = style_net.eval() if style_net is not None else None
        assert style_net is not None or style_encoding_type != "example", "checkpoint has no style encoder"

    # exemplar / first-pose BVH files are parsed on a host thread (the C parser releases the GIL) while this thread reads the WAV
    # and enqueues the audio front-end: every distinct file once
    from concurrent.futures import ThreadPoolExecutor
    bvh_paths = [str(Path(s_[0])) for s_ in styles
                 if style_encoding_type == "example" and isinstance(s_, (tuple, list)) and isinstance(s_[0], (pathlib.PurePath, str))]
    if audio_file is not None and isinstance(first_pose, (pathlib.PurePath, str)):
        bvh_paths.append(str(Path(first_pose)))
    loader = ThreadPoolExecutor(max_workers=1) if (audio_file is not None and bvh_paths) else None
    loading = {}
    if loader is not None:
        for pth in dict.fromkeys(bvh_paths):
            loading[pth] = loader.submit(anim.bvh_load, pth)
        loader.shutdown(wait=False)

    def load_bvh(pth):
        """parsed clip of a BVH path (a private copy: callers trim / modify it)"""
        fut = loading.get(str(Path(pth)))
        clip = fut.result() if fut is not None else anim.bvh_load(pth)
        return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in clip.items()}

    with torch.no_grad():
        if audio_file is not None:
            with _stage("wav_read_host"):
                _, wav = read_wav_mono16k(audio_file)
            n_frames = audio.n_anim_frames(len(wav))
            with _stage("loudness+mel_device"):
                feats = audio.preprocess_audio_device(wav, 60, n_frames, pipe_conf["audio_conf"],
                                                      pipe_conf["audio_feature_type"], device)
            with _stage("speech_encoder_device"):
                speech = speech_net(((feats[None] - audio_mean) / audio_std).contiguous())

        encodings, feat = [], None
        anim_name = "style"
        parsed = {}                     # exemplar files already parsed in this call: (path, trim) -> feature arrays
        for style in styles:
            if style_encoding_type == "example":
                if isinstance(style[0], (pathlib.PurePath, str)):
                    anim_name = Path(style[0]).stem
                    with _stage("exemplar_bvh_parse_host"):
                        clip = load_bvh(style[0])
                    if style[1] is not None:
                        clip["rotations"] = clip["rotations"][style[1][0]:style[1][1]]
                        clip["positions"] = clip["positions"][style[1][0]:style[1][1]]
                    assert int(np.ceil(1 / clip["frametime"])) == 60
                    with _stage("exemplar_features_device"):
                        feat = anim.preprocess_animation(clip, device)
                        parsed[(str(Path(style[0]).resolve()), None if style[1] is None else tuple(style[1]))] = feat
                        ex = (_example_features(feat) - in_mean) / in_std
                    with _stage("style_encoder_device"):
                        z, _, _ = style_net(ex[None].contiguous(), temperature)
                    encodings.append(z)
                elif isinstance(style[0], np.ndarray):
                    anim_name = style[1]
                    encodings.append(torch.as_tensor(style[0], dtype=torch.float32, device=device)[None])
            elif style_encoding_type == "label":
                onehot = torch.zeros((1, nlabels), device=device)
                onehot[0, label_names.index(style)] = 1.0
                encodings.append(onehot)
                anim_name = style
                assert first_pose is not None
            else:
                raise ValueError("Unknown style encoding type")

        if blend_type == "stitch" and len(encodings) > 1:
            if audio_file is None:
                final = encodings
            else:
                assert len(styles) == len(blend_ratio)
                se = split_by_ratio(n_frames, blend_ratio)
                final = torch.cat([z.unsqueeze(1).repeat(1, e - s, 1) for z, (s, e) in zip(encodings, se)], dim=1)
        elif blend_type == "add" and len(encodings) > 1:
            assert len(encodings) == len(blend_ratio)
            final = torch.matmul(torch.stack(encodings, dim=1).transpose(2, 1),
                                 torch.tensor(blend_ratio, device=device, dtype=torch.float32))
        else:
            final = encodings[0]

        if audio_file is not None:
            if first_pose is not None:
                key = (str(Path(first_pose).resolve()), None) if isinstance(first_pose, (pathlib.PurePath, str)) else None
                if key in parsed:       # the first pose is an (untrimmed) style exemplar of this call: parsed already
                    feat = parsed[key]
                else:
                    with _stage("first_pose_bvh_parse_host"):
                        clip = load_bvh(first_pose) if key is not None else dict(first_pose)
                    with _stage("first_pose_features_device"):
                        feat = anim.preprocess_animation(clip, device)
            g = lambda a: a[0:1].to(torch.float32).contiguous()  # noqa: E731
            root_pos, root_rot, root_vel, root_vrt, lpos, lrot, ltxy, lvel, lvrt = feat[:9]
            gaze_pos = feat[14]
            if final.dim() == 2:
                final = final.unsqueeze(1).repeat(1, speech.shape[1], 1)
            if file_name is None:
                file_name = f"audio_{Path(audio_file).stem}_label_{anim_name}"
            T = speech.shape[1]
            film = hasattr(decoder.recurrent_decoder, "gammas_predictor")
            if T > STREAM_MIN_FRAMES and not film:
                # long clip: chunked persistent decode with the BVH text formatted and written underneath it
                pose0 = torch.cat([g(x).reshape(1, -1) for x in (root_vel, root_vrt, lpos, ltxy, lvel, lvrt)], dim=1)
                # the WAV copy runs beside the decode; its errors surface where the one-launch path's do (a bare Thread
                # sent a PermissionError to threading.excepthook instead of the `except` below)
                from concurrent.futures import ThreadPoolExecutor as _TPE
                with _TPE(max_workers=1) as copy_pool:
                    try:
                        copier = copy_pool.submit(copyfile, audio_file, str(results_path / (file_name + ".wav")))
                        with _stage("decode+pose_to_bvh_device_with_bvh_text_write_host_underneath"):
                            _decode_to_bvh_streaming(decoder, pose0, g(root_pos), g(root_rot), g(gaze_pos), speech,
                                                     final.contiguous(), (in_mean, in_std, out_mean, out_std), dt,
                                                     str(results_path / (file_name + ".bvh")), parents, bone_names)
                        copier.result()
                    except (PermissionError, OSError) as e:
                        print(e)
                return final
            gaze = g(gaze_pos).repeat(T, 1)[None]
            with _stage("decode_device"):
                out = decoder(g(root_pos), g(root_rot), g(root_vel), g(root_vrt), g(lpos), g(ltxy), g(lvel), g(lvrt),
                              gaze.contiguous(), speech, final.contiguous(), None, in_mean, in_std, out_mean, out_std, dt)
            V_root_pos, V_root_rot, _, _, V_lpos, V_ltxy = out[0], out[1], out[2], out[3], out[4], out[5]
            try:
                with _stage("pose_to_bvh_device"):
                    channels = anim.bvh_channels(V_root_pos[0], V_root_rot[0], V_lpos[0], V_ltxy[0], np.array([0, 0, 0]),
                                                 np.array([1, 0, 0, 0]))
                with _stage("bvh_text_write_host"):
                    anim.write_bvh_channels(str(results_path / (file_name + ".bvh")), *channels, parents=parents,
                                            names=bone_names, order="zyx", dt=dt)
                with _stage("wav_copy_host"):
                    copyfile(audio_file, str(results_path / (file_name + ".wav")))
            except (PermissionError, OSError) as e:
                print(e)
    return final
