#!/usr/bin/env python
"""Headline benchmark: gesture frames/sec of the ZeroEGGS training step on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or as the plain command above -- bench.py then
re-executes itself under torch.distributed.run on 127.0.0.1 with a free port (one rank per GPU over RCCL).

A "step" is ONE full training iteration of BASELINE.json configs[1]: configs_v1 networks (random-init, seed 1234),
batch 32 per GPU of 256-frame windows cut from synthetic 60-fps 2-minute clips that are resident in HBM, style
examples of length 384: gather -> speech encoder -> style VAE -> 255-step decoder rollout -> FK/L1 loss -> BPTT ->
(RCCL all-reduce) -> fused RAdam.  Nothing is skipped or cached inside the timed region; training-mode dropout is on.
Prints ONE JSON line on rank 0 (contract: task statement / DESIGN.md "Measurement").  At N = 1 the line also carries
the other BASELINE.json configs as extra keys: `decode` (B=1, 1800 frames), `decode_30min` (configs[4]),
`generate_30min` (configs[4] through generate_gesture(), per-stage ms), `v2_label_b64` (configs[3]) and the CPU baselines
(the unmodified reference timed on this box: oracle/ref_timing.py).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # three engine streams + RCCL's: see zeggs/__init__.py (before HIP initialises)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "ubisoft-laforge-zeroeggs_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

from zeggs import engine, modules, ops, synth  # noqa: E402

BATCH, WINDOW, EXAMPLE_LEN, CLIP_FRAMES = 32, 256, 384, 7200
H, SP, ST = 1024, 64, 64
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3
FP64_VECTOR_PEAK_TFLOPS = 78.6          # MI355X datasheet (fp64 matrix = fp64 vector rate)


def step_weight_bytes(style=ST):
    """algorithmic bytes of ONE decoder step: every per-step weight + bias read once"""
    xd = synth.POSE_IN + SP + style
    return 4 * (H * xd + 3 * H * (H + xd) + 3 * H * H + 3 * H * H + 3 * H * H + synth.POSE_OUT * H
                + H + 4 * 3 * H + synth.POSE_OUT)


def step_bytes(batch, style=ST):
    return step_weight_bytes(style) + batch * 4 * (synth.POSE_IN + SP + style + 2 * H + synth.POSE_OUT + 2 * H)


def step_flops(batch, style=ST):
    xd = synth.POSE_IN + SP + style
    return 2 * batch * (H * xd + 3 * H * (H + xd) + 3 * H * H + 3 * H * H + 3 * H * H + synth.POSE_OUT * H)


def build_dataset(n_train=64, n_unique=4, seed=0, shared=None):
    """64 two-minute 60-fps clips (n_unique distinct, tiled: the content does not affect the timing).
    shared = (path, is_builder, barrier): multi-rank runs synthesise the distinct clips ONCE -- the builder rank writes them
    to `path` (/dev/shm), the others map them after the barrier -- so that start-up does not put N copies of the generator
    on the host's cores."""
    stats = synth.make_stats()
    if shared is not None:
        path, is_builder, barrier = shared
        if is_builder:
            base = [synth.make_clip(CLIP_FRAMES, seed=seed * 1000 + i, stats=stats) for i in range(n_unique)]
            np.savez(path, **{f"{i}.{k}": v for i, c in enumerate(base) for k, v in c.items()})
        barrier()
        if not is_builder:
            z = np.load(path, mmap_mode="r")
            base = [{k.split(".", 1)[1]: z[k] for k in z.files if k.startswith(f"{i}.")} for i in range(n_unique)]
    else:
        base = [synth.make_clip(CLIP_FRAMES, seed=seed * 1000 + i, stats=stats) for i in range(n_unique)]
    clips = [base[i % n_unique] for i in range(n_train)]
    data = {k: np.concatenate([c[k] for c in clips], axis=0) for k in clips[0]}
    bounds = np.arange(n_train + 1) * CLIP_FRAMES
    data["ranges_train"] = np.stack([bounds[:-1], bounds[1:]], axis=1).astype(np.int32)
    data["ranges_train_labels"] = (np.arange(n_train) % 19).astype(np.int32)
    data.update(stats)
    return data


def build_nets(device, style=ST, with_style_encoder=True):
    torch.manual_seed(1234)
    se = modules.SpeechEncoder(synth.N_AUDIO, 64, SP)
    de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, SP, style, H, 2)
    st = modules.StyleEncoder(synth.POSE_IN, 512, ST, type="attn", use_vae=True) if with_style_encoder else None
    return se.to(device).train(), de.to(device).train(), (st.to(device).train() if st is not None else None)


def profile_stamp(path):
    """Short git-style identity of a committed profile file a bench figure is read from (sha1 of its bytes, first 12 hex digits) and
    its modification date: a stale figure is visible in the line itself."""
    import hashlib
    p = ROOT / path
    if not p.exists():
        return None
    return {"file": path, "sha1_12": hashlib.sha1(p.read_bytes()).hexdigest()[:12],
            "mtime": time.strftime("%Y-%m-%d", time.gmtime(p.stat().st_mtime))}


def kernel_shares(path, names):
    """Share of the kernel time of each kernel whose name contains one of `names`, read from a kernel-statistics csv under profiles/
    (rocprofv3 --stats: columns Name, ..., Percentage; the tools/rocpd_stats.py summaries: kernel, calls, total_us, avg_us, pct);
    None if the file or a kernel is missing."""
    import csv
    p = ROOT / path
    if not p.exists():
        return None
    out = {}
    with open(p, newline="") as f:
        for row in csv.DictReader(f):
            nm, pct = row.get("Name") or row.get("kernel") or "", row.get("Percentage") or row.get("pct")
            for n in names:
                if n in nm and pct is not None:
                    out[n] = round(out.get(n, 0.0) + float(pct), 1)
    return out if all(n in out for n in names) else None


KERNEL_STATS_CSV = ("profiles/r06_train_only_kernel_stats_12steps.csv", "profiles/r05_train_only_kernel_stats_12steps.csv")


def dominant_kernel_note():
    for path in KERNEL_STATS_CSV:
        sh = kernel_shares(path, ("train_bwd_persistent_k", "train_fwd_persistent_k"))
        if sh:
            return (f"train_bwd_persistent_k ({sh['train_bwd_persistent_k']} % of kernel time, read from {path}): its figures are "
                    f"roofline.backward; train_fwd_persistent_k ({sh['train_fwd_persistent_k']} %) is the top-level entry")
    return "train_bwd_persistent_k (no kernel-statistics csv found under profiles/): its figures are roofline.backward"


def sweep_ms(which):
    import ctypes as C
    ms = C.c_float(0)
    ops._check(ops.lib().zeggs_timing_ms(int(which), C.byref(ms)), "timing_ms")
    return float(ms.value)


# ----------------------------------------------------------------------------- CPU baselines
def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return os.cpu_count() or 1


def cpu_port_train(data, threads, iters=2):
    """The CPU oracle (torch-CPU restatement of the reference, oracle/) on the SAME workload shape (B=32, T=256,
    example 384): `iters` full iterations fwd+loss+bwd (RAdam is negligible)."""
    from oracle import loss as oloss
    from oracle import nets as onets
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    se = modules.SpeechEncoder(synth.N_AUDIO, 64, SP)
    de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, SP, ST, H, 2)
    st = modules.StyleEncoder(synth.POSE_IN, 512, ST, type="attn", use_vae=True)
    ws = [{k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()} for m in (se, de, st)]
    B, T = BATCH, WINDOW
    f = lambda k: torch.as_tensor(np.stack([data[k][i * 100:i * 100 + T] for i in range(B)]))  # noqa: E731
    W = {k: f(k) for k in ("X_audio_features", "Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos",
                           "Y_ltxy", "Y_lvel", "Y_lvrt", "Y_gaze_pos")}
    t = lambda k: torch.as_tensor(np.asarray(data[k]), dtype=torch.float32)  # noqa: E731
    ex = torch.randn(B, EXAMPLE_LEN, synth.POSE_IN)
    eps = torch.randn(B, ST)
    times = []
    for it in range(iters + 1):
        t0 = time.perf_counter()
        speech = onets.speech_encoder(ws[0], (W["X_audio_features"] - t("audio_input_mean")) / t("audio_input_std"))
        z, mu, lv = onets.style_encoder(ws[2], ex, eps)
        O = onets.decoder_rollout(ws[1], W["Y_root_pos"][:, 0], W["Y_root_rot"][:, 0], W["Y_root_vel"][:, 0],
                                  W["Y_root_vrt"][:, 0], W["Y_lpos"][:, 0], W["Y_ltxy"][:, 0], W["Y_lvel"][:, 0],
                                  W["Y_lvrt"][:, 0], W["Y_gaze_pos"], speech, z.unsqueeze(1).repeat(1, T, 1),
                                  t("anim_input_mean"), t("anim_input_std"), t("anim_output_mean"),
                                  t("anim_output_std"), synth.DT)
        loss, _ = oloss.training_loss(O, [W[k] for k in ("Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt",
                                                         "Y_lpos", "Y_ltxy", "Y_lvel", "Y_lvrt")], W["Y_gaze_pos"],
                                      synth.PARENTS, synth.DT, mu, lv, iteration=it)
        loss.backward()
        for w in ws:
            for v in w.values():
                v.grad = None
        times.append(time.perf_counter() - t0)
    dt_ = float(np.mean(times[1:]))          # first iteration warms allocators / thread pools
    return B * T / dt_, dt_


def cpu_port_decode(frames=400):
    from oracle import nets as onets
    torch.set_num_threads(1)                 # generate.py:88 runs the decode on one thread
    torch.manual_seed(1234)
    de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, SP, ST, H, 2)
    sd = {k: v.detach() for k, v in de.state_dict().items()}
    stats = synth.make_stats()
    c = synth.make_clip(frames, seed=1, stats=stats)
    t = lambda k: torch.as_tensor(np.asarray(stats[k]), dtype=torch.float32)  # noqa: E731
    W = {k: torch.as_tensor(v[None]) for k, v in c.items()}
    speech, style = torch.randn(1, frames, 64) * 0.5, torch.randn(1, frames, 64) * 0.5
    with torch.no_grad():
        t0 = time.perf_counter()
        onets.decoder_rollout(sd, W["Y_root_pos"][:, 0], W["Y_root_rot"][:, 0], W["Y_root_vel"][:, 0], W["Y_root_vrt"][:, 0],
                              W["Y_lpos"][:, 0], W["Y_ltxy"][:, 0], W["Y_lvel"][:, 0], W["Y_lvrt"][:, 0], W["Y_gaze_pos"],
                              speech, style, t("anim_input_mean"), t("anim_input_std"), t("anim_output_mean"),
                              t("anim_output_std"), synth.DT)
        dt_ = time.perf_counter() - t0
    return (frames - 1) / dt_


def cpu_port_mel(seconds=10):
    from oracle import mel as omel
    wav = synth.synth_wav(16000 * seconds, seed=0).astype(np.float32) / 32768.0
    t0 = time.perf_counter()
    omel.preprocess_audio(wav, 60 * seconds)
    return 60 * seconds / (time.perf_counter() - t0)


def cpu_baselines(data):
    """`cpu_baseline` (train), plus decode / mel legs.  kind = "reference": the unmodified reference timed here
    through oracle/ref_timing.py -- from /root/reference in the build container, from the oracle/_ref snapshot
    (oracle/build_ref.py, travels with the built .so) on the GPU box; only if neither exists kind = "port": the oracle
    restatement timed on this host's cores, next to the reference's figures recorded on the build box
    (profiles/r02_cpu_reference.json) and the port/reference ratio measured there."""
    from oracle import ref_shims
    rec_file = ROOT / "profiles" / "r02_cpu_reference.json"
    recorded = json.load(open(rec_file)) if rec_file.exists() else None
    if ref_shims.available():
        # the reference ITSELF on this box's host cores (on the GPU box: the byte-for-byte snapshot oracle/build_ref.py left in
        # oracle/_ref/): train() with every core (the headline baseline) and with thread_count = 1 (the shipped config value,
        # configs_v1.json:37); bounded samples (5 + 1 steady iterations; iteration 0 = checkpoint + sample rendering is skipped)
        from oracle import ref_timing
        ncpu = os.cpu_count() or 1
        # "all cores" = the physical cores, capped at 32 threads: torch's intra-op pool gets SLOWER beyond that on this
        # workload (hundreds of sub-millisecond ops per decoder step, each a fork-join over the pool)
        nthr = max(1, min(physical_cores(), 32))
        r = ref_timing.measure(iters=5, frames=600, train_threads=(nthr,), legs=("train", "decode", "mel"))      # ~30 s of CPU work
        tr = next(iter(r["train"].values()))
        train = {"value": tr["frames_per_s"], "unit": "frames/s", "cores": tr["threads"], "kind": "reference",
                 "sample": f"{tr['iterations_timed']} steady iterations of the unmodified reference train() "
                           f"(ZEGGS/train.py:29), B={BATCH} x {WINDOW}, {np.mean(tr['s_per_iteration']):.2f} s/iteration, "
                           f"thread_count={tr['threads']} (of {ncpu} logical / {physical_cores()} physical cores of this box)",
                 "source": r["reference"], "cpu": r["cpu"]}
        if ncpu > 1 and not os.environ.get("ZEGGS_BENCH_SKIP_1THREAD"):
            r1 = ref_timing.measure(iters=1, train_threads=(1,), legs=("train",))
            t1 = next(iter(r1["train"].values()))
            train["thread_count_1"] = {"value": t1["frames_per_s"], "cores": 1, "iterations_timed": t1["iterations_timed"],
                                       "s_per_iteration": round(float(np.mean(t1["s_per_iteration"])), 2),
                                       "note": "the shipped configs_v1.json value (thread_count: 1)"}
        dec = {"value": r["decode"]["threads_1"]["frames_per_s"], "unit": "frames/s", "cores": 1, "kind": "reference",
               "sample": "reference Decoder.forward, B=1, no_grad, 600 frames"}
        mel = {"value": r["mel"]["anim_frames_per_s"], "unit": "frames/s", "cores": 1, "kind": "reference",
               "sample": "reference preprocess_audio on 10 s of 16 kHz audio"}
        return train, dec, mel
    threads = min(physical_cores(), 16)
    fps, dt_ = cpu_port_train(data, threads)
    train = {"value": round(fps, 1), "unit": "frames/s", "cores": threads, "kind": "port",
             "sample": f"2 full training iterations (B={BATCH}, T={WINDOW}, example {EXAMPLE_LEN}) of the torch-CPU "
                       f"oracle, {dt_:.2f} s/iteration"}
    dec = {"value": round(cpu_port_decode(), 1), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "oracle decoder rollout, B=1, no_grad, 400 frames, 1 thread"}
    mel = {"value": round(cpu_port_mel(), 1), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "oracle preprocess_audio on 10 s of 16 kHz audio"}
    if recorded:
        rt = recorded["train"]
        train["reference_recorded"] = {"where": "build container, " + recorded["cpu"]["model"],
                                       **{k: {"threads": v["threads"], "frames_per_s": v["frames_per_s"]} for k, v in rt.items()},
                                       "port_on_same_box": recorded.get("port_on_build_box", {}).get("train")}
        dec["reference_recorded"] = {k: v["frames_per_s"] for k, v in recorded["decode"].items()}
        dec["reference_recorded"]["port_on_same_box"] = recorded.get("port_on_build_box", {}).get("decode")
        mel["reference_recorded"] = {"anim_frames_per_s": recorded["mel"]["anim_frames_per_s"],
                                     "port_on_same_box": recorded.get("port_on_build_box", {}).get("mel")}
    return train, dec, mel


# ----------------------------------------------------------------------------- extra configs (N = 1)
def decode_args(de, dev, T):
    st = {k: torch.as_tensor(v, dtype=torch.float32, device=dev) for k, v in synth.make_stats().items() if k.startswith("anim")}
    g = torch.Generator(device="cpu").manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)  # noqa: E731
    return (de.eval(), r(1, synth.POSE_OUT), torch.zeros(1, 3, device=dev), torch.tensor([[1.0, 0, 0, 0]], device=dev),
            r(1, T, 3) * 10, r(1, T, SP) * 0.3, r(1, T, ST) * 0.3, st["anim_input_mean"], st["anim_input_std"],
            st["anim_output_mean"], st["anim_output_std"], synth.DT)


def wgrad_gemm_alone(dev, reps=8):
    """The decoder's two largest weight-gradient products ALONE on the chip, through the LDS-tiled stream-K kernel (round 4) and the
    barrier-free direct kernel (round 5; csrc/gemm.hip: gemm_tn_direct_kernel): TFLOP/s of fp32 MFMA from HIP events, the zero fill of
    the output included.  (In the training iteration these products share the chip with two other queues: DESIGN section 3.2.)"""
    shapes = {"dW_hh 3072x1024 K=8160": (3072, 1024, 8160), "dW_ih0 3072x2286 K=8160": (3072, 2286, 8160)}
    before = {k: ops._OPTIONS.get(k) for k in ("gemm_direct", "gemm_direct_wgs", "gemm_direct_shield", "gemm_direct_depth")}
    out = {"bound": "mfma", "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "kernels": {}}
    try:
        # direct_shield: the variant that owns its SIMDs' register files, 8 operand pairs in flight -- what the engine's
        # three-queue schedule runs (alone on the chip it has nothing to be shielded from: the same rate as `direct`)
        for tag, mode, shield, depth in (("lds_tiled_streamk", 0, 0, 4), ("direct", 1, 0, 4), ("direct_shield", 1, 1, 8)):
            ops.set_option("gemm_direct", mode)
            ops.set_option("gemm_direct_wgs", 0)
            ops.set_option("gemm_direct_shield", shield)
            ops.set_option("gemm_direct_depth", depth)
            res = {}
            for name, (M, N, K) in shapes.items():
                A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
                Cm = torch.zeros(M, N, device=dev)
                f = lambda: ops.gemm(A, B, Cm, M, N, K, (1, M), (N, 1), (N, 1))  # noqa: E731
                f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    f()
                e1.record()
                e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                res[name] = {"us": round(us, 1), "achieved": round(2.0 * M * N * K / us / 1e6, 1),
                             "frac": round(2.0 * M * N * K / us / 1e6 / MFMA_F32_PEAK_TFLOPS, 3)}
            out["kernels"][tag] = res
    finally:
        for k, v in before.items():
            if v is not None:
                ops.set_option(k, v)
    return out


def decode_rate(de, dev, T=1801):
    """second half of BASELINE.json's metric: autoregressive decode frames/s (B=1, no_grad ring path, speech/style
    already encoded; 30 s of 60-fps frames), with its own roofline entry (75.7 MB of weights per frame)."""
    args = decode_args(de, dev, T)
    with torch.no_grad():
        ops.decoder_core(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.decoder_core(*args)
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        sweep = sweep_ms(0) * 1e-3 / (T - 1)
        # the same rollout as a chain of stage launches (3 per frame), for comparison
        ops.set_option("persistent", 0)
        ops.decoder_core(*args)
        torch.cuda.synchronize()
        chain_us = sweep_ms(0) * 1e3 / (T - 1)
        ops.set_option("persistent", 1)
    de.train()
    ach = step_bytes(1) / sweep / 1e9
    return {"value": round((T - 1) / dt_, 1), "unit": "frames/s", "us_per_frame": round(dt_ * 1e6 / (T - 1), 2),
            "config": f"B=1 autoregressive rollout of {T - 1} frames, no_grad: weight-stationary persistent kernel "
                      f"(one launch, weights in registers, granule exchange between CUs)",
            "stage_launch_chain_us_per_frame": round(chain_us, 2),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "us_per_step": round(sweep * 1e6, 2),
                         "algorithmic_bytes_per_step": step_bytes(1)}}


def decode_30min(se, de, dev, minutes=30.0, reps=3):
    """BASELINE.json configs[4] shape: 30 min of 16 kHz audio -> device mel front-end -> speech encoder -> B=1
    autoregressive decode of 108 000 frames (random-init nets, synthetic audio); decode repeated `reps` times."""
    from zeggs import audio
    n = int(minutes * 60 * 16000)
    wav = (0.1 * np.random.default_rng(0).standard_normal(n)).astype(np.float32)
    stats = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32, device=dev) for k, v in synth.make_stats().items()}
    T = audio.n_anim_frames(n)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0

    se.eval(), de.eval()
    with torch.no_grad():
        audio.mel_features(wav[:16000], 60)                                    # warm-up
        wav_dev, t_up = timed(lambda: torch.as_tensor(wav).to(dev))           # 115 MB over PCIe (not part of mel_ms)
        feats, t_mel = timed(lambda: audio.mel_features(wav_dev, T))
        x = feats[None].contiguous()
        ops.normalize_rows_(x, stats["audio_input_mean"], float(stats["audio_input_std"]))
        se(x[:, :512].contiguous())                                           # warm-up
        sp, t_se = timed(lambda: se(x))
        args = decode_args(de, dev, T)
        args = args[:5] + (sp,) + args[6:]
        ts = []
        for _ in range(reps):
            out, t_dec = timed(lambda: ops.decoder_core(*args))
            ts.append(t_dec)
        finite = bool(torch.isfinite(out[0]).all() and torch.isfinite(feats).all())
    se.train(), de.train()
    t_dec = float(np.mean(ts))                        # (mean of the repetitions; decode_s_all lists them)
    n_stft = audio.stft_frame_count(n)
    mel_bytes = n * 4 + T * 81 * 4                   # 200 new samples in per STFT frame, (80 + 1) x 60/80 floats out
    return {"frames": T, "mel_ms": round(t_mel * 1e3, 2), "speech_encoder_ms": round(t_se * 1e3, 2),
            "wav_upload_ms": round(t_up * 1e3, 2),
            "mel_roofline": {"bound": "hbm", "kernel": "mel_stft_fft_k (round 4): the 800-point real STFT of 4 frames per workgroup as a "
                                                       "half-length complex FFT (mixed-radix Stockham 4 4 5 5 in LDS, float64) + split + mel "
                                                       "bands from LDS + log chain; mel_resample_k",
                             "achieved": round(mel_bytes / t_mel / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(mel_bytes / t_mel / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes": int(mel_bytes),
                             "us_at_hbm_peak": round(mel_bytes / (HBM_PEAK_GBS * 1e9) * 1e6, 1), "stft_frames": int(n_stft),
                             "fft_flop_per_stft_frame": 19000,
                             "note": "an FFT now, as the reference's np.fft.rfft (round 3: the DFT as an fp64 matrix-core product, 67x "
                                     "the arithmetic, 6.5 ms; option mel_fft = 0); the launch is bound by the float64 log10 / pow / log / "
                                     "exp chain of 80 mel values per frame and LDS traffic, not by HBM; 0.1 % of configs[4]"},
            "decode_s": round(t_dec, 3), "decode_s_all": [round(v, 3) for v in ts],
            "value": round((T - 1) / t_dec, 1), "unit": "frames/s",
            "x_realtime": round(minutes * 60.0 / (t_mel + t_se + t_dec), 1), "finite": finite,
            "frac_hbm": round(step_bytes(1) * (T - 1) / t_dec / 1e9 / HBM_PEAK_GBS, 4),
            "config": f"{minutes:g} min WAV -> mel -> speech encoder -> B=1 decode of {T - 1} frames, 1 GPU"}


def generate_30min(dev, minutes=30.0, exemplar_frames=CLIP_FRAMES):
    """BASELINE.json configs[4] through the REAL entry point: zeggs.generate.generate_gesture() (ZEGGS/generate.py:22-411)
    on a 30-minute 16 kHz WAV with a 7 200-frame exemplar BVH (also the first pose), random-init configs_v1 nets saved as
    the reference's whole-module pickles, loudness normalisation on.  Per-stage wall ms (device stages bracketed by a
    synchronisation; `_host` stages are Python / NumPy on the host: file parsing and BVH text, which north_star leaves there)."""
    import shutil
    import tempfile
    import scipy.io.wavfile as wavfile
    from zeggs import anim, generate
    tmp = Path(tempfile.mkdtemp(prefix="zeggs_gen30_"))
    try:
        net, data, res = tmp / "net", tmp / "data", tmp / "res"
        net.mkdir(), data.mkdir()
        torch.manual_seed(1234)
        se = modules.SpeechEncoder(synth.N_AUDIO, 64, SP)
        de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, SP, ST, H, 2)
        st = modules.StyleEncoder(synth.POSE_IN, 512, ST, type="attn", use_vae=True)
        torch.save(se, net / "speech_encoder.pt"), torch.save(de, net / "decoder.pt"), torch.save(st, net / "style_encoder.pt")
        np.savez(data / "stats.npz", **synth.make_stats())
        json.dump(synth.data_definition(), open(data / "data_definition.json", "w"))
        conf = dict(audio_conf=dict(pre_emphasis=False, pre_emph_coeff=0.97, centered=True, real_amplitude=True,
                                    normalize_mel_bins=True, normalize_range=True, min_clipping=1e-5, sampling_rate=16000,
                                    mel_fmin=20, mel_fmax=7600, n_mel_channels=80, filter_length=800, hop_length=200,
                                    resample_method="linear", normalize_loudness=True),
                    audio_feature_type=["mel_spec", "energy"])
        json.dump(conf, open(data / "data_pipeline_conf.json", "w"))
        n = int(minutes * 60 * 16000)
        rng = np.random.default_rng(2)
        env = 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * np.arange(n) / 16000.0)
        wavfile.write(tmp / "long.wav", 16000, (rng.standard_normal(n) * env * 3000).astype(np.int16))
        wavfile.write(tmp / "short.wav", 16000, synth.synth_wav(160000, seed=0))
        anim.bvh_save(tmp / "ex.bvh", synth.make_bvh_clip(exemplar_frames, seed=3))
        kw = dict(style_encoding_type="example", blend_type="add", blend_ratio=[1.0], first_pose=tmp / "ex.bvh",
                  temperature=1.0, seed=1234)
        generate.generate_gesture(tmp / "short.wav", [(tmp / "ex.bvh", None)], net, data, res, file_name="warm", **kw)
        torch.cuda.synchronize()
        generate.PROFILE = prof = {}
        t0 = time.perf_counter()
        enc = generate.generate_gesture(tmp / "long.wav", [(tmp / "ex.bvh", None)], net, data, res, file_name="out", **kw)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        generate.PROFILE = None
        with open(res / "out.bvh") as fh:
            for line in fh:
                if line.startswith("Frames:"):
                    frames = int(line.split()[1])
                    break
        # (round 4: long clips decode in chunks with the BVH rows converted / downloaded / formatted / written underneath the
        #  next chunks -- that one stage is device-bound with the host work hidden in it: counted with the device stages)
        dev_ms = sum(v for k, v in prof.items() if (k.endswith("_device") or "_device_with_" in k) and isinstance(v, (int, float)))
        host_ms = sum(v for k, v in prof.items() if k.endswith("_host"))
        return {"frames": frames, "total_s": round(total, 2), "device_stages_ms": round(dev_ms, 1),
                "host_stages_ms": round(host_ms, 1), "stages_ms": {k: (round(v, 2) if isinstance(v, (int, float)) else v) for k, v in prof.items()},
                "x_realtime_device_stages": round(minutes * 60e3 / dev_ms, 1), "x_realtime_total": round(minutes * 60 / total, 1),
                "bvh_bytes": (res / "out.bvh").stat().st_size, "style_encoding_finite": bool(torch.isfinite(enc).all()),
                "config": f"generate_gesture(): {minutes:g} min 16 kHz WAV, {exemplar_frames}-frame exemplar BVH (= first pose), "
                          f"loudness normalisation on, B=1 decode of {frames - 1} frames, BVH written"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def time_regions(step, first, steps, regions=3, after=None):
    """Seconds per step of the extras' engines: `regions` (>= 3) consecutive timed regions of `steps` steps each (synchronised on
    both sides); the MEDIAN of the regions is what is reported, every region is listed beside it (ms_per_step_regions: the spread is
    visible -- the extras run first, in fresh processes on a box whose files are still paging in: in two of three full runs of the
    second session the FIRST region of the first extra came out 4 % slow, 28.3 against 27.1 / 27.1 ms, and never when that extra was
    run on a warm box -- five variants, fifteen regions, 26.89-27.03).  The headline keeps the contract's single region of exactly K steps.
    after(it): called behind every step but a region's last (round 6, second session: the extras prefetch the next batch like the
    headline loop; until then they gathered every batch at the head of its own step, 0.3 ms on the caller's queue)."""
    per = []
    for r in range(regions):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for it in range(first + r * steps, first + (r + 1) * steps):
            out = step(it)
            if after is not None and it + 1 < first + (r + 1) * steps:      # (the next batch's prefetch, as the headline loop and
                after(it)                                                    #  zeggs.train() issue it: never across a region's end)
        torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / steps)
    return float(np.median(per)), [round(x * 1e3, 3) for x in per], out


def v2_label_b64(ds, dev, steps=10, warmup=6, batch=64, nlabels=9):
    """BASELINE.json configs[3]: configs_v2.json = label conditioning (one-hot over 9 labels, no style encoder),
    batch 64 x 256-frame windows: the MFMA-bound regime of the stage kernels (B >= 40)."""
    se, de, _ = build_nets(dev, style=nlabels, with_style_encoder=False)
    eng = engine.TrainEngine(se, de, None, ds, synth.PARENTS, synth.DT, style_encoding_type="label")
    perm = np.random.default_rng(42).permutation(len(ds))
    table = torch.as_tensor(np.eye(nlabels, dtype=np.float32)[np.arange(len(ds)) % nlabels]).to(dev)

    def indices(it):
        return engine.shard_indices(perm, it % (len(ds) // batch), batch, 1, 0)

    def step(it):
        idx = indices(it)
        # (the label rows through the dataset's pinned upload ring: a copy from pageable memory would put the host in line with the GPU
        #  once per step -- engine._IndexUploader)
        return eng.step(idx, None, labels=ops.gather_rows(table, ds.upload_indices(idx)))

    for it in range(warmup):        # (with the prefetch: its two buffer sets are allocated here, not in the first timed region)
        step(it)
        if it + 1 < warmup:
            eng.prefetch(indices(it + 1), None)
    dt_, regions_ms, _ = time_regions(step, warmup, steps, after=lambda it: eng.prefetch(indices(it + 1), None))
    fwd, bwd = sweep_ms(0) * 1e-3 / (WINDOW - 1), sweep_ms(1) * 1e-3 / (WINDOW - 1)
    fl = step_flops(batch, nlabels)
    return {"value": round(batch * WINDOW / dt_, 1), "unit": "frames/s", "ms_per_step": round(dt_ * 1e3, 3),
            "ms_per_step_regions": regions_ms, "batch": batch, "window": WINDOW, "steps": steps,
            "roofline": {"bound": "mfma", "kernel": ("train_fwd_persistent_k<4> (one weight-stationary launch per window)"
                                                     if ops.lib().zeggs_persistent_state(1) == 1 else
                                                     "3 launches of stage_k<4,...> per step") + ", forward step, fp32 MFMA",
                         "achieved": round(fl / fwd / 1e12, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(fl / fwd / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "us_per_step": round(fwd * 1e6, 2),
                         "backward_us_per_step": round(bwd * 1e6, 2),
                         "backward_kernel": ("train_bwd_persistent_k, two sweeps of 32 batch rows"
                                             if ops.lib().zeggs_persistent_state(2) == 1 else "3 launches of stage_k<4,...> per step"),
                         "backward_frac": round(fl / bwd / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                         "iteration_tflops": round(3 * (WINDOW - 1) * fl / dt_ / 1e12, 2), "traffic": None},
            "config": "configs_v2.json shape: label conditioning (9 one-hot labels), no style encoder, batch 64 x 256"}


def variants_b32(ds, dev, steps=5, warmup=3):
    """The option surface beyond the shipped configs: rnn_cond = "film" (RecurrentDecoderFiLM, ZEGGS/modules.py:188-227) and
    style_encoder.type = "gru" (StyleEncoderGRU, :307-343) at the headline shape.  Since round 4 both run on the fragment-packed
    stage kernels: the FiLM step is 4 launches per direction (its two modulated ELU layers add a dependent stage to the normal
    decoder's 3), the style recurrence one launch per exemplar frame and direction, replayed as a hipGraph.  Reference-pinned
    (tests/golden/variants.npz); the persistent H = 1024 sweeps decline rnn_cond = "film"."""
    torch.manual_seed(1234)
    se = modules.SpeechEncoder(synth.N_AUDIO, 64, SP).to(dev).train()
    de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, SP, ST, H, 2, rnn_cond="film").to(dev).train()
    st = modules.StyleEncoder(synth.POSE_IN, 512, ST, type="gru", use_vae=True).to(dev).train()
    eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT)
    perm = np.random.default_rng(42).permutation(len(ds))
    ops.set_option("timing", 1)
    ind = lambda it: engine.shard_indices(perm, it % (len(ds) // BATCH), BATCH, 1, 0)  # noqa: E731
    for it in range(warmup):        # (with the prefetch: its two buffer sets are allocated here, not in the first timed region)
        eng.step(ind(it), EXAMPLE_LEN)
        if it + 1 < warmup:
            eng.prefetch(ind(it + 1), EXAMPLE_LEN)
    dt_, regions_ms, loss = time_regions(lambda it: eng.step(ind(it), EXAMPLE_LEN), warmup, steps,
                                        after=lambda it: eng.prefetch(ind(it + 1), EXAMPLE_LEN))
    ms = ctypes.c_float(0.0)
    fwd_us = bwd_us = None
    if ops.lib().zeggs_timing_ms(0, ctypes.byref(ms)) == 0:
        fwd_us = ms.value * 1e3 / (WINDOW - 1)
    if ops.lib().zeggs_timing_ms(1, ctypes.byref(ms)) == 0:
        bwd_us = ms.value * 1e3 / (WINDOW - 1)
    ops.set_option("timing", 0)
    xd = synth.POSE_IN + SP       # film: the style is not a step input, it modulates (gamma / beta [B, 2H] per step)
    wbytes = 4 * (H * xd + 3 * H * (H + xd) + 3 * 3 * H * H + H * H + synth.POSE_OUT * H + 2 * H + 12 * H + synth.POSE_OUT)
    abytes = wbytes + BATCH * 4 * (xd + 2 * 2 * H + 2 * H + synth.POSE_OUT + 2 * H)
    out = {"value": round(BATCH * WINDOW / dt_, 1), "unit": "frames/s", "ms_per_step": round(dt_ * 1e3, 2), "ms_per_step_regions": regions_ms,
           "steps": steps,
           "finite": bool(torch.isfinite(loss)), "algorithmic_bytes_per_step": abytes,
           "config": "FiLM decoder + GRU style encoder (VAE), batch 32 x 256, example 384: fragment-packed stage kernels"}
    pmc = None
    if (ROOT / "profiles" / "r06_film_step_pmc.json").exists():      # counter passes of tools/pmc_variants.sh (separate runs)
        pmc = json.load(open(ROOT / "profiles" / "r06_film_step_pmc.json"))
    if fwd_us:
        out["roofline"] = {"bound": "hbm", "kernel": "stage_k (4 launches per FiLM decoder step and direction)",
                           "us_per_step": round(fwd_us, 2), "achieved": round(abytes / fwd_us / 1e3, 1), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(abytes / fwd_us / 1e3 / HBM_PEAK_GBS, 4),
                           "backward_us_per_step": round(bwd_us, 2) if bwd_us else None,
                           "backward_frac": round(abytes / bwd_us / 1e3 / HBM_PEAK_GBS, 4) if bwd_us else None,
                           "traffic": pmc["forward"]["traffic_bytes_per_step"] if pmc else None,
                           "backward_traffic": pmc["backward"]["traffic_bytes_per_step"] if pmc else None,
                           "traffic_source": "profiles/r06_film_step_pmc.json (FETCH_SIZE x2 + WRITE_SIZE)" if pmc else None}
    return out


def nhidden_512_b32(ds, dev, steps=5, warmup=3):
    """configs_v1.json with decoder.nhidden = 512 (ZEGGS/train.py:129 honours the option): the persistent sweeps are built for
    H = 1024 and decline, the fragment-packed stage kernels serve it (3 launches per step and direction) -- the fall-back
    path, measured (parity at this width: tests/test_gpu_parity.py::test_decoder_other_hidden_width_vs_oracle)."""
    torch.manual_seed(1234)
    H2 = 512
    se = modules.SpeechEncoder(synth.N_AUDIO, 64, SP).to(dev).train()
    de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, SP, ST, H2, 2).to(dev).train()
    st = modules.StyleEncoder(synth.POSE_IN, 512, ST, type="attn", use_vae=True).to(dev).train()
    eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT)
    perm = np.random.default_rng(42).permutation(len(ds))
    ops.set_option("timing", 1)
    ind = lambda it: engine.shard_indices(perm, it % (len(ds) // BATCH), BATCH, 1, 0)  # noqa: E731
    for it in range(warmup):        # (with the prefetch: its two buffer sets are allocated here, not in the first timed region)
        eng.step(ind(it), EXAMPLE_LEN)
        if it + 1 < warmup:
            eng.prefetch(ind(it + 1), EXAMPLE_LEN)
    dt_, regions_ms, loss = time_regions(lambda it: eng.step(ind(it), EXAMPLE_LEN), warmup, steps,
                                        after=lambda it: eng.prefetch(ind(it + 1), EXAMPLE_LEN))
    ms = ctypes.c_float(0.0)
    fwd_us = bwd_us = None
    if ops.lib().zeggs_timing_ms(0, ctypes.byref(ms)) == 0:
        fwd_us = ms.value * 1e3 / (WINDOW - 1)
    if ops.lib().zeggs_timing_ms(1, ctypes.byref(ms)) == 0:
        bwd_us = ms.value * 1e3 / (WINDOW - 1)
    ops.set_option("timing", 0)
    xd = synth.POSE_IN + SP + ST
    wbytes = 4 * (H2 * xd + 3 * H2 * (H2 + xd) + 3 * 3 * H2 * H2 + synth.POSE_OUT * H2 + H2 + 12 * H2 + synth.POSE_OUT)
    out = {"value": round(BATCH * WINDOW / dt_, 1), "unit": "frames/s", "ms_per_step": round(dt_ * 1e3, 2), "ms_per_step_regions": regions_ms,
           "steps": steps,
           "finite": bool(torch.isfinite(loss)), "algorithmic_bytes_per_step": wbytes,
           "config": "configs_v1 nets with decoder.nhidden = 512, batch 32 x 256, example 384: fragment-packed stage kernels "
                     "(the persistent sweeps serve H = 1024 only)"}
    if fwd_us:
        out["roofline"] = {"bound": "hbm", "kernel": "stage_k (3 launches per step)", "us_per_step": round(fwd_us, 2),
                           "achieved": round(wbytes / fwd_us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(wbytes / fwd_us / 1e3 / HBM_PEAK_GBS, 4),
                           "backward_us_per_step": round(bwd_us, 2) if bwd_us else None}
    return out


def tail_split_bf16(ds, dev, steps=10, warmup=4, nplanes=6):
    """EXPERIMENT, reported BESIDE the headline, never as it: the headline configuration (configs[1], B = 32 x 256)
    with the fp32 TN products of the tail on the bf16 matrix cores through the fp32-exact three-plane operand split (csrc/gemm_split.hip,
    option "gemm_split_bf16" = 6 plane products; default off; acceptance record profiles/r06_gemm_split_bf16.txt: error against float64
    <= the native fp32 MFMA kernel's on all five weight-gradient shapes).  roofline: the decoder's two largest weight-gradient
    products alone on the chip, priced against 2 500 / n TFLOP/s fp32-EQUIVALENT (n bf16 matrix instructions per fp32 product)."""
    se, de, st = build_nets(dev)
    ops.set_option("gemm_split_bf16", nplanes)
    try:
        eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT)
        perm = np.random.default_rng(42).permutation(len(ds))
        idx = lambda it: engine.shard_indices(perm, it % (len(ds) // BATCH), BATCH, 1, 0)  # noqa: E731
        loss = None
        for it in range(warmup):        # (the headline's loop: the next batch prefetched behind every step)
            loss = eng.step(idx(it), EXAMPLE_LEN)
            if it + 1 < warmup:
                eng.prefetch(idx(it + 1), EXAMPLE_LEN)
        dt_, regions_ms, loss = time_regions(lambda it: eng.step(idx(it), EXAMPLE_LEN), warmup, steps,
                                            after=lambda it: eng.prefetch(idx(it + 1), EXAMPLE_LEN))
        del eng
        peak = 2500.0 / nplanes
        kern = {}
        for name, (M, N, K) in {"dW_hh 3072x1024 K=8160": (3072, 1024, 8160), "dW_ih0 3072x2286 K=8160": (3072, 2288, 8160)}.items():
            A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
            Cm = torch.zeros(M, N, device=dev)
            f = lambda: ops.gemm(A, B, Cm, M, N, K, (1, M), (N, 1), (N, 1))  # noqa: E731
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                f()
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 8
            kern[name] = {"us": round(us, 1), "achieved": round(2.0 * M * N * K / us / 1e6, 1), "frac": round(2.0 * M * N * K / us / 1e6 / peak, 3)}
    finally:
        ops.set_option("gemm_split_bf16", 0)
    return {"value": round(BATCH * WINDOW / dt_, 1), "unit": "frames/s", "ms_per_step": round(dt_ * 1e3, 3), "ms_per_step_regions": regions_ms,
            "final_loss": round(float(loss.detach()), 4), "option": f"gemm_split_bf16={nplanes}", "default": "off (the headline is the native fp32 run)",
            "roofline": {"bound": "mfma", "peak": round(peak, 1), "unit": "TFLOP/s fp32-equivalent (2 M N K / time; 2 500 dense bf16 / n plane products)",
                         "kernel": f"gemm_tn_split_kernel<{nplanes}>: v_mfma_f32_32x32x16_bf16 on three truncated bf16 planes per fp32 operand, alone on the chip",
                         "kernels": kern, "acceptance": profile_stamp("profiles/r06_gemm_split_bf16.txt")}}


EXTRA_ENGINES = {"v2_label_b64": v2_label_b64, "variants_film_gru_b32": variants_b32, "nhidden_512_b32": nhidden_512_b32,
                 "tail_split_bf16": tail_split_bf16}


def extra_in_fresh_process(key, timeout=240):
    """`python bench.py --extra <key>` in a child process: the extra's engine is then the FIRST of its process, like the headline's
    (a later engine of one process gets its streams from further down PyTorch's pool; measured up to 2.3 x slower on some boxes).
    Returns the child's JSON, or None (the caller then runs the extra in-process)."""
    try:
        r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--extra", key], capture_output=True, text=True,
                           timeout=timeout, env=launch_env())
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            res = json.loads(lines[-1])
            res["process"] = "own (python bench.py --extra %s)" % key
            return res
    except (subprocess.SubprocessError, OSError, ValueError):
        pass
    return None


# ----------------------------------------------------------------------------- launcher
def launch_command(argv, gpus, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)


def launch_env(base=None):
    """Environment of the ranks: dmabuf IPC for RCCL, 8 hardware queues (the engine's three streams + RCCL's must not share
    one: DESIGN.md section 6), a bounded host thread pool per rank."""
    env = dict(os.environ if base is None else base)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    env.setdefault("OMP_NUM_THREADS", "4")
    return env


def rank_device(local_rank):
    """one process per GPU: LOCAL_RANK k drives GPU k of the node"""
    return torch.device("cuda", int(local_rank))


def self_launch(a):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return subprocess.call(launch_command(sys.argv[1:], a.gpus, port), env=launch_env())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="ONE blocking all-reduce after the backward instead of the overlapped decoder-slice exchange")
    ap.add_argument("--no-wgrad-overlap", action="store_true",
                    help="decoder weight-gradient GEMMs on the main stream instead of beside the encoders' backward")
    ap.add_argument("--no-early-step", action="store_true",
                    help="the whole optimizer step at the end of the iteration (default: the decoder's slice on the weight-gradient stream)")
    ap.add_argument("--no-prefetch", action="store_true", help="gather every batch at the start of its own step")
    ap.add_argument("--no-extras", action="store_true", help="skip decode / decode_30min / v2_label_b64 (N = 1 only)")
    ap.add_argument("--no-generate", action="store_true", help="skip generate_30min (configs[4] through generate_gesture())")
    ap.add_argument("--force-process-group", action="store_true",
                    help="initialise the RCCL process group and run the gradient all-reduce even at --gpus 1")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="only rendezvous (gloo on CPU when no GPU is visible) and print the rank census")
    ap.add_argument("--extra", choices=sorted(EXTRA_ENGINES), default=None,
                    help="run ONE of the extra training configurations alone and print its JSON (what the default run does in a "
                         "fresh process per extra: a second or third engine in one process maps its streams onto hardware queues "
                         "less luckily than the first, tools/prio_probe.py)")
    a = ap.parse_args()
    if a.extra:
        dev = rank_device(0)
        torch.cuda.set_device(dev)
        ops.set_option("timing", 1)
        ds = engine.DeviceDataset(build_dataset(), WINDOW, dev)
        print(json.dumps(EXTRA_ENGINES[a.extra](ds, dev)))
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    # The extra training configurations run FIRST, each in a process of its own, before this process has touched the GPU: a
    # second engine in one process runs slower than the first on some boxes (stream -> hardware-queue mapping), and a child
    # that starts while its parent holds queues on the same GPU is scheduled against them (FiLM: 42 instead of 36 ms).
    pre_extras = {}
    if a.gpus == 1 and "WORLD_SIZE" not in os.environ and not a.no_extras and not a.launch_selftest:
        for key in EXTRA_ENGINES:
            pre_extras[key] = extra_in_fresh_process(key)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    have_gpu = torch.cuda.is_available()
    use_pg = world > 1 or a.force_process_group
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        if have_gpu:
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world,
                                                 device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: run `python bench.py --gpus {a.gpus}` (self-launching) "
                         f"or torch.distributed.run with --nproc-per-node {a.gpus}")
    census = None
    if use_pg:
        me = {"rank": rank, "local_rank": local,
              "device": str(torch.cuda.get_device_properties(local).uuid) if have_gpu else "cpu"}
        census = [None] * world
        torch.distributed.all_gather_object(census, me)
    if a.launch_selftest:
        if rank == 0:
            print(json.dumps({"launch_selftest": True, "world_size": world, "ranks": census}))
        if use_pg:
            torch.distributed.destroy_process_group()
        return
    if have_gpu and local >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {torch.cuda.device_count()} GPUs are visible")
    dev = rank_device(local)
    torch.cuda.set_device(dev)
    ops.set_option("timing", 1)
    shared = None
    if world > 1:       # one synthetic-dataset build per node: rank 0 writes /dev/shm, the others map it
        shm = Path("/dev/shm" if Path("/dev/shm").is_dir() else "/tmp") / f"zeggs_bench_{os.environ.get('MASTER_PORT', '0')}.npz"
        shared = (shm, rank == 0, torch.distributed.barrier)
    t_data = time.perf_counter()
    data = build_dataset(shared=shared)
    t_data = time.perf_counter() - t_data
    if shared is not None:
        torch.distributed.barrier()
        if rank == 0:
            shm.unlink(missing_ok=True)
    ds = engine.DeviceDataset(data, WINDOW, dev)
    se, de, st = build_nets(dev)
    eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT, world_size=world, rank=rank,
                             force_allreduce=a.force_process_group, overlap_allreduce=not a.no_overlap,
                             overlap_wgrads=not a.no_wgrad_overlap, early_decoder_step=not a.no_early_step)
    ops.manual_seed(1000 + rank)                            # per-rank noise streams (dropout masks, VAE eps)
    perm = np.random.default_rng(42).permutation(len(ds))   # same permutation on every rank
    gb = BATCH * world

    def indices(it):
        return engine.shard_indices(perm, it % (len(ds) // gb), BATCH, world, rank)

    def sync():
        if use_pg:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    step_events = []

    def run(first, last):       # steps first .. last-1; the next batch is gathered beside the current step (never across `last`:
        for it in range(first, last):                       # the timed region gathers exactly its own K batches)
            eng.step(indices(it), EXAMPLE_LEN)
            if it + 1 < last and not (a.no_wgrad_overlap or a.no_prefetch):
                eng.prefetch(indices(it + 1), EXAMPLE_LEN)
            if not os.environ.get("ZEGGS_BENCH_NO_STEP_EVENTS"):      # (tools/gap_probe.sh: does the per-step event cost the iteration anything?)
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                step_events.append(e)

    run(0, a.warmup)
    sync()
    eng.allreduce_events = [] if use_pg else None
    sync()
    t0 = time.perf_counter()
    run(a.warmup, a.warmup + a.steps)
    sync()
    mine = time.perf_counter() - t0
    if os.environ.get("ZEGGS_BENCH_STEP_TIMES"):
        print("step ends (ms since the previous):", " ".join(f"{x.elapsed_time(y):.2f}" for x, y in zip(step_events, step_events[1:])),
              file=sys.stderr)
    fwd_in, bwd_in = sweep_ms(0), sweep_ms(1)               # the LAST timed iteration's stage sweeps (HIP events)
    # per-iteration times from the HIP events recorded behind every optimizer step of the timed region
    ev = step_events[a.warmup:a.warmup + a.steps]          # (intervals between the timed steps only: K - 1 of them)
    step_ms = np.array([x.elapsed_time(y) for x, y in zip(ev, ev[1:])]) if len(ev) > 1 else np.zeros(0)
    el = torch.tensor([mine], device=dev, dtype=torch.float64)
    per_rank = None
    if use_pg:
        allt = [torch.zeros_like(el) for _ in range(world)]
        torch.distributed.all_gather(allt, el)
        per_rank = [round(float(t) / a.steps * 1e3, 3) for t in allt]
        torch.distributed.all_reduce(el, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(el)
    fw, bw = [fwd_in], [bwd_in]
    loss = None
    for k in range(3):                                       # a few more iterations, sweeps read after each one
        loss = eng.step(indices(a.warmup + a.steps + k), EXAMPLE_LEN)
        fw.append(sweep_ms(0))
        bw.append(sweep_ms(1))
    loss = float(loss.detach())
    ar_blocking = None
    if use_pg:      # the whole exchange un-overlapped (3 more iterations): what the overlap hides = blocking - exposed
        keep, eng.overlap_allreduce, eng.allreduce_events = (eng.overlap_allreduce, list(eng.allreduce_events or [])), False, []
        for k in range(3):
            eng.step(indices(a.warmup + a.steps + 3 + k), EXAMPLE_LEN)
        torch.cuda.synchronize()
        ar_blocking = float(np.mean([e0.elapsed_time(e1) for e0, e1 in eng.allreduce_events]))
        eng.overlap_allreduce, eng.allreduce_events = keep
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        out = {
            "metric": "train_frames_per_sec", "value": round(gb * WINDOW * a.steps / elapsed, 1), "unit": "frames/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs_v1.json nets (25.5M params, random-init), batch 32/GPU x 256-frame windows "
                                   "from synthetic 60fps 2-min clips, style example 384 frames, full train step "
                                   "(gather+fwd+loss+bwd+allreduce+RAdam)",
                       "global_batch": gb, "window": WINDOW, "parallelism": f"dp{world}"},
            "final_loss": round(loss, 4),
        }
        if len(step_ms):
            out["step_ms_hip_events"] = {"n": int(len(step_ms)), "p50": round(float(np.percentile(step_ms, 50)), 3),
                                         "p95": round(float(np.percentile(step_ms, 95)), 3),
                                         "min": round(float(step_ms.min()), 3), "max": round(float(step_ms.max()), 3)}
        out["dataset_build_s"] = round(t_data, 2)
        nst = WINDOW - 1
        f_us, b_us = float(np.mean(fw)) * 1e3 / nst, float(np.mean(bw)) * 1e3 / nst
        pmc = None
        for name in ("r06_decoder_step_pmc.json", "r05_decoder_step_pmc.json", "r04_decoder_step_pmc.json", "r03_decoder_step_pmc.json", "r02_decoder_step_pmc.json", "r01_decoder_step_pmc.json"):
            if (ROOT / "profiles" / name).exists():
                pmc = json.load(open(ROOT / "profiles" / name))
                pmc["file"] = "profiles/" + name
                break
        ach = step_bytes(BATCH) / (f_us * 1e-6) / 1e9
        ach_b = step_bytes(BATCH) / (b_us * 1e-6) / 1e9
        out["roofline"] = {
            "bound": "hbm", "kernel": ("train_fwd_persistent_k: the 255 forward steps of a window as ONE weight-stationary "
                                       "launch (3 phases per step, weights in registers)"
                                       if ops.lib().zeggs_persistent_state(1) == 1 else
                                       "stage_k, decoder forward step = 3 launches (GRU l0, GRU l1, layer2 + pose "
                                       "integration + next step's layer0)") + ", per-step figures at B=32",
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": pmc.get("traffic_bytes_per_step") if pmc else None,
            "traffic_source": profile_stamp(pmc["file"]) if pmc else None,
            "us_per_step": round(f_us, 2), "us_per_step_in_timed_region": round(fwd_in * 1e3 / nst, 2),
            "algorithmic_bytes_per_step": step_bytes(BATCH),
            "backward": {"kernel": ("train_bwd_persistent_k: the 255 BPTT steps of a window as ONE weight-stationary launch "
                                    "(4 phases per step, transposed weights as 4-row v_mfma_f32_4x4x1 tiles in registers / LDS)"
                                    if ops.lib().zeggs_persistent_state(2) == 1 else
                                    "stage_k, BPTT step = 3 launches (transposed packs)"), "achieved": round(ach_b, 1),
                         "frac": round(ach_b / HBM_PEAK_GBS, 4), "us_per_step": round(b_us, 2),
                         "us_per_step_in_timed_region": round(bwd_in * 1e3 / nst, 2),
                         "traffic": pmc.get("traffic_bytes_per_step_backward") if pmc else None},
            "mfma_frac_at_b32": round(step_flops(BATCH) / (f_us * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
            "dominant_kernel": dominant_kernel_note(),
            "note": "HIP events (library hook zeggs_timing_ms, recorded on the stream the kernels run on) around the "
                    "255-step stage sweeps: the last timed iteration + 3 more; traffic = FETCH_SIZE(x2)+WRITE_SIZE "
                    f"from {pmc['file'] if pmc else 'n/a'}; at B=32 the step is also at the fp32 MFMA ridge"}
        if use_pg:
            ar = [e0.elapsed_time(e1) for e0, e1 in eng.allreduce_events[:a.steps]]
            out["rccl_ranks"] = {"world_size": torch.distributed.get_world_size(), "ranks": census}
            out["per_rank_ms_per_step"] = per_rank
            out["allreduce_ms"] = round(float(np.mean(ar)), 3) if ar else None
            out["allreduce_exposed_ms"] = out["allreduce_ms"]
            out["allreduce_blocking_ms"] = round(ar_blocking, 3)
            out["allreduce_overlapped_ms"] = round(max(0.0, ar_blocking - (out["allreduce_ms"] or 0.0)), 3) if eng.overlap_allreduce else 0.0
            out["per_gpu_frames_per_sec"] = round(BATCH * WINDOW * a.steps / elapsed, 1)
            # what one of these ranks would do alone: the iteration without the exposed part of the exchange (to be checked
            # against the N = 1 line of the same build)
            out["n1_equivalent_frames_per_sec"] = round(BATCH * WINDOW / max(1e-9, (ms - (out["allreduce_ms"] or 0.0)) * 1e-3), 1)
            out["allreduce_bytes"] = int(eng.flat_gx.numel() * 4)
            out["allreduce_note"] = ("allreduce_ms = the EXPOSED part: the decoder slice (91 % of the payload) is reduced "
                                     "underneath the encoders' backward" if eng.overlap_allreduce else
                                     "one blocking all-reduce of the flat gradient buffer after the backward")
        out["streams"] = ("single stream" if eng.wgrad_stream is None else
                          "decoder weight-gradient GEMMs and the weight-only packs of the next sweeps on the library's second "
                          "stream, speech encoder and the next batch's gather on a third"
                          + ("" if not a.no_prefetch else " (no batch prefetch)"))
        out["cpu_baseline"] = None
        if world == 1 and not a.no_extras:
            out["wgrad_gemm_alone"] = wgrad_gemm_alone(dev)
            out["decode"] = decode_rate(de, dev)
            out["roofline"]["decode_b1"] = out["decode"]["roofline"]
            out["decode_30min"] = decode_30min(se, de, dev)
            if not a.no_generate:
                out["generate_30min"] = generate_30min(dev)
            del eng
            for key, fn in EXTRA_ENGINES.items():
                out[key] = pre_extras.get(key) or fn(ds, dev)
        if world == 1 and not a.no_cpu_baseline:
            import contextlib
            with contextlib.redirect_stdout(sys.stderr):      # the reference's train() writes its progress bar to stdout
                train_b, dec_b, mel_b = cpu_baselines(data)
            sys.stderr.write("\n")
            out["cpu_baseline"] = train_b
            if "decode" in out:
                out["decode"]["cpu_baseline"] = dec_b
            out["mel_cpu_baseline"] = mel_b
    # RCCL writes its version banner through C stdio, which a pipe buffers until exit: every rank pushes it out BEFORE the last
    # barrier, so that rank 0's JSON line is the last thing on stdout
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if use_pg:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
