"""Time the UNMODIFIED reference's CPU path (the baseline BASELINE.md section 3 / SURVEY.md 8(d) ask for).

TEST / MEASUREMENT INFRASTRUCTURE -- see oracle/__init__.py.  Needs the reference: /root/reference (build container) or
the snapshot oracle/build_ref.py left in oracle/_ref/ (it travels to the GPU box); bench.py calls `measure()` for its
`cpu_baseline` leg when either is present (kind = "reference") and otherwise falls back to the oracle port plus the
figures this script recorded in profiles/r02_cpu_reference.json.

    python -m oracle.ref_timing [--iters 5] [--frames 3000] [--out profiles/r02_cpu_reference.json]

Three legs, all through oracle/ref_shims.py (third-party stubs only, no reference source modified):
  train   train.train() (ZEGGS/train.py:29) on the configs[1] workload shape -- batch 32 x 256-frame windows of
          synthetic 60-fps 2-minute clips, configs_v1 nets, CPU -- timestamps taken in a wrapped RAdam.step,
          iteration 0 (checkpoint + sample rendering) skipped; once with thread_count=1 (the shipped value,
          configs_v1.json:37) and once with every core
  decode  Decoder.forward, B=1, torch.no_grad, 1 thread (what generate.py:88 enforces) and all threads
  mel     data_pipeline.preprocess_audio (ZEGGS/data_pipeline.py:33) on the 10 s synthetic WAV
"""
import argparse
import json
import os
import platform
import random
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "ubisoft-laforge-zeroeggs_amd"))
from oracle import ref_shims  # noqa: E402
from zeggs import synth  # noqa: E402

BATCH, WINDOW, CLIP_FRAMES = 32, 256, 7200
NET_OPT = {
    "decoder": {"nhidden": 1024, "num_rnn_layers": 2, "rnn_cond": "normal"},
    "speech_encoder": {"nhidden": 64, "speech_encoding_size": 64},
    "style_encoder": {"nhidden": 512, "style_encoding_size": 64, "example_length": 256, "type": "attn", "use_vae": True},
}


class _Done(Exception):
    pass


def cpu_info():
    model = platform.processor() or "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model, "logical_cores": os.cpu_count()}


def time_train(ref, data_dir, threads, iters):
    """-> seconds per steady-state iteration of the reference train() (list), frames/s."""
    stamps = []
    orig_step = ref.optimizers.RAdam.step

    def step(self, closure=None):
        r = orig_step(self, closure)
        stamps.append(time.perf_counter())
        if len(stamps) >= iters + 2:        # iteration 0 (+ checkpoint/samples) and one more warm-up are dropped
            raise _Done()
        return r

    ref.optimizers.RAdam.step = step
    ref.train.RAdam.step = step
    # Iteration 0 of the reference always renders six 30-second sample clips (train.py:477-760: ~11 000 frames of B=1 decode,
    # about a minute of CPU) before the steady state starts; that interval is dropped from the timing anyway, so the harness
    # asks for 1-second clips there (a wrapper around SGDataset.get_sample: no reference source is touched)
    orig_sample = ref.dataset.SGDataset.get_sample

    def short_sample(self, dataset, length=None, range_index=None):
        return orig_sample(self, dataset, 1, range_index)

    ref.dataset.SGDataset.get_sample = short_sample
    tmp = Path(tempfile.mkdtemp(prefix="zeggs_reftime_"))
    (tmp / "models").mkdir(), (tmp / "logs").mkdir()
    random.seed(0)
    opt = dict(niterations=160, batchsize=BATCH, window=WINDOW, change_pace=True, learning_rate=1e-4,
               learning_rate_decay=0.995, eps=1e-5, resume=False, use_gpu=False, thread_count=threads, seed=1234,
               use_tensorboard=False, style_encoding_type="example", generate_samples_step=5000, use_script=False)
    try:
        ref.train.train(tmp / "models", tmp / "logs", data_dir / "processed_data.npz", data_dir / "data_definition.json",
                        opt, NET_OPT)
    except _Done:
        pass
    finally:
        ref.optimizers.RAdam.step = orig_step
        ref.train.RAdam.step = orig_step
        ref.dataset.SGDataset.get_sample = orig_sample
    dts = np.diff(np.array(stamps))[1:]          # drop the interval that contains iteration 0's checkpoint + samples
    return [float(x) for x in dts], float(BATCH * WINDOW / np.mean(dts))


def time_decode(ref, threads, frames):
    torch.set_num_threads(threads)
    torch.manual_seed(1234)
    de = ref.modules.Decoder(pose_input_size=synth.POSE_IN, pose_output_size=synth.POSE_OUT, speech_encoding_size=64,
                             style_encoding_size=64, hidden_size=1024, num_rnn_layers=2).eval()
    stats = synth.make_stats()
    c = synth.make_clip(frames, seed=1, stats=stats)
    t = lambda k: torch.as_tensor(np.asarray(stats[k]), dtype=torch.float32)  # noqa: E731
    W = {k: torch.as_tensor(v[None]) for k, v in c.items()}
    speech, style = torch.randn(1, frames, 64) * 0.5, torch.randn(1, frames, 64) * 0.5
    with torch.no_grad():
        t0 = time.perf_counter()
        de(W["Y_root_pos"][:, 0], W["Y_root_rot"][:, 0], W["Y_root_vel"][:, 0], W["Y_root_vrt"][:, 0], W["Y_lpos"][:, 0],
           W["Y_ltxy"][:, 0], W["Y_lvel"][:, 0], W["Y_lvrt"][:, 0], W["Y_gaze_pos"], speech, style,
           torch.LongTensor(synth.PARENTS), t("anim_input_mean"), t("anim_input_std"), t("anim_output_mean"),
           t("anim_output_std"), synth.DT)
        dt = time.perf_counter() - t0
    return dt, (frames - 1) / dt


def time_mel(ref, seconds=10):
    conf = json.load(open(ref_shims.root() / "data" / "processed_v1" / "data_pipeline_conf.json"))
    conf["audio_conf"]["normalize_loudness"] = False       # pyloudnorm is not installed
    ac = ref.DictConfig(conf["audio_conf"])
    n = 16000 * seconds
    wav = synth.synth_wav(n, seed=0).astype(np.float32) / 32768.0
    nfr = int(round(60.0 * seconds))
    torch.set_num_threads(1)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        ref.data_pipeline.preprocess_audio(wav, 60, nfr, ac, feature_type=conf["audio_feature_type"])
        ts.append(time.perf_counter() - t0)
    return min(ts), nfr / min(ts)


def measure(iters=5, frames=3000, train_threads=(1, None), legs=("train", "decode", "mel")):
    assert ref_shims.available(), "/root/reference or the oracle/_ref snapshot (oracle/build_ref.py) is required"
    ref = ref_shims.load()
    try:
        return _measure(ref, iters, frames, train_threads, legs)
    finally:
        ref_shims.release()


def _measure(ref, iters, frames, train_threads, legs):
    ncpu = os.cpu_count() or 1
    out = {"cpu": cpu_info(), "reference": ref_shims.source() + " through oracle/ref_shims.py",
           "workload": f"configs_v1.json nets, batch {BATCH} x {WINDOW}-frame windows of synthetic 60-fps 2-minute "
                       f"clips (8 clips resident), style example length drawn in [256, 512] as train.py:228 does"}
    if "train" in legs:
        tmp = Path(tempfile.mkdtemp(prefix="zeggs_refdata_"))
        synth.write_dataset(tmp, n_train=8, n_valid=1, nframes=CLIP_FRAMES, seed=0)
        out["train"] = {}
        for th in train_threads:
            th = ncpu if th is None else th
            dts, fps = time_train(ref, tmp, th, iters)
            out["train"][f"threads_{th}"] = {"threads": th, "iterations_timed": len(dts), "s_per_iteration": dts,
                                             "frames_per_s": round(fps, 1)}
    if "decode" in legs:
        out["decode"] = {}
        for th in sorted({1, min(ncpu, 16)}):      # (the B=1 rollout does not scale with threads: generate.py:88 pins it to 1)
            dt, fps = time_decode(ref, th, frames)
            out["decode"][f"threads_{th}"] = {"threads": th, "frames": frames, "seconds": round(dt, 3),
                                              "frames_per_s": round(fps, 1)}
    if "mel" in legs:
        dt, fps = time_mel(ref)
        out["mel"] = {"threads": 1, "audio_seconds": 10, "seconds": round(dt, 4), "anim_frames_per_s": round(fps, 1),
                      "x_realtime": round(10.0 / dt, 1)}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--out", default=str(ROOT / "profiles" / "r02_cpu_reference.json"))
    a = ap.parse_args()
    res = measure(a.iters, a.frames)
    # the oracle PORT on the same box (bench.py falls back to it where /root/reference is absent): the port/reference
    # ratio measured here is what makes a port figure from another host comparable
    sys.path.insert(0, str(ROOT))
    import bench
    threads = min(bench.physical_cores(), 16)
    fps, dt_ = bench.cpu_port_train(bench.build_dataset(n_train=4), threads)
    res["port_on_build_box"] = {"train": {"threads": threads, "frames_per_s": round(fps, 1), "s_per_iteration": round(dt_, 2)},
                                "decode": {"threads": 1, "frames_per_s": round(bench.cpu_port_decode(), 1)},
                                "mel": {"threads": 1, "anim_frames_per_s": round(bench.cpu_port_mel(), 1)}}
    Path(a.out).write_text(json.dumps(res, indent=1))
    print(json.dumps(res))
