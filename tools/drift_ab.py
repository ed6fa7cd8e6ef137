"""Attribution of the long free-running decode's drift: the 108 000-frame B = 1 rollout of
tests/golden/full_rollout108k.npz on ONE variant of the engine, against the reference's fp64 run (and its own fp32 run).

    ZEGGS_LIB=<alt build> ZEGGS_OPTIONS=<switches> python tools/drift_ab.py <tag> [out.jsonl] [perturb]

Variants (tools/drift_ab.sh runs them all): the shipped persistent kernel; the same built with -DZEGGS_EXACT_GATES=1 (libm
expf / tanhf instead of v_exp_f32 / v_rcp_f32), with -DZEGGS_EXACT_SINCOS=1 (sinf / cosf instead of the two polynomials of the
root integration), with both; the stage launches (persistent=0: same folds M / N0, another summation order); the generic
per-step GEMM path (decoder_fast=0: NO algebraic folds), also with both exact builds -- the closest this engine gets to the
reference's arithmetic, only the summation order inside the GEMMs differs.  `perturb` adds 1e-7-relative noise to the speech
encoding: the distance between two runs that differ by one rounding error in the input is the intrinsic divergence of the
rollout, the yardstick for every other row.  Prints one JSON row (max |HIP - fp64| per frame range and channel group)."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "ubisoft-laforge-zeroeggs_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import helpers  # noqa: E402
from zeggs import ops, synth  # noqa: E402

tag = sys.argv[1]
out = sys.argv[2] if len(sys.argv) > 2 else None
perturb = len(sys.argv) > 3 and sys.argv[3] == "perturb"
DEV = "cuda:0"
gd = np.load(ROOT / "tests" / "golden" / "full_rollout108k.npz")
_, de, _ = helpers.build_nets()
T = int(gd["T"])
W, speech, style = helpers.long_decoder_inputs(helpers.real_stats("v1"), T, int(gd["seed"]))
if perturb:
    speech = speech * (1.0 + 1e-7 * torch.as_tensor(np.random.default_rng(1).standard_normal(tuple(speech.shape)).astype(np.float32)))
s = helpers.real_stats_tensors("v1", device=DEV)
g = lambda t: t.to(DEV)  # noqa: E731
fp = [g(W[k][:, 0].contiguous()) for k in ("Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos", "Y_ltxy",
                                           "Y_lvel", "Y_lvrt")]
import time  # noqa: E402
with torch.no_grad():
    de = de.to(DEV).eval()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    O = de(*fp, g(W["Y_gaze_pos"]), g(speech), g(style), None, s["in_mean"], s["in_std"], s["out_mean"], s["out_std"], synth.DT)
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
O = [o.detach().cpu().double() for o in O]
pose = helpers.pack_pose(*O[2:]).numpy()[0][::500]
J = synth.NJ
err = np.abs(pose - gd["pose_every500"])
ltxy = err[:, 6 + 3 * J:6 + 9 * J].max(axis=1)
allc = err.max(axis=1)
ref = gd["ref_fp32_pose_err_every500"]
e_pos = np.abs(O[0].numpy()[0][::100] - gd["root_pos_every100"]).max(axis=1)
ref_pos = gd["ref_fp32_root_pos_err_every100"]
e_rot = np.abs(O[1].numpy()[0][::100] - gd["root_rot_every100"]).max(axis=1)
rng = ((0, 5), (5, 21), (21, 61), (61, 121), (121, 217))
row = dict(tag=tag, lib=str(ops._LIB_PATH.name), options=dict(ops._OPTIONS), perturbed_input=perturb, seconds=round(secs, 3),
           persistent_state=int(ops.lib().zeggs_persistent_state(0)),
           frames=[f"<= {(hi - 1) * 500}" for _, hi in rng],
           all_channels=[float(allc[lo:hi].max()) for lo, hi in rng],
           ltxy=[float(ltxy[lo:hi].max()) for lo, hi in rng],
           reference_fp32_all_channels=[float(ref[lo:hi].max()) for lo, hi in rng],
           root_pos_at=[10000, 30000, 60000, T - 1],
           root_pos=[float(e_pos[100]), float(e_pos[300]), float(e_pos[600]), float(e_pos[-1])],
           reference_fp32_root_pos=[float(ref_pos[100]), float(ref_pos[300]), float(ref_pos[600]), float(ref_pos[-1])],
           root_rot_end=float(e_rot.max()), finite=bool(np.isfinite(pose).all()))
print(json.dumps(row))
if out:
    with open(out, "a") as f:
        f.write(json.dumps(row) + "\n")
    np.save(Path(out).with_suffix("").as_posix() + f"_{tag}_pose500.npy", pose.astype(np.float32))
    np.save(Path(out).with_suffix("").as_posix() + f"_{tag}_root100.npy",
            np.concatenate([O[0].numpy()[0][::100], O[1].numpy()[0][::100]], axis=1))
