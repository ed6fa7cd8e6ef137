"""One training-shaped rollout (B = 32, T = 64, forward + BPTT) and one B = 1 decode of 2 000 frames on WHATEVER build ZEGGS_LIB
names -> npz (outputs, gradient samples).  tests/test_gpu_full_shapes.py::test_fast_gate_build_equals_exact_gate_build runs it
once per build (the shipped library: hardware exp2 / rcp gates, polynomial sin / cos; libzeggs_exact.so: -DZEGGS_EXACT_GATES=1
-DZEGGS_EXACT_SINCOS=1, libm throughout) and compares."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "ubisoft-laforge-zeroeggs_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import helpers  # noqa: E402
from zeggs import ops, synth  # noqa: E402

DEV = "cuda:0"
out = {}
_, de, _ = helpers.build_nets()
de = de.to(DEV)
s = helpers.real_stats_tensors("v1", device=DEV)
g = lambda t: t.to(DEV)  # noqa: E731
KEYS = ("Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos", "Y_ltxy", "Y_lvel", "Y_lvrt")
# training rollout + BPTT
W, speech, style = helpers.full_decoder_inputs(helpers.real_stats("v1"), 32, 64, 515)
sp = g(speech).requires_grad_(True)
O = de.train()(*[g(W[k][:, 0].contiguous()) for k in KEYS], g(W["Y_gaze_pos"]), sp, g(style), None, s["in_mean"], s["in_std"],
               s["out_mean"], s["out_std"], synth.DT)
sum((o * o).mean() for o in O).backward()
torch.cuda.synchronize()
out["train_pose"] = helpers.pack_pose(*[o.detach().cpu() for o in O[2:]]).numpy()
out["train_root"] = torch.cat([O[0].detach().cpu(), O[1].detach().cpu()], dim=-1).numpy()
out["train_grads"] = np.concatenate([p.grad.flatten()[torch.as_tensor(helpers.sample_idx(p.numel()), device=DEV)].cpu().numpy()
                                     for p in de.parameters()] + [sp.grad.flatten()[::97].cpu().numpy()])
out["train_gmax"] = np.array([float(p.grad.abs().max()) for p in de.parameters()])
# B = 1 decode
W, speech, style = helpers.long_decoder_inputs(helpers.real_stats("v1"), 2000, 9300)
with torch.no_grad():
    O = de.eval()(*[g(W[k][:, 0].contiguous()) for k in KEYS], g(W["Y_gaze_pos"]), g(speech), g(style), None, s["in_mean"],
                  s["in_std"], s["out_mean"], s["out_std"], synth.DT)
torch.cuda.synchronize()
out["decode_pose"] = helpers.pack_pose(*[o.cpu() for o in O[2:]]).numpy()[0, ::10]
out["decode_root"] = torch.cat([O[0].cpu(), O[1].cpu()], dim=-1).numpy()[0, ::10]
out["persistent"] = np.array([ops.lib().zeggs_persistent_state(k) for k in range(3)])
np.savez(sys.argv[1], **out)
print("ab_rollout:", ops._LIB_PATH.name, out["persistent"])
