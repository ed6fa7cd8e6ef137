"""The fused training loss alone at the headline shape (B 32 x T 256, the 75-joint rig): wall time per call (forward + backward
kernels, HIP events around N calls) and a hash of every output -- run once per library build (ZEGGS_LIB=...) to A/B a kernel
change and to see whether it is bit-identical.  usage: python tools/loss_probe.py [B T]"""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ubisoft-laforge-zeroeggs_amd"))
from zeggs import ops, synth  # noqa: E402

B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 256)
dev = torch.device("cuda:0")
torch.manual_seed(0)
stats = synth.make_stats()
J = len(synth.PARENTS)
PO = 6 + 15 * J
rng = np.random.default_rng(7)
clip = synth.make_clip(T, seed=5, stats=stats)
keys = ("Y_root_vel", "Y_root_vrt", "Y_lpos", "Y_ltxy", "Y_lvel", "Y_lvrt")
row = np.concatenate([clip[k].reshape(T, -1) for k in keys], axis=1).astype(np.float32)          # [T, PO]
w_pose = torch.as_tensor(np.broadcast_to(row, (B, T, PO)).copy(), device=dev)
o_pose = (w_pose + torch.as_tensor(0.05 * rng.standard_normal((B, T, PO)), dtype=torch.float32, device=dev)).requires_grad_(True)
w_rpos = torch.as_tensor(np.broadcast_to(clip["Y_root_pos"], (B, T, 3)).copy(), dtype=torch.float32, device=dev)
w_rrot = torch.as_tensor(np.broadcast_to(clip["Y_root_rot"], (B, T, 4)).copy(), dtype=torch.float32, device=dev)
o_rpos = (w_rpos + 0.01 * torch.randn(B, T, 3, device=dev)).requires_grad_(True)
o_rrot = (w_rrot + 0.01 * torch.randn(B, T, 4, device=dev)).requires_grad_(True)
gaze = torch.as_tensor(np.broadcast_to(clip["Y_gaze_pos"], (B, T, 3)).copy(), dtype=torch.float32, device=dev)
mu = torch.randn(B, 64, device=dev, requires_grad=True)
lv = (0.3 * torch.randn(B, 64, device=dev)).requires_grad_(True)
parents = torch.as_tensor(synth.PARENTS, dtype=torch.int32, device=dev)
one = torch.ones((), device=dev)


def call():
    for t in (o_pose, o_rpos, o_rrot, mu, lv):
        t.grad = None
    loss, terms = ops.training_loss(o_pose, o_rpos, o_rrot, w_pose, w_rpos, w_rrot, gaze, parents, synth.DT, mu, lv,
                                    kl_weight=0.3, unit_grad=True)
    loss.backward(one)
    return loss, terms


for _ in range(5):
    loss, terms = call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 50
e0.record()
for _ in range(N):
    call()
e1.record()
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (loss, terms, o_pose.grad, o_rpos.grad, o_rrot.grad, mu.grad, lv.grad):
    h.update(t.detach().cpu().numpy().tobytes())
print(f"lib {os.environ.get('ZEGGS_LIB', 'default')}: B {B} T {T}: {1e3 * e0.elapsed_time(e1) / N:.1f} us per call "
      f"(host-launch inclusive), loss {float(loss):.6f}, sha256 {h.hexdigest()[:16]}")
