"""The fused attention kernels of the style encoder (csrc/attention.hip: softmax(QK^T) -> dropout -> .V in one launch, backward
recomputed from the row log-sum-exp) against the GEMM + softmax + GEMM path they replace, through the whole StyleEncoder
forward + backward: sequence lengths that are / are not multiples of the 32-key and 128-query tiles, dropout masks on (the
same counter-hash masks on both paths) and off.  Parity with the reference itself: tests/test_gpu_parity.py and
tests/test_gpu_full_shapes.py run on the fused path (the default)."""
import numpy as np
import pytest
import torch

import helpers
from zeggs import ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(st, x, eps, w, fused, dropout):
    ops.set_option("fused_attention", fused)
    ops.manual_seed(77)
    st.train() if dropout else st.eval()
    st.zero_grad()
    z, mu, lv = st(x, 1.0, eps=eps)
    (z * w[0] + mu * w[1] + lv * w[2]).sum().backward()
    torch.cuda.synchronize()
    return [t.detach().clone() for t in (z, mu, lv)], {k: p.grad.detach().clone() for k, p in st.named_parameters()}


@pytest.mark.parametrize("one_launch", [1, 0])
@pytest.mark.parametrize("L,dropout", [(384, True), (77, True), (200, False), (33, False), (512, True)])
def test_fused_attention_matches_the_gemm_softmax_path(L, dropout, one_launch):
    """one_launch = 1 (default since round 6): the dQ and the dK / dV pass as ONE grid (blockIdx.z = pass; the dK / dV workgroups
    form rowsum(dO . O) themselves); 0: two launches, the second reading what the first stored."""
    ops.set_option("attn_bwd_one_launch", one_launch)
    _, _, st = helpers.build_nets()
    st = st.to(DEV)
    g = torch.Generator().manual_seed(L)
    B = 3
    x = torch.randn(B, L, synth.POSE_IN, generator=g).to(DEV)
    eps = torch.randn(B, 64, generator=g).to(DEV)
    w = [torch.randn(B, 64, generator=g).to(DEV) for _ in range(3)]
    try:
        out1, g1 = _run(st, x, eps, w, 1, dropout)
        out0, g0 = _run(st, x, eps, w, 0, dropout)
    finally:
        ops.set_option("fused_attention", 1)
        ops.set_option("attn_bwd_one_launch", 1)
    for a, b in zip(out1, out0):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))
    worst = 0.0
    for k in g0:
        scale = max(1e-12, float(g0[k].abs().max()))
        e = float((g1[k] - g0[k]).abs().max()) / scale
        worst = max(worst, e)
        assert e < 2e-4, (k, e)
    print(f"\nL={L} dropout={dropout}: fused vs unfused worst gradient difference {worst:.2e} of the tensor's max")
