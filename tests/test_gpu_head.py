"""Round 6: the head of an iteration.  The style example gathered straight into the padded input of the attention encoder's first
convolution (zeggs_gather_example + zeggs_style_encoder_input_offset + fwd_part | 4), and the prefetched batch in two alternating
buffer sets that are never handed back to the allocator (no record_stream events on the caller's stream)."""
import numpy as np
import pytest
import torch

import helpers
from zeggs import engine, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,L,W,extra,pad", [(3, 7, 5, 3, 1), (32, 384, synth.POSE_OUT, 3, 1), (2, 1, 130, 0, 2), (1, 64, 64, 1, 0)])
def test_gather_example_is_bit_identical_to_fill_gather_normalise_pad(B, L, W, extra, pad):
    """reference dataset.py:176-204 (get_example: the gaze slot stays zero) + train.py:239 (normalisation), then the zero rows the
    style encoder's first convolution pads with (modules.py ConvNorm padding = 1): the one-pass kernel against the separate ones."""
    g = torch.Generator().manual_seed(B * 1000 + L)
    frames = torch.randn(500, W, generator=g).to(DEV)
    rows = torch.randint(0, 500, (B, L), generator=g).to(DEV)
    mean = torch.randn(W + extra, generator=g).to(DEV)
    std = (torch.rand(W + extra, generator=g) + 0.5).to(DEV)
    ex = ops.fill_(torch.empty(B, L, W + extra, device=DEV))
    ops.gather_rows(frames, rows, out=ex, out_ld=W + extra)
    ops.normalize_rows_(ex, mean, std)
    want = torch.zeros(B, L + 2 * pad, W + extra, device=DEV)
    want[:, pad:pad + L] = ex
    out = torch.full((B, L + 2 * pad, W + extra), 7.0, device=DEV)
    ops.gather_example(frames, rows, mean, std, out, pad=pad)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    # ... and the host formula
    f, r, m, s = frames.cpu().numpy(), rows.cpu().numpy(), mean.cpu().numpy(), std.cpu().numpy()
    host = (np.concatenate([f[r], np.zeros((B, L, extra), np.float32)], axis=2) - m) / s
    assert np.array_equal(out[:, pad:pad + L].cpu().numpy(), host.astype(np.float32))


def _style(B=4, L=48, train=False):
    torch.manual_seed(7)
    _, _, st = helpers.build_nets()
    st = st.to(DEV)
    st.train(train)
    return st, torch.randn(B, L, synth.POSE_IN, device=DEV)


def test_style_encoder_runs_in_the_workspace_that_holds_its_padded_input():
    """ops.style_input_buffer + example_view: the forward recognises the tensor, runs in that workspace and skips its padding copy;
    outputs and every gradient equal the ordinary call's (split-K atomics: 1e-6, not bitwise); a COPY of the tensor, or a tensor of
    another shape, takes the ordinary path."""
    st, x = _style()
    enc = st.encoder
    eps = torch.randn(x.shape[0], 64, device=DEV)

    def run(inp):
        for p in st.parameters():
            p.grad = None
        z, mu, lv = st(inp, 1.0, eps=eps)
        (z.sum() + (mu * mu).sum() + lv.sum()).backward()
        torch.cuda.synchronize()
        return [t.detach().clone() for t in (z, mu, lv)] + [p.grad.detach().clone() for p in st.parameters()]

    ref = run(x)
    ws, xp = ops.style_input_buffer(enc, x.shape[0], x.shape[1], x.shape[2], st.training, DEV)
    xp.zero_()
    xp[:, 1:-1] = x
    n0 = ops.COUNTERS.get("style_in_place", 0)
    got = run(ops.example_view(ws, xp))
    assert ops.COUNTERS.get("style_in_place", 0) == n0 + 1
    for a, b in zip(ref, got):
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-6 * float(a.abs().max()) + 1e-9), float((a - b).abs().max())
    # the view's values are the example's: a plain copy of it is an ordinary input
    n1 = ops.COUNTERS.get("style_in_place", 0)
    again = run(ops.example_view(ws, xp).clone())
    assert ops.COUNTERS.get("style_in_place", 0) == n1
    for a, b in zip(ref, again):
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-6 * float(a.abs().max()) + 1e-9)
    # a workspace made for another length does not fit this input: ordinary path, same numbers
    ws2, xp2 = ops.style_input_buffer(enc, x.shape[0], x.shape[1] + 8, x.shape[2], st.training, DEV)
    fake = xp2[:, 1:1 + x.shape[1]]
    fake.copy_(x)
    fake._zeggs_style_ws = ws2
    other = run(fake)
    assert ops.COUNTERS.get("style_in_place", 0) == n1
    for a, b in zip(ref, other):
        assert torch.allclose(a, b, rtol=2e-5, atol=1e-6 * float(a.abs().max()) + 1e-9)


def _train(prefetch, steps=5, B=8, T=24, L=32, twice=False, in_place=True):
    se, de, st = (m.to(DEV).eval() for m in helpers.build_nets())
    ds = engine.DeviceDataset(synth.make_processed(3, 0, T + 40, seed=11), T, torch.device(DEV))
    eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT)
    eng.example_in_place = in_place
    perm = np.random.default_rng(5).permutation(len(ds))
    gen = torch.Generator().manual_seed(3)
    n0 = ops.COUNTERS.get("style_in_place", 0)
    for k in range(steps):
        eng.step(engine.shard_indices(perm, k, B, 1, 0), L, eps=torch.randn(B, 64, generator=gen).to(DEV))
        if prefetch and k + 1 < steps:
            if twice:       # a prefetch nobody picks up, then the right one: the same buffer set is written again two calls later
                eng.prefetch(engine.shard_indices(perm, k + 7, B, 1, 0), L)
                eng.prefetch(engine.shard_indices(perm, k + 8, B, 1, 0), L)
            eng.prefetch(engine.shard_indices(perm, k + 1, B, 1, 0), L)
    torch.cuda.synchronize()
    return eng, eng.flat_p.detach().cpu().numpy().copy(), ops.COUNTERS.get("style_in_place", 0) - n0


def test_prefetched_batches_in_persistent_buffers_give_the_weights_of_fresh_gathers():
    """Five optimizer steps: batches prefetched into the two alternating buffer sets (example in place) against batches gathered
    at the start of every step into fresh tensors (the ordinary style-encoder input path)."""
    e1, p1, n1 = _train(True)
    assert e1.prefetch_hits == 4 and n1 == 4
    bufs = [dict(s) for s in e1._pf_sets]
    assert all("style_ws" in s and "loss_ws" in s for s in bufs)
    e0, p0, n0 = _train(False)
    assert e0.prefetch_hits == 0 and n0 == 0
    assert np.isfinite(p1).all() and np.abs(p1 - p0).max() <= 2e-6, np.abs(p1 - p0).max()
    # the sets are reused, not reallocated: one more prefetch / step pair writes the same storage
    ptr = {k: v.data_ptr() for k, v in bufs[(e1._pf_n + 1) % 2].items() if hasattr(v, "data_ptr")}
    e1.prefetch(np.arange(8), 32)
    now = e1._pf_sets[e1._pf_n % 2]
    assert {k: v.data_ptr() for k, v in now.items() if hasattr(v, "data_ptr")} == ptr
    # ZEGGS_EXAMPLE_IN_PLACE=0 (the A/B switch): prefetched, ordinary input path, same weights
    e2, p2, n2 = _train(True, in_place=False)
    assert e2.prefetch_hits == 4 and n2 == 0 and np.abs(p2 - p0).max() <= 2e-6


def test_prefetches_that_nobody_picks_up_do_not_disturb_the_batch_in_use():
    """Three prefetches between two steps: the buffer set of the batch the previous step may still be reading is written again
    only behind an explicit wait of the third stream for the caller's."""
    e1, p1, _ = _train(True, twice=True)
    assert e1.prefetch_hits == 4
    _, p0, _ = _train(False)
    assert np.abs(p1 - p0).max() <= 2e-6, np.abs(p1 - p0).max()
