"""The engine's second / third stream (decoder weight-gradient GEMMs beside the encoders' backward, speech encoder beside
the style encoder): several optimizer steps give the weights of the single-stream schedule."""
import numpy as np
import pytest
import torch

import helpers
from zeggs import engine, synth

pytestmark = pytest.mark.gpu


def _run(overlap, steps=4, B=8, T=24, L=32, clip=None):
    dev = torch.device("cuda:0")
    se, de, st = helpers.build_nets()
    se, de, st = se.to(dev).eval(), de.to(dev).eval(), st.to(dev).eval()        # eval: no dropout masks to agree on
    data = synth.make_processed(3, 0, clip or T + 40, seed=11)
    ds = engine.DeviceDataset(data, T, dev)
    eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT, overlap_wgrads=overlap)
    assert (eng.wgrad_stream is not None) == overlap and (eng.aux_stream is not None) == overlap
    perm = np.random.default_rng(5).permutation(len(ds))
    gen = torch.Generator().manual_seed(3)
    losses = []
    for k in range(steps):
        eps = torch.randn(B, 64, generator=gen).to(dev)
        idx = engine.shard_indices(perm, k, B, 1, 0)
        losses.append(eng.step(idx, L, eps=eps))
        if overlap and k + 1 < steps:
            eng.prefetch(engine.shard_indices(perm, k + 1, B, 1, 0), L)      # picked up by the next step
            assert eng._prefetched is not None
    torch.cuda.synchronize()
    # with the side streams the decoder's slice of every optimizer step runs early, on the weight-gradient stream
    assert eng.opt.early_pieces == (steps if overlap else 0), eng.opt.early_pieces
    return eng.flat_p.detach().cpu().numpy().copy(), [float(x.detach()) for x in losses]


def test_side_streams_give_the_single_stream_weights():
    p1, l1 = _run(True)
    p0, l0 = _run(False)
    assert np.isfinite(p1).all() and np.isfinite(l1).all()
    assert np.abs(p1 - p0).max() <= 2e-6, np.abs(p1 - p0).max()      # split-K atomics: not bitwise run to run
    assert np.allclose(l1, l0, rtol=1e-5, atol=1e-6), (l1, l0)
    p2, _ = _run(True)                                            # and run to run
    assert np.abs(p1 - p2).max() <= 2e-6


def test_side_streams_at_the_bench_shape():
    """batch 32 x 256 frames, example 384 (both persistent sweeps, their packs prepared on the second stream, the next batch
    prefetched): five optimizer steps end in the weights of the single-stream schedule"""
    from zeggs import ops
    hits = ops.prepared_hits
    p1, l1 = _run(True, steps=5, B=32, T=256, L=384, clip=900)
    # from the second step on (the first one validates the kernels) the forward picks up the workspace whose packs were made
    # on the second stream: a silent miss costs 0.4 ms per iteration (the packs run twice)
    assert ops.prepared_hits - hits >= 3, ops.prepared_hits - hits
    p0, l0 = _run(False, steps=5, B=32, T=256, L=384, clip=900)
    assert np.isfinite(p1).all() and np.isfinite(l1).all()
    assert np.abs(p1 - p0).max() <= 5e-6, np.abs(p1 - p0).max()
    assert np.allclose(l1, l0, rtol=2e-5, atol=1e-6), (l1, l0)
