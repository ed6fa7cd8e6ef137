"""The engine's second / third stream (decoder weight-gradient GEMMs beside the encoders' backward, speech encoder beside
the style encoder): several optimizer steps give the weights of the single-stream schedule."""
import numpy as np
import pytest
import torch

import helpers
from zeggs import engine, synth

pytestmark = pytest.mark.gpu


def _run(overlap, steps=4, B=8, T=24, L=32, clip=None, edit_after=None, **kw):
    dev = torch.device("cuda:0")
    se, de, st = helpers.build_nets()
    se, de, st = se.to(dev).eval(), de.to(dev).eval(), st.to(dev).eval()        # eval: no dropout masks to agree on
    data = synth.make_processed(3, 0, clip or T + 40, seed=11)
    ds = engine.DeviceDataset(data, T, dev)
    eng = engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT, overlap_wgrads=overlap, **kw)
    assert (eng.wgrad_stream is not None) == overlap and (eng.aux_stream is not None) == overlap
    perm = np.random.default_rng(5).permutation(len(ds))
    gen = torch.Generator().manual_seed(3)
    losses = []
    for k in range(steps):
        eps = torch.randn(B, 64, generator=gen).to(dev)
        idx = engine.shard_indices(perm, k, B, 1, 0)
        losses.append(eng.step(idx, L, eps=eps))
        if edit_after is not None and edit_after[0] == k:     # somebody edits the weights between two steps
            with torch.no_grad():
                if edit_after[1] == "flat":
                    eng.flat_p.mul_(1.001)
                else:
                    for p in eng.params:
                        p.mul_(1.001)
        if overlap and k + 1 < steps:
            eng.prefetch(engine.shard_indices(perm, k + 1, B, 1, 0), L)      # picked up by the next step
            assert eng._prefetched is not None
    torch.cuda.synchronize()
    # with the side streams the decoder's slice of every optimizer step runs early, on the weight-gradient stream
    assert eng.opt.early_pieces == (steps if overlap else 0), eng.opt.early_pieces
    # ... and the two side queues are released from INSIDE the style encoder's forward, behind its first convolution (the speech
    # encoder's graph is still built: its weights above move exactly as in the single-stream schedule)
    assert eng.head_first_releases == (steps if overlap and eng.style_head_first else 0), eng.head_first_releases
    _run.last_engine = eng
    return eng.flat_p.detach().cpu().numpy().copy(), [float(x.detach()) for x in losses]


def test_side_streams_give_the_single_stream_weights():
    p1, l1 = _run(True)
    p0, l0 = _run(False)
    assert np.isfinite(p1).all() and np.isfinite(l1).all()
    assert np.abs(p1 - p0).max() <= 2e-6, np.abs(p1 - p0).max()      # split-K atomics: not bitwise run to run
    assert np.allclose(l1, l0, rtol=1e-5, atol=1e-6), (l1, l0)
    p2, _ = _run(True)                                            # and run to run
    assert np.abs(p1 - p2).max() <= 2e-6


def test_deferred_style_weight_gradients_give_the_same_weights():
    """defer_style_wgrads (off by default: measured slower): the style encoder's backward enqueues its chain only
    (zeggs_style_encoder_bwd_part, part 1) and the engine runs the six weight-gradient products (part 2) on the third queue behind
    it -- the operands are parked in buffers of their own -- and joins before the optimizer: the weights of the inline schedule."""
    kw = dict(steps=3, B=8, T=64, L=128, clip=400)
    p1, l1 = _run(True, defer_style_wgrads=True, **kw)
    assert _run.last_engine.defer_style_wgrads and not _run.last_engine.ctx.deferred_wgrads
    p0, l0 = _run(True, defer_style_wgrads=False, **kw)
    assert np.isfinite(p1).all() and np.isfinite(l1).all()
    assert np.abs(p1 - p0).max() <= 2e-6, np.abs(p1 - p0).max()
    assert np.allclose(l1, l0, rtol=1e-5, atol=1e-6), (l1, l0)


@pytest.mark.parametrize("how", ["flat", "param"])
def test_packs_made_ahead_are_not_used_after_the_weights_were_edited(how):
    """prepare_ahead: the next step's weight-only packs are made right behind the optimizer.  An in-place torch edit of the weights
    between two steps -- of the engine's flat buffer, or of the modules' parameters (views of it with version counters of their
    own: what load_state_dict does) -- must make the next step pack again: five steps with such an edit end in the weights of an
    engine that prepares nothing ahead (stale packs = the unscaled weights in the sweeps: off by 1e-3, not 5e-6)."""
    kw = dict(steps=5, B=32, T=256, L=384, clip=900, edit_after=(2, how))
    p1, l1 = _run(True, **kw)
    assert _run.last_engine.prepare_ahead and _run.last_engine.ctx.prepared_hits >= 3
    p0, l0 = _run(True, prepare_ahead=False, style_head_first=0, **kw)
    assert np.abs(p1 - p0).max() <= 5e-6, np.abs(p1 - p0).max()
    assert np.allclose(l1, l0, rtol=2e-5, atol=1e-6), (l1, l0)


def test_side_streams_at_the_bench_shape():
    """batch 32 x 256 frames, example 384 (both persistent sweeps, their packs prepared on the second stream, the next batch
    prefetched): five optimizer steps end in the weights of the single-stream schedule"""
    p1, l1 = _run(True, steps=5, B=32, T=256, L=384, clip=900)
    hits = _run.last_engine.ctx.prepared_hits          # (per engine: ops.EngineContext)
    # from the second step on (the first one validates the kernels) the forward picks up the workspace whose packs were made
    # on the second stream: a silent miss costs 0.4 ms per iteration (the packs run twice)
    assert hits >= 3, hits
    p0, l0 = _run(False, steps=5, B=32, T=256, L=384, clip=900)
    assert np.isfinite(p1).all() and np.isfinite(l1).all()
    assert np.abs(p1 - p0).max() <= 5e-6, np.abs(p1 - p0).max()
    assert np.allclose(l1, l0, rtol=2e-5, atol=1e-6), (l1, l0)


def test_two_engines_on_two_threads_are_isolated():
    """Two TrainEngines stepping CONCURRENTLY from two host threads (own nets, batch sizes, windows, side streams, status
    words, noise-seed streams; training mode, dropout on) end in exactly the weights each of them reaches when it runs
    alone: what a step needs beyond the arguments of its calls travels in the engine's ops.EngineContext, not in module
    globals of the binding (round 3 flipped `direct_param_grads` / `set_status` / `set_wgrad_stream` / hooks around step()).
    The persistent sweeps are off here: they need the whole chip, two tenants would make each other give up."""
    import threading
    from zeggs import ops
    dev = torch.device("cuda:0")
    cfg = [dict(seed=1234, B=4, T=12, L=16, noise=11, data=21), dict(seed=4321, B=6, T=20, L=24, noise=12, data=22)]
    steps = 6

    def make(c):
        se, de, st = helpers.build_nets(seed=c["seed"])
        se, de, st = se.to(dev).train(), de.to(dev).train(), st.to(dev).train()
        ds = engine.DeviceDataset(synth.make_processed(3, 0, c["T"] + 40, seed=c["data"]), c["T"], dev)
        return engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT, noise_seed=c["noise"])

    def drive(eng, c, out, barrier=None):
        try:
            stream = torch.cuda.Stream(device=dev)
            perm = np.random.default_rng(c["seed"]).permutation(len(eng.ds))
            with torch.cuda.stream(stream):
                for k in range(steps):
                    if barrier is not None:
                        barrier.wait()              # both threads are inside step() at the same time
                    eng.step(engine.shard_indices(perm, k, c["B"], 1, 0), c["L"])
                stream.synchronize()
                eng.flush()
            out.append(eng.flat_p.detach().cpu().numpy().copy())
        except BaseException as e:      # noqa: BLE001  (reported by the main thread)
            out.append(e)
            if barrier is not None:
                barrier.abort()

    saved = {k: ops._OPTIONS.get(k, 1) for k in ("train_persistent", "bwd_persistent")}
    for k in saved:
        ops.set_option(k, 0)
    try:
        alone = []
        for c in cfg:
            out = []
            drive(make(c), c, out)
            assert not isinstance(out[0], BaseException), out[0]
            alone.append(out[0])
        engines = [make(c) for c in cfg]
        outs, barrier = [[], []], threading.Barrier(2)
        threads = [threading.Thread(target=drive, args=(e, c, o, barrier)) for e, c, o in zip(engines, cfg, outs)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        for k, v in saved.items():
            ops.set_option(k, v)
    for i, (o, a) in enumerate(zip(outs, alone)):
        assert o and not isinstance(o[0], BaseException), o
        assert np.isfinite(o[0]).all()
        assert np.abs(o[0] - a).max() <= 2e-6, (i, np.abs(o[0] - a).max())      # split-K atomics: not bitwise run to run
    assert engines[0].ctx is not engines[1].ctx and ops.current() is not engines[0].ctx
