"""A persistent kernel that gives up AFTER its validated first use must be harmless:
the give-up is forced with the tuning option "persistent_spin" = 0 (the first unsatisfied device-side wait runs out at once --
what a co-tenant that holds CUs does to the bounded waits), and
  * training: the fused RAdam step of that iteration is a no-op ON THE DEVICE (zeggs_radam_step_guarded reads the sticky
    status word the kernels OR into), the host notices STATUS_LAG iterations later, disables the persistent sweeps and
    re-runs the lost steps on the stage kernels -- the weights end up those of a run on the stage kernels throughout;
  * inference: the B = 1 rollout is redone on the stage launches before anybody reads its frames;
  * two decoders with their own workspaces, streams and ZeggsDecCall structs interleave in one process (no global per-call
    state in the ABI)."""
import warnings

import numpy as np
import pytest
import torch

import helpers
from zeggs import engine, ops, synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SPIN = 1 << 21


@pytest.fixture
def restore_options():
    yield
    for k, v in (("persistent_spin", SPIN), ("train_persistent", 1), ("bwd_persistent", 1), ("persistent", 1)):
        ops.set_option(k, v)


def _engine(B, T, clip):
    se, de, st = helpers.build_nets()
    se, de, st = se.to(DEV).train(), de.to(DEV).train(), st.to(DEV).train()      # dropout ON: the replay re-draws the same masks
    data = synth.make_processed(3, 0, clip, seed=11)
    ds = engine.DeviceDataset(data, T, DEV)
    return engine.TrainEngine(se, de, st, ds, synth.PARENTS, synth.DT), ds


def _train(steps, fail_at, persistent, B=32, T=64, L=96):
    for k in ("train_persistent", "bwd_persistent"):
        ops.set_option(k, int(persistent))
    ops.set_option("persistent_spin", SPIN)
    ops.manual_seed(99)
    eng, ds = _engine(B, T, T + 200)
    perm = np.random.default_rng(5).permutation(len(ds))
    for k in range(steps):
        if k == fail_at:
            torch.cuda.synchronize()
            assert ops.lib().zeggs_persistent_state(1) == 1 and ops.lib().zeggs_persistent_state(2) == 1   # validated by now
            ops.set_option("persistent_spin", 0)
        eng.step(engine.shard_indices(perm, k, B, 1, 0), L)
        if k == fail_at:
            torch.cuda.synchronize()
            st = eng.status.cpu().numpy()
            assert st[0] & 2 and st[1] == 1, st          # the rollout gave up, the step was skipped on the device
            ops.set_option("persistent_spin", SPIN)
    torch.cuda.synchronize()
    return eng


def test_training_giveup_is_skipped_on_device_and_replayed_on_the_stage_kernels(restore_options):
    steps, fail_at = 7, 2
    ref = _train(steps, -1, persistent=False)
    w_ref = ref.flat_p.detach().cpu().numpy().copy()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        eng = _train(steps, fail_at, persistent=True)
    assert any("gave up" in str(w.message) for w in rec)
    assert eng.recovered_steps == engine.TrainEngine.STATUS_LAG and eng.iteration == steps == eng.opt._step
    assert int(eng.status.cpu()[0]) == 0 and int(eng.status.cpu()[1]) == 0
    w = eng.flat_p.detach().cpu().numpy()
    assert np.isfinite(w).all()
    # steps 0..1 ran on the persistent kernels (1e-6 apart from the stage kernels), everything after on the stage kernels
    assert np.abs(w - w_ref).max() <= 5e-6, np.abs(w - w_ref).max()


def test_persistent_sweeps_are_rearmed_after_a_giveup_and_back_off(restore_options):
    """a give-up must not switch the persistent sweeps off for the life of the process.  Give-up at
    iteration 2 (persistent_spin = 0) -> skipped on the device, noticed STATUS_LAG later, replayed on the stage kernels; after
    `rearm_after` clean iterations (counted from the first lost step, so every data-parallel rank gets the same number) the
    sweeps are back ON -- proven by forcing a second give-up, which only a running persistent kernel can produce -- and that
    second give-up, coming right after the re-arm, doubles the probation.  The weights end up those of an undisturbed run."""
    steps, B, T, L = 14, 32, 64, 96
    ref = _train(steps, -1, persistent=False)
    w_ref = ref.flat_p.detach().cpu().numpy().copy()
    for k in ("train_persistent", "bwd_persistent"):
        ops.set_option(k, 1)
    ops.set_option("persistent_spin", SPIN)
    ops.manual_seed(99)
    eng, ds = _engine(B, T, T + 200)
    eng.rearm_after = eng._rearm_wait = 4
    perm = np.random.default_rng(5).permutation(len(ds))
    lag = engine.TrainEngine.STATUS_LAG
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for k in range(steps):
            if k in (2, 8):
                torch.cuda.synchronize()
                ops.set_option("persistent_spin", 0)
            eng.step(engine.shard_indices(perm, k, B, 1, 0), L)
            if k in (2, 8):
                torch.cuda.synchronize()
                st = eng.status.cpu().numpy()
                assert st[0] & 2 and st[1] == 1, (k, st)      # a persistent rollout ran and gave up: the step was skipped on the device
                ops.set_option("persistent_spin", SPIN)
            if k == 2 + lag:        # the host has just noticed: sweeps off, probation of 4 iterations from the first lost step
                assert ops._OPTIONS["train_persistent"] == 0 and eng._rearm_at == 2 + 4 and eng.rearm_count == 0
            if k == 6:              # ... over: back on (and validated as before)
                assert ops._OPTIONS["train_persistent"] == 1 and ops._OPTIONS["bwd_persistent"] == 1 and eng.rearm_count == 1
                assert ops.lib().zeggs_persistent_state(1) == 1 and ops.lib().zeggs_persistent_state(2) == 1
            if k == 8 + lag:        # second give-up within the probation window of the re-arm: the wait doubles
                assert ops._OPTIONS["train_persistent"] == 0 and eng._rearm_wait == 8 and eng._rearm_at == 8 + 8
        eng.flush()
    torch.cuda.synchronize()
    assert sum("gave up" in str(w.message) for w in rec) == 2
    assert eng.recovered_steps == 2 * lag and eng.iteration == steps == eng.opt._step and eng.rearm_count == 1
    w = eng.flat_p.detach().cpu().numpy()
    assert np.isfinite(w).all()
    # (four of the 14 steps ran on the persistent kernels, 1e-6 apart per gradient; an early RAdam step moves an entry by up to lr = 1e-4)
    assert np.abs(w - w_ref).max() <= 3e-5, np.abs(w - w_ref).max()


def _cotenant(workgroups, ms, stream, scratch):
    import ctypes as C
    ops._check(ops.lib().zeggs_test_cotenant(int(workgroups), 256, C.c_float(ms), C.c_void_p(scratch.data_ptr()),
                                             C.c_long(scratch.numel()), C.c_void_p(stream.cuda_stream)), "test_cotenant")


def test_giveup_under_a_resident_cotenant_is_replayed_and_rearmed(restore_options):
    """the give-up path driven by a RESIDENT co-tenant, not by the persistent_spin = 0 hook.  64 stand-in
    workgroups (zeggs_test_cotenant: what a collective's resident workgroups are to the chip) sit on 64 CUs for 60 ms across the
    sweep boundary of iteration 3: the forward sweep needs all 256 CUs to itself, 64 of its workgroups cannot be placed, the
    bounded waits of the 192 that can run out (persistent_spin lowered to ~4 ms of polls for the WHOLE run: nothing is forced) --
    the step is skipped on the device, noticed STATUS_LAG iterations later, replayed on the stage kernels, and after the probation
    the sweeps are back on (validated again) for the rest of the run.  Weights = those of an undisturbed run."""
    steps, B, T, L, at = 12, 32, 64, 96, 3
    spin = 4096
    ref = _train(steps, -1, persistent=False)
    w_ref = ref.flat_p.detach().cpu().numpy().copy()
    for k in ("train_persistent", "bwd_persistent"):
        ops.set_option(k, 1)
    ops.set_option("persistent_spin", spin)
    ops.manual_seed(99)
    eng, ds = _engine(B, T, T + 200)
    eng.rearm_after = eng._rearm_wait = 4
    perm = np.random.default_rng(5).permutation(len(ds))
    lag = engine.TrainEngine.STATUS_LAG
    side = torch.cuda.Stream()
    scratch = torch.zeros(1 << 16, device=DEV)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        for k in range(steps):
            if k == at:
                torch.cuda.synchronize()
                assert ops.lib().zeggs_persistent_state(1) == 1 and ops.lib().zeggs_persistent_state(2) == 1   # validated by now
                _cotenant(64, 60.0, side, scratch)          # resident from here, across this iteration's sweeps
            eng.step(engine.shard_indices(perm, k, B, 1, 0), L)
            if k == at:
                torch.cuda.synchronize()
                st = eng.status.cpu().numpy()
                assert st[0] & (2 | 4) and st[1] == 1, st          # a sweep gave up under the co-tenant: the step was skipped on the device
            if k == at + lag:
                assert ops._OPTIONS["train_persistent"] == 0 and eng._rearm_at == at + 4 and eng.rearm_count == 0
            if k == at + 4:
                assert ops._OPTIONS["train_persistent"] == 1 and ops._OPTIONS["bwd_persistent"] == 1 and eng.rearm_count == 1
        eng.flush()
    torch.cuda.synchronize()
    assert ops.lib().zeggs_persistent_state(1) == 1 and ops.lib().zeggs_persistent_state(2) == 1       # back on and validated
    assert sum("gave up" in str(w.message) for w in rec) == 1
    assert eng.recovered_steps == lag and eng.iteration == steps == eng.opt._step and eng.rearm_count == 1
    assert int(eng.status.cpu()[0]) == 0 and int(eng.status.cpu()[1]) == 0
    w = eng.flat_p.detach().cpu().numpy()
    assert np.isfinite(w).all()
    assert np.abs(w - w_ref).max() <= 3e-5, np.abs(w - w_ref).max()


def _rollout(de, T, seed=4242):
    W, speech, style = helpers.full_decoder_inputs(helpers.real_stats("v1"), 1, T, seed)
    s = helpers.real_stats_tensors("v1", device=DEV)
    g = lambda t: t.to(DEV)  # noqa: E731
    fp = [g(W[k][:, 0].contiguous()) for k in ("Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos", "Y_ltxy",
                                               "Y_lvel", "Y_lvrt")]
    with torch.no_grad():
        return [o.cpu() for o in de(*fp, g(W["Y_gaze_pos"]), g(speech), g(style), None, s["in_mean"], s["in_std"],
                                    s["out_mean"], s["out_std"], synth.DT)]


def test_inference_giveup_redoes_the_rollout_on_the_stage_launches(restore_options):
    _, de, _ = helpers.build_nets()
    de = de.to(DEV).eval()
    ops.set_option("persistent", 0)
    stage = _rollout(de, 40)
    ops.set_option("persistent", 1)
    first = _rollout(de, 40)                                  # validates the kernel on this process
    assert ops.lib().zeggs_persistent_state(0) == 1
    assert max(float((a - b).abs().max()) for a, b in zip(first, stage)) < 1e-4
    ops.set_option("persistent_spin", 0)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        redone = _rollout(de, 40)
    assert any("gave up" in str(w.message) for w in rec)
    for a, b in zip(redone, stage):      # the stage launches' frames (run to run they differ by the prologue GEMMs' split-K atomics)
        assert torch.isfinite(a).all() and float((a - b).abs().max()) < 2e-6


def test_two_decoders_interleave_in_one_process(restore_options):
    """Per-call state travels in ZeggsDecCall: two training decoders with different shapes, their own workspaces, prepared
    on their own side streams, forward / backward calls interleaved on two streams -> each equals its isolated run."""
    def make(B, T, seed):
        _, de, _ = helpers.build_nets(seed=seed)
        de = de.to(DEV).train()
        W, speech, style = helpers.full_decoder_inputs(helpers.real_stats("v1"), B, T, 700 + seed)
        g = lambda t: t.to(DEV)  # noqa: E731
        fp = [g(W[k][:, 0].contiguous()) for k in ("Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos",
                                                   "Y_ltxy", "Y_lvel", "Y_lvrt")]
        return de, fp, g(W["Y_gaze_pos"]), g(speech).requires_grad_(True), g(style)

    s = helpers.real_stats_tensors("v1", device=DEV)

    def fwd(m):
        de, fp, gaze, speech, style = m
        return de(*fp, gaze, speech, style, None, s["in_mean"], s["in_std"], s["out_mean"], s["out_std"], synth.DT)

    def loss(O):
        return sum((o * o).mean() for o in O)

    def grads(m):
        return [p.grad.detach().clone() for p in m[0].parameters()] + [m[3].grad.detach().clone()]

    def clear(m):
        m[0].zero_grad()
        m[3].grad = None

    A, Bm = make(32, 24, 1), make(8, 40, 2)
    for m in (A, Bm):                      # warm-up: validates the persistent kernels, fills the caches
        loss(fwd(m)).backward()
        clear(m)
    iso = []
    for m in (A, Bm):
        loss(fwd(m)).backward()
        torch.cuda.synchronize()
        iso.append(grads(m))
        clear(m)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    sa.wait_stream(torch.cuda.current_stream())
    sb.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sa):
        la = loss(fwd(A))
    with torch.cuda.stream(sb):
        lb = loss(fwd(Bm))
    with torch.cuda.stream(sa):
        la.backward()
    with torch.cuda.stream(sb):
        lb.backward()
    torch.cuda.synchronize()
    for m, ref in zip((A, Bm), iso):
        for got, want in zip(grads(m), ref):
            scale = max(1e-12, float(want.abs().max()))
            assert float((got - want).abs().max()) <= 2e-5 * scale


def test_flush_replays_a_giveup_in_the_last_iterations(restore_options):
    """a sweep that gives up within the last STATUS_LAG iterations (or right before a checkpoint) is only known to
    the device -- TrainEngine.flush() drains it: the skipped step is re-run on the stage kernels, the optimizer's step count
    and the weights are those of an undisturbed run on the stage kernels, and the replayed loss is handed to the caller."""
    steps = 5
    ref = _train(steps, -1, persistent=False)
    w_ref = ref.flat_p.detach().cpu().numpy().copy()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        eng = _train(steps, steps - 1, persistent=True)              # the LAST step gives up: no later step() would notice
        assert eng.recovered_steps == 0 and int(eng.status.cpu()[1]) == 1
        assert eng.flush() == 1
    assert any("gave up" in str(w.message) for w in rec)
    assert eng.recovered_steps == 1 and eng.iteration == steps == eng.opt._step
    assert [it for it, _, _ in eng.replayed] == [steps - 1] and np.isfinite(float(eng.replayed[0][1]))
    assert int(eng.status.cpu()[0]) == 0 and int(eng.status.cpu()[1]) == 0 and eng.flush() == 0
    w = eng.flat_p.detach().cpu().numpy()
    assert np.isfinite(w).all() and np.abs(w - w_ref).max() <= 5e-6, np.abs(w - w_ref).max()


def test_plain_autograd_training_giveup_is_redone(restore_options):
    """Training-mode decoder calls OUTSIDE an engine (plain autograd: the tests, the reference's own loop of INTEGRATION.md
    route 2) have nobody who looks at a status word later: the binding inspects its own word right after the rollout and
    after the BPTT sweep and redoes them on the stage kernels, so neither outputs nor gradients of a give-up reach the caller."""
    _, de, _ = helpers.build_nets()
    de = de.to(DEV).train()
    B, T = 32, 40
    W, speech, style = helpers.full_decoder_inputs(helpers.real_stats("v1"), B, T, 911)
    s = helpers.real_stats_tensors("v1", device=DEV)
    g = lambda t: t.to(DEV)  # noqa: E731
    fp = [g(W[k][:, 0].contiguous()) for k in ("Y_root_pos", "Y_root_rot", "Y_root_vel", "Y_root_vrt", "Y_lpos", "Y_ltxy",
                                               "Y_lvel", "Y_lvrt")]

    def run():
        de.zero_grad()
        sp = g(speech).requires_grad_(True)
        O = de(*fp, g(W["Y_gaze_pos"]), sp, g(style), None, s["in_mean"], s["in_std"], s["out_mean"], s["out_std"], synth.DT)
        sum((o * o).mean() for o in O).backward()
        torch.cuda.synchronize()
        return [o.detach().clone() for o in O], [p.grad.detach().clone() for p in de.parameters()] + [sp.grad.clone()]

    for k in ("train_persistent", "bwd_persistent"):
        ops.set_option(k, 0)
    out_s, grad_s = run()                                       # stage kernels
    for k in ("train_persistent", "bwd_persistent"):
        ops.set_option(k, 1)
    run()                                                       # validates both sweeps on this process
    assert ops.lib().zeggs_persistent_state(1) == 1 and ops.lib().zeggs_persistent_state(2) == 1
    ops.set_option("persistent_spin", 0)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        out_g, grad_g = run()
    assert any("gave up" in str(w.message) for w in rec)
    for a, b in zip(out_g, out_s):     # (run to run the stage path differs by the split-K atomics of its prologue GEMMs)
        assert torch.isfinite(a).all() and float((a - b).abs().max()) < 1e-6 * max(1.0, float(b.abs().max()))
    for a, b in zip(grad_g, grad_s):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 2e-5 * max(1e-12, float(b.abs().max()))


def test_streaming_giveup_redoes_the_chunk(restore_options):
    """GestureStream feeds a chunk's last frame and GRU state into the next chunk -- a persistent decode kernel that
    gives up mid-stream must not be carried forward: the stream owns a status word, checks it per chunk and redoes the
    chunk on the stage launches; the streamed frames equal an undisturbed stream's."""
    from zeggs import anim, stream
    se, de, _ = helpers.build_nets()
    se, de = se.to(DEV).eval(), de.to(DEV).eval()
    stats = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32, device=DEV) for k, v in synth.make_stats().items()}
    conf = dict(pre_emphasis=False, pre_emph_coeff=0.97, centered=True, real_amplitude=True, normalize_mel_bins=True,
                normalize_range=True, min_clipping=1e-5, sampling_rate=16000, mel_fmin=20, mel_fmax=7600,
                n_mel_channels=80, filter_length=800, hop_length=200, resample_method="linear", normalize_loudness=False)
    wav = synth.synth_wav(40000, seed=5).astype(np.float32) / 32768.0
    first = anim.preprocess_animation(synth.make_bvh_clip(8, seed=3), DEV)
    torch.manual_seed(5)
    style = torch.randn(1, 64, device=DEV) * 0.5
    cuts = [0, 9000, 21000, 30000, 40000]

    def run(fail_chunk):
        gs = stream.GestureStream(se, de, first, style, stats, conf, synth.DT)
        outs = []
        for i in range(len(cuts) - 1):
            if i == fail_chunk:
                assert ops.lib().zeggs_persistent_state(0) == 1
                ops.set_option("persistent_spin", 0)
            outs.append(gs.push(wav[cuts[i]:cuts[i + 1]]))
            ops.set_option("persistent_spin", SPIN)
        outs.append(gs.finish())
        return gs, {k: torch.cat([o[k] for o in outs if o], dim=0) for k in ("pose", "rpos", "rrot")}

    ops.set_option("persistent", 1)
    _, good = run(-1)
    if ops.lib().zeggs_persistent_state(0) != 1:
        pytest.skip("the B = 1 persistent decode kernel is not available on this device")
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        gs, got = run(2)
    assert any("gave up" in str(w.message) for w in rec) and gs.redone_chunks == 1
    for k in good:
        assert got[k].shape == good[k].shape and torch.isfinite(got[k]).all()
        assert float((got[k] - good[k]).abs().max()) < 5e-5, k
