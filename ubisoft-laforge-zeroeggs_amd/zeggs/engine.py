"""Training engine: HBM-resident dataset, flat parameter buffers, one fused training step.

This is the host-side runtime that `zeggs.train.train()` and `bench.py` drive.  Per iteration:
  window/example index rule on the host (integers, reference dataset.py:79-96,176-204) ->
  HIP gathers from the device-resident frame tables -> SpeechEncoder -> StyleEncoder (VAE) ->
  Decoder rollout -> loss (fwd+bwd fused) -> BPTT -> [RCCL all-reduce of the flat gradient] ->
  fused RAdam over the flat parameter buffer.
Data parallelism: one process per GPU, every rank holds the full nets + dataset and takes its
contiguous slice of the global batch; gradients are averaged with ONE all-reduce per iteration.
"""
import collections
import copy
import os
import warnings

import numpy as np
import torch

from . import ops
from .modules import kl_div_weight
from .optimizers import RAdam


class _IndexUploader:
    """Host -> device copies of the (small) index arrays of a batch through a ring of PINNED staging buffers: a copy from
    pageable memory blocks the host until everything queued before it on the stream has finished, i.e. once per step the
    host would fall in line with the GPU and then feed the ~100 small launches in front of the forward rollout one at a
    time.  A slot is reused only after the copy out of it has completed (event) -- which also bounds how far the host runs
    ahead: two batches with 4 slots.  That is deliberate: while the host enqueues further ahead than that (after a device
    synchronisation it does, unthrottled) the iterations in flight measured 0.9 ms slower each (8 slots: 4 slow iterations
    after every synchronisation, 16 slots: 7, 4 slots: 1)."""

    SLOTS = 4

    def __init__(self, device):
        self.device = torch.device(device)
        self.slots = [None] * self.SLOTS          # (pinned int64 buffer, event of its last copy)
        self.k = 0

    def __call__(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.int64)
        if self.device.type != "cuda":
            return torch.as_tensor(arr).to(self.device)
        i, self.k = self.k % self.SLOTS, self.k + 1
        buf, ev = self.slots[i] if self.slots[i] is not None else (None, None)
        if ev is not None:
            ev.synchronize()
        if buf is None or buf.numel() < arr.size:
            buf = torch.empty(max(arr.size, 1024), dtype=torch.int64).pin_memory()
        host = buf[:arr.size].view(arr.shape)
        host.copy_(torch.from_numpy(arr))
        out = host.to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.slots[i] = (buf, ev)
        return out


class DeviceDataset:
    """The arrays of processed_data.npz resident in HBM (reference dataset.py:41-96 keeps them on the host)."""

    def __init__(self, data, window, device):
        f = lambda k: torch.as_tensor(np.asarray(data[k]), dtype=torch.float32)  # noqa: E731
        n = len(data["Y_root_pos"])
        self.n_frames = n
        self.window = window
        self.device = device
        self._upload = _IndexUploader(device)
        pose = torch.cat([f("Y_root_vel").reshape(n, -1), f("Y_root_vrt").reshape(n, -1), f("Y_lpos").reshape(n, -1),
                          f("Y_ltxy").reshape(n, -1), f("Y_lvel").reshape(n, -1), f("Y_lvrt").reshape(n, -1)], dim=1)
        self.pose = pose.to(device).contiguous()                 # [N, PO] reference output-vector layout
        self.audio = f("X_audio_features").to(device).contiguous()
        self.rpos = f("Y_root_pos").to(device).contiguous()
        self.rrot = f("Y_root_rot").to(device).contiguous()
        self.gaze = f("Y_gaze_pos").to(device).contiguous()
        self.PO = self.pose.shape[1]
        self.ranges_train = np.asarray(data["ranges_train"]).astype(np.int64)
        self.ranges_train_labels = np.asarray(data["ranges_train_labels"]).astype(np.int64)
        # validation clips: only rendered as sample animations at checkpoints (train.py:632-760), never trained on
        has_valid = "ranges_valid" in getattr(data, "files", data)
        self.ranges_valid = np.asarray(data["ranges_valid"]).astype(np.int64).reshape(-1, 2) if has_valid \
            else np.zeros((0, 2), np.int64)
        self.ranges_valid_labels = np.asarray(data["ranges_valid_labels"]).astype(np.int64) if has_valid \
            else np.zeros(0, np.int64)
        # window table (dataset.py:79-96): one window per start frame in [range_start, range_end - window)
        counts = np.maximum(self.ranges_train[:, 1] - window - self.ranges_train[:, 0], 0)
        self.win_sample = np.repeat(np.arange(len(counts)), counts).astype(np.int16)
        self.win_start = np.concatenate([np.arange(a, a + c) for (a, _), c in zip(self.ranges_train, counts)]
                                        or [np.zeros(0, np.int64)]).astype(np.int64)
        t = lambda k, dt=torch.float32: torch.as_tensor(np.asarray(data[k]), dtype=dt).to(device)  # noqa: E731
        self.audio_mean, self.audio_std = t("audio_input_mean"), float(np.asarray(data["audio_input_std"]))
        self.in_mean, self.in_std = t("anim_input_mean").contiguous(), t("anim_input_std").contiguous()
        self.out_mean, self.out_std = t("anim_output_mean").contiguous(), t("anim_output_std").contiguous()

    def __len__(self):
        return len(self.win_start)

    def upload_indices(self, arr):
        """An int64 index array on the device through the pinned staging ring (no host stall: _IndexUploader)."""
        return self._upload(arr)

    def example_rows(self, idx, example_len):
        """Source frame of every row of the style example (dataset.py:176-204), int64 [B, example_len]."""
        W = self.window
        r0 = self.win_start[idx]
        rng = self.ranges_train[self.win_sample[idx]]
        rs, re = rng[:, 0], rng[:, 1]
        r_last = r0 + W - 1
        ext = (example_len - W) // 2
        ws = np.minimum(ext, r0 - rs)
        we = np.minimum(ext, re - r_last)
        start = np.maximum(r0 - (ws + ext - we), rs)
        end = np.minimum(np.minimum(r_last + (we + ext - ws), re) + 1, self.n_frames)
        cur = end - start
        k = np.arange(example_len)[None, :]
        rows = np.where(k < cur[:, None], start[:, None] + k, end[:, None] - example_len + k)
        return rows.astype(np.int64)

    def sample_clip(self, split, seconds=None, range_index=None):
        """dataset.py:206-233 `get_sample`: one whole clip (cut to `seconds` at 60 fps), batch 1, on the device.
        -> dict(audio [1,T,F] normalised, pose [1,T,PO], rpos, rrot, gaze, label, frames (s, e), range_index)."""
        ranges, labels = (self.ranges_train, self.ranges_train_labels) if split == "train" \
            else (self.ranges_valid, self.ranges_valid_labels)
        if range_index is None:
            range_index = int(np.random.randint(len(ranges)))
        s, e = (int(v) for v in ranges[range_index])
        if seconds is not None:
            e = min(s + seconds * 60, e)
        audio = self.audio[s:e][None].clone()
        ops.normalize_rows_(audio, self.audio_mean, self.audio_std)
        return dict(audio=audio, pose=self.pose[s:e][None], rpos=self.rpos[s:e][None], rrot=self.rrot[s:e][None],
                    gaze=self.gaze[s:e][None].contiguous(), label=int(labels[range_index]), frames=(s, e),
                    range_index=range_index)

    def clip_example(self, frames, example_len):
        """dataset.py:176-204 `get_example(se, se, L)` as the reference calls it for sample rendering (train.py:544):
        the clip's own frames [s, e] (no gaze), tail repeated up to `example_len`, normalised.  [1, L', PO+3]"""
        s, e = frames
        end = min(e + 1, self.n_frames)
        ex = ops.fill_(torch.empty(end - s, self.PO + 3, device=self.device))
        ex[:, :self.PO] = self.pose[s:end]
        if end - s < example_len:
            ex = torch.cat([ex, ex[-example_len + (end - s):]], dim=0)
        ex = ex[None].contiguous()
        ops.normalize_rows_(ex, self.in_mean, self.in_std)
        return ex

    def batch(self, idx, example_len, bufs=None, style_encoder=None):
        """Gather one batch (normalised where the reference normalises before the nets).
        bufs: a dict of tensors of an earlier call with the same shapes to gather into (a caller that alternates two such sets
        never hands a batch tensor back to the allocator: TrainEngine.prefetch).  style_encoder: the attention encoder that will
        read the example -- the example is then gathered straight into the padded input region of a workspace of that encoder
        (ops.style_input_buffer / gather_example: one pass instead of fill + gather + normalise + the encoder's padding copy)."""
        dev = self.device
        B, T = len(idx), self.window
        bufs = {} if bufs is None else bufs
        starts = self._upload(self.win_start[idx])
        def old(k, shape):       # the buffer of an earlier call, if it still has the shape
            t = bufs.get(k)
            return t if t is not None and tuple(t.shape) == shape and t.is_contiguous() else None
        gw = lambda k, src: ops.gather_windows(src, starts, T, out=old(k, (B, T, src.shape[1])))  # noqa: E731
        gr = lambda k, src: ops.gather_rows(src, starts, out=old(k, (B, src.shape[1])), out_ld=src.shape[1])  # noqa: E731
        out = dict(audio=gw("audio", self.audio), pose=gw("pose", self.pose), rpos=gw("rpos", self.rpos),
                   rrot=gw("rrot", self.rrot), gaze=gw("gaze", self.gaze),
                   # first frame of every window (the decoder's initial pose), gathered directly: no strided copies
                   pose0=gr("pose0", self.pose), rpos0=gr("rpos0", self.rpos), rrot0=gr("rrot0", self.rrot))
        ops.normalize_rows_(out["audio"], self.audio_mean, self.audio_std)
        if example_len is not None:
            rows = self._upload(self.example_rows(idx, example_len))
            if style_encoder is not None:
                ws, xp = ops.style_input_buffer(style_encoder, B, example_len, self.PO + 3, style_encoder.training, dev,
                                                ws=bufs.get("style_ws"))
                ops.gather_example(self.pose, rows, self.in_mean, self.in_std, xp)
                out["style_ws"] = ws
                out["example"] = ops.example_view(ws, xp)
            else:
                ex = ops.fill_(torch.empty(B, example_len, self.PO + 3, device=dev))      # gaze slot = 0 (dataset.py:194)
                ops.gather_rows(self.pose, rows, out=ex, out_ld=self.PO + 3)
                ops.normalize_rows_(ex, self.in_mean, self.in_std)
                out["example"] = ex
        return out


def shard_indices(perm, batch_index, per_rank_batch, world_size, rank):
    """Window indices of `rank` for global batch `batch_index`: the global batch is the next world*per_rank entries of
    the (identical on every rank) permutation, split contiguously by rank -- drop_last semantics."""
    gb = per_rank_batch * world_size
    s = batch_index * gb
    return perm[s + rank * per_rank_batch: s + (rank + 1) * per_rank_batch]


def allreduce_mean_(flat_grad, world_size, group=None, prescaled=True, force=False):
    """ONE collective per iteration over the flat gradient buffer (RCCL on GPUs, gloo in the CPU tests).  With
    prescaled=True every rank already multiplied its gradients by 1/world (the loss kernel's gscale)."""
    if world_size > 1 or force:      # force: exercise the collective on a single rank (bench.py --force-process-group)
        torch.distributed.all_reduce(flat_grad, op=torch.distributed.ReduceOp.SUM, group=group)
        if not prescaled:
            flat_grad.mul_(1.0 / world_size)
    return flat_grad


def flatten_parameters(modules):
    """Re-home all parameters of `modules` into ONE flat fp32 buffer (+ one flat grad buffer).
    Parameter objects are kept (their .data / .grad become views), so state_dict() is unchanged."""
    params = [p for m in modules for p in m.parameters()]
    dev = params[0].device
    total = sum(p.numel() for p in params)
    flat_p = torch.empty(total, device=dev, dtype=torch.float32)
    # 4 floats behind the gradients: [0] = "a persistent sweep gave up on some rank" (rides on the gradient all-reduce,
    # TrainEngine._guard), reachable as flat_g.tail
    flat_gx = torch.zeros((total + 3) // 4 * 4 + 4, device=dev, dtype=torch.float32)
    flat_g = flat_gx[:total]
    off = 0
    for p in params:
        n = p.numel()
        flat_p[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat_p[off:off + n].view(p.shape)
        p.grad = flat_g[off:off + n].view(p.shape)
        off += n
    for m in modules:           # nn.GRU caches flattened weights; harmless on ROCm but keep it coherent
        for sub in m.modules():
            if isinstance(sub, torch.nn.GRU):
                sub._flat_weights = [getattr(sub, n) for n in sub._flat_weights_names]
    return params, flat_p, flat_g, flat_gx


class TrainEngine:
    def __init__(self, speech_encoder, decoder, style_encoder, dataset, parents, dt, lr=1e-4, eps=1e-5,
                 style_encoding_type="example", world_size=1, rank=0, process_group=None, force_allreduce=False,
                 overlap_allreduce=True, overlap_wgrads=True, early_decoder_step=True, noise_seed=None,
                 style_head_first=3, prepare_ahead=True, defer_style_wgrads=False):
        self.se, self.de, self.st = speech_encoder, decoder, style_encoder
        # the attention style encoder's six weight-gradient products leave its backward chain for the third queue (ops._StyleFn,
        # zeggs_style_encoder_bwd_part); ZEGGS_DEFER_STYLE_WGRADS=0/1 for the A/B.  OFF: measured 16.96 against 16.86 ms -- every queue is
        # busy with chip-filling products until the chain ends, the total is conserved, and the shorter chain moves the attention
        # backward into a collision with the second queue's last product (profiles/r06_style_wgrads_deferred.txt)
        self.defer_style_wgrads = bool(int(os.environ.get("ZEGGS_DEFER_STYLE_WGRADS", int(defer_style_wgrads))))
        # everything the binding needs beyond the arguments of a call (gradient targets, side stream, status words, hooks,
        # the prepared decoder workspace, optionally an own noise-seed stream) travels in THIS engine's context object --
        # two engines stepping from two threads do not see each other's (ops.EngineContext)
        self.ctx = ops.EngineContext()
        if noise_seed is not None:
            self.ctx.seed_rng = np.random.default_rng(int(noise_seed))
        self.early_decoder_step = early_decoder_step
        self.prepare_ahead = bool(int(os.environ.get("ZEGGS_PREPARE_AHEAD", int(prepare_ahead))))
        self._ahead_version = None
        self.head_first_releases = 0        # steps whose side queues were released from inside the style encoder's forward
        self.style_head_first = int(os.environ.get("ZEGGS_STYLE_HEAD_FIRST", style_head_first))
        self.ds = dataset
        self.dt = float(dt)
        self.world, self.rank, self.pg = world_size, rank, process_group
        self.force_allreduce = force_allreduce
        self.overlap_allreduce = overlap_allreduce
        # the decoder's weight-gradient GEMMs on the library's second stream, beside the encoders' backward
        on_gpu = overlap_wgrads and torch.device(dataset.device).type == "cuda"
        self.wgrad_stream = ops.side_stream(dataset.device) if on_gpu else None
        if on_gpu and not any(k in ops._OPTIONS for k in ("gemm_direct", "gemm_direct_shield", "gemm_direct_depth", "gemm_direct_reserve")):
            # Three queues share the chip in the iteration's tail.  The barrier-free stream-K product (gemm.hip:
            # gemm_tn_direct_kernel, +25-40 % on a weight-gradient product that has the chip to itself) LOSES there as it stands --
            # it needs no LDS and half the registers, so other queues' waves move in beside it (17.9 ms per iteration against 17.2
            # with the LDS-tiled kernel, whose footprint keeps a CU to itself).  Its "shield" variant allocates the whole register
            # file of its SIMDs (one wave per SIMD, 512 registers, 8 operand pairs in flight): the same isolation, the direct
            # kernel's rate -- 16.8 ms (profiles/r05_gemm_direct_ab.txt).  With more than one rank it leaves the CUs of the gradient
            # exchange out of its grids: RCCL's workgroups are resident for the whole all-reduce, and a stream-K product whose
            # workgroups cannot all be resident waits for its stragglers (gemm.hip; profiles/r06_reserve_ab.txt).
            # The routing belongs to THIS engine's calls (zeggs_gemm_route through ops.EngineContext): another engine or a plain
            # caller in the process keeps the library's defaults; a caller who set the options himself keeps his.
            self.ctx.gemm_route = (1, 1, 8, int(os.environ.get("ZEGGS_GEMM_RESERVE", 32)) if world_size > 1 else 0)
        # the speech encoder (a short chain of small launches, forward and -- autograd replays a node on the stream of its
        # forward -- backward) beside the style encoder
        self.aux_stream = torch.cuda.Stream(device=dataset.device) if on_gpu else None
        self.style_type = style_encoding_type
        dev = dataset.device
        self.parents = torch.as_tensor(np.asarray(parents), dtype=torch.int32, device=dev)
        mods = [speech_encoder, decoder] + ([style_encoder] if style_encoding_type == "example" else [])
        self.params, self.flat_p, self.flat_g, self.flat_gx = flatten_parameters(mods)
        self._gflag = self.flat_gx[self.flat_gx.numel() - 4:self.flat_gx.numel() - 3]
        # the decoder's slice of the flat buffers (module order above): its gradients are final when the decoder
        # backward returns, 91 % of the payload, while the encoders' backward still has to run
        lo = sum(p.numel() for p in speech_encoder.parameters())
        self._dec_range = (lo, lo + sum(p.numel() for p in decoder.parameters()))
        # ... in two halves when the weight-gradient GEMMs run on the side stream: [lo, split) = layer0 + GRU layer 0 is computed
        # last (ops._DecoderFn.backward), underneath the exchange of [split, hi)
        names = [n for n, _ in decoder.named_parameters()]
        cut = names.index("recurrent_decoder.layer1.weight_ih_l1") if "recurrent_decoder.layer1.weight_ih_l1" in names else 0
        self._dec_split = lo + sum(p.numel() for p in list(decoder.parameters())[:cut])
        self._dec_work = None
        self._early_dp = False              # this step applies the decoder's optimizer slices behind their own exchange
        self._flag_sent = False
        self._dec_shape = None              # (B, speech width, style width) of the last decoder call: what to prepare for
        self._prefetched = None             # (key, batch, event): the next step's batch, gathered on the third stream
        self._pf_sets, self._pf_n = ({}, {}), 0     # ... into two alternating sets of persistent buffers (prefetch)
        self._pf_used = [None, None]        # number of the step that read a set last
        self._step_no, self._last_sync_step = 0, -1     # steps started; the step in which the third stream last waited for the caller's
        self._keep_next, self._keep_prev = [], []   # tensors of the third stream's pool read on the caller's stream (step)
        # the style example gathered straight into the attention encoder's padded input (ZEGGS_EXAMPLE_IN_PLACE=0: A/B)
        self.example_in_place = bool(int(os.environ.get("ZEGGS_EXAMPLE_IN_PLACE", 1)))
        self.prefetch_hits = 0
        self.opt = RAdam(self.params, lr=lr, eps=eps)
        self.opt.attach_flat(self.flat_p, self.flat_g)
        # Give-up protocol of the persistent sweeps (include/zeggs_hip.h: ZeggsDecCall.status).  The kernels OR into status[0]
        # (sticky); the fused RAdam step reads it ON THE DEVICE (in data-parallel runs: the flag of all ranks, summed with the
        # gradients) and skips itself, counting in status[1] -- nothing invalid reaches the weights however far the host runs
        # ahead.  The host reads the counter back asynchronously with a fixed lag (the same iteration on every rank) and then
        # re-runs the lost steps on the stage kernels (_recover).
        self.status = self._status_host = None
        self._status_ring = []
        self._history = collections.deque(maxlen=8)
        self.recovered_steps = 0
        self.replayed = []                  # (iteration, loss, terms) of steps re-run by _recover(): for the caller's log
        # Re-arming (round 5).  A give-up is usually a TRANSIENT (a co-tenant held CUs for a while: RCCL kernels on an 8-GPU node,
        # another process's launch); until round 4 it switched the persistent sweeps off for the life of the process -- a permanent
        # 1.3x slowdown for one bad moment.  Now the sweeps come back after `rearm_after` clean iterations on the stage kernels
        # (every rank reaches that iteration together: all ranks skip and replay the same steps, so the schedule needs no extra
        # exchange); a give-up soon after a re-arm doubles the wait (`rearm_backoff`), up to `rearm_max`.  0 = never re-arm.
        self.rearm_after = self.REARM_AFTER
        self._rearm_wait = self.rearm_after
        self._rearm_at = None               # iteration at which the persistent sweeps are switched back on
        self._rearmed_at = None             # iteration of the last re-arm (a give-up within `_rearm_wait` of it backs off)
        self.rearm_count = 0
        if torch.device(dev).type == "cuda" and not os.environ.get("ZEGGS_NO_GUARD"):     # (env: A/B measurement of its cost)
            self.status = ops.new_status(dev)
            self.opt.attach_guard(self.status, self._gflag if (world_size > 1 or force_allreduce) else None)
        self.iteration = 0
        self.last_terms = None
        self.decoder_fwd_events = None      # bench.py: list of (start, end) HIP events around the forward rollout
        self.decoder_bwd_events = None      # bench.py: same around loss.backward() (BPTT + encoder backward)
        self.allreduce_events = None        # bench.py: (start, end) HIP events around the gradient all-reduce
        self._one = torch.ones((), device=dev, dtype=torch.float32)     # upstream gradient of loss.backward()

    def _reduce_decoder_grads(self, part=None):
        """ops hook (decoder gradients final in stream order): start the all-reduce of the decoder's gradient slice -- all of
        it (part None), or its second / first half in parameter order (part 0 / 1, ops.set_after_decoder_backward); it runs on
        the process group's stream, ordered after the kernels already on the current stream, underneath the encoders'
        backward.  Same collective sequence on every rank: decoder slice(s), then the encoder slices."""
        lo, hi = self._dec_range
        a, b = (lo, hi) if part is None else (self._dec_split, hi) if part == 0 else (lo, self._dec_split)
        early = self._early_dp and part is not None
        if early and part == 0:
            # the decoder's slices of the optimizer step are applied as soon as THEIR exchange is complete (below), underneath
            # the encoders' backward -- so the give-up flag of all ranks has to be known first: it is final here (both sweeps
            # of this iteration are behind this point in stream order) and travels ahead of the gradients, 16 bytes
            ops.status_flag(self.status, self._gflag)
            self._dec_work = (self._dec_work or []) + [torch.distributed.all_reduce(
                self.flat_gx[self.flat_gx.numel() - 4:], op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=True)]
            self._flag_sent = True
            self._early_ranges = []
        if b > a:
            self._dec_work = (self._dec_work or []) + [torch.distributed.all_reduce(
                self.flat_g[a:b], op=torch.distributed.ReduceOp.SUM, group=self.pg, async_op=True)]
            if early:
                self._early_ranges.append(((a + 3) // 4 * 4, b // 4 * 4, len(self._dec_work)))
        if early and part == 1:
            # both halves are in flight (the second one's GEMMs ran underneath the first one's exchange): now the CURRENT
            # (weight-gradient) stream -- not the host -- waits for each exchange and updates that slice behind it
            done = 0
            for a4, b4, upto in self._early_ranges:
                for w in self._dec_work[done:upto]:
                    w.wait()
                done = upto
                if b4 > a4:
                    self.opt.early(a4, b4)

    def prefetch(self, idx, example_len):
        """Gather the batch of a LATER step now, on the third stream (beside whatever the chip is doing: the gather is a
        handful of small copies that depend on nothing but the resident dataset).  step() picks it up if it is called with
        the same indices.  No-op without the side streams."""
        if self.aux_stream is None:
            return
        ex_len = example_len if self.style_type == "example" else None
        # Two alternating sets of batch buffers that live as long as the engine (round 6).  Set k is written here, on the third
        # stream, for step n and again for step n + 2 -- behind step n + 1's work on that stream, which starts with a wait for the
        # caller's stream (launch_speech), i.e. behind everything step n read it for.  Nothing goes back to the allocator, so
        # nothing needs record_stream (whose events, recorded on the CALLER's stream when a batch died, were 60 us of barrier
        # packets in front of every iteration), and the batch costs no allocation either.
        self._pf_n += 1
        k = self._pf_n % 2
        bufs = self._pf_sets[k]
        if self._pf_used[k] is not None and self._last_sync_step <= self._pf_used[k]:
            # (two prefetches without a step in between: the third stream has not waited for the caller's since the step that read
            #  this set -- do it now)
            self.aux_stream.wait_stream(torch.cuda.current_stream())
        self._pf_used[k] = None
        enc = getattr(self.st, "encoder", None) if ex_len is not None else None
        if type(enc).__name__ != "StyleEncoderAttn" or not self.example_in_place:
            enc = None                  # (the GRU style encoder has no padded first convolution: plain example tensor)
        with torch.cuda.stream(self.aux_stream):
            b = self.ds.batch(idx, ex_len, bufs=bufs, style_encoder=enc)
            # ... and the half of the loss's feature pass that depends on the batch only (ground-truth rows transposed, their
            # forward kinematics): off the serial section between the two sweeps
            b["loss_ws"] = ops.loss_prepare_truth(b["pose"], b["rpos"], b["rrot"], b["gaze"], self.parents, self.dt,
                                                  ws=bufs.get("loss_ws"))
            ev = torch.cuda.Event()
            ev.record(self.aux_stream)
        bufs.clear()
        bufs.update(b)
        self._prefetched = ((np.asarray(idx).tobytes(), ex_len), dict(b), ev, k)

    STATUS_LAG = 3       # iterations between a step and the host's look at its skip counter (identical on every rank)
    REARM_AFTER = 200    # clean stage-kernel iterations before the persistent sweeps are tried again (rearm_after; 0: never)
    REARM_MAX = 20000

    def _maybe_rearm(self):
        """Top of a (non-replay) step: switch the persistent sweeps back on when their probation is over.  The library
        re-validates a re-enabled kernel on its first launch (zeggs_set_option: state -1), and a give-up of that launch is
        handled like any other (skipped on the device, replayed)."""
        if self._rearm_at is None or self.iteration < self._rearm_at:
            return
        self._rearm_at = None
        self._rearmed_at = self.iteration
        self.rearm_count += 1
        for k in getattr(self, "_rearm_what", ("train_persistent", "bwd_persistent")):
            ops.set_option(k, 1)

    def _post_status(self):
        """After the optimizer step: copy the status words to a pinned slot (asynchronous, on the CURRENT stream: the caller's, or
        the weight-gradient stream when the step hands this to decoder_prepare) for the look STATUS_LAG steps on."""
        k = self.iteration % (self.STATUS_LAG + 1)
        while len(self._status_ring) <= self.STATUS_LAG:
            self._status_ring.append([torch.zeros(ops.STATUS_WORDS, dtype=torch.int32).pin_memory(), None])
        slot = self._status_ring[k]
        # (round 5: moving this copy to a stream of its own was tried -- the ~75 us between the optimizer's last kernel and the next
        #  iteration's first are not the copy's: they stayed -- and cost the data-parallel schedule 2.7 ms: a fifth stream beside
        #  RCCL's oversubscribes the hardware queues, bench.launch_env)
        slot[0].copy_(self.status, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()

    def _check_status(self):
        """Before a step: the skip counter as of STATUS_LAG steps ago (its copy has long completed: no stall)."""
        if not self._status_ring or self.iteration < self.STATUS_LAG:
            return
        slot = self._status_ring[(self.iteration - self.STATUS_LAG) % (self.STATUS_LAG + 1)]
        if slot[1] is None:
            return
        slot[1].synchronize()
        if int(slot[0][1]) > 0:
            self._recover()

    def _recover(self):
        """A persistent sweep gave up (on this or another rank): the device skipped every optimizer step since.  Disable the
        persistent training kernels for this process, clear the words, and re-run the lost steps -- same windows, same noise
        seeds -- on the stage kernels."""
        torch.cuda.synchronize()
        st = self.status.cpu()
        bits, n = int(st[0]), int(st[1])
        if self.rearm_after > 0:
            # probation: back on after `_rearm_wait` clean iterations; a give-up that follows a re-arm closely doubles the wait
            if self._rearmed_at is not None and self.iteration - n - self._rearmed_at <= self._rearm_wait:
                self._rearm_wait = min(2 * self._rearm_wait, self.REARM_MAX)
            else:
                self._rearm_wait = self.rearm_after
            self._rearm_at = self.iteration - n + self._rearm_wait      # (counted from the first lost step: the same on every rank)
        warnings.warn(f"zeggs: persistent sweep gave up on this rank: {[v for b, v in ops.GAVE_UP.items() if bits & b]}; "
                      f"{n} optimizer step(s) were skipped on the device and are re-run on the stage kernels "
                      "(train_persistent / bwd_persistent off"
                      + (f", back on after {self._rearm_wait} clean iterations)" if self.rearm_after > 0 else " for this process)"))
        # (what the caller had on before: only that is re-armed later -- a sweep he switched off for an A/B stays off)
        self._rearm_what = [k for k in ("train_persistent", "bwd_persistent") if ops._OPTIONS.get(k, 1)]
        ops.set_option("train_persistent", 0)
        ops.set_option("bwd_persistent", 0)
        ops.fill_(self.status.view(torch.float32))
        for slot in self._status_ring:
            slot[1] = None
        redo = list(self._history)[-n:] if n > 0 else []
        if len(redo) < n:
            raise RuntimeError(f"zeggs: {n} optimizer steps were skipped but only {len(redo)} are in the replay history")
        self.iteration -= n
        self.opt.rewind(n)
        self._prefetched = None
        self.recovered_steps += n
        rng = self.ctx.rng()
        after = copy.deepcopy(rng.bit_generator.state)       # the stream continues where the (skipped) steps left it
        # a replayed step runs with the learning rate it had when it was first tried (the caller may have applied the
        # (iteration + 1) % 1000 decay since -- zeggs.train does it before the lagged look at the skip counter)
        lr_now = [grp["lr"] for grp in self.opt.param_groups]
        try:
            for h in redo:
                rng.bit_generator.state = copy.deepcopy(h["seed_state"])
                for grp, lr in zip(self.opt.param_groups, h.get("lr", lr_now)):
                    grp["lr"] = lr
                loss = self.step(h["idx"], h["example_len"], eps=h["eps"], labels=h["labels"], _replay=True)
                self.replayed.append((self.iteration - 1, loss, self.last_terms))
        finally:
            for grp, lr in zip(self.opt.param_groups, lr_now):
                grp["lr"] = lr
        rng.bit_generator.state = after

    def flush(self):
        """Drain the give-up protocol NOW (a device synchronisation): _check_status() looks at an optimizer step STATUS_LAG
        iterations after it ran, so steps skipped within the last STATUS_LAG iterations are still unknown to the host.  Call
        before anything reads the weights for keeps -- a checkpoint, rendered samples, the end of training (zeggs.train does);
        on every rank at the same iteration in data-parallel runs (all ranks skip the same steps).  Returns the number of steps
        that were re-run."""
        if self.status is None:
            return 0
        torch.cuda.synchronize()
        n = int(self.status.cpu()[1])
        if n > 0:
            self._recover()
        for slot in self._status_ring:
            slot[1] = None
        return n

    def step(self, idx, example_len, eps=None, labels=None, _replay=False):
        """One training iteration on window indices `idx` (this rank's slice). Returns the loss tensor (device)."""
        if self.status is not None and not _replay:
            self._check_status()
            self._maybe_rearm()
            self._history.append(dict(idx=np.array(idx, copy=True), example_len=example_len, eps=eps, labels=labels,
                                      seed_state=copy.deepcopy(self.ctx.rng().bit_generator.state),
                                      lr=[grp["lr"] for grp in self.opt.param_groups]))
        ds, T = self.ds, self.ds.window
        ex_len = example_len if self.style_type == "example" else None
        pre, self._prefetched = self._prefetched, None
        self._step_no += 1
        hit = pre is not None and pre[0] == (np.asarray(idx).tobytes(), ex_len)
        if hit:
            b = pre[1]
            self._pf_used[pre[3]] = self._step_no
            self.prefetch_hits += 1
            torch.cuda.current_stream().wait_event(pre[2])
            # (no record_stream: the tensors are views of this engine's two alternating buffer sets (prefetch), never returned to the
            #  allocator.  Recorded, every one of the ten cost an event on THIS stream when the batch went out of scope -- 60 us of
            #  barrier packets between the optimizer's last kernel and the next iteration's first, profiles/r06_iteration_head.txt)
        else:
            b = ds.batch(idx, ex_len)
        if self.aux_stream is None:
            ops.fill_(self.flat_gx)
        # (with the third stream the zero fill of the flat gradient buffer -- 100 MB, 18 us -- runs THERE, in front of the speech
        #  encoder (launch_speech): behind that stream's wait for this one, i.e. behind the optimizer kernels that read the previous
        #  gradients, and in front of every gradient write of this step: this stream joins the third one before the decoder's
        #  forward, the weight-gradient stream joins this one after the BPTT sweep)
        ctx = self.ctx
        ctx.direct_grads = True             # *_bwd kernels write straight into the flat gradient buffer
        ctx.status = self.status
        overlap = self.overlap_allreduce and (self.world > 1 or self.force_allreduce)
        self._dec_work = None
        ctx.after_decoder_backward = self._reduce_decoder_grads if overlap else None
        ctx.wgrad_stream = self.wgrad_stream
        ctx.defer_style_wgrads = bool(self.defer_style_wgrads and self.aux_stream is not None)
        ctx.deferred_wgrads = []
        # data-parallel twin of the early decoder step below: needs the two-halves exchange on the weight-gradient stream
        # (the slices are then updated there, behind their own all-reduce) and the device-side guard (flag first)
        self._early_dp = bool(self.early_decoder_step and overlap and self.wgrad_stream is not None and self.status is not None)
        self._flag_sent = False
        ctx.decoder_grads_final = None
        if self.early_decoder_step and not overlap and self.world == 1 and not self.force_allreduce:
            # no exchange to wait for: the decoder's 88 % of the optimizer step (HBM-bound) runs on the weight-gradient stream as
            # soon as its gradients are final, underneath the encoders' backward (matrix-core-bound) instead of after it
            lo, hi = self._dec_range
            ctx.decoder_grads_final = lambda: self.opt.early((lo + 3) // 4 * 4, hi // 4 * 4)
        done = False
        try:
            with ops.use(ctx):
                cur = torch.cuda.current_stream() if self.aux_stream is not None else None
                box = {}

                def launch_prepare():
                    if ctx.prepared is not None and ctx.prepared[0][0] == len(idx):
                        if self._ahead_version == self.flat_p._version:
                            return          # made at the end of the previous step (prepare_ahead)
                        # the flat weight buffer was edited in place since (the parameters are views of it with version counters
                        # of their own, which ops checks): those packs are stale
                        with ops.use(ctx):
                            ops._drop_prepared(ctx)
                    if self.wgrad_stream is not None and self._dec_shape is not None and self._dec_shape[0] == len(idx):
                        # the weight-only packs of the decoder sweeps, beside the encoders' forward
                        Bd, SP, ST = self._dec_shape
                        ops.decoder_prepare(self.de, Bd, T, SP, ST, ds.in_mean, ds.in_std, ds.out_mean, ds.out_std, self.dt,
                                            self.wgrad_stream)

                def launch_speech():
                    if self.aux_stream is not None:
                        self.aux_stream.wait_stream(cur)            # the batch was gathered on the current stream
                        self._last_sync_step = self._step_no
                        # (what the previous step parked: whatever takes these blocks from the third stream's pool from here on runs
                        #  behind that wait, i.e. behind the last read on the caller's stream)
                        self._keep_prev, self._keep_next = self._keep_next, []
                        self._keep_prev.clear()
                        if not hit:
                            b["audio"].record_stream(self.aux_stream)
                        with torch.cuda.stream(self.aux_stream):
                            ops.fill_(self.flat_gx)
                            box["speech"] = self.se(b["audio"])
                    else:
                        box["speech"] = self.se(b["audio"])

                # style_head_first (bit 0: the decoder's packs, bit 1: the speech encoder): these queues are released only BEHIND
                # the style encoder's first convolution (ops._StyleFn calls the hook between the two parts of
                # zeggs_style_encoder_fwd_part), which fills the chip on its own and heads the longest chain before the sweep
                held = []
                example = self.style_type == "example"
                side = self.aux_stream is not None and self.wgrad_stream is not None
                # (only the attention encoder's op has the two-part forward that calls the hook: with the GRU style encoder nothing
                #  is held back -- held work would start BEHIND that encoder instead of beside it)
                hooked = example and side and type(getattr(self.st, "encoder", None)).__name__ == "StyleEncoderAttn"
                for bit, fn in ((1, launch_prepare), (2, launch_speech)):
                    if hooked and self.style_head_first & bit:
                        held.append(fn)
                    else:
                        fn()

                def release(inside=False):
                    ctx.after_style_head = None
                    self.head_first_releases += bool(held and inside)
                    with torch.enable_grad():       # (called from inside an autograd.Function's forward, where grad mode is off)
                        while held:
                            held.pop(0)()
                ctx.after_style_head = (lambda: release(True)) if held else None
                mu = logvar = None
                if example:
                    z, mu, logvar = self.st(b["example"], 1.0, eps=eps)
                else:
                    z = labels
                release()                   # (a style module that did not go through ops.style_encoder: nothing was held back)
                speech = box["speech"]
                style = ops.broadcast_time(z, T)
                if self.aux_stream is not None:
                    cur.wait_stream(self.aux_stream)
                    # (the encoding was allocated on the third stream and is read on this one: kept alive until that stream has
                    #  waited for this one again -- launch_speech of the next step -- instead of a record_stream, whose event lands on
                    #  this stream when the tensor dies)
                    self._keep_next.append(speech)
                ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
                if self.decoder_fwd_events is not None:
                    e0 = ev()
                    e0.record()
                self._dec_shape = (speech.shape[0], speech.shape[2], style.shape[2])
                pose, orp, orr = ops.decoder_core(self.de, b["pose0"], b["rpos0"], b["rrot0"], b["gaze"], speech, style,
                                                  ds.in_mean, ds.in_std, ds.out_mean, ds.out_std, self.dt)
                if self.decoder_fwd_events is not None:
                    e1 = ev()
                    e1.record()
                    self.decoder_fwd_events.append((e0, e1))
                klw = kl_div_weight(self.iteration) if mu is not None else 0.0
                loss, terms = ops.training_loss(pose, orp, orr, b["pose"], b["rpos"], b["rrot"], b["gaze"], self.parents,
                                                self.dt, mu, logvar, kl_weight=klw, gscale=1.0 / self.world,
                                                unit_grad=True, truth_ws=b.get("loss_ws"))
                if self.decoder_bwd_events is not None:
                    e2 = ev()
                    e2.record()
                loss.backward(self._one)
                if ctx.deferred_wgrads:
                    # the style encoder's weight-gradient products, enqueued LAST on the third queue (behind the speech encoder's
                    # backward, which autograd has enqueued by now): they run beside the end of the chain and the second queue's last
                    # work instead of inside the chain (joined below, with the speech encoder's gradients)
                    with torch.cuda.stream(self.aux_stream):
                        for evd, fn in ctx.deferred_wgrads:
                            self.aux_stream.wait_event(evd)
                            fn()
                    ctx.deferred_wgrads = []
                if self.decoder_bwd_events is not None:
                    e3 = ev()
                    e3.record()
                    self.decoder_bwd_events.append((e2, e3))
            done = True
        finally:
            ctx.after_style_head = None
            ctx.deferred_wgrads = []
            if not done:
                # the step did not complete (OOM, an error in a backward, KeyboardInterrupt): forget the slices the optimizer
                # may already have applied early -- a stale list would make the NEXT step() skip the decoder slice -- and give
                # the decoder workspaces back once the side stream is done with them
                self.opt._early = []
                if self.wgrad_stream is not None:
                    torch.cuda.current_stream().wait_stream(self.wgrad_stream)
                ctx.release_wgrad_workspaces()
        if self.wgrad_stream is not None:      # join: every decoder gradient is final from here on in stream order
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            ctx.release_wgrad_workspaces()
        if self.aux_stream is not None:        # ... and the speech encoder's
            torch.cuda.current_stream().wait_stream(self.aux_stream)
        if self.allreduce_events is not None:       # with the overlap on: the EXPOSED part of the exchange
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
        guarded_dp = self.status is not None and (self.world > 1 or self.force_allreduce)
        if guarded_dp and not self._flag_sent:      # this rank's give-up flag joins the last slice of the exchange (summed over the ranks)
            ops.status_flag(self.status, self._gflag)
        if self._dec_work is not None:
            # the decoder slice has been in flight since the decoder backward returned; now the encoders' slices
            lo, hi = self._dec_range
            works = list(self._dec_work)
            tail = self.flat_g[hi:] if self._flag_sent else self.flat_gx[hi:]      # (flag already exchanged: not summed twice)
            for part in (self.flat_g[:lo], tail):
                if part.numel():
                    works.append(torch.distributed.all_reduce(part, op=torch.distributed.ReduceOp.SUM, group=self.pg,
                                                              async_op=True))
            for w in works:
                w.wait()
            self._dec_work = None
        else:
            allreduce_mean_(self.flat_gx, self.world, self.pg, prescaled=True, force=self.force_allreduce)
        if self.allreduce_events is not None:
            a1.record()
            self.allreduce_events.append((a0, a1))
        self.opt.step()
        if self.prepare_ahead and self.wgrad_stream is not None and self._dec_shape is not None and not _replay:
            # the NEXT step's weight-only packs, as soon as the optimizer is through: the second queue does the fold products in
            # the gap between two iterations and under the style encoder's input staging instead of beside its tail (the
            # forward checks shape, weight pointers and version counters before it picks the workspace up)
            Bd, SP, ST = self._dec_shape
            # (the status read-back rides on that queue too, behind the packs: it waits for this stream's optimizer kernels anyway,
            #  and on the caller's stream the copy + its event stood in front of the next iteration's first kernel)
            with ops.use(ctx):
                ops.decoder_prepare(self.de, Bd, T, SP, ST, ds.in_mean, ds.in_std, ds.out_mean, ds.out_std, self.dt,
                                    self.wgrad_stream, after=self._post_status if self.status is not None else None)
            self._ahead_version = self.flat_p._version
        elif self.status is not None:
            self._post_status()
        self.iteration += 1
        self.last_terms = terms
        return loss
