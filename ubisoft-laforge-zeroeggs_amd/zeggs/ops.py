"""ctypes binding of libzeggs_hip.so (C ABI: include/zeggs_hip.h) + autograd glue.

PyTorch is used here only as plumbing: device memory (tensors), the current HIP
stream and autograd bookkeeping.  Every arithmetic op of the hot path is a call
into the hand-written gfx950 library.  There is deliberately NO fallback: if the
shared library is missing or a tensor is not on a GPU, these functions raise.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np
import torch

_LIB = None
_LIB_PATH = Path(os.environ.get("ZEGGS_LIB") or Path(__file__).resolve().parent / "libzeggs_hip.so")   # ZEGGS_LIB: A/B builds
c_f = C.c_void_p  # device pointers are passed as raw addresses


class HipLibraryMissing(RuntimeError):
    pass


# ----------------------------------------------------------------------------- structs (mirror zeggs_hip.h)
class SpeechDims(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("F", C.c_int), ("H", C.c_int), ("O", C.c_int), ("KW", C.c_int),
                ("dropout_p", C.c_float), ("seed", C.c_uint64)]


class StyleDims(C.Structure):
    _fields_ = [("B", C.c_int), ("L", C.c_int), ("C", C.c_int), ("H", C.c_int), ("E", C.c_int), ("NH", C.c_int),
                ("dropout", C.c_int), ("seed", C.c_uint64)]


class DecDims(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("PI", C.c_int), ("PO", C.c_int), ("SP", C.c_int), ("ST", C.c_int),
                ("H", C.c_int), ("dt", C.c_float), ("film", C.c_int)]


class LossDims(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("J", C.c_int), ("S", C.c_int), ("dt", C.c_float)]


def _ptr_struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": [(f, C.c_void_p) for f in fields]})


SPEECH_FIELDS = ("w0", "b0", "w1", "b1", "w2", "b2")
STYLE_FIELDS = ("c0_w", "c0_b", "ln0_g", "ln0_b", "c4_w", "c4_b", "ln1_g", "ln1_b", "in_w", "in_b", "out_w", "out_b",
                "lna_g", "lna_b", "ff0_w", "ff0_b", "ff2_w", "ff2_b", "lnf_g", "lnf_b")
DEC_FIELDS = ("l0_w", "l0_b", "w_ih0", "w_hh0", "b_ih0", "b_hh0", "w_ih1", "w_hh1", "b_ih1", "b_hh1", "l2_w", "l2_b",
              "c0_w", "c0_b", "c1_w", "c1_b", "c2_w", "c2_b",
              "l3_w", "l3_b", "g_w", "g_b", "be_w", "be_b")          # last six: rnn_cond "film" only (else NULL)
SpeechPtrs = _ptr_struct("SpeechPtrs", SPEECH_FIELDS)
StylePtrs = _ptr_struct("StylePtrs", STYLE_FIELDS)
DecPtrs = _ptr_struct("DecPtrs", DEC_FIELDS)
DecStats = _ptr_struct("DecStats", ("in_mean", "in_std", "out_mean", "out_std"))


class DecCall(C.Structure):       # mirrors ZeggsDecCall: the per-call controls of zeggs_decoder_fwd_ex / _bwd_ex
    _fields_ = [("prepared", C.c_int), ("defer_wgrads", C.c_int), ("wgrad_stream", C.c_void_p), ("status", C.c_void_p),
                ("grads_zeroed", C.c_int)]


COUNTERS = {}                                      # which optional paths ran (tests): "style_in_place"
STATUS_WORDS = 4                                   # ZEGGS_STATUS_WORDS
GAVE_UP = {1: "B=1 decode kernel", 2: "training rollout", 4: "BPTT sweep"}     # ZEGGS_GAVE_UP_* bits of status[0]


def lib():
    """Load the HIP library (built by __graft_entry__.build() / csrc/build.sh)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not _LIB_PATH.exists():
        raise HipLibraryMissing(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). The ZeroEGGS MI355X engine has no CPU fallback.")
    L = C.CDLL(str(_LIB_PATH))
    L.zeggs_last_error.restype = C.c_char_p
    for n in ("zeggs_speech_encoder_workspace_bytes", "zeggs_style_encoder_workspace_bytes",
              "zeggs_decoder_workspace_bytes", "zeggs_loss_workspace_bytes", "zeggs_style_encoder_input_offset"):
        getattr(L, n).restype = C.c_size_t
    _LIB = L
    # tuning switches for experiments, e.g. ZEGGS_OPTIONS="bwd_chunks=4,stage_variant=0" (see zeggs_set_option)
    for kv in filter(None, os.environ.get("ZEGGS_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        if L.zeggs_set_option(k.strip().encode(), int(v)) != 0:
            raise RuntimeError(f"ZEGGS_OPTIONS: {L.zeggs_last_error().decode()}")
        _OPTIONS[k.strip()] = int(v)
    return L


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {lib().zeggs_last_error().decode()}")


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("zeggs.ops: tensor is not on a GPU (the HIP engine has no CPU path)")
    if t.dtype not in (torch.float32, torch.int64, torch.int32, torch.uint8):
        raise RuntimeError(f"zeggs.ops: unsupported dtype {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError("zeggs.ops: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


def _ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


_seed_rng = np.random.default_rng(0x5EED)


def next_seed():
    rng = current().seed_rng
    return int((rng if rng is not None else _seed_rng).integers(1, 2 ** 62))


def manual_seed(s):
    """Seed of the dropout-mask / VAE-noise stream (counter-based hash inside the kernels).  Process-wide default stream; an
    EngineContext with its own `seed_rng` (TrainEngine(noise_seed=...)) draws from that instead."""
    global _seed_rng
    _seed_rng = np.random.default_rng(int(s))


# ----------------------------------------------------------------------------- per-engine call context
# What a training loop needs beyond the arguments of a call -- where gradients are written, the stream of the deferred
# weight-gradient GEMMs, the status words, hooks, a prepared workspace -- lives in an EngineContext OBJECT, not in module
# globals: `with ops.use(ctx):` makes it the context of the calling THREAD, every autograd Function captures the context of its
# forward and its backward (which autograd runs on another thread) uses that one.  Two engines stepping from two threads
# therefore never see each other's switches (tests/test_gpu_streams.py::test_two_engines_on_two_threads).
import threading  # noqa: E402


class EngineContext:
    def __init__(self):
        self.direct_grads = False          # *_bwd kernels write parameter gradients straight into the parameters' .grad
        self.wgrad_stream = None           # torch stream of the decoder's deferred weight-gradient GEMMs (ZeggsDecCall)
        self.status = None                 # device status words of the training-mode decoder calls (ZeggsDecCall.status)
        self.after_style_head = None       # hook fn(): the style encoder's first product is enqueued (engine.style_head_first)
        self.after_decoder_backward = None # hook fn(part): decoder gradients final in stream order (data-parallel exchange)
        self.decoder_grads_final = None    # hook fn(): runs on wgrad_stream behind ALL decoder gradients (early RAdam slice)
        self.prepared = None               # (key, workspace, event, mask) of decoder_prepare, until a forward picks it up
        self.prepared_hits = 0
        self.wgrad_keepalive = []          # workspaces the deferred GEMMs still read, until the caller joined wgrad_stream
        self.seed_rng = None               # own noise-seed generator (None: the process-wide stream of ops.manual_seed)
        self.gemm_route = None             # (direct, shield, depth, reserve) of THIS context's weight-gradient products (zeggs_gemm_route)
        self.defer_style_wgrads = False    # the attention style encoder's backward enqueues its chain only and leaves its six weight-
        self.deferred_wgrads = []          # gradient products here as (event, fn()): the caller runs them on another stream and joins

    def release_wgrad_workspaces(self):
        """After the caller has made its stream wait for wgrad_stream: the decoder workspaces the deferred weight-gradient
        GEMMs were reading may go back to the allocator (kept alive here instead of `record_stream`-ed, see decoder_prepare)."""
        self.wgrad_keepalive.clear()

    def rng(self):
        return self.seed_rng if self.seed_rng is not None else _seed_rng


_DEFAULT_CTX = EngineContext()          # plain autograd use (tests, the reference's own loop): nothing special
_tls = threading.local()


def _route(ectx):
    """Entry of every call that launches matrix products: the GEMM routing of the context the call belongs to becomes the calling
    thread's (forward: the thread inside `ops.use`, backward: autograd's worker thread, with the context its forward captured).  One
    ctypes call when the thread's route changes, nothing otherwise."""
    r = ectx.gemm_route or (-1, -1, -1, -1)
    if getattr(_tls, "route", (-1, -1, -1, -1)) != r:
        lib().zeggs_gemm_route(*[int(x) for x in r])
        _tls.route = r


def current():
    return getattr(_tls, "ctx", None) or _DEFAULT_CTX


class use:
    """`with ops.use(ctx):` -- ctx is the EngineContext of this thread inside the block."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        self.prev = getattr(_tls, "ctx", None)
        _tls.ctx = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        _tls.ctx = self.prev


_SIDE_STREAMS = {}


def side_stream(device=None):
    """The library's low-priority second stream of `device` as a torch stream (zeggs_side_stream)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _SIDE_STREAMS:
        with torch.cuda.device(idx):
            h = C.c_void_p(0)
            _check(lib().zeggs_side_stream(C.byref(h)), "side_stream")
            _SIDE_STREAMS[idx] = torch.cuda.ExternalStream(h.value, device=torch.device("cuda", idx))
    return _SIDE_STREAMS[idx]




def new_status(device):
    """Zeroed status words (ZeggsDecCall.status): [0] sticky ZEGGS_GAVE_UP_* bits, [1] optimizer steps skipped because of them."""
    t = torch.empty(STATUS_WORDS, dtype=torch.int32, device=device)
    fill_(t.view(torch.float32))
    return t


_PLAIN_TLS = threading.local()


def _plain_status(dev):
    """Status words of a decoder call made OUTSIDE an engine context: one per device and host thread (concurrent rollouts do not
    share it), held in thread-local storage so that it is freed with the thread (a dict keyed by thread id grew for
    ever in programs that spawn threads).  Cost to know about: the caller of such a call reads the word back right after the
    rollout -- ONE 4-byte device-to-host read (a host synchronisation) per rollout, forward and backward, and only once a
    persistent kernel has been validated on this process; an engine (ops.use(EngineContext) with a status of its own, as
    zeggs.engine.TrainEngine) never pays it -- it reads its words with a lag of three iterations."""
    per_dev = getattr(_PLAIN_TLS, "words", None)
    if per_dev is None:
        per_dev = _PLAIN_TLS.words = {}
    st = per_dev.get(dev.index)
    if st is None:
        st = per_dev[dev.index] = new_status(dev)
    return st


def _grad_targets(orig_params, params, ectx):
    """-> (tensors the backward kernel writes, values returned to autograd).  Direct mode (EngineContext.direct_grads, only
    engine.TrainEngine turns it on): the *_bwd entry points write parameter gradients STRAIGHT into the parameters' existing
    `.grad` tensors (views of the engine's flat gradient buffer) and the autograd Functions return None for them, so no
    AccumulateGrad add runs.  Semantics are OVERWRITE (every parameter is used by exactly one op per iteration); the `.grad`
    tensors MUST be zero when the backward starts (the engine zeroes its flat buffer once per step): the kernels are told so
    (`grads_zeroed`) and accumulate onto them instead of zero-filling each one first."""
    outs, rets = [], []
    for o, t in zip(orig_params, params):
        g = getattr(o, "grad", None) if ectx.direct_grads else None
        if g is not None and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.shape == t.shape:
            outs.append(g)
            rets.append(None)
        else:
            e = torch.empty_like(t)
            outs.append(e)
            rets.append(e)
    return outs, rets


def fill_(t, value=0.0):
    _check(lib().zeggs_fill(_p(t), C.c_long(t.numel()), C.c_float(value), _stream()), "fill")
    return t


def scale_copy(src, dev_scale=None, alpha=1.0, out=None):
    """out = alpha * dev_scale[0] * src (dev_scale: 1-element device tensor or None)"""
    src = src if src.is_contiguous() else src.contiguous()
    out = torch.empty_like(src) if out is None else out
    _check(lib().zeggs_scale_copy(_p(out), _p(src), C.c_long(src.numel()), _p(dev_scale) if dev_scale is not None else None,
                                  C.c_float(alpha), _stream()), "scale_copy")
    return out


def randn(shape, device, seed=None):
    """Standard normal samples from the library's counter-hash stream (seeded by ops.manual_seed)."""
    out = torch.empty(*shape, device=device, dtype=torch.float32)
    _check(lib().zeggs_randn(_p(out), C.c_long(out.numel()), C.c_uint64(next_seed() if seed is None else int(seed)),
                             _stream()), "randn")
    return out


class _BroadcastTimeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, T):
        z = _f32c(z)
        B, S = z.shape
        out = torch.empty(B, T, S, device=z.device, dtype=torch.float32)
        _check(lib().zeggs_broadcast_time(_p(out), _p(z), B, T, S, _stream()), "broadcast_time")
        ctx.dims = (B, T, S)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, S = ctx.dims
        dz = torch.empty(B, S, device=dout.device, dtype=torch.float32)
        _check(lib().zeggs_sum_time(_p(dz), _p(_f32c(dout)), B, T, S, _stream()), "sum_time")
        return dz, None


def broadcast_time(z, T):
    """z [B,S] -> [B,T,S] contiguous (style_encoding.unsqueeze(1).repeat_interleave(T, 1), ZEGGS/train.py:256)"""
    return _BroadcastTimeFn.apply(z, int(T))


def _ptrs(cls, fields, tensors):
    s = cls()
    for f, t in zip(fields, tensors):
        setattr(s, f, t.data_ptr() if t is not None else None)
    return s


# ----------------------------------------------------------------------------- raw GEMM (tests)
def gemm(A, B, Cmat, M, N, K, sa, sb, sc, bias=None, nbatch=1, bs=(0, 0, 0), alpha=1.0, beta=0.0, act=0):
    rc = lib().zeggs_gemm(_p(A), _p(B), _p(Cmat), _p(bias), M, N, K, C.c_long(sa[0]), C.c_long(sa[1]),
                          C.c_long(sb[0]), C.c_long(sb[1]), C.c_long(sc[0]), C.c_long(sc[1]), nbatch,
                          C.c_long(bs[0]), C.c_long(bs[1]), C.c_long(bs[2]), C.c_float(alpha), C.c_float(beta), act,
                          _stream())
    _check(rc, "zeggs_gemm")
    return Cmat


# ----------------------------------------------------------------------------- speech encoder
class _SpeechFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1, w2, b2, p, seed):
        x = _f32c(x)
        ctx.ectx = current()
        _route(ctx.ectx)
        ctx.orig = (w0, b0, w1, b1, w2, b2)
        params = [_f32c(t) for t in (w0, b0, w1, b1, w2, b2)]
        B, T, F = x.shape
        d = SpeechDims(B, T, F, w0.shape[0], w2.shape[0], w1.shape[2], float(p), int(seed))
        L = lib()
        ws = _ws(L.zeggs_speech_encoder_workspace_bytes(C.byref(d)), x.device)
        out = torch.empty(B, T, d.O, device=x.device, dtype=torch.float32)
        P = _ptrs(SpeechPtrs, SPEECH_FIELDS, params)
        _check(L.zeggs_speech_encoder_fwd(C.byref(d), C.byref(P), _p(x), _p(out), _p(ws), C.c_size_t(ws.numel()),
                                          _stream()), "speech_encoder_fwd")
        ctx.d, ctx.ws = d, ws
        ctx.save_for_backward(x, out, *params)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, out, *params = ctx.saved_tensors
        L = lib()
        _route(ctx.ectx)
        grads, rets = _grad_targets(ctx.orig, params, ctx.ectx)
        P = _ptrs(SpeechPtrs, SPEECH_FIELDS, params)
        G = _ptrs(SpeechPtrs, SPEECH_FIELDS, grads)
        zeroed = int(ctx.ectx.direct_grads and all(r is None for r in rets))     # the engine zeroes its flat gradient buffer per step
        _check(L.zeggs_speech_encoder_bwd_ex(C.byref(ctx.d), C.byref(P), _p(x), _p(out), _p(_f32c(dout)), C.byref(G),
                                             _p(ctx.ws), C.c_size_t(ctx.ws.numel()), _stream(), zeroed), "speech_encoder_bwd")
        return (None, *rets, None, None)


def speech_encoder(x, w0, b0, w1, b1, w2, b2, p=0.0):
    return _SpeechFn.apply(x, w0, b0, w1, b1, w2, b2, p, next_seed() if p > 0 else 0)


# ----------------------------------------------------------------------------- style encoder
_POS_CACHE = {}


def positional_table(length, dim, device):
    """Sinusoidal table of the reference PositionalEncoding (modules.py:450-459), built on the host exactly as
    the reference does (float32 torch ops) and cached on the device."""
    key = (dim, str(device))
    tab = _POS_CACHE.get(key)
    if tab is None or tab.shape[0] < length:
        n = max(length, 1024)
        pos = torch.arange(0, n, dtype=torch.float).unsqueeze(1)
        div = torch.exp(torch.arange(0, dim, 2).float() * (-np.log(10000.0) / dim))
        t = torch.zeros(n, dim)
        t[:, 0::2] = torch.sin(pos * div)
        t[:, 1::2] = torch.cos(pos * div)
        tab = t.to(device)
        _POS_CACHE[key] = tab
    return tab


def style_param_list(enc):
    """Parameters of a StyleEncoderAttn module in the C-ABI order (STYLE_FIELDS)."""
    c, b = enc.convs, enc.blocks[0]
    mha = b.attention.multi_head_attention
    return [c[0].conv.weight, c[0].conv.bias, c[2].weight, c[2].bias, c[4].conv.weight, c[4].conv.bias,
            c[6].weight, c[6].bias, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias,
            b.attention.layer_norm.weight, b.attention.layer_norm.bias,
            b.feed_forward.convs[0].conv.weight, b.feed_forward.convs[0].conv.bias,
            b.feed_forward.convs[2].conv.weight, b.feed_forward.convs[2].conv.bias,
            b.feed_forward.layer_norm.weight, b.feed_forward.layer_norm.bias]


class _StyleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, dropout, seed, nheads, *params):
        ctx.ectx = current()
        _route(ctx.ectx)
        ctx.orig = params
        params = [_f32c(t) for t in params]
        B, Lx, Cx = x.shape
        H, E = params[0].shape[0], params[4].shape[0]
        d = StyleDims(B, Lx, Cx, H, E, nheads, int(dropout), int(seed))
        L = lib()
        need = int(L.zeggs_style_encoder_workspace_bytes(C.byref(d)))
        # an example that style_input_buffer() + gather_example() put where the workspace keeps the first convolution's padded input:
        # the encoder runs IN that workspace and skips its own padding copy (zeggs_style_encoder_fwd_part, part | 4); anything that
        # does not match exactly (another shape, a copy of the tensor, a workspace of another size) takes the ordinary path
        ws, inplace = getattr(x, "_zeggs_style_ws", None), 0
        if ws is not None and x.dtype == torch.float32 and ws.numel() == need and x.stride() == ((Lx + 2) * Cx, Cx, 1) and \
                x.data_ptr() == ws.data_ptr() + int(L.zeggs_style_encoder_input_offset(C.byref(d))) + 4 * Cx:
            inplace, xptr = 4, C.c_void_p(x.data_ptr())
            COUNTERS["style_in_place"] = COUNTERS.get("style_in_place", 0) + 1
        else:
            x = _f32c(x)
            ws, xptr = _ws(need, x.device), _p(x)
        out = torch.empty(B, E, device=x.device, dtype=torch.float32)
        P = _ptrs(StylePtrs, STYLE_FIELDS, params)
        hook = getattr(ctx.ectx, "after_style_head", None)
        if hook is not None:       # (an engine that releases its other queues once the chip-filling first product is enqueued)
            _check(L.zeggs_style_encoder_fwd_part(C.byref(d), C.byref(P), xptr, _p(pos), _p(out), _p(ws),
                                                  C.c_size_t(ws.numel()), _stream(), 1 | inplace), "style_encoder_fwd (head)")
            hook()
            _check(L.zeggs_style_encoder_fwd_part(C.byref(d), C.byref(P), xptr, _p(pos), _p(out), _p(ws),
                                                  C.c_size_t(ws.numel()), _stream(), 2 | inplace), "style_encoder_fwd (rest)")
        else:
            _check(L.zeggs_style_encoder_fwd_part(C.byref(d), C.byref(P), xptr, _p(pos), _p(out), _p(ws),
                                                  C.c_size_t(ws.numel()), _stream(), 3 | inplace), "style_encoder_fwd")
        ctx.d, ctx.ws = d, ws
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def backward(ctx, dout):
        params = list(ctx.saved_tensors)
        L = lib()
        _route(ctx.ectx)
        grads, rets = _grad_targets(ctx.orig, params, ctx.ectx)
        P = _ptrs(StylePtrs, STYLE_FIELDS, params)
        G = _ptrs(StylePtrs, STYLE_FIELDS, grads)
        zeroed = int(ctx.ectx.direct_grads and all(r is None for r in rets))
        dout = _f32c(dout)
        ectx = ctx.ectx
        if ectx.defer_style_wgrads and zeroed:
            # the chain now, on this stream; the six weight-gradient products later, where the caller puts them (they read the operands
            # the chain parked in the workspace: zeggs_style_encoder_bwd_part)
            _check(L.zeggs_style_encoder_bwd_part(C.byref(ctx.d), C.byref(P), _p(dout), C.byref(G), _p(ctx.ws),
                                                  C.c_size_t(ctx.ws.numel()), _stream(), zeroed, 1), "style_encoder_bwd (chain)")
            ev = torch.cuda.Event()
            ev.record()
            d, ws, keep = ctx.d, ctx.ws, (params, grads, dout)

            def products():
                _route(ectx)
                _check(L.zeggs_style_encoder_bwd_part(C.byref(d), C.byref(P), _p(keep[2]), C.byref(G), _p(ws),
                                                      C.c_size_t(ws.numel()), _stream(), zeroed, 2), "style_encoder_bwd (products)")
            ectx.deferred_wgrads.append((ev, products))
            return (None, None, None, None, None, *rets)
        _check(L.zeggs_style_encoder_bwd_ex(C.byref(ctx.d), C.byref(P), _p(dout), C.byref(G), _p(ctx.ws),
                                            C.c_size_t(ctx.ws.numel()), _stream(), zeroed), "style_encoder_bwd")
        return (None, None, None, None, None, *rets)


STYLE_GRU_FIELDS = ("c0_w", "c0_b", "c2_w", "c2_b", "w_ih", "w_hh", "b_ih", "b_hh", "w_ih_r", "w_hh_r", "b_ih_r",
                    "b_hh_r", "p_w", "p_b")
StyleGruPtrs = _ptr_struct("StyleGruPtrs", STYLE_GRU_FIELDS)


class StyleGruDims(C.Structure):       # mirrors ZeggsStyleGruDims
    _fields_ = [("B", C.c_int), ("L", C.c_int), ("C", C.c_int), ("H", C.c_int), ("O", C.c_int)]


class _StyleGruFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, *params):
        x = _f32c(x)
        ctx.ectx = current()
        _route(ctx.ectx)
        ctx.orig = params
        params = [_f32c(t) for t in params]
        B, Lx, Cx = x.shape
        d = StyleGruDims(B, Lx, Cx, params[0].shape[0], params[12].shape[0])
        L = lib()
        L.zeggs_style_encoder_gru_workspace_bytes.restype = C.c_size_t
        ws = _ws(L.zeggs_style_encoder_gru_workspace_bytes(C.byref(d)), x.device)
        out = torch.empty(B, d.O, device=x.device, dtype=torch.float32)
        P = _ptrs(StyleGruPtrs, STYLE_GRU_FIELDS, params)
        _check(L.zeggs_style_encoder_gru_fwd(C.byref(d), C.byref(P), _p(x), _p(out), _p(ws), C.c_size_t(ws.numel()),
                                             _stream()), "style_encoder_gru_fwd")
        ctx.d, ctx.ws = d, ws
        ctx.save_for_backward(*params)
        return out

    @staticmethod
    def backward(ctx, dout):
        params = list(ctx.saved_tensors)
        _route(ctx.ectx)
        grads, rets = _grad_targets(ctx.orig, params, ctx.ectx)
        P = _ptrs(StyleGruPtrs, STYLE_GRU_FIELDS, params)
        G = _ptrs(StyleGruPtrs, STYLE_GRU_FIELDS, grads)
        _check(lib().zeggs_style_encoder_gru_bwd(C.byref(ctx.d), C.byref(P), _p(_f32c(dout)), C.byref(G), _p(ctx.ws),
                                                 C.c_size_t(ctx.ws.numel()), _stream()), "style_encoder_gru_bwd")
        return (None, *rets)


def style_encoder_gru(x, enc):
    """StyleEncoderGRU.forward (reference modules.py:339-343) on the HIP engine"""
    g, pl = enc.rnn_layer, enc.projection_layer.linear_layer
    return _StyleGruFn.apply(x, enc.convs[0].conv.weight, enc.convs[0].conv.bias, enc.convs[2].conv.weight,
                             enc.convs[2].conv.bias, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0,
                             g.weight_ih_l0_reverse, g.weight_hh_l0_reverse, g.bias_ih_l0_reverse, g.bias_hh_l0_reverse,
                             pl.weight, pl.bias)


def style_input_buffer(enc, B, L, Cx, training, device, ws=None):
    """A workspace of the attention style encoder for a [B, L, Cx] example whose padded-input region the CALLER fills
    (gather_example): -> (ws, xp [B, L + 2, Cx] view of it).  `xp[:, 1:-1]` carries the workspace with it (attribute
    _zeggs_style_ws); style_encoder_attn() on that tensor runs in this workspace without its padding copy.  `ws`: a buffer of an
    earlier call to use again when it still fits."""
    params = style_param_list(enc)
    d = StyleDims(int(B), int(L), int(Cx), params[0].shape[0], params[4].shape[0], 4, 1 if training else 0, 0)
    Lb = lib()
    need = int(Lb.zeggs_style_encoder_workspace_bytes(C.byref(d)))
    if ws is None or ws.numel() != need or ws.device != torch.device(device):
        ws = _ws(need, device)
    off = int(Lb.zeggs_style_encoder_input_offset(C.byref(d)))
    xp = ws[off:off + 4 * B * (L + 2) * Cx].view(torch.float32).view(B, L + 2, Cx)
    return ws, xp


def example_view(ws, xp):
    """The [B, L, C] example inside the padded buffer `xp` of style_input_buffer(), tagged with its workspace."""
    ex = xp[:, 1:-1]
    ex._zeggs_style_ws = ws
    return ex


def gather_example(frames, rows, mean, std, out, pad=1):
    """zeggs_gather_example: frames [N, W], rows int64 [B, L], mean / std [Wo] -> out [B, L + 2 pad, Wo] (contiguous):
    normalised rows with the columns W.. zero BEFORE the normalisation (the example's empty gaze slot, reference dataset.py:194),
    edge rows zero."""
    B, Lr = rows.shape
    Wo = out.shape[2]
    assert out.shape == (B, Lr + 2 * pad, Wo) and out.is_contiguous() and mean.numel() == Wo and std.numel() == Wo
    _check(lib().zeggs_gather_example(_p(frames), frames.shape[1], _p(rows.contiguous()), int(B), int(Lr), _p(_f32c(mean)),
                                      _p(_f32c(std)), _p(out), int(Wo), int(pad), _stream()), "gather_example")
    return out


def style_encoder_attn(x, enc, training):
    pos = positional_table(x.shape[1], enc.convs[6].normalized_shape[0], x.device)
    return _StyleFn.apply(x, pos, 1 if training else 0, next_seed() if training else 0, 4, *style_param_list(enc))


class _VaeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, eps, temperature, S):
        enc, eps = _f32c(enc), _f32c(eps)
        B = enc.shape[0]
        z, mu, lv = (torch.empty(B, S, device=enc.device, dtype=torch.float32) for _ in range(3))
        _check(lib().zeggs_vae_reparam_fwd(_p(enc), _p(eps), _p(z), _p(mu), _p(lv), B, S, C.c_float(temperature),
                                           _stream()), "vae_reparam_fwd")
        ctx.save_for_backward(enc, eps)
        ctx.t, ctx.S = temperature, S
        ctx.set_materialize_grads(False)
        return z, mu, lv

    @staticmethod
    def backward(ctx, dz, dmu, dlv):
        enc, eps = ctx.saved_tensors
        denc = torch.empty_like(enc)
        c = lambda g: _p(_f32c(g)) if g is not None else None  # noqa: E731
        _check(lib().zeggs_vae_reparam_bwd(_p(enc), _p(eps), c(dz), c(dmu), c(dlv), _p(denc), enc.shape[0], ctx.S,
                                           C.c_float(ctx.t), _stream()), "vae_reparam_bwd")
        return denc, None, None, None


def vae_reparam(enc, eps, temperature, S):
    """-> z, mu, logvar (contiguous [B,S] each; mu / logvar are the two halves of enc)"""
    return _VaeFn.apply(enc, eps, float(temperature), S)


# ----------------------------------------------------------------------------- decoder
def decoder_param_list(dec):
    r, c = dec.recurrent_decoder, dec.cell_state_encoder
    g = r.layer1
    ps = [r.layer0.weight, r.layer0.bias, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0,
          g.weight_ih_l1, g.weight_hh_l1, g.bias_ih_l1, g.bias_hh_l1, r.layer2.weight, r.layer2.bias,
          c.layer0.weight, c.layer0.bias, c.layer1.weight, c.layer1.bias, c.layer2.weight, c.layer2.bias]
    if hasattr(r, "gammas_predictor"):      # RecurrentDecoderFiLM
        ps += [r.layer3.weight, r.layer3.bias, r.gammas_predictor.linear_layer.weight,
               r.gammas_predictor.linear_layer.bias, r.betas_predictor.linear_layer.weight,
               r.betas_predictor.linear_layer.bias]
    return ps


def _drop_prepared(ectx):
    """Forget a preparation nobody picked up (an eval / no_grad call, another batch size, an exception in between): the
    prepare kernels may still be WRITING the workspace on the side stream, and the block goes back to the current stream's
    pool -- so the current stream waits for them first."""
    prep, ectx.prepared = ectx.prepared, None
    if prep is not None:
        torch.cuda.current_stream().wait_event(prep[2])


def _versions(params):
    """autograd version counters of the weights a preparation was made from: an in-place torch operation on them in between (a
    loaded checkpoint, a replayed snapshot, a hand edit) makes the forward ignore the packs and make its own.  (The engine's
    own optimizer kernels write through raw pointers and do not count -- they run BEFORE the preparation.)"""
    return tuple(int(t._version) for t in params)


def decoder_prepare(dec, B, T, SP, ST, in_mean, in_std, out_mean, out_std, dt, stream, after=None):
    """Weight-only preparation of the NEXT training-mode decoder_core call of THIS context (ops.use) with these dimensions
    (zeggs_decoder_prepare) on `stream`, beside whatever the current stream does meanwhile (the encoders' forward); that
    call picks the prepared workspace up and waits for it.  The weights must not change in between."""
    ectx = current()
    _route(ectx)
    _drop_prepared(ectx)
    params = [_f32c(t) for t in decoder_param_list(dec)]
    stats = [_f32c(t) for t in (in_mean, in_std, out_mean, out_std)]
    PO, H = int(stats[2].numel()), dec.recurrent_decoder.layer1.hidden_size
    d = DecDims(int(B), int(T), PO + 3, PO, int(SP), int(ST), H, float(dt), 1 if len(params) == len(DEC_FIELDS) else 0)
    L = lib()
    ws = _ws(L.zeggs_decoder_workspace_bytes(C.byref(d), 1), params[0].device)
    P = _ptrs(DecPtrs, DEC_FIELDS, params)
    S = _ptrs(DecStats, ("in_mean", "in_std", "out_mean", "out_std"), stats)
    stream.wait_stream(torch.cuda.current_stream())       # the optimizer step that produced these weights
    # (no record_stream: the workspace belongs to the current stream's pool and its next user there -- the forward that picks
    #  it up, or whoever gets the block after it -- comes after a wait for `stream`; a recorded stream would park the block
    #  in the allocator's pending list, and every device synchronisation would then cost a few iterations of re-allocation)
    with torch.cuda.stream(stream):
        mask = L.zeggs_decoder_prepare(C.byref(d), C.byref(P), C.byref(S), _p(ws), C.c_size_t(ws.numel()),
                                       C.c_void_p(stream.cuda_stream))
        _check(min(mask, 0), "decoder_prepare")
        ev = torch.cuda.Event()
        ev.record(stream)
        if after is not None:       # (more work of the caller for `stream`, behind the packs and outside what the forward waits for)
            after()
    if mask > 0:
        key = (d.B, d.T, d.PI, d.PO, d.SP, d.ST, d.H, d.film, tuple(t.data_ptr() for t in params), _versions(params))
        ectx.prepared = (key, ws, ev, int(mask))
    return int(mask)


def _warn_gave_up(bits, what):
    import warnings
    warnings.warn("zeggs: a persistent kernel gave up (" + ", ".join(n for b, n in GAVE_UP.items() if bits & b) +
                  "; another tenant on the GPU?); it is disabled for this process and " + what +
                  " is redone on the stage kernels")


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose0, rpos0, rrot0, gaze, speech, style, in_mean, in_std, out_mean, out_std, dt, H, grad_mode,
                *params):
        pose0, rpos0, rrot0, gaze, speech, style = (_f32c(t) for t in (pose0, rpos0, rrot0, gaze, speech, style))
        stats = [_f32c(t) for t in (in_mean, in_std, out_mean, out_std)]
        ectx = ctx.ectx = current()
        _route(ctx.ectx)
        ctx.orig = params
        params = [_f32c(t) for t in params]
        B, T, SP = speech.shape
        ST, PO = style.shape[2], pose0.shape[1]
        # needs_input_grad ignores torch.no_grad(): the caller's grad mode selects the BPTT workspace / ring path
        training = bool(grad_mode) and any(ctx.needs_input_grad)
        d = DecDims(B, T, PO + 3, PO, SP, ST, H, float(dt), 1 if len(params) == len(DEC_FIELDS) else 0)
        L = lib()
        dev = pose0.device
        prep = ectx.prepared
        mask = 0
        if prep is not None and training and prep[0] == (d.B, d.T, d.PI, d.PO, d.SP, d.ST, d.H, d.film,
                                                         tuple(t.data_ptr() for t in params), _versions(params)):
            ectx.prepared = None
            _, ws, ev, mask = prep                          # packs of this step's weights, made on a second stream
            ectx.prepared_hits += 1
            torch.cuda.current_stream().wait_event(ev)
        else:
            _drop_prepared(ectx)                            # (waits for the side stream before the block is released)
            ws = _ws(L.zeggs_decoder_workspace_bytes(C.byref(d), int(training)), pose0.device)
        pose = torch.empty(B, T, PO, device=dev, dtype=torch.float32)
        rpos = torch.empty(B, T, 3, device=dev, dtype=torch.float32)
        rrot = torch.empty(B, T, 4, device=dev, dtype=torch.float32)
        P = _ptrs(DecPtrs, DEC_FIELDS, params)
        S = _ptrs(DecStats, ("in_mean", "in_std", "out_mean", "out_std"), stats)
        capturing = torch.cuda.is_current_stream_capturing()
        # Give-up detection (ZeggsDecCall.status).  An engine passes ITS status words and looks at them itself (device-guarded
        # optimizer step, lagged read-back: engine.TrainEngine).  Every other caller gets a per-thread word that is inspected
        # right here, where a validated persistent kernel may have run: the outputs are consumed next (BVH export, sample
        # rendering, the reference's own loss / optimizer) and nothing else would notice.  No read-back -- and no host
        # synchronisation -- when no persistent kernel can have been involved, or inside a stream capture (there the NaN
        # poisoning of the outputs is what is left).
        own = None
        if training:
            status = ectx.status
            if status is None and not capturing and (_persistent_live(1) or _persistent_live(2)):
                status = own = _plain_status(dev)
        else:
            status = None
            if not capturing and B == 1 and _persistent_live(0):
                status = own = _plain_status(dev)
        call = DecCall(int(mask) & 1, 0, None, status.data_ptr() if status is not None else None, 0)

        def run():
            _check(L.zeggs_decoder_fwd_ex(C.byref(d), C.byref(P), C.byref(S), _p(pose0), _p(rpos0), _p(rrot0), _p(gaze),
                                          _p(speech), _p(style), _p(pose), _p(rpos), _p(rrot), int(training), _p(ws),
                                          C.c_size_t(ws.numel()), _stream(), C.byref(call)), "decoder_fwd")
        run()
        if own is not None:
            bits = int(own[0].item())                      # one 4-byte read-back per rollout
            if bits:
                _warn_gave_up(bits, "the rollout")
                for name in ("persistent", "train_persistent", "bwd_persistent"):
                    set_option(name, 0)
                fill_(own.view(torch.float32))
                call.prepared = 0
                run()
        global _LAST_DECODER_WS
        if _CHAIN_DIAGNOSTICS:          # (diagnostic of the off-by-default "chain" option only: pins the workspace)
            _LAST_DECODER_WS = (d, int(training), ws)
        if training:
            ctx.d, ctx.ws = d, ws
            ctx.status, ctx.own_status = status, own is not None
            ctx.bwd_prepared = bool(mask & 2) and call.prepared != 0
            ctx.save_for_backward(gaze, pose, rpos, rrot, *stats, *params)
            ctx.set_materialize_grads(False)      # missing output gradients are zero-filled by our own kernel in backward
        return pose, rpos, rrot

    @staticmethod
    def backward(ctx, dpose, drpos, drrot):
        gaze, pose, rpos, rrot, *rest = ctx.saved_tensors
        stats, params = rest[:4], rest[4:]
        d = ctx.d
        ectx = ctx.ectx
        _route(ectx)
        L = lib()
        dev = pose.device
        z = lambda g, ref: fill_(torch.empty_like(ref)) if g is None else _f32c(g)  # noqa: E731
        dpose, drpos, drrot = z(dpose, pose), z(drpos, rpos), z(drrot, rrot)
        grads, rets = _grad_targets(ctx.orig, params, ectx)
        dspeech = torch.empty(d.B, d.T, d.SP, device=dev, dtype=torch.float32)
        dstyle = torch.empty(d.B, d.T, d.ST, device=dev, dtype=torch.float32)
        P = _ptrs(DecPtrs, DEC_FIELDS, params)
        G = _ptrs(DecPtrs, DEC_FIELDS, grads)
        S = _ptrs(DecStats, ("in_mean", "in_std", "out_mean", "out_std"), stats)
        direct = ectx.direct_grads and all(r is None for r in rets)
        side = ectx.wgrad_stream if direct and not torch.cuda.is_current_stream_capturing() else None
        # with a gradient exchange waiting (engine hook) the deferred GEMMs come in two halves of the parameter order, so that
        # the all-reduce of one runs underneath the GEMMs of the other
        after = ectx.after_decoder_backward
        chunked = side is not None and after is not None
        call = DecCall(2 if ctx.bwd_prepared else 0, 0 if side is None else (2 if chunked else 1),
                       side.cuda_stream if side is not None else None,
                       ctx.status.data_ptr() if ctx.status is not None else None, int(direct))

        def run():
            _check(L.zeggs_decoder_bwd_ex(C.byref(d), C.byref(P), C.byref(S), _p(gaze), _p(pose), _p(rpos), _p(rrot),
                                          _p(dpose), _p(drpos), _p(drrot), C.byref(G), _p(dspeech), _p(dstyle), _p(ctx.ws),
                                          C.c_size_t(ctx.ws.numel()), _stream(), C.byref(call)), "decoder_bwd")
        run()
        if ctx.own_status and not torch.cuda.is_current_stream_capturing():
            # plain autograd use (no engine looking at the word later): the gradients go to the caller's optimizer next
            bits = int(ctx.status[0].item())
            if bits:
                _warn_gave_up(bits, "the BPTT sweep")
                for name in ("train_persistent", "bwd_persistent"):
                    set_option(name, 0)
                fill_(ctx.status.view(torch.float32))
                call.prepared = 0
                run()                   # the forward's saved activations are intact: only the sweep is repeated
        if side is not None:
            # the library has put the recurrent layers' weight-gradient GEMMs on its second stream (they read only what the
            # sweep left in the workspace), beside the CellStateEncoder / encoder backward on this one
            ectx.wgrad_keepalive.append(ctx.ws)      # read by `side` until the caller joins it (release_wgrad_workspaces)
            if chunked:
                side.wait_stream(torch.cuda.current_stream())       # the CellStateEncoder gradients come from this stream
                with torch.cuda.stream(side):
                    after(0)         # layer2, GRU layer 1, CellStateEncoder: final behind the GEMMs already on `side`
                    _check(L.zeggs_decoder_wgrads(C.byref(d), C.byref(G), _p(ctx.ws), C.c_size_t(ctx.ws.numel()), 4 | 8,
                                                  C.c_void_p(side.cuda_stream)), "decoder_wgrads")
                    after(1)         # layer0, GRU layer 0
            elif ectx.decoder_grads_final is not None:
                side.wait_stream(torch.cuda.current_stream())       # the CellStateEncoder gradients come from this stream
                with torch.cuda.stream(side):
                    ectx.decoder_grads_final()
        elif after is not None and direct:
            after(None)
        return (None, None, None, None, dspeech, dstyle, None, None, None, None, None, None, None, *rets)


_LAST_DECODER_WS = None
_CHAIN_DIAGNOSTICS = False      # set by set_option("chain", 1): keep the last decoder workspace for last_decoder_chain_errors()


def last_decoder_chain_errors():
    """Error word of the chained (run-ahead) stage launches of the most recent decoder rollout (option "chain"):
    0 = every device-side hand-off wait was satisfied.  Synchronises the device."""
    if _LAST_DECODER_WS is None:
        return 0
    d, training, ws = _LAST_DECODER_WS
    out = C.c_int(0)
    _check(lib().zeggs_decoder_chain_errors(C.byref(d), training, _p(ws), C.c_size_t(ws.numel()), C.byref(out)),
           "decoder_chain_errors")
    return int(out.value)


def decoder_core(dec, pose0, rpos0, rrot0, gaze, speech, style, in_mean, in_std, out_mean, out_std, dt):
    """-> pose [B,T,PO] (de-normalised output vectors), rpos [B,T,3], rrot [B,T,4]"""
    H = dec.recurrent_decoder.layer1.hidden_size
    return _DecoderFn.apply(pose0, rpos0, rrot0, gaze, speech, style, in_mean, in_std, out_mean, out_std, dt, H,
                            torch.is_grad_enabled(), *decoder_param_list(dec))


def decoder_rollout(dec, Z_root_pos, Z_root_rot, Z_root_vel, Z_root_vrt, Z_lpos, Z_ltxy, Z_lvel, Z_lvrt, gaze,
                    speech, style, in_mean, in_std, out_mean, out_std, dt):
    """Reference-shaped Decoder.forward: 8 tensors [B,T,...] (views of the engine's pose rows)."""
    B, J = Z_lpos.shape[0], Z_lpos.shape[1]
    pose0 = torch.cat([Z_root_vel.reshape(B, -1), Z_root_vrt.reshape(B, -1), Z_lpos.reshape(B, -1),
                       Z_ltxy.reshape(B, -1), Z_lvel.reshape(B, -1), Z_lvrt.reshape(B, -1)], dim=1)
    pose, rpos, rrot = decoder_core(dec, pose0, Z_root_pos, Z_root_rot, gaze, speech, style, in_mean, in_std,
                                    out_mean, out_std, dt)
    T = pose.shape[1]
    return (rpos, rrot, pose[..., 0:3], pose[..., 3:6], pose[..., 6:6 + 3 * J].reshape(B, T, J, 3),
            pose[..., 6 + 3 * J:6 + 9 * J].reshape(B, T, J, 2, 3),
            pose[..., 6 + 9 * J:6 + 12 * J].reshape(B, T, J, 3),
            pose[..., 6 + 12 * J:6 + 15 * J].reshape(B, T, J, 3))


def decoder_chunk(dec, pose0, rpos0, rrot0, gaze, speech, style, in_mean, in_std, out_mean, out_std, dt, h_in=None,
                  status=None):
    """Inference rollout resumed from a given state (zeggs_decoder_fwd_state_ex): frame 0 of the chunk is the last frame already
    produced (pose0 / rpos0 / rrot0; index 0 of gaze / speech / style belongs to it), h_in [2,B,H] the GRU state after it (None:
    first chunk, CellStateEncoder).  -> pose [B,N,PO], rpos, rrot, h_out [2,B,H].  Nothing is read back: `status` (device
    words, ops.new_status) collects a give-up bit of the persistent kernel for the CALLER to look at when it chooses to."""
    pose0, rpos0, rrot0, gaze, speech, style = (_f32c(t) for t in (pose0, rpos0, rrot0, gaze, speech, style))
    stats = [_f32c(t) for t in (in_mean, in_std, out_mean, out_std)]
    params = [_f32c(t) for t in decoder_param_list(dec)]
    B, N, SP = speech.shape
    PO, H = pose0.shape[1], dec.recurrent_decoder.layer1.hidden_size
    d = DecDims(B, N, PO + 3, PO, SP, style.shape[2], H, float(dt), 1 if len(params) == len(DEC_FIELDS) else 0)
    L = lib()
    dev = pose0.device
    ws = _ws(L.zeggs_decoder_workspace_bytes(C.byref(d), 0), dev)
    pose = torch.empty(B, N, PO, device=dev, dtype=torch.float32)
    rpos = torch.empty(B, N, 3, device=dev, dtype=torch.float32)
    rrot = torch.empty(B, N, 4, device=dev, dtype=torch.float32)
    h_out = torch.empty(2, B, H, device=dev, dtype=torch.float32)
    P = _ptrs(DecPtrs, DEC_FIELDS, params)
    S = _ptrs(DecStats, ("in_mean", "in_std", "out_mean", "out_std"), stats)
    call = DecCall(0, 0, None, status.data_ptr() if status is not None else None, 0)
    _check(L.zeggs_decoder_fwd_state_ex(C.byref(d), C.byref(P), C.byref(S), _p(pose0), _p(rpos0), _p(rrot0), _p(gaze),
                                        _p(speech), _p(style), _p(pose), _p(rpos), _p(rrot),
                                        _p(_f32c(h_in)) if h_in is not None else None, _p(h_out), _p(ws),
                                        C.c_size_t(ws.numel()), _stream(), C.byref(call)), "decoder_fwd_state_ex")
    return pose, rpos, rrot, h_out


# ----------------------------------------------------------------------------- free functions of the Networks layer
# (reference ZEGGS/modules.py:673-813; csrc/funcs.hip).  Differentiable: the reference's own training loop (INTEGRATION.md
# route 2) calls normalize / compute_KL_div inside its inline loss and back-propagates through them.
class _NormalizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        x = _f32c(x)
        W = x.shape[-1]
        y = torch.empty_like(x)
        _check(lib().zeggs_normalize_vec_fwd(_p(x), _p(y), C.c_long(x.numel() // max(W, 1)), W, C.c_float(eps), _stream()),
               "normalize_vec_fwd")
        ctx.save_for_backward(x)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        W = x.shape[-1]
        dx = torch.empty_like(x)
        _check(lib().zeggs_normalize_vec_bwd(_p(x), _p(_f32c(dy)), _p(dx), C.c_long(x.numel() // max(W, 1)), W,
                                             C.c_float(ctx.eps), _stream()), "normalize_vec_bwd")
        return dx, None


def normalize_vec(x, eps=1e-8):
    """x / (||x||_2 + eps) over the last dimension (reference modules.py:673-675)"""
    return _NormalizeFn.apply(x, float(eps))


class _VectorizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, root_pos, root_rot, root_vel, root_vrt, lpos, ltxy, lvel, lvrt, gaze_pos, in_mean, in_std):
        a = [_f32c(t) for t in (root_pos, root_rot, root_vel, root_vrt, lpos, ltxy, lvel, lvrt, gaze_pos)]
        in_mean, in_std = _f32c(in_mean), _f32c(in_std)
        B, J = a[4].shape[0], a[4].shape[1]
        out = torch.empty(B, 9 + 15 * J, device=a[0].device, dtype=torch.float32)
        _check(lib().zeggs_vectorize_input_fwd(B, J, *[_p(t) for t in a], _p(in_mean), _p(in_std), _p(out), _stream()),
               "vectorize_input_fwd")
        ctx.save_for_backward(a[0], a[1], a[8], in_std)
        ctx.shapes = [t.shape for t in a]
        ctx.dims = (B, J)
        return out

    @staticmethod
    def backward(ctx, dout):
        root_pos, root_rot, gaze_pos, in_std = ctx.saved_tensors
        B, J = ctx.dims
        g = [torch.empty(sh, device=root_pos.device, dtype=torch.float32) for sh in ctx.shapes]
        _check(lib().zeggs_vectorize_input_bwd(B, J, _p(root_pos), _p(root_rot), _p(gaze_pos), _p(in_std), _p(_f32c(dout)),
                                               *[_p(t) for t in g], _stream()), "vectorize_input_bwd")
        return (*g, None, None)


def vectorize_input(root_pos, root_rot, root_vel, root_vrt, lpos, ltxy, lvel, lvrt, gaze_pos, in_mean, in_std):
    """-> [B, 9 + 15 J] normalised autoregressive input (reference modules.py:677-713)"""
    return _VectorizeFn.apply(root_pos, root_rot, root_vel, root_vrt, lpos, ltxy, lvel, lvrt, gaze_pos, in_mean, in_std)


class _DevectorizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, predicted, root_pos, root_rot, J, dt, out_mean, out_std):
        predicted, root_pos, root_rot, out_mean, out_std = (_f32c(t) for t in (predicted, root_pos, root_rot, out_mean,
                                                                                 out_std))
        B = predicted.shape[0]
        dev = predicted.device
        pose = torch.empty(B, 6 + 15 * J, device=dev, dtype=torch.float32)
        nrp = torch.empty(B, 3, device=dev, dtype=torch.float32)
        nrr = torch.empty(B, 4, device=dev, dtype=torch.float32)
        _check(lib().zeggs_devectorize_output_fwd(B, J, C.c_float(dt), _p(predicted), _p(root_pos), _p(root_rot),
                                                  _p(out_mean), _p(out_std), _p(pose), _p(nrp), _p(nrr), _stream()),
               "devectorize_output_fwd")
        ctx.save_for_backward(predicted, root_pos, root_rot, out_mean, out_std)
        ctx.dims = (B, J, dt)
        ctx.set_materialize_grads(False)
        return pose, nrp, nrr

    @staticmethod
    def backward(ctx, dpose, dnrp, dnrr):
        predicted, root_pos, root_rot, out_mean, out_std = ctx.saved_tensors
        B, J, dt = ctx.dims
        c = lambda g: _p(_f32c(g)) if g is not None else None  # noqa: E731
        dpred, drp, drr = torch.empty_like(predicted), torch.empty_like(root_pos), torch.empty_like(root_rot)
        _check(lib().zeggs_devectorize_output_bwd(B, J, C.c_float(dt), _p(predicted), _p(root_pos), _p(root_rot),
                                                  _p(out_mean), _p(out_std), c(dpose), c(dnrp), c(dnrr), _p(dpred), _p(drp),
                                                  _p(drr), _stream()), "devectorize_output_bwd")
        return dpred, drp, drr, None, None, None, None


def devectorize_output(predicted, root_pos, root_rot, njoints, dt, out_mean, out_std):
    """-> pose [B, 6 + 15 J] (de-normalised output vector), new root_pos [B,3], new root_rot [B,4] (reference modules.py:716-742)"""
    return _DevectorizeFn.apply(predicted, root_pos, root_rot, int(njoints), float(dt), out_mean, out_std)


class _KLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = _f32c(mu), _f32c(logvar)
        out = torch.empty((), device=mu.device, dtype=torch.float32)
        _check(lib().zeggs_kl_div_fwd(_p(mu), _p(logvar), mu.shape[0], mu.numel() // mu.shape[0], _p(out), _stream()),
               "kl_div_fwd")
        ctx.save_for_backward(mu, logvar)
        return out

    @staticmethod
    def backward(ctx, dout):
        mu, logvar = ctx.saved_tensors
        dmu, dlv = torch.empty_like(mu), torch.empty_like(logvar)
        _check(lib().zeggs_kl_div_bwd(_p(mu), _p(logvar), mu.shape[0], mu.numel() // mu.shape[0], _p(_f32c(dout).reshape(1)),
                                      _p(dmu), _p(dlv), _stream()), "kl_div_bwd")
        return dmu, dlv


def kl_div(mu, logvar):
    """mean_b(-0.5 mean_s(1 + logvar - mu^2 - exp(logvar))) as a device scalar (reference modules.py:778-779)"""
    return _KLFn.apply(mu, logvar)


def mask_from_lengths(lengths):
    """bool [B, max(lengths)]: position < length (reference modules.py:802-813; like the reference's `torch.arange(0,
    max_len)` on a device scalar, sizing the result reads max(lengths) back to the host)"""
    lengths = lengths.to(torch.int64).contiguous()
    B = lengths.shape[0]
    max_len = int(lengths.max().item())
    mask = torch.empty(B, max_len, device=lengths.device, dtype=torch.uint8)
    _check(lib().zeggs_mask_from_lengths(_p(lengths), B, max_len, _p(mask), _stream()), "mask_from_lengths")
    return mask.view(torch.bool)


# ----------------------------------------------------------------------------- loss
class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, o_pose, o_rpos, o_rrot, mu, logvar, w_pose, w_rpos, w_rrot, gaze, parents, kl_weight, dt, gscale,
                unit_grad, truth_ws):
        o_pose, o_rpos, o_rrot, w_pose, w_rpos, w_rrot, gaze = (
            _f32c(t) for t in (o_pose, o_rpos, o_rrot, w_pose, w_rpos, w_rrot, gaze))
        B, T, PO = o_pose.shape
        J = (PO - 6) // 15
        has_kl = mu is not None and kl_weight > 0
        S = mu.shape[1] if mu is not None else 0
        d = LossDims(B, T, J, S, float(dt))
        L = lib()
        dev = o_pose.device
        need = L.zeggs_loss_workspace_bytes(C.byref(d))
        prepared = truth_ws is not None and truth_ws.numel() >= need       # (loss_prepare_truth filled its truth half)
        ws = truth_ws if prepared else _ws(need, dev)
        terms = torch.empty(19, device=dev, dtype=torch.float32)
        dpose, drpos, drrot = torch.empty_like(o_pose), torch.empty_like(o_rpos), torch.empty_like(o_rrot)
        dmu = torch.empty(B, S, device=dev) if mu is not None else None
        dlv = torch.empty(B, S, device=dev) if mu is not None else None
        _check(L.zeggs_loss_fwd_bwd_ex(C.byref(d), _p(parents), _p(o_pose), _p(o_rpos), _p(o_rrot), _p(w_pose),
                                       _p(w_rpos), _p(w_rrot), _p(gaze), _p(_f32c(mu)) if mu is not None else None,
                                       _p(_f32c(logvar)) if mu is not None else None,
                                       C.c_float(kl_weight if has_kl else 0.0), _p(terms), _p(dpose), _p(drpos),
                                       _p(drrot), _p(dmu), _p(dlv), C.c_float(gscale), _p(ws), C.c_size_t(ws.numel()),
                                       _stream(), int(prepared)), "loss_fwd_bwd")
        ctx.save_for_backward(dpose, drpos, drrot, dmu, dlv)
        ctx.unit_grad = bool(unit_grad)
        ctx.mark_non_differentiable(terms)
        ctx.set_materialize_grads(False)      # no zero-filled gradient for the (non-differentiable) terms vector
        loss = torch.empty((), device=dev, dtype=torch.float32)
        scale_copy(terms[18:19], out=loss)
        return loss, terms

    @staticmethod
    def backward(ctx, g, _gterms):
        dpose, drpos, drrot, dmu, dlv = ctx.saved_tensors
        if g is None:
            return (None,) * 15
        if not ctx.unit_grad:        # general case: scale by the upstream scalar (device side, no host sync)
            g = _f32c(g).reshape(1)
            dpose, drpos, drrot = scale_copy(dpose, g), scale_copy(drpos, g), scale_copy(drrot, g)
            if dmu is not None:
                dmu, dlv = scale_copy(dmu, g), scale_copy(dlv, g)
        return (dpose, drpos, drrot, dmu, dlv, None, None, None, None, None, None, None, None, None, None)


def loss_prepare_truth(w_pose, w_rpos, w_rrot, gaze, parents, dt, ws=None):
    """The ground-truth half of the loss's feature pass on the current stream (zeggs_loss_prepare_truth) -> the workspace to
    hand to training_loss(truth_ws=...) for the same batch.  `ws`: a workspace of an earlier call to use again when it fits."""
    w_pose, w_rpos, w_rrot, gaze = (_f32c(t) for t in (w_pose, w_rpos, w_rrot, gaze))
    B, T, PO = w_pose.shape
    d = LossDims(B, T, (PO - 6) // 15, 0, float(dt))
    L = lib()
    need = int(L.zeggs_loss_workspace_bytes(C.byref(d)))
    if ws is None or ws.numel() != need or ws.device != w_pose.device:
        ws = _ws(need, w_pose.device)
    _check(L.zeggs_loss_prepare_truth(C.byref(d), _p(parents), _p(w_pose), _p(w_rpos), _p(w_rrot), _p(gaze), _p(ws),
                                      C.c_size_t(ws.numel()), _stream()), "loss_prepare_truth")
    return ws


def training_loss(o_pose, o_rpos, o_rrot, w_pose, w_rpos, w_rrot, gaze, parents, dt, mu=None, logvar=None,
                  kl_weight=0.0, gscale=1.0, unit_grad=False, truth_ws=None):
    """-> (loss scalar tensor, terms[19]); terms[0:18] are the reference's weighted loss terms.
    unit_grad=True: the caller promises to call backward() on the returned loss itself (upstream gradient 1), so
    the gradients computed in the fused forward+backward kernel are handed on without a scaling pass."""
    return _LossFn.apply(o_pose, o_rpos, o_rrot, mu, logvar, w_pose, w_rpos, w_rrot, gaze, parents,
                         float(kl_weight), float(dt), float(gscale), bool(unit_grad), truth_ws)


# ----------------------------------------------------------------------------- optimizer / data
def radam_step(p, g, m, v, beta1, beta2, eps, step_scale, rectified, status=None, gflag=None, count=True, decay=0.0):
    """status (int32[STATUS_WORDS], device): the guarded step -- a no-op on the device, counted in status[1], when a
    persistent sweep of the iteration gave up here (status[0]) or on another rank (gflag, a device float).  A step applied in
    pieces (slices of the flat buffers) counts its skip in ONE of them: count=False for the others.
    decay = weight_decay * lr (reference optimizers.py:88-95), 0 when the step is not applied."""
    if decay != 0.0:
        _check(lib().zeggs_radam_step_wd(_p(p), _p(g), _p(m), _p(v), C.c_long(p.numel()), C.c_float(beta1), C.c_float(beta2),
                                         C.c_float(eps), C.c_float(step_scale), int(rectified), C.c_float(decay),
                                         C.c_void_p(status.data_ptr()) if status is not None else None,
                                         _p(gflag) if gflag is not None else None, int(bool(count)), _stream()), "radam_step_wd")
    elif status is None:
        _check(lib().zeggs_radam_step(_p(p), _p(g), _p(m), _p(v), C.c_long(p.numel()), C.c_float(beta1),
                                      C.c_float(beta2), C.c_float(eps), C.c_float(step_scale), int(rectified),
                                      _stream()), "radam_step")
    else:
        _check(lib().zeggs_radam_step_guarded_part(_p(p), _p(g), _p(m), _p(v), C.c_long(p.numel()), C.c_float(beta1),
                                                   C.c_float(beta2), C.c_float(eps), C.c_float(step_scale), int(rectified),
                                                   C.c_void_p(status.data_ptr()), _p(gflag) if gflag is not None else None,
                                                   int(bool(count)), _stream()), "radam_step_guarded")


def status_flag(status, dst):
    """dst[0] (device float) = 1.0 if status[0] has a give-up bit, else 0.0 -- the word that rides on the gradient all-reduce"""
    _check(lib().zeggs_status_flag(C.c_void_p(status.data_ptr()), _p(dst), _stream()), "status_flag")


def gather_windows(frames, starts, T, out=None):
    """frames [N, W] device, starts int64 [B] device -> [B, T, W] (into `out` when it has that shape)"""
    B, W = starts.shape[0], frames.shape[1]
    if out is None or out.shape != (B, T, W) or not out.is_contiguous():
        out = torch.empty(B, T, W, device=frames.device, dtype=torch.float32)
    _check(lib().zeggs_gather_windows(_p(frames), W, _p(starts), B, T, _p(out), _stream()), "gather_windows")
    return out


def gather_rows(frames, rows, out=None, out_ld=None):
    """frames [N, W], rows int64 [...]-> [..., W] (or into a strided `out`)"""
    W = frames.shape[1]
    n = rows.numel()
    if out is None:
        out = torch.empty(*rows.shape, W, device=frames.device, dtype=torch.float32)
        out_ld = W
    _check(lib().zeggs_gather_rows(_p(frames), W, _p(rows.contiguous()), C.c_long(n), _p(out), int(out_ld), _stream()),
           "gather_rows")
    return out


def normalize_rows_(x, mean, std):
    """in place (x - mean) / std over the last dim; std is a [W] tensor or a python float"""
    W = x.shape[-1]
    rows = x.numel() // W
    vec = std if torch.is_tensor(std) and std.dim() > 0 else None
    sc = 1.0 if vec is not None else float(std)
    _check(lib().zeggs_normalize_rows(_p(x), C.c_long(rows), W, C.c_long(W), _p(mean), _p(vec) if vec is not None else None,
                                      C.c_float(sc), _stream()), "normalize_rows")
    return x


_OPTIONS = {}            # switches set through set_option / ZEGGS_OPTIONS (the library's defaults otherwise)
_PERSISTENT_OPTION = ("persistent", "train_persistent", "bwd_persistent")


def _persistent_live(which):
    """True if persistent kernel `which` (0 B=1 decode, 1 training rollout, 2 BPTT sweep) is switched on and validated on this
    process -- only then can a call have ended in an undetected give-up (an unvalidated kernel is checked, with a stream
    synchronisation, by the library itself)."""
    return _OPTIONS.get(_PERSISTENT_OPTION[which], 1) != 0 and lib().zeggs_persistent_state(which) == 1


def set_option(name, value):
    """Runtime switches of the library (e.g. "decoder_fast": 1 packed stage kernels / 0 generic GEMM path)."""
    global _CHAIN_DIAGNOSTICS, _LAST_DECODER_WS
    _check(lib().zeggs_set_option(name.encode(), int(value)), "set_option")
    _OPTIONS[name] = int(value)
    if name == "chain":
        _CHAIN_DIAGNOSTICS = bool(value)
        if not value:
            _LAST_DECODER_WS = None
