"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/zeggs_hip.h declares, answers host-only queries, and the product path refuses to run
without a GPU (no CPU fallback)."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

from zeggs import modules, ops, synth

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "zeggs_hip.h").read_text()
    names = set(re.findall(r"\b(zeggs_[a-z0-9_]+)\s*\(", header))
    assert len(names) >= 18
    L = ops.lib()
    for n in sorted(names):
        assert hasattr(L, n), f"libzeggs_hip.so does not export {n}"
    assert L.zeggs_version() >= 100


def test_workspace_queries_run_on_host():
    L = ops.lib()
    d = ops.DecDims(32, 256, synth.POSE_IN, synth.POSE_OUT, 64, 64, 1024, synth.DT)
    train, infer = L.zeggs_decoder_workspace_bytes(ctypes.byref(d), 1), L.zeggs_decoder_workspace_bytes(ctypes.byref(d), 0)
    assert train > infer > 0
    s = ops.StyleDims(32, 512, synth.POSE_IN, 512, 128, 4, 1, 0)
    assert L.zeggs_style_encoder_workspace_bytes(ctypes.byref(s)) > 0
    sp = ops.SpeechDims(32, 256, 81, 64, 64, 31, 0.2, 1)
    assert L.zeggs_speech_encoder_workspace_bytes(ctypes.byref(sp)) > 0
    ld = ops.LossDims(32, 256, 75, 64, synth.DT)
    assert L.zeggs_loss_workspace_bytes(ctypes.byref(ld)) > 0


def test_gemm_routing_is_per_thread_not_per_process():
    """zeggs_gemm_route: a caller's routing of the TN products (what TrainEngine wants for its three-queue tail) is the calling
    THREAD's; the process-wide options stay what they were for every other thread, and -1 falls back to them (host-only check)."""
    import threading
    L = ops.lib()
    out = (ctypes.c_int * 4)()
    L.zeggs_gemm_route_get(out)
    base = list(out)
    L.zeggs_gemm_route(1, 1, 8, 32)
    L.zeggs_gemm_route_get(out)
    assert list(out) == [1, 1, 8, 32]
    seen = []

    def other():
        o = (ctypes.c_int * 4)()
        L.zeggs_gemm_route_get(o)
        seen.append(list(o))
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert seen == [base]                       # another thread: untouched
    L.zeggs_gemm_route(-1, -1, 6, -1)
    L.zeggs_gemm_route_get(out)
    assert list(out) == [base[0], base[1], 6, base[3]]
    L.zeggs_gemm_route(-1, -1, -1, -1)
    L.zeggs_gemm_route_get(out)
    assert list(out) == base
    # the Python side: a context's route becomes the thread's at the entry of its calls, the default context resets it
    ctx = ops.EngineContext()
    ctx.gemm_route = (1, 2, 4, 16)
    ops._route(ctx)
    L.zeggs_gemm_route_get(out)
    assert list(out) == [1, 2, 4, 16]
    ops._route(ops._DEFAULT_CTX)
    L.zeggs_gemm_route_get(out)
    assert list(out) == base


def test_no_cpu_fallback():
    se = modules.SpeechEncoder(synth.N_AUDIO, 64, 64)
    with pytest.raises(RuntimeError, match="GPU"):
        se(torch.zeros(1, 4, synth.N_AUDIO))


def test_state_dict_keys_match_reference_layout():
    torch.manual_seed(0)
    de = modules.Decoder(synth.POSE_IN, synth.POSE_OUT, 64, 64, 1024, 2)
    st = modules.StyleEncoder(synth.POSE_IN, 512, 64, type="attn", use_vae=True)
    keys = set(de.state_dict())
    for k in ("recurrent_decoder.layer0.weight", "recurrent_decoder.layer1.weight_ih_l0",
              "recurrent_decoder.layer1.bias_hh_l1", "recurrent_decoder.layer2.bias",
              "cell_state_encoder.layer2.weight"):
        assert k in keys
    skeys = set(st.state_dict())
    for k in ("encoder.convs.0.conv.weight", "encoder.convs.2.weight", "encoder.convs.6.bias",
              "encoder.blocks.0.attention.multi_head_attention.in_proj_weight",
              "encoder.blocks.0.attention.multi_head_attention.out_proj.bias",
              "encoder.blocks.0.attention.layer_norm.weight", "encoder.blocks.0.feed_forward.convs.2.conv.weight",
              "encoder.blocks.0.feed_forward.layer_norm.bias"):
        assert k in skeys
    assert sum(p.numel() for p in de.parameters()) == 23301227
    assert sum(p.numel() for p in st.parameters()) == 2105472


def test_streaming_frame_accounting_is_consistent_host_only():
    """zeggs_mel_frames_ready (host integer rule of the streaming front-end): monotone in the sample count, never
    promises a frame whose STFT support reaches past the received samples, and reaches the offline frame count's
    neighbourhood as the signal grows (the tail is flushed by the final call)."""
    import ctypes as C
    import math
    from zeggs import audio, ops
    L = ops.lib()
    L.zeggs_mel_frames_ready.restype = C.c_long
    for flags in (0, 1):        # centered (the shipped configuration) / uncentered frames (audio_conf.centered = false)
        _frame_accounting(L, audio.MelDims(800, 200, 80, 16000, 60.0, 1e-5, 0.0, flags), flags)


def _frame_accounting(L, d, flags):
    import ctypes as C
    import math
    from zeggs import audio
    prev = 0
    for n in list(range(0, 3000, 37)) + [16000, 16001, 48000, 480000]:
        k = int(L.zeggs_mel_frames_ready(C.byref(d), C.c_long(n)))
        assert k >= prev or n < 3000 and k >= 0
        prev = max(prev, k)
        for kk in (k - 1,):
            if kk >= 0:
                hi = max(math.ceil((80.0 / 60.0) * kk), 1)          # last STFT frame that animation frame kk interpolates
                assert 200 * hi + (800 if flags & 1 else 400) <= n, (n, kk, hi)      # its window ends inside the received samples
        assert k <= audio.n_anim_frames(n) + 1
    assert int(L.zeggs_mel_frames_ready(C.byref(d), C.c_long(480000))) >= audio.n_anim_frames(480000) - 3
