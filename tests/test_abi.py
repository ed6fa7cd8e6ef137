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
