"""Checkpoint compatibility with the reference.

The reference saves WHOLE modules with torch.save (train.py:482-491), pickled under the top-level module name
`modules` (its scripts run with ZEGGS/ as cwd).  `alias_reference_modules()` maps that name to this package's
drop-in classes, so reference checkpoints un-pickle straight into the HIP engine; checkpoints written by this
package's train() are pickled as `zeggs.modules.*`, and `export_for_reference()` re-saves plain state_dicts for the
other direction.
"""
import sys

import torch


def alias_reference_modules():
    from . import modules, optimizers
    sys.modules.setdefault("modules", modules)
    sys.modules.setdefault("optimizers", optimizers)


def load_module(path, map_location=None):
    alias_reference_modules()
    return torch.load(path, map_location=map_location, weights_only=False)


def export_for_reference(module, path):
    torch.save(module.state_dict(), path)


# ----------------------------------------------------------------------------- pickle-free checkpoints
# Whole-module pickles (the reference's format) tie a checkpoint to importable class paths.  The portable form is
# `weights.safetensors` (every tensor of the three state_dicts under "<net>/<key>") + `arch.json` (constructor
# arguments); both formats are written by train(), either loads into the same modules.
def _arch_of(se, de, st):
    from . import modules
    rd = de.recurrent_decoder
    film = isinstance(rd, modules.RecurrentDecoderFiLM)
    H = rd.layer1.hidden_size
    PO = (rd.layer3 if film else rd.layer2).out_features
    ST = de.cell_state_encoder.layer0.in_features - (PO + 3)
    arch = {"speech_encoder": {"input_size": se.layer0.in_channels, "hidden_size": se.layer0.out_channels,
                               "output_size": se.layer2.out_features},
            "decoder": {"pose_input_size": PO + 3, "pose_output_size": PO, "speech_encoding_size": se.layer2.out_features,
                        "style_encoding_size": ST, "hidden_size": H, "num_rnn_layers": 2,
                        "rnn_cond": "film" if film else "normal"}}
    if st is not None:
        gru = isinstance(st.encoder, modules.StyleEncoderGRU)
        conv0 = st.encoder.convs[0].conv
        arch["style_encoder"] = {"input_size": conv0.in_channels, "hidden_size": conv0.out_channels,
                                 "style_embedding_size": st.style_embedding_size, "type": "gru" if gru else "attn",
                                 "use_vae": bool(st.use_vae)}
    return arch


def save_state(directory, se, de, st=None, meta=None):
    """write <directory>/weights.safetensors + arch.json"""
    import json
    from pathlib import Path
    from safetensors.torch import save_file
    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    tensors = {}
    for name, net in (("speech_encoder", se), ("decoder", de), ("style_encoder", st)):
        if net is not None:
            tensors.update({f"{name}/{k}": v.detach().cpu().contiguous() for k, v in net.state_dict().items()})
    save_file(tensors, str(directory / "weights.safetensors"))
    with open(directory / "arch.json", "w") as f:
        json.dump({"arch": _arch_of(se, de, st), "meta": meta or {}}, f, indent=1)


def load_state(directory, device=None):
    """-> (speech_encoder, decoder, style_encoder or None, meta) rebuilt from save_state() files"""
    import json
    from pathlib import Path
    from safetensors.torch import load_file
    from . import modules
    directory = Path(directory)
    with open(directory / "arch.json") as f:
        info = json.load(f)
    arch = info["arch"]
    tensors = load_file(str(directory / "weights.safetensors"))
    nets = {"speech_encoder": modules.SpeechEncoder(**arch["speech_encoder"]), "decoder": modules.Decoder(**arch["decoder"]),
            "style_encoder": modules.StyleEncoder(**arch["style_encoder"]) if "style_encoder" in arch else None}
    for name, net in nets.items():
        if net is not None:
            net.load_state_dict({k[len(name) + 1:]: v for k, v in tensors.items() if k.startswith(name + "/")})
            if device is not None:
                net.to(device)
    return nets["speech_encoder"], nets["decoder"], nets["style_encoder"], info.get("meta", {})
