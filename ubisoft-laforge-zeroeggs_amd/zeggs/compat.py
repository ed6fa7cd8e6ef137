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
