"""Import the UNMODIFIED reference: /root/reference/ZEGGS in the build container, or -- where that does not exist
(the GPU box) -- the byte-for-byte snapshot oracle/build_ref.py left in the git-ignored oracle/_ref/, unpacked into a
temporary directory.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Only used by
oracle/make_golden*.py (fixture generation, build container only) and by bench.py's
cpu_baseline kind="reference" leg (oracle/ref_timing.py); nothing on the GPU box reads /root/reference.

The shims replace *missing third-party packages / binaries* only (tensorboard,
omegaconf, sox, ffmpeg check, scipy.signal.hann alias); no reference source is
modified or copied.  Recipe documented in SURVEY.md Appendix A.
"""
import importlib.util
import sys
import types
from pathlib import Path

REF = Path("/root/reference/ZEGGS")
SNAPSHOT = Path(__file__).resolve().parent / "_ref" / "zeggs_reference.tar.gz"
_root = None


def root():
    """Directory that holds ZEGGS/, configs/ and data/processed_v*/ of the reference (None: neither source exists)."""
    global _root
    if _root is None:
        if REF.is_dir():
            _root = REF.parent
        elif SNAPSHOT.exists():
            import atexit
            import shutil
            import tarfile
            import tempfile
            tmp = Path(tempfile.mkdtemp(prefix="zeggs_ref_snapshot_"))
            atexit.register(shutil.rmtree, str(tmp), ignore_errors=True)
            with tarfile.open(SNAPSHOT) as tar:
                tar.extractall(tmp)
            _root = tmp
    return _root


def available():
    return root() is not None


def source():
    return "unmodified /root/reference/ZEGGS" if REF.is_dir() else "oracle/_ref snapshot of the unmodified reference (oracle/build_ref.py)"


def load():
    """Returns a namespace with the reference modules (modules, train, ...)."""
    import scipy.signal as sps
    import torch

    if "zeggs_ref_loaded" in sys.modules:
        ns = sys.modules["zeggs_ref_loaded"]
        for k, v in ns.ref_entries.items():      # (re-)claim the top-level names, see release()
            _saved.setdefault(k, sys.modules.get(k))
            sys.modules[k] = v
        return ns
    # The reference's files import each other by TOP-LEVEL name (`from modules import ...`) and its checkpoints pickle
    # classes under those names; zeggs.compat.alias_reference_modules() may have bound `modules` / `optimizers` to the drop-in
    # classes in this process.  The reference takes the names while it runs; release() gives them back.
    for k in ("modules", "optimizers", "train", "generate", "data_pipeline", "dataset", "helpers", "utils", "anim"):
        if k in sys.modules:
            _saved[k] = sys.modules.pop(k)
    REF = root() / "ZEGGS"              # noqa: N806  (the live checkout or the unpacked snapshot)
    sys.path.insert(0, str(REF))
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = type("SummaryWriter", (), {
        m: (lambda self, *a, **k: None) for m in ("__init__", "add_scalar", "add_scalars", "add_hparams")})
    sys.modules["torch.utils.tensorboard"] = tb
    sys.modules.setdefault("sox", types.ModuleType("sox"))
    if not hasattr(sps, "hann"):
        sps.hann = sps.windows.hann
    pkg = types.ModuleType("audio")
    pkg.__path__ = [str(REF / "audio")]
    sys.modules["audio"] = pkg
    for n in ("logs", "signal_manipulation", "spectrograms", "audio_files"):
        spec = importlib.util.spec_from_file_location("audio." + n, REF / "audio" / f"{n}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules["audio." + n] = m
        spec.loader.exec_module(m)

    class DictConfig(dict):
        def __init__(self, d):
            super().__init__({k: DictConfig(v) if isinstance(v, dict) else v for k, v in d.items()})
        __getattr__ = dict.__getitem__

    oc = types.ModuleType("omegaconf")
    oc.DictConfig = DictConfig
    sys.modules["omegaconf"] = oc

    import warnings
    warnings.filterwarnings("ignore")
    import modules, train, data_pipeline, generate, optimizers, dataset, helpers  # noqa: E401
    from anim import tquat, txform, bvh, quat

    ns = types.ModuleType("zeggs_ref_loaded")
    ns.modules, ns.train, ns.data_pipeline, ns.generate = modules, train, data_pipeline, generate
    ns.optimizers, ns.dataset, ns.helpers = optimizers, dataset, helpers
    ns.tquat, ns.txform, ns.bvh, ns.quat = tquat, txform, bvh, quat
    ns.DictConfig = DictConfig
    # torch >= 2.6 defaults weights_only=True; the reference pickles whole modules
    _orig_load = torch.load

    def _load(*a, **k):
        k.setdefault("weights_only", False)
        return _orig_load(*a, **k)

    ns.torch_load = _load
    ns.ref_entries = {k: sys.modules[k] for k in ("modules", "optimizers", "train", "generate", "data_pipeline", "dataset",
                                                  "helpers") if k in sys.modules}
    sys.modules["zeggs_ref_loaded"] = ns
    return ns


def load_dropin(dropin_modules):
    """The reference's OWN `train.py` and `generate.py` (unmodified files, imported under private names) with the top-level
    name `modules` bound to `dropin_modules` while they are imported -- i.e. `from modules import Decoder, SpeechEncoder,
    StyleEncoder, compute_KL_div, normalize` (train.py:20-24) resolves to the drop-in, everything else (dataset, optimizers,
    anim, helpers, utils, data_pipeline) stays the reference's.  INTEGRATION.md route 2; used by
    tests/test_gpu_reference_side.py.  Returns a namespace with .train / .generate (modules) and .ref (load())."""
    ref = load()                         # third-party stubs, sys.path, the reference's helper modules
    ns = types.SimpleNamespace(ref=ref)
    mine = sys.modules.get("modules")
    sys.modules["modules"] = dropin_modules
    try:
        for name in ("train", "generate"):
            spec = importlib.util.spec_from_file_location("zeggs_ref_dropin_" + name, root() / "ZEGGS" / f"{name}.py")
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
            setattr(ns, name, m)
    finally:
        if mine is None:
            sys.modules.pop("modules", None)
        else:
            sys.modules["modules"] = mine
    return ns


_saved = {}


def release():
    """Give the top-level module names the reference occupied back to whoever held them before load() (the drop-in aliases of
    zeggs.compat) -- the loaded reference stays usable through the namespace load() returned until the next load()."""
    for k, v in list(_saved.items()):
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v
    _saved.clear()
