"""Recipe: snapshot what the measurement legs need of the UNMODIFIED reference into oracle/_ref/ (git-ignored, NOT
gpurun-ignored: it travels to the GPU box with the built .so, exactly like oracle/_ref/ binaries of a compiled reference).

TEST / MEASUREMENT INFRASTRUCTURE -- see oracle/__init__.py.  Run in the build container (needs /root/reference), from
`__graft_entry__.build()` or by hand:

    python -m oracle.build_ref

Writes
  oracle/_ref/zeggs_reference.tar.gz   the reference's Python sources (ZEGGS/**/*.py, configs/*.json) and the small data files
                                       its CPU path reads (data/processed_v{1,2}/{stats.npz,data_definition.json,
                                       data_pipeline_conf.json}), byte for byte, as ONE archive -- no reference source file is
                                       added to this repository's tree or history; oracle/ref_shims.py unpacks it into a
                                       temporary directory when /root/reference is absent (the GPU box), so that bench.py's
                                       `cpu_baseline` can time the reference ITSELF (kind = "reference") on the same box in
                                       the same run
  oracle/_ref/speech_encoder_v1.pt     the trained artefact the reference ships (data/outputs/v1/saved_models), loaded by
                                       tests/test_gpu_full_shapes.py through zeggs.compat.load_module
  oracle/_ref/MANIFEST.json            file list + sha256 of the archive members
"""
import hashlib
import io
import json
import shutil
import tarfile
from pathlib import Path

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent / "_ref"
DATA = ("stats.npz", "data_definition.json", "data_pipeline_conf.json")


def members():
    files = sorted(p for p in (REF / "ZEGGS").rglob("*.py") if "__pycache__" not in p.parts)
    files += sorted((REF / "configs").glob("*.json"))
    for v in ("processed_v1", "processed_v2"):
        files += [REF / "data" / v / n for n in DATA if (REF / "data" / v / n).exists()]
    return files


def build(force=False):
    """-> True if the snapshot exists afterwards.  No-op (False) where /root/reference is absent and nothing was built."""
    arc = OUT / "zeggs_reference.tar.gz"
    if not REF.is_dir():
        return arc.exists()
    OUT.mkdir(parents=True, exist_ok=True)
    files = members()
    manifest = {str(p.relative_to(REF)): hashlib.sha256(p.read_bytes()).hexdigest() for p in files}
    mpath = OUT / "MANIFEST.json"
    if not force and arc.exists() and mpath.exists() and json.loads(mpath.read_text()).get("files") == manifest:
        return True
    buf = io.BytesIO()
    with tarfile.open(fileobj=buf, mode="w:gz") as tar:
        for p in files:
            tar.add(str(p), arcname=str(p.relative_to(REF)))
    arc.write_bytes(buf.getvalue())
    pt = REF / "data" / "outputs" / "v1" / "saved_models" / "speech_encoder.pt"
    if pt.exists():
        shutil.copyfile(pt, OUT / "speech_encoder_v1.pt")
    mpath.write_text(json.dumps({"source": str(REF), "files": manifest}, indent=1))
    return True


if __name__ == "__main__":
    ok = build(force=True)
    print("oracle/_ref:", "built" if ok else "skipped (/root/reference absent)",
          sorted(p.name for p in OUT.glob("*")) if OUT.exists() else [])
