"""Command line of the drop-in: the option surface of the reference's two scripts on the MI355X engine.

    python -m zeggs.cli train    -o options.json [-n NAME]                     (ZEGGS/main.py:10-74)
    python -m zeggs.cli generate -o options.json -s style.bvh -a speech.wav ... (ZEGGS/generate.py:414-525)
    python -m zeggs.cli generate -o options.json -c pairs.csv                   (batch mode, same CSV columns)

`options.json` is the reference's file (configs/configs_v*.json before training, <output_dir>/options.json after): keys
`train_opt`, `net_opt`, `paths` {base_path, path_processed_data, output_dir, models_dir}.  Differences from the reference's
scripts: there is no CPU path (`-g/--use_gpu` is accepted and implied), and the run's environment is not snapshotted
(`helpers.save_useful_info`: pip freeze / git state -- logging, out of scope)."""
import argparse
import csv
import datetime
import json
import sys
from pathlib import Path


def _paths(options):
    p = options["paths"]
    base = Path(p["base_path"])
    return p, base, base / p["path_processed_data"]


def cmd_train(a):
    from .train import train
    options = json.loads(Path(a.options).read_text())
    if a.name:
        options["name"] = a.name
    p, base, data = _paths(options)
    if p.get("output_dir") is None:      # a time-stamped run directory under <base>/outputs, as the reference does
        p["output_dir"] = str(base / "outputs" / datetime.datetime.now().strftime("%Y_%m_%d_%H_%M_%S"))
    out = Path(p["output_dir"])
    out.mkdir(parents=True, exist_ok=True)
    if p.get("models_dir") is None and not options["train_opt"]["resume"]:
        p["models_dir"] = str(out / "saved_models")
    models, logs = Path(p["models_dir"]), out / "logs"
    models.mkdir(parents=True, exist_ok=True)
    logs.mkdir(exist_ok=True)
    (out / "options.json").write_text(json.dumps(options, indent=4))      # what `generate -o` reads afterwards
    train(models_dir=models, logs_dir=logs, path_processed_data=data / "processed_data.npz",
          path_data_definition=data / "data_definition.json", train_options=options["train_opt"],
          network_options=options["net_opt"])
    return 0


def _truthy(v):
    return str(v).strip().lower() not in ("", "0", "false", "no", "nan", "none")


def cmd_generate(a):
    from .generate import generate_gesture
    options = json.loads(Path(a.options).read_text())
    p, _, data = _paths(options)
    network, results = Path(p["models_dir"]), Path(a.results_path) if a.results_path else Path(p["output_dir"]) / "results"
    results.mkdir(parents=True, exist_ok=True)
    kind = a.style_encoding_type
    jobs = []
    if a.csv:       # columns: base_path, audio, style, file_name, temperature, seed, use_gpu, frames, first_pose, generate
        with open(a.csv, newline="") as fh:
            for row in csv.DictReader(fh):
                if not _truthy(row.get("generate", "1")):
                    continue
                root = Path(row["base_path"])
                frames = [int(x) for x in row["frames"].split()] if _truthy(row.get("frames", "")) else None
                jobs.append(dict(audio=root / row["audio"], style=(root / row["style"], frames) if kind == "example" else row["style"],
                                 file_name=row.get("file_name") or None,
                                 first_pose=root / row["first_pose"] if _truthy(row.get("first_pose", "")) else None,
                                 temperature=float(row.get("temperature") or 1.0), seed=int(float(row.get("seed") or 1234))))
    else:
        if not (a.audio and a.style):
            raise SystemExit("generate: give -a AUDIO and -s STYLE (or -c CSV)")
        jobs.append(dict(audio=Path(a.audio), style=(Path(a.style), a.frames) if kind == "example" else a.style,
                         file_name=a.file_name, first_pose=Path(a.first_pose) if a.first_pose else None,
                         temperature=a.temperature, seed=a.seed))
    for k, j in enumerate(jobs):
        print(f"[{k + 1}/{len(jobs)}] {j['audio']}  style {j['style']}", flush=True)
        generate_gesture(audio_file=j["audio"], styles=[j["style"]], network_path=network, data_path=data, results_path=results,
                         style_encoding_type=kind, file_name=j["file_name"], first_pose=j["first_pose"],
                         temperature=j["temperature"], seed=j["seed"], use_gpu=True)
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="zeggs", description="ZeroEGGS on the MI355X engine: train / generate")
    sub = ap.add_subparsers(dest="cmd", required=True)
    t = sub.add_parser("train", help="train the networks (reference: python main.py -o ... -n ...)")
    t.add_argument("-o", "--options", required=True, help="options file (configs_v*.json layout)")
    t.add_argument("-n", "--name", help="run name stored in the options")
    t.set_defaults(fn=cmd_train)
    g = sub.add_parser("generate", help="generate gesture BVHs (reference: python generate.py -o ... -s ... -a ...)")
    g.add_argument("-o", "--options", required=True, help="options.json written by training")
    g.add_argument("-p", "--results_path", nargs="?", default=None, help="where the BVH / WAV pairs go (default <output_dir>/results)")
    g.add_argument("-se", "--style_encoding_type", default="example", choices=("example", "label"))
    g.add_argument("-s", "--style", help="style exemplar BVH (or the label name with -se label)")
    g.add_argument("-a", "--audio", help="16 kHz speech WAV")
    g.add_argument("-n", "--file_name", help="output base name")
    g.add_argument("-fp", "--first_pose", default=None, help="BVH whose first frame starts the animation")
    g.add_argument("-t", "--temperature", type=float, nargs="?", default=1.0, help="VAE temperature")
    g.add_argument("-r", "--seed", type=int, nargs="?", default=1234)
    g.add_argument("-g", "--use_gpu", action="store_true", help="accepted for compatibility: the engine always runs on the GPU")
    g.add_argument("-f", "--frames", type=int, nargs=2, help="start and end frame of the exemplar")
    g.add_argument("-c", "--csv", help="CSV with one audio / style pair per row (evaluation_example_based.csv layout)")
    g.set_defaults(fn=cmd_generate)
    a = ap.parse_args(argv)
    return a.fn(a)


if __name__ == "__main__":
    sys.exit(main())
