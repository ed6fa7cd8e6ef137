"""train(): the reference's training entry point on the HIP engine.

Same signature and on-disk artefacts as ZEGGS/train.py:29-36 (options dictionaries = configs_v*.json blocks):
reads processed_data.npz + data_definition.json, writes models_dir/{speech_encoder,decoder,style_encoder,
checkpoints}.pt (+ models_dir/<iteration>/) every `generate_samples_step` iterations INCLUDING iteration 0, steps
the exponential LR decay every 1000 iterations, stops after niterations*1000 iterations (checked per epoch, like the
reference).  Differences (documented in DESIGN.md): the dataset lives in HBM and batches are gathered by a HIP
kernel; data parallelism over torch.distributed (RCCL) when WORLD_SIZE > 1.  Like the reference it renders three
training and three validation clips (ground truth + prediction) to logs_dir/samples/*.bvh at every checkpoint
(train.py:516-760) and logs the loss terms under logs_dir/tb when `use_tensorboard` is set (TensorBoard event files
when the tensorboard package is importable, a scalars.jsonl with the same tags otherwise).
"""
import copy
import datetime
import json
import os
import random
import sys
from pathlib import Path

import numpy as np
import torch

from . import engine, modules, ops


last_engine = None       # the TrainEngine of the most recent train() call (train() itself returns None, like the reference)


def compact_copy(module):
    """Deep copy of a module whose parameters are views of the engine's flat buffers, with every parameter in its
    OWN storage (torch.save of the views would write the whole flat buffer into each file)."""
    grads = [p.grad for p in module.parameters()]
    for q in module.parameters():
        q.grad = None
    try:
        return copy.deepcopy(module)        # Parameter.__deepcopy__ clones .data -> compact storage
    finally:
        for q, g in zip(module.parameters(), grads):
            q.grad = g


LOSS_TAGS = ("loss_root_pos", "loss_root_rot", "loss_root_vel", "loss_root_vrt", "loss_lpos", "loss_lrot", "loss_lvel",
             "loss_lvrt", "loss_cpos", "loss_crot", "loss_cvel", "loss_cvrt", "loss_ldvl", "loss_ldvt", "loss_cdvl",
             "loss_cdvt", "loss_gaze", "loss_kl_div")          # train.py:440-462, the order of the loss kernel's terms


class ScalarLog:
    """`use_tensorboard` sink (train.py:100-103,437-463): a SummaryWriter when tensorboard is installed, else the
    same tags as JSON lines in <dir>/scalars.jsonl."""

    def __init__(self, directory):
        directory = Path(directory)
        directory.mkdir(parents=True, exist_ok=True)
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer, self.file = SummaryWriter(log_dir=str(directory), flush_secs=10), None
        except ImportError:
            self.writer, self.file = None, open(directory / "scalars.jsonl", "a")

    def add(self, iteration, loss, terms):
        terms = terms[:len(LOSS_TAGS)].tolist()            # one device -> host copy
        loss = float(loss.detach())
        if self.writer is not None:
            self.writer.add_scalar("losses/total_loss", loss, iteration)
            self.writer.add_scalars("losses/losses", dict(zip(LOSS_TAGS, terms)), iteration)
        else:
            self.file.write(json.dumps({"iteration": iteration, "losses/total_loss": loss,
                                        "losses/losses": dict(zip(LOSS_TAGS, terms))}) + "\n")
            self.file.flush()

    def close(self):
        (self.writer or self.file).close()


def render_samples(samples_dir, iteration, ds, se, de, st, details, example_len, count=3, seconds=30):
    """train.py:516-760: `count` random training clips and `count` validation clips (cut to 30 s), each written as
    ground truth and as the decoder's free-running prediction from the clip's first pose, speech and style (its own
    frames as the style example, or its label).  Nets in eval mode, no_grad; B = 1 rollouts."""
    from . import anim
    J = len(details["parents"])
    parents, names, dt = np.asarray(details["parents"]), details["bone_names"], details["dt"]
    nlabels = len(details["label_names"])
    was_training = [m.training for m in (se, de, st) if m is not None]
    for m in (se, de, st):
        if m is not None:
            m.eval()
    written = []
    try:
        with torch.no_grad():
            for split in ("train", "valid"):
                if len(ds.ranges_train if split == "train" else ds.ranges_valid) == 0:
                    continue
                for i in range(count):
                    c = ds.sample_clip(split, seconds)
                    T = c["pose"].shape[1]
                    speech = se(c["audio"])
                    if st is not None:
                        z, _, _ = st(ds.clip_example(c["frames"], example_len))
                    else:
                        z = torch.zeros(1, nlabels, device=ds.device)
                        z[0, c["label"]] = 1.0
                    style = z[:, None].repeat(1, T, 1).contiguous()
                    pose, rpos, rrot = ops.decoder_core(de, c["pose"][:, 0].contiguous(), c["rpos"][:, 0].contiguous(),
                                                        c["rrot"][:, 0].contiguous(), c["gaze"], speech, style,
                                                        ds.in_mean, ds.in_std, ds.out_mean, ds.out_std, dt)
                    label = details["label_names"][c["label"]]
                    for kind, P, rp, rr in (("ground", c["pose"], c["rpos"], c["rrot"]), ("predict", pose, rpos, rrot)):
                        path = samples_dir / f"iteration_{iteration}_{split}_{kind}_{i}_{label}.bvh"
                        try:
                            anim.write_bvh(str(path), rp[0], rr[0], P[0, :, 6:6 + 3 * J].reshape(T, J, 3),
                                           P[0, :, 6 + 3 * J:6 + 9 * J].reshape(T, J, 2, 3), parents, names, "zyx", dt)
                            written.append(path)
                        except (PermissionError, OSError) as e:     # train.py:626-627: report and go on
                            print(e)
    finally:
        for m, tr in zip([m for m in (se, de, st) if m is not None], was_training):
            m.train(tr)
    return written


def train(models_dir, logs_dir, path_processed_data, path_data_definition, train_options, network_options):
    global last_engine
    models_dir, logs_dir = Path(models_dir), Path(logs_dir)
    np.random.seed(train_options["seed"])
    torch.manual_seed(train_options["seed"])
    ops.manual_seed(train_options["seed"])
    if not (train_options["use_gpu"] and torch.cuda.is_available()):
        raise RuntimeError("the ZeroEGGS MI355X engine has no CPU path (use_gpu=false / no GPU visible)")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not torch.distributed.is_initialized():
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    window, batchsize = train_options["window"], train_options["batchsize"]
    if batchsize * world == 3 or window == 3:
        # torch.cross without `dim` takes the FIRST axis of size 3 (ZEGGS/anim/txform.py:25-26, ZEGGS/train.py:301,315): with a
        # batch of 3 clips or a window of 3 frames the reference crosses along the batch / time axis.  The kernels always cross
        # along the coordinate axis -- the intended value -- so THIS is the one configuration whose losses differ from the reference's.
        import warnings
        warnings.warn("zeggs.train: batchsize == 3 or window == 3 -- the reference's torch.cross (no dim) crosses along that axis "
                      "instead of the coordinate axis there; this engine computes the intended cross product, so its loss and "
                      "gradients differ from the reference's for this configuration")
    se_opt, st_opt, de_opt = (network_options["speech_encoder"], network_options["style_encoder"],
                              network_options["decoder"])
    with open(path_data_definition) as f:
        details = json.load(f)
    nlabels, parents, dt = len(details["label_names"]), details["parents"], details["dt"]
    ds = engine.DeviceDataset(np.load(path_processed_data), window, device)
    style_type = train_options["style_encoding_type"]
    style_size = nlabels if style_type == "label" else st_opt["style_encoding_size"]
    paths = {k: models_dir / f"{k}.pt" for k in ("speech_encoder", "decoder", "style_encoder", "checkpoints")}
    resume = train_options["resume"] and all(paths[k].exists() for k in ("speech_encoder", "decoder", "checkpoints"))
    from . import compat
    if resume:
        se = compat.load_module(paths["speech_encoder"], device).to(device)
        de = compat.load_module(paths["decoder"], device).to(device)
        st = compat.load_module(paths["style_encoder"], device).to(device) if style_type == "example" else None
    else:      # construction order of the reference (train.py:118-139): same seed -> same initial weights
        se = modules.SpeechEncoder(ds.audio.shape[1], se_opt["nhidden"], se_opt["speech_encoding_size"]).to(device)
        de = modules.Decoder(ds.PO + 3, ds.PO, se_opt["speech_encoding_size"], style_size, de_opt["nhidden"], 2).to(device)
        st = None
        if style_type == "example":
            st = modules.StyleEncoder(ds.PO + 3, st_opt["nhidden"], style_size, type=st_opt["type"],
                                      use_vae=st_opt["use_vae"]).to(device)
    eng = engine.TrainEngine(se, de, st, ds, parents, dt, lr=train_options["learning_rate"], eps=train_options["eps"],
                             style_encoding_type=style_type, world_size=world, rank=rank)
    last_engine = eng
    # the common seed gave identical initial weights and gives the identical window permutation on every rank; the
    # NOISE streams (dropout masks, VAE eps: the library's counter-hash RNG) must differ per rank, otherwise the
    # global batch would see `world` copies of the same noise
    ops.manual_seed(train_options["seed"] + 7919 * rank)
    iteration = epoch = 0
    if resume:
        ck = torch.load(paths["checkpoints"], map_location=device, weights_only=False)
        iteration, epoch = ck["iteration"], ck["epoch"]
        eng.opt.load_state_dict(ck["optimizer_state_dict"])
        eng.opt.attach_flat(eng.flat_p, eng.flat_g, keep_state=True)
        eng.iteration = iteration
        # the noise stream (dropout masks, VAE eps) continues as a function of the ITERATION, not from its iteration-0 state
        # (a resumed run would otherwise replay the noise of the start of training)
        ops.manual_seed(train_options["seed"] + 7919 * rank + 104729 * iteration)
    (logs_dir / "samples").mkdir(parents=True, exist_ok=True)
    scalars = ScalarLog(logs_dir / "tb") if (train_options.get("use_tensorboard") and rank == 0) else None
    example_len = st_opt["example_length"]
    if resume and iteration > 0:     # the length the interrupted run had drawn for this iteration (seeded per iteration below)
        example_len = 2 * random.Random(train_options["seed"] * 1000003 + iteration - 1).randint(
            st_opt["example_length"] // 2, st_opt["example_length"])
    gb = batchsize * world
    labels_onehot = None
    if style_type == "label":       # one-hot row per training range (dataset.py:150-151), gathered per window by a HIP kernel
        labels_onehot = torch.as_tensor(np.eye(nlabels, dtype=np.float32)[ds.ranges_train_labels]).to(device)
    perm_rng = np.random.default_rng(train_options["seed"])           # identical on every rank
    for _ in range(epoch):                                             # resume: replay the permutations already consumed
        perm_rng.permutation(len(ds))
    while iteration < 1000 * train_options["niterations"]:
        start = datetime.datetime.now()
        perm = perm_rng.permutation(len(ds))
        nb = len(ds) // gb                                            # drop_last
        # resume: the batches of this epoch's permutation that the interrupted run had already consumed are skipped (the
        # checkpointed iteration itself is re-run, as the reference does: train.py:166-172)
        first = iteration - epoch * nb if (resume and 0 < iteration - epoch * nb < nb) else 0
        resume = False
        for bi in range(first, nb):
            se.train(), de.train()
            if st is not None:
                st.train()
            idx = engine.shard_indices(perm, bi, batchsize, world, rank)
            lab = ops.gather_rows(labels_onehot, ds.upload_indices(ds.win_sample[idx])) if labels_onehot is not None else None
            loss = eng.step(idx, example_len, labels=lab)
            # the example length of the NEXT iteration (train.py:228); seeded here so that all ranks agree
            example_len = 2 * random.Random(train_options["seed"] * 1000003 + iteration).randint(
                st_opt["example_length"] // 2, st_opt["example_length"])
            if bi + 1 < nb:         # the next batch is gathered on a side stream while this iteration is still running
                eng.prefetch(engine.shard_indices(perm, bi + 1, batchsize, world, rank), example_len)
            if (iteration + 1) % 1000 == 0:
                for g in eng.opt.param_groups:
                    g["lr"] *= train_options["learning_rate_decay"]
            checkpoint = iteration % train_options["generate_samples_step"] == 0
            if checkpoint:
                # the weights are about to be read for keeps: drain the give-up protocol first (steps a persistent sweep lost
                # within the last few iterations are re-run on the stage kernels NOW; every rank, same iteration)
                eng.flush()
            if eng.replayed:                 # steps the engine re-ran: their losses replace the NaN ones logged at the time
                for it_r, loss_r, terms_r in eng.replayed:
                    if scalars is not None:
                        scalars.add(it_r, loss_r, terms_r)
                    if rank == 0:
                        print(f"\n| it {it_r} re-run on the stage kernels | loss {float(loss_r):.4f} |")
                if eng.replayed[-1][0] == iteration:
                    loss = eng.replayed[-1][1]
                eng.replayed.clear()
            if scalars is not None:
                scalars.add(iteration, loss, eng.last_terms)
            if rank == 0 and iteration % 50 == 0:
                sys.stdout.write(f"\r| epoch {epoch} | it {iteration} | batch {bi}/{nb} | loss {float(loss.detach()):.4f} "
                                 f"| {datetime.datetime.now() - start} |")
            if rank == 0 and checkpoint:
                snap = [compact_copy(m) if m is not None else None for m in (se, de, st)]
                osd = eng.opt.state_dict()                 # moments are views of the flat buffers: store them compact
                osd["state"] = {k: {kk: (vv.detach().clone() if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
                                for k, v in osd["state"].items()}
                for d in (models_dir, models_dir / str(iteration)):
                    d.mkdir(parents=True, exist_ok=True)
                    torch.save(snap[0], d / "speech_encoder.pt")
                    torch.save(snap[1], d / "decoder.pt")
                    if st is not None:
                        torch.save(snap[2], d / "style_encoder.pt")
                    torch.save({"iteration": iteration, "epoch": epoch, "loss": loss.detach(),
                                "optimizer_state_dict": osd}, d / "checkpoints.pt")
                    try:                                   # pickle-free twin (optional dependency: safetensors)
                        compat.save_state(d, se, de, st, meta={"iteration": iteration, "epoch": epoch})
                    except ImportError as e:
                        print(f"\nwarning: safetensors twin not written ({e})")
                render_samples(logs_dir / "samples", iteration, ds, se, de, st, details, st_opt["example_length"])
            iteration += 1
        epoch += 1
    eng.flush()                               # nothing skipped on the device may be left behind when train() returns
    for it_r, loss_r, terms_r in eng.replayed:
        if scalars is not None:
            scalars.add(it_r, loss_r, terms_r)
        if rank == 0:
            print(f"\n| it {it_r} re-run on the stage kernels | loss {float(loss_r):.4f} |")
    eng.replayed.clear()
    if scalars is not None:
        scalars.close()
    if rank == 0:
        print("\nDone!")
    return None
