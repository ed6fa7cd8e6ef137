"""world_size-2 data-parallel path on CPU (gloo): index sharding + ONE all-reduce of the flat gradient buffer
reproduces the single-process global-batch gradients.  The arithmetic is the CPU oracle (no GPU here); the
collective/sharding code is the engine's (zeggs.engine.shard_indices / allreduce_mean_)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    sys.path[:0] = [str(root), str(root / "ubisoft-laforge-zeroeggs_amd"), str(root / "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle import loss as oloss
    from oracle import nets as onets
    from zeggs import engine, synth
    import helpers
    se, de, st = helpers.build_nets()
    s = helpers.stats_tensors()
    per_rank, T, L = 1, 4, 6
    data = synth.make_processed(2, 0, T + 6, seed=8)
    ds = engine.DeviceDataset(data, T, torch.device("cpu"))
    perm = np.random.default_rng(5).permutation(len(ds))
    eps_all = torch.randn(per_rank * world, 64, generator=torch.Generator().manual_seed(3))

    def grads_for(idx, eps, gscale):
        ws = [{k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()} for m in (se, de, st)]
        st_rows = ds.example_rows(idx, L)
        g = lambda t, r: torch.stack([t[a:a + T] for a in r])  # noqa: E731
        starts = ds.win_start[idx]
        audio = (g(ds.audio, starts) - ds.audio_mean) / ds.audio_std
        pose, rpos, rrot, gaze = g(ds.pose, starts), g(ds.rpos, starts), g(ds.rrot, starts), g(ds.gaze, starts)
        ex = torch.cat([ds.pose[torch.as_tensor(st_rows)], torch.zeros(len(idx), L, 3)], dim=-1)
        ex = (ex - ds.in_mean) / ds.in_std
        J = 75
        sp = lambda p: (p[..., 0:3], p[..., 3:6], p[..., 6:6 + 3 * J].reshape(*p.shape[:-1], J, 3),  # noqa: E731
                        p[..., 6 + 3 * J:6 + 9 * J].reshape(*p.shape[:-1], J, 2, 3),
                        p[..., 6 + 9 * J:6 + 12 * J].reshape(*p.shape[:-1], J, 3),
                        p[..., 6 + 12 * J:].reshape(*p.shape[:-1], J, 3))
        speech = onets.speech_encoder(ws[0], audio)
        z, mu, lv = onets.style_encoder(ws[2], ex, eps)
        W = (rpos, rrot) + sp(pose)
        O = onets.decoder_rollout(ws[1], rpos[:, 0], rrot[:, 0], *[w[:, 0] for w in sp(pose)], gaze, speech,
                                  z.unsqueeze(1).repeat(1, T, 1), s["in_mean"], s["in_std"], s["out_mean"],
                                  s["out_std"], synth.DT)
        loss, _ = oloss.training_loss(O, W, gaze, synth.PARENTS, synth.DT, mu, lv, iteration=0)
        (loss * gscale).backward()
        return torch.cat([v.grad.flatten() for w in ws for v in w.values()])

    # this rank's shard of global batch 0, gradients pre-scaled by 1/world, ONE all-reduce
    idx = engine.shard_indices(perm, 0, per_rank, world, rank)
    flat = grads_for(idx, eps_all[rank * per_rank:(rank + 1) * per_rank], 1.0 / world)
    engine.allreduce_mean_(flat, world, prescaled=True)
    # single-process reference: the whole global batch at once
    full = grads_for(perm[:per_rank * world], eps_all, 1.0)
    err = float((flat - full).abs().max() / full.abs().max())
    other = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(other, torch.tensor([float(flat.double().sum())]))
    out[rank] = (err, [float(o) for o in other], [int(i) for i in idx])
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_equals_global_batch():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (e0, sums0, idx0), (e1, sums1, idx1) = out[0], out[1]
    assert e0 < 1e-4 and e1 < 1e-4, (e0, e1)               # averaged shard grads == global-batch grads
    assert sums0 == sums1                                    # both ranks hold identical reduced gradients
    assert set(idx0).isdisjoint(idx1) and len(idx0) == len(idx1) == 1


def test_bench_self_launches_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment re-executes itself under
    torch.distributed.run on 127.0.0.1 (one rank per GPU; gloo on this GPU-less box) -- VERDICT r1 item 2."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--launch-selftest"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["world_size"] == 2 and [x["rank"] for x in out["ranks"]] == [0, 1]
    sys.path.insert(0, str(root))
    import bench
    cmd = bench.launch_command(["--gpus", "8", "--steps", "20"], 8, 29511)
    assert "--nproc-per-node=8" in cmd and cmd[-4:] == ["--gpus", "8", "--steps", "20"] and "127.0.0.1" in cmd
    # the ranks' environment: dmabuf IPC for RCCL, 8 hardware queues (engine streams + RCCL's), and LOCAL_RANK k -> GPU k
    env8 = bench.launch_env({})
    assert env8["GPU_MAX_HW_QUEUES"] == "8" and env8["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "OMP_NUM_THREADS" in env8
    assert bench.launch_env({"GPU_MAX_HW_QUEUES": "4"})["GPU_MAX_HW_QUEUES"] == "4"          # the user's setting wins
    assert [bench.rank_device(k).index for k in range(8)] == list(range(8))


def test_bench_dataset_is_built_once_and_shared_between_ranks(tmp_path):
    """multi-rank start-up: the builder rank writes the distinct synthetic clips, the others map them -- same dataset"""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    saved = bench.CLIP_FRAMES
    bench.CLIP_FRAMES = 300
    try:
        calls = []
        a = bench.build_dataset(n_train=4, n_unique=2, shared=(tmp_path / "d.npz", True, lambda: calls.append(1)))
        b = bench.build_dataset(n_train=4, n_unique=2, shared=(tmp_path / "d.npz", False, lambda: calls.append(1)))
        c = bench.build_dataset(n_train=4, n_unique=2)
    finally:
        bench.CLIP_FRAMES = saved
    assert calls == [1, 1]
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])) and np.array_equal(np.asarray(a[k]), np.asarray(c[k])), k


def test_ranks_draw_different_noise_streams():
    """train() re-seeds the library's noise stream with seed + 7919 * rank after the (common-seed) weight init, so two
    data-parallel ranks draw different dropout masks / VAE eps."""
    from zeggs import ops
    seeds = []
    for rank in (0, 1):
        ops.manual_seed(1234 + 7919 * rank)
        seeds.append([ops.next_seed() for _ in range(4)])
    assert not set(seeds[0]) & set(seeds[1])
    ops.manual_seed(1234)
    again = [ops.next_seed() for _ in range(4)]
    assert again == seeds[0]                     # and the stream is reproducible per rank
