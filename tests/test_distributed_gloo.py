"""world_size-2 CPU test (gloo) of the multi-GPU path: shard assignment + observation all-gather."""
import os

import numpy as np
import pytest


def _worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from real_robots_amd.distributed import gather_observations, shard_range, synthetic_actions
    total = 10
    a, b = shard_range(total, rank, world)
    acts = torch.from_numpy(synthetic_actions(range(a, b), step=3))
    ids = torch.arange(a, b, dtype=torch.int64)
    # pad to equal shard size for all_gather (the library shards evenly when N % world == 0)
    n = 5
    g = gather_observations({'act': acts[:n], 'id': ids[:n]})
    full = synthetic_actions(range(total), step=3)
    ok = bool((g['id'].numpy() == np.arange(total)).all() and (g['act'].numpy() == full).all())
    # image slabs (bench.py --gather images): every rank contributes its envs' RGB + depth, order = global env id
    from real_robots_amd.distributed import gather_images
    rgb = torch.full((n, 4, 8, 3), rank + 1, dtype=torch.uint8) + ids[:n].to(torch.uint8).view(n, 1, 1, 1)
    depth = torch.full((n, 4, 8), 0.25 * (rank + 1), dtype=torch.float32)
    grgb, gdepth = gather_images(rgb, depth)
    ok = ok and tuple(grgb.shape) == (total, 4, 8, 3) and tuple(gdepth.shape) == (total, 4, 8)
    want = np.concatenate([np.full(n, r + 1) + np.arange(r * n, (r + 1) * n) for r in range(world)])
    ok = ok and bool((grgb[:, 0, 0, 0].numpy() == want).all()) and bool((gdepth[n:, 0, 0].numpy() == 0.5).all())
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and t.item() == world
    with open(os.path.join(tmp, 'ok%d' % rank), 'w') as f:
        f.write('1' if ok else '0')
    dist.destroy_process_group()


def test_two_rank_shard_and_gather(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert open(os.path.join(str(tmp_path), 'ok%d' % r)).read() == '1'


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_plumbing_two_ranks_stub_env(scaling):
    """bench.py's own launch path with two ranks on CPU (gloo): `python -m torch.distributed.run ... bench.py --gpus 2` with
    --stub-env replacing the simulator.  Exercises the argument handling, the env-id sharding (weak: envs-per-gpu on every rank;
    strong: a fixed total split over the ranks), the barrier + max-over-ranks timing, the per-step observation gather and the
    JSON assembly -- so that the first multi-GPU run cannot fail on plumbing.  Rank 0 prints exactly one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 23000 + (os.getpid() * 7 + (0 if scaling == "weak" else 1)) % 4000
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
           '--presettle', '3', '--envs-per-gpu', '8', '--scaling', scaling, '--gather', 'lowdim', '--stub-env']
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    total = 16 if scaling == "weak" else 8
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2 and out["scaling"] == scaling
    assert out["config"]["envs_total"] == total and out["config"]["world"] == 2 and out["config"]["envs_per_gpu"] == total // 2
    assert out["config"]["gathered_bytes_per_step_per_rank"] == total * (9 + 4 + 21) * 4
    assert out["timed_steps"] == [5, 10] and out["value"] > 0 and "STUB" in out["data"] and out["roofline"] is None
    assert abs(out["value"] - total * 5 / (out["ms_per_step"] * 5e-3)) < 1e-2 * out["value"]
    ranks = sorted(ln for ln in r.stderr.splitlines() if ln.startswith('RANK '))
    half = total // 2
    assert ranks == ["RANK 0 ids 0 %d steps 10" % half, "RANK 1 ids %d %d steps 10" % (half, total)], ranks
