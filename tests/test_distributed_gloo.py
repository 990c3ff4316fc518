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
