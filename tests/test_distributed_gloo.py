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


def _delta_worker(rank, world, port, tmp):
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from real_robots_amd.distributed import DeltaImageGather, gather_images
    n, H, W = 3, 6, 8
    full = n * H * W * 7
    notes = []
    for sync_free in (False, True):
        g = torch.Generator().manual_seed(100 + rank)
        rgb = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, generator=g)
        depth = torch.rand((n, H, W), generator=g)
        dg = DeltaImageGather(sync_free=sync_free, slack=1.5, margin=24)
        ok, sent, exact, stale = True, [], [], []
        for t in range(10):
            if t > 0:
                # a frame differs from the one before in a few pixels: none at t = 3; at t = 5 ONE env of rank 0 changes almost
                # everywhere (the others four pixels: records are prefix-packed, nobody pays for that env but its rank); pixel 0 itself
                # at t = 6; at t = 8 every env of every rank changes everywhere (the records would outweigh the slabs)
                for e in range(n):
                    k = 0 if t == 3 else (H * W - 5 if (t == 5 and rank == 0 and e == 1) else (H * W if t == 8 else 4 + rank))
                    pix = torch.randperm(H * W, generator=g)[:k]
                    if t == 6:
                        pix = torch.cat([pix, torch.zeros(1, dtype=torch.int64)])
                    rgb.view(n, H * W, 3)[e, pix] = torch.randint(0, 256, (len(pix), 3), dtype=torch.uint8, generator=g)
                    depth.view(n, H * W)[e, pix[::2]] = torch.rand(len(pix[::2]), generator=g)
            a_rgb, a_dep = dg.step(rgb, depth)
            f_rgb, f_dep = gather_images(rgb, depth)
            exact.append(bool(torch.equal(a_rgb, f_rgb)) and bool(torch.equal(a_dep.view(torch.int32), f_dep.view(torch.int32))))
            sent.append(dg.bytes_last)
            stale.append(dg.stale_last)
        if not sync_free:
            # always exact; the seed frame and t = 8 ship the slabs; an unchanged frame ships one (empty) record slot; the frame with
            # one busy env costs that env's records on every rank's payload (cap = the largest rank total), not n times them
            ok = all(exact) and sent[0] == full and sent[8] == full and sent[3] == 12 + 8 and max(sent[1:3]) < full // 2
            ok = ok and sent[5] <= (H * W - 5 + 2 * (4 + world)) * 12 + 8 and dg.slab_steps == 2
        else:
            # no host read of this step's totals: cap = 1.5 x the previous step's largest total + 24 records.  Step 1 has no history
            # (slabs).  A frame that outgrows its cap -- the all-change frame of t = 8 for sure, the busy env of t = 5 with two ranks --
            # is incomplete on the ranks, flagged ONE step late (stale_last) and repaired by the slabs in that step; every frame
            # that is not flagged afterwards is exact
            ok = all(exact[t] == (not stale[t + 1]) for t in range(9)) and exact[9]
            ok = ok and all(sent[t] == full and exact[t] for t in range(10) if stale[t])
            ok = ok and stale[9] and not exact[8] and sent[0] == full and sent[1] == full and min(sent) < full // 2
        notes.append((sync_free, ok, sent, exact, stale))
    with open(os.path.join(tmp, 'okd%d' % rank), 'w') as f:
        f.write('1' if all(nt[1] for nt in notes) else '0 %r' % (notes,))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_delta_image_gather_is_bitwise_the_full_slab_gather(tmp_path, world):
    """bench.py --gather images-delta: persistent gathered images + per-step records of the changed pixels give,
    on every rank and after every step, bit for bit what the full-slab all-gather gives -- with frames that do not change at all,
    one env that changes almost everywhere (prefix-packed records: only its rank's total grows), a change of pixel 0, and frames
    that change everywhere (the step ships the slabs instead).  With `sync_free=True` (the payload sized from the PREVIOUS step's
    totals, no host read per step) every frame is exact except one that outgrows its cap, which is flagged and repaired a step later."""
    import torch.multiprocessing as mp
    port = 25500 + (os.getpid() * 3 + world) % 2000
    mp.spawn(_delta_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), 'okd%d' % r)).read() == '1'


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
    import re
    ranks = sorted(re.findall(r'RANK \d+ ids \d+ \d+ steps \d+', r.stderr))      # (the ranks' lines may interleave on the shared pipe)
    half = total // 2
    assert ranks == ["RANK 0 ids 0 %d steps 10" % half, "RANK 1 ids %d %d steps 10" % (half, total)], ranks


def test_bench_plumbing_four_ranks_gather_images_and_plan_check():
    """Four ranks on CPU (gloo), `--gather images`: every rank's env-id block, the image slabs' byte count, `ranks_seen` from the
    collective itself (an all_gather of ones with every rank's device ordinal and env-id block) and the plan the ranks reported."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 27000 + (os.getpid() * 13) % 4000
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '4', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '4', '--steps', '4', '--warmup', '1',
           '--presettle', '2', '--envs-per-gpu', '6', '--image', '16x8', '--gather', 'images', '--stub-env']
    env = dict(os.environ, OMP_NUM_THREADS='1')
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == 4 and cfg["world"] == 4 and cfg["ranks_seen"] == 4 and cfg["envs_total"] == 24
    assert [row[2:4] for row in cfg["plan"]] == [[0, 6], [6, 12], [12, 18], [18, 24]] and [row[4] for row in cfg["plan"]] == [0, 1, 2, 3]
    assert cfg["gathered_bytes_per_step_per_rank"] == 24 * (9 + 4 + 21) * 4 + 24 * 16 * 8 * (3 + 4)
    import re
    ranks = sorted(re.findall(r'RANK \d+ ids \d+ \d+ steps \d+', r.stderr))      # (the ranks' lines may interleave on the shared pipe)
    assert ranks == ["RANK %d ids %d %d steps 7" % (k, 6 * k, 6 * k + 6) for k in range(4)], ranks
    # the same launch with the delta gather: the stub's images never change, so a step ships one (empty) 12-byte record slot per rank
    cmd2 = [a if a != 'images' else 'images-delta' for a in cmd]
    r2 = subprocess.run(cmd2, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=env, cwd=root)
    assert r2.returncode == 0, r2.stderr[-2000:]
    out2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.strip().startswith('{')][0])
    assert out2["config"]["gather"] == 'images-delta'
    assert out2["config"]["gathered_bytes_per_step_per_rank"] == 24 * (9 + 4 + 21) * 4 + 4 * (1 * 12 + 8)


def test_plan_check_names_every_violation():
    """bench.check_plan: what makes `bench.py --gpus N` exit non-zero before the first step."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    good = [[1, 0, 0, 8, 0], [1, 1, 8, 16, 1]]
    assert bench.check_plan(good, 2, 16) is None
    assert 'saw 1 ranks' in bench.check_plan(good[:1], 2, 16)
    assert 'share a device' in bench.check_plan([[1, 0, 0, 8, 0], [1, 0, 8, 16, 1]], 2, 16)
    assert bench.check_plan([[1, 0, 0, 8, 0], [1, 0, 8, 16, 1]], 2, 16, distinct_devices=False) is None
    assert 'do not tile' in bench.check_plan([[1, 0, 0, 8, 0], [1, 1, 9, 16, 1]], 2, 16)
    assert 'end at 15' in bench.check_plan([[1, 0, 0, 8, 0], [1, 1, 8, 15, 1]], 2, 16)
    assert 'out of order' in bench.check_plan([[1, 1, 8, 16, 1], [1, 0, 0, 8, 0]], 2, 16)
