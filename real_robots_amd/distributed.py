"""Multi-GPU sharding of a batch of REALRobot envs: one process per GPU, contiguous env-index blocks, no collective
on the stepping path (envs are independent; the reference runs exactly one env per process). The only collective is
the optional observation gather (RCCL over xGMI on GPUs via torch.distributed backend "nccl"; "gloo" in CPU tests).
"""
import numpy as np


def shard_range(num_envs_total, rank, world_size):
    """Contiguous block [start, stop) of global env ids owned by `rank` (block sizes differ by at most one)."""
    base, rem = divmod(int(num_envs_total), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_of(global_env_id, num_envs_total, world_size):
    for r in range(world_size):
        a, b = shard_range(num_envs_total, r, world_size)
        if a <= global_env_id < b:
            return r, global_env_id - a
    raise IndexError(global_env_id)


def synthetic_actions(global_env_ids, step, seed=1234, hold_prob=0.05):
    """README-style resample-and-hold joint commands keyed by (seed, global env id), so a shard produces the same
    commands for its envs whatever the world size (bitwise shard equivalence). float32 [n, 9].
    The command of env e at step t is the sample drawn at the last resample time <= t."""
    lo = np.array([-2.09, -2.09, -2.96, -2.09, -2.96, -2.09, -3.05, 0.0, 0.0])
    hi = np.array([2.09, 2.09, 2.96, 2.09, 2.96, 2.09, 3.05, 1.5708, 1.5708])
    out = np.empty((len(global_env_ids), 9), np.float32)
    period = int(round(1.0 / hold_prob))
    for i, e in enumerate(global_env_ids):
        epoch = (int(step) + int(e) % period) // period      # resample every `period` steps, phase-shifted per env
        rng = np.random.default_rng([seed, int(e), epoch])
        out[i] = rng.uniform(lo, hi)
    return out


def gather_observations(local_obs, group=None):
    """All-gather a dict of equally-shaped local torch tensors along dim 0 (the optional policy-side gather;
    SURVEY.md 8e). Works on any torch.distributed backend."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = {}
    for k, t in local_obs.items():
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous(), group=group)
        out[k] = torch.cat(parts, dim=0)
    return out


def gather_images(rgb, depth, group=None):
    """All-gather of the per-rank image slabs (RGB u8 [n, H, W, 3], depth f32 [n, H, W]) into [world * n, ...] tensors:
    one collective per slab (470 MB per rank at 4096 envs x 128x128), which RCCL moves over the seven xGMI links of a
    fully connected node directly -- bucket sizes far above the latency-bound regime.  Falls back to the
    list form of all_gather on backends without all_gather_into_tensor."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    outs = []
    for t in (rgb, depth):
        t = t.contiguous()
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        try:
            dist.all_gather_into_tensor(out, t, group=group)
        except (RuntimeError, NotImplementedError, AttributeError):
            parts = list(out.chunk(world, dim=0))
            dist.all_gather(parts, t, group=group)
        outs.append(out)
    return outs[0], outs[1]
