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


class DeltaImageGather:
    """The policy-side image gather without the full slabs: every rank keeps PERSISTENT copies of all ranks' images and a step
    ships only the pixels that changed since the step before.

    Why (SURVEY.md 8e with the measured rate instead of the 1e6 target): at 6.4 M env-steps/s per GPU a full-slab all-gather
    (`gather_images`, 114 688 B per env-step) makes every rank SEND 0.74 TB/s and receive seven times that -- more than its seven
    xGMI links (7 x ~153 GB/s) carry.  An env's frame differs from its previous one in ~1 250 pixels (what the renderer's
    fragment lists hold: pixels won by moving geometry + pixels vacated): {pixel index 4 B, RGB 3 B, depth 4 B} = 11 B each,
    ~14 KB per env-step instead of 115 KB -- 88 GB/s per rank at that rate, a tenth of a link each way.

    One step: (1) the changed pixels of the local slab against the rank's own slab in the persistent copy (bit compare; a
    renderer-side list of changed pixels can replace this pass), (2) one small all-reduce (MAX) of the per-env counts fixes the
    padded length K of this step's records, (3) three all-gathers of [n, K] index / RGB / depth records, (4) a scatter into the
    persistent images.  Pad records rewrite pixel 0 with its own new value.  The result is bit for bit the full-slab gather
    (tests/test_distributed_gloo.py, 2 and 4 ranks).  The first call seeds the copies with one full-slab gather."""

    def __init__(self, group=None):
        self.group = group
        self.rgb = self.depth = None
        self.bytes_last = 0

    def step(self, rgb, depth):
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        n, H, W = depth.shape
        if self.rgb is None:
            self.rgb, self.depth = gather_images(rgb, depth, self.group)
            self.bytes_last = rgb.numel() + depth.numel() * 4
            return self.rgb, self.depth
        mine_rgb = self.rgb[rank * n:(rank + 1) * n].view(n, H * W, 3)
        mine_dep = self.depth[rank * n:(rank + 1) * n].view(n, H * W)
        new_rgb, new_dep = rgb.contiguous().view(n, H * W, 3), depth.contiguous().view(n, H * W)
        changed = (new_rgb != mine_rgb).any(-1) | (new_dep.view(torch.int32) != mine_dep.view(torch.int32))
        counts = changed.sum(1)
        kmax = counts.max().reshape(1).to(torch.int64)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX, group=self.group)
        K = max(int(kmax.item()), 1)
        # [n, K] records: the changed pixels of an env in pixel order, padded with pixel 0
        order = torch.cumsum(changed, 1) - 1
        idx = torch.zeros((n, K), dtype=torch.int64, device=depth.device)
        rows, pix = torch.nonzero(changed, as_tuple=True)
        idx[rows, order[rows, pix]] = pix
        rec_rgb = torch.gather(new_rgb, 1, idx.unsqueeze(-1).expand(n, K, 3)).contiguous()
        rec_dep = torch.gather(new_dep, 1, idx).contiguous()
        idx32 = idx.to(torch.int32)
        outs = []
        for t in (idx32, rec_rgb, rec_dep):
            out = torch.empty((world * n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            try:
                dist.all_gather_into_tensor(out, t, group=self.group)
            except (RuntimeError, NotImplementedError, AttributeError):
                dist.all_gather(list(out.chunk(world, dim=0)), t, group=self.group)
            outs.append(out)
        all_idx = outs[0].to(torch.int64)
        self.rgb.view(world * n, H * W, 3).scatter_(1, all_idx.unsqueeze(-1).expand(world * n, K, 3), outs[1])
        self.depth.view(world * n, H * W).scatter_(1, all_idx, outs[2])
        self.bytes_last = n * K * 11 + 8
        return self.rgb, self.depth
