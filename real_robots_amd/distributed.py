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
    ships only the pixels that may differ from the step before.

    Why (SURVEY.md 8e with the measured rate instead of the 1e6 target): at 6.9 M env-steps/s per GPU a full-slab all-gather
    (`gather_images`, 114 688 B per env-step) makes every rank SEND 0.79 TB/s and receive seven times that -- more than its seven
    xGMI links (7 x ~153 GB/s) carry.  An env's frame differs from its previous one in ~1 250 pixels -- exactly what the renderer's
    fragment lists hold (pixels won by moving geometry + pixels vacated) -- 12 B per record {pixel address, RGB, depth}: ~15 KB per
    env-step instead of 115 KB, ~100 GB/s per rank at that rate.

    One step: (1) the local records, PREFIX-PACKED (an env with a full-frame change costs its own records, not everybody's):
    with `env` (a BatchedREALRobotEnv whose last step rendered) straight from the renderer's lists by a HIP kernel
    (rr_pack_image_delta; one cumsum over RR_F_FRAG_COUNT for the offsets, no compare pass over the slabs); without it -- CPU tensors,
    the gloo tests -- from a bit compare against the rank's own block of the persistent copy; (2) one small all-gather of the ranks'
    record totals; (3) one all-gather of `cap` records per rank; (4) the records of every rank applied to the persistent images
    (rr_apply_image_delta on a GPU).  `cap`, the per-rank length of the payload collective, is
      * the largest total of THIS step, read on the host (one small blocking read per step) -- the default, always exact; or
      * with `sync_free=True`, `slack` x the largest total of the PREVIOUS step + `margin`, known without waiting for THIS step: the
        only wait is for the previous step's totals (an event recorded a whole step earlier) -- the host keeps one step of slack.  A step whose total outgrows that (resets, goal set-up, a camera move -- host-initiated
        events: call `invalidate()` with them and the next step ships the slabs) drops the surplus records; it is reported one step
        late in `stale_last` and repaired by a slab gather in the step after.
    When the records would outweigh the slabs (cap x 12 >= n x H x W x 7) the step ships the slabs and reseeds the copies.
    The result of an exact step is bit for bit the full-slab gather (tests/test_distributed_gloo.py, 2 and 4 ranks; on a GPU
    tests/test_gpu_round6.py).  The first call seeds the copies with one full-slab gather.

    The returned tensors ARE the persistent copies: the next call updates them in place -- clone what must outlive a step."""

    def __init__(self, group=None, env=None, sync_free=False, slack=1.5, margin=4096):
        self.group, self.env, self.sync_free, self.slack, self.margin = group, env, bool(sync_free), float(slack), int(margin)
        self.rgb = self.depth = None
        self.bytes_last = 0
        self.stale_last = False          # sync_free: the step BEFORE the last one dropped records (it has been repaired since)
        self.slab_steps = 0              # steps that shipped the slabs (seed, fallback, repair)
        self._lag = None                 # (pinned host copy of the last step's totals and overflow flag, event)
        self._rec = None

    def invalidate(self):
        """The next step ships the full slabs (call with whatever rewrites whole frames: resets of many envs, set_goal, rr_set_camera)."""
        self.rgb = self.depth = None
        self._lag = None

    def _seed(self, rgb, depth):
        self.rgb, self.depth = gather_images(rgb, depth, self.group)
        self.bytes_last = rgb.numel() + depth.numel() * 4
        self.slab_steps += 1
        self._lag = None
        return self.rgb, self.depth

    def _all_gather(self, t, world):
        import torch
        import torch.distributed as dist
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        try:
            dist.all_gather_into_tensor(out, t, group=self.group)
        except (RuntimeError, NotImplementedError, AttributeError):
            dist.all_gather(list(out.chunk(world, dim=0)), t, group=self.group)
        return out

    def step(self, rgb, depth):
        import torch
        import torch.distributed as dist
        from . import _native as nat
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        n, H, W = depth.shape
        npix = n * H * W
        if self.rgb is None:
            return self._seed(rgb, depth)
        dev = depth.device
        on_gpu = self.env is not None and dev.type == 'cuda'
        # ---- (1) the local records and their number
        if on_gpu:
            fc = torch.as_tensor(self.env.device_buffer(nat.F_FRAG_COUNT), device=dev).view(torch.int32).reshape(-1)
            csum = torch.cumsum(fc, 0)
            total = csum[-1:].to(torch.int64)
            offsets = (csum - fc).to(torch.int32)
            local = None
        else:
            mine_rgb = self.rgb[rank * n:(rank + 1) * n].reshape(npix, 3)
            mine_dep = self.depth[rank * n:(rank + 1) * n].reshape(npix)
            new_rgb, new_dep = rgb.contiguous().reshape(npix, 3), depth.contiguous().reshape(npix)
            changed = (new_rgb != mine_rgb).any(-1) | (new_dep.view(torch.int32) != mine_dep.view(torch.int32))
            pix = torch.nonzero(changed).reshape(-1)
            c = new_rgb[pix].to(torch.int32)
            local = torch.stack([pix.to(torch.int32), c[:, 0] | (c[:, 1] << 8) | (c[:, 2] << 16), new_dep[pix].view(torch.int32)], 1)
            total = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
        # ---- (2) every rank's total
        totals = self._all_gather(total, world)
        # ---- the length of the payload collective
        self.stale_last = False
        if self.sync_free:
            lag = self._lag
            self._lag = None
            if lag is None:
                cap = None                                  # (no history yet: this step ships the slabs)
            else:
                if lag[2] is not None:
                    lag[2].synchronize()                    # (recorded a step ago: long complete)
                lag_tot, lag_cap = lag[0], lag[1]
                if int(lag_tot.max()) > lag_cap:            # the previous step dropped records: repair with the slabs
                    self.stale_last = True
                    cap = None
                else:
                    cap = int(self.slack * int(lag_tot.max())) + self.margin
        else:
            cap = max(1, int(totals.max().item()))
        if cap is None or cap * 12 >= npix * 7:
            out = self._seed(rgb, depth)
            if self.sync_free:
                self._note_totals(totals, 1 << 62, dev)
            return out
        # ---- (3) the payload: cap records per rank
        if on_gpu:
            if self._rec is None or self._rec.shape[0] < cap:
                self._rec = torch.empty((max(cap, 1), 3), dtype=torch.int32, device=dev)
            rec = self._rec[:cap]
            nat.check(self.env.L.rr_pack_image_delta(self.env.h, offsets.data_ptr(), rec.data_ptr(), cap))
        else:
            rec = torch.zeros((cap, 3), dtype=torch.int32, device=dev)
            k = min(cap, local.shape[0])
            rec[:k] = local[:k]
        allrec = self._all_gather(rec.contiguous(), world)
        # ---- (4) into the persistent images
        if on_gpu:
            # (on torch's current stream -- the one the collectives above were enqueued on; the env's own stream is expected to be that
            # stream too, as in bench.py: rr_pack_image_delta runs on the library's stream)
            tot32 = totals.to(torch.int32)
            nat.check(nat.load_library().rr_apply_image_delta(allrec.data_ptr(), tot32.data_ptr(), world, cap, npix, self.rgb.data_ptr(),
                                                              self.depth.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        else:
            allrec = allrec.view(world, cap, 3)
            valid = torch.arange(cap, device=dev).unsqueeze(0) < totals.clamp(max=cap).unsqueeze(1)
            r_idx, j_idx = torch.nonzero(valid, as_tuple=True)
            v = allrec[r_idx, j_idx]
            addr = r_idx * npix + v[:, 0].to(torch.int64)
            flat_rgb, flat_dep = self.rgb.view(world * npix, 3), self.depth.view(world * npix)
            flat_rgb[addr] = torch.stack([v[:, 1] & 255, (v[:, 1] >> 8) & 255, (v[:, 1] >> 16) & 255], 1).to(torch.uint8)
            flat_dep[addr] = v[:, 2].contiguous().view(torch.float32)
        if self.sync_free:
            self._note_totals(totals, cap, dev)
        self.bytes_last = cap * 12 + 8
        return self.rgb, self.depth

    def _note_totals(self, totals, cap, dev):
        """sync_free: this step's totals travel to the host behind the step (pinned copy + event); the NEXT step reads them."""
        import torch
        if dev.type == 'cuda':
            host = torch.empty(totals.shape, dtype=totals.dtype, pin_memory=True)
            host.copy_(totals, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._lag = (host, cap, ev)
        else:
            self._lag = (totals.clone(), cap, None)
