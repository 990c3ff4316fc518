#!/usr/bin/env python3
"""Headline benchmark: aggregate env-steps/s of the batched REALRobot env.step() hot path.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[2], the configuration the metric is quoted on): REALRobot2020-R2J3-v0, 4096 envs
per GPU, 3 objects, 128x128 top-down RGB + depth rendered every step; one "step" = one env.step() of every env in the
batch.  Joint commands: README-style resample-and-hold `action_space.sample()` over the FULL joint limits
(robot.py:58-67), keyed by global env id, resident in HBM before the timed region.
Weak scaling by default (every rank owns --envs-per-gpu envs; `--scaling strong` splits a fixed total over the ranks);
no collective on the stepping path; `--gather lowdim|images` adds the optional policy-side observation all-gather
(RCCL).  Rank 0 prints ONE JSON line.  Extra objects in that line:
  roofline      per-kernel HIP-event durations (library stream; the launches of a step one after the other: k_solve, k_raster
                ... are what the main stream runs for the light envs, k_solve_heavy / render_heavy what the side stream runs
                beside them for the few heavy ones) and, for the unit that dominates the step's critical path, ALGORITHMIC
                bytes per launch / duration against the 8 TB/s HBM peak (`achieved` is that ratio, not a measured HBM rate; the
                measured HBM bytes of the same configuration, when a committed PMC profile matches it, are `traffic`).
                `roofline.hbm_achievable_GBs`: device copy / triad bandwidth measured in this run (rr_device_microbench).
                `roofline.valu` prices the dominant kernel against the VALU issue rate a sample-test-like instruction mix reaches
                in this run on the same device (rr_device_microbench kind 2), which is what actually bounds it; the sweep over
                instruction kinds and occupancies behind that choice is profiles/r04_valu_issue.txt.
  timed_steps / heavy_envs   which steps of the workload the timed region covered and how many envs were heavy / very heavy
                (solved and rendered on the side streams) at its start and end: full-range random commands
                press more and more arms onto the table, so the rate depends on the window
  secondary     (N=1 only) the same library on the other workloads a reader needs to judge the headline: the LATE window
                (steps 2000-2200, SURVEY 8(d) config 3's horizon), half-range commands (round 1's headline), macro-action
                pushing and a batched evaluate() (BASELINE config 5), BASELINE config 2 (1024 envs, 1 object, no render),
                the reference's default 320x240 camera and the R1 shape (with mask) at 4096 envs, BASELINE config 1 (one env
                through real_robots.make(...).step, camera off / on, with the CPU oracle's single-core rate beside it)
  cpu_baseline  PyBullet ("reference") when importable on this box, else the CPU oracle ("port"; oracle/ is the checker,
                never the product), timed on the host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
N_OBJECTS = 3
W = H = 128
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
# VALU issue ceiling in wave64 instructions / s that `roofline.valu.frac` is priced against: a FIXED figure -- 256 CUs x 4 SIMDs x
# 2.4 GHz, one wave64 instruction per 2 cycles on a SIMD-32.  No real instruction stream reaches it on this part
# (profiles/r04_valu_issue.txt, tools/ubench/valu_issue.hip: single-kind streams saturate at 505-553 G, a sample-test-like mix at
# 806-975 G); the rate of that mix in THIS run (rr_device_microbench kind 2) is reported beside the fraction as
# `reference_mix_rate` -- it depends on the mix and on the clocks, so it is a reference point, not a ceiling.
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 2
RENDER_KERNELS = ('k_image_setup', 'k_raster', 'k_shade')                      # together they produce the observation image
SIDE_STREAM_KERNELS = ('k_solve_heavy', 'render_heavy')                        # the heavy envs' share, beside the main stream


def algo_bytes(n_obj, w, h):
    """Algorithmic bytes per env-step and kernel (DESIGN.md "Roofline accounting")."""
    state = (22 + 13 * n_obj + 11) * 4 * 2 + 9 * 4 + (9 + 4 + 7 * n_obj) * 4          # state R/W + command + low-dim obs
    inst = 22 * 12 * 4
    return {'k_prep': state, 'k_collide': state, 'k_solve': state, 'k_render_setup': 11 * 4 + 13 * n_obj * 4 + inst,
            'k_raster': inst,           # instance matrices in; + 8 B per listed fragment out (measured, added at run time)
            'k_image_setup': 0,         # steady state: does not run (first frame / earlier image-update schemes only)
            'k_shade': inst,            # + (8 B list entry in + 7 B pixel out) per list entry (measured, added at run time)
            '_state': state, '_image': w * h * 7 + inst}


# ---------------------------------------------------------------------------------------------- CPU baseline
def _cpu_worker(args):
    seconds, seed = args
    import numpy as np
    from oracle.oracle import Oracle
    from real_robots_amd.distributed import synthetic_actions
    o = Oracle(N_OBJECTS, W, H)
    acts = [synthetic_actions([seed], t)[0].astype(np.float64) for t in range(0, 400, 20)]
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            o.step(acts[(n // 20) % len(acts)])
            o.render()
            n += 1
    return n, time.perf_counter() - t0


def cpu_baseline(seconds=10.0):
    """Same per-env workload (full-range commands, step + 128x128 render), one env per process on every host core this
    process may use; time-bounded sample.  kind "reference" = PyBullet through oracle/pybullet_ref.py when `import
    pybullet` works on this box; otherwise kind "port" = oracle/rr_oracle.c."""
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    from oracle import pybullet_ref
    if pybullet_ref.available():
        try:
            return pybullet_ref.cpu_baseline(seconds=seconds, cores=cores, n_objects=N_OBJECTS, width=W, height=H)
        except Exception as ex:        # a broken install must not take the GPU measurement down with it
            sys.stderr.write("bench.py: PyBullet baseline failed (%r); falling back to the CPU oracle\n" % (ex,))
    from oracle import oracle as orc
    orc.build()
    ctx = mp.get_context('spawn')
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(seconds, i) for i in range(cores)])
    rate = sum(n / t for n, t in res)
    # BASELINE config 1 (one env, REALRobot2020-R2J1, reference default 320x240 eye): the oracle's single-core rate, camera off / on
    import numpy as np
    c1 = {}
    o = orc.Oracle(1, 320, 240)
    rng = np.random.default_rng(0)
    for cam in (False, True):
        n, t0 = 0, time.perf_counter()
        a = np.zeros(9)
        while time.perf_counter() - t0 < 2.0:
            if n % 20 == 0:
                a = rng.uniform(-1.5, 1.5, 9)
                a[7:] = np.abs(a[7:]) * 0.5
            o.step(a)
            if cam:
                o.render()
            n += 1
        c1["camera_on" if cam else "camera_off"] = round(n / (time.perf_counter() - t0), 1)
    return {"value": round(rate, 1), "unit": "env-steps/s", "cores": cores, "kind": "port", "config1_single_core": c1,
            "sample": "%d processes x 1 env x %.0f s each (%d env-steps in total), 3 objects, full-range commands, 128x128 "
                      "RGB+depth render every step (oracle/rr_oracle.c, float64 physics); PyBullet not importable on this box"
                      % (cores, seconds, sum(n for n, _ in res))}


# ---------------------------------------------------------------------------------------------- GPU legs
def make_commands(torch, np, ids, n_steps, scale, device):
    """One [n_local, 9] tensor per 20-step resample epoch, resident in HBM; returns the tensor of every step."""
    from real_robots_amd.distributed import synthetic_actions
    epochs, cmd_of_step = {}, []
    for t in range(n_steps):
        key = t // 20
        if key not in epochs:
            epochs[key] = torch.from_numpy(synthetic_actions(ids, key * 20, hold_prob=0.05) * np.float32(scale)).to(device)
        cmd_of_step.append(epochs[key])
    return cmd_of_step


def kernel_table(env, nat, n_local, n_obj, w, h, render, step_fn, nprof):
    """Per-kernel HIP-event durations (every kernel alone on the library's stream) over nprof steps of step_fn."""
    env.set_timing(1)
    for t in range(nprof):
        step_fn(t)
    timing = env.get_timing()
    env.set_timing(0)
    algo = algo_bytes(n_obj, w, h)
    frags = 0.0
    if render:
        frags = float(env.host(nat.F_FRAG_COUNT).sum()) / n_local      # list entries (won + vacated pixels), mean per env
        algo['k_raster'] += 8 * frags
        algo['k_shade'] += 15 * frags
    if render and not timing.get('k_render_setup', (0.0, 0))[1]:
        algo['k_solve'] += algo['k_render_setup']      # the light solve sets up the render instances of its envs itself
    kernels = {}
    for k, (ms, n) in timing.items():
        if not n:
            continue
        if k in SIDE_STREAM_KERNELS:
            # what an untimed step runs on the side stream for its few heavy envs: listed, not priced
            # (their number is known on the device only) and never the roofline unit -- they are off the step's critical path
            kernels[k] = {"avg_ms": round(ms / n, 4), "launches": n, "stream": "side (beside the main stream's solve + render)"}
        else:
            kernels[k] = {"avg_ms": round(ms / n, 4), "launches": n,
                          "algorithmic_GBs": round(algo[k] * n_local / (ms / n * 1e-3) / 1e9, 2)}
    return kernels, algo, frags


def heavy_counts(env, nat):
    """Envs the last step solved / rendered on the side streams (RR_F_ENV_CLASS: 1 heavy, 2 very heavy)."""
    cls = env.host(nat.F_ENV_CLASS)
    return {"heavy": int((cls == 1).sum()), "very_heavy": int((cls == 2).sum())}


def load_profile(name, cfg):
    """A committed PMC-derived profile (profiles/<name>) applies to a run only when it was collected on the same
    configuration; returns (dict, None) or (None, reason)."""
    path = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(path):
        return None, "no profiles/%s" % name
    try:
        prof = json.load(open(path))
    except Exception as ex:
        return None, "unreadable profiles/%s: %r" % (name, ex)
    pc = prof.get('config')
    if pc != cfg:
        return None, "profiles/%s was collected on %s, this run is %s" % (name, json.dumps(pc), json.dumps(cfg))
    # a profile is only quoted for the kernels it was collected on: the hash of the kernel source recorded by tools/profile_round.sh
    if prof.get('source_sha256') != kernel_source_hash():
        return None, "stale profile: profiles/%s was collected on another build of the kernel sources (real_robots_amd/csrc)" % name
    return prof, None


def kernel_source_hash():
    """sha256 over the kernel sources of the library (realrobot.hip and its rr_*.inc parts, in name order)."""
    import glob
    import hashlib
    d = os.path.join(ROOT, 'real_robots_amd', 'csrc')
    files = sorted(glob.glob(os.path.join(d, '*.hip')) + glob.glob(os.path.join(d, 'rr_*.inc')))
    if not files:
        return None
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b'\0' + open(f, 'rb').read())
    return h.hexdigest()


def secondary_workloads(torch, np, nat, BatchedREALRobotEnv, device, steps):
    """Short runs of the other workloads (each its own env handle), N=1 only; env-steps/s with the k_solve / k_raster
    HIP-event times so that a reader sees where the difference to the headline comes from."""
    out = []
    dev = 'cuda:%d' % device

    def timed(env, n_local, step_fn, presettle, steps_, label, n_obj, w, h, render, presettle_label=None):
        env.sync()
        tp = time.perf_counter()
        for t in range(presettle):
            step_fn(t)
        env.sync()
        if presettle_label:         # the run-up itself is a workload SURVEY 8(d) names (config 3 "over 2 000 steps"): the average from reset
            elp = time.perf_counter() - tp
            out.append({"workload": presettle_label, "value": round(n_local * presettle / elp, 1), "unit": "env-steps/s",
                        "ms_per_step": round(elp / presettle * 1e3, 4), "steps": presettle, "timed_steps": [0, presettle],
                        "heavy_envs": {"end": heavy_counts(env, nat)}})
        h0 = heavy_counts(env, nat)
        t0 = time.perf_counter()
        for t in range(presettle, presettle + steps_):
            step_fn(t)
        env.sync()
        el = time.perf_counter() - t0
        h1 = heavy_counts(env, nat)
        ok = bool((env.host(nat.F_ERRFLAGS) == 0).all())
        kern, _, _ = kernel_table(env, nat, n_local, n_obj, w, h, render, lambda t: step_fn(presettle + steps_ + t), 10)
        out.append({"workload": label, "value": round(n_local * steps_ / el, 1), "unit": "env-steps/s",
                    "ms_per_step": round(el / steps_ * 1e3, 4), "steps": steps_, "timed_steps": [presettle, presettle + steps_],
                    "heavy_envs": {"start": h0, "end": h1}, "all_envs_finite": ok,
                    "kernels_ms": {k: v["avg_ms"] for k, v in kern.items()}})

    ids = np.arange(ENVS_PER_GPU)
    # (0) the headline workload in its LATE window: SURVEY 8(d) quotes config 3 over 2000 steps; by then a quarter of the arms
    # press on the table (heavy / very heavy envs) and the rate is lower than in the default run's early window
    env = BatchedREALRobotEnv(ENVS_PER_GPU, objects=3, width=W, height=H, device=device, want_mask=False)
    cmds = make_commands(torch, np, ids, 2000 + 200 + 10, 1.0, dev)
    timed(env, ENVS_PER_GPU, lambda t: env.step(device_ptr=cmds[t].data_ptr(), render=True), 2000, 200,
          "config 3, LATE window: the headline workload (4096 envs, 3 objects, full-range commands, 128x128 render every step) "
          "timed over steps 2000-2200", 3, W, H, True,
          presettle_label="config 3 over its first 2000 steps (SURVEY 8(d)'s horizon): the headline workload from reset, average over steps 0-2000 "
                          "(4096 envs, 3 objects, full-range commands, 128x128 render every step)")
    env.close()
    del cmds
    # (1) round 1's headline: commands scaled by 0.5 (arms rarely reach the objects)
    env = BatchedREALRobotEnv(ENVS_PER_GPU, objects=3, width=W, height=H, device=device, want_mask=False)
    cmds = make_commands(torch, np, ids, 150 + steps + 10, 0.5, dev)
    timed(env, ENVS_PER_GPU, lambda t: env.step(device_ptr=cmds[t].data_ptr(), render=True), 150, steps,
          "config 3 with HALF-range commands (x0.5; round 1's headline workload): 4096 envs, 3 objects, 128x128 render every step",
          3, W, H, True)
    env.close()
    # (2) macro actions (BASELINE config 5 shape): the gripper sweeps over the table and pushes the objects
    env = BatchedREALRobotEnv(ENVS_PER_GPU, objects=3, width=W, height=H, device=device, want_mask=False)
    rng = np.random.default_rng(0)
    env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(ENVS_PER_GPU, 2, 2)))      # macro_space, env.py:49-52
    timed(env, ENVS_PER_GPU, lambda t: env.step_plan(render=True), 300, min(steps * 2, 400),
          "config 5 shape: 4096 envs, random macro actions planned on the device (k_plan_macro), steps 300.. of the 1000-step plans "
          "(gripper pushing objects), 3 objects, 128x128 render every step", 3, W, H, True)
    env.close()
    # (3) BASELINE config 2: dynamics only
    n2 = 1024
    env = BatchedREALRobotEnv(n2, objects=1, width=64, height=64, device=device, want_mask=False)
    cmds2 = make_commands(torch, np, np.arange(n2), 10000 + 10, 1.0, dev)
    timed(env, n2, lambda t: env.step(device_ptr=cmds2[t].data_ptr(), render=False), 0, 10000,
          "config 2: REALRobot2020-R2J1, 1024 envs, 1 object, full-range joint commands, no render (dynamics-only), 10 000 steps from reset "
          "(SURVEY 8(d)'s horizon)", 1, 64, 64, False)
    env.close()
    # (3b) the reference's DEFAULT camera (robot.py:30-31: 320x240) at 4096 envs, and R1 (mask observation, robot.py:99-112) at the
    # headline's 128x128: 7 B x 76 800 = 538 KB resp. 11 B x 16 384 = 180 KB of image per env-step (SURVEY 8d accounting)
    for (w_, h_, mask_, label) in ((320, 240, False, "config 3 at the reference's default camera: 4096 envs, 3 objects, full-range commands, 320x240 RGB+depth "
                                                        "every step (538 KB of image per env-step)"),
                                   (W, H, True, "REALRobot2020-R1J3 shape: 4096 envs, 3 objects, full-range commands, 128x128 RGB+depth+MASK every step "
                                                "(180 KB of image per env-step)")):
        try:
            env = BatchedREALRobotEnv(ENVS_PER_GPU, objects=3, width=w_, height=h_, device=device, want_mask=mask_)
            cmds = make_commands(torch, np, ids, 150 + steps + 10, 1.0, dev)
            timed(env, ENVS_PER_GPU, lambda t: env.step(device_ptr=cmds[t].data_ptr(), render=True), 150, steps, label, 3, w_, h_, True)
            out[-1]["image_algorithmic_GBs"] = round((w_ * h_ * (11 if mask_ else 7)) * ENVS_PER_GPU / (out[-1]["ms_per_step"] * 1e-3) / 1e9, 1)
            env.close()
            del cmds
        except Exception as ex:            # the headline must not depend on it
            out.append({"workload": label, "error": repr(ex)})
    # (4) BASELINE config 5 at size: batched evaluate() -- 4096 envs, macro actions, intrinsic phase + extrinsic trials with
    # goals of a seeded synthetic dataset, a device-side batched policy, scores computed on the device
    try:
        from real_robots_amd.evaluate import bench_evaluate_batched
        # (SURVEY 8(d)'s shape: intrinsic 2 000 steps + 5 trials x 2 000 steps, a new macro action every 1 000 steps; once without a
        # per-step render -- what a policy on joints / touch / object positions needs -- and once with the retina every step)
        out.append(bench_evaluate_batched(ENVS_PER_GPU, device=device, width=W, height=H, render_every=0))
        out.append(bench_evaluate_batched(ENVS_PER_GPU, device=device, width=W, height=H, render_every=1))
    except Exception as ex:            # the headline must not depend on it
        out.append({"workload": "config 5 end to end (evaluate_batched at 4096 envs)", "error": repr(ex)})
    # (5) BASELINE config 1: ONE env through the drop-in facade (what an unchanged BasePolicy agent sees), camera off / on
    import real_robots_amd as rr
    e1 = rr.make('REALRobot2020-R2J1-v0', device=device)
    e1.reset()
    rng = np.random.default_rng(0)
    res = {}
    for cam, n_steps in ((False, 2000), (True, 300)):
        a = np.zeros(9)
        for t in range(20):
            e1.step({'joint_command': a, 'render': cam})
        t0 = time.perf_counter()
        for t in range(n_steps):
            if rng.random() < 0.05:                 # README policy: a new action_space sample now and then, held in between
                a = e1.action_space['joint_command'].sample()
            e1.step({'joint_command': a, 'render': cam})
        res["camera_on" if cam else "camera_off"] = {"steps": n_steps, "value": round(n_steps / (time.perf_counter() - t0), 1),
                                                      "unit": "env-steps/s"}
    e1.close()
    out.append({"workload": "config 1: REALRobot2020-R2J1-v0, ONE env through real_robots.make(...).step (gym facade: rr_step + "
                            "observation read-back every step, 320x240 eye when the action asks for it)", **res,
                "note": "single-env drop-in latency: the step is one device-side latency chain (solve -> preparation -> collision pass, "
                        "~85 us) plus the facade's host time; observations come back through mapped host mirrors with one wait per step "
                        "(rr_map_observations / rr_sync_observations, DESIGN.md 5.3); the CPU oracle's single-core rate on the "
                        "same shape is cpu_baseline.config1_single_core"})
    return out


def check_plan(rows, world, total, distinct_devices=True):
    """rows[r] = [1, device ordinal, first env id, end env id, rank] as gathered from rank r.  Returns None when the plan holds,
    else the reason: wrong rank count, a rank out of place, two ranks on one device, env-id blocks that do not tile [0, total)."""
    if len(rows) != world or sum(r[0] for r in rows) != world:
        return "the collective saw %d ranks, the launch promised %d" % (sum(r[0] for r in rows), world)
    if [r[4] for r in rows] != list(range(world)):
        return "ranks out of order in the gather: %s" % [r[4] for r in rows]
    if distinct_devices and len({r[1] for r in rows}) != world:
        return "two ranks share a device ordinal: %s" % [r[1] for r in rows]
    pos = 0
    for r in rows:
        if r[2] != pos or r[3] < r[2]:
            return "env-id blocks do not tile [0, %d): rank %d holds [%d, %d), expected to start at %d" % (total, r[4], r[2], r[3], pos)
        pos = r[3]
    if pos != total:
        return "env-id blocks end at %d, not at %d" % (pos, total)
    return None


class _StubEnv:
    """Plumbing test double (`--stub-env`, tests/test_distributed_gloo.py): the surface of BatchedREALRobotEnv that main() uses,
    on CPU tensors, so that argument handling, sharding, the barrier / max-over-ranks timing and the JSON assembly run under
    gloo without a GPU.  It simulates nothing; a line produced with it says so in `data` and carries no roofline."""

    def __init__(self, torch, np, nat, n, n_obj, w, h):
        self.torch, self.np, self.nat, self.N = torch, np, nat, n
        self.buf = {nat.F_JOINTS: torch.zeros(n, 9), nat.F_TOUCH: torch.zeros(n, 4), nat.F_OBJ_POSE: torch.zeros(n, n_obj, 7),
                    nat.F_RGB: torch.zeros(n, h, w, 3, dtype=torch.uint8), nat.F_DEPTH: torch.zeros(n, h, w)}
        self.steps = 0

    def step(self, device_ptr=None, render=False, cmd=None):
        self.steps += 1
        if cmd is not None:
            self.buf[self.nat.F_JOINTS] += 1e-3 * cmd

    def device_buffer(self, field):
        return self.buf[field]

    def host(self, field):
        np = self.np
        if field == self.nat.F_ERRFLAGS:
            return np.zeros(self.N, np.uint32)
        if field == self.nat.F_ENV_CLASS:
            return np.zeros(self.N, np.int32)
        if field == self.nat.F_FRAG_COUNT:
            return np.zeros((self.N, 1), np.uint32)
        return self.buf[field].numpy()

    def set_timing(self, on):
        pass

    def get_timing(self):
        return {k: (0.0, 0) for k in self.nat.KERNEL_NAMES}

    def sync(self):
        pass

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--envs-per-gpu', type=int, default=ENVS_PER_GPU)
    ap.add_argument('--objects', type=int, default=N_OBJECTS)
    ap.add_argument('--envs-per-block', type=int, default=0)
    ap.add_argument('--solver-iters', type=int, default=50)
    ap.add_argument('--command-scale', type=float, default=1.0,
                    help='scale of the synthetic joint commands (1.0 = the full joint limits; recorded in config)')
    ap.add_argument('--presettle', type=int, default=150,
                    help='untimed steps before the warmup so that the timed region runs in steady state '
                         '(objects landed on the table, arms moving, contacts active)')
    ap.add_argument('--image', default=None, help='WxH of the rendered observation (development A/B; the headline config is 128x128)')
    ap.add_argument('--no-render', action='store_true', help='dynamics-only variant (not the headline metric)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary workloads (profiling runs)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak: --envs-per-gpu envs on every rank; strong: --envs-per-gpu envs in total, split over the ranks')
    ap.add_argument('--gather', nargs='?', const='lowdim', default='none', choices=('none', 'lowdim', 'images', 'images-delta'),
                    help='also all-gather observations every step (RCCL): joints/touch/object poses, or those + RGB + depth as full slabs '
                         '(images) or as per-step records of the changed pixels applied to persistent copies (images-delta)')
    ap.add_argument('--stub-env', action='store_true',
                    help='plumbing test on CPU (gloo): a stub replaces the simulator; the printed line is marked as such')
    args = ap.parse_args()
    if args.image:
        global W, H
        W, H = (int(x) for x in args.image.lower().split('x'))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.stub_env:
        cpu = cpu_baseline()             # before this process touches the GPU

    import numpy as np
    import torch
    import torch.distributed as dist
    from real_robots_amd import _native as nat
    from real_robots_amd.batched import BatchedREALRobotEnv
    from real_robots_amd.distributed import DeltaImageGather, gather_images, gather_observations, shard_range

    stub = args.stub_env
    if stub:
        dev = 'cpu'
        device_sync = lambda: None                                                     # noqa: E731
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
        torch.cuda.set_device(local_rank)
        dev = 'cuda:%d' % local_rank
        device_sync = torch.cuda.synchronize
    if world > 1 or os.environ.get('RR_BENCH_FORCE_PG'):
        # (RR_BENCH_FORCE_PG: diagnostics -- a one-rank process group and one collective in front of the env's creation, as every rank
        # of an N > 1 run has: does RCCL's own set-up change what the step's streams get?  scratch/README.md)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if world == 1:
            os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('gloo' if stub else 'nccl')          # "nccl" = RCCL on ROCm
        if world == 1:
            probe = torch.ones(8, device=dev)
            dist.all_reduce(probe)
            device_sync()

    if args.scaling == 'strong':
        total = args.envs_per_gpu
        start, stop = shard_range(total, rank, world)
        n_local = stop - start
    else:
        n_local = args.envs_per_gpu
        total = n_local * world
        start, stop = shard_range(total, rank, world)
    ids = np.arange(start, stop)
    # N > 1: the collective backend itself has to confirm the plan -- every rank contributes {1, its device ordinal, its env-id
    # range} to one all-gather (RCCL on GPUs): the number of ranks seen, distinct devices, and env-id blocks that tile
    # [0, total) without gap or overlap.  Anything else ends the run with a one-line reason and a non-zero exit code.
    ranks_seen, plan_rows = 1, None
    if world > 1:
        mine = torch.tensor([1, local_rank, start, stop, rank], dtype=torch.int64, device=dev)
        rows = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        plan_rows = [[int(x) for x in r.cpu().tolist()] for r in rows]
        ranks_seen = sum(r[0] for r in plan_rows)
        why = check_plan(plan_rows, world, total, distinct_devices=not stub)
        if why:
            if rank == 0:
                print("bench.py: multi-GPU plan violated: " + why, file=sys.stderr)
            dist.destroy_process_group()
            raise SystemExit(3)
    n_obj = args.objects
    # An explicit stream for the library AND for torch's work of this process (the observation gather): the collectives are
    # ordered after the step that produced their buffers because both sit on this one stream, not because both happen to use
    # the legacy default stream.
    stream = None
    if stub:
        env = _StubEnv(torch, np, nat, n_local, n_obj, W, H)
    else:
        stream = torch.cuda.Stream(device=dev)
        env = BatchedREALRobotEnv(n_local, objects=n_obj, width=W, height=H, device=local_rank,
                                  envs_per_block=args.envs_per_block, solver_iters=args.solver_iters,
                                  want_mask=False, stream=stream.cuda_stream)     # R2 observations carry no mask (robot.py:99-112)
    render = not args.no_render

    n_total_steps = args.presettle + args.warmup + args.steps + 20
    cmd_of_step = make_commands(torch, np, ids, n_total_steps, args.command_scale, dev)
    device_sync()                          # the commands were uploaded on torch's default stream
    views = None
    if args.gather != 'none':
        views = {'joints': torch.as_tensor(env.device_buffer(nat.F_JOINTS), device=dev),
                 'touch': torch.as_tensor(env.device_buffer(nat.F_TOUCH), device=dev),
                 'objpose': torch.as_tensor(env.device_buffer(nat.F_OBJ_POSE), device=dev)}
        if args.gather in ('images', 'images-delta'):
            views['rgb'] = torch.as_tensor(env.device_buffer(nat.F_RGB), device=dev)
            views['depth'] = torch.as_tensor(env.device_buffer(nat.F_DEPTH), device=dev)
    gathered_bytes = 0

    # (records straight from the renderer's fragment lists on a GPU: rr_pack_image_delta; the stub / CPU path compares the slabs)
    delta_gather = DeltaImageGather(env=None if stub else env)

    def one_step(t):
        nonlocal gathered_bytes
        if stub:
            env.step(render=render, cmd=cmd_of_step[t])
        else:
            env.step(device_ptr=cmd_of_step[t].data_ptr(), render=render)
        if views is not None and world > 1:
            # (torch's current stream is the library's stream: see `stream` above)
            low = gather_observations({k: views[k] for k in ('joints', 'touch', 'objpose')})
            gathered_bytes = sum(v.numel() * v.element_size() for v in low.values())
            if args.gather == 'images':
                rgb, depth = gather_images(views['rgb'], views['depth'])
                gathered_bytes += rgb.numel() + depth.numel() * 4
            elif args.gather == 'images-delta':
                delta_gather.step(views['rgb'], views['depth'])
                gathered_bytes += world * delta_gather.bytes_last

    import contextlib
    on_stream = (lambda: torch.cuda.stream(stream)) if stream is not None else contextlib.nullcontext
    with on_stream():
        for t in range(args.presettle + args.warmup):
            one_step(t)
        device_sync()
        heavy_start = heavy_counts(env, nat)
        if world > 1:
            dist.barrier()
        device_sync()
        t0 = time.perf_counter()
        for t in range(args.presettle + args.warmup, args.presettle + args.warmup + args.steps):
            one_step(t)
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()
        elapsed = time.perf_counter() - t0
        heavy_end = heavy_counts(env, nat)
        if world > 1:
            tt = torch.tensor([elapsed], device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
    assert (env.host(nat.F_ERRFLAGS) == 0).all(), "an env reported a non-finite state"
    timed_steps = [args.presettle + args.warmup, args.presettle + args.warmup + args.steps]
    if stub:
        if rank == 0:
            print(json.dumps({"metric": "env-steps/sec (whole node), 4096 envs, R2J3 3-obj + 128x128 cam", "value": round(total * args.steps / elapsed, 1),
                              "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
                              "vs_baseline": None, "dtype": "f32", "data": "STUB ENV -- plumbing test, nothing was simulated",
                              "config": {"workload": "stub", "envs_total": total, "envs_per_gpu": n_local, "world": world,
                                         "gather": args.gather, "gathered_bytes_per_step_per_rank": gathered_bytes,
                                         "rank_env_ids": [int(start), int(stop)], "parallelism": "env-shard x%d" % world,
                                         "ranks_seen": ranks_seen, "plan": plan_rows},
                              "timed_steps": timed_steps, "roofline": None}))
        if world > 1:
            # every rank reports its shard for the test that launched it
            print("RANK %d ids %d %d steps %d" % (rank, start, stop, env.steps), file=sys.stderr)
            dist.destroy_process_group()
        return

    # per-kernel device time (HIP events on the library's stream), separate pass right behind the timed region
    nprof = min(20, args.steps)
    base_t = args.presettle + args.warmup + args.steps
    views_saved, views = views, None              # the timing pass measures the kernels, not the gather
    kernels, algo, frags = kernel_table(env, nat, n_local, n_obj, W, H, render, lambda t: one_step(base_t + t), nprof)
    views = views_saved
    # the dominant kernel = the largest share of the step's critical path: the main stream's launches (the side stream's
    # solve + render of the few heavy envs end before the main stream's render does, profiles/README.md)
    dom_kernel = max((k for k in kernels if k not in SIDE_STREAM_KERNELS), key=lambda k: kernels[k]["avg_ms"])
    dom = dom_kernel
    if render:
        # the image is produced by two kernels (visibility; deferred shading incl. putting vacated pixels back to the
        # static layer): the stage as a whole is what SURVEY 8(d)'s image bytes belong to
        members = [k for k in RENDER_KERNELS if k in kernels]
        rms = sum(kernels[k]["avg_ms"] for k in members) + (kernels['render_heavy']["avg_ms"] if 'render_heavy' in kernels else 0.0)
        kernels['render_stage'] = {"avg_ms": round(rms, 4), "launches": kernels['k_raster']["launches"], "members": members,
                                   "algorithmic_GBs": round(algo['_image'] * n_local / (rms * 1e-3) / 1e9, 2),
                                   "list_entries_per_env": round(frags, 1)}
        algo['render_stage'] = algo['_image']
        if dom_kernel in RENDER_KERNELS:
            dom = 'render_stage'      # the image is the unit SURVEY 8(d) prices; its kernels are reported together
    run_cfg = {"envs": n_local, "objects": n_obj, "width": W, "height": H, "render": bool(render),
               "command_scale": args.command_scale, "solver_iters": args.solver_iters}
    traffic, traffic_src = None, None
    prof, why = load_profile('traffic_latest.json', run_cfg)
    if prof is not None:
        traffic = prof.get(dom)
        traffic_src = "profiles/traffic_latest.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this configuration, not of this run" % prof.get('source', '?')
    else:
        traffic_src = "null: " + why
    # what this box achieves, measured in this run by the library (rr_device_microbench): HBM copy / triad bandwidth and the VALU
    # issue rate of a sample-test-like mix at the raster kernel's occupancy ("measure achievable ... in the same run")
    ubench, ubench_err = None, None
    try:
        ubench = nat.device_microbench(local_rank)
    except Exception as ex:            # the headline must not depend on it
        ubench_err = repr(ex)
    # `frac` is priced against a FIXED ceiling (256 CUs x 4 SIMDs x 2.4 GHz, one wave64 VALU instruction per 2 cycles on a
    # SIMD-32 = 1228.8 G wave-instr/s); the rate a sample-test-like instruction mix reaches in this run is reported beside it
    # (`reference_mix_rate`): it depends on the mix and on the clocks and is no ceiling
    valu_peak = VALU_PEAK_WAVE_INSTR
    mix_rate = ubench['valu_mix_G_wave_instr_s'] if ubench else None
    valu = None
    sq, why_sq = load_profile('sq_latest.json', run_cfg)
    if sq is not None and dom_kernel in sq.get('valu_wave_instr_per_launch', {}):
        wi = float(sq['valu_wave_instr_per_launch'][dom_kernel])
        dur = kernels[dom_kernel]["avg_ms"] * 1e-3
        valu = {"kernel": dom_kernel, "wave_instr_per_launch": round(wi), "wave_instr_per_env_step": round(wi / n_local, 1),
                "achieved": round(wi / dur / 1e9, 2), "peak": round(valu_peak / 1e9, 1), "unit": "G wave64-instr/s",
                "frac": round(wi / dur / valu_peak, 4),
                "peak_is": "fixed: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (SIMD-32)",
                "reference_mix_rate": mix_rate, "frac_of_reference_mix": (round(wi / dur / (mix_rate * 1e9), 4) if mix_rate else None),
                "reference_mix_is": ("measured in this run: rr_device_microbench kind 2, a sample-test-like mix (sub, mul, fma, cmp, cndmask) at "
                                     "eight waves per SIMD on every SIMD" if ubench else "not measured (%s)" % ubench_err) +
                                    "; the sweep over instruction kinds and 1..8 waves per SIMD is profiles/r04_valu_issue.txt: single-kind "
                                    "streams saturate at 505-553 G, the mix at 806-975 G",
                "source": "SQ_INSTS_VALU of profiles/sq_latest.json (%s, same configuration) / this run's HIP-event duration" % sq.get('source', '?')}
    else:
        valu = {"kernel": dom_kernel, "frac": None, "note": "null: " + (why_sq or "kernel not in the profile")}
    ach = kernels[dom]["algorithmic_GBs"]
    # what the counters say bounds the dominant kernel: VALU issue when its issue fraction exceeds its HBM fraction (it does,
    # tenfold); the HBM figures stay the contract's accounting (achieved / peak / frac)
    traffic_lo = prof.get('lower_bound', {}).get(dom) if prof is not None else None
    dom_ms = kernels[dom]["avg_ms"]
    traffic_frac = round(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6) if traffic else None
    bound = "valu" if (valu.get("frac") or 0.0) > max(ach / HBM_PEAK_GBS, traffic_frac or 0.0) else "hbm"
    roofline = {"bound": bound, "kernel": dom if dom != 'render_stage' else "+".join(kernels['render_stage']['members']),
                "dominant_single_kernel": dom_kernel,
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 6),
                "hbm_achievable_GBs": ({"copy": ubench['hbm_copy_GBs'], "triad": ubench['hbm_triad_GBs'],
                                        "frac_of_copy": round(ach / ubench['hbm_copy_GBs'], 6),
                                        "note": "device copy / triad of 256 MiB arrays measured in this run (rr_device_microbench); "
                                                "`peak` stays the 8 TB/s spec figure the contract prescribes"}
                                       if ubench else {"error": ubench_err}),
                "achieved_is": "ALGORITHMIC bytes per launch / HIP-event duration (full images, although a frame rewrites only "
                               "the pixels that changed) -- the contract's accounting, not a measured HBM rate; `traffic` is the measured one",
                "traffic": traffic, "traffic_lower_bound": traffic_lo, "achieved_traffic_frac": traffic_frac,
                "traffic_note": "HBM bytes of the unit per step from PMC counters; `traffic` applies the guide's x2 FETCH_SIZE correction "
                                "(calibrated for 16-B-per-lane coalesced reads only) to every read, `traffic_lower_bound` to none: "
                                "the unit's reads are mostly scattered 8/16-byte records, so the truth lies between; "
                                "achieved_traffic_frac = traffic / duration / 8 TB/s",
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(algo[dom] * n_local),
                "whole_step_algorithmic_GBs": round((algo['_state'] + (W * H * 7 if render else 0)) * n_local * args.steps / elapsed / 1e9, 2),
                "bound_note": "`bound` comes from the counters: the dominant kernel's VALU issue fraction (roofline.valu) against its HBM "
                              "fractions. achieved / peak / frac stay in the contract's HBM terms (algorithmic bytes): at 116 KB and "
                              "~0.3 M instructions per env-step the path is instruction-issue / latency bound (DESIGN.md 5): k_raster's "
                              "own VALU stream alone issues at 780-810 G wave-instr/s, with its 26.8 M LDS instructions per launch (2-6 "
                              "CU-cycles each) at 350-700 G -- VALU issue and LDS instruction issue load the CU about equally "
                              "(profiles/r04_raster_stream_issue.txt, r04_valu_lds_issue.txt), and 0.86 of the workgroup slots are busy "
                              "(profiles/r04_raster_wg_timeline.txt)",
                "valu": valu,
                "kernels": kernels,
                "timing_note": "per-kernel durations: HIP events on the library's stream, the launches of a step one after the "
                               "other (no side-stream overlap), %d steps of the same workload right after the timed region; "
                               "k_solve / k_render_setup / k_raster / k_shade = the main stream's launches (light envs), "
                               "k_solve_heavy / render_heavy = the side streams' (heavy envs); k_prep / k_collide = the look-ahead "
                               "of the next step, which an untimed step runs under the render; render_stage = the image of every "
                               "env = k_raster + k_shade + render_heavy; rocprofv3 --stats of the overlapped run is under profiles/" % nprof}
    env.close()

    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary and render and n_local == ENVS_PER_GPU:
        secondary = secondary_workloads(torch, np, nat, BatchedREALRobotEnv, local_rank, min(args.steps, 100))

    if rank == 0:
        rccl = None
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            pass
        cmd_txt = "full-range" if args.command_scale == 1.0 else "x%g-scaled" % args.command_scale
        out = {
            "metric": "env-steps/sec (whole node), 4096 envs, R2J3 3-obj + 128x128 cam",
            "value": round(total * args.steps / elapsed, 1), "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "timed_steps": timed_steps, "heavy_envs": {"start": heavy_start, "end": heavy_end, "of": n_local,
                                                       "note": "envs with generic contact rows at the start / end of the timed region "
                                                               "(rank 0): solved and rendered on the side streams; their number grows "
                                                               "with the step index, and the rate falls with it (secondary: late window)"},
            "config": {"workload": "REALRobot2020-R2J%d-v0, %d envs/GPU, %d object(s) + contact solver, %s joint commands "
                                   "(README resample-and-hold over the joint limits, robot.py:58-67), %s" %
                                   (n_obj, n_local, n_obj, cmd_txt,
                                    "128x128 top-down RGB+depth render every step" if render else "no render"),
                       "envs_total": total, "envs_per_gpu": n_local, "command_scale": args.command_scale,
                       "solver_iters": args.solver_iters, "dt": 0.005, "parallelism": "env-shard x%d" % world,
                       "gather": args.gather, "gathered_bytes_per_step_per_rank": gathered_bytes,
                       "world": world, "ranks_seen": ranks_seen, "rccl_version": rccl if world > 1 else None,
                       "ranks_seen_note": "sum over an RCCL all_gather of ones (with every rank's device ordinal and env-id block, checked "
                                          "against the plan before the first step; a violation exits non-zero)" if world > 1 else None,
                       "device": torch.cuda.get_device_name(local_rank),
                       "multi_gpu_note": None if world > 1 else "N>1 is unmeasured until the driver's SCALE run exists; "
                                                                "the step path has no collective (DESIGN.md 6)"},
            "roofline": roofline}
        if secondary is not None:
            out["secondary"] = secondary
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
