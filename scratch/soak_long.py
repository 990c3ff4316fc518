"""Long soak of both workloads (random joint commands with renders, macro actions with renders): no env may report a
non-finite state; images stay consistent with a from-scratch render of the final state (a fresh env given the same state)."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=True)
ids = np.arange(N)
t0 = time.time()
for t in range(6000):
    if t % 20 == 0: cmd = torch.from_numpy(synthetic_actions(ids, t, hold_prob=0.05) * (1.0 if (t // 1000) % 2 else 0.5)).cuda()
    env.step(device_ptr=cmd.data_ptr(), render=True)
    if t % 1000 == 999:
        print("random cmds step", t + 1, "errflags", int((env.host(nat.F_ERRFLAGS) != 0).sum()), "finite", bool(np.isfinite(env.state).all()), flush=True)
rng = np.random.default_rng(9)
for ep in range(4):
    env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
    for t in range(1000): env.step_plan(render=True)
    print("macro episode", ep, "errflags", int((env.host(nat.F_ERRFLAGS) != 0).sum()), "finite", bool(np.isfinite(env.state).all()), flush=True)
# images of the long-running env (10 000 incremental frames) against a fresh env rendering the same state from scratch
st = env.state; rgb, dep, msk = env.host(nat.F_RGB), env.host(nat.F_DEPTH), env.host(nat.F_MASK)
fresh = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=True)
fresh.state = st
fresh.render()
print("images equal a from-scratch render:", bool((fresh.host(nat.F_RGB) == rgb).all()), bool((fresh.host(nat.F_DEPTH) == dep).all()), bool((fresh.host(nat.F_MASK) == msk).all()))
print("elapsed %.1f s" % (time.time() - t0))
