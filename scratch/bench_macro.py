"""SURVEY 8(d) config 5 shape, measured: 4096 envs, R1M3-style random macro actions (a new one every 1000 steps, planned on
the device by k_plan_macro), 128x128 RGB+depth render every step.  Prints env-steps/s over whole episodes (planning
included) and the per-kernel times at the contact-rich part of an episode."""
import sys, time, json; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
N, EPISODES = 4096, 2
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
rng = np.random.default_rng(0)
lo, hi = np.array([-0.25, -0.5]), np.array([0.05, 0.5])          # macro_space, env.py:49-52
env.plan_macro(rng.uniform(lo, hi, size=(N, 2, 2)))
for t in range(50): env.step_plan(render=True)                   # warm-up
env.sync(); t0 = time.perf_counter(); steps = 0
for ep in range(EPISODES):
    env.plan_macro(rng.uniform(lo, hi, size=(N, 2, 2)))
    for t in range(1000):
        env.step_plan(render=True); steps += 1
env.sync(); dt = time.perf_counter() - t0
assert (env.host(nat.F_ERRFLAGS) == 0).all()
out = {"workload": "macro actions, %d envs, 3 objects, 128x128 render every step, %d episodes of 1000 steps" % (N, EPISODES),
       "env_steps_per_s": round(N * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4)}
env.plan_macro(rng.uniform(lo, hi, size=(N, 2, 2)))
for t in range(500): env.step_plan(render=True)
env.set_timing(1)
for t in range(20): env.step_plan(render=True)
tm = env.get_timing(); env.set_timing(0)
out["kernels_ms_at_step_500"] = {k: round(ms / max(n, 1), 4) for k, (ms, n) in tm.items() if n}
print(json.dumps(out))
