"""Development tool: the macro-action workload of bench.py's secondary entry (4096 envs, device-side plans, render every step),
steps 300..300+n, for profiling runs (rocprofv3 --kernel-trace -- python3 scratch/macro_run.py 60)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from real_robots_amd import _native as nat                    # noqa: E402
from real_robots_amd.batched import BatchedREALRobotEnv       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
rng = np.random.default_rng(0)
env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
for t in range(300):
    env.step_plan(render=True)
env.sync()
t0 = time.perf_counter()
for t in range(n):
    env.step_plan(render=True)
env.sync()
el = time.perf_counter() - t0
cls = env.host(nat.F_ENV_CLASS)
print("macro: %.1f env-steps/s, %.4f ms/step, heavy %d very heavy %d" % (N * n / el, el / n * 1e3, (cls == 1).sum(), (cls == 2).sum()))
env.close()
