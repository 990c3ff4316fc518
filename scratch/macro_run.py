"""The macro workload of bench.py's secondary leg (config 5 shape: 4096 envs, random macro actions, render every step) as a plain run,
for rocprofv3 --kernel-trace + scratch/timeline.py.  argv[1]: steps (default 450)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
T = int(sys.argv[1]) if len(sys.argv) > 1 else 450
RENDER = os.environ.get('RENDER', '1') != '0'
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
env.plan_macro(np.random.default_rng(0).uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
K = int(os.environ.get('SYNC_EVERY', '0'))          # bound the host's run-ahead: a sync every K steps (0: never)
for t in range(300):
    env.step_plan(render=RENDER)
    if K and t % K == K - 1: env.sync()
env.sync(); t0 = time.perf_counter()
for t in range(T - 300):
    env.step_plan(render=RENDER)
    if K and t % K == K - 1: env.sync()
env.sync()
cls = env.host(nat.F_ENV_CLASS)
print('macro workload: %.4f ms per step over steps 300..%d; heavy %d very heavy %d' % ((time.perf_counter() - t0) / (T - 300) * 1e3, T, (cls == 1).sum(), (cls == 2).sum()))
env.close()
