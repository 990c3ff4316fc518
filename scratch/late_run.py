"""Development tool: the headline workload run on to its LATE window (steps 2000..2000+n: ~650 heavy envs) for kernel traces."""
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                  # noqa: E402
from real_robots_amd import _native as nat                    # noqa: E402
from real_robots_amd.batched import BatchedREALRobotEnv       # noqa: E402

spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N, T0 = 4096, 2000
cmds = bench.make_commands(torch, np, np.arange(N), T0 + n, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
for t in range(T0):
    env.step(device_ptr=cmds[t].data_ptr(), render=True)
env.sync()
t0 = time.perf_counter()
for t in range(T0, T0 + n):
    env.step(device_ptr=cmds[t].data_ptr(), render=True)
env.sync()
el = time.perf_counter() - t0
cls = env.host(nat.F_ENV_CLASS)
print("late window: %.1f env-steps/s, %.4f ms/step, heavy %d very heavy %d" % (N * n / el, el / n * 1e3, (cls == 1).sum(), (cls == 2).sum()))
env.close()
