"""A/B helper: the headline workload from reset -- average over steps 0-2000 and the late window 2000-2200 (ms per step)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import importlib.util
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N = 4096
cmds = bench.make_commands(torch, np, np.arange(N), 2200, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
env.sync(); t0 = time.perf_counter()
for t in range(2000):
    env.step(device_ptr=cmds[t].data_ptr(), render=True)
env.sync(); t1 = time.perf_counter()
for t in range(2000, 2200):
    env.step(device_ptr=cmds[t].data_ptr(), render=True)
env.sync(); t2 = time.perf_counter()
cls = env.host(nat.F_ENV_CLASS)
print('%s steps 0-2000 %.4f ms (%.3f M)  late %.4f ms (%.3f M)  heavy %d vh %d  state sum %.6f' % (os.environ.get('RR_HEAVY_ON_MAIN'), (t1 - t0) / 2000 * 1e3, N * 2000 / (t1 - t0) / 1e6,
      (t2 - t1) / 200 * 1e3, N * 200 / (t2 - t1) / 1e6, (cls == 1).sum(), (cls == 2).sum(), float(np.abs(env.state).sum())), flush=True)
