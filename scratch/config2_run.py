"""BASELINE config 2 (1024 envs, 1 object, no render) as a plain run for rocprofv3 --kernel-trace + scratch/timeline.py; argv[1]: steps (default 2500)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
n = int(os.environ.get('N', '1024'))
cmds = bench.make_commands(torch, np, np.arange(n), T + 1, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(n, objects=1, width=64, height=64, want_mask=False)
for t in range(T - 400): env.step(device_ptr=cmds[t].data_ptr(), render=False)
env.sync(); t0 = time.perf_counter()
for t in range(T - 400, T): env.step(device_ptr=cmds[t].data_ptr(), render=False)
env.sync()
cls = env.host(nat.F_ENV_CLASS)
print('config 2 (%d envs): %.4f ms per step over steps %d..%d; heavy %d very heavy %d' % (n, (time.perf_counter() - t0) / 400 * 1e3, T - 400, T, (cls == 1).sum(), (cls == 2).sum()))
env.close()
