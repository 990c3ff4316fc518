"""BASELINE config 2 (1024 envs, 1 object, no render, full-range commands): env-steps/s and kernel times; RR_COOP_ALL=0 for the A/B."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
for n in (1024, 512, 256):
    cmds = bench.make_commands(torch, np, np.arange(n), 600, 1.0, 'cuda:0')
    env = BatchedREALRobotEnv(n, objects=1, width=64, height=64, want_mask=False)
    for t in range(150):
        env.step(device_ptr=cmds[t].data_ptr(), render=False)
    env.sync()
    t0 = time.perf_counter()
    for t in range(150, 550):
        env.step(device_ptr=cmds[t].data_ptr(), render=False)
    env.sync()
    el = time.perf_counter() - t0
    env.set_timing(True)
    for t in range(550, 570):
        env.step(device_ptr=cmds[t].data_ptr(), render=False)
    env.sync()
    tm = env.get_timing()
    print('N', n, 'RR_COOP_ALL', os.environ.get('RR_COOP_ALL'), round(n * 400 / el), 'env-steps/s', round(el / 400 * 1e3, 4), 'ms/step',
          {k: round(ms / max(c, 1), 4) for k, (ms, c) in tm.items() if c}, flush=True)
    env.close()
