"""Late window of the headline workload (steps 2000-2100, ~650 heavy envs: the heavy list is rendered by k_raster_list) and the macro
workload (steps 300-400): ms per step."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N = 4096
cmds = bench.make_commands(torch, np, np.arange(N), 2110, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
for t in range(2000):
    env.step(device_ptr=cmds[t].data_ptr(), render=(t >= 1990))
env.sync(); t0 = time.perf_counter()
for t in range(2000, 2100):
    env.step(device_ptr=cmds[t].data_ptr(), render=True)
env.sync(); print('late window ms/step', round((time.perf_counter() - t0) * 10, 4), flush=True)
env.close()
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
env.plan_macro(np.random.default_rng(0).uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
for t in range(300):
    env.step_plan(render=True)
env.sync(); t0 = time.perf_counter()
for t in range(100):
    env.step_plan(render=True)
env.sync(); print('macro ms/step', round((time.perf_counter() - t0) * 10, 4), flush=True)
env.close()
