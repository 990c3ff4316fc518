"""320x240 at 4096 envs (the reference's default camera): 64x64 raster tiles against full-width strips (RR_TILE_W=320)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N = 4096
cmds = bench.make_commands(torch, np, np.arange(N), 300, 1.0, 'cuda:0')
for tw in (None, '128', '32', '40'):
    if tw: os.environ['RR_TILE_W'] = tw
    env = BatchedREALRobotEnv(N, objects=3, width=320, height=240, want_mask=False)
    for t in range(160):
        env.step(device_ptr=cmds[t].data_ptr(), render=(t >= 150))
    env.sync()
    t0 = time.perf_counter()
    for t in range(160, 220):
        env.step(device_ptr=cmds[t].data_ptr(), render=True)
    env.sync()
    el = (time.perf_counter() - t0) / 60
    env.set_timing(True)
    for t in range(220, 230):
        env.step(device_ptr=cmds[t].data_ptr(), render=True)
    env.sync()
    tm = env.get_timing()
    print('RR_TILE_W', tw, round(N / el), 'env-steps/s', round(el * 1e3, 4), 'ms', {k: round(ms / max(n, 1), 4) for k, (ms, n) in tm.items() if n and k in ('k_raster', 'k_shade', 'render_heavy')}, flush=True)
    env.close()
