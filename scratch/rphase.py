"""Where a raster wave spends its life: cycles between the phase marks of the window loop, summed over all waves (development
build: `make -C real_robots_amd/csrc stats`; RR_ABLATE=4096).  Bench workload, 4096 envs, one rendered frame."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['RR_LIB'] = os.environ.get('RR_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so'))
os.environ['RR_ABLATE'] = '4096'
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N = 4096
cmds = bench.make_commands(torch, np, np.arange(N), 200, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
lib = nat.load_library()
for t in range(170):
    env.step(device_ptr=cmds[t].data_ptr(), render=(t >= 165))
env.sync()
out = (ctypes.c_ulonglong * 16)()
lib.rr_debug_raster_phase(out, 1)
env.set_timing(True)
for t in range(170, 171):
    env.step(device_ptr=cmds[t].data_ptr(), render=True)
env.sync()
lib.rr_debug_raster_phase(out, 0)
v = np.array(list(out)[:10], dtype=np.float64)
names = ['window fetch', 'loads + projection + gather', 'bbox + set-up', 'in-lane points', 'scan', 'records + rounds', 'large triangles', 'last fetch', 'barrier wait', 'near-plane pass']
for n, x in zip(names, v):
    print('%-30s %6.1f %%   %8.0f cycles per (env, frame)' % (n, 100 * x / v.sum(), x / N))
print({k: round(ms / max(n, 1), 4) for k, (ms, n) in env.get_timing().items() if n and k in ('k_raster', 'k_shade')})
