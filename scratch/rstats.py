"""Raster work counters on the bench workload (development; needs `make -C real_robots_amd/csrc stats`)."""
import os, sys, ctypes
os.environ['RR_ABLATE'] = os.environ.get('RR_ABLATE', '32768')      # enables the work counters of the development build
sys.path.insert(0, '/root/repo')
os.environ['RR_LIB'] = os.path.join(os.path.dirname(__file__), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
SC = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
lib = nat.load_library()
ids = list(range(N))
for t in range(160):
    env.step(synthetic_actions(ids, (t // 20) * 20, hold_prob=0.05) * SC, render=False)
out = (ctypes.c_ulonglong * 16)()
lib.rr_debug_raster_stats(out, 1)
env.step(synthetic_actions(ids, 160, hold_prob=0.05) * SC, render=True); env.sync() if hasattr(env, 'sync') else None
torch.cuda.synchronize()
lib.rr_debug_raster_stats(out, 0)
v = np.array(list(out), dtype=np.float64) / N
names = ['windows', 'windows after cluster cull', 'live tris', 'big tris', 'sum small area', 'sum window-max small area', 'windows with live', 'windows with small', 'sum big area', 'hier blocks rasterised', 'hier tris', 'clusters that own a pixel at the end', 'pixlist', 'blocks', 'wave cycles idle at loop-end barrier (sum over 16 waves)', 'wave cycles in the window loop (sum over 16 waves)']
for n, x in zip(names, v): print(f'{n:32s} {x:10.1f} per env')
