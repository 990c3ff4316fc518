"""Solver phase cycles of the block that holds a robot-contact env (development build)."""
import os, sys, ctypes
sys.path.insert(0, '/root/repo')
BLK = 2536 // 4
os.environ['RR_LIB'] = os.path.join(os.path.dirname(__file__), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so')
os.environ['RR_ABLATE'] = str((BLK << 16) | 0x4000)
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
lib = nat.load_library()
ids = np.arange(N); cache = {}
def act(t):
    k = t // 20
    if k not in cache: cache[k] = synthetic_actions(ids, k * 20, hold_prob=0.05) * 0.5
    return cache[k]
for t in range(196): env.step(act(t))
out = (ctypes.c_ulonglong * 16)()
torch.cuda.synchronize(); lib.rr_debug_solver_prof(out, 1)
K = 4
for t in range(196, 196 + K): env.step(act(t))
torch.cuda.synchronize(); lib.rr_debug_solver_prof(out, 0)
c = env.contacts(2536); print('env 2536 contacts', len(c), 'robot', int((c[:, 0] < 16).sum()))
v = np.array(list(out), dtype=np.float64) / K
names = ['stage Minv', 'gather contacts + build rows', 'motor+limit rows', 'limmask + register rows', 'PGS iterations', 'integrate', 'touch/forces']
for n, x in zip(names, v): print(f'{n:30s} {x:10.0f} cycles')
