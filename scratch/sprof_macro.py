"""Solver phase cycles of ONE workgroup during a macro-action episode (development build).  argv: step, block."""
import os, sys, ctypes
sys.path.insert(0, '/root/repo')
T = int(sys.argv[1]); BLK = int(sys.argv[2])
os.environ['RR_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so')
os.environ['RR_ABLATE'] = str((BLK << 16) | 0x4000)
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
lib = nat.load_library()
rng = np.random.default_rng(0)
env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
for t in range(T): env.step_plan(render=False)
out = (ctypes.c_ulonglong * 16)()
env.sync(); lib.rr_debug_solver_prof(out, 1)
env.step_plan(render=False)
env.sync(); lib.rr_debug_solver_prof(out, 0)
names = ['stage Minv', 'gather contacts + build rows', 'motor+limit rows', 'limmask + register rows', 'PGS iterations', 'integrate', 'touch/forces']
for n, x in zip(names, list(out)): print(f'{n:30s} {x:10d} cycles')
for i in range(4 * BLK, 4 * BLK + 4):
    c = env.contacts(i); a, b = c[:, 0].astype(int), c[:, 1].astype(int)
    print('env', i, 'nc', len(c), 'robot', int(((a >= 0) & (a < 16)).sum()), 'objobj', int(((a >= 16) & (b >= 16)).sum()))
