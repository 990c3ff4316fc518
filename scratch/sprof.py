"""Solver phase cycle stamps on the bench workload (development; needs `make -C real_robots_amd/csrc stats`)."""
import os, sys, ctypes
sys.path.insert(0, '/root/repo')
os.environ.setdefault('RR_LIB', os.path.join(os.path.dirname(__file__), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so'))
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = int(os.environ.get('N', '4096'))
SCALE = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
lib = nat.load_library()
ids = list(range(N))
T0 = int(sys.argv[2]) if len(sys.argv) > 2 else 160
for t in range(T0):
    env.step(synthetic_actions(ids, (t // 20) * 20, hold_prob=0.05) * SCALE, render=False)
out = (ctypes.c_ulonglong * 16)()
torch.cuda.synchronize()
lib.rr_debug_solver_prof(out, 1)
K = 10
for t in range(K):
    env.step(synthetic_actions(ids, (T0 // 20) * 20, hold_prob=0.05) * SCALE, render=False)
torch.cuda.synchronize()
lib.rr_debug_solver_prof(out, 0)
DIV = K if (int(os.environ.get('RR_ABLATE', '0')) & 0x4000) else K * N / 4
if 'per_sweep' in sys.argv: DIV *= 50
v = np.array(list(out), dtype=np.float64) / DIV
names = ['stage Minv', 'row build (+ block repack)', 'motor+limit rows', 'limmask + register rows', 'PGS iterations (rest)', 'sweep: generic blocks, lateral pass (new)', 'sweep: generic blocks, torsional pass (new)',
         'sweep: motors + limits', 'sweep: pass prologue + object-lane rows', 'sweep: to slots', 'sweep: generic row blocks (new: normal pass)', 'sweep: from slots + list build', '-', '-', '-', '-']
tot = v[:12].sum()
for n, x in zip(names, v): print(f'{n:28s} {x:10.0f} ticks  {100 * x / tot:5.1f} %')
print('total', tot, 'ticks (shader clock cycles)')
