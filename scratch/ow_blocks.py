"""Development (stats build): start / end times (100 MHz counter) of every workgroup of k_solve_light_ow in one step of the bench workload."""
import ctypes
import os
import sys
sys.path.insert(0, '/root/repo')
os.environ['RR_LIB'] = os.environ.get('RR_STATS_LIB', os.path.join(os.path.dirname(__file__), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so'))
import numpy as np
import torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
lib = nat.load_library()
ids = list(range(N))
acts = {}
for t in range(220):
    k = t // 20
    if k not in acts:
        acts[k] = synthetic_actions(ids, k * 20)
    env.step(acts[k], render=True)
torch.cuda.synchronize()
nb = N // 16
buf = (ctypes.c_uint * (8 * 4096))()
lib.rr_debug_solver_blocks(buf, 4096)
a = np.array(list(buf), dtype=np.int64).reshape(4096, 8)[2048:2048 + nb]
st, en = a[:, 0], a[:, 1]
t0 = st.min()
en = st + ((en - st) % (1 << 32))
print("workgroups %d: start offsets (us) min %.1f median %.1f p90 %.1f max %.1f | object-wave duration mean %.1f max %.1f | last end %.1f us after the first start"
      % (nb, 0.0, np.median(st - t0) / 100, np.percentile(st - t0, 90) / 100, (st - t0).max() / 100, ((en - st) / 100).mean(), ((en - st) / 100).max(), (en.max() - t0) / 100))
d = (en - st) / 100.0
ok = d < 1000
print("object-wave durations (us) of the %d workgroups whose lane 0 stepped: min %.1f median %.1f p90 %.1f max %.1f; last end %.1f us after the first start"
      % (ok.sum(), d[ok].min(), np.median(d[ok]), np.percentile(d[ok], 90), d[ok].max(), (en[ok].max() - t0) / 100))
late = np.argsort(-(st - t0))[:10]
print("latest starts:", [(int(b), round(float(st[b] - t0) / 100, 1)) for b in late])
