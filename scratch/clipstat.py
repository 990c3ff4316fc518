"""Development build: how many triangles cross the near plane in the bench workload, and what the clip pass costs."""
import os, sys, ctypes, time
sys.path.insert(0, '/root/repo')
os.environ['RR_LIB'] = os.path.join(os.path.dirname(__file__), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
lib = nat.load_library()
ids = list(range(N))
for t in range(200): env.step(synthetic_actions(ids, (t // 20) * 20) * scale, render=(t > 190))
env.set_timing(1)
for t in range(20): env.step(synthetic_actions(ids, 200) * scale, render=True)
tm = env.get_timing(); env.set_timing(0)
print('scale', scale, 'ablate', os.environ.get('RR_ABLATE'), {k: round(ms / max(n, 1), 4) for k, (ms, n) in tm.items() if n})
if int(os.environ.get('RR_ABLATE', '0')) & 32768:
    out = (ctypes.c_ulonglong * 16)()
    lib.rr_debug_raster_stats(out, 0)
    v = list(out)
    print('frames', v[13], 'clipped triangles per frame %.2f' % (v[14] / max(v[13], 1)), 'frames with any %.4f' % (v[15] / max(v[13], 1)))
