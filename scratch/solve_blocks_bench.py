"""Headline workload (random resample-and-hold commands x 0.5): the slowest solver workgroup of each step and its envs'
contact population (development build)."""
import os, sys, ctypes
sys.path.insert(0, '/root/repo')
os.environ['RR_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
lib = nat.load_library()
ids = np.arange(N)
out = (ctypes.c_uint * (5 * (N // 4)))()
hist = []
for t in range(420):
    if t % 20 == 0: cmd = torch.from_numpy(synthetic_actions(ids, t, hold_prob=0.05) * 0.5).cuda()
    env.step(device_ptr=cmd.data_ptr(), render=False)
    if t >= 170:
        env.sync(); lib.rr_debug_solver_blocks(out, N // 4)
        a = np.frombuffer(out, dtype=np.uint32).reshape(N // 4, 5).astype(np.int64)
        b = int(a[:, 0].argmax()); d = a[b, 1:]
        hist.append((int(a[b, 0]), int(np.median(a[:, 0])), [(int(x & 255), int((x >> 8) & 255)) for x in d]))
cyc = np.array([h[0] for h in hist]); med = np.array([h[1] for h in hist])
print("steps", len(hist), "slowest block cycles: median %d p90 %d max %d; median block %d" % (np.median(cyc), np.percentile(cyc, 90), cyc.max(), np.median(med)))
slow = [h for h in hist if h[0] > 1.5 * np.median(cyc)]
print("steps whose slowest block is > 1.5x the typical slowest:", len(slow))
for h in sorted(slow, key=lambda h: -h[0])[:12]:
    print("  cycles %8d  (nc, generic) per env: %s" % (h[0], h[2]))
