"""Per-env phase cycles of k_collide on the bench workload (development; needs `make -C real_robots_amd/csrc stats`)."""
import os, sys, ctypes
sys.path.insert(0, '/root/repo')
os.environ['RR_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 4096
SCALE = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
lib = nat.load_library()
ids = list(range(N))
for t in range(170):
    env.step(synthetic_actions(ids, (t // 20) * 20, hold_prob=0.05) * SCALE, render=False)
torch.cuda.synchronize()
WAVES = 4
out = np.zeros((N, WAVES, 12), np.uint32)
assert lib.rr_debug_collide_prof(out.ctypes.data_as(ctypes.c_void_p), N) == 0
cnt = out[:, :, 7:12].sum(1).astype(np.float64)
print('per env: close pairs %.2f, apart in direction 0 %.2f, apart only in direction 1 %.2f, not apart but no candidate %.2f, with contacts %.2f' % tuple(cnt.mean(0)))
busy = out[:, :, 2:6].astype(np.float64).sum(2)
print('pair-loop cycles per wave: mean of waves %.0f, slowest wave %.0f (mean over envs)' % (busy.mean(), busy.max(1).mean()))
names = ['stage', 'sphere tests', 'loads + cull', 'prefilter', 'all-plane pass', 'reduction', 'record write']
wtot = out[:, :, :7].astype(np.float64).sum(2)          # per wave
slow = wtot.argmax(1)
cyc = out[np.arange(N), slow, :7].astype(np.float64)      # the env's slowest wave
tot = cyc.sum(1)
out = np.concatenate([out[np.arange(N), slow, :7], out[:, :, 7].sum(1)[:, None]], 1)
order = np.argsort(-tot)
print("per-env cycles: mean %.0f  median %.0f  p99 %.0f  max %.0f; close pairs mean %.1f max %d" % (tot.mean(), np.median(tot), np.percentile(tot, 99), tot.max(), out[:, 7].mean(), out[:, 7].max()))
print("%-16s %10s %10s %10s" % ("phase", "mean", "worst env", "p99 envs"))
top = order[:max(1, N // 100)]
for i, n in enumerate(names):
    print("%-16s %10.0f %10.0f %10.0f" % (n, cyc[:, i].mean(), cyc[order[0], i], cyc[top, i].mean()))
print("worst envs:", [(int(e), int(tot[e]), int(out[e, 7]), int(env.host(nat.F_CCOUNT)[e]) if hasattr(nat, 'F_CCOUNT') else 0) for e in order[:8]])
