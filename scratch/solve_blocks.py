"""Which solver workgroups are slow during a macro-action episode (development build): cycles per block against the
contact population of its four envs."""
import os, sys, ctypes
sys.path.insert(0, '/root/repo')
os.environ['RR_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
lib = nat.load_library()
rng = np.random.default_rng(0)
m = rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2))
env.plan_macro(m)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 600
for t in range(T): env.step_plan(render=False)
env.sync()
out = (ctypes.c_uint * (5 * (N // 4)))()
assert lib.rr_debug_solver_blocks(out, N // 4) == 0
a = np.array(list(out), dtype=np.int64).reshape(N // 4, 5)
cyc = a[:, 0]
order = np.argsort(-cyc)
print("block cycles: mean %.0f median %.0f p90 %.0f p99 %.0f max %d" % (cyc.mean(), np.median(cyc), np.percentile(cyc, 90), np.percentile(cyc, 99), cyc.max()))
for b in order[:12]:
    d = a[b, 1:]
    print("block %4d cycles %8d  " % (b, cyc[b]) + "  ".join("nc %2d gen %2d ovf %2d lean %2d" % (x & 255, (x >> 8) & 255, (x >> 16) & 255, (x >> 24) & 255) for x in d))
gen = (a[:, 1:] >> 8) & 255; ovf = (a[:, 1:] >> 16) & 255
print("envs with overflow contacts:", int((ovf > 0).sum()), " max generic per block -> mean cycles:")
mg = gen.max(1)
for lo, hi in ((0, 1), (1, 5), (5, 13), (13, 21), (21, 29), (29, 49)):
    sel = (mg >= lo) & (mg < hi)
    if sel.any(): print("   max generic in [%2d,%2d): %4d blocks, mean %8.0f cycles, max %8d" % (lo, hi, sel.sum(), cyc[sel].mean(), cyc[sel].max()))
