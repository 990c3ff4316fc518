"""Development build (make -C real_robots_amd/csrc stats): cycles of every solver workgroup against the contact population
of its four envs, for the headline workload (full-range commands) or the macro workload (argv[1] == 'macro')."""
import os, sys, ctypes
sys.path.insert(0, '/root/repo')
os.environ.setdefault('RR_LIB', os.path.join(os.path.dirname(__file__), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so'))
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = int(os.environ.get('N', '4096'))
macro = len(sys.argv) > 1 and sys.argv[1] == 'macro'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
lib = nat.load_library()
ids = list(range(N))
if macro:
    env.plan_macro(np.random.default_rng(0).uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
    for t in range(420): env.step_plan()
else:
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    for t in range(T): env.step(synthetic_actions(ids, (t // 20) * 20, hold_prob=0.05) * scale, render=False)
torch.cuda.synchronize()
nb = N // 4 if N > 1024 else N          # (up to 1024 envs: one env per wave -- coop form)
buf = (ctypes.c_uint * (8 * nb))()
lib.rr_debug_solver_blocks(buf, nb)
a = np.array(list(buf), dtype=np.int64).reshape(nb, 8)
if N <= 1024: a = a[:min(nb, 4096)]
cyc, build, pgs = a[:, 0], a[:, 1], a[:, 2]
print("workgroups %d: total cycles mean %.0f median %.0f p99 %.0f max %.0f | build mean %.0f max %.0f | pgs mean %.0f max %.0f" % (
    nb, cyc.mean(), np.median(cyc), np.percentile(cyc, 99), cyc.max(), build.mean(), build.max(), pgs.mean(), pgs.max()))
order = np.argsort(-cyc)[:12]
for b in order:
    envs = [(int(x & 255), int((x >> 8) & 255), int((x >> 16) & 255), int(x >> 24)) for x in a[b, 4:8]]
    print("  wg %4d cycles %7d build %6d pgs %7d  envs (nc, generic, os, F-list): %s" % (b, cyc[b], build[b], pgs[b], envs))
ng = ((a[:, 4:8] >> 8) & 255)
print("generic contacts per env: mean %.2f, share of envs with any %.3f, max %d; waves with any %.3f" % (ng.mean(), (ng > 0).mean(), ng.max(), (ng.max(1) > 0).mean()))
