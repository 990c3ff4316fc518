"""Trajectory generated with RR_SOLVER_POOL=<p> (rows in global memory); at step T the same state is stepped once by that env,
by a default env and by the fp32 oracle."""
import os, sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from oracle.oracle import Oracle
N = 34
T, I, pool = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ['RR_SOLVER_POOL'] = pool
env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
del os.environ['RR_SOLVER_POOL']
rng = np.random.default_rng(5)
env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
plans = [env.get_plan(i) for i in range(N)]
for t in range(T): env.step_plan(render=False)
st0 = env.state
ref = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
ref.state = st0
cmd = np.stack([plans[i][T] for i in range(N)]).astype(np.float32)
env.step(cmd); ref.step(cmd)
o = Oracle(3, 64, 64, f32=True); o.state = st0[I].astype(np.float64); o.step(plans[I][T].astype(np.float64))
a, b, r = env.state[I], ref.state[I], o.state
np.set_printoptions(precision=6, suppress=True, linewidth=200)
print("pool env vs oracle   joints %.2e objects %.2e" % (np.abs(a[:22] - r[:22]).max(), np.abs(a[22:] - r[22:]).max()))
print("default  vs oracle   joints %.2e objects %.2e" % (np.abs(b[:22] - r[:22]).max(), np.abs(b[22:] - r[22:]).max()))
for k in range(3):
    print(" object", k, "vel pool   ", a[22 + 13 * k + 7:22 + 13 * k + 13], "\n          vel default", b[22 + 13 * k + 7:22 + 13 * k + 13], "\n          vel oracle ", r[22 + 13 * k + 7:22 + 13 * k + 13])
c = env.contacts(I); c2 = ref.contacts(I); oc = o.contacts()
print("contacts", len(c), len(c2), len(oc))
for row, row2, row3 in zip(c, c2, oc):
    print("  A %3d B %3d link %2d dist %+.4f  force pool %9.3f  default %9.3f oracle %9.3f" % (row[0], row[1], row[2], row[9], row[10], row2[10], row3[10]))
print("object 2 pool   ", a[22 + 26:22 + 39])
print("object 2 default", b[22 + 26:22 + 39])
print("object 2 oracle ", r[22 + 26:22 + 39])
print("object 2 before ", st0[I][22 + 26:22 + 39])
