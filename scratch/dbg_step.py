"""One-step differential triage of k_collide / k_solve against the float oracle in a few canned scenarios."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from oracle.oracle import Oracle
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
np.set_printoptions(precision=6, suppress=True, linewidth=220)

def report(tag, env, o, i, st0, cmd):
    o.state = st0[i].astype(np.float64); o.step(cmd.astype(np.float64))
    st1 = env.state[i]; cd, co = env.contacts(i), o.contacts()
    keep = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11]
    same = len(cd) == len(co) and (len(cd) == 0 or (cd[:, keep] == co[:, keep].astype(np.float32)).all())
    dj = np.abs(st1[:22] - o.state[:22]).max(); do = np.abs(st1[22:] - o.state[22:]).max()
    print("%-28s contacts dev %2d orc %2d list-identical %s | d(q,qd) %.2e d(obj) %.2e | force diff %.3g of %.3g" % (
        tag, len(cd), len(co), same, dj, do, np.abs(cd[:, 10] - co[:, 10]).max() if same and len(cd) else -1, co[:, 10].max() if len(co) else 0))
    if not same and len(cd) == len(co):
        bad = np.flatnonzero((cd[:, keep] != co[:, keep].astype(np.float32)).any(1))
        print("   first differing contact", bad[0], "\n   dev", cd[bad[0]], "\n   orc", co[bad[0]].astype(np.float32))
    return dj, do

nobj = 3
env = BatchedREALRobotEnv(5, objects=nobj, width=64, height=64); o = Oracle(nobj, 64, 64, f32=True)
z = np.zeros((5, 9), np.float32)
for t in range(60):
    st0 = env.state; env.step(z)
    if t in (0, 20, 40, 59): report("rest t=%d" % t, env, o, 1, st0, z[1])
# arm pressed on the table: full-range command
cmd = np.tile(np.array([0.3, 1.6, 0, -1.2, 0, 0.8, 0, 0.5, 0.2], np.float32), (5, 1))
for t in range(140):
    st0 = env.state; env.step(cmd)
    if t % 20 == 19: report("arm down t=%d" % t, env, o, 2, st0, cmd[2])
# macro pushing
env.reset(); env.plan_macro(np.tile(np.array([[-0.1, -0.15], [-0.1, 0.3]]), (5, 1, 1))); plan = env.get_plan(0)
for t in range(420):
    st0 = env.state; env.step_plan()
    if t >= 240 and t % 20 == 0: report("push t=%d" % t, env, o, 3, st0, plan[t])
print("errflags", env.host(nat.F_ERRFLAGS))
