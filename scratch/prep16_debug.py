import os, sys
import numpy as np
sys.path.insert(0, '.')
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
names = [('BR', 0, 99), ('BP', 99, 132), ('BAX', 132, 165), ('MINV', 165, 286), ('QDS', 286, 297), ('OR', 297, 324), ('OIINV', 324, 351), ('OVS', 351, 360), ('OWS', 360, 369), ('OP', 369, 378)]
rng = np.random.default_rng(0)
os.environ['RR_PREP_SCALAR'] = '1'
e0 = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
for t in range(100):
    e0.step(rng.uniform(-1, 1, (N, 9)).astype(np.float32))
st = e0.state
e0.close()
recs = {}
for name, envv in (('scalar', {'RR_PREP_SCALAR': '1', 'RR_NO_LOOKAHEAD': '1'}), ('p16', {'RR_PREP_SCALAR': '0', 'RR_NO_LOOKAHEAD': '1'})):
    os.environ.update(envv)
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    env.state = st
    env.step(None)
    recs[name] = env.host(nat.F_PREP)
    env.close()
a, b = recs['scalar'], recs['p16']
for nm, lo, hi in names:
    d = np.abs(a[:, lo:hi].astype(np.float64) - b[:, lo:hi])
    print(nm, 'nan a/b', np.isnan(a[:, lo:hi]).sum(), np.isnan(b[:, lo:hi]).sum(), 'max diff', np.nanmax(d) if d.size else 0, 'max |a|', np.nanmax(np.abs(a[:, lo:hi])))
np.set_printoptions(precision=5, linewidth=220, suppress=True)
print('MINV scalar env0\n', a[0, 165:286].reshape(11, 11))
print('MINV p16 env0\n', b[0, 165:286].reshape(11, 11))
print('QDS', a[0, 286:297], '\n   ', b[0, 286:297])
# now the look-ahead form
os.environ.pop('RR_NO_LOOKAHEAD')
os.environ['RR_PREP_SCALAR'] = '1'
rl = {}
for name, v in (('scalar', '1'), ('p16', '0')):
    os.environ['RR_PREP_SCALAR'] = v
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    env.state = st
    os.environ['RR_PREP_SCALAR'] = v
    env.step(None)
    rl[name] = env.host(nat.F_PREP)
    env.close()
print('look-ahead form (note: the step before it used different M^-1 -> states differ slightly)')
for nm, lo, hi in names:
    d = np.abs(rl['scalar'][:, lo:hi].astype(np.float64) - rl['p16'][:, lo:hi])
    print(nm, 'nan', np.isnan(rl['p16'][:, lo:hi]).sum(), 'max diff', np.nanmax(d))
