import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle.oracle import Oracle
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
sys.path.insert(0, '/root/repo/tests')
from test_gpu_parity import _grasp_script
cmds = _grasp_script(np.zeros(11))
env = BatchedREALRobotEnv(4, objects=1, width=64, height=64)
o = Oracle(1, 64, 64, f32=True)
for _ in range(100):
    env.step(None); o.step(None)
for t, c in enumerate(cmds):
    env.step(np.tile(c.astype(np.float32), (4, 1))); o.step(c.astype(np.float32).astype(np.float64))
    st = env.state
    cont = env.contacts(0)
    nrob = int(((cont[:, 0] >= 0) & (cont[:, 0] < 16)).sum()) if len(cont) else 0
    if t > 265 or not np.isfinite(st).all():
        print(t, 'nc', len(cont), 'nrob', nrob, 'finite', np.isfinite(st).all(), 'err', env.host(nat.F_ERRFLAGS)[0], 'dq', np.abs(st[0][:11] - o.state[:11]).max(), 'touch', env.host(nat.F_TOUCH)[0], o.obs()[1])
    if not np.isfinite(st).all():
        break
