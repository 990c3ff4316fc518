import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle.oracle import Oracle
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
q=np.array([-0.00481712,0.44336477,-0.23572378,-0.10927426,0.3090867,0.31713215,-0.13455392,0.3776641,-0.3915679,0.37826854,-0.39188972])
env = BatchedREALRobotEnv(2, objects=3, width=128, height=128)
o = Oracle(3, 128, 128)
st = env.state; st[:, :11] = q; env.state = st
o.state = env.state[0].astype(np.float64)
env.render()
m = env.host(nat.F_MASK)[0]; d = env.host(nat.F_DEPTH)[0]
r, do, mo = o.render()
print('gpu robot px', (m == 0).sum(), 'oracle', (mo == 0).sum(), 'mismatch', (m != mo).sum())
print('gpu min depth', d.min(), 'oracle', do.min())
print('link poses diff', np.abs(env.link_poses()[0][:, :3] - np.array([o.link_pose(i)[:3] for i in range(17)])).max())
bad = np.argwhere(m != mo)
print(bad[:10], do[m != mo][:10])
# histogram of oracle depth in robot pixels: bad vs good
print('oracle depth of bad px: min %.4f max %.4f' % (do[m != mo].min(), do[m != mo].max()))
good = (m == mo) & (mo == 0)
print('oracle depth of good robot px: min %.4f' % do[good].min(), 'gpu', d[good].min())
