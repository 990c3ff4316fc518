import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd import _native as nat
from oracle.oracle import Oracle
np.set_printoptions(precision=5, suppress=True, linewidth=220)
env = BatchedREALRobotEnv(64, objects=3, width=128, height=128)
o = Oracle(3, 128, 128)
for t in range(7):
    env.step(None); o.step(None)
    s = env.state
    print(t, 'gpu q', s[0][:11]); print(t, 'orc q', o.state[:11])
    print(t, 'gpu qd', s[0][11:22]); print(t, 'orc qd', o.state[11:22])
    print(t, 'gpu obj0', s[0][22:35]); print(t, 'orc obj0', o.state[22:35])
    print('err', env.host(nat.F_ERRFLAGS)[:4], 'spread', np.abs(s - s[0]).max())
