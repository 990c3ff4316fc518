import os, sys
import numpy as np
sys.path.insert(0, '.')
os.environ['RR_LIB'] = os.path.abspath('real_robots_amd/csrc/librealrobot_hip_dbg.so')
os.environ['RR_NO_LOOKAHEAD'] = '1'
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
st = np.load('/tmp/st.npy'); a = np.load('/tmp/a.npy')
N = len(st)
env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
env.state = st
env.step(None)
b = env.host(nat.F_PREP)
np.set_printoptions(precision=5, linewidth=220, suppress=True)
print('per lane: mass axis(3) bax(3) bp(3) q\n', b[0, 165:286].reshape(11, 11))
print('scalar BAX\n', a[0, 132:165].reshape(11, 3), '\nBP\n', a[0, 99:132].reshape(11, 3), '\nq', st[0, :11])
