import os, sys
import numpy as np
sys.path.insert(0, '.')
N = 3
rng = np.random.default_rng(0)
os.environ['RR_PREP_SCALAR'] = '1'
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
e0 = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
for t in range(100):
    e0.step(rng.uniform(-1, 1, (N, 9)).astype(np.float32))
st = e0.state
os.environ['RR_NO_LOOKAHEAD'] = '1'
env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
env.state = st
env.step(None)
a = env.host(nat.F_PREP)
env.close(); e0.close()
np.save('/tmp/st.npy', st); np.save('/tmp/a.npy', a)
