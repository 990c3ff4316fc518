import sys; sys.path.insert(0, '/root/repo')
import numpy as np, faulthandler
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
env = BatchedREALRobotEnv(8, objects=3, width=128, height=128)
print('created', flush=True)
for t in range(3):
    env.step(np.zeros((8, 9), np.float32), render=False)
    print('step', t, env.host(nat.F_JOINTS)[0, :3], flush=True)
