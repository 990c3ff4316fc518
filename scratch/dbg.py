import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd import _native as nat
mode = sys.argv[1]
env = BatchedREALRobotEnv(64, objects=3, width=128, height=128)
rng = np.random.default_rng(0)
lo = np.array([-2.09, -2.09, -2.09, -2.09, -2.09, -2.09, -2.09, 0, 0]); hi = np.array([2.09] * 7 + [1.57, 1.57])
act = rng.uniform(lo, hi, size=(64, 9)) * 0.5
for t in range(100):
    print('step', t, flush=True)
    sys.stderr.write('step %d\n' % t); sys.stderr.flush()
    if mode == 'zero': env.step(None)
    elif mode == 'arm': a = act.copy(); a[:, 7:] = 0; env.step(a.astype(np.float32))
    else: env.step(act.astype(np.float32))
    env.sync()
print('ok', mode, env.state[0][:11])
