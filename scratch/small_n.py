"""Device-side floor of the single-env step (BASELINE config 1): HIP-event time of every kernel of a step at N = 1 and 16, no camera /
camera, and the wall time per step of the batched API and of the gym facade."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
for N, w, h in ((1, 320, 240), (16, 320, 240)):
    env = BatchedREALRobotEnv(N, objects=1, width=w, height=h)
    for cam in (False, True):
        for t in range(50):
            env.step(synthetic_actions(range(N), t, seed=3).astype(np.float32), render=cam)
        env.sync()
        t0 = time.perf_counter()
        for t in range(50, 350):
            env.step(synthetic_actions(range(N), 50, seed=3).astype(np.float32), render=cam)
        env.sync()
        wall = (time.perf_counter() - t0) / 300
        env.set_timing(True)
        for t in range(350, 370):
            env.step(synthetic_actions(range(N), 50, seed=3).astype(np.float32), render=cam)
        env.sync()
        tm = env.get_timing()
        env.set_timing(False)
        print('N', N, 'camera', cam, 'wall us/step (async launches, one sync at the end)', round(wall * 1e6, 1),
              {k: round(ms / max(n, 1) * 1e3, 1) for k, (ms, n) in tm.items() if n}, 'us per kernel', flush=True)
    env.close()
import real_robots_amd as rr
e = rr.make('REALRobot2020-R2J1-v0')
e.reset()
a = np.zeros(9)
for cam in (False, True):
    for t in range(30):
        e.step({'joint_command': a, 'render': cam})
    n = 1000 if not cam else 200
    t0 = time.perf_counter()
    for t in range(n):
        e.step({'joint_command': a, 'render': cam})
    print('facade camera', cam, 'steps/s', round(n / (time.perf_counter() - t0), 1), flush=True)
e.close()
