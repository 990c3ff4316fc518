import os, sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
for N, pool in ((4095, None), (1021, "3000"), (131, "1500")):
    if pool: os.environ['RR_SOLVER_POOL'] = pool
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64, want_mask=False)
    os.environ.pop('RR_SOLVER_POOL', None)
    ref = BatchedREALRobotEnv(N, objects=3, width=64, height=64, want_mask=False)
    rng = np.random.default_rng(4)
    m = rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2))
    env.plan_macro(m); ref.plan_macro(m)
    worst = 0.0
    for t in range(500):
        if t % 50 == 0:                       # resynchronise: compare one-step results from identical states
            ref.state = env.state
        env.step_plan(render=False); ref.step_plan(render=False)
        if t % 50 == 0:
            worst = max(worst, float(np.abs(env.state - ref.state).max()))
    print(N, pool, "timesteps ok", bool((env.host(nat.F_TIMESTEP) == 500).all()), "errflags", int((env.host(nat.F_ERRFLAGS) != 0).sum()),
          "worst one-step difference to the default pool from the same state %.2e" % worst, flush=True)
    env.close(); ref.close()
