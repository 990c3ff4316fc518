"""Per-kernel time over a macro-action episode (SURVEY 8d config 5 shape: 4096 envs, R1M3-style random macro actions,
one new macro action per 1000 steps, 128x128 render every step)."""
import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
rng = np.random.default_rng(0)
lo, hi = np.array([-0.25, -0.5]), np.array([0.05, 0.5])          # macro_space, env.py:49-52
for ep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    m = rng.uniform(lo, hi, size=(N, 2, 2))
    env.plan_macro(m)
    env.sync(); t0 = time.perf_counter()
    for t in range(1000):
        if t % 100 == 0:
            env.set_timing(1)
        env.step_plan(render=True)
        if t % 100 == 19:
            tm = env.get_timing(); env.set_timing(0)
            ngen = 0; worst = (0, 0)
            for i in range(0, N, 8):
                c = env.contacts(i)
                if len(c):
                    rob = int(((c[:, 0] >= 0) & (c[:, 0] < 16)).sum()); oo = int(((c[:, 0] >= 16) & (c[:, 1] >= 16)).sum())
                    ngen += (rob + oo) > 0; worst = max(worst, (rob + oo, len(c)))
            print(ep, t, {k[2:]: round(ms / max(n, 1), 3) for k, (ms, n) in tm.items() if n},
                  '| sampled(1/8) envs with generic contacts', ngen, 'worst generic/nc', worst, flush=True)
    env.sync(); dt = time.perf_counter() - t0
    print("episode %d: %.1f ms per step incl. timing passes, %.2f M env-steps/s" % (ep, dt, N * 1000 / dt / 1e6), flush=True)
assert (env.host(nat.F_ERRFLAGS) == 0).all()
