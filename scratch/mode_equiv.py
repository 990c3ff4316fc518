"""Stepping with per-kernel timing on (every kernel bracketed by events, single stream) must give bit for bit the same
states and images as the normal two-stream mode; so must a handle without a mask buffer."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 130
def run(timing, want_mask):
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=want_mask)
    env.set_timing(1 if timing else 0)
    rng = np.random.default_rng(1)
    env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
    for t in range(400):
        if t < 300: env.step_plan(render=(t % 3 == 0))
        else: env.step(synthetic_actions(range(N), t, seed=2), render=True)
    out = (env.state.copy(), env.host(nat.F_RGB).copy(), env.host(nat.F_DEPTH).copy())
    env.close()
    return out
a = run(False, True); b = run(True, True); c = run(False, False)
print("timing mode == normal mode:", all((x == y).all() for x, y in zip(a, b)))
print("no-mask handle == mask handle:", all((x == y).all() for x, y in zip(a, c)))
