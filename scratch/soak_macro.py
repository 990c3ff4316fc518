"""Macro-action soak: 4096 envs, 3 episodes of random macro actions (the gripper pushes the objects around), no env may
report a non-finite state, and two runs must agree bit for bit (k_balance's env order, the shared LDS row pool and the
per-env friction lists are deterministic)."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
N = 4096
def run():
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
    rng = np.random.default_rng(3)
    worst = 0
    for ep in range(3):
        env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
        for t in range(1000):
            env.step_plan(render=(t % 50 == 0))
        ef = env.host(nat.F_ERRFLAGS)
        st = env.state
        print("episode", ep, "errflags", int((ef != 0).sum()), "finite", bool(np.isfinite(st).all()),
              "obj z range %.3f..%.3f" % (st[:, 24::13][:, :3].min() if False else st[:, [24, 37, 50]].min(), st[:, [24, 37, 50]].max()), flush=True)
    st, rgb = env.state.copy(), env.host(nat.F_RGB).copy()
    env.close()
    return ef, st, rgb
ef, st, rgb = run()
ef2, st2, rgb2 = run()
print("reproducible state", bool((st == st2).all()), "images", bool((rgb == rgb2).all()), "errflags", int((ef != 0).sum()))
