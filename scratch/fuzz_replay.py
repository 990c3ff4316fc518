"""Replays one case of tests/test_gpu_contacts_fuzz.py and prints a hash of the whole device state after every step (to compare
two builds of the library, RR_LIB=..., or two runs of one build)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions

case, seed0 = int(sys.argv[1]), 2
rng = np.random.default_rng(seed0 * 1000 + case)
N = int(rng.choice([1, 3, 5, 17, 34, 63, 130]))
nobj = int(rng.integers(1, 4))
W, H = [(64, 48), (64, 64), (128, 128), (160, 120), (320, 240)][int(rng.integers(0, 5))]
pool = rng.choice([None, None, "0", "900", "2500"])
if pool:
    os.environ['RR_SOLVER_POOL'] = str(pool)
env = BatchedREALRobotEnv(N, objects=nobj, width=W, height=H)
macro = rng.random() < 0.6
plans = None
if macro:
    env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
    plans = [env.get_plan(i) for i in range(N)]
scale = rng.choice([0.5, 0.8, 1.0])
T = int(rng.integers(60, 200))
t_off = int(rng.integers(0, 400)) if macro else 0
if macro and t_off:
    plans = [np.roll(p, -t_off, axis=0) for p in plans]
print("case", case, "N", N, "nobj", nobj, W, H, "pool", pool, "macro", macro, "T", T)
out = []
allst = []
for t in range(T):
    if macro:
        cmd = np.stack([plans[i][t % 1000] for i in range(N)]).astype(np.float32)
    else:
        cmd = (synthetic_actions(range(N), t, seed=case) * scale).astype(np.float32)
    if rng.random() < 0.01:
        env.reset((rng.random(N) < 0.3).astype(np.uint8))
    if rng.random() < 0.01:
        env.set_object_pose(int(rng.integers(0, N)), int(rng.integers(0, nobj)),
                            np.array([rng.uniform(-0.2, 0.0), rng.uniform(-0.3, 0.3), rng.uniform(0.3, 0.6), 0, 0, 0, 1], np.float32))
    flags = (rng.random(N) < 0.5).astype(np.uint8)
    chk = t % 20 == 19
    if chk:
        ncs = np.array([len(env.contacts(i)) for i in range(N)])
        sel = sorted(set(list(np.argsort(-ncs)[:2]) + [int(rng.integers(0, N))]))
        flags[sel] = 1
    env.step(cmd, render=flags if N > 1 else bool(flags[0]))
    st = env.state
    out.append(hashlib.md5(np.ascontiguousarray(st).tobytes()).hexdigest()[:10])
    allst.append(st.copy())
print(" ".join(out))
if len(sys.argv) > 2:
    np.save(sys.argv[2], np.stack(allst))
