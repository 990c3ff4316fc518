"""N beyond k_balance's 64 x 1024 bit-mask range (identity order path): stepping works and matches a small batch."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 70001
big = BatchedREALRobotEnv(N, objects=3, width=64, height=48, want_mask=False)
small = BatchedREALRobotEnv(8, objects=3, width=64, height=48, want_mask=False)
for t in range(120):
    a = synthetic_actions(range(8), t, seed=1)
    big.step(np.tile(a, (N // 8 + 1, 1))[:N], render=(t % 40 == 39))
    small.step(a, render=(t % 40 == 39))
sb, ss = big.state, small.state
print("errflags", int((big.host(nat.F_ERRFLAGS) != 0).sum()), "timesteps ok", bool((big.host(nat.F_TIMESTEP) == 120).all()))
print("first 8 envs equal small batch:", bool((sb[:8] == ss).all()), " last 8-block equal:", bool((sb[70000 - 8:70000] == ss).all()))
print("images equal:", bool((big.host(nat.F_RGB)[:8] == small.host(nat.F_RGB)).all()))
