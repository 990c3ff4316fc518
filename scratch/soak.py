"""Soak: 4096 envs, aggressive random commands (full joint range), 3000 steps, render every 7th step; checks that no env
reports a non-finite state, objects stay in the workspace (or are re-posed), and the run is reproducible."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 4096
def run():
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
    ids = np.arange(N); cache = {}
    mx = 0
    for t in range(3000):
        k = t // 20
        if k not in cache: cache[k] = torch.from_numpy(synthetic_actions(ids, k * 20, hold_prob=0.05)).cuda()
        env.step(device_ptr=cache[k].data_ptr(), render=(t % 7 == 0))
        if t % 500 == 499:
            ef = env.host(nat.F_ERRFLAGS); st = env.state
            nrob = sum(int((env.contacts(i)[:, 0] < 16).any()) if len(env.contacts(i)) else 0 for i in range(0, N, 16))
            print(t, 'errflags', int((ef != 0).sum()), 'finite', bool(np.isfinite(st).all()), 'obj z range %.3f..%.3f' % (st[:, 22:].reshape(N, 3, 13)[:, :, 2].min(), st[:, 22:].reshape(N, 3, 13)[:, :, 2].max()), 'sampled envs with robot contacts', nrob, flush=True)
    st = env.state.copy(); rgb = env.host(nat.F_RGB).copy()
    env.close()
    return st, rgb
a, ra = run()
b, rb = run()
print('reproducible state', bool((a == b).all()), 'images', bool((ra == rb).all()))
