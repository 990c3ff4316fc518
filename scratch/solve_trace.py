import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
ids = np.arange(N); cache = {}
def act(t):
    k = t // 20
    if k not in cache: cache[k] = synthetic_actions(ids, k * 20, hold_prob=0.05) * 0.5
    return cache[k]
for t in range(160): env.step(act(t))
env.set_timing(1)
for t in range(160, 460):
    env.step(act(t), render=False)
    if t % 10 == 9:
        tm = env.get_timing()
        line = ' '.join('%s %.3f' % (k[2:], ms / n) for k, (ms, n) in tm.items() if n and k in ('k_solve', 'k_collide', 'k_prep'))
        nrob = 0; ncmax = 0
        for i in range(0, N, 8):
            c = env.contacts(i)
            if len(c):
                nrob += int(((c[:, 0] >= 0) & (c[:, 0] < 16)).any()); ncmax = max(ncmax, len(c))
        print(t, line, '| envs (of 512 sampled) with robot contacts:', nrob, 'max nc', ncmax, flush=True)
