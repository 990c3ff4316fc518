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
    if k not in cache: cache[k] = torch.from_numpy(synthetic_actions(ids, k * 20, hold_prob=0.05)).cuda()
    return cache[k]
for t in range(0, 700, 20): act(t)
for t in range(300): env.step(device_ptr=act(t).data_ptr())
env.set_timing(1)
for t in range(300, 600):
    env.step(device_ptr=act(t).data_ptr(), render=False)
    if t % 50 == 49:
        tm = env.get_timing()
        worst = (0, 0); ngen = 0
        for i in range(0, N, 4):
            c = env.contacts(i)
            if len(c):
                rob = int(((c[:, 0] >= 0) & (c[:, 0] < 16)).sum()); oo = int(((c[:, 0] >= 16) & (c[:, 1] >= 16)).sum())
                ngen += (rob + oo) > 0; worst = max(worst, (rob + oo, len(c)))
        print(t, {k[2:]: round(ms / max(n, 1), 3) for k, (ms, n) in tm.items() if n}, '| sampled(1/4) envs with generic contacts', ngen, 'worst generic/nc', worst, flush=True)
