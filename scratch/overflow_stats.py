import sys; sys.path.insert(0, '/root/repo')
import numpy as np
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
for t in range(190): env.step(act(t))
env.set_timing(1)
for t in range(190, 225):
    env.step(act(t), render=False)
    tm = env.get_timing()
    ms = tm['k_solve'][0] / max(1, tm['k_solve'][1])
    if ms > 0.15 or t % 10 == 0:
        worst = (0, 0, 0, 0, -1); n_gen = 0
        for i in range(N):
            c = env.contacts(i)
            if len(c) <= 12 and not ((c[:, 0] < 16).any() if len(c) else False): continue
            rob = int(((c[:, 0] >= 0) & (c[:, 0] < 16)).sum()); oo = int(((c[:, 0] >= 16) & (c[:, 1] >= 16)).sum())
            n_gen += 1
            if (rob, oo, len(c)) > worst[:3]: worst = (rob, oo, len(c), 0, i)
        print('t %d k_solve %.3f ms | envs with generic contacts %d | worst env %d: robot %d objobj %d nc %d' % (t, ms, n_gen, worst[4], worst[0], worst[1], worst[2]), flush=True)
