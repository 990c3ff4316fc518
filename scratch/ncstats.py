import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
ids = np.arange(N)
cache = {}
for t in range(300):
    key = t // 20
    if key not in cache: cache[key] = synthetic_actions(ids, key * 20, hold_prob=0.05) * 0.5
    env.step(cache[key])
    if t in (160, 180, 200, 220, 240, 259, 280, 299):
        ncs, nrob, nact = [], [], []
        for i in range(0, N, 17):
            c = env.contacts(i)
            ncs.append(len(c)); nrob.append(int(((c[:, 0] >= 0) & (c[:, 0] < 16)).sum()) if len(c) else 0)
            nact.append(int((c[:, 10] > 0).sum()) if len(c) else 0)
        ncs = np.array(ncs); print(t, 'nc mean %.1f max %d p90 %d | robot-involved mean %.1f max %d | active mean %.1f' % (ncs.mean(), ncs.max(), np.percentile(ncs, 90), np.mean(nrob), np.max(nrob), np.mean(nact)))
        g = ncs[:len(ncs)//4*4].reshape(-1, 4).max(1); print('   max over groups of 4 (wave): mean %.1f' % g.mean())
