"""Distribution of generic contact rows per env over a long run of the bench workload (development)."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 1024
env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
ids = list(range(N))
for t in range(4001):
    env.step(synthetic_actions(ids, (t // 20) * 20, hold_prob=0.05), render=False)
    if t in (200, 370, 1000, 2000, 4000):
        ng = []
        for i in range(0, N, 2):
            c = env.contacts(i)
            if not len(c): ng.append(0); continue
            os_ = (c[:, 0] >= 16) & (c[:, 1] < 0)
            g = int((~os_).sum())
            for o in range(3):
                g += max(0, int((os_ & (c[:, 0] == 16 + o)).sum()) - 4)
            ng.append(g)
        ng = np.array(ng)
        print(t, 'heavy frac %.2f' % (ng > 0).mean(), 'hist', np.histogram(ng, bins=[0, 1, 3, 6, 10, 16, 24, 49])[0], 'mean', ng.mean().round(2), 'p90', np.percentile(ng, 90), 'max', ng.max())
