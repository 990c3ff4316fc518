"""Contact population by type during macro-action episodes (robot-static / robot-object / object-object / object-static)."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from collections import Counter
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
N = 1024
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
rng = np.random.default_rng(0)
lo, hi = np.array([-0.25, -0.5]), np.array([0.05, 0.5])
m = rng.uniform(lo, hi, size=(N, 2, 2))
env.plan_macro(m)
for t in range(1000):
    env.step_plan(render=False)
    if t in (150, 250, 450, 700, 950):
        rs = []; ro = []; oo = []; os_ = []; links = Counter(); pos = []; act = []
        for i in range(0, N, 2):
            c = env.contacts(i)
            if not len(c): rs.append(0); ro.append(0); oo.append(0); os_.append(0); continue
            a, b = c[:, 0].astype(int), c[:, 1].astype(int)
            isrob = (a >= 0) & (a < 16)
            r_s = isrob & (b < 0); r_o = (isrob & (b >= 16)) | ((a >= 16) & (b >= 0) & (b < 16)); o_o = (a >= 16) & (b >= 16); o_s = (a >= 16) & (b < 0)
            rs.append(r_s.sum()); ro.append(r_o.sum()); oo.append(o_o.sum()); os_.append(o_s.sum())
            for k in np.nonzero(r_s)[0]: links[(int(c[k, 2]), int(b[k]))] += 1
            pos += list(c[r_s, 9]); act += list(c[r_s, 10] > 0)
        rs, ro, oo, os_ = map(np.array, (rs, ro, oo, os_))
        print("t", t, "mean RS %.1f RO %.1f OO %.1f OS %.1f | max RS %d RO %d OO %d OS %d | envs RS>14: %d, RS+RO>14: %d, total>32: %d of %d" %
              (rs.mean(), ro.mean(), oo.mean(), os_.mean(), rs.max(), ro.max(), oo.max(), os_.max(), (rs > 14).sum(), (rs + ro > 14).sum(), (rs + ro + oo + os_ > 32).sum(), len(rs)))
        print("   RS hist", np.bincount(rs // 4)[:12], "(bins of 4)  RS dist>0: %.0f%%  RS with force: %.0f%%" % (100 * np.mean(np.array(pos) > 0), 100 * np.mean(act)))
        print("   RS (link, static shape) top:", links.most_common(12))
