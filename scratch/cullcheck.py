import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle.oracle import Oracle
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 32
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
o = Oracle(3, 128, 128)
tot = 0; worst = 0
for t in range(300):
    env.step(synthetic_actions(range(N), t, seed=3) * 0.6, render=(t % 100 == 99))
    if t % 100 == 99:
        st = env.state; rgb = env.host(nat.F_RGB); msk = env.host(nat.F_MASK)
        for i in range(N):
            o.state = st[i].astype(np.float64); r, d, m = o.render()
            fr = (np.abs(r.astype(int) - rgb[i].astype(int)).max(-1) > 2).mean(); mm = (m != msk[i]).mean()
            worst = max(worst, fr, mm); tot += fr
print('mean rgb mismatch frac', tot / (3 * N), 'worst', worst)
# find worst frame
worst=(0,None)
env2 = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
for t in range(300):
    env2.step(synthetic_actions(range(N), t, seed=3) * 0.6, render=(t % 100 == 99))
    if t % 100 == 99:
        st = env2.state; rgb = env2.host(nat.F_RGB); msk = env2.host(nat.F_MASK); dep = env2.host(nat.F_DEPTH)
        for i in range(N):
            o.state = st[i].astype(np.float64); r, d, m = o.render()
            mm = (m != msk[i]).mean()
            if mm > worst[0]: worst = (mm, (t, i, r.copy(), rgb[i].copy(), m.copy(), msk[i].copy(), d.copy(), dep[i].copy(), st[i].copy()))
mm, (t, i, r, g, m, mg, d, dg, sti) = worst
print('worst', mm, 't', t, 'env', i)
from PIL import Image
Image.fromarray(np.concatenate([r, g, (np.abs(r.astype(int)-g.astype(int)).max(-1)>2).astype(np.uint8)[...,None].repeat(3,-1)*255],1)).save('/root/repo/gpurun_out/worst.png')
bad = np.argwhere(m != mg)
print('bad pixels', len(bad), 'rows', bad[:,0].min(), bad[:,0].max(), 'cols', bad[:,1].min(), bad[:,1].max())
print('oracle ids', np.unique(m[m!=mg], return_counts=True), 'gpu ids', np.unique(mg[m!=mg], return_counts=True))
print('depth oracle/gpu at bad', d[m!=mg][:5], dg[m!=mg][:5])
print('q', sti[:11])
