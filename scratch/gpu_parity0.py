import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle.oracle import Oracle
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd import _native as nat
np.set_printoptions(precision=5, suppress=True, linewidth=200)
N = 64
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
rng = np.random.default_rng(0)
orc = [Oracle(3, 128, 128, f32=(i % 2 == 1)) for i in range(4)]
lo = np.array([-2.09, -2.09, -2.09, -2.09, -2.09, -2.09, -2.09, 0, 0]); hi = np.array([2.09] * 7 + [1.57, 1.57])
act = rng.uniform(lo, hi, size=(N, 9))
for t in range(300):
    if t % 40 == 0:
        act = rng.uniform(lo, hi, size=(N, 9)) * 0.5
    env.step(act.astype(np.float32), render=False)
    for i, o in enumerate(orc):
        o.step(act[i].astype(np.float32).astype(np.float64))
    if t % 50 == 49:
        st = env.state
        for i, o in enumerate(orc):
            d = np.abs(st[i] - o.state)
            print(t, i, 'max|dq| %.2e max|dqd| %.2e objpos %.2e %.2e %.2e objquat %.2e' % (d[:11].max(), d[11:22].max(), d[22:25].max(), d[35:38].max(), d[48:51].max(), max(d[25:29].max(), d[38:42].max(), d[51:55].max())))
print('touch gpu', env.host(nat.F_TOUCH)[:4]); print('touch orc', [o.obs()[1] for o in orc])
env.render()
rgb = env.host(nat.F_RGB); dep = env.host(nat.F_DEPTH); msk = env.host(nat.F_MASK)
for i, o in enumerate(orc):
    o.state = st[i].astype(np.float64)
    r, d, m = o.render()
    print('img', i, 'rgb mismatch frac', (np.abs(r.astype(int) - rgb[i].astype(int)).max(-1) > 2).mean(), 'mask mismatch', (m != msk[i]).mean(), 'depth max diff', np.abs(d - dep[i]).max())
from PIL import Image
Image.fromarray(rgb[0]).save('/root/repo/gpurun_out/gpu_r0.png')
# timing
env.set_timing(1)
for t in range(20): env.step(act.astype(np.float32), render=True)
print(env.get_timing())
