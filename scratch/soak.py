"""Soak run (development): 4096 envs, full-range commands, render every step, random per-env resets and object teleports;
reports error flags / non-finite states and the contact population at the end."""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
N = 4096
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
ids = np.arange(N)
rng = np.random.default_rng(3)
cmds = {}
t0 = time.time()
for t in range(T):
    k = t // 20
    if k not in cmds:
        cmds = {k: torch.from_numpy(synthetic_actions(ids, k * 20, hold_prob=0.05)).cuda()}
    env.step(device_ptr=cmds[k].data_ptr(), render=True)
    if t % 500 == 499:
        env.reset((rng.random(N) < 0.05).astype(np.uint8))
        for _ in range(8):
            env.set_object_pose(int(rng.integers(0, N)), int(rng.integers(0, 3)), np.array([rng.uniform(-0.2, 0.0), rng.uniform(-0.3, 0.3), rng.uniform(0.3, 0.6), 0, 0, 0, 1], np.float32))
    if t % 2000 == 1999:
        ef = env.host(nat.F_ERRFLAGS)
        st = env.state
        print(t + 1, 'errflags nonzero', int((ef != 0).sum()), 'finite', bool(np.isfinite(st).all()), 'max |q|', float(np.abs(st[:, :11]).max()), 'obj z range', float(st[:, 24::13][:, :3].min()), float(st[:, 24::13][:, :3].max()), '%.1f s' % (time.time() - t0), flush=True)
rgb = env.host(nat.F_RGB)
print('image mean', float(rgb.mean()), 'nonwhite frac', float((rgb != 255).any(-1).mean()))
