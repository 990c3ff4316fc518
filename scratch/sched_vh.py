"""Off-bench schedule checks: scratch/sched_vh.py <fraction of envs> [target height of the gripper base: 0.33 crushes it onto the table
(very heavy), 0.40 presses it (heavy)]; the automatic placement against a grid of forced readings of the two heavy counts."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from oracle.kinematics import inverse_kinematics, quat_from_euler
N = 4096
zt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.33       # 0.33: crushed (very heavy); 0.40: pressed (heavy)
press = inverse_kinematics(np.zeros(11), [-0.15, 0.25, zt], quat_from_euler(0, 3.14, -1.57))
press = np.concatenate([press[:7], [0.0, 0.0]]).astype(np.float32)
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
cmd = np.zeros((N, 9), np.float32)
cmd[np.arange(N) % 100 < int(frac * 100)] = press
cmd_dev = torch.from_numpy(cmd).cuda()

def ms_per_step(force):
    if force is None: os.environ.pop('RR_FORCE_HCOUNT', None)
    else: os.environ['RR_FORCE_HCOUNT'] = '%d,%d' % force
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
    for _ in range(200): env.step(device_ptr=cmd_dev.data_ptr(), render=True)
    best = 1e9
    for _ in range(3):
        env.sync(); t0 = time.perf_counter()
        for _ in range(120): env.step(device_ptr=cmd_dev.data_ptr(), render=True)
        env.sync(); best = min(best, (time.perf_counter() - t0) / 120 * 1e3)
    cls = env.host(nat.F_ENV_CLASS); env.close()
    return best, int((cls == 1).sum()), int((cls == 2).sum())

auto, nh, nvh = ms_per_step(None)
auto = min(auto, ms_per_step(None)[0])
forced = {}
for h in (0, 150, 600, 1300, 2600):
    for vh in (0, 30, 100, 400, 1200):
        forced[(h, vh)] = ms_per_step((h, vh))[0]
best = min(forced, key=forced.get)
print('frac %.2f: %d heavy, %d very heavy; automatic %.4f; best forced %.4f at %s; worst %.4f at %s; auto/best %.3f' % (frac, nh, nvh, auto, forced[best], best, max(forced.values()), max(forced, key=forced.get), auto / forced[best]))
print(sorted((round(v, 4), k) for k, v in forced.items())[:6])
