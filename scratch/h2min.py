"""RR_HEAVY2_MIN off the bench: ms per step of a workload (fraction of envs pressing / crushing the gripper on the table) per threshold."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from oracle.kinematics import inverse_kinematics, quat_from_euler
N = 4096
frac, zt = float(sys.argv[1]), float(sys.argv[2])
press = inverse_kinematics(np.zeros(11), [-0.15, 0.25, zt], quat_from_euler(0, 3.14, -1.57))
cmd = np.zeros((N, 9), np.float32); cmd[np.arange(N) % 100 < int(frac * 100)] = np.concatenate([press[:7], [0.0, 0.0]]).astype(np.float32)
cmd_dev = torch.from_numpy(cmd).cuda()
for thr in (8, 12, 16, 24, 1000):
    os.environ['RR_HEAVY2_MIN'] = str(thr)
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
    for _ in range(200): env.step(device_ptr=cmd_dev.data_ptr(), render=True)
    best = 1e9
    for _ in range(3):
        env.sync(); t0 = time.perf_counter()
        for _ in range(120): env.step(device_ptr=cmd_dev.data_ptr(), render=True)
        env.sync(); best = min(best, (time.perf_counter() - t0) / 120 * 1e3)
    cls = env.host(nat.F_ENV_CLASS); env.close()
    print('frac %.2f z %.2f RR_HEAVY2_MIN %4d: %.4f ms  heavy %d very heavy %d' % (frac, zt, thr, best, (cls == 1).sum(), (cls == 2).sum()), flush=True)
