import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
for (N, W, H) in ((1024, 320, 240), (4096, 320, 240), (4096, 64, 64), (1024, 128, 128)):
    env = BatchedREALRobotEnv(N, objects=3, width=W, height=H, want_mask=False)
    ids = np.arange(N); cache = {}
    def act(t):
        k = t // 20
        if k not in cache: cache[k] = torch.from_numpy(synthetic_actions(ids, k * 20, hold_prob=0.05) * 0.5).cuda()
        return cache[k]
    for t in range(0, 240, 20): act(t)       # commands precomputed outside the timed loop
    for t in range(170): env.step(device_ptr=act(t).data_ptr(), render=(t > 160))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(170, 230): env.step(device_ptr=act(t).data_ptr(), render=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
    env.set_timing(1)
    for t in range(230, 240): env.step(device_ptr=act(t).data_ptr(), render=True)
    tm = env.get_timing(); env.set_timing(0)
    print('%d envs %dx%d: %.3f ms/step, %.2f M env-steps/s, %s' % (N, W, H, dt * 1e3, N / dt / 1e6, {k[2:]: round(ms / max(n, 1), 3) for k, (ms, n) in tm.items()}), 'frags/env %.0f' % (env.host(nat.F_FRAG_COUNT).sum() / N))
    env.close()
