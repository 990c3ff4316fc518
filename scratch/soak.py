"""T (10 000) steps x 4096 envs with a render every step, random masked resets and teleports: error flags, finiteness, heavy counts."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
import importlib.util
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = 4096
cmds = bench.make_commands(torch, np, np.arange(N), 2000, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
rng = np.random.default_rng(0)
t0 = time.perf_counter()
for t in range(T):
    if t % 500 == 250:
        env.reset((rng.random(N) < 0.1).astype(np.uint8))
    if t % 700 == 350:
        p = env.host(nat.F_OBJ_POSE); p[:, rng.integers(0, 3), 2] += 0.1
        env.set_object_poses(p, (rng.random(N) < 0.05).astype(np.uint8))
    env.step(device_ptr=cmds[t % 2000].data_ptr(), render=True)
    if t % 1000 == 999:
        ef = env.host(nat.F_ERRFLAGS); cls = env.host(nat.F_ENV_CLASS); st = env.state
        print('step %5d: %.3f ms/step so far; error flags set in %d envs (render overflow bit in %d); heavy %d, very heavy %d; state finite: %s; contacts mean %.2f'
              % (t + 1, (time.perf_counter() - t0) / (t + 1) * 1e3, int(((ef & np.uint32(0xFFFFFFF7)) != 0).sum()), int(((ef & np.uint32(8)) != 0).sum()), int((cls == 1).sum()), int((cls == 2).sum()),
                 bool(np.isfinite(st).all()), float(env.host(nat.F_CONTACT_COUNT).mean())), flush=True)
assert ((env.host(nat.F_ERRFLAGS) & np.uint32(0xFFFFFFF7)) == 0).all() and np.isfinite(env.state).all()
print('soak ok: %d steps x %d envs' % (T, N))
