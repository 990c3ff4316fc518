"""Development build (make -C real_robots_amd/csrc stats): how long does a solver wave last against the generic contacts of its
four envs -- and against WHO is in them (robot or objects only)?  Headline workload at step argv[1] (default 2100), all envs in one
launch (a step without camera), four envs to a wave."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('RR_LIB', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so'))
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N, T = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 2100
cmds = bench.make_commands(torch, np, np.arange(N), T + 1, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(N, objects=3, width=32, height=32, want_mask=False)
lib = nat.load_library()
for t in range(T): env.step(device_ptr=cmds[t].data_ptr(), render=False)
env.sync()
nb = N // 4
buf = (ctypes.c_uint * (8 * nb))()
lib.rr_debug_solver_blocks(buf, nb)
a = np.array(list(buf), dtype=np.int64).reshape(nb, 8)
cyc = a[:, 0]
ng = (a[:, 4:8] >> 8) & 255
# who is in the generic rows of the LAST solved step's list?  (the contact list of the step just solved)
robot_gen = np.zeros(N, bool)
for i in np.flatnonzero(ng.reshape(-1) > 0):
    ct = env.contacts(int(i))
    ab = ct[:, :2].astype(int)
    robot_gen[i] = bool((((ab >= 0) & (ab < 16)).any(1)).any())
wave_robot = robot_gen.reshape(nb, 4).any(1)
mx = ng.max(1)
print('step %d: waves %d, with generic rows %d (robot in them: %d)' % (T, nb, (mx > 0).sum(), wave_robot.sum()))
for lo, hi in ((0, 0), (1, 2), (3, 4), (5, 8), (9, 16), (17, 48)):
    for rob in (False, True):
        m = (mx >= lo) & (mx <= hi) & (wave_robot == rob)
        if m.any(): print('  max generic %2d..%2d %-12s waves %4d  cycles median %7d  p90 %7d  max %7d' % (lo, hi, 'robot' if rob else 'objects only', m.sum(), np.median(cyc[m]), np.percentile(cyc[m], 90), cyc[m].max()))
env.close()
