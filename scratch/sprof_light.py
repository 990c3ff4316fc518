"""Phase cycle stamps of k_solve_light on the bench workload (development; needs `make -C real_robots_amd/csrc stats`).
One workgroup (= one wave = four envs) at a time: RR_ABLATE = block << 16 | 0x4000; blocks >= 300 have no counterpart in the
coop launches of the heavy classes (their lists are shorter than 1200), so the stamps are the light kernel's alone."""
import ctypes
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, '/root/repo')
    os.environ['RR_LIB'] = os.environ.get('RR_STATS_LIB', os.path.join(os.path.dirname(__file__), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so'))
    import numpy as np
    import torch
    from real_robots_amd import _native as nat
    from real_robots_amd.batched import BatchedREALRobotEnv
    from real_robots_amd.distributed import synthetic_actions
    N = 4096
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
    lib = nat.load_library()
    ids = list(range(N))
    T0 = 200
    acts = {}
    def act(t):
        k = t // 20
        if k not in acts:
            acts[k] = synthetic_actions(ids, k * 20)
        return acts[k]
    for t in range(T0):
        env.step(act(t), render=True)
    out = (ctypes.c_ulonglong * 16)()
    torch.cuda.synchronize()
    lib.rr_debug_solver_prof(out, 1)
    K = 20
    for t in range(K):
        env.step(act(T0 + t), render=True)
    torch.cuda.synchronize()
    lib.rr_debug_solver_prof(out, 0)
    v = np.array(list(out), dtype=np.float64) / K
    print(' '.join('%.0f' % x for x in v[:13]))
else:
    names = ['stage-in + command', 'row build', 'motor + limit rows', 'register rows', 'PGS sweeps', 'integrate', 'forces/touch', '-',
             'OW: wait for groups', 'OW: load rows', 'OW: sweeps', 'OW: forces+integr.', 'OW: instances']
    print('%-8s' % 'block' + ''.join('%20s' % n for n in names) + '%10s' % 'total')
    for b in (70, 130, 190, 250):
        env = dict(os.environ, RR_ABLATE=str((b << 16) | 0x4000))
        r = subprocess.run([sys.executable, __file__, 'child'], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        v = [float(x) for x in r.stdout.strip().split('\n')[-1].split()]
        print('%-8d' % b + ''.join('%20.0f' % x for x in v) + '   groups %.0f  object wave %.0f' % (sum(v[:8]), sum(v[8:])))
