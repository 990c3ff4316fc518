"""Time breakdown of k_raster by ablation (development build: `make -C real_robots_amd/csrc stats`): RR_ABLATE bits 1 = nothing
after projection + set-up, 2 = no wave-cooperative path, 8 = no triangles at all (fill, cull, restore marks, compaction only),
16 = no near-plane clipping.  HIP-event time of k_raster alone on the stream (rr_set_timing), bench workload, 4096 envs."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['RR_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so')
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N = 4096
cmds = bench.make_commands(torch, np, np.arange(N), 200, 1.0, 'cuda:0')
for abl in [int(a) for a in (sys.argv[1:] or ['0', '1', '2', '8', '16'])]:
    os.environ['RR_ABLATE'] = str(abl)
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
    for t in range(170):
        env.step(device_ptr=cmds[t].data_ptr(), render=(t >= 165))
    env.sync()
    env.set_timing(True)
    for t in range(170, 190):
        env.step(device_ptr=cmds[t].data_ptr(), render=True)
    env.sync()
    tm = env.get_timing()
    env.set_timing(False)
    print('RR_ABLATE', abl, {k: round(ms / max(n, 1), 4) for k, (ms, n) in tm.items() if n and k in ('k_raster', 'k_shade')}, flush=True)
    env.close()
