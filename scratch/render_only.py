"""k_raster alone on the machine: the bench workload stepped to t = 170, then `rr_render` of all envs R times (k_render_setup,
k_raster, k_shade and nothing beside them).  For PMC passes: rocprofv3 --pmc ... -- python3 scratch/render_only.py [R]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N, R = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 20
cmds = bench.make_commands(torch, np, np.arange(N), 200, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
for t in range(172):
    env.step(device_ptr=cmds[t].data_ptr(), render=(t >= 168))
env.sync()
env.set_timing(True)
for r in range(R):
    env.render()
env.sync()
tm = env.get_timing()
print({k: round(ms / max(n, 1), 4) for k, (ms, n) in tm.items() if n}, flush=True)
env.close()
