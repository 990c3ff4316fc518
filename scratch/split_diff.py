"""First divergence between the split step (three streams, look-ahead) and the in-line one (RR_NO_SPLIT RR_NO_LOOKAHEAD) on the bench
workload (development).  Two processes' worth in one: envs are created one after the other with different environments."""
import importlib.util
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
N, T = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 60
cmds = bench.make_commands(torch, np, np.arange(N), T, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
os.environ['RR_NO_SPLIT'] = '1'
os.environ['RR_NO_LOOKAHEAD'] = '1'
plain = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
np.set_printoptions(precision=9, linewidth=220)
for t in range(T):
    cls = env.host(nat.F_ENV_CLASS).copy()
    env.step(device_ptr=cmds[t].data_ptr(), render=True)
    plain.step(device_ptr=cmds[t].data_ptr(), render=True)
    a, b = env.state, plain.state
    if not np.array_equal(a, b, equal_nan=True):
        bad = np.argwhere(a != b)
        envs = np.unique(bad[:, 0])
        print("step", t, ":", len(envs), "envs differ; classes of the first:", cls[envs[:10]].tolist())
        e = int(envs[0])
        print("env", e, "fields", bad[bad[:, 0] == e][:, 1].tolist())
        print("split ", a[e][bad[bad[:, 0] == e][:, 1]])
        print("inline", b[e][bad[bad[:, 0] == e][:, 1]])
        print("contacts", len(env.contacts(e)), "\n", env.contacts(e)[:, [0, 1, 2, 10]])
        break
else:
    print("no divergence in", T, "steps")
