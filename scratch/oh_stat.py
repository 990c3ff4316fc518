"""Sizing of an object-heavy class: among the heavy envs whose generic rows hold no robot body -- static contacts per object,
object-object contacts per env (headline workload at step argv[1], default 2100)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from collections import Counter
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N, nobj = 4096, 3
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2100
cmds = bench.make_commands(torch, np, np.arange(N), T + 1, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(N, objects=nobj, width=32, height=32, want_mask=False)
for t in range(T): env.step(device_ptr=cmds[t].data_ptr(), render=False)
env.sync()
cls = env.host(nat.F_ENV_CLASS)
ids = np.flatnonzero(cls >= 1)
n_robot = 0; per_obj = Counter(); oo = Counter(); tot = Counter(); robot_any = 0
for i in ids:
    ct = env.contacts(int(i))
    a, b = ct[:, 0].astype(int), ct[:, 1].astype(int)
    rob = ((a >= 0) & (a < 16)) | ((b >= 0) & (b < 16))
    if rob.any(): n_robot += 1; continue
    st = (a >= 16) & (b < 0)
    mx = max(int((st & (a == 16 + o)).sum()) for o in range(nobj))
    n_oo = int(((a >= 16) & (b >= 16)).sum())
    per_obj[mx] += 1; oo[n_oo] += 1; tot[len(ct)] += 1
print('step %d: heavy + very heavy %d, robot in the list %d, objects only %d' % (T, len(ids), n_robot, len(ids) - n_robot))
print('  max static contacts of one object:', sorted(per_obj.items()))
print('  object-object contacts:', sorted(oo.items()))
print('  contacts in all:', sorted(tot.items()))
env.close()
