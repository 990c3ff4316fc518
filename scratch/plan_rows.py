import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.kinematics import generate_plan
perimeter = [(a, b) for a in (-0.25, 0.05) for b in (-0.5, 0.0, 0.5)]
pairs = [(p, (0.05, 0.0)) for p in perimeter]
env = BatchedREALRobotEnv(len(pairs), objects=1, width=64, height=64)
env.plan_macro(np.array(pairs, dtype=np.float32))
np.set_printoptions(precision=3, suppress=True, linewidth=200)
for i, (p1, p2) in enumerate(pairs):
    pl = env.get_plan(i); host = generate_plan(np.zeros(11), (p1, p2))
    print("p1", p1, "device p1_h", pl[150][:7], " p1_l", pl[220][:7])
    print("            host   p1_h", host[150][:7], " p1_l", host[220][:7])
