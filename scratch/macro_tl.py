"""The bench's macro workload (4096 envs, random macro actions, render every step) for a kernel trace: steps 0..520 of the plans.
rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 scratch/macro_tl.py ; python scratch/timeline.py <dir> 30"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from real_robots_amd.batched import BatchedREALRobotEnv
N = 4096
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
rng = np.random.default_rng(5)
env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
for t in range(520):
    env.step_plan(render=True)
env.sync()
env.close()
