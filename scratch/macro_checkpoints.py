"""Distance of the gripper base to the reference test's checkpoints for all 36 perimeter pairs (tests/test_actions.py)."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
perimeter = [(a, b) for a in (-0.25, 0.05) for b in (-0.5, 0.0, 0.5)]
pairs = [(p1, p2) for p1 in perimeter for p2 in perimeter]
N = len(pairs)
env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
for i in range(N):
    env.set_object_pose(i, 0, [0.2, 0.0, 0.75, 0, 0, 0, 1]); env.set_object_pose(i, 1, [0.2, -0.3, 0.75, 0, 0, 0, 1]); env.set_object_pose(i, 2, [0.2, 0.3, 0.75, 0, 0, 0, 1])
env.plan_macro(np.array(pairs, dtype=np.float32))
base = nat.LINK_NAMES.index('base')
home, home2 = np.array([-0.55, 0.0, 1.27]), np.array([-0.41, 0.0, 1.14])
res = {}
for t in range(1000):
    env.step_plan()
    if t in (150, 199, 249, 749, 849, 999):
        lp = env.link_poses()[:, base, :3]
        for i, (p1, p2) in enumerate(pairs):
            tgt = {150: [p1[0], p1[1], 0.6], 199: [p1[0], p1[1], 0.6], 249: [p1[0], p1[1], 0.46], 749: [p2[0], p2[1], 0.46], 849: home2, 999: home}[t]
            res[(t, i)] = (np.linalg.norm(lp[i] - tgt), lp[i] - np.array(tgt))
for t in (150, 199, 249, 749, 849, 999):
    d = np.array([res[(t, i)][0] for i in range(N)])
    print("t %3d: max %.4f mean %.4f  worst pair %s offset %s" % (t, d.max(), d.mean(), pairs[int(d.argmax())], np.round(res[(t, int(d.argmax()))][1], 4)))
print("per p1 at t=199:", {p: round(max(res[(199, i)][0] for i, pr in enumerate(pairs) if pr[0] == p), 4) for p in perimeter})
print("contacts of the worst env at the end:", len(env.contacts(6)))
