"""Position error of the plan's IK solutions themselves (FK of the commanded joints vs the Cartesian target), host planner,
and how long the rate-limited first-order motor needs between the plan's way points."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd.kinematics import generate_plan, link_pose, EE_LINK
perimeter = [(a, b) for a in (-0.25, 0.05) for b in (-0.5, 0.0, 0.5)]
maxdiff = np.array([.2, .2, .2, .2, .2, .3, .3, .1, .1])
def simulate(q0, plan, t0, t1):
    q = q0.copy()
    for t in range(t0, t1):
        a = q + np.clip(plan[t] - q, -maxdiff, maxdiff)
        q = q + 0.1 * (a - q)
    return q
for p1 in perimeter:
    plan = generate_plan(np.zeros(11), (p1, (0.05, 0.0)))
    def ee(q9):
        q = np.zeros(11); q[:7] = q9[:7]
        return link_pose(q, EE_LINK)[0]
    e_h = np.linalg.norm(ee(plan[150]) - np.array([p1[0], p1[1], 0.6]))
    e_l = np.linalg.norm(ee(plan[220]) - np.array([p1[0], p1[1], 0.46]))
    q = simulate(np.zeros(9), plan, 0, 200)
    lag199 = np.linalg.norm(ee(q) - ee(plan[150]))
    print("p1", p1, "IK error high %.4f low %.4f | kinematic motor model at t=199: %.4f m from the commanded pose, joint gap max %.3f rad (home2 -> p1_h needs max %.2f rad)" %
          (e_h, e_l, lag199, np.abs(q - plan[150]).max(), np.abs(plan[150] - plan[50]).max()))
