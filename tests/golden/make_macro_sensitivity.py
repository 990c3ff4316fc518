#!/usr/bin/env python3
"""Generates tests/golden/macro_sensitivity.json: the reference's macro tracking script (tests/test_actions.py:42-71,101-117,
147-152: all 36 ordered pairs of the six perimeter points; gripper `base` expected within 0.01 m of (p1, 0.6) @199,
(p1, 0.46) @249, (p2, 0.46) @749, raw_xy[849] @849 and home @999) run on the CPU oracle for the motor models a first PyBullet
run has to decide between:

    kp in {0.1, 0.2, 0.5, 1.0}        position gain of the 11 position motors (pybullet default 0.1)
    rate limit on / off               limitActionByJoint (env.py:314-321) applied to the command before the motor, or not

For every variant and pair: the distance of the gripper base from each of the five check points.  With the documented
semantics (kp 0.1 + rate limit) the check point at t = 849 is out of reach; the table shows which variant
would satisfy the script's 1 cm everywhere, so that ONE comparison with `python -m oracle.pybullet_ref record` output
(key `macro_waypoints`) settles the motor model.  Plans: oracle/kinematics.py (independent of the motor model).

Run here (CPU only, ~2 min on 8 cores):  python tests/golden/make_macro_sensitivity.py
"""
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHECK_T = (199, 249, 749, 849, 999)
KPS = (0.1, 0.2, 0.5, 1.0)
HOME = np.array([-0.55, 0.0, 1.27])
HOME2 = np.array([-0.419, 0.0, 1.14])          # FK of home2 (tests/test_actions.py:65-66 quotes (-0.41, 0, 1.14))


def perimeter_pairs():
    pts = [(a, b) for a in (-0.25, 0.05) for b in (-0.5, 0.0, 0.5)]
    return [(p1, p2) for p1 in pts for p2 in pts]


def targets(p1, p2):
    """raw_xy[t] of the reference script at the five check points (index 849 lies 50 steps into the home2 segment)."""
    return {199: np.array([p1[0], p1[1], 0.6]), 249: np.array([p1[0], p1[1], 0.46]), 749: np.array([p2[0], p2[1], 0.46]),
            849: HOME2, 999: HOME}


def _plan(pair):
    from oracle.kinematics import generate_plan
    return generate_plan(np.zeros(11), pair)


def _run(args):
    from oracle.oracle import Oracle
    kp, no_limit, pair, plan = args
    o = Oracle(3, 64, 64, motor_kp=kp, no_rate_limit=int(no_limit))
    o.reset()
    shelf = [[0.2, 0.0, 0.75], [0.2, -0.3, 0.75], [0.2, 0.3, 0.75]]          # objects parked on the shelf (test_actions.py:95-98)
    for i, p in enumerate(shelf):
        o.set_object_pose(i, np.array(p + [0, 0, 0, 1.0]))
    tg = targets(*pair)
    out = []
    for t in range(1000):
        o.step(plan[t])
        if t in CHECK_T:
            out.append(float(np.linalg.norm(o.link_pose('base')[:3] - tg[t])))
    return out


def main():
    pairs = perimeter_pairs()
    with mp.get_context('spawn').Pool(min(8, os.cpu_count() or 1)) as pool:
        plans = pool.map(_plan, pairs)
        table = {}
        for kp in KPS:
            for no_limit in (False, True):
                res = pool.map(_run, [(kp, no_limit, pairs[i], plans[i]) for i in range(len(pairs))])
                table["kp=%g,rate_limit=%s" % (kp, "off" if no_limit else "on")] = res
    out = {"check_steps": list(CHECK_T), "pairs": pairs, "tolerance_of_the_reference_script_m": 0.01,
           "distance_m": {k: [[round(d, 5) for d in row] for row in v] for k, v in table.items()},
           "pairs_within_tolerance": {k: [int(sum(1 for row in v if row[j] < 0.01)) for j in range(len(CHECK_T))] for k, v in table.items()},
           "worst_m": {k: [round(max(row[j] for row in v), 4) for j in range(len(CHECK_T))] for k, v in table.items()}}
    path = os.path.join(ROOT, 'tests', 'golden', 'macro_sensitivity.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=0)
    for k in table:
        print(k, "pairs within 1 cm at t =", dict(zip(CHECK_T, out["pairs_within_tolerance"][k])), " worst", out["worst_m"][k])


if __name__ == '__main__':
    main()
