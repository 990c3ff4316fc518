#!/usr/bin/env python3
"""Generates tests/golden/fk_golden.json: forward kinematics of the Kuka+gripper computed *directly from the
reference's URDF text* with a few lines of numpy (no use of the model blob, the oracle or the HIP library), plus the
known answers the reference's own tests hold. Run in the build container (reads /root/reference data files):
    python tests/golden/make_golden.py
The fixture is data: joint vectors in, link positions out.
Known answers cited: /root/reference/tests/test_actions.py:60,65-66 (gripper `base` at home / home2, tolerance 0.01 m,
test_actions.py:150)."""
import json
import os
import xml.etree.ElementTree as ET

import numpy as np

URDF = "/root/reference/real_robots/data/kuka_gripper_description/urdf/kuka_gripper.urdf"
ROBOT_POS = np.array([-0.55, 0.0, -0.04])     # real_robots/envs/robot.py:46


def rpy(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
                     [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def rot(axis, a):
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def main():
    root = ET.parse(URDF).getroot()
    joints = {}
    for j in root.findall('joint'):
        o = j.find('origin')
        ax = j.find('axis')
        joints[j.find('child').get('link')] = dict(
            name=j.get('name'), type=j.get('type'), parent=j.find('parent').get('link'),
            xyz=np.array([float(x) for x in o.get('xyz').split()]),
            R=rpy(*[float(x) for x in o.get('rpy').split()]),
            axis=np.array([float(x) for x in ax.get('xyz').split()]) if ax is not None else None)
    # link COM offsets: BodyPart.get_position() reports the inertial (COM) frame (pybullet getLinkState[0])
    coms = {}
    for l in root.findall('link'):
        o = l.find('inertial').find('origin')
        coms[l.get('name')] = np.array([float(x) for x in o.get('xyz').split()])
    # command (9) -> joint angle by joint name (robot.py:195-201)
    def joint_angles(cmd):
        q = {'lbr_iiwa_joint_%d' % (i + 1): cmd[i] for i in range(7)}
        q['base_to_finger00_joint'] = cmd[7]
        q['base_to_finger10_joint'] = cmd[7]
        q['finger00_to_finger01_joint'] = -cmd[8]
        q['finger10_to_finger11_joint'] = -cmd[8]
        return q

    def link_frame(link, q):
        if link not in joints:
            return np.eye(3), ROBOT_POS.copy()
        j = joints[link]
        Rp, pp = link_frame(j['parent'], q)
        R = Rp @ j['R']
        p = pp + Rp @ j['xyz']
        if j['type'] == 'revolute':
            R = R @ rot(j['axis'], q[j['name']])
        return R, p

    rng = np.random.default_rng(2020)
    cmds = [np.zeros(9)]
    h2 = np.zeros(9)
    h2[5] = h2[6] = np.pi / 2
    cmds.append(h2)
    for _ in range(6):
        c = rng.uniform(-2.0, 2.0, 9)
        c[7:] = rng.uniform(0, 1.5, 2)
        cmds.append(c)
    links = ['lbr_iiwa_link_3', 'lbr_iiwa_link_7', 'base', 'finger_00', 'finger_01', 'finger_11', 'skin_00', 'skin_11']
    cases = []
    for c in cmds:
        q = joint_angles(c)
        def com_pos(l):
            R, p = link_frame(l, q)
            return (p + R @ coms[l]).tolist()
        cases.append(dict(cmd=c.tolist(), links={l: com_pos(l) for l in links}))
    out = dict(
        source="kuka_gripper.urdf via tests/golden/make_golden.py",
        reference_known_answers=dict(
            home_base=[-0.55, 0.0, 1.27], home2_base=[-0.41, 0.0, 1.14], tolerance=0.01,
            cite="tests/test_actions.py:60,65-66,150 (reference repo)"),
        cases=cases)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fk_golden.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', path)


if __name__ == '__main__':
    main()
