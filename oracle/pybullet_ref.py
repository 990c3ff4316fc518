"""PyBullet-side harness: the reference's env.step() path driven through the pybullet API, for boxes that HAVE pybullet.

TEST INFRASTRUCTURE (lives under oracle/ with the other checkers; imported only by tests/, bench.py's cpu_baseline leg
and the command line below -- never by real_robots_amd).  Written from SURVEY.md / the reference's call sites, not copied:
it restates, call by call, what REALRobotEnv + Kuka ask of pybullet, so that it needs neither `gym` nor `pybullet_envs`:

  scene / reset   env.py:202-219, robot.py:120-129,165-185  resetSimulation; loadURDF(kuka_gripper.urdf) moved to
                  [-0.55, 0, -0.04]; table + objects with URDF_USE_INERTIA_FROM_FILE at Kuka.object_poses; every joint
                  resetJointState(0, 0) with the motor disabled; gravity -9.81; fixedTimeStep 0.005 (set BEFORE the
                  resetSimulation like SingleRobotEmptyScene does, SURVEY A.1.2: solver iterations / ERP fall back to
                  Bullet's defaults afterwards)
  step            env.py:314-356, 257-264; robot.py:188-211  rate limit, out-of-bounds rule, clip + gripper coupling,
                  11 x setJointMotorControl2(POSITION_CONTROL, targetPosition), stepSimulation, joint / contact read-back
  touch / contacts  robot.py:131-163   getContactPoints per link, |distance| < 0.1, max normal force per skin link
  camera          env.py:136-141,249-255,536-567   computeViewMatrix(eye [0.01,0,1.2] -> table position, up z),
                  computeProjectionMatrixFOV(80, W/H, 0.1, 100), getCameraImage(TinyRenderer)

It cannot run in the build container (no pybullet, no network); `available()` is False there and everything that needs
pybullet raises `PyBulletUnavailable`.  Where it can run it serves three purposes:
  python -m oracle.pybullet_ref record  OUT.npz   golden vectors (states, contacts, images) for seeded action streams and
                                                  the 36-pair macro script of tests/test_actions.py -> tests/golden/
  python -m oracle.pybullet_ref compare           the same streams through oracle/rr_oracle.c, differences printed
  bench.py                                        cpu_baseline.kind = "reference" (one process per host core)
The model data (URDF / OBJ / PNG) is the reference's; it is located at run time (RR_REFERENCE_DATA, an installed
`real_robots`, pybullet_data -- the reference copies its data there on import --, /root/reference) and never copied here.
"""
import os
import sys
import time

import numpy as np

JOINT_NAMES = ['lbr_iiwa_joint_%d' % (i + 1) for i in range(7)] + [
    'base_to_finger00_joint', 'finger00_to_finger01_joint', 'base_to_finger10_joint', 'finger10_to_finger11_joint']
OBJECTS = ['cube', 'tomato', 'mustard']                              # robot.py:49-50 (after "table")
OBJECT_POSES = {                                                     # robot.py:19-24
    'table': [0.0, 0.0, 0.08, 0.0, 0.0, 0.0], 'cube': [-0.1, 0.0, 0.45, 0.0, 0.0, 0.0],
    'tomato': [-0.1, -0.3, 0.45, 0.0, 0.0, 0.0], 'mustard': [-0.1, 0.3, 0.45, 1.5708, 3.14159, 0.0]}
ROBOT_POSITION = [-0.55, 0.0, -0.04]                                 # robot.py:46
MAX_DIFF = np.array([0.2, 0.2, 0.2, 0.2, 0.2, 0.3, 0.3, 0.1, 0.1])   # env.py:317
SKINS = ['skin_00', 'skin_01', 'skin_10', 'skin_11']                 # robot.py:156
URDF_DIR = os.path.join('kuka_gripper_description', 'urdf')


class PyBulletUnavailable(RuntimeError):
    pass


def available():
    try:
        import pybullet  # noqa: F401
        return True
    except Exception:
        return False


def find_reference_data():
    """Directory that contains kuka_gripper_description/urdf/kuka_gripper.urdf, or None."""
    cands = [os.environ.get('RR_REFERENCE_DATA')]
    try:
        import pybullet_data
        cands.append(pybullet_data.getDataPath())
    except Exception:
        pass
    for mod in ('real_robots',):
        try:
            m = __import__(mod)
            cands.append(os.path.join(os.path.dirname(os.path.abspath(m.__file__)), 'data'))
        except Exception:
            pass
    cands.append('/root/reference/real_robots/data')
    for c in cands:
        if c and os.path.exists(os.path.join(c, URDF_DIR, 'kuka_gripper.urdf')):
            return c
    return None


def joint_limits():
    """min_joints / max_joints of Kuka.__init__ (robot.py:58-67)."""
    lo, hi = -np.ones(9) * np.pi * 0.944, np.ones(9) * np.pi * 0.944
    for k in (0, 1, 3, 5):
        lo[k], hi[k] = -np.pi * 0.666, np.pi * 0.666
    lo[6], hi[6] = -np.pi * 0.972, np.pi * 0.972
    lo[7], hi[7] = 0.0, np.pi / 2
    lo[8], hi[8] = 0.0, np.pi / 2
    return lo, hi


class PyBulletRef:
    """One REALRobot env on PyBullet (DIRECT mode), restated from the reference's call sites."""

    def __init__(self, n_objects=3, width=320, height=240, data_dir=None):
        if not available():
            raise PyBulletUnavailable("pybullet is not importable here")
        import pybullet
        self.p = pybullet
        self.data = data_dir or find_reference_data()
        if self.data is None:
            raise PyBulletUnavailable("the reference's model data was not found (set RR_REFERENCE_DATA)")
        self.n_objects, self.W, self.H = int(n_objects), int(width), int(height)
        self.used = ['table'] + OBJECTS[:self.n_objects]
        self.lo, self.hi = joint_limits()
        self.cid = pybullet.connect(pybullet.DIRECT)
        self.reset()

    def close(self):
        if getattr(self, 'cid', None) is not None:
            self.p.disconnect(self.cid)
            self.cid = None

    # ---------------------------------------------------------------- reset (env.py:206-219; robot.py:120-129,165-185)
    def reset(self):
        p, c = self.p, self.cid
        # World.clean_everything of SingleRobotEmptyScene(gravity 9.81, timestep 0.005, frame_skip 1) runs first ...
        p.setGravity(0, 0, -9.81, physicsClientId=c)
        p.setDefaultContactERP(0.9, physicsClientId=c)
        p.setPhysicsEngineParameter(fixedTimeStep=0.005, numSolverIterations=5, numSubSteps=1, physicsClientId=c)
        # ... then Kuka.reset wipes the world (robot.py:121)
        p.resetSimulation(physicsClientId=c)
        self.robot = p.loadURDF(os.path.join(self.data, URDF_DIR, 'kuka_gripper.urdf'), physicsClientId=c)
        self.joint, self.link = {}, {}
        for j in range(p.getNumJoints(self.robot, physicsClientId=c)):
            info = p.getJointInfo(self.robot, j, physicsClientId=c)
            self.joint[info[1].decode()] = j
            self.link[info[12].decode()] = j
            # addToScene: every joint motor starts with force 0
            p.setJointMotorControl2(self.robot, j, p.POSITION_CONTROL, positionGain=0.1, velocityGain=0.1, force=0, physicsClientId=c)
        _, orn = p.getBasePositionAndOrientation(self.robot, physicsClientId=c)
        p.resetBasePositionAndOrientation(self.robot, ROBOT_POSITION, orn, physicsClientId=c)      # reset_position keeps the orientation
        self.body = {}
        for name in self.used:
            x = OBJECT_POSES[name]
            quat = p.getQuaternionFromEuler(x[3:])
            b = p.loadURDF(os.path.join(self.data, URDF_DIR, name + '.urdf'), basePosition=x[:3], baseOrientation=quat,
                           useFixedBase=False, flags=p.URDF_USE_INERTIA_FROM_FILE, physicsClientId=c)
            self.body[name] = b
            p.resetBasePositionAndOrientation(b, x[:3], quat, physicsClientId=c)
        for j in self.joint.values():                                                              # robot.py:181-182
            p.resetJointState(self.robot, j, targetValue=0, targetVelocity=0, physicsClientId=c)
            p.setJointMotorControl2(self.robot, j, p.POSITION_CONTROL, targetPosition=0, targetVelocity=0, positionGain=0.1,
                                    velocityGain=0.1, force=0, physicsClientId=c)
        p.setGravity(0, 0, -9.81, physicsClientId=c)                                               # env.py:208
        self.timestep = 0

    # ---------------------------------------------------------------- step (env.py:326-356)
    def calc_state(self):
        q = [self.p.getJointState(self.robot, self.joint[n], physicsClientId=self.cid)[0] for n in JOINT_NAMES[:9]]
        q[8] = -q[8]
        return np.array(q)

    def step(self, action9=None):
        p, c = self.p, self.cid
        a = np.zeros(9) if action9 is None else np.asarray(action9, dtype=np.float64)
        assert np.isfinite(a).all() and len(a) == 9
        cur = self.calc_state()
        a = cur + np.maximum(-MAX_DIFF, np.minimum(MAX_DIFF, a - cur))                             # env.py:314-321
        for name in self.used:                                                                     # env.py:257-264
            x, y, z = p.getBasePositionAndOrientation(self.body[name], physicsClientId=c)[0]
            if z < OBJECT_POSES['table'][2] or (x > 0.11 and z < 0.29):
                pose = OBJECT_POSES[name]
                p.resetBasePositionAndOrientation(self.body[name], pose[:3], p.getQuaternionFromEuler(pose[3:]), physicsClientId=c)
        a = np.maximum(self.lo, np.minimum(a, self.hi))                                            # robot.py:192-193
        a[8] = max(0.0, min(2 * a[7], a[8]))
        tgt = list(a[:7]) + [a[7], -a[8], a[7], -a[8]]                                             # robot.py:195-201
        for n, x in zip(JOINT_NAMES, tgt):
            p.setJointMotorControl2(self.robot, self.joint[n], p.POSITION_CONTROL, targetPosition=float(x), physicsClientId=c)
        p.stepSimulation(physicsClientId=c)                                                        # env.py:340
        self.timestep += 1

    # ---------------------------------------------------------------- read-back
    def state61(self):
        """q[11] qd[11] then per object pos3 quat4 (xyzw) lin3 ang3 -- the layout of RR_F_STATE / rro_get_state."""
        p, c = self.p, self.cid
        js = [p.getJointState(self.robot, self.joint[n], physicsClientId=c) for n in JOINT_NAMES]
        out = [s[0] for s in js] + [s[1] for s in js]
        for k in range(3):
            if k < self.n_objects:
                b = self.body[OBJECTS[k]]
                pos, orn = p.getBasePositionAndOrientation(b, physicsClientId=c)
                lin, ang = p.getBaseVelocity(b, physicsClientId=c)
                out += list(pos) + list(orn) + list(lin) + list(ang)
            else:
                x = OBJECT_POSES[OBJECTS[k]]
                out += list(x[:3]) + list(p.getQuaternionFromEuler(x[3:])) + [0.0] * 6
        return np.array(out)

    def link_position(self, name):
        """robot.parts[name].get_position(): COM frame of the link."""
        return np.array(self.p.getLinkState(self.robot, self.link[name], physicsClientId=self.cid)[0])

    def contacts(self):
        """Rows {link index of the robot part, other body id, x, y, z, nx, ny, nz, distance, normal force} of the last step."""
        rows = []
        for j in [-1] + sorted(self.joint.values()):
            for ct in self.p.getContactPoints(bodyA=self.robot, linkIndexA=j, physicsClientId=self.cid):
                rows.append([j, ct[2]] + list(ct[5]) + list(ct[7]) + [ct[8], ct[9]])
        return np.array(rows).reshape(-1, 10)

    def touch_sensors(self):                                                                       # robot.py:152-163
        out = np.zeros(4)
        for i, s in enumerate(SKINS):
            for ct in self.p.getContactPoints(bodyA=self.robot, linkIndexA=self.link[s], physicsClientId=self.cid):
                if abs(ct[8]) < 0.1:
                    out[i] = max(out[i], ct[9])
        return out

    def render(self):                                                                              # env.py:249-255,536-567
        p, c = self.p, self.cid
        tgt = p.getBasePositionAndOrientation(self.body['table'], physicsClientId=c)[0]
        view = p.computeViewMatrix([0.01, 0.0, 1.2], list(tgt), [0, 0, 1])
        proj = p.computeProjectionMatrixFOV(80, float(self.W) / self.H, 0.1, 100.0)
        _, _, px, depth, mask = p.getCameraImage(self.W, self.H, view, proj, renderer=p.ER_TINY_RENDERER, physicsClientId=c)
        rgb = np.array(px, dtype=np.uint8).reshape(self.H, self.W, 4)[:, :, :3]
        return rgb, np.array(depth, dtype=np.float32).reshape(self.H, self.W), np.array(mask, dtype=np.int32).reshape(self.H, self.W)

    def inverse_kinematics(self, pos, orn):                                                        # env.py:372-375,423-426
        return np.array(self.p.calculateInverseKinematics(self.robot, 7, list(pos), list(orn), maxNumIterations=1000,
                                                          residualThreshold=0.001, physicsClientId=self.cid))

    def engine_parameters(self):
        return dict(self.p.getPhysicsEngineParameters(physicsClientId=self.cid))


# -------------------------------------------------------------------------------------------------- scripted streams
def macro_plan(ref, point_1, point_2):
    """generate_plan (env.py:414-457) with this env's IK; 1000 x 9."""
    home, home2 = np.zeros(9), np.zeros(9)
    home2[5] = home2[6] = np.pi / 2
    from real_robots_amd.mathutil import quat_from_euler          # = pybullet.getQuaternionFromEuler
    orn = quat_from_euler(0, 3.14, -1.57)

    def go(xyz):
        return ref.inverse_kinematics(xyz, orn)[:9]

    def interp(p1, p2, steps):
        p1, p2 = np.array(p1), np.array(p2)
        pieces = min(int(np.linalg.norm(p2 - p1) / 0.05) + 1, steps)
        coords = np.linspace(p1, p2, pieces + 1)
        joints = np.zeros((steps, 9))
        chunk = int(steps / pieces)
        for i, cxyz in enumerate(coords[1:]):
            joints[i * chunk:, :] = go(cxyz)
        return joints

    p1h, p1l = go(np.hstack([point_1, 0.6])), go(np.hstack([point_1, 0.46]))
    p2h = go(np.hstack([point_2, 0.6]))
    return np.vstack([np.tile(home2, (100, 1)), np.tile(p1h, (100, 1)), np.tile(p1l, (50, 1)),
                      interp(np.hstack([point_1, 0.46]), np.hstack([point_2, 0.46]), 500), np.tile(p2h, (50, 1)),
                      np.tile(home2, (100, 1)), np.tile(home, (100, 1))])


def perimeter_pairs():
    pts = [(y, x) for x in (-0.5, 0.0, 0.5) for y in (-0.25, 0.05)]      # tests/test_actions.py:101-117
    return [(np.array(a), np.array(b)) for a in pts for b in pts]


def seeded_actions(seed, steps, scale=1.0):
    """The resample-and-hold joint commands of real_robots_amd.distributed.synthetic_actions for env id `seed`."""
    from real_robots_amd.distributed import synthetic_actions
    return np.stack([synthetic_actions([seed], t)[0].astype(np.float64) * scale for t in range(steps)])


GOLDEN_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'pybullet_golden.npz')
GOLDEN_STREAMS = (('free_0.4', 7, 300, 0.4), ('full_1.0', 11, 400, 1.0))     # name, env id of the seeded commands, steps, command scale
MACRO_CHECK_STEPS = (199, 249, 749, 799, 849, 899, 999)


class OracleBackend:
    """The recorder's backend interface on top of oracle/rr_oracle.c + oracle/kinematics.py: lets the recorder / consumer pair
    be exercised end to end without pybullet (tests/test_pybullet_golden.py) -- what it records is NOT a PyBullet vector."""

    def __init__(self, n_objects=3, width=128, height=128):
        from oracle.oracle import Oracle
        self.o = Oracle(n_objects, width, height)
        self.n_objects = n_objects

    def reset(self):
        self.o.reset()
        self._prefer = None

    def step(self, action9=None):
        self.o.step(action9)

    def state61(self):
        return self.o.state

    def touch_sensors(self):
        return self.o.obs()[1]

    def contacts(self):
        return self.o.contacts()

    def render(self):
        return self.o.render()

    def link_position(self, name):
        return self.o.link_pose(name)[:3]

    def inverse_kinematics(self, pos, orn):
        from oracle.kinematics import inverse_kinematics
        if not hasattr(self, '_prefer'):
            self._prefer = None
        q = inverse_kinematics(np.zeros(11), pos, orn, prefer=self._prefer)
        self._prefer = q
        return q[:9]

    def engine_parameters(self):
        return {'backend': 'oracle/rr_oracle.c (NOT pybullet)'}

    def close(self):
        pass


def record(out_path=None, width=128, height=128, backend=None):
    """Golden vectors from a live PyBullet (default out_path: tests/golden/pybullet_golden.npz, where the tests pick them up
    automatically): per stream the action sequence, the state after every step, touch sensors, the contacts of the last step and
    a rendered frame; plus the gripper-base positions at the check steps of the 36-pair macro script."""
    out_path = out_path or GOLDEN_PATH
    ref = backend if backend is not None else PyBulletRef(3, width, height)
    gold = {'engine_parameters': repr(ref.engine_parameters())}
    for name, acts in ((n_, seeded_actions(sd, st, sc)) for n_, sd, st, sc in GOLDEN_STREAMS):
        ref.reset()
        states, touch = [], []
        for a in acts:
            ref.step(a)
            states.append(ref.state61())
            touch.append(ref.touch_sensors())
        rgb, depth, mask = ref.render()
        gold.update({name + '/actions': acts, name + '/states': np.array(states), name + '/touch': np.array(touch),
                     name + '/rgb': rgb, name + '/depth': depth, name + '/mask': mask, name + '/contacts': ref.contacts()})
    way = []
    for p1, p2 in perimeter_pairs():
        ref.reset()
        for _ in range(100):
            ref.step(np.zeros(9))
        plan = macro_plan(ref, p1, p2)
        row = []
        for t, a in enumerate(plan):
            ref.step(a)
            if t in MACRO_CHECK_STEPS:
                row.append(ref.link_position('base'))
        way.append(np.concatenate([p1, p2] + row))
    gold['macro_waypoints'] = np.array(way)
    np.savez_compressed(out_path, **gold)
    ref.close()
    return out_path


def divergence(gold, make_stepper, streams=GOLDEN_STREAMS):
    """Replays the recorded action streams through another simulator and returns, per stream, the largest deviation of joints
    / object positions from the recorded states in windows of 50 steps.  `make_stepper()` -> object with reset(), step(a),
    state61().  Used by tests/test_pybullet_golden.py for the oracle and for the HIP path."""
    out = {}
    for name, _, steps, _ in streams:
        acts, states = gold[name + '/actions'], gold[name + '/states']
        sim = make_stepper()
        sim.reset()
        rows = []
        for t, a in enumerate(acts):
            sim.step(a)
            d = np.abs(np.asarray(sim.state61(), dtype=np.float64) - states[t])
            rows.append((d[:11].max(), max(d[22 + 13 * k: 25 + 13 * k].max() for k in range(3))))
        rows = np.array(rows)
        out[name] = {'joints_rad': [float(rows[i:i + 50, 0].max()) for i in range(0, len(rows), 50)],
                     'object_pos_m': [float(rows[i:i + 50, 1].max()) for i in range(0, len(rows), 50)]}
    return out


def compare():
    """Seeded streams through PyBullet and through oracle/rr_oracle.c side by side; prints the divergence over time."""
    from oracle.oracle import Oracle
    ref, orc = PyBulletRef(3, 128, 128), Oracle(3, 128, 128)
    print("engine parameters:", ref.engine_parameters())
    for scale, steps in ((0.4, 300), (1.0, 400)):
        ref.reset()
        orc.reset()
        acts = seeded_actions(7, steps, scale)
        for t, a in enumerate(acts):
            ref.step(a)
            orc.step(a)
            if t % 50 == 49:
                d = np.abs(ref.state61() - orc.state)
                print("scale %.1f step %3d  |dq| %.2e  |dqd| %.2e  |dpos| %.2e" % (scale, t + 1, d[:11].max(), d[11:22].max(),
                      max(d[22 + 13 * k: 25 + 13 * k].max() for k in range(3))))
        r, dep, m = ref.render()
        orc.state = ref.state61()
        r2, dep2, m2 = orc.render()
        print("image: mask differs in %d px, rgb > 8 levels in %d px, depth max diff %.2e" %
              (int((m != m2).sum()), int((np.abs(r.astype(int) - r2.astype(int)).max(-1) > 8).sum()), float(np.abs(dep - dep2).max())))
    ref.close()


# -------------------------------------------------------------------------------------------------- CPU baseline
def _baseline_worker(args):
    seconds, seed, n_objects, width, height = args
    ref = PyBulletRef(n_objects, width, height)
    acts = seeded_actions(seed, 400)[::20]
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            ref.step(acts[(n // 20) % len(acts)])
            ref.touch_sensors()
            ref.render()
            n += 1
    dt = time.perf_counter() - t0
    ref.close()
    return n, dt


def cpu_baseline(seconds=10.0, cores=1, n_objects=3, width=128, height=128):
    """bench.py's cpu_baseline with kind "reference": one PyBullet env per process on `cores` host cores, the same
    per-env workload as the GPU line (step + touch read-back + render every step)."""
    if not available():
        raise PyBulletUnavailable("pybullet is not importable here")
    import multiprocessing as mp
    import pybullet
    ctx = mp.get_context('spawn')
    with ctx.Pool(cores) as pool:
        res = pool.map(_baseline_worker, [(seconds, i, n_objects, width, height) for i in range(cores)])
    rate = sum(n / t for n, t in res)
    return {"value": round(rate, 1), "unit": "env-steps/s", "cores": cores, "kind": "reference",
            "sample": "PyBullet %s DIRECT, %d processes x 1 env x %.0f s each (%d env-steps in total), %d objects, full-range "
                      "commands, %dx%d TinyRenderer frame every step (oracle/pybullet_ref.py)"
                      % (getattr(pybullet, '__version__', '?'), cores, seconds, sum(n for n, _ in res), n_objects, width, height)}


if __name__ == '__main__':
    if not available():
        sys.exit("pybullet is not importable on this machine; nothing to do")
    if len(sys.argv) > 1 and sys.argv[1] == 'record':
        print(record(sys.argv[2] if len(sys.argv) > 2 else None))
    elif len(sys.argv) > 1 and sys.argv[1] == 'compare':
        compare()
    else:
        sys.exit("usage: python -m oracle.pybullet_ref record [OUT.npz, default tests/golden/pybullet_golden.npz] | compare")
