"""Kuka facade: the attributes of the reference's `Kuka` robot object that callers reach through `env.robot`
(real_robots/envs/robot.py:10-211), backed by the batched HIP environment instead of a pybullet client."""
import numpy as np

from .. import _native as nat
from .. import spaces
from ..mathutil import quat_from_euler


class BodyHandle:
    """`object_bodies[name]` / `parts[name]`: get_position / get_pose / reset_pose (pybullet_envs BodyPart API)."""

    def __init__(self, robot, kind, index):
        self._robot, self._kind, self._index = robot, kind, index

    def get_pose(self):
        return self._robot._pose(self._kind, self._index)

    def get_position(self):
        return self.get_pose()[:3]

    def current_position(self):
        return self.get_position()

    def get_orientation(self):
        return self.get_pose()[3:]

    def reset_pose(self, position, orientation):
        if self._kind != 'object':
            raise NotImplementedError("only the free objects can be re-posed")
        self._robot._env._backend().set_object_pose(0, self._index, list(position) + list(orientation))

    def reset_position(self, position):
        self.reset_pose(position, self.get_pose()[3:])


class Kuka:
    used_objects = ["table", "orange", "mustard", "cube", "tomato"]
    object_poses = {                                   # robot.py:19-24
        "table": [0.0, 0.0, 0.08, 0.0, 0.0, 0.0],
        "orange": [0.2, -0.15, 0.45, 0.0, 0.0, 0.0],
        "mustard": [-0.1, 0.3, 0.45, 1.5708, 3.14159, 0.0],
        "cube": [-0.1, 0.0, 0.45, 0.0, 0.0, 0.0],
        "tomato": [-0.1, -0.3, 0.45, 0.0, 0.0, 0.0]}
    num_joints = 9
    num_kuka_joints = 7
    num_gripper_joints = 2
    num_touch_sensors = 4
    eye_width = 320
    eye_height = 240

    class ObsSpaces:
        JOINT_POSITIONS = "joint_positions"
        TOUCH_SENSORS = "touch_sensors"
        RETINA = "retina"
        DEPTH = "depth"
        MASK = "mask"
        OBJ_POS = "object_positions"
        GOAL = "goal"
        GOAL_MASK = "goal_mask"
        GOAL_POS = "goal_positions"

    def __init__(self, additional_obs=False, objects=3, eye_width=None, eye_height=None, env=None):
        self._env = env
        self.robot_position = [-0.55, 0, -0.04]
        self.contact_threshold = 0.1
        self.used_objects = ["table", "cube", "tomato", "mustard"][:objects + 1]
        self.object_poses = {k: list(v) for k, v in Kuka.object_poses.items()}
        self.action_dim = self.num_joints
        if eye_width:
            self.eye_width = eye_width
        if eye_height:
            self.eye_height = eye_height
        self.min_joints = np.ones(9) * -np.pi * 0.944      # robot.py:58-67
        self.max_joints = np.ones(9) * np.pi * 0.944
        self.min_joints[0] = -np.pi * 0.666
        self.max_joints[0] = np.pi * 0.666
        self.min_joints[1:9:2] = -np.pi * 0.666
        self.max_joints[1:9:2] = np.pi * 0.666
        self.min_joints[6] = -np.pi * 0.972
        self.max_joints[6] = np.pi * 0.972
        self.min_joints[-2:] = 0
        self.max_joints[-2:] = np.pi / 2
        self.action_space = spaces.Box(low=self.min_joints, high=self.max_joints, dtype=float)
        H, W = self.eye_height, self.eye_width
        O = self.ObsSpaces
        obs = {
            O.JOINT_POSITIONS: spaces.Box(-np.inf, np.inf, [self.num_joints], dtype=float),
            O.TOUCH_SENSORS: spaces.Box(0, np.inf, [self.num_touch_sensors], dtype=float),
            O.RETINA: spaces.Box(0, 255, [H, W, 3], dtype=np.uint8),
            O.DEPTH: spaces.Box(0, 1, [H, W], dtype=float),
            O.GOAL: spaces.Box(0, 255, [H, W, 3], dtype=np.uint8)}
        if additional_obs:
            fmax = np.finfo(np.float32).max
            obj_obs = {o: spaces.Box(-np.array([fmax] * 3), np.array([fmax] * 3), dtype=float) for o in self.used_objects[1:]}
            obs[O.MASK] = spaces.Box(0, 255, [H, W], dtype=np.int32)
            obs[O.GOAL_MASK] = spaces.Box(0, 255, [H, W], dtype=np.int32)
            obs[O.OBJ_POS] = spaces.Dict(obj_obs)
            obs[O.GOAL_POS] = spaces.Dict(obj_obs)
        self.observation_space = spaces.Dict(obs)
        self.target = "orange"
        self.object_names = {0: "kuka", 1: "table"}
        for i, n in enumerate(self.used_objects[1:]):
            self.object_names[2 + i] = n
        self.object_bodies = {"kuka": BodyHandle(self, 'link', 0), "table": BodyHandle(self, 'table', 0)}
        for i, n in enumerate(self.used_objects[1:]):
            self.object_bodies[n] = BodyHandle(self, 'object', i)
        self.parts = {n: BodyHandle(self, 'link', i) for i, n in enumerate(nat.LINK_NAMES)}
        self.parts['kuka0'] = self.parts[nat.LINK_NAMES[0]]
        self.robot_parts = {i: n for i, n in enumerate(nat.LINK_NAMES)}

    # ------------------------------------------------------------------ queries
    def _pose(self, kind, index):
        if kind == 'table':
            return np.array(self.object_poses['table'][:3] + [0.0, 0.0, 0.0, 1.0])
        be = self._env._backend()
        if kind == 'object':
            return self._mirror()['obj_pose'][0, index].astype(np.float64)
        return be.link_poses()[0, index].astype(np.float64)

    def _mirror(self):
        """The backend's host mirror of the low-dimensional observations (rr_map_observations): one wait, no copy calls --
        the single-env facade reads joints and touch sensors after every step (env.py:336-339)."""
        be = self._env._backend()
        m = be.map_observations()
        be.sync_observations()
        return m

    def calc_state(self):                               # robot.py:203-211
        return self._mirror()['joints'][0].tolist()

    def get_touch_sensors(self):                        # robot.py:152-163
        return self._mirror()['touch'][0].astype(np.float64)

    def get_contacts(self, forces=False):               # robot.py:131-150
        out = {}
        for c in self._env._backend().contacts(0):
            bodyA, bodyB, link, dist, force = int(c[0]), int(c[1]), int(c[2]), c[9], c[10]
            if not (0 <= bodyA < 16) or abs(dist) >= self.contact_threshold:
                continue
            part = nat.LINK_NAMES[link]
            other = "table" if bodyB < 0 else self.object_names[2 + bodyB - 16]
            out.setdefault(part, []).append([other, float(force)] if forces else other)
        return out

    def reset_object(self, obj_name):                   # robot.py:125-129
        pose = self.object_poses[obj_name]
        self.object_bodies[obj_name].reset_pose(pose[:3], quat_from_euler(*pose[3:]))
