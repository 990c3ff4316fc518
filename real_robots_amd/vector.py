"""Vector-env adapter: N REALRobot envs behind the gymnasium `VectorEnv` calling convention (reset(seed, options) ->
(obs, info); step(actions) -> (obs, rewards, terminations, truncations, infos)), on top of BatchedREALRobotEnv.

The reference has no vectorised env (one env per process); this is the adoption surface SURVEY 8(f4) asks for.
When `gymnasium` is importable the class derives from `gymnasium.vector.VectorEnv`, otherwise it is a plain class with
the same methods and attributes (`num_envs`, `single_action_space`, `single_observation_space`, `action_space`,
`observation_space`).  Episodes end like the reference's (env.py:345-352): `truncated` once `timestep >=
max_episode_steps`.  Autoreset is gymnasium's SAME-STEP mode (`metadata["autoreset_mode"] = "same_step"`): a truncated env
is reset inside the step() that truncated it, the returned observation is the first one of its next episode and the last
observation of the finished episode is in `infos["final_obs"]` (an object array: a dict of the low-dim entries for every finished
env, None elsewhere; `infos["_final_obs"]` is the mask of the envs it applies to) -- every action acts on the episode its observation came from, no action is dropped.
Observations are batched numpy arrays; with `device_obs=True` the image / low-dim entries are the library's device
buffers instead (DLPack / __cuda_array_interface__, zero copy into torch on ROCm).
"""
import numpy as np

from . import _native as nat
from . import spaces
from .batched import BatchedREALRobotEnv
from .envs.robot import Kuka

try:                                    # optional dependency
    from gymnasium.vector import VectorEnv as _Base
except Exception:                       # pragma: no cover - gymnasium is not installed in the build container
    _Base = object


def _batch_dict_space(space, n):
    """Dict of Boxes -> the same Dict with a leading axis of n (other entry types are kept as they are)."""
    out = {}
    for k, sp in space.spaces.items():
        if hasattr(sp, 'low') and hasattr(sp, 'high'):
            out[k] = spaces.Box(low=np.broadcast_to(sp.low, (n,) + tuple(sp.shape)).copy(),
                                high=np.broadcast_to(sp.high, (n,) + tuple(sp.shape)).copy(), dtype=sp.dtype)
        else:
            out[k] = sp
    return spaces.Dict(out)


class REALRobotVectorEnv(_Base):
    def __init__(self, num_envs, objects=3, additional_obs=False, eye_width=320, eye_height=240, device=0,
                 max_episode_steps=int(15e6), render_every_step=True, device_obs=False, solver=None):
        self.num_envs = int(num_envs)
        self._robot = Kuka(additional_obs, objects, eye_width, eye_height, env=None)
        self.single_action_space = spaces.Dict({"joint_command": self._robot.action_space, "render": spaces.MultiBinary(1)})
        self.single_observation_space = self._robot.observation_space
        # batched spaces = the single spaces with a leading env axis (gymnasium.vector.utils.batch_space)
        self.action_space = spaces.Dict({
            "joint_command": spaces.Box(low=np.tile(self._robot.min_joints, (self.num_envs, 1)),
                                        high=np.tile(self._robot.max_joints, (self.num_envs, 1)), dtype=float),
            "render": spaces.MultiBinary(self.num_envs)})
        self.observation_space = _batch_dict_space(self.single_observation_space, self.num_envs)
        self.metadata = {"autoreset_mode": "same_step"}
        self.max_episode_steps = int(max_episode_steps)
        self.render_every_step, self.device_obs, self.additional_obs = bool(render_every_step), bool(device_obs), bool(additional_obs)
        self._be = BatchedREALRobotEnv(self.num_envs, objects=objects, width=eye_width, height=eye_height, device=device,
                                       want_mask=additional_obs, solver=solver)
        # host-side episode clocks (no device read-back per step): the batched env behind this adapter is private to it, every
        # path that resets an env goes through reset() / step() below and resets its clock with it
        self._steps = np.zeros(self.num_envs, np.int64)

    # ------------------------------------------------------------------ observations
    def _obs(self, rendered):
        be = self._be
        get = be.device_buffer if self.device_obs else be.host
        obs = {"joint_positions": get(nat.F_JOINTS), "touch_sensors": get(nat.F_TOUCH)}
        if rendered:
            obs["retina"], obs["depth"] = get(nat.F_RGB), get(nat.F_DEPTH)
            if self.additional_obs:
                obs["mask"] = get(nat.F_MASK)
        if self.additional_obs:
            obs["object_positions"] = get(nat.F_OBJ_POSE)
        return obs

    def reset(self, *, seed=None, options=None):
        self._be.reset()
        self._steps[:] = 0
        if self.render_every_step:
            self._be.render()
        return self._obs(self.render_every_step), {}

    def step(self, actions):
        """actions: float array [N, 9] of joint commands (or a dict with "joint_command" [N, 9] and optional "render")."""
        render = self.render_every_step
        if isinstance(actions, dict):
            render = bool(np.any(actions.get("render", render)))
            actions = actions["joint_command"]
        self._be.step(np.asarray(actions, dtype=np.float32), render=render)
        self._steps += 1
        truncated = self._steps >= self.max_episode_steps
        n = self.num_envs
        infos = {}
        if truncated.any():
            # same-step autoreset: keep the finished episodes' last low-dim observation, reset those envs, re-render them
            # (gymnasium's layout: an object array with one dict per finished env and None elsewhere, plus the boolean mask)
            j, tc = self._be.host(nat.F_JOINTS), self._be.host(nat.F_TOUCH)
            op = self._be.host(nat.F_OBJ_POSE) if self.additional_obs else None
            fin = np.full(n, None, dtype=object)
            for i in np.flatnonzero(truncated):
                fin[i] = {"joint_positions": j[i], "touch_sensors": tc[i]}
                if op is not None:
                    fin[i]["object_positions"] = op[i]
            infos["final_obs"] = fin
            infos["_final_obs"] = truncated.copy()
            self._be.reset(truncated.astype(np.uint8))
            self._steps[truncated] = 0
            if render:
                self._be.render()
        return self._obs(render), np.zeros(n), np.zeros(n, bool), truncated, infos

    def close(self, **kwargs):
        self._be.close()
