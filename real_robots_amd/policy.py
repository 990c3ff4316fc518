"""Controller interface of the REAL competition (mirror of real_robots/policy.py:1-103).

A controller is constructed as `Controller(action_space, observation_space)` (evaluate.py:189) and asked for one
action per environment step; the six phase callbacks are optional.
"""


class BasePolicy:
    def __init__(self, action_space, observation_space):
        self.action_space = action_space
        self.observation_space = observation_space

    def step(self, observation, reward, done):
        """Return the next action dict for `observation` (dict with "joint_positions", "touch_sensors", "retina",
        "goal" and, in R1 environments, "object_positions", "goal_positions", "mask", "goal_mask"). `reward` is
        always 0; `done` turns True when the intrinsic phase or an extrinsic trial ends."""
        raise NotImplementedError("Derive your controller from BasePolicy and implement step().")

    def start_intrinsic_phase(self):
        """Called once before the first intrinsic-phase observation."""

    def end_intrinsic_phase(self, observation, reward, done):
        """Called with the last observation of the intrinsic phase."""

    def start_extrinsic_phase(self):
        """Called once before the extrinsic trials."""

    def end_extrinsic_phase(self):
        """Called after the last extrinsic trial."""

    def start_extrinsic_trial(self):
        """Called before each extrinsic trial; the next observation carries a new goal."""

    def end_extrinsic_trial(self, observation, reward, done):
        """Called with the last observation of an extrinsic trial."""


class BatchedPolicy:
    """One controller object for a whole batch of N envs stepped in lock-step on one GPU (`evaluate_batched`): the batched
    counterpart of BasePolicy -- same six phase callbacks, same per-step question, asked ONCE per step for all envs.

    Constructed as `Controller(num_envs, action_space, observation_space)` (the single-env spaces).  `step` receives a dict
    of batched observations and returns the actions of all envs:
      * observations: "joint_positions" [N, 9], "touch_sensors" [N, 4] and, in R1 environments, "object_positions"
        [N, n_objects, 7 (xyz + xyzw)] and "goal_positions" [N, n_objects, 3] (NaN where a goal does not name the object) as
        numpy arrays (low-dimensional: the only per-step read-back); "retina" [N, H, W, 3] u8, "depth" [N, H, W] f32 and "mask"
        [N, H, W] i32 as DEVICE buffers (`__dlpack__` / `__cuda_array_interface__`: `torch.from_dlpack(obs["retina"])`, zero
        copy) -- rewritten in place by every step that renders; "goal" [N, H, W, 3] u8 (numpy, constant during a trial);
      * return value: macro actions [N, 2, 2] (action_type 'macro_action') or joint commands [N, 9] ('joints') as a numpy
        array, or a dict {"macro_action" | "joint_command": array, "render": bool} to ask for the camera on that step.
    """

    def __init__(self, num_envs, action_space, observation_space):
        self.num_envs = int(num_envs)
        self.action_space = action_space
        self.observation_space = observation_space

    def step(self, observations, reward, done):
        raise NotImplementedError("Derive your controller from BatchedPolicy and implement step().")

    def start_intrinsic_phase(self):
        pass

    def end_intrinsic_phase(self, observations, reward, done):
        pass

    def start_extrinsic_phase(self):
        pass

    def end_extrinsic_phase(self):
        pass

    def start_extrinsic_trial(self):
        pass

    def end_extrinsic_trial(self, observations, reward, done):
        pass
