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
