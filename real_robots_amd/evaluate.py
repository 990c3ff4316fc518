"""Local evaluation harness: intrinsic phase then N extrinsic trials with goal scoring
(mirror of real_robots/evaluate.py:16-446; same argument checks, callbacks, score object).

The AIcrowd event sink and the video maker of the reference are out of scope: the
evaluation-state dictionary keeps the reference schema (evaluate.py:100-121) and is handed to an optional
`state_sink` callable instead of aicrowd_api.
"""
import numpy as np

from .policy import BasePolicy
from .registry import make


class EvaluationService:
    def __init__(self, Controller, environment='R1', action_type='macro_action', n_objects=1,
                 intrinsic_timesteps=15e6, extrinsic_timesteps=10e3, extrinsic_trials=50, visualize=True,
                 goals_dataset_path="./goals.npy.npz", video=None, state_sink=None, env_kwargs=None):
        self.ControllerClass = Controller
        self.intrinsic_timesteps = intrinsic_timesteps
        self.extrinsic_timesteps = extrinsic_timesteps
        self.extrinsic_trials = extrinsic_trials
        self.visualize = visualize          # accepted for compatibility; there is no GUI (headless)
        self.goals_dataset_path = goals_dataset_path
        self.state_sink = state_sink
        self.video = None
        self.setup_gym_env(environment, action_type, n_objects, env_kwargs or {})
        self.setup_controller()
        self.setup_evaluation_state()
        self.scores = {}

    def setup_evaluation_state(self):
        self.evaluation_state = {
            "state": "PENDING", "intrinsic_phase_state": "PENDING", "extrinsic_phase_state": "PENDING",
            "max_intrinsic_timesteps": self.intrinsic_timesteps, "max_extrinsic_timesteps": self.extrinsic_timesteps,
            "current_intrinsic_timestep": 0, "max_extrinsic_trials": self.extrinsic_trials,
            "num_extrinsic_trials_complete": 0, "progress_in_current_extrinsic_trial": 0,
            "evaluation_score": {"score": 0, "score_2D": 0, "score_2.5D": 0, "score_3D": 0, "score_total": 0},
            "score": {"score": 0, "score_secondary": 0}}

    def sync_evaluation_state(self):
        if self.state_sink is not None:
            self.state_sink(self.evaluation_state)

    def setup_gym_env(self, environment, action_type, n_objects, env_kwargs):
        if environment not in ["R1", "R2"]:
            raise Exception("Environment type has to be either R1 or R2")
        if action_type == 'macro_action' and environment == 'R2':
            raise Exception("Action type cannot be macro_action in Round 2")
        if action_type not in ['joints', 'cartesian', 'macro_action']:
            raise Exception("Action type has to be either 'joints', 'cartesian',or 'macro_action'")
        if not (isinstance(n_objects, int) and 1 <= n_objects <= 3):
            raise Exception("Number of objects has to be 1, 2 or 3.")
        self.env = make('REALRobot2020-{}{}{}-v0'.format(environment, action_type[0].upper(), n_objects), **env_kwargs)
        self.env.set_goals_dataset_path(self.goals_dataset_path)
        self.env.intrinsic_timesteps = self.intrinsic_timesteps
        self.env.extrinsic_timesteps = self.extrinsic_timesteps
        self.env.extrinsic_trials = self.extrinsic_trials

    def setup_controller(self):
        if not issubclass(self.ControllerClass, BasePolicy):
            raise Exception("Supplied Controller is not a Sub-Class of real_robots.policy.BasePolicy")
        self.controller = self.ControllerClass(self.env.action_space, self.env.observation_space)

    def add_scores(self, challenge, score):
        self.scores.setdefault(challenge, []).append(score)

    def run_intrinsic_phase(self):
        try:
            self._run_intrinsic_phase()
        except Exception:
            self.evaluation_state["state"] = "ERROR"
            self.evaluation_state["intrinsic_phase_state"] = "INTRINSIC_PHASE_ERROR"
            self.sync_evaluation_state()
            raise

    def _run_intrinsic_phase(self):
        if not self.intrinsic_timesteps:
            self.intrinsic_timesteps = 0
        if self.intrinsic_timesteps > 0:
            observation, reward, done = self.env.reset(), 0, False
            self.evaluation_state["intrinsic_phase_state"] = "INTRINSIC_PHASE_IN_PROGRESS"
            self.evaluation_state["state"] = "INTRINSIC_PHASE_IN_PROGRESS"
            self.sync_evaluation_state()
            steps = 0
            self.controller.start_intrinsic_phase()
            while not done:
                action = self.controller.step(observation, reward, done)
                observation, reward, done, _ = self.env.step(action)
                steps += 1
                self.evaluation_state["current_intrinsic_timestep"] = steps
                self.sync_evaluation_state()
            self.evaluation_state["intrinsic_phase_state"] = "INTRINSIC_PHASE_COMPLETE"
            self.evaluation_state["state"] = "INTRINSIC_PHASE_COMPLETE"
            self.sync_evaluation_state()
            self.controller.end_intrinsic_phase(observation, reward, done)
        else:
            print("[WARNING] Skipping Intrinsic Phase as intrinsic_timesteps = 0 or False")
            self.evaluation_state["state"] = "INTRINSIC_PHASE_SKIPPED"
            self.sync_evaluation_state()

    def run_extrinsic_trial(self, trial_number):
        self.env.reset()
        reward, done = 0, False
        observation = self.env.set_goal()
        self.controller.start_extrinsic_trial()
        steps = 0
        while not done:
            action = self.controller.step(observation, reward, done)
            observation, reward, done, _ = self.env.step(action)
            steps += 1
            self.evaluation_state["progress_in_current_extrinsic_trial"] = float(steps) / self.extrinsic_timesteps
            self.sync_evaluation_state()
        self.add_scores(*self.env.evaluateGoal())
        self.evaluation_state["num_extrinsic_trials_complete"] = trial_number + 1
        self.sync_evaluation_state()
        self.controller.end_extrinsic_trial(observation, reward, done)

    def run_extrinsic_phase(self):
        try:
            return self._run_extrinsic_phase()
        except Exception:
            self.evaluation_state["state"] = "ERROR"
            self.evaluation_state["extrinsic_phase_state"] = "EXTRINSIC_PHASE_ERROR"
            self.sync_evaluation_state()
            raise

    def _run_extrinsic_phase(self):
        self.evaluation_state["extrinsic_phase_state"] = "EXTRINSIC_PHASE_IN_PROGRESS"
        self.evaluation_state["state"] = "EXTRINSIC_PHASE_IN_PROGRESS"
        self.sync_evaluation_state()
        self.controller.start_extrinsic_phase()
        for trial in range(int(self.extrinsic_trials)):
            self.run_extrinsic_trial(trial)
            self.build_score_object()
        self.evaluation_state["extrinsic_phase_state"] = "EXTRINSIC_PHASE_COMPLETE"
        self.evaluation_state["score"] = {"score": self.evaluation_state["evaluation_score"]["score_total"],
                                          "score_secondary": self.evaluation_state["evaluation_score"]["score_2D"]}
        self.evaluation_state["meta"] = self.evaluation_state["evaluation_score"]
        self.evaluation_state["state"] = "EVALUATION_COMPLETE"
        self.sync_evaluation_state()
        self.controller.end_extrinsic_phase()
        return self.build_score_object()

    def build_score_object(self):
        total_results = []
        score_object = {}
        for key in ['2D', '2.5D', '3D']:
            results = self.scores.get(key, [])
            score_object["score_{}".format(key)] = np.mean(results) if results else 0
            total_results += results
        score_object["score_total"] = np.mean(total_results) if total_results else float('nan')
        self.evaluation_state["evaluation_score"] = score_object
        self.sync_evaluation_state()
        return score_object


def evaluate(Controller, environment='R1', action_type='macro_action', n_objects=1, intrinsic_timesteps=15e6,
             extrinsic_timesteps=10e3, extrinsic_trials=50, visualize=True, goals_dataset_path="./goals.npy.npz",
             video=None, **kwargs):
    """Signature of real_robots.evaluate (evaluate.py:420-446); returns (score_object, scores)."""
    service = EvaluationService(Controller, environment, action_type, n_objects, intrinsic_timesteps,
                                extrinsic_timesteps, extrinsic_trials, visualize, goals_dataset_path, video, **kwargs)
    service.run_intrinsic_phase()
    service.run_extrinsic_phase()
    return service.build_score_object(), service.scores


def evaluate_batched(Controller, num_envs, environment='R1', action_type='macro_action', n_objects=1,
                     intrinsic_timesteps=1000, extrinsic_timesteps=1000, extrinsic_trials=5,
                     goals_dataset_path="./goals.npy.npz", eye_width=320, eye_height=240, device=0, render_every=0):
    """`evaluate()` for `num_envs` independent agents stepped in lock-step on one GPU (BASELINE config 5): env i owns
    its own `Controller` instance and walks the goal list from goal i (cyclically), so every env sees
    `extrinsic_trials` different goals. Phases, callbacks and the scoring formula are those of the single-env harness
    (evaluate.py:249-259, 282-324; env.py:181-200). Observations are assembled on the host per env; the retina is
    only rendered/fetched on the steps where `render_every` divides the step index (0: never) -- policies that need
    images every step should consume the device buffers of `BatchedREALRobotEnv` instead.
    Returns (score_object, scores) aggregated over all envs and trials."""
    from . import _native as nat
    from .batched import BatchedREALRobotEnv, OBJECT_NAMES
    from .envs.env import REALRobotEnv
    if environment not in ("R1", "R2"):
        raise Exception("Environment type has to be either R1 or R2")
    if action_type == 'macro_action' and environment == 'R2':
        raise Exception("Action type cannot be macro_action in Round 2")
    if action_type not in ('joints', 'macro_action'):
        raise Exception("evaluate_batched supports 'joints' and 'macro_action'")
    if not issubclass(Controller, BasePolicy):
        raise Exception("Supplied Controller is not a Sub-Class of real_robots.policy.BasePolicy")
    proto = REALRobotEnv(objects=n_objects, action_type=action_type, additional_obs=(environment == 'R1'),
                         eye_width=eye_width, eye_height=eye_height)
    proto.set_goals_dataset_path(goals_dataset_path)
    proto.load_goals()
    goals = list(proto.goals)
    names = OBJECT_NAMES[:n_objects]
    N = int(num_envs)
    env = BatchedREALRobotEnv(N, objects=n_objects, width=eye_width, height=eye_height, device=device,
                              want_mask=(environment == 'R1'))
    ctrls = [Controller(proto.action_space, proto.observation_space) for _ in range(N)]
    zero_img = np.zeros((eye_height, eye_width, 3), np.uint8)
    zero_depth = np.zeros((eye_height, eye_width))
    zero_mask = np.zeros((eye_height, eye_width), np.int32)

    def observations(goal_of_env, rendered):
        joints, touch = env.host(nat.F_JOINTS).astype(np.float64), env.host(nat.F_TOUCH).astype(np.float64)
        poses = env.host(nat.F_OBJ_POSE).astype(np.float64)
        rgb = env.host(nat.F_RGB) if rendered else None
        dep = env.host(nat.F_DEPTH) if rendered else None
        msk = env.host(nat.F_MASK) if rendered and environment == 'R1' else None
        out = []
        for i in range(N):
            g = goal_of_env[i]
            o = {"joint_positions": list(joints[i]), "touch_sensors": touch[i],
                 "retina": rgb[i] if rendered else zero_img, "depth": dep[i].astype(np.float64) if rendered else zero_depth,
                 "goal": g.retina if g is not None else zero_img}
            if environment == 'R1':
                o["mask"] = msk[i] if rendered else zero_mask
                o["object_positions"] = {n: poses[i, k, :3] for k, n in enumerate(names)}
                o["goal_mask"] = g.mask if g is not None else zero_mask
                o["goal_positions"] = {n: np.asarray(g.final_state[n])[:3] for n in g.final_state} if g is not None else None
            out.append(o)
        return out

    def run_phase(n_steps, goal_of_env):
        obs = observations(goal_of_env, False)
        done = False
        for t in range(int(n_steps)):
            acts = [c.step(o, 0, done) for c, o in zip(ctrls, obs)]
            rendered = bool(render_every) and (t % render_every == 0)
            if action_type == 'macro_action':
                env.step_macro([a['macro_action'] for a in acts], render=rendered)       # entries may be None (env.py:391-393)
            else:
                env.step(np.array([np.zeros(9) if a['joint_command'] is None else a['joint_command'] for a in acts],
                                  dtype=np.float32), render=rendered)
            done = (t + 1) >= n_steps
            obs = observations(goal_of_env, rendered)
        return obs

    scores = {}
    if intrinsic_timesteps and intrinsic_timesteps > 0:
        env.reset()
        for c in ctrls:
            c.start_intrinsic_phase()
        obs = run_phase(intrinsic_timesteps, [None] * N)
        for c, o in zip(ctrls, obs):
            c.end_intrinsic_phase(o, 0, True)
    for c in ctrls:
        c.start_extrinsic_phase()
    pos_const = -np.log(0.25) / 0.10
    for trial in range(int(extrinsic_trials)):
        env.reset()
        goal_of_env = [goals[(i + trial) % len(goals)] for i in range(N)]
        start = env.host(nat.F_OBJ_POSE)                     # one upload for the whole batch (env.py:159-162 per env)
        for i, g in enumerate(goal_of_env):
            for n in g.initial_state:
                start[i, names.index(n)] = np.asarray(g.initial_state[n], dtype=np.float32)
        env.set_object_poses(start)
        for c in ctrls:
            c.start_extrinsic_trial()
        obs = run_phase(extrinsic_timesteps, goal_of_env)
        poses = env.host(nat.F_OBJ_POSE).astype(np.float64)
        for i, g in enumerate(goal_of_env):
            sc = sum(np.exp(-pos_const * np.linalg.norm(np.asarray(g.final_state[n])[:3] - poses[i, names.index(n), :3]))
                     for n in g.final_state)
            scores.setdefault(g.challenge, []).append(sc)
        for c, o in zip(ctrls, obs):
            c.end_extrinsic_trial(o, 0, True)
    for c in ctrls:
        c.end_extrinsic_phase()
    env.close()
    total, score_object = [], {}
    for key in ['2D', '2.5D', '3D']:
        r = scores.get(key, [])
        score_object["score_{}".format(key)] = np.mean(r) if r else 0
        total += r
    score_object["score_total"] = np.mean(total) if total else float('nan')
    return score_object, scores
