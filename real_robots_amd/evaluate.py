"""Local evaluation harness: intrinsic phase then N extrinsic trials with goal scoring
(mirror of real_robots/evaluate.py:16-446; same argument checks, callbacks, score object).

The AIcrowd event sink of the reference is out of scope: the evaluation-state dictionary keeps
the reference schema (evaluate.py:100-121) and is handed to an optional `state_sink` callable instead of aicrowd_api.
`video=(intrinsic, extrinsic, debug)` makes videos like the reference's (evaluate.py:69-71 -> real_robots_amd/videomaker.py).
"""
import numpy as np

from .policy import BasePolicy, BatchedPolicy
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
        self.video = video
        self.setup_gym_env(environment, action_type, n_objects, env_kwargs or {})
        if self.video:                      # evaluate.py:69-71
            from .videomaker import VideoMaker
            self.videomaker = VideoMaker(self.env, *self.video)
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
            if self.video:
                self.videomaker.start_intrinsic()
            while not done:
                action = self.controller.step(observation, reward, done)
                observation, reward, done, _ = self.env.step(action)
                steps += 1
                self.evaluation_state["current_intrinsic_timestep"] = steps
                self.sync_evaluation_state()
                if self.video:
                    self.videomaker.update_intrinsic(steps)
            if self.video:
                self.videomaker.end_intrinsic()
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
        if self.video:
            self.videomaker.start_trial(observation, trial_number)
        while not done:
            action = self.controller.step(observation, reward, done)
            observation, reward, done, _ = self.env.step(action)
            steps += 1
            self.evaluation_state["progress_in_current_extrinsic_trial"] = float(steps) / self.extrinsic_timesteps
            self.sync_evaluation_state()
            if self.video:
                self.videomaker.extrinsic_trial(observation, action, steps, self.evaluation_state["evaluation_score"])
        if self.video:
            self.videomaker.end_trial()
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
                     goals_dataset_path="./goals.npy.npz", eye_width=320, eye_height=240, device=0, render_every=0, solver=None):
    """`evaluate()` for `num_envs` independent agents stepped in lock-step on one GPU (BASELINE config 5): env i owns
    its own `Controller` instance and walks the goal list from goal i (cyclically), so every env sees
    `extrinsic_trials` different goals. Phases, callbacks and the scoring formula are those of the single-env harness
    (evaluate.py:249-259, 282-324; env.py:181-200). Observations are assembled on the host per env; the retina is
    only rendered/fetched on the steps where `render_every` divides the step index (0: never) -- policies that need
    images every step should consume the device buffers of `BatchedREALRobotEnv` instead.
    A `Controller` derived from `real_robots_amd.policy.BatchedPolicy` takes the batched path instead: ONE controller
    object, one `step()` call per step for all envs, images stay on the device, the only per-step read-back is the
    low-dimensional observation, goal scores are computed on the device (rr_evaluate_goals) -- BASELINE config 5 at 4096 envs.
    `solver`: the constants the reference leaves to pybullet's defaults (`BatchedREALRobotEnv(solver=...)`).
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
    if isinstance(Controller, type) and issubclass(Controller, BatchedPolicy):
        return _evaluate_batched_policy(Controller, num_envs, environment, action_type, n_objects, intrinsic_timesteps,
                                        extrinsic_timesteps, extrinsic_trials, goals_dataset_path, eye_width, eye_height,
                                        device, render_every, solver=solver)[:2]
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
                              want_mask=(environment == 'R1'), solver=solver)
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


def _score_object(scores):
    total, score_object = [], {}
    for key in ['2D', '2.5D', '3D']:
        r = scores.get(key, [])
        score_object["score_{}".format(key)] = np.mean(r) if r else 0
        total += r
    score_object["score_total"] = np.mean(total) if total else float('nan')
    return score_object


def _evaluate_batched_policy(Controller, num_envs, environment, action_type, n_objects, intrinsic_timesteps,
                             extrinsic_timesteps, extrinsic_trials, goals_dataset_path, eye_width, eye_height, device,
                             render_every, env=None, solver=None):
    """The batched-policy path of evaluate_batched (phases: evaluate.py:203-324 of the reference; goal set-up env.py:151-166;
    scoring env.py:181-200).  Per step: one `Controller.step` call, one rr_step / rr_step_plan, one read-back of joints +
    touch (+ object poses in R1).  Env i walks the goal list from goal i (cyclically), trial k takes goal (i + k) % len(goals).
    Returns (score_object, scores, per_env_scores [trials][N], timing dict)."""
    import time
    from . import _native as nat
    from .batched import BatchedREALRobotEnv, OBJECT_NAMES
    from .envs.env import REALRobotEnv
    proto = REALRobotEnv(objects=n_objects, action_type=action_type, additional_obs=(environment == 'R1'),
                         eye_width=eye_width, eye_height=eye_height, device=device)
    proto.set_goals_dataset_path(goals_dataset_path)
    proto.load_goals()
    goals = list(proto.goals)
    names = OBJECT_NAMES[:n_objects]
    N = int(num_envs)
    own_env = env is None
    if own_env:
        env = BatchedREALRobotEnv(N, objects=n_objects, width=eye_width, height=eye_height, device=device,
                                  want_mask=(environment == 'R1'), solver=solver)
    ctrl = Controller(N, proto.action_space, proto.observation_space)
    r1 = environment == 'R1'
    dev_imgs = {"retina": env.device_buffer(nat.F_RGB), "depth": env.device_buffer(nat.F_DEPTH)}
    if r1:
        dev_imgs["mask"] = env.device_buffer(nat.F_MASK)
    H, W = eye_height, eye_width
    # per-goal arrays, gathered once: positions of the final state (NaN: object not named), which objects count, start poses
    G = len(goals)
    g_final = np.full((G, n_objects, 3), np.nan, np.float32)
    g_mask = np.zeros((G, n_objects), np.uint8)
    g_init = np.full((G, n_objects, 7), np.nan, np.float32)
    for k, g in enumerate(goals):
        for n_, pose in g.final_state.items():
            if n_ in names:
                g_final[k, names.index(n_)] = np.asarray(pose, np.float32)[:3]
                g_mask[k, names.index(n_)] = 1
        for n_, pose in g.initial_state.items():
            if n_ in names:
                g_init[k, names.index(n_)] = np.asarray(pose, np.float32)
    challenges = [g.challenge for g in goals]
    zero_goal = np.zeros((N, H, W, 3), np.uint8)
    timing = {"env_steps": 0, "step_seconds": 0.0}

    # the low-dimensional observations come back through the mapped host mirror that the last launch of every step fills
    # (rr_map_observations): ONE wait per step on the event behind it instead of three synchronising copies
    mirror = env.map_observations()

    def observations(goal_idx, goal_img):
        env.sync_observations()
        obs = {"joint_positions": mirror['joints'].copy(), "touch_sensors": mirror['touch'].copy(), "goal": goal_img}
        obs.update(dev_imgs)
        if r1:
            obs["object_positions"] = mirror['obj_pose'].copy()
            obs["goal_positions"] = g_final[goal_idx] if goal_idx is not None else None
        return obs

    def run_phase(n_steps, goal_idx, goal_img):
        obs = observations(goal_idx, goal_img)
        done = False
        t0 = time.perf_counter()
        for t in range(int(n_steps)):
            act = ctrl.step(obs, 0, done)
            rendered = bool(render_every) and (t % render_every == 0)
            if isinstance(act, dict):
                rendered = rendered or bool(np.any(act.get("render", False)))
                act = act["macro_action" if action_type == 'macro_action' else "joint_command"]
            if hasattr(act, 'detach'):                       # a torch tensor: the actions are tiny, bring them to the host
                act = act.detach().cpu().numpy()
            if action_type == 'macro_action':
                env.step_macro(np.asarray(act, dtype=np.float64).reshape(N, 2, 2), render=rendered)
            else:
                env.step(np.asarray(act, dtype=np.float32).reshape(N, 9), render=rendered)
            done = (t + 1) >= n_steps
            obs = observations(goal_idx, goal_img)
        env.sync()
        timing["env_steps"] += N * int(n_steps)
        timing["step_seconds"] += time.perf_counter() - t0
        return obs

    scores, per_env = {}, []
    if intrinsic_timesteps and intrinsic_timesteps > 0:
        env.reset()
        ctrl.start_intrinsic_phase()
        obs = run_phase(intrinsic_timesteps, None, zero_goal)
        ctrl.end_intrinsic_phase(obs, 0, True)
    ctrl.start_extrinsic_phase()
    for trial in range(int(extrinsic_trials)):
        env.reset()
        gi = (np.arange(N) + trial) % G
        start = env.host(nat.F_OBJ_POSE)                     # one upload for the whole batch (env.py:159-162 per env)
        named = ~np.isnan(g_init[gi][:, :, 0])
        start[named] = g_init[gi][named]
        env.set_object_poses(start)
        goal_img = np.stack([goals[k].retina for k in gi]) if N <= 256 else _goal_images(goals, gi, H, W)
        ctrl.start_extrinsic_trial()
        obs = run_phase(extrinsic_timesteps, gi, goal_img)
        sc = env.evaluate_goals(np.nan_to_num(g_final[gi]), g_mask[gi]).astype(np.float64)      # on the device (env.py:181-200)
        per_env.append(sc)
        for i in range(N):
            scores.setdefault(challenges[gi[i]], []).append(float(sc[i]))
        ctrl.end_extrinsic_trial(obs, 0, True)
    ctrl.end_extrinsic_phase()
    if own_env:
        env.close()
    proto.close()
    return _score_object(scores), scores, per_env, timing


def _goal_images(goals, gi, H, W):
    """[N, H, W, 3] goal retinas for a large batch without stacking N Python objects twice: one row per distinct goal, indexed."""
    uniq, inv = np.unique(gi, return_inverse=True)
    table = np.stack([goals[k].retina for k in uniq])
    return table[inv]


def bench_evaluate_batched(num_envs, device=0, width=128, height=128, intrinsic=2000, trials=5, extrinsic=2000, n_goals=512, seed=2020,
                           macro_every=1000, render_every=0):
    """bench.py's config-5 leg in SURVEY 8(d)'s shape: intrinsic phase of 2 000 steps + 5 extrinsic trials x 2 000 steps
    (evaluate.py:249-259,306-318 of the reference), a BatchedPolicy that draws a new uniform macro action per env every 1 000
    steps -- a whole plan of env.py:447-454 runs to its end --, a seeded synthetic goals dataset (generate_goals on the batched
    simulator, reference file format), evaluate_batched end to end.  `render_every` 1: the retina is rendered every step (it stays
    on the device), 0: never.  Returns the `secondary` entry (env-steps/s of the stepping loops incl. policy calls and the
    observation read-back)."""
    import os
    import tempfile
    import time
    from .generate_goals import generate_goals, save_goals
    from .registry import make  # noqa: F401

    class RandomMacro(BatchedPolicy):
        def __init__(self, n, a, o):
            super().__init__(n, a, o)
            self.rng = np.random.default_rng(seed)
            self.t = 0
            self.cur = None

        def start_extrinsic_trial(self):
            self.t = 0

        def step(self, observations, reward, done):
            if self.cur is None or self.t % macro_every == 0:
                self.cur = self.rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(self.num_envs, 2, 2))      # macro_space, env.py:49-52
            self.t += 1
            return self.cur

    t0 = time.perf_counter()
    path = os.path.join(tempfile.gettempdir(), 'rr_goals_s%d_n%d_%dx%d.npy.npz' % (seed, n_goals, width, height))
    if not os.path.exists(path):
        n2, n25 = n_goals // 2, (3 * n_goals) // 10
        goals = generate_goals(n_2d_goals=n2, n_25d_goals=n25, n_3d_goals=n_goals - n2 - n25, n_obj=3, seed=seed,
                               batch=min(2048, max(64, n_goals)), width=width, height=height, device=device, max_rounds=80)
        save_goals(path, goals)
    t_goals = time.perf_counter() - t0
    so, scores, per_env, timing = _evaluate_batched_policy(RandomMacro, num_envs, 'R1', 'macro_action', 3, intrinsic, extrinsic, trials,
                                                          path, width, height, device, render_every)
    return {"workload": "config 5 end to end: real_robots.evaluate_batched, REALRobot2020-R1M3, %d envs, one BatchedPolicy (uniform macro "
                        "actions, a new one every %d steps), intrinsic phase %d steps + %d extrinsic trials x %d steps, %s, %d seeded "
                        "synthetic goals (generate_goals, %dx%d), device-side scores, low-dimensional observations through the mapped host mirror"
                        % (num_envs, macro_every, intrinsic, trials, extrinsic,
                           "retina + depth + mask rendered every step" if render_every == 1 else "no per-step render", n_goals, width, height),
            "value": round(timing["env_steps"] / timing["step_seconds"], 1), "unit": "env-steps/s",
            "ms_per_step": round(timing["step_seconds"] / (timing["env_steps"] / num_envs) * 1e3, 4),
            "steps": timing["env_steps"] // num_envs, "goal_dataset_seconds": round(t_goals, 1),
            "score_total": float(so["score_total"]), "scores_per_challenge": {k: round(float(np.mean(v)), 4) for k, v in scores.items()}}
