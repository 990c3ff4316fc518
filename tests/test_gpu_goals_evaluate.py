"""GPU tests (-m gpu) of the callers either side of the path: goal dataset generation on the
batched simulator, the reference-compatible dataset format, and real_robots.evaluate() end to end."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_generate_goals_and_evaluate_end_to_end(tmp_path):
    import real_robots            # the alias package: existing agents import this name
    from real_robots.policy import BasePolicy
    from real_robots_amd.generate_goals import generate_goals, is_on_table, save_goals
    goals = generate_goals(n_2d_goals=2, n_25d_goals=1, n_3d_goals=1, n_obj=1, seed=7, batch=32, width=64, height=64)
    assert [g.challenge for g in goals] == ['2D', '2D', '2.5D', '3D']
    g = goals[0]
    assert set(g.initial_state) == {'cube'} and g.retina.shape == (64, 64, 3) and g.mask.shape == (64, 64)
    assert is_on_table('cube', g.final_state['cube'][2]) and is_on_table('cube', g.initial_state['cube'][2])
    assert np.linalg.norm(g.final_state['cube'][:2] - g.initial_state['cube'][:2]) >= 0.2
    assert (g.mask == 2).sum() > 0 and g.retina.any()
    path = str(tmp_path / 'goals.npy.npz')
    save_goals(path, goals)

    calls = []

    class Pusher(BasePolicy):
        def start_intrinsic_phase(self):
            calls.append('si')

        def start_extrinsic_trial(self):
            calls.append('st')

        def end_extrinsic_trial(self, observation, reward, done):
            calls.append('et')

        def step(self, observation, reward, done):
            assert set(observation) >= {'joint_positions', 'touch_sensors', 'retina', 'goal', 'object_positions', 'goal_positions'}
            return {'macro_action': np.array([[-0.1, -0.2], [0.0, 0.2]]), 'render': False}

    result, scores = real_robots.evaluate(Pusher, environment='R1', action_type='macro_action', n_objects=1,
                                          intrinsic_timesteps=30, extrinsic_timesteps=40, extrinsic_trials=3,
                                          visualize=False, goals_dataset_path=path,
                                          env_kwargs=dict(eye_width=64, eye_height=64))
    assert calls == ['si', 'st', 'et', 'st', 'et', 'st', 'et']
    assert set(result) == {'score_2D', 'score_2.5D', 'score_3D', 'score_total'}
    assert len(scores['2D']) == 2 and len(scores['2.5D']) == 1
    assert 0 <= result['score_total'] <= 1          # one object: exp(-ln4/0.1 * dist) in (0, 1]
    # the goal image shown during the extrinsic phase is the dataset's final-state retina
    env = real_robots.make('REALRobot2020-R1J1-v0', eye_width=64, eye_height=64)
    env.set_goals_dataset_path(path)
    env.reset()
    obs = env.set_goal()
    assert (obs['goal'] == goals[0].retina).all() and env.goal_idx == 0
    p = env.get_obj_pos('cube')
    assert np.allclose(p, goals[0].initial_state['cube'][:3], atol=1e-3)
    challenge, score = env.evaluateGoal()
    d = np.linalg.norm(goals[0].final_state['cube'][:3] - p)
    assert challenge == '2D' and abs(score - np.exp(np.log(0.25) / 0.10 * d)) < 1e-3     # env.py:181-200
    env.close()


def test_evaluate_batched_matches_single_env_scores(tmp_path):
    """N envs with identical scripted controllers: every env reproduces the single-env harness' score for the goal
    it is given (same physics, same scoring), and the per-challenge aggregation has N x trials entries."""
    import real_robots_amd as rr
    from real_robots_amd.generate_goals import generate_goals, save_goals
    goals = generate_goals(n_2d_goals=2, n_25d_goals=0, n_3d_goals=0, n_obj=1, seed=3, batch=16, width=64, height=64)
    path = str(tmp_path / 'g.npy.npz')
    save_goals(path, goals)

    class Scripted(rr.BasePolicy):
        def step(self, observation, reward, done):
            return {'macro_action': np.array([[-0.1, -0.25], [-0.1, 0.25]]), 'render': False}

    kw = dict(environment='R1', action_type='macro_action', n_objects=1, intrinsic_timesteps=20,
              extrinsic_timesteps=60, extrinsic_trials=2, goals_dataset_path=path)
    res_b, scores_b = rr.evaluate_batched(Scripted, 6, eye_width=64, eye_height=64, **kw)
    res_s, scores_s = rr.evaluate(Scripted, visualize=False, env_kwargs=dict(eye_width=64, eye_height=64), **kw)
    assert len(scores_b['2D']) == 12
    # env 0 walks the goals in the same order as the single env: trial 0 -> goal 0, trial 1 -> goal 1
    assert abs(scores_b['2D'][0] - scores_s['2D'][0]) < 1e-3 and abs(scores_b['2D'][6] - scores_s['2D'][1]) < 1e-3
    assert 0 < res_b['score_total'] <= 1


def test_generated_goals_are_rest_states_of_the_oracle_and_render_the_same():
    """SURVEY 8(f2) against the oracle, not only its own properties: every generated goal state (settled on the HIP
    path) is a rest state of the CPU restatement too - placed there, the oracle's objects stay within 1 mm / 0.5 deg
    over 60 zero-command steps - and the oracle's render of that state reproduces the goal's retina and mask."""
    from oracle.oracle import Oracle
    from real_robots_amd.generate_goals import generate_goals, OBJECT_NAMES
    W, H, n_obj = 96, 72, 3
    goals = generate_goals(n_2d_goals=2, n_25d_goals=2, n_3d_goals=2, n_obj=n_obj, seed=11, batch=64, width=W, height=H)
    assert [g.challenge for g in goals] == ['2D'] * 2 + ['2.5D'] * 2 + ['3D'] * 2
    o = Oracle(n_objects=n_obj, width=W, height=H, f32=True)
    for g in goals:
        for state, check_image in ((g.initial_state, False), (g.final_state, True)):
            o.reset()
            for _ in range(30):
                o.step(None)
            for i, name in enumerate(OBJECT_NAMES[:n_obj]):
                o.set_object_pose(i, np.asarray(state[name], np.float64))
            if check_image:
                rgb, depth, mask = o.render()
                assert (mask == g.mask).mean() > 0.995, (g.challenge, (mask == g.mask).mean())
                close = (np.abs(rgb.astype(int) - g.retina.astype(int)).max(-1) <= 1)
                assert close.mean() > 0.99, (g.challenge, close.mean())
            for _ in range(60):
                o.step(None)
            s = o.state[22:].reshape(3, 13)          # per object: position, quaternion, velocities
            for i, name in enumerate(OBJECT_NAMES[:n_obj]):
                assert np.linalg.norm(s[i, :3] - state[name][:3]) < 1e-3, (g.challenge, name, s[i, :3], state[name][:3])
                qdot = abs(np.dot(s[i, 3:7], state[name][3:7]))
                assert qdot > np.cos(np.radians(0.25)), (g.challenge, name, qdot)      # rotated by < 0.5 deg


def test_evaluate_batched_at_4096_envs_matches_the_single_env_harness(tmp_path):
    """BASELINE config 5 at its size: 4096 envs, REALRobot2020-R1M3 (macro actions, 3 objects), a seeded 4096-goal dataset
    from generate_goals, ONE BatchedPolicy object driving all envs (images stay on the device, low-dim read-back only, scores
    from rr_evaluate_goals on the device), intrinsic phase + 2 extrinsic trials.  Envs 0..7 reproduce the scores the single-env
    harness (real_robots.evaluate, per-env BasePolicy, facade env) gets for the goals they were given."""
    import time
    import real_robots_amd as rr
    from real_robots_amd.evaluate import _evaluate_batched_policy
    from real_robots_amd.generate_goals import generate_goals, save_goals
    N, W, H = 4096, 64, 64
    t0 = time.time()
    goals = generate_goals(n_2d_goals=2048, n_25d_goals=1229, n_3d_goals=819, n_obj=3, seed=2020, batch=2048, width=W, height=H,
                           max_rounds=80)
    assert len(goals) == N
    path = str(tmp_path / 'goals4096.npy.npz')
    save_goals(path, goals)
    t_goals = time.time() - t0
    macro = np.array([[-0.1, -0.25], [-0.1, 0.25]])
    calls = []

    class ScriptedB(rr.BatchedPolicy):
        def start_intrinsic_phase(self):
            calls.append('si')

        def start_extrinsic_trial(self):
            calls.append('st')

        def end_extrinsic_trial(self, observations, reward, done):
            calls.append('et')
            assert done and observations["object_positions"].shape == (self.num_envs, 3, 7)

        def step(self, observations, reward, done):
            n = self.num_envs
            assert observations["joint_positions"].shape == (n, 9) and observations["touch_sensors"].shape == (n, 4)
            assert hasattr(observations["retina"], '__dlpack__') and observations["goal"].shape == (n, H, W, 3)
            return np.broadcast_to(macro, (n, 2, 2))

    t0 = time.time()
    so, scores, per_env, timing = _evaluate_batched_policy(ScriptedB, N, 'R1', 'macro_action', 3, 200, 200, 2, path, W, H, 0, 0)
    t_eval = time.time() - t0
    print("config 5 at size: %d goals in %.1f s; evaluate_batched %.1f s, %.0f env-steps/s in the stepping loops"
          % (N, t_goals, t_eval, timing["env_steps"] / timing["step_seconds"]))
    assert calls == ['si', 'st', 'et', 'st', 'et'] and timing["env_steps"] == N * 600
    del calls[:]
    assert sum(len(v) for v in scores.values()) == 2 * N and set(scores) == {'2D', '2.5D', '3D'}
    assert len(per_env) == 2 and per_env[0].shape == (N,) and 0 < so['score_total'] <= 3
    # public entry point, same run
    so2, scores2 = rr.evaluate_batched(ScriptedB, 64, environment='R1', action_type='macro_action', n_objects=3,
                                       intrinsic_timesteps=20, extrinsic_timesteps=30, extrinsic_trials=1,
                                       goals_dataset_path=path, eye_width=W, eye_height=H)
    assert sum(len(v) for v in scores2.values()) == 64

    class Scripted(rr.BasePolicy):
        def step(self, observation, reward, done):
            return {'macro_action': macro, 'render': False}

    for i in range(8):
        sub = str(tmp_path / ('g%d.npy.npz' % i))
        save_goals(sub, [goals[i], goals[(i + 1) % N]])
        _, sc = rr.evaluate(Scripted, environment='R1', action_type='macro_action', n_objects=3, intrinsic_timesteps=200,
                            extrinsic_timesteps=200, extrinsic_trials=2, visualize=False, goals_dataset_path=sub,
                            env_kwargs=dict(eye_width=W, eye_height=H))
        flat = [sc[goals[i].challenge][0]] if goals[i].challenge != goals[(i + 1) % N].challenge else None
        single = [s for k in (goals[i].challenge, goals[(i + 1) % N].challenge) for s in sc[k]]
        if flat is None:
            single = sc[goals[i].challenge]                   # both trials under one challenge, in trial order
        else:
            single = [sc[goals[i].challenge][0], sc[goals[(i + 1) % N].challenge][0]]
        assert abs(single[0] - per_env[0][i]) < 1e-3 and abs(single[1] - per_env[1][i]) < 1e-3, (i, single, per_env[0][i], per_env[1][i])


def test_goal_generator_repeatability_and_object_distance_predicate():
    """generate_goals.py:229-246 checkRepeatability (batched: a goal's objects put back at its initial state settle where they
    were) and :296-338 max_objects_dist (3D goals with several objects: two of them within the distance in the initial or the
    final state) -- the two pieces of the reference's generator that were not restated before round 5."""
    from real_robots_amd.generate_goals import generate_goals, two_near_objects
    goals, rep = generate_goals(n_2d_goals=2, n_25d_goals=1, n_3d_goals=3, n_obj=3, seed=11, batch=48, width=64, height=64,
                                max_objects_dist=0.3, repeatability=True)
    assert [g.challenge for g in goals] == ['2D', '2D', '2.5D', '3D', '3D', '3D']
    assert rep != 1000000                                   # every placement settled again
    max_pos, max_or = rep
    assert max_pos < 2e-3 and max_or < 2e-2, rep            # settled states are fixed points of "re-pose and settle" (m / quaternion norm)
    for g in goals[3:]:
        ini = np.vstack([g.initial_state[k] for k in g.initial_state])
        fin = np.vstack([g.final_state[k] for k in g.final_state])
        assert two_near_objects(ini, 0.3) or two_near_objects(fin, 0.3)
