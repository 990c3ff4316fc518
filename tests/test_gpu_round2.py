"""GPU tests (-m gpu) added in round 2: EnvCamera view against the oracle, CLI demo, BASELINE config 2 at its size,
checkpoint restore semantics, batched object teleport, macro steps with None actions."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions

pytestmark = pytest.mark.gpu


def test_env_camera_view_matches_the_oracle_pixel_exact():
    """render('rgb_array') (EnvCamera: distance 1.2, yaw 30, pitch -30, roll 0, target [0,0,.4], fov 80, 320x240;
    env.py:83-90,470-513) against an oracle render through the same view / projection matrices: mask identical, RGB
    within one grey level (but for at most two texel-boundary pixels), depth within 1e-6 -- for the reset pose and after 80 steps of arm motion."""
    import real_robots_amd as rr
    from real_robots_amd.mathutil import perspective, view_from_yaw_pitch_roll
    env = rr.make('REALRobot2020-R1J3-v0', eye_width=64, eye_height=64)
    env.reset()
    view = view_from_yaw_pitch_roll([0, 0, .4], 1.2, 30, -30, 0)
    proj = perspective(80, 320.0 / 240.0, 0.1, 100.0)
    o = Oracle(3, 320, 240)
    o.set_camera(view, proj)
    for phase in range(2):
        img = env.render('rgb_array')
        assert img.shape == (240, 320, 3) and img.dtype == np.uint8
        be = env.envCamera._be
        o.state = env._backend().state[0].astype(np.float64)
        r, d, m = o.render()
        assert (m == be.host(nat.F_MASK)[0]).all()
        # coverage and depth are exact; the shading is not contraction-free on the device: a nearest-texel lookup at a texel
        # boundary may flip for a pixel or two (the criterion of the seeded differential test)
        assert (np.abs(r.astype(int) - img.astype(int)).max(-1) > 1).sum() <= 2
        assert np.abs(d - be.host(nat.F_DEPTH)[0]).max() < 1e-6
        assert set(np.unique(m).tolist()) >= {-1, 0, 1}            # background, robot and table are in the oblique view
        for _ in range(80):
            env.step({'joint_command': np.array([0.8, 0.6, 0, -1.0, 0, 0.5, 0, 0.3, 0.2]), 'render': False})
    # the eye camera object of the reference API (env.py:516-567): same view as the observation's retina
    obs, _, _, _ = env.step({'joint_command': np.array([0.8, 0.6, 0, -1.0, 0, 0.5, 0, 0.3, 0.2]), 'render': True})
    rgb, mask, depth = env.eyes['eye'].render(env.robot.object_bodies['table'].get_position())
    assert rgb.shape == (64, 64, 3) and (rgb == obs['retina']).mean() > 0.999 and (mask == obs['mask']).mean() > 0.999
    env.close()


def test_cli_demo_runs(capsys):
    """real-robots-demo (cli.py:23-64): a random policy on REALRobot2020-R2J3-v0, headless."""
    from real_robots_amd import cli
    cli.main(['--steps', '20'])
    out = capsys.readouterr().out
    assert 'ran 20 steps of REALRobot2020-R2J3-v0' in out


def test_config2_1024_envs_one_object_no_render():
    """BASELINE config 2 at its size: REALRobot2020-R2J1, 1024 envs, 1 object (cube), joint control, no render.
    Full-range README-style commands for 400 steps; four envs are followed by the float64 oracle for the first 150 steps
    (before contact switching amplifies rounding), every env must stay finite, and two runs must agree bit for bit."""
    N, T = 1024, 400
    ids = np.arange(N)

    def run(follow):
        env = BatchedREALRobotEnv(N, objects=1, width=64, height=64, want_mask=False)
        orcs = [Oracle(1, 64, 64) for _ in range(4)] if follow else []
        for t in range(T):
            act = synthetic_actions(ids, t, seed=1234)
            env.step(act)
            if t < 150:
                for k, o in enumerate(orcs):
                    o.step(act[k * 300].astype(np.float64))
            if t == 149 and follow:
                st = env.state
                for k, o in enumerate(orcs):
                    assert np.abs(st[k * 300][:11] - o.state[:11]).max() < 2e-3, k
                    assert np.abs(st[k * 300][22:25] - o.state[22:25]).max() < 2e-3, k
        st, ef, ts = env.state, env.host(nat.F_ERRFLAGS), env.host(nat.F_TIMESTEP)
        env.close()
        return st, ef, ts

    a, ef, ts = run(True)
    assert np.isfinite(a).all() and (ef == 0).all() and (ts == T).all()
    z = a[:, 24]
    # a cube hit by the swinging arm may be in flight or on its way down (the out-of-bounds rule re-poses it at the start of the
    # next step once z < 0.08, env.py:257-264); most cubes rest on the table
    assert (z > -1.0).all() and (z < 5.0).all() and (np.abs(z - 0.319) < 2e-3).mean() > 0.8
    b, _, _ = run(False)
    assert (a == b).all()


def test_set_state_clears_the_freeze_bit_contacts_and_touch():
    """rr_set_state is a fresh start: an env frozen by the NaN guard steps again after a valid state is restored, and the
    contact list / touch sensors of the previous step are gone until the next step."""
    N = 3
    env = BatchedREALRobotEnv(N, objects=2, width=64, height=64)
    for _ in range(60):
        env.step(None)
    good = env.state
    assert len(env.contacts(1)) > 0
    bad = good.copy()
    bad[1, 22] = np.nan                        # object position -> non-finite after the next integration
    env.state = bad
    env.step(None)
    assert env.host(nat.F_ERRFLAGS)[1] & 1
    frozen = env.state[1].copy()
    env.step(None)
    assert np.array_equal(env.state[1], frozen, equal_nan=True)          # frozen envs are skipped
    env.state = good
    assert (env.host(nat.F_ERRFLAGS) == 0).all()
    assert len(env.contacts(1)) == 0 and (env.host(nat.F_TOUCH) == 0).all()
    q_before = env.state[1].copy()
    env.step(np.full((N, 9), 0.3, np.float32))
    assert (env.host(nat.F_ERRFLAGS) == 0).all() and np.abs(env.state[1] - q_before).max() > 1e-4
    assert len(env.contacts(1)) > 0
    env.close()


def test_batched_object_teleport_equals_the_per_object_calls():
    N = 5
    a = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    b = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    for e in (a, b):
        for _ in range(20):
            e.step(np.full((N, 9), 0.2, np.float32))
    rng = np.random.default_rng(0)
    poses = a.host(nat.F_OBJ_POSE)
    mask = np.array([1, 0, 1, 1, 0], np.uint8)
    for i in np.flatnonzero(mask):
        for k in range(3):
            poses[i, k] = [rng.uniform(-0.2, 0.0), rng.uniform(-0.3, 0.3), 0.4 + 0.1 * k, 0, 0, 0, 1]
            a.set_object_pose(int(i), k, poses[i, k])
    b.set_object_poses(poses, mask)
    assert (a.state == b.state).all() and (a.host(nat.F_OBJ_POSE) == b.host(nat.F_OBJ_POSE)).all()
    for e in (a, b):
        e.step(None)
    assert (a.state == b.state).all()
    a.close()
    b.close()


def test_step_macro_with_none_actions_keeps_the_plan_position():
    """step_macro with macro_action None: that env steps with zeros(9) and its plan does not advance (env.py:391-393)."""
    N = 4
    env = BatchedREALRobotEnv(N, objects=1, width=64, height=64)
    ref = BatchedREALRobotEnv(N, objects=1, width=64, height=64)
    m = np.tile(np.array([[-0.1, 0.2], [-0.2, -0.3]]), (N, 1, 1))
    env.step_macro([None] * N)                         # nobody has a plan yet: zeros everywhere
    ref.step(None)
    assert (env.state == ref.state).all()
    acts = [m[i] for i in range(N)]
    for t in range(130):
        env.step_macro(acts)
        ref.step_macro(m)
    assert (env.state == ref.state).all()
    plan = env.get_plan(2)
    acts[2] = None
    env.step_macro(acts)                               # env 2 idles for one step ...
    cmd = np.stack([plan[130]] * N)
    cmd[2] = 0
    ref.step(cmd.astype(np.float32))
    assert (env.state == ref.state).all()
    acts[2] = m[2]
    env.step_macro(acts)                               # ... and resumes at the row it had reached
    cmd = np.stack([plan[131]] * N)
    cmd[2] = plan[130]
    ref.step(cmd.astype(np.float32))
    assert (env.state == ref.state).all()
    env.close()
    ref.close()


@pytest.mark.parametrize("W,H", [(128, 128), (320, 240)])
def test_near_plane_clipping_matches_the_oracle(W, H):
    """Links passing within 10 cm of the eye camera (eye (0.01, 0, 1.2) looking down, near plane 0.1, env.py:136-141,548-551):
    triangles that cross the near plane are clipped against it (the reference's TinyRenderer clips at the eye plane and
    discards fragments nearer than the near plane -- the same coverage) instead of being dropped.  Postures that put the
    gripper at z = 1.02 .. 1.12 under the camera, plus the sweep between them: mask and depth identical, RGB within one grey
    level; the frames must actually contain geometry cut by the plane (depth ~ 0)."""
    from oracle.kinematics import inverse_kinematics, quat_from_euler
    targets = [[0.0, 0.0, 1.02], [0.0, 0.05, 1.08], [-0.05, 0.0, 1.12], [0.02, -0.04, 1.10]]
    qs = [inverse_kinematics(np.zeros(11), t, quat_from_euler(0, 0, 0)) for t in targets]
    frames = []
    for a, b in zip(qs[:-1], qs[1:]):
        frames += [a + (b - a) * f for f in np.linspace(0.0, 1.0, 6)[:-1]]
    frames.append(qs[-1])
    N = len(frames)
    env = BatchedREALRobotEnv(N, objects=3, width=W, height=H)
    st = env.state
    for i, q in enumerate(frames):
        st[i, :11] = q
        st[i, 7:11] = [0.4, -0.3, 0.4, -0.3]
    env.state = st
    env.render()
    rgb, dep, msk = env.host(nat.F_RGB), env.host(nat.F_DEPTH), env.host(nat.F_MASK)
    # the float build of the oracle shares the device's forward kinematics bit for bit (test_gpu_contacts_fuzz.py), and the
    # coverage / depth arithmetic is contraction-free on both sides: masks and depths must be identical.  (Against the
    # float64 build a depth next to the near plane moves by 20 x the rounding of w: dz/dw = 2 n f / ((f - n) w^2).)
    o = Oracle(3, W, H, f32=True)
    o64 = Oracle(3, W, H)
    cut = 0
    for i in range(N):
        o.state = env.state[i].astype(np.float64)
        r, d, m = o.render()
        assert (m == msk[i]).all(), i
        assert np.abs(r.astype(int) - rgb[i].astype(int)).max() <= 1, i
        assert (d == dep[i]).all(), i
        o64.state = env.state[i].astype(np.float64)
        r64, d64, m64 = o64.render()
        assert (m64 != msk[i]).sum() <= 2 and (np.abs(d64 - dep[i]) > 1e-4).sum() <= 4, i      # (a sample on an edge may change hands)
        cut += int((d < 0.02).sum())
        assert (m == 0).sum() > 0.05 * W * H, i                  # the arm fills a good part of the frame
    assert cut > 50 * N                                          # ... and is cut by the near plane in these frames
    # a second render after moving on: the incremental image update must cope with fragments of clipped triangles
    env.step(None, render=True)
    o.state = env.state[0].astype(np.float64)
    r, d, m = o.render()
    assert (m == env.host(nat.F_MASK)[0]).all() and (d == env.host(nat.F_DEPTH)[0]).all()
    env.close()


def test_dlpack_and_vector_env_adapter():
    """f4: torch.from_dlpack on the library's device buffers (zero copy) and the gymnasium-style vector env."""
    import torch
    from real_robots_amd.vector import REALRobotVectorEnv
    venv = REALRobotVectorEnv(6, objects=2, eye_width=64, eye_height=64, max_episode_steps=5, device_obs=True)
    obs, info = venv.reset(seed=0)
    assert info == {} and set(obs) == {'joint_positions', 'touch_sensors', 'retina', 'depth'}
    j = torch.from_dlpack(obs['joint_positions'])
    rgb = torch.from_dlpack(obs['retina'])
    assert j.is_cuda and tuple(j.shape) == (6, 9) and tuple(rgb.shape) == (6, 64, 64, 3) and rgb.dtype == torch.uint8
    act = np.tile(np.array([0.3, 0.5, 0, -0.8, 0, 0.4, 0, 0.2, 0.1]), (6, 1))
    for t in range(5):
        obs, rew, term, trunc, info = venv.step(act)
        assert rew.shape == (6,) and not term.any() and trunc.all() == (t == 4)
    # zero copy: the tensors made before the steps see the new observations
    venv._be.sync()
    assert np.allclose(j.cpu().numpy(), venv._be.host(nat.F_JOINTS))
    assert (rgb.cpu().numpy() == venv._be.host(nat.F_RGB)).all()
    # same-step autoreset: the step that truncated the episodes returned their final low-dim observation in infos and reset them
    assert info["_final_obs"].all() and float(info["final_obs"][0]["joint_positions"][1]) > 0.05
    assert (venv._be.host(nat.F_TIMESTEP) == 0).all() and abs(float(j[0, 1])) < 1e-6
    obs, rew, term, trunc, info = venv.step(act)
    assert (venv._be.host(nat.F_TIMESTEP) == 1).all() and not trunc.any() and info == {}
    assert venv.action_space["joint_command"].shape == (6, 9) and venv.observation_space["joint_positions"].shape[0] == 6
    venv.close()


def test_heavy_light_split_is_bitwise_equivalent(monkeypatch):
    """rr_step solves and renders the envs with generic contact rows on side streams, beside the others (DESIGN.md 5.1:
    light / heavy / very heavy).  That is scheduling only: with full-range commands (arms pressed on the table, grippers in the objects:
    a few dozen heavy groups) 240 steps with a render every step must leave bit-identical states, contact forces and
    images whether the split is on or off, and with per-env render flags."""
    N, T = 512, 240
    ids = np.arange(N)
    flags = (np.arange(N) % 3 != 0).astype(np.uint8)

    def run(**envvars):
        for k, v in envvars.items():
            monkeypatch.setenv(k, v)
        env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
        for k in envvars:
            monkeypatch.delenv(k)
        heavy = 0
        for t in range(T):
            env.step(synthetic_actions(ids, t, seed=77), render=(flags if t % 2 else True))
            if t % 40 == 39:
                heavy = max(heavy, sum(len(env.contacts(i)) > 12 for i in range(0, N, 8)))
        out = (env.state, env.host(nat.F_RGB), env.host(nat.F_DEPTH), env.host(nat.F_MASK), env.host(nat.F_TOUCH), env.host(nat.F_ERRFLAGS))
        env.close()
        return out, heavy

    a, heavy = run()
    b, _ = run(RR_NO_SPLIT="1")
    # the third class (very heavy envs on a stream of their own) with a threshold that puts most heavy envs into it, and
    # the heavy envs' render by the three kernels instead of the fused list kernel
    c, _ = run(RR_HEAVY2_MIN="2")
    d, _ = run(RR_HEAVY2_MIN="1000")
    assert heavy >= 2                                   # the run did have envs with generic contacts
    assert (a[5] == 0).all()
    for other in (b, c, d):
        for x, y in zip(a, other):
            assert np.array_equal(x, y)
