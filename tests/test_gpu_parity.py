"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs, plus size-independent properties at the benchmark's full size.

Tolerances (fp32 HIP path vs float64 oracle, stated per test):
  free motion / resting contact, <= 300 steps : |dq| < 1e-4 rad, object position < 1e-3 m
  robot-object contact (grasp script)         : compared against the fp32 build of the oracle over a short
                                                horizon (contact switching amplifies rounding): < 5e-3 m
  images                                      : mask identical, RGB within 1 grey level, depth within 1e-6
                                                (coverage math runs without FMA contraction on both sides)
"""
import numpy as np
import pytest

from oracle.oracle import Oracle
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions

pytestmark = pytest.mark.gpu


def _objs(state):
    return state[22:].reshape(3, 13)


def test_free_motion_and_resting_contact_parity():
    N, T = 16, 300
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    orcs = [Oracle(3, 64, 64) for _ in range(4)]
    for t in range(T):
        act = synthetic_actions(range(N), t, seed=7) * 0.4
        act[:, 7:] = np.abs(act[:, 7:])
        env.step(act)
        for i, o in enumerate(orcs):
            o.step(act[i].astype(np.float64))
    st = env.state
    joints = env.host(nat.F_JOINTS)
    for i, o in enumerate(orcs):
        ref = o.state
        assert np.abs(st[i][:11] - ref[:11]).max() < 1e-4
        assert np.abs(st[i][11:22] - ref[11:22]).max() < 1e-3
        assert np.abs(_objs(st[i])[:, :3] - _objs(ref)[:, :3]).max() < 1e-3
        assert np.abs(joints[i] - o.obs()[0]).max() < 1e-4
    assert (env.host(nat.F_TIMESTEP) == T).all()
    assert (env.host(nat.F_ERRFLAGS) == 0).all()
    env.close()


def _grasp_script(q_home):
    from oracle.kinematics import inverse_kinematics, quat_from_euler
    orient = quat_from_euler(0, 3.14, -1.57)
    q_hi = inverse_kinematics(q_home, [-0.1, 0.0, 0.55], orient)
    q_lo = inverse_kinematics(q_hi, [-0.1, 0.0, 0.47], orient)
    seq = [(q_hi, [0.5, 0.0], 150), (q_lo, [0.5, 0.0], 120), (q_lo, [0.0, 0.0], 60)]
    cmds = []
    for q, g, n in seq:
        cmds += [np.concatenate([q[:7], g])] * n
    return np.array(cmds)


def test_robot_object_contact_parity_and_touch_sensors():
    cmds = _grasp_script(np.zeros(11))
    env = BatchedREALRobotEnv(4, objects=1, width=64, height=64)
    o = Oracle(1, 64, 64, f32=True)
    for _ in range(100):
        env.step(None)
        o.step(None)
    touch_max = 0
    for t, c in enumerate(cmds):
        env.step(np.tile(c.astype(np.float32), (4, 1)))
        o.step(c.astype(np.float32).astype(np.float64))
        if t >= 270:
            touch_max = max(touch_max, env.host(nat.F_TOUCH).max())
        if t in (149, 269, 290):
            st = env.state[0]
            assert np.abs(st[:11] - o.state[:11]).max() < 2e-3, t
            assert np.abs(_objs(st)[0, :3] - _objs(o.state)[0, :3]).max() < 5e-3, t
    st = env.state
    assert np.abs(st - st[0]).max() == 0.0          # identical envs stay bitwise identical
    assert touch_max > 1.0 and o.obs()[1].max() > 1.0
    touch = env.host(nat.F_TOUCH)[0]
    assert touch[[1, 3]].min() > 1.0                # distal skins of both fingers press on the cube
    cont = env.contacts(0)
    assert len(cont) > 0 and (cont[:, 10] >= 0).all()
    env.close()


@pytest.mark.parametrize("W,H", [(128, 128), (320, 240), (256, 256), (64, 48)])
def test_raster_parity(W, H):
    N = 4
    env = BatchedREALRobotEnv(N, objects=3, width=W, height=H)
    o = Oracle(3, W, H)
    rng = np.random.default_rng(3)
    act = rng.uniform(-1.2, 1.2, (N, 9)).astype(np.float32)
    act[:, 7:] = np.abs(act[:, 7:])
    for t in range(120):
        env.step(act, render=(t == 119))
    st = env.state
    rgb, dep, msk = env.host(nat.F_RGB), env.host(nat.F_DEPTH), env.host(nat.F_MASK)
    assert rgb.shape == (N, H, W, 3) and dep.shape == (N, H, W) and msk.dtype == np.int32
    for i in range(N):
        o.state = st[i].astype(np.float64)
        r, d, m = o.render()
        # coverage math is evaluated without FMA contraction on both sides -> the images agree exactly
        same = (m == msk[i])
        assert same.all()
        assert np.abs(r.astype(int) - rgb[i].astype(int)).max() <= 1
        assert np.abs(d - dep[i]).max() < 1e-6
        assert set(np.unique(msk[i]).tolist()) <= {-1, 0, 1, 2, 3, 4}
    env.close()


def test_per_env_render_flags_and_none_action():
    N = 8
    env = BatchedREALRobotEnv(N, objects=2, width=64, height=64)
    env.render()
    before = env.host(nat.F_RGB).copy()
    flags = np.zeros(N, np.uint8)
    flags[::2] = 1
    act = np.zeros((N, 9), np.float32)
    act[:, 1] = 0.8
    for _ in range(40):
        env.step(act, render=flags)
    after = env.host(nat.F_RGB)
    assert (after[1::2] == before[1::2]).all()            # not rendered: buffers untouched
    assert (after[::2] != before[::2]).any()
    q0 = env.state[:, :11].copy()
    env.step(None)                                        # env.py:333-334: None -> zeros(9)
    assert (np.abs(env.state[:, 1]) < np.abs(q0[:, 1])).all()
    env.close()


def test_shard_equivalence_and_determinism():
    """N envs on one device == the same envs split into two shards (bitwise); a repeated run is bitwise equal."""
    N, T = 64, 60
    def run(ids):
        env = BatchedREALRobotEnv(len(ids), objects=3, width=64, height=64)
        for t in range(T):
            env.step(synthetic_actions(ids, t, seed=11) * 0.5)
        s = env.state
        env.close()
        return s
    full = run(list(range(N)))
    again = run(list(range(N)))
    assert (full == again).all()
    a, b = run(list(range(0, 32))), run(list(range(32, 64)))
    assert (np.concatenate([a, b]) == full).all()


def test_reset_mask_set_state_and_oob_reset():
    N = 6
    env = BatchedREALRobotEnv(N, objects=2, width=64, height=64)
    act = np.full((N, 9), 0.3, np.float32)
    for _ in range(30):
        env.step(act)
    moved = env.state
    mask = np.array([1, 0, 1, 0, 0, 0], np.uint8)
    env.reset(mask)
    st = env.state
    assert (st[[0, 2], :22] == 0).all() and np.abs(st[1] - moved[1]).max() == 0
    ts = env.host(nat.F_TIMESTEP)
    assert ts[0] == 0 and ts[1] == 30
    # checkpoint / restore
    env.state = moved
    assert (env.state == moved).all()
    # out-of-bounds object is re-posed on the next step (env.py:257-264)
    env.set_object_pose(3, 0, [0.5, 0.5, 0.02, 0, 0, 0, 1])
    env.step(act)
    p = env.host(nat.F_OBJ_POSE)[3, 0, :3]
    assert np.allclose(p, [-0.1, 0.0, 0.45], atol=1e-3)
    env.close()


def test_link_poses_match_fk_fixture():
    import json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'fk_golden.json')))
    cases = gold['cases']
    env = BatchedREALRobotEnv(len(cases), objects=1, width=64, height=64)
    st = env.state
    for i, c in enumerate(cases):
        cmd = np.array(c['cmd'])
        st[i, :7] = cmd[:7]
        st[i, 7] = st[i, 9] = cmd[7]
        st[i, 8] = st[i, 10] = -cmd[8]
    env.state = st
    lp = env.link_poses()
    for i, c in enumerate(cases):
        for link, pos in c['links'].items():
            assert np.allclose(lp[i, nat.LINK_NAMES.index(link), :3], pos, atol=5e-6), link
    env.close()


def test_full_size_properties_4096_envs():
    """BASELINE config 3 size (4096 envs, 3 objects, 128x128): size-independent checks."""
    N = 4096
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
    ids = np.arange(N)
    for t in range(150):
        act = synthetic_actions(ids % 64, t, seed=5) * 0.3       # 64 distinct action streams, each repeated 64x
        env.step(act, render=(t == 149))
    st = env.state
    assert np.isfinite(st).all() and (env.host(nat.F_ERRFLAGS) == 0).all()
    # replicas of the same action stream are bitwise identical, wherever they sit in the batch
    assert (st.reshape(64, 64, 61) == st.reshape(64, 64, 61)[0:1]).all()
    objs = st[:, 22:].reshape(N, 3, 13)
    top = 0.08 + 0.199403
    assert (objs[:, :, 2] > top).all() and (objs[:, :, 2] < 0.46).all()
    assert np.abs(np.linalg.norm(objs[:, :, 3:7], axis=2) - 1).max() < 1e-5
    msk = env.host(nat.F_MASK)
    assert (msk[:, 64, 64] >= -1).all() and ((msk == 1).sum(axis=(1, 2)) > 500).all()
    rgb = env.host(nat.F_RGB)
    assert (rgb.reshape(64, 64, -1) == rgb.reshape(64, 64, -1)[0:1]).all()
    env.close()


def test_gym_facade_step_contract():
    import real_robots_amd as rr
    env = rr.make('REALRobot2020-R1J3-v0', eye_width=64, eye_height=64)
    obs = env.reset()
    assert set(obs) == {'joint_positions', 'touch_sensors', 'retina', 'depth', 'mask', 'object_positions', 'goal',
                        'goal_mask', 'goal_positions'}
    assert obs['retina'].shape == (64, 64, 3) and obs['retina'].dtype == np.uint8 and obs['retina'].any()
    assert set(obs['object_positions']) == {'cube', 'tomato', 'mustard'}
    env.intrinsic_timesteps = 3
    a = {'joint_command': np.full(9, 0.2), 'render': False}
    for k in range(3):
        obs, reward, done, info = env.step(a)
        assert reward == 0 and info == {} and done == (k == 2)
    assert not obs['retina'].any() and len(obs['joint_positions']) == 9       # camera off -> zero placeholders
    with pytest.raises(AssertionError):
        env.step({'joint_command': np.array([np.nan] * 9), 'render': False})
    with pytest.raises(AssertionError):
        env.step({'joint_command': np.zeros(8), 'render': False})
    assert np.allclose(env.get_part_pos('base'), [-0.55, 0, 1.27], atol=0.05)
    env.close()


def test_env_camera_rgb_array():
    """render('rgb_array'): EnvCamera (distance 1.2, yaw 30, pitch -30, target [0,0,.4], 320x240; env.py:83-90,470-513)."""
    import real_robots_amd as rr
    env = rr.make('REALRobot2020-R2J3-v0', eye_width=64, eye_height=64)
    env.reset()
    assert env.render('human').size == 0
    img = env.render('rgb_array')
    assert img.shape == (240, 320, 3) and img.dtype == np.uint8
    frac_bg = (img == 255).all(-1).mean()
    assert 0.3 < frac_bg < 0.95                     # an oblique view of robot + table, not an empty or full frame
    for _ in range(60):
        env.step({'joint_command': np.array([0.8, 0.6, 0, -1.0, 0, 0.5, 0, 0, 0]), 'render': False})
    img2 = env.render('rgb_array')
    assert (img2 != img).any()                       # follows the simulation state
    env.close()


@pytest.mark.parametrize("N,nobj", [(1, 1), (5, 2), (37, 3), (130, 3)])
def test_odd_batch_sizes_and_object_counts(N, nobj):
    """Batch sizes that are not multiples of the 4-env solver groups / 32-64 thread blocks; 1-3 objects."""
    env = BatchedREALRobotEnv(N, objects=nobj, width=64, height=64)
    o = Oracle(nobj, 64, 64)
    for t in range(80):
        act = synthetic_actions(range(N), t, seed=9) * 0.4
        env.step(act, render=(t == 79))
        o.step(act[N - 1].astype(np.float64))
    st = env.state
    assert np.isfinite(st).all()
    assert np.abs(st[N - 1][:11] - o.state[:11]).max() < 1e-4          # the last env of a ragged block is right
    assert np.abs(_objs(st[N - 1])[:nobj, :3] - _objs(o.state)[:nobj, :3]).max() < 1e-3
    o.state = st[N - 1].astype(np.float64)
    r, d, m = o.render()
    assert (m == env.host(nat.F_MASK)[N - 1]).all()
    assert set(np.unique(m).tolist()) <= set([-1, 0, 1] + list(range(2, 2 + nobj)))
    env.close()


def test_device_side_nonfinite_command_is_flagged_and_skipped():
    """robot.py:189 asserts on the host; a device-resident command cannot be asserted, so the env is flagged
    (RR_F_ERRFLAGS bit 1) and its step is skipped while the other envs advance."""
    import torch
    N = 4
    env = BatchedREALRobotEnv(N, objects=1, width=64, height=64)
    cmd = torch.zeros(N, 9, device='cuda')
    cmd[2, 3] = float('nan')
    env.step(device_ptr=cmd.data_ptr())
    torch.cuda.synchronize()
    flags = env.host(nat.F_ERRFLAGS)
    ts = env.host(nat.F_TIMESTEP)
    assert flags.tolist() == [0, 0, 2, 0] and ts.tolist() == [1, 1, 0, 1]
    cmd[2, 3] = 0.0
    env.step(device_ptr=cmd.data_ptr())
    assert env.host(nat.F_ERRFLAGS).tolist() == [0, 0, 0, 0] and env.host(nat.F_TIMESTEP).tolist() == [2, 2, 1, 2]
    # zero-copy view of an observation buffer as a torch tensor
    j = torch.as_tensor(env.device_buffer(nat.F_JOINTS), device='cuda')
    assert j.shape == (N, 9) and torch.isfinite(j).all()
    env.close()


@pytest.mark.parametrize("pool", [None, 700, 0])
def test_piled_objects_generic_and_overflow_rows_match_oracle(pool, monkeypatch):
    """Three objects dropped into each other next to the gripper: object-object, object-table and robot-object contacts
    at once (up to ~30 per env).  With the default LDS row pool every row is swept from LDS by the pipelined loop; with
    RR_SOLVER_POOL=700 the first env of the workgroup keeps 19 base parts and the others nothing, with 0 every row of
    every env lives in global memory (the overflow path, object-static rows included).  fp32 oracle, short horizon
    (stacking is chaotic): joints < 2e-3 rad, object positions < 5e-3 m for every env; the contact lists must agree
    exactly in size and bodies at the first step."""
    if pool is not None:
        monkeypatch.setenv("RR_SOLVER_POOL", str(pool))
    N = 3
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    o = Oracle(3, 64, 64, f32=True)
    poses = [np.array([-0.10, 0.00, 0.33, 0, 0, 0, 1], np.float32),
             np.array([-0.08, 0.02, 0.36, 0, 0, 0, 1], np.float32),
             np.array([-0.12, -0.01, 0.40, 0.7071068, 0, 0, 0.7071068], np.float32)]
    for k, p in enumerate(poses):
        o.set_object_pose(k, p.astype(np.float64))
        for i in range(N):
            env.set_object_pose(i, k, p)
    from oracle.kinematics import inverse_kinematics, quat_from_euler
    q = inverse_kinematics(np.zeros(11), [-0.1, 0.0, 0.50], quat_from_euler(0, 3.14, -1.57))
    cmd = np.concatenate([q[:7], [0.3, 0.0]]).astype(np.float32)
    max_objobj = max_nc = 0
    for t in range(60):
        env.step(np.tile(cmd, (N, 1)))
        o.step(cmd.astype(np.float64))
        c = env.contacts(0)
        oc = o.contacts()
        if t == 0:
            assert len(c) == len(oc)
            assert (c[:, :3] == oc[:, :3]).all()
        if len(c):
            max_nc = max(max_nc, len(c))
            max_objobj = max(max_objobj, int(((c[:, 0] >= 16) & (c[:, 1] >= 16)).sum()))
        if t in (0, 9, 29):
            for i in range(N):
                st = env.state[i]
                assert np.abs(st[:11] - o.state[:11]).max() < 2e-3, (t, i)
                assert np.abs(_objs(st)[:, :3] - _objs(o.state)[:, :3]).max() < 5e-3, (t, i)
    assert max_objobj > 8 and max_nc > 16, (max_objobj, max_nc)
    st = env.state
    if pool is None:
        assert np.abs(st - st[0]).max() == 0.0      # identical envs on the same path stay bitwise identical
    assert (env.host(nat.F_ERRFLAGS) == 0).all()
    env.close()


def test_every_env_heavy_and_batch_not_a_multiple_of_four(monkeypatch):
    """k_balance with every env in the heavy class (RR_SOLVER_POOL=0 makes any contact 'too much') and 34 envs: the
    positions of the solver order that stay empty are not the last ones then.  Every env must be stepped exactly once
    per step (an env served by two solver groups integrates twice): 120 steps against a run with the default pool,
    joints < 1e-3 rad and object positions < 2e-3 m (rows in global memory take a differently compiled path)."""
    N = 34
    cmds = [synthetic_actions(range(N), t, seed=4) * 0.7 for t in range(120)]

    def run(pool):
        if pool is not None:
            monkeypatch.setenv("RR_SOLVER_POOL", pool)
        env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
        if pool is not None:
            monkeypatch.delenv("RR_SOLVER_POOL")
        for t in range(120):
            env.step(cmds[t])
        st, ts, ef = env.state, env.host(nat.F_TIMESTEP), env.host(nat.F_ERRFLAGS)
        env.close()
        return st, ts, ef

    a, ta, ea = run(None)
    b, tb, eb = run("0")
    assert (ta == 120).all() and (tb == 120).all() and (ea == 0).all() and (eb == 0).all()
    assert np.abs(a[:, :11] - b[:, :11]).max() < 1e-3
    for i in range(N):
        assert np.abs(_objs(a[i])[:, :3] - _objs(b[i])[:, :3]).max() < 2e-3, i


def test_pushing_gripper_one_step_parity():
    """Macro actions (the gripper sweeps over the table and pushes the objects: dozens of robot-object and some
    object-object contacts per env, most of them speculative) on 34 envs.  Every 25 steps the envs with the most
    contacts are checked one step at a time: the fp32 oracle starts from the device state before the step, takes the
    same plan row, and must land on the device state after the step (joints and joint velocities < 2e-4, object
    positions < 1e-4 m, object velocities < 5e-3): this pins the pipelined generic sweep, the skipped friction rows of
    contacts without normal impulse, the shared LDS row pool and the env-to-workgroup dealing of k_balance."""
    N = 34                                      # not a multiple of the four envs of a solver workgroup
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    o = Oracle(3, 64, 64, f32=True)
    rng = np.random.default_rng(5)
    env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
    plans = [env.get_plan(i) for i in range(N)]
    checked = heavy = objobj = 0
    for t in range(800):
        check = t >= 150 and t % 25 == 0
        if check:
            ncs = np.array([len(env.contacts(i)) for i in range(N)])
            sel = np.argsort(-ncs)[:3]
            st0 = env.state
            cont0 = [env.contacts(i) for i in sel]
        env.step_plan(render=False)
        if check:
            st1 = env.state
            for k_, i in enumerate(sel):
                o.state = st0[i].astype(np.float64)
                o.set_contact_cache(cont0[k_])          # contact history of the warm start
                o.step(plans[i][t].astype(np.float64))
                c = env.contacts(i)
                rob = int((((c[:, 0] >= 0) & (c[:, 0] < 16)) | ((c[:, 1] >= 0) & (c[:, 1] < 16))).sum()) if len(c) else 0
                heavy = max(heavy, rob)
                objobj = max(objobj, int(((c[:, 0] >= 16) & (c[:, 1] >= 16)).sum()) if len(c) else 0)
                assert len(c) == len(o.contacts()), (t, i)
                d = np.abs(st1[i] - o.state)
                assert d[:22].max() < 2e-4, (t, i, d[:22].max())
                objs = _objs(st1[i]) - _objs(o.state)
                assert np.abs(objs[:, :3]).max() < 1e-4, (t, i)
                assert np.abs(objs[:, 7:]).max() < 5e-3, (t, i)
                checked += 1
    assert checked > 50 and heavy > 12, (checked, heavy, objobj)    # sweeps with more than a dozen robot contacts were checked
    assert (env.host(nat.F_ERRFLAGS) == 0).all()
    assert (env.host(nat.F_TIMESTEP) == 800).all()                  # k_balance's order reaches every env exactly once per step
    env.close()


def test_soak_full_range_commands_stay_finite_and_reproducible():
    """1024 envs, full-range random commands (arms swing through the objects: robot-object, object-object and
    many-contact states all occur), 400 steps with a render every 7th: no env may report a non-finite state, and two runs
    must agree bit for bit in state and images although fragment lists and work queues are filled in atomic order."""
    N = 1024
    ids = np.arange(N)
    cmds = {k: synthetic_actions(ids, k * 20, hold_prob=0.05) for k in range(20)}

    def run():
        env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
        for t in range(400):
            env.step(cmds[t // 20], render=(t % 7 == 0))
        ef, st, rgb, dep = env.host(nat.F_ERRFLAGS), env.state.copy(), env.host(nat.F_RGB).copy(), env.host(nat.F_DEPTH).copy()
        robot = sum(int((env.contacts(i)[:, 0] < 16).any()) for i in range(0, N, 8) if len(env.contacts(i)))
        env.close()
        return ef, st, rgb, dep, robot

    ef, st, rgb, dep, robot = run()
    assert (ef == 0).all() and np.isfinite(st).all()
    assert robot > 0                                   # the generic solver rows were exercised
    ef2, st2, rgb2, dep2, _ = run()
    assert (st == st2).all() and (rgb == rgb2).all() and (dep == dep2).all()


def test_raster_parity_many_poses_near_camera():
    """24 envs with wide joint commands (links pass close to the camera: near-plane drops, screen-filling slivers, the
    hierarchical block path), three frames each.  Against the float32 build of the oracle (same algorithm, same
    precision as the device): mask identical, RGB within 1 grey level, depth within 1e-5 at every pixel.  Against the
    float64 oracle the same bounds hold except at isolated pixels whose centre lies on a triangle edge to within the
    float32 rounding of the forward kinematics (a few centimetres from the near plane 1/w amplifies the last bit of an
    instance matrix to ~1e-4 pixel): at most 2 such pixels per frame and 4 over the 72 frames x 16384 pixels, and the
    mask must still be identical (measured: 1 pixel)."""
    N, W, H = 24, 128, 128
    env = BatchedREALRobotEnv(N, objects=3, width=W, height=H)
    o, o32 = Oracle(3, W, H), Oracle(3, W, H, f32=True)
    flips = 0
    for t in range(180):
        act = synthetic_actions(range(N), t, seed=11) * 0.8
        env.step(act, render=(t % 60 == 59))
        if t % 60 == 59:
            st, rgb, dep, msk = env.state, env.host(nat.F_RGB), env.host(nat.F_DEPTH), env.host(nat.F_MASK)
            for i in range(N):
                o32.state = st[i].astype(np.float64)
                r, d, m = o32.render()
                assert (m == msk[i]).all(), (t, i)
                assert np.abs(r.astype(int) - rgb[i].astype(int)).max() <= 1, (t, i)
                assert np.abs(d - dep[i]).max() < 1e-5, (t, i)
                o.state = st[i].astype(np.float64)
                r, d, m = o.render()
                assert (m == msk[i]).all(), (t, i)
                bad = (np.abs(r.astype(int) - rgb[i].astype(int)).max(-1) > 1) | (np.abs(d - dep[i]) >= 1e-5)
                assert bad.sum() <= 2, (t, i, int(bad.sum()))
                flips += int(bad.sum())
    assert flips <= 4, flips
    env.close()


@pytest.mark.parametrize("mode,W,H", [("RR_FULL_COPY", 128, 128), ("RR_SEPARATE_RESTORE", 128, 128), ("RR_FULL_COPY", 320, 240)])
def test_incremental_image_update_equals_full_copy(mode, W, H, monkeypatch):
    """The images persist in HBM: a frame only rewrites the pixels of its fragments and puts the pixels the previous frame's
    fragments vacated back to the static layer.  Over 150 steps with wide commands (links sweeping through the image),
    per-env render flags that skip frames at random, an object teleported away, an env reset and a camera change in between, every
    rendered image must equal bit for bit what the two earlier schemes produce: the full copy of the static layer into
    every image before each frame, and the separate restore pass."""
    N = 12                      # 320x240 is rendered in several tiles: fragment lists and markers per tile
    rng = np.random.default_rng(2)
    flags = [(rng.random(N) < 0.6).astype(np.uint8) for _ in range(150)]

    def run(env_var):
        if env_var:
            monkeypatch.setenv(env_var, "1")
        env = BatchedREALRobotEnv(N, objects=3, width=W, height=H)
        if env_var:
            monkeypatch.delenv(env_var)
        frames = []
        for t in range(150):
            act = synthetic_actions(range(N), t, seed=3) * 0.9
            if t == 70:
                env.set_object_pose(2, 0, np.array([0.05, 0.3, 0.5, 0, 0, 0, 1], np.float32))
            if t == 100:
                m = np.zeros(N, np.uint8); m[5] = 1
                env.reset(m)
            if t == 120:                            # a new camera: new static layer, the next frame starts from a full copy
                from real_robots_amd.mathutil import look_at, perspective
                env.set_camera(look_at(np.array([0.3, 0.2, 1.1]), np.array([0.0, 0.0, 0.2]), np.array([0.0, 0.0, 1.0])),
                               perspective(70.0, W / H, 0.1, 100.0))
            env.step(act, render=flags[t])
            if t % 10 == 9:
                frames.append((env.host(nat.F_RGB).copy(), env.host(nat.F_DEPTH).copy(), env.host(nat.F_MASK).copy()))
        env.close()
        return frames

    a, b = run(None), run(mode)
    assert len(a) == len(b) == 15
    for k, ((r0, d0, m0), (r1, d1, m1)) in enumerate(zip(a, b)):
        assert (r0 == r1).all() and (d0 == d1).all() and (m0 == m1).all(), k
    assert len({fr[0].tobytes() for fr in a}) == 15           # the frames differ from each other


def test_known_answers_hold_on_the_device_path():
    """The analytic known answers of tests/test_oracle_pins.py evaluated on the HIP path itself (no oracle involved):
    resting contact forces sum to m g; a cube launched at 0.4 m/s along +y decelerates at mu g + v (0.04 + 0.04 v)
    with mu = 0.5 and stops; free fall follows semi-implicit Euler."""
    env = BatchedREALRobotEnv(4, objects=3, width=64, height=64)
    z0 = env.state[0, 22 + 2]
    env.step(None)
    z1 = env.state[0, 22 + 2]
    assert abs((z0 - z1) - 9.81 * 0.005 ** 2 * (1 - 0.0)) < 2e-6          # first step of free fall: dz = g dt^2 (v0 = 0)
    for _ in range(400):
        env.step(None)
    c = env.contacts(0)
    for ob, mass in enumerate((1.5, 3.0, 2.0)):
        sel = c[:, 0] == 16 + ob
        assert 3 <= sel.sum() <= 4
        assert abs(c[sel, 10].sum() - mass * 9.81) < 5e-3 * mass * 9.81, (ob, c[sel, 10].sum())
    st = env.state.copy()
    st[:, 22 + 8] = 0.4
    env.state = st
    v_prev, sliding = 0.4, 0
    for k in range(40):
        env.step(None)
        v = float(env.state[0, 22 + 8])
        if v > 0.02:
            a = (v_prev - v) / 0.005
            assert abs(a - (0.5 * 9.81 + v_prev * (0.04 + 0.04 * v_prev))) < 2e-2, (k, a)
            sliding += 1
        v_prev = v
    assert sliding >= 12 and abs(env.state[0, 22 + 8]) < 1e-5
    env.close()


def test_object_home_poses_follow_edits_of_object_poses():
    """Kuka.object_poses is edited in place by callers of the reference (tests/test_actions.py:95-98 parks the objects on the
    shelf: x and z + 0.3, then reset_object); reset and the out-of-bounds rule (env.py:257-264) must use the edited poses."""
    import real_robots_amd as rr
    env = rr.make('REALRobot2020-R2J3-v0', eye_width=64, eye_height=64)
    env.reset()
    for obj in env.robot.used_objects[1:]:
        env.robot.object_poses[obj][0] += 0.3
        env.robot.object_poses[obj][2] += 0.3
        env.robot.reset_object(obj)
    zero = {'joint_command': np.zeros(9), 'render': False}
    for _ in range(200):
        env.step(zero)
    shelf = {o: env.get_obj_pos(o) for o in env.robot.used_objects[1:]}
    assert all(p[0] > 0.15 and p[2] > 0.4 for p in shelf.values()), shelf           # resting on the shelf
    # push the cube off the world: the out-of-bounds rule brings it back to the EDITED pose, not the blob's
    env.robot.object_bodies['cube'].reset_pose([0.5, 0.0, 0.02], [0, 0, 0, 1])
    env.step(zero)
    p = env.get_obj_pos('cube')
    assert abs(p[0] - 0.2) < 0.01 and abs(p[2] - 0.75) < 0.02, p
    env.reset()                                                                      # reset uses the edited poses as well
    p = env.get_obj_pos('tomato')
    assert abs(p[0] - 0.2) < 1e-3 and abs(p[1] + 0.3) < 1e-3 and abs(p[2] - 0.75) < 1e-3, p
    env.close()
    # the batched API: per-env homes
    b = BatchedREALRobotEnv(4, objects=3, width=64, height=64)
    b.set_object_home(2, 0, [0.0, 0.2, 0.5, 0, 0, 0, 1])
    b.reset()
    pose = b.host(nat.F_OBJ_POSE)
    assert np.abs(pose[2, 0, :3] - [0.0, 0.2, 0.5]).max() < 1e-6 and np.abs(pose[1, 0, :3] - [-0.1, 0.0, 0.45]).max() < 1e-6
    b.close()
