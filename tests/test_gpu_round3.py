"""GPU tests (-m gpu) of round 3, through the C ABI:
* the look-ahead (state part of step t+1 under the render of step t) is scheduling only: bitwise the same run
  as preparing every step in line, with resets, teleports, home-pose edits and state restores in between;
* checkpoint save / restore continues a run bit for bit (state + contact history of the warm start);
* bench.py's exact workload at its size (4096 envs, full-range device-resident commands, render every step, three-stream
  split) followed by the float oracle, and bitwise equal to the unsplit / in-line step;
* an edited eye camera reaches the retina (env.py:136-141, 249-255).
"""
import os

import numpy as np
import pytest

from oracle.oracle import Oracle
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions

pytestmark = pytest.mark.gpu


def _make(monkeypatch, env_vars, *args, **kw):
    for k, v in env_vars.items():
        monkeypatch.setenv(k, v)
    try:
        return BatchedREALRobotEnv(*args, **kw)
    finally:
        for k in env_vars:
            monkeypatch.delenv(k, raising=False)


def _snapshot(env):
    return (env.state, env.host(nat.F_TOUCH), env.host(nat.F_CONTACT_COUNT), env.host(nat.F_RGB), env.host(nat.F_DEPTH),
            env.host(nat.F_MASK), env.host(nat.F_JOINTS), env.host(nat.F_OBJ_POSE), env.host(nat.F_ERRFLAGS))


def _same(a, b):
    return all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))


def test_lookahead_is_bitwise_equivalent_to_inline_preparation(monkeypatch):
    """Same seeded run twice: with the look-ahead (k_prep_a / k_prep_b / k_collide of step t+1 launched behind the solve of
    step t, class by class, on side streams under the render) and with RR_NO_LOOKAHEAD=1 (every step prepares itself in line).
    Full-range commands (heavy and very heavy envs appear), per-env render flags, and everything that invalidates a
    look-ahead in between: reset masks, single and batched teleports, home-pose edits, set_state, a rejected command."""
    N = 384
    envs = [_make(monkeypatch, v, N, objects=3, width=128, height=128) for v in ({}, {'RR_NO_LOOKAHEAD': '1'})]
    rng = np.random.default_rng(11)
    for t in range(260):
        cmd = synthetic_actions(range(N), t, seed=3).astype(np.float32)
        ev = rng.random()
        mask = (rng.random(N) < 0.2).astype(np.uint8)
        pose = np.array([rng.uniform(-0.2, 0.0), rng.uniform(-0.3, 0.3), rng.uniform(0.3, 0.6), 0, 0, 0, 1], np.float32)
        i, o = int(rng.integers(0, N)), int(rng.integers(0, 3))
        mode = int(rng.integers(0, 3))
        flags = (rng.random(N) < 0.6).astype(np.uint8)
        for e in envs:
            if ev < 0.03:
                e.reset(mask)
            elif ev < 0.06:
                e.set_object_pose(i, o, pose)
            elif ev < 0.08:
                e.set_object_home(i, o, pose)
            elif ev < 0.10:
                e.state = e.state
            elif ev < 0.12:
                p = e.host(nat.F_OBJ_POSE)
                p[:, o, 2] += 0.05
                e.set_object_poses(p, mask)
            if t == 100:        # a rejected (non-finite) device-resident command: that env does not step, its contact history is dropped
                import torch
                dc = torch.from_numpy(cmd).cuda()
                dc[5, 2] = float('nan')
                e.step(device_ptr=dc.data_ptr(), render=True)
                torch.cuda.synchronize()
                assert e.host(nat.F_ERRFLAGS)[5] == 2 and len(e.contacts(5)) == 0
            else:
                e.step(cmd, render=[False, True, flags][mode])
        if t % 20 == 19 or t < 3 or t in (100, 101):
            a, b = _snapshot(envs[0]), _snapshot(envs[1])
            assert _same(a, b), t
            ca, cb = envs[0].contacts(i), envs[1].contacts(i)
            assert np.array_equal(ca, cb), t
    assert (envs[0].host(nat.F_ENV_CLASS) > 0).any()            # heavy envs took the side streams
    assert (envs[0].host(nat.F_ERRFLAGS) == 0).all()
    for e in envs:
        e.close()


def test_checkpoint_restore_continues_bit_for_bit():
    """save -> 60 steps == restore -> the same 60 steps, bitwise (state, contact lists with their forces, touch, images): the
    checkpoint carries the contact history of the warm start, which the 61-float state does not (rr_set_state starts cold).
    Restoring into a fresh env handle works too."""
    N = 96
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    cmds = [synthetic_actions(range(N), t, seed=5).astype(np.float32) for t in range(200)]
    for t in range(120):
        env.step(cmds[t], render=(t % 7 == 0))
    ck = env.checkpoint()
    st_at_save = env.state
    for t in range(120, 180):
        env.step(cmds[t], render=True)
    ref = _snapshot(env)
    ref_c = [env.contacts(i) for i in range(0, N, 7)]
    ts_ref = env.host(nat.F_TIMESTEP)
    env.restore(ck)
    assert np.array_equal(env.state, st_at_save, equal_nan=True) and (env.host(nat.F_TIMESTEP) == 120).all()
    for t in range(120, 180):
        env.step(cmds[t], render=True)
    assert _same(_snapshot(env), ref) and np.array_equal(env.host(nat.F_TIMESTEP), ts_ref)
    assert all(np.array_equal(a, env.contacts(i)) for a, i in zip(ref_c, range(0, N, 7)))
    # ... while a plain state restore (no contact history) does not reproduce the run bit for bit
    env.restore(ck)
    env.state = st_at_save
    for t in range(120, 180):
        env.step(cmds[t], render=True)
    assert not np.array_equal(env.state, ref[0])
    assert np.abs(env.state - ref[0])[:, :11].max() < 0.2
    fresh = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    fresh.restore(ck)
    for t in range(120, 180):
        fresh.step(cmds[t], render=True)
    assert _same(_snapshot(fresh), ref)
    with pytest.raises(nat.NativeError):
        fresh.restore(ck[:-4])
    fresh.close()
    env.close()


def test_bench_workload_at_size_against_the_oracle(monkeypatch):
    """BASELINE config 3 exactly as bench.py runs it -- 4096 envs, 3 objects, full-range resample-and-hold commands resident in
    HBM (bench.make_commands), 128x128 RGB + depth rendered every step, the three-stream heavy / light split and the look-ahead
    active -- for 420 steps.  Every 50 steps the 8 envs with the most contacts (the heavy and very heavy ones: arms pressed on
    the table, grippers in the objects) are stepped once by the float oracle from the device state and contact history:
    contact lists bit-identical, states within the force-scaled one-step bounds of tests/test_gpu_contacts_fuzz.py, image
    masks and depths exact.  The same run with RR_NO_SPLIT=1 RR_NO_LOOKAHEAD=1 (one launch per kernel, in line) is bitwise equal."""
    import importlib.util
    import torch
    from tests.test_gpu_contacts_fuzz import _lists_identical, state_bounds
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    N, T = bench.ENVS_PER_GPU, 420
    assert N == 4096 and bench.N_OBJECTS == 3 and (bench.W, bench.H) == (128, 128)
    cmds = bench.make_commands(torch, np, np.arange(N), T, 1.0, 'cuda:0')
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
    plain = _make(monkeypatch, {'RR_NO_SPLIT': '1', 'RR_NO_LOOKAHEAD': '1'}, N, objects=3, width=128, height=128, want_mask=False)
    o = Oracle(3, 128, 128, f32=True)
    checks, n_heavy_checked, n_light_checked, worst = 0, 0, 0, dict(dj=0.0, do=0.0, dv=0.0)
    for t in range(T):
        chk = t % 50 == 49
        if chk:
            torch.cuda.synchronize()
            st0 = env.state
            nc = env.host(nat.F_CONTACT_COUNT)
            # the 8 envs with the most contacts + 8 seeded-random ones (the light-env kernel, 98 % of the batch, meets the oracle
            # at the benchmark's size too -- not only through its bitwise equality with the generic kernel)
            heavy8 = np.argsort(-nc, kind='stable')[:8]
            rnd8 = np.random.default_rng(1000 + t).choice(np.setdiff1d(np.arange(N), heavy8), 8, replace=False)
            sel = np.concatenate([heavy8, rnd8])
            cache = {int(i): env.contacts(int(i)) for i in sel}
        env.step(device_ptr=cmds[t].data_ptr(), render=True)
        plain.step(device_ptr=cmds[t].data_ptr(), render=True)
        if not chk:
            continue
        st1 = env.state
        cls = env.host(nat.F_ENV_CLASS)
        rgb, dep = env.host(nat.F_RGB), env.host(nat.F_DEPTH)
        cmd_h = cmds[t].cpu().numpy()
        for i in sel:
            i = int(i)
            o.state = st0[i].astype(np.float64)
            o.set_contact_cache(cache[i])
            o.step(cmd_h[i].astype(np.float64))
            cd, co = env.contacts(i), o.contacts()
            assert _lists_identical(cd, co), (t, i, len(cd), len(co))
            fmax = float(cd[:, 10].max()) if len(cd) else 0.0
            dj = float(np.abs(st1[i][:22] - o.state[:22]).max())
            dobj = np.abs((st1[i][22:] - o.state[22:]).reshape(3, 13))
            bj, bo, bv = state_bounds(fmax)
            assert dj <= bj and dobj[:, :7].max() <= bo and dobj[:, 7:].max() <= bv, (t, i, fmax, dj, dobj[:, :7].max(), dobj[:, 7:].max())
            o.state = st1[i].astype(np.float64)
            r, d, m = o.render()
            diff = np.abs(r.astype(int) - rgb[i].astype(int)).max(-1)
            assert (diff > 1).sum() <= 2 and np.abs(d - dep[i]).max() <= 1e-5, (t, i, int((diff > 1).sum()))
            assert ((d < 1.0) == (dep[i] < 1.0)).all()
            checks += 1
            n_heavy_checked += int(cls[i] > 0)
            n_light_checked += int(cls[i] == 0)
        # the unsplit, in-line run is the same run
        assert np.array_equal(st1, plain.state, equal_nan=True), t
        assert np.array_equal(rgb, plain.host(nat.F_RGB)) and np.array_equal(dep, plain.host(nat.F_DEPTH)), t
    assert checks == 16 * (T // 50) and n_heavy_checked >= checks // 4 and n_light_checked >= checks // 4, (checks, n_heavy_checked, n_light_checked)
    assert (env.host(nat.F_ERRFLAGS) == 0).all()
    cls = env.host(nat.F_ENV_CLASS)
    assert (cls == 1).sum() > 20 and (cls == 2).sum() >= 1, ((cls == 1).sum(), (cls == 2).sum())
    env.close()
    plain.close()


def test_edited_eye_camera_reaches_the_retina():
    """get_retina is eyes["eye"].render(table position) in the reference (env.py:249-255): moving that camera changes the
    observation.  The facade pushes an edited eye into its backend (rr_set_camera); the frame equals the oracle's for the same
    view and differs from the default view."""
    import real_robots_amd as rr
    from real_robots_amd.mathutil import look_at, perspective
    e = rr.make('REALRobot2020-R1J2-v0', eye_width=128, eye_height=96)
    obs0 = e.reset()
    e.eyes["eye"].eyePosition = [0.3, -0.2, 1.0]
    act = {'joint_command': np.array([0.2, 0.4, 0, -0.6, 0, 0.3, 0, 0.1, 0.05]), 'render': True}
    for _ in range(3):
        obs, _, _, _ = e.step(act)
    assert (obs['retina'] != obs0['retina']).mean() > 0.05
    o = Oracle(2, 128, 96)
    o.state = e._backend().state[0].astype(np.float64)
    o.set_camera(look_at([0.3, -0.2, 1.0], [0, 0, 0.08], [0, 0, 1]), perspective(80, 128.0 / 96.0, 0.1, 100.0))
    r, d, m = o.render()
    assert (m == obs['mask']).all() and np.abs(r.astype(int) - obs['retina'].astype(int)).max() <= 1
    e.set_eye("eye")                                   # back to the default eye (env.py:136-141)
    obs2, _, _, _ = e.step(act)
    o2 = Oracle(2, 128, 96)
    o2.state = e._backend().state[0].astype(np.float64)
    r2, d2, m2 = o2.render()
    assert (m2 == obs2['mask']).all()
    e.close()


def test_object_wave_form_of_the_light_solve_is_bitwise_equivalent(monkeypatch):
    """k_solve_light_ow (workgroups of five waves for sixteen envs: four waves of 16-lane env groups for the command part, the row
    build and the robot's rows, the fifth wave one lane per (env, object) for the object chains, integration and the objects'
    render instances) against k_solve_light (RR_NO_OBJECT_WAVE=1: everything of an env on its 16-lane group): states, joints,
    touch sensors, object poses, contact forces, images bitwise equal over 200 steps of full-range commands with resets, a
    teleport and a rejected (non-finite) command in between, at an env count that is not a multiple of 16 and with 1..3 objects."""
    for n_obj, N in ((3, 130), (2, 49), (1, 16)):
        a = BatchedREALRobotEnv(N, objects=n_obj, width=64, height=64)
        b = _make(monkeypatch, {'RR_NO_OBJECT_WAVE': '1'}, N, objects=n_obj, width=64, height=64)
        rng = np.random.default_rng(7 + n_obj)
        ids = list(range(N))
        for t in range(200):
            cmd = synthetic_actions(ids, t, seed=5).astype(np.float32)
            if t == 60:
                m = (rng.random(N) < 0.3).astype(np.uint8)
                a.reset(m); b.reset(m)
            if t == 120:
                pose = np.array([-0.1, 0.2, 0.5, 0, 0, 0, 1], np.float32)
                a.set_object_pose(5, 0, pose); b.set_object_pose(5, 0, pose)
            if t == 90:         # a rejected (non-finite) device-resident command: env 3 does not step
                import torch
                dc = torch.from_numpy(cmd).cuda()
                dc[3, 2] = float('nan')
                a.step(device_ptr=dc.data_ptr(), render=True); b.step(device_ptr=dc.data_ptr(), render=True)
                torch.cuda.synchronize()
                assert a.host(nat.F_ERRFLAGS)[3] == 2
            else:
                a.step(cmd, render=True); b.step(cmd, render=True)
            if t % 10 == 9 or t in (90, 91):
                assert _same(_snapshot(a), _snapshot(b)), (n_obj, t)
                for i in (0, 5, N - 1):
                    assert np.array_equal(a.contacts(i), b.contacts(i)), (n_obj, t, i)
        assert (a.host(nat.F_ERRFLAGS)[3] & 2) == 0      # (the rejected command was one step only)
        a.close(); b.close()


def test_collide_launch_order_does_not_change_results(monkeypatch):
    """k_collide takes the envs by falling duration of their last collision pass (batches of more than 1 024 envs: the order is sorted
    by extra workgroups of the preparation launch, rr_collide.inc); RR_COLLIDE_ORDER=0 keeps env order.  The order only decides
    which workgroup handles which env and in which order the envs are appended to the heavy lists: states, contacts, classes and
    images are bitwise the same over 240 full-range steps with resets in between (with rendering; tests/test_gpu_round4.py has the
    same at 1 531 envs together with the pair cull, without images)."""
    N = 1160
    a = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    b = _make(monkeypatch, {'RR_COLLIDE_ORDER': '0'}, N, objects=3, width=64, height=64)
    rng = np.random.default_rng(21)
    ids = list(range(N))
    for t in range(240):
        cmd = synthetic_actions(ids, t, seed=9).astype(np.float32)
        if t in (80, 160):
            m = (rng.random(N) < 0.4).astype(np.uint8)
            a.reset(m); b.reset(m)
        a.step(cmd, render=True); b.step(cmd, render=True)
        if t % 20 == 19:
            assert _same(_snapshot(a), _snapshot(b)), t
            assert np.array_equal(a.host(nat.F_ENV_CLASS), b.host(nat.F_ENV_CLASS)), t
    assert (a.host(nat.F_ENV_CLASS) > 0).sum() >= 3      # (heavy envs exist: the first workgroups had list entries to take)
    a.close(); b.close()
