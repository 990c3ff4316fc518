"""GPU tests (-m gpu) of round 4, through the C ABI:
* every placement rr_step's host side picks from the lagged heavy counters (csrc/realrobot.hip, "Look-ahead" block of rr_step)
  is FORCED through the knobs the library reads at rr_create and compared bitwise with the unsplit, in-line step
  (RR_NO_SPLIT=1 RR_NO_LOOKAHEAD=1): the schedule may never change a result (env.py:326-356 is one sequential step);
* the macro workload and the late window of the headline workload exactly as bench.secondary_workloads runs them (4096 envs,
  render every step) are followed by the float oracle at size, the way tests/test_gpu_round3.py follows the headline window;
* RR_F_CONTACT_COUNT / RR_F_ENV_CLASS have stable storage: a pointer obtained once stays valid over steps (realrobot.h).
"""
import os

import numpy as np
import pytest

from oracle.oracle import Oracle
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions

pytestmark = pytest.mark.gpu

PLAIN = {'RR_NO_SPLIT': '1', 'RR_NO_LOOKAHEAD': '1'}


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
            env.host(nat.F_MASK), env.host(nat.F_JOINTS), env.host(nat.F_OBJ_POSE), env.host(nat.F_ERRFLAGS),
            env.host(nat.F_ENV_CLASS))


def _first_difference(a, b):
    names = ('state', 'touch', 'contact_count', 'rgb', 'depth', 'mask', 'joints', 'obj_pose', 'errflags', 'env_class')
    for n, x, y in zip(names, a, b):
        if not np.array_equal(x, y, equal_nan=True):
            bad = np.argwhere(np.asarray(x != y).reshape(len(x), -1).any(1)).ravel()
            return '%s differs in envs %s' % (n, bad[:8].tolist())
    return None


# Every placement of realrobot.hip's rr_step, by the knob that forces it.  RR_FORCE_HCOUNT pins what the host-side decisions
# read instead of the lagged counters ("heavy,very heavy"): launch shapes (coop form <= 256, list-walking render <= 768 items,
# the look-ahead's placement) and placements then differ from what the device-side lists actually hold -- which is exactly the
# situation of a lagged counter, and must not matter.
PLACEMENTS = [
    ('1: default (look-ahead behind the very heavy envs\' solve, their render behind the heavy envs\')', {}),
    ('1, empty lists assumed: coop solves, list-walking renders', {'RR_FORCE_HCOUNT': '0,0'}),
    ('1 with a long heavy list assumed: packed heavy solve, three-kernel render of the list, very heavy render at the main stream\'s tail',
     {'RR_FORCE_HCOUNT': '2000,10'}),
    ('2: many very heavy envs assumed: packed solves, kinematics + collide on the heavy stream, dynamics on the very heavy one',
     {'RR_FORCE_HCOUNT': '2000,300'}),
    ('3: split off (mostly heavy): one k_solve for all, look-ahead beside the render', {'RR_SPLIT_MAX_PCT': '0', 'RR_FORCE_HCOUNT': '1,0'}),
    ('1 <-> 3: the split switches off mid-run (more than 2 % heavy envs)', {'RR_SPLIT_MAX_PCT': '2'}),
    ('1 with separate k_render_setup launches (no set-up in the solve kernels) and k_collide in env order', {'RR_NO_FUSED_SETUP': '1', 'RR_COLLIDE_ORDER': '0'}),
    ("1' with the streams' events recorded by markers instead of completing with their launches", {'RR_FORCE_HCOUNT': '40,3', 'RR_NO_EXT_EVENTS': '1'}),
    ('5: look-ahead without the split', {'RR_NO_SPLIT': '1'}),
    ('5: split without the look-ahead', {'RR_NO_LOOKAHEAD': '1'}),
]


def test_every_schedule_placement_is_bitwise_the_inline_step(monkeypatch):
    """448 envs, 250 full-range steps (heavy and very heavy envs appear: arms pressed on the table), per-env render flags,
    resets, teleports, home-pose edits and state restores in between; states, touch, contact counts and lists with forces,
    images, classes and error flags of every forced placement are bitwise those of the unsplit in-line step."""
    N, T = 448, 250
    names = list(PLACEMENTS)
    envs = [_make(monkeypatch, PLAIN, N, objects=3, width=128, height=128)]
    envs += [_make(monkeypatch, v, N, objects=3, width=128, height=128) for _, v in names]
    rng = np.random.default_rng(41)
    for t in range(T):
        cmd = synthetic_actions(range(N), t, seed=3).astype(np.float32)
        ev = rng.random()
        mask = (rng.random(N) < 0.15).astype(np.uint8)
        pose = np.array([rng.uniform(-0.2, 0.0), rng.uniform(-0.3, 0.3), rng.uniform(0.3, 0.6), 0, 0, 0, 1], np.float32)
        i, o = int(rng.integers(0, N)), int(rng.integers(0, 3))
        mode = int(rng.integers(0, 4))
        flags = (rng.random(N) < 0.6).astype(np.uint8)
        for e in envs:
            if ev < 0.02:
                e.reset(mask)
            elif ev < 0.04:
                e.set_object_pose(i, o, pose)
            elif ev < 0.05:
                e.set_object_home(i, o, pose)
            elif ev < 0.06:
                e.state = e.state
            e.step(cmd, render=[False, True, True, flags][mode])
        if t % 25 == 24 or t < 2:
            ref = _snapshot(envs[0])
            cref = [envs[0].contacts(k) for k in range(0, N, 31)]
            for (name, _), e in zip(names, envs[1:]):
                d = _first_difference(ref, _snapshot(e))
                assert d is None, (name, t, d)
                assert all(np.array_equal(c, e.contacts(k)) for c, k in zip(cref, range(0, N, 31))), (name, t)
    cls = envs[0].host(nat.F_ENV_CLASS)
    assert (cls == 1).sum() >= 3 and (cls == 2).sum() >= 1, ((cls == 1).sum(), (cls == 2).sum())     # both side streams had work
    assert (envs[0].host(nat.F_ERRFLAGS) == 0).all()
    for e in envs:
        e.close()


def test_collision_pass_order_and_pair_cull_change_no_result(monkeypatch):
    """k_collide dispatches its workgroups by falling duration of the env's last pass (batches of more than 1 024 envs) and drops
    shape pairs in its broad phase by the grown-radius rule (rr_collide.inc; tests/test_pair_cull.py states the rule against the
    oracle): 1 531 envs -- not a multiple of eight: the order's padding entries --, 160 full-range steps with resets and teleports;
    states, touch, contact counts, lists with forces, classes and error flags are bitwise those of the env-order pass without the cull."""
    N, T = 1531, 160
    a = _make(monkeypatch, {}, N, objects=3, width=32, height=32)
    b = _make(monkeypatch, {'RR_COLLIDE_ORDER': '0', 'RR_NO_PAIR_CULL': '1'}, N, objects=3, width=32, height=32)
    rng = np.random.default_rng(5)
    for t in range(T):
        cmd = synthetic_actions(range(N), t, seed=9).astype(np.float32)
        ev = rng.random()
        mask = (rng.random(N) < 0.1).astype(np.uint8)
        pose = np.array([rng.uniform(-0.2, 0.0), rng.uniform(-0.3, 0.3), rng.uniform(0.3, 0.6), 0, 0, 0, 1], np.float32)
        i, o = int(rng.integers(0, N)), int(rng.integers(0, 3))
        for e in (a, b):
            if ev < 0.03:
                e.reset(mask)
            elif ev < 0.06:
                e.set_object_pose(i, o, pose)
            e.step(cmd, render=False)
        if t % 20 == 19 or t < 2:
            for f in (nat.F_TOUCH, nat.F_CONTACT_COUNT, nat.F_JOINTS, nat.F_OBJ_POSE, nat.F_ERRFLAGS, nat.F_ENV_CLASS):
                assert np.array_equal(a.host(f), b.host(f)), (t, f)
            assert np.array_equal(a.state, b.state), t
            for k in range(0, N, 61):
                assert np.array_equal(a.contacts(k), b.contacts(k)), (t, k)
    assert a.host(nat.F_CONTACT_COUNT).max() >= 20        # arms pressed on the table: the pass had real work
    a.close(); b.close()


def test_steps_without_camera_side_by_side_are_bitwise_the_one_launch_step(monkeypatch):
    """Placement 3b (rr_host.inc step_single): a batch of more than 1 024 envs stepping WITHOUT camera solves its classes side by side on
    the three streams once the very heavy list is long (macro actions without the retina) -- forced here by the reading, on a population
    that has both heavy classes; mixed with rendered steps (the split placements and their look-ahead in between).  Bitwise the default run."""
    N, T = 1100, 170
    a = _make(monkeypatch, {}, N, objects=3, width=32, height=32)
    b = _make(monkeypatch, {'RR_FORCE_HCOUNT': '400,100'}, N, objects=3, width=32, height=32)
    c = _make(monkeypatch, {'RR_FORCE_HCOUNT': '30,600'}, N, objects=3, width=32, height=32)      # (a very heavy list read longer than the one-env-per-wave form's cap)
    rng = np.random.default_rng(8)
    for t in range(T):
        cmd = synthetic_actions(range(N), t, seed=12).astype(np.float32)
        ev = rng.random()
        mask = (rng.random(N) < 0.1).astype(np.uint8)
        render = bool(t % 7 == 3)
        for e in (a, b, c):
            if ev < 0.03:
                e.reset(mask)
            e.step(cmd, render=render)
        if t % 20 == 19 or t < 2:
            for e in (b, c):
                for f in (nat.F_TOUCH, nat.F_CONTACT_COUNT, nat.F_JOINTS, nat.F_OBJ_POSE, nat.F_ERRFLAGS, nat.F_ENV_CLASS, nat.F_RGB, nat.F_DEPTH):
                    assert np.array_equal(a.host(f), e.host(f)), (t, f)
                assert np.array_equal(a.state, e.state), t
                for k in range(0, N, 53):
                    assert np.array_equal(a.contacts(k), e.contacts(k)), (t, k)
    cls = a.host(nat.F_ENV_CLASS)
    assert (cls == 1).sum() >= 3 and (cls == 2).sum() >= 1, ((cls == 1).sum(), (cls == 2).sum())
    for e in (a, b, c):
        e.close()


def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _oracle_check(env, o, sel, st0, cache, st1, cmd_of, rgb, dep, t):
    """One oracle step for each env of `sel` from the device state st0 + contact history: lists bit-identical, states within the
    force-scaled one-step bounds, image mask exact, depth 1e-5, RGB <= 1 except <= 2 texel-boundary pixels."""
    from tests.test_gpu_contacts_fuzz import _lists_identical, state_bounds
    worst = 0.0
    for i in sel:
        i = int(i)
        o.state = st0[i].astype(np.float64)
        o.set_contact_cache(cache[i])
        o.step(cmd_of(i).astype(np.float64))
        cd, co = env.contacts(i), o.contacts()
        assert _lists_identical(cd, co), (t, i, len(cd), len(co))
        fmax = float(cd[:, 10].max()) if len(cd) else 0.0
        dj = float(np.abs(st1[i][:22] - o.state[:22]).max())
        dobj = np.abs((st1[i][22:] - o.state[22:]).reshape(3, 13))
        bj, bo, bv = state_bounds(fmax)
        assert dj <= bj and dobj[:, :7].max() <= bo and dobj[:, 7:].max() <= bv, (t, i, fmax, dj, dobj[:, :7].max(), dobj[:, 7:].max())
        worst = max(worst, dj / bj, dobj[:, :7].max() / bo, dobj[:, 7:].max() / bv)
        o.state = st1[i].astype(np.float64)
        r, d, m = o.render()
        diff = np.abs(r.astype(int) - rgb[i].astype(int)).max(-1)
        assert (diff > 1).sum() <= 2 and np.abs(d - dep[i]).max() <= 1e-5, (t, i, int((diff > 1).sum()))
        assert ((d < 1.0) == (dep[i] < 1.0)).all()
    return worst


def _sample16(nc, t):
    """The envs an at-size check point hands to the oracle: the 8 with the most contacts (heavy / very heavy: the generic kernel)
    and 8 seeded-random others (mostly light: `k_solve_light_ow`, 98 % of the batch, meets the oracle at 4096 envs directly)."""
    heavy8 = np.argsort(-nc, kind='stable')[:8]
    rnd8 = np.random.default_rng(1000 + t).choice(np.setdiff1d(np.arange(len(nc)), heavy8), 8, replace=False)
    return np.concatenate([heavy8, rnd8])


def test_macro_workload_at_size_against_the_oracle(monkeypatch):
    """BASELINE config 5's shape exactly as bench.secondary_workloads runs it: 4096 envs, macro actions drawn with seed 0 from
    macro_space (env.py:49-52) and planned on the device, one plan row per step (env.py:388-412), render every step, steps
    0..340 -- by then 1 500-2 000 envs are heavy and hundreds very heavy, the split switches itself off and on around 60 % and
    the look-ahead moves between the streams.  Every 50 steps the 8 envs with the most contacts and 8 seeded-random ones are stepped by the float oracle
    from the device state and contact history with the plan row the device consumed; every 50 steps the whole batch is
    compared bitwise with the unsplit in-line run."""
    import torch
    bench = _bench_module()
    N, T = bench.ENVS_PER_GPU, 340
    assert N == 4096
    macro = np.random.default_rng(0).uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2))
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
    plain = _make(monkeypatch, PLAIN, N, objects=3, width=128, height=128, want_mask=False)
    env.plan_macro(macro)
    plain.plan_macro(macro)
    o = Oracle(3, 128, 128, f32=True)
    checks, heavy_seen, vheavy_seen = 0, 0, 0
    for t in range(T):
        chk = t % 50 == 49
        if chk:
            torch.cuda.synchronize()
            st0 = env.state
            nc = env.host(nat.F_CONTACT_COUNT)
            sel = _sample16(nc, t)
            cache = {int(i): env.contacts(int(i)) for i in sel}
            rows = {int(i): env.get_plan(int(i))[t] for i in sel}
        env.step_plan(render=True)
        plain.step_plan(render=True)
        if not chk:
            continue
        st1 = env.state
        cls = env.host(nat.F_ENV_CLASS)
        rgb, dep = env.host(nat.F_RGB), env.host(nat.F_DEPTH)
        _oracle_check(env, o, sel, st0, cache, st1, lambda i: rows[i], rgb, dep, t)
        checks += len(sel)
        heavy_seen, vheavy_seen = max(heavy_seen, int((cls == 1).sum())), max(vheavy_seen, int((cls == 2).sum()))
        assert np.array_equal(st1, plain.state, equal_nan=True), t
        assert np.array_equal(rgb, plain.host(nat.F_RGB)) and np.array_equal(dep, plain.host(nat.F_DEPTH)), t
        assert np.array_equal(env.host(nat.F_CONTACT_COUNT), plain.host(nat.F_CONTACT_COUNT)), t
    assert checks == 16 * (T // 50)
    assert heavy_seen > 500 and vheavy_seen > 64, (heavy_seen, vheavy_seen)      # the macro placements were in play
    assert (env.host(nat.F_ERRFLAGS) == 0).all()
    env.close()
    plain.close()


def test_late_window_of_the_headline_workload_against_the_oracle(monkeypatch):
    """The headline workload (bench.make_commands, 4096 envs, render every step) in its LATE window, as bench.py's first secondary
    entry times it: the first 2000 steps run without camera (the state does not depend on it; the images persist from frame to
    frame, so a first rendered frame is a full frame), then steps 2000..2100 with a render every step -- ~650 heavy envs, three
    render launches for their list, the very heavy envs' render at the tail of the main stream.  Oracle steps for the 8 envs
    with the most contacts and 8 seeded-random ones every 25 steps, whole batch bitwise against the unsplit in-line run at the same points."""
    import torch
    bench = _bench_module()
    N, T0, T = bench.ENVS_PER_GPU, 2000, 100
    cmds = bench.make_commands(torch, np, np.arange(N), T0 + T, 1.0, 'cuda:0')
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
    plain = _make(monkeypatch, PLAIN, N, objects=3, width=128, height=128, want_mask=False)
    for t in range(T0):
        env.step(device_ptr=cmds[t].data_ptr(), render=False)
        plain.step(device_ptr=cmds[t].data_ptr(), render=False)
    torch.cuda.synchronize()
    assert np.array_equal(env.state, plain.state, equal_nan=True)
    o = Oracle(3, 128, 128, f32=True)
    checks, heavy_seen = 0, 0
    for t in range(T0, T0 + T):
        chk = t % 25 == 24
        if chk:
            torch.cuda.synchronize()
            st0 = env.state
            nc = env.host(nat.F_CONTACT_COUNT)
            sel = _sample16(nc, t)
            cache = {int(i): env.contacts(int(i)) for i in sel}
        env.step(device_ptr=cmds[t].data_ptr(), render=True)
        plain.step(device_ptr=cmds[t].data_ptr(), render=True)
        if not chk:
            continue
        st1 = env.state
        cls = env.host(nat.F_ENV_CLASS)
        rgb, dep = env.host(nat.F_RGB), env.host(nat.F_DEPTH)
        cmd_h = cmds[t].cpu().numpy()
        _oracle_check(env, o, sel, st0, cache, st1, lambda i: cmd_h[i], rgb, dep, t)
        checks += len(sel)
        heavy_seen = max(heavy_seen, int((cls == 1).sum()))
        assert np.array_equal(st1, plain.state, equal_nan=True), t
        assert np.array_equal(rgb, plain.host(nat.F_RGB)) and np.array_equal(dep, plain.host(nat.F_DEPTH)), t
    assert checks == 16 * (T // 25)
    assert heavy_seen * 4 > 768, heavy_seen             # the long-list placement (h_long) was in play
    assert (env.host(nat.F_ERRFLAGS) == 0).all()
    env.close()
    plain.close()


def test_contact_count_and_env_class_pointers_are_stable_over_steps():
    """realrobot.h: a pointer from rr_get_buffer stays valid until rr_destroy.  RR_F_CONTACT_COUNT and RR_F_ENV_CLASS used to
    point into the double-buffered contact frame, which changes roles every step: a DLPack view taken once
    must show, after every step, what rr_copy_to_host returns."""
    import torch
    N = 256
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    cc = torch.from_dlpack(env.device_buffer(nat.F_CONTACT_COUNT))
    cl = torch.from_dlpack(env.device_buffer(nat.F_ENV_CLASS))
    seen_heavy = False
    for t in range(200):
        env.step(synthetic_actions(range(N), t, seed=3).astype(np.float32), render=(t % 3 == 0))
        env.sync()
        a, b = cc.cpu().numpy().reshape(-1), env.host(nat.F_CONTACT_COUNT)
        assert np.array_equal(a, b), t
        c, d = cl.cpu().numpy().reshape(-1), env.host(nat.F_ENV_CLASS)
        assert np.array_equal(c, d), t
        assert all(len(env.contacts(i)) == b[i] for i in (0, 17, N - 1)), t
        seen_heavy = seen_heavy or bool((d > 0).any())
    assert seen_heavy
    env.close()


def test_host_mirrors_follow_every_step():
    """rr_map_observations / rr_map_images / rr_sync_observations (the single-env facade's read-back, env.py:336-339, 536-567): after
    every step -- with and without camera, after resets and teleports -- the mapped host blocks hold what rr_copy_to_host returns, for
    one env (class-by-class chain on the main stream) and for a small batch."""
    for N in (1, 5):
        env = BatchedREALRobotEnv(N, objects=3, width=96, height=64)
        m = env.map_observations()
        rgb, dep, msk = env.map_images(mask=True)
        rng = np.random.default_rng(N)
        for t in range(120):
            if t == 40:
                env.reset()
            if t == 70:
                env.set_object_pose(0, 1, np.array([-0.1, 0.1, 0.5, 0, 0, 0, 1], np.float32))
            cam = bool(rng.random() < 0.4)
            env.step(synthetic_actions(range(N), t, seed=3).astype(np.float32), render=cam)
            env.sync_observations()
            j, tc, op, ts = m['joints'].copy(), m['touch'].copy(), m['obj_pose'].copy(), m['timestep'].copy()
            img = (rgb.copy(), dep.copy(), msk.copy()) if cam else None
            assert np.array_equal(j, env.host(nat.F_JOINTS)) and np.array_equal(tc, env.host(nat.F_TOUCH)), (N, t)
            assert np.array_equal(op, env.host(nat.F_OBJ_POSE)) and np.array_equal(ts, env.host(nat.F_TIMESTEP)), (N, t)
            assert np.array_equal(m['errflags'], env.host(nat.F_ERRFLAGS))
            if cam:
                assert np.array_equal(img[0], env.host(nat.F_RGB)) and np.array_equal(img[1], env.host(nat.F_DEPTH)), (N, t)
                assert np.array_equal(img[2], env.host(nat.F_MASK)), (N, t)
        env.close()


def test_deselected_image_mirror_is_not_refreshed_and_comes_back_up_to_date():
    """rr_select_image_mirror (once the mask was mapped every rendered step paid its copy): a deselected block keeps
    its last contents over rendered steps while the selected ones follow; selected again it holds the device buffer at once; and the
    gym facade deselects the mask on a plain observation and gets a correct one on the next extended observation."""
    env = BatchedREALRobotEnv(2, objects=3, width=96, height=64)
    rgb, dep, msk = env.map_images(mask=True)
    act = lambda t: synthetic_actions(range(2), t, seed=5).astype(np.float32)
    for t in range(30):
        env.step(act(t), render=True)
    env.sync_observations()
    assert np.array_equal(msk, env.host(nat.F_MASK))
    old_mask = msk.copy()
    env.select_image_mirror(mask=False)
    for t in range(30, 110):
        env.step(act(t), render=True)
    env.sync_observations()
    assert np.array_equal(rgb, env.host(nat.F_RGB)) and np.array_equal(dep, env.host(nat.F_DEPTH))
    assert np.array_equal(msk, old_mask) and not np.array_equal(old_mask, env.host(nat.F_MASK))      # the arm has moved; the block has not
    env.select_image_mirror()
    env.sync_observations()
    assert np.array_equal(msk, env.host(nat.F_MASK))
    with pytest.raises(Exception):
        nat.check(env.L.rr_select_image_mirror(env.h, 8))
    env.close()
    import real_robots_amd as rr
    e = rr.make('REALRobot2020-R2J3-v0', eye_width=64, eye_height=64)
    e.reset()
    a = {'joint_command': np.array([0.3, 0.5, 0, -1.0, 0, 0.5, 0, 0.5, 0.5]), 'render': True}
    for _ in range(20):
        obs, _, _, _ = e.step(a)
    ext = e.get_observation_extended(_rendered=True)
    for _ in range(20):
        obs, _, _, _ = e.step(a)                       # plain observations: the mask block is deselected
    be = e._backend()
    assert be._img_sel == 3                            # (the selection is tracked by the backend: RGB + depth)
    # any OTHER holder of the views asks map_images(mask=True) and gets a block that is refreshed again -- no stale
    # mask without an error -- and brought up to date at once
    views = be.map_images(mask=True)
    assert be._img_sel == 7
    be.sync_observations()
    assert np.array_equal(views[2], be.host(nat.F_MASK))
    for _ in range(5):
        obs, _, _, _ = e.step(a)
    assert be._img_sel == 3
    ext2 = e.get_observation_extended(_rendered=True)
    assert be._img_sel == 7
    assert np.array_equal(ext2['mask'], be.host(nat.F_MASK)[0]) and np.array_equal(ext2['retina'], be.host(nat.F_RGB)[0])
    assert not np.array_equal(ext2['mask'], ext['mask'])
    e.close()


def test_single_env_chain_is_bitwise_the_batched_step():
    """One env (the gym facade's backend) is stepped class by class on the main stream with the look-ahead as two plain launches
    (rr_step's small-N path); as a member of a batch of 80 the same env takes the generic kernels and the side streams.  The batch
    runs first (300 full-range steps, a camera frame every third step); an env that was light at some check points and heavy at
    others is then replayed alone with its command stream: bitwise the same trajectory, contacts, touch sensors and frames."""
    N, T = 80, 300
    b = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    cmds = [synthetic_actions(range(N), t, seed=6).astype(np.float32) for t in range(T)]
    rec, cls_seen = {}, np.zeros((N, 3), bool)
    for t in range(T):
        b.step(cmds[t], render=(t % 3 == 0))
        if t % 10 == 9:
            rec[t] = (b.state, b.host(nat.F_TOUCH), b.host(nat.F_RGB), b.host(nat.F_DEPTH), [b.contacts(i) for i in range(N)])
            c = b.host(nat.F_ENV_CLASS)
            cls_seen[np.arange(N), c] = True
    cand = np.flatnonzero(cls_seen[:, 0] & (cls_seen[:, 1] | cls_seen[:, 2]))
    assert len(cand) > 0, cls_seen.sum(0)
    b.close()
    for k in cand[:2]:
        a = BatchedREALRobotEnv(1, objects=3, width=64, height=64)
        for t in range(T):
            a.step(cmds[t][k:k + 1], render=(t % 3 == 0))
            if t % 10 == 9:
                st, tc, rgb, dep, con = rec[t]
                assert np.array_equal(a.state[0], st[k], equal_nan=True), (k, t)
                assert np.array_equal(a.host(nat.F_TOUCH)[0], tc[k]) and np.array_equal(a.contacts(0), con[k]), (k, t)
                assert np.array_equal(a.host(nat.F_RGB)[0], rgb[k]) and np.array_equal(a.host(nat.F_DEPTH)[0], dep[k]), (k, t)
        a.close()


def test_checkpoint_of_other_step_parameters_is_rejected(monkeypatch):
    """A checkpoint continues bit for bit only in a handle that steps the same way: the header carries dt / ERP / margin / sweeps /
    warm-start factor / object-lane capacity / inertia source / edge contacts, and a blob taken with other values is refused
    (round 3 accepted it and diverged silently).  After a restore the published contact count follows the restored list and the
    class diagnostic starts clean."""
    a = BatchedREALRobotEnv(32, objects=3, width=64, height=64)
    for t in range(80):
        a.step(synthetic_actions(range(32), t, seed=3).astype(np.float32), render=False)
    ck = a.checkpoint()
    nc = a.host(nat.F_CONTACT_COUNT)
    for kw, env_vars in (({'solver_iters': 20}, {}), ({'dt': 0.004}, {}), ({}, {'RR_NO_WARMSTART': '1'}), ({}, {'RR_SOLVER_POOL': '600'})):
        b = _make(monkeypatch, env_vars, 32, objects=3, width=64, height=64, **kw)
        with pytest.raises(nat.NativeError, match='other step parameters'):
            b.restore(ck)
        b.close()
    c = BatchedREALRobotEnv(32, objects=3, width=64, height=64)
    c.restore(ck)
    assert np.array_equal(c.host(nat.F_CONTACT_COUNT), nc) and (c.host(nat.F_ENV_CLASS) == 0).all()
    a.step(synthetic_actions(range(32), 80, seed=3).astype(np.float32), render=True)
    c.step(synthetic_actions(range(32), 80, seed=3).astype(np.float32), render=True)
    assert np.array_equal(a.state, c.state, equal_nan=True) and np.array_equal(a.host(nat.F_RGB), c.host(nat.F_RGB))
    a.close()
    c.close()


def test_device_microbench_reports_plausible_ceilings():
    """rr_device_microbench (bench.py's roofline): HBM copy / triad bandwidth and the VALU issue rate of a sample-test-like mix,
    measured on the device.  Plausibility only: between a tenth of and the full spec figure (8 TB/s; 256 CU x 4 SIMD x 2.4 GHz / 2
    cycles = 1 229 G wave-instr/s is the most any reading of the hardware allows)."""
    r = nat.device_microbench(0)
    assert 800.0 < r['hbm_copy_GBs'] < 8000.0 and 800.0 < r['hbm_triad_GBs'] < 8000.0, r
    assert 300.0 < r['valu_mix_G_wave_instr_s'] < 1229.0, r


def test_one_wave_per_env_solve_of_small_batches_is_bitwise_the_packed_form(monkeypatch):
    """A step that solves all envs in one launch (no camera; config 2) gives every env a wave of its own up to 1024 envs -- its four
    16-lane groups build the rows side by side (the coop form, until now only for the heavy lists) -- instead of four envs to a
    wave (RR_COOP_ALL=0): bitwise the same states, contacts, touch sensors and classes over 260 full-range steps with resets, at a
    batch size that is not a multiple of four; config 2 2.39 -> 2.61 M env-steps/s."""
    N = 203
    a = BatchedREALRobotEnv(N, objects=1, width=64, height=64)
    b = _make(monkeypatch, {'RR_COOP_ALL': '0'}, N, objects=1, width=64, height=64)
    rng = np.random.default_rng(5)
    for t in range(260):
        cmd = synthetic_actions(range(N), t, seed=8).astype(np.float32)
        if t in (90, 180):
            m = (rng.random(N) < 0.3).astype(np.uint8)
            a.reset(m); b.reset(m)
        a.step(cmd, render=False); b.step(cmd, render=False)
        if t % 20 == 19:
            assert np.array_equal(a.state, b.state, equal_nan=True), t
            assert np.array_equal(a.host(nat.F_TOUCH), b.host(nat.F_TOUCH)) and np.array_equal(a.host(nat.F_CONTACT_COUNT), b.host(nat.F_CONTACT_COUNT)), t
            assert np.array_equal(a.host(nat.F_ENV_CLASS), b.host(nat.F_ENV_CLASS)), t
            assert all(np.array_equal(a.contacts(i), b.contacts(i)) for i in range(0, N, 17)), t
    assert (a.host(nat.F_ENV_CLASS) > 0).any() and (a.host(nat.F_ERRFLAGS) == 0).all()
    a.close(); b.close()


def test_raster_tile_shape_does_not_change_an_image(monkeypatch):
    """Images wider than 128 columns are rasterised in 64 x 64 tiles (a cluster met three of the 12-row strips a full-width tile is
    at 320 x 240), narrower ones in full-width strips; RR_TILE_W forces a shape.  The tile shape decides which workgroup owns a
    pixel, never its value: RGB, depth and mask bitwise equal for squares, strips and two odd shapes at 320 x 240 (a partial last
    column of tiles at 48 and 100) and at 128 x 128, over 40 steps with per-env render flags (incremental image update)."""
    for (w, h, shapes) in ((320, 240, ('320', '48', '100')), (128, 128, ('64', '32'))):
        ref = BatchedREALRobotEnv(24, objects=3, width=w, height=h)
        others = [_make(monkeypatch, {'RR_TILE_W': tw}, 24, objects=3, width=w, height=h) for tw in shapes]
        rng = np.random.default_rng(w)
        for t in range(40):
            cmd = synthetic_actions(range(24), t, seed=2).astype(np.float32)
            flags = (rng.random(24) < 0.7).astype(np.uint8)
            for e in [ref] + others:
                e.step(cmd, render=(flags if t % 3 else True))
            if t % 5 == 4:
                a = (ref.host(nat.F_RGB), ref.host(nat.F_DEPTH), ref.host(nat.F_MASK))
                for tw, e in zip(shapes, others):
                    b = (e.host(nat.F_RGB), e.host(nat.F_DEPTH), e.host(nat.F_MASK))
                    assert all(np.array_equal(x, y) for x, y in zip(a, b)), (w, tw, t)
        assert (ref.host(nat.F_ERRFLAGS) == 0).all()
        for e in [ref] + others:
            e.close()


def test_raster_dispatch_order_does_not_change_an_image(monkeypatch):
    """From 2048 (env, tile) items on, k_raster's workgroups are dispatched in the order of the previous frame's costs (the extra
    workgroups of k_shade sort the durations k_raster measured; env << 8 | tile per XCD class, ~0 padding when the batch is not a
    multiple of eight).  The order decides where and when a tile is rasterised, never what is in it: states, contact lists, RGB,
    depth, mask and error flags bitwise equal to the env-major grid (RR_NO_RASTER_ORDER=1) over 60 steps with resets, per-env
    render flags and a render-less step in between -- at 515 envs (a partial last group of eight) and 128 x 128, and at 128 envs
    with the 20 tiles of 320 x 240."""
    for (n, w, h, steps) in ((515, 128, 128, 60), (128, 320, 240, 25)):
        ref = _make(monkeypatch, {'RR_NO_RASTER_ORDER': '1'}, n, objects=3, width=w, height=h)
        env = BatchedREALRobotEnv(n, objects=3, width=w, height=h)
        rng = np.random.default_rng(n)
        for t in range(steps):
            cmd = synthetic_actions(range(n), t, seed=3).astype(np.float32)
            flags = (rng.random(n) < 0.8).astype(np.uint8)
            if t % 17 == 16:
                mask = (rng.random(n) < 0.1).astype(np.uint8)
                for e in (ref, env):
                    e.reset(mask)
            for e in (ref, env):
                e.step(cmd, render=(False if t % 11 == 10 else (flags if t % 3 else True)))
            if t % 6 == 5 or t == steps - 1:
                for x, y in zip(_snapshot(ref), _snapshot(env)):
                    assert np.array_equal(x, y), (n, w, t)
        assert (env.host(nat.F_ERRFLAGS) == 0).all()
        ref.close(); env.close()


def test_a_camera_inside_the_arm_mesh_does_not_stop_the_physics():
    """a render-only condition must never touch simulation-visible state.  With the eye INSIDE an arm link hundreds
    of triangles cross the near plane in every tile (the clip queue of a raster tile holds 2 048; its overflow is error flag 8, a
    render status).  Whatever the renderer reports, the physics keeps stepping: no flag the solver's dead mask knows (1, 4) appears,
    the clocks advance, and the states are bitwise those of a run that never renders."""
    from real_robots_amd.mathutil import look_at, perspective
    N, T = 32, 60
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
    ref = BatchedREALRobotEnv(N, objects=3, width=128, height=128)
    # inside link 1 / link 2 of the arm (robot base at [-0.55, 0, -0.04]), looking along the arm
    env.set_camera(look_at(np.array([-0.55, 0.0, 0.30]), np.array([-0.55, 0.05, 1.2]), np.array([1.0, 0.0, 0.0])), perspective(120.0, 1.0, 0.1, 100.0))
    for t in range(T):
        cmd = synthetic_actions(range(N), t, seed=5).astype(np.float32)
        env.step(cmd, render=True)
        ref.step(cmd, render=False)
    ef = env.host(nat.F_ERRFLAGS)
    assert (ef & ~np.uint32(8) == 0).all(), np.unique(ef)            # nothing but (possibly) the render status
    assert (env.host(nat.F_TIMESTEP) == T).all()
    assert np.array_equal(env.state, ref.state, equal_nan=True)
    assert (env.host(nat.F_MASK) == 0).any()                         # the arm is in front of the lens
    env.close()
    ref.close()
