"""GPU parity tests (-m gpu) that pin WHICH contacts exist and what they carry, through the C ABI:

* a seeded randomised differential run (batch sizes 1..130, 1-3 objects, five resolutions, full / shrunken / empty solver
  row pool, random joint commands or macro plans, per-env render flags, resets and teleports): the envs with the most
  contacts are checked one step at a time against the float build of the oracle started from the device state.  The
  collision pipeline (forward kinematics, shape transforms, sphere cull, vertex-in-polytope tests, manifold reduction)
  runs without FMA contraction and with a shared explicit sin/cos on both sides, so the contact LIST -- bodies, points,
  normals, distances, friction -- must be bit-identical; the solver (fp32, contracted) is held to tolerances;
* numeric parity of Kuka.get_touch_sensors / get_contacts normal forces (robot.py:131-163) in a grasp and in a pushing
  sweep.
"""
import os

import numpy as np
import pytest

from oracle.oracle import Oracle
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions

pytestmark = pytest.mark.gpu

CRUSH_FORCE = 2000.0      # N; above this a link is crushed onto the table / an object under the 100 kN position motors: 50
                          # Gauss-Seidel sweeps are far from converged there, the impulses are 1e3 times those of resting contact and
                          # the float rounding of device and oracle (same rows, same order, different association) grows with them


def state_bounds(fmax):
    """One-step bounds device vs float oracle from the same state and contact history, as a function of the largest normal
    force of the step: (joints q / qd [rad, rad/s], object pose, object velocity).  Up to CRUSH_FORCE the bounds are the flat
    ones measured over 1200 seeded cases (about twice the worst seen: 1.6e-4 / 4.6e-7 / 1.8e-4); above it they scale with the
    force -- a rounding error of relative size 1e-7 in an impulse of f dt acts on the same inverse inertias."""
    k = max(1.0, fmax / (2 * CRUSH_FORCE))      # (measured over 300 cases, 343 checks above 2 kN: at most 0.3 of these bounds)
    return 3e-4 * k, 1e-6 * k, 4e-4 * k


SENS_RUNS = 8             # oracle re-runs from a state moved by one unit in the last place, per check above the flat bounds
SENS_FACTOR = 4.0         # the device may differ from the oracle by this many times the oracle's own spread over those re-runs


def oracle_sensitivity(o, st0, cache, cmd, ref, nobj, rng):
    """How far the float oracle's OWN one-step result moves when every entry of the start state is moved by one unit in the
    last place (SENS_RUNS random sign patterns): the conditioning of this step's contact problem under 50 projected
    Gauss-Seidel sweeps, measured rather than assumed (a clamp that flips in one sweep moves the result by far more than
    the rounding that flipped it).  Returns the largest deviation from the unperturbed result `ref` in the three groups of
    state_bounds (joints, object pose, object velocity)."""
    sj = so = sv = 0.0
    st0 = st0.astype(np.float32)
    for _ in range(SENS_RUNS):
        up = rng.random(st0.shape) < 0.5
        stp = np.where(up, np.nextafter(st0, np.float32(np.inf)), np.nextafter(st0, np.float32(-np.inf)))
        o.state = stp.astype(np.float64)
        o.set_contact_cache(cache)
        o.step(cmd.astype(np.float64))
        d = np.abs(o.state - ref)
        dobj = d[22:22 + 13 * nobj].reshape(nobj, 13)
        sj, so, sv = max(sj, float(d[:22].max())), max(so, float(dobj[:, :7].max())), max(sv, float(dobj[:, 7:].max()))
    return sj, so, sv


def solver_independent_checks(o, st1, cd, co):
    """What holds for ANY correct solver run on the step's contact problem, whatever its rounding: evaluated on the float
    oracle's rows (o has just stepped) for the device's solution (post-step state st1, normal forces cd[:, 10]) and for the
    oracle's own.  Returns (ok, details): same active set apart from contacts whose force is below 2 % of the largest one on
    either side, force sums within 5 %, and the device's complementarity residual (what one more Gauss-Seidel update of the
    normal rows would still change) not above 1.05 x the oracle's + 0.1 % of the force sum."""
    if not len(cd):
        return True, {}
    f_dev, f_orc = cd[:, 10].astype(np.float64), co[:, 10].astype(np.float64)
    fm = max(f_dev.max(), f_orc.max())
    r_dev = o.solution_residual(st1, f_dev, 0.02 * fm)
    r_orc = o.solution_residual(o.state, f_orc, 0.02 * fm)
    differ = r_dev['active'] ^ r_orc['active']
    # a contact that is active on one side only must be a marginal one there too
    marginal = all(max(f_dev[c], f_orc[c]) < 0.04 * fm for c in range(len(cd)) if (differ >> c) & 1)
    sums = abs(r_dev['f_sum'] - r_orc['f_sum']) <= 0.05 * r_orc['f_sum'] + 1.0
    resid = r_dev['res_sum'] <= 1.05 * r_orc['res_sum'] + 1e-3 * r_orc['f_sum'] + 0.05
    d = dict(fmax=fm, active_xor=bin(differ).count('1'), f_sum=(r_dev['f_sum'], r_orc['f_sum']), res=(r_dev['res_sum'], r_orc['res_sum']))
    return bool(marginal and sums and resid), d


def _lists_identical(c_dev, c_orc):
    """Contact records {bodyA, bodyB, linkA, x, n, dist, force, mu}: everything but the force (column 10) bit for bit."""
    if len(c_dev) != len(c_orc):
        return False
    if not len(c_dev):
        return True
    keep = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11]
    return bool((c_dev[:, keep] == c_orc[:, keep].astype(np.float32)).all())


def _fuzz_case(case, seed0, stats, bad, solver=None):
    """solver: the product's `solver=` dict (None: documented defaults); the oracle gets the same constants."""
    from oracle.oracle import params_from_solver
    rng = np.random.default_rng(seed0 * 1000 + case)
    N = int(rng.choice([1, 3, 5, 17, 34, 63, 130]))
    nobj = int(rng.integers(1, 4))
    W, H = [(64, 48), (64, 64), (128, 128), (160, 120), (320, 240)][int(rng.integers(0, 5))]
    pool = rng.choice([None, None, "0", "900", "2500"])
    if pool:
        os.environ['RR_SOLVER_POOL'] = str(pool)
    try:
        env = BatchedREALRobotEnv(N, objects=nobj, width=W, height=H, solver=solver)
    finally:
        os.environ.pop('RR_SOLVER_POOL', None)
    o = Oracle(nobj, W, H, f32=True, **params_from_solver(solver))
    macro = rng.random() < 0.6
    plans = None
    if macro:
        env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
        plans = [env.get_plan(i) for i in range(N)]
    scale = rng.choice([0.5, 0.8, 1.0])
    T = int(rng.integers(60, 200))
    t_off = int(rng.integers(0, 400)) if macro else 0
    if macro and t_off:
        plans = [np.roll(p, -t_off, axis=0) for p in plans]
    for t in range(T):
        if macro:
            cmd = np.stack([plans[i][t % 1000] for i in range(N)]).astype(np.float32)
        else:
            cmd = (synthetic_actions(range(N), t, seed=case) * scale).astype(np.float32)
        if rng.random() < 0.01:
            env.reset((rng.random(N) < 0.3).astype(np.uint8))
        if rng.random() < 0.01:
            env.set_object_pose(int(rng.integers(0, N)), int(rng.integers(0, nobj)),
                                np.array([rng.uniform(-0.2, 0.0), rng.uniform(-0.3, 0.3), rng.uniform(0.3, 0.6), 0, 0, 0, 1], np.float32))
        flags = (rng.random(N) < 0.5).astype(np.uint8)
        chk = t % 20 == 19
        if chk:
            st0 = env.state
            ncs = np.array([len(env.contacts(i)) for i in range(N)])
            sel = sorted(set(list(np.argsort(-ncs)[:2]) + [int(rng.integers(0, N))]))
            flags[sel] = 1
            cache = {int(i): env.contacts(int(i)) for i in sel}      # contact history (warm start) of the envs that are checked
        env.step(cmd, render=flags if N > 1 else bool(flags[0]))
        if not chk:
            continue
        st1 = env.state
        rgb, dep, msk = env.host(nat.F_RGB), env.host(nat.F_DEPTH), env.host(nat.F_MASK)
        for i in sel:
            o.state = st0[i].astype(np.float64)
            o.set_contact_cache(cache[int(i)])
            o.step(cmd[i].astype(np.float64))
            cd, co = env.contacts(i), o.contacts()
            stats['checks'] += 1
            stats['contacts'] += len(cd)
            tag = (case, N, nobj, W, H, pool, bool(macro), t, int(i))
            if not _lists_identical(cd, co):
                bad.append(tag + ('contact list', len(cd), len(co)))
                if os.environ.get('RR_FUZZ_VERBOSE'):
                    np.set_printoptions(precision=9, linewidth=250, suppress=True)
                    print(tag, "device contacts\n", cd, "\noracle contacts\n", co.astype(np.float32))
                continue
            fmax = float(cd[:, 10].max()) if len(cd) else 0.0
            dj = float(np.abs(st1[i][:22] - o.state[:22]).max())
            dobj = np.abs((st1[i][22:22 + 13 * nobj] - o.state[22:22 + 13 * nobj]).reshape(nobj, 13))
            do, dv = float(dobj[:, :7].max()), float(dobj[:, 7:].max())      # pose; linear and angular velocity
            bj, bo, bv = state_bounds(fmax)
            crush = fmax > CRUSH_FORCE
            stats['crush'] += int(crush)
            key = 'c_' if crush else ''
            stats[key + 'dj'] = max(stats[key + 'dj'], dj / bj)
            stats[key + 'do'] = max(stats[key + 'do'], do / bo)
            stats[key + 'dv'] = max(stats[key + 'dv'], dv / bv)
            # every check is held to a stated bound: the flat one up to CRUSH_FORCE, the force-scaled one above it.  A step
            # that is over its bound has to be explained by the conditioning of ITS contact problem, measured on the oracle
            # itself: within SENS_FACTOR x the oracle's spread under one-ulp changes of the start state (a clamp of the 50
            # Gauss-Seidel sweeps that flips moves the result by far more than the rounding that flipped it) -- and the
            # solver-independent properties below hold for it like for every other check.  Such steps are counted and capped.
            over = dj > bj or do > bo or dv > bv
            ref_state = o.state.copy()
            ok, det = solver_independent_checks(o, st1[i], cd, co)
            if not np.isfinite(st1[i]).all():
                bad.append(tag + ('state', dj, do, dv, fmax))
            elif over:
                sj, so_, sv = oracle_sensitivity(o, st0[i], cache[int(i)], cmd[i], ref_state, nobj, np.random.default_rng(case * 1000 + t))
                stats['ill'] += 1
                stats['ill_share'] = max(stats['ill_share'], dj / max(SENS_FACTOR * sj, bj), do / max(SENS_FACTOR * so_, bo), dv / max(SENS_FACTOR * sv, bv))
                if dj > max(SENS_FACTOR * sj, bj) or do > max(SENS_FACTOR * so_, bo) or dv > max(SENS_FACTOR * sv, bv):
                    bad.append(tag + ('state (ill-conditioned step)', dj, do, dv, fmax, sj, so_, sv))
            if not ok:
                bad.append(tag + ('solver-independent', det))
            o.state = st1[i].astype(np.float64)
            r, d, m = o.render()
            diff = np.abs(r.astype(int) - rgb[i].astype(int)).max(-1)
            # coverage and depth are contraction-free on both sides (exact); the shading is not: a nearest-texel lookup at a
            # texel boundary may flip for a pixel or two
            if (m != msk[i]).any() or (diff > 1).sum() > 2 or np.abs(d - dep[i]).max() > 1e-5:
                bad.append(tag + ('image', int((m != msk[i]).sum()), int((diff > 1).sum()), float(np.abs(d - dep[i]).max())))
    if (env.host(nat.F_ERRFLAGS) != 0).any() or (env.host(nat.F_TIMESTEP) > T).any():
        bad.append((case, 'errflags/timestep'))
    env.close()


def test_seeded_differential_run_contact_lists_bit_identical():
    """>= 300 seeded cases; zero disagreements in the contact lists (in particular no candidate that sits at the 2 cm margin
    on one side only); EVERY check's state within the stated one-step bounds (state_bounds: flat up to 2 kN of normal force,
    scaled with the force above; a step over its bound -- at most 0.2 % of the checks -- is held to SENS_FACTOR x the
    oracle's own measured spread under one-ulp changes of the start state, oracle_sensitivity) and the
    solver-independent properties of the device's solution (active set, force sum, complementarity residual) as good as
    the oracle's; image masks and depths exact, RGB within one grey level except at most two texel-boundary pixels per frame."""
    stats = dict(checks=0, contacts=0, crush=0, dj=0.0, do=0.0, dv=0.0, c_dj=0.0, c_do=0.0, c_dv=0.0, ill=0, ill_share=0.0)
    bad = []
    n_cases = int(os.environ.get('RR_FUZZ_CASES', '300'))
    only = os.environ.get('RR_FUZZ_ONLY')
    for case in ([int(only)] if only else range(n_cases)):
        _fuzz_case(case, 2, stats, bad)
    print("fuzz: %d cases, %d one-step checks, %d contacts compared, %d of them above %.0f N; worst share of the bound used -- joints %.2f object pose %.2f "
          "object velocity %.2f (above: %.2f %.2f %.2f); %d steps over their bound and held to %.0f x the oracle's own "
          "one-ulp spread instead, worst share %.2f; %d violations"
          % (n_cases, stats['checks'], stats['contacts'], stats['crush'], CRUSH_FORCE, stats['dj'], stats['do'], stats['dv'],
             stats['c_dj'], stats['c_do'], stats['c_dv'], stats['ill'], SENS_FACTOR, stats['ill_share'], len(bad)))
    for b in bad[:20]:
        print("   violation:", b)
    assert not bad, bad[:20]
    if not only:
        assert stats['checks'] >= 3 * n_cases and stats['contacts'] > 20 * n_cases
        assert stats['crush'] <= 0.15 * stats['checks']      # full-range commands press links into the table now and then
        assert stats['ill'] <= 0.002 * stats['checks'] + 1    # the measured-conditioning bound is the exception (<= 10 of ~4600), not a second regime


def _grasp_script():
    from oracle.kinematics import inverse_kinematics, quat_from_euler
    orient = quat_from_euler(0, 3.14, -1.57)
    q_hi = inverse_kinematics(np.zeros(11), [-0.1, 0.0, 0.55], orient)
    q_lo = inverse_kinematics(q_hi, [-0.1, 0.0, 0.47], orient)
    cmds = []
    for q, g, n in [(q_hi, [0.5, 0.0], 150), (q_lo, [0.5, 0.0], 120), (q_lo, [0.0, 0.0], 60)]:
        cmds += [np.concatenate([q[:7], g])] * n
    return np.array(cmds, np.float32)


def _check_forces(env, o, i, st0, cache, cmd, where, stats):
    """One step of env i from the device state st0: contact list identical, normal forces and the four touch sensors
    (max normal force per skin link, robot.py:152-163) within 0.1 % (+ 0.02 N) of the float oracle."""
    o.state = st0[i].astype(np.float64)
    o.set_contact_cache(cache)                  # the device's contact list before the step: the history of the warm start
    o.step(cmd.astype(np.float64))
    cd, co = env.contacts(i), o.contacts()
    assert _lists_identical(cd, co), where
    if len(cd):
        f_dev, f_orc = cd[:, 10].astype(np.float64), co[:, 10]
        # 0.1 % of the largest force (+ 0.02 N) up to CRUSH_FORCE; above it the relative bound grows with the force like the state
        # bounds do (the warm-started squeeze of the gripper reaches 2.8 kN on a skin) -- no check is waived
        rel = 1e-3 * max(1.0, f_orc.max() / CRUSH_FORCE)
        err = float(np.abs(f_dev - f_orc).max())
        stats['worst_force_share'] = max(stats.get('worst_force_share', 0.0), err / (rel * f_orc.max() + 0.02))
        assert err <= rel * f_orc.max() + 0.02, (where, err, float(f_orc.max()))
        stats['forces'] += int((f_orc > 1.0).sum())
    touch = env.host(nat.F_TOUCH)[i].astype(np.float64)
    t_orc = o.obs()[1]
    rel = 1e-3 * max(1.0, t_orc.max() / CRUSH_FORCE)
    assert np.abs(touch - t_orc).max() <= rel * max(t_orc.max(), 1.0) + 0.02, (where, touch, t_orc)
    stats['touch'] += int((t_orc > 1.0).sum())


def test_touch_sensors_and_normal_forces_match_the_oracle():
    """a7: get_touch_sensors / get_contacts (robot.py:131-163) compared numerically, one step at a time from the device
    state, (i) while the fingers close on the cube and (ii) while a macro action pushes the objects over the table."""
    stats = dict(forces=0, touch=0)
    cmds = _grasp_script()
    env = BatchedREALRobotEnv(4, objects=1, width=64, height=64)
    o = Oracle(1, 64, 64, f32=True)
    for _ in range(100):
        env.step(None)
    for t, c in enumerate(cmds):
        chk = t >= 262 and t % 2 == 0
        if chk:
            st0, cache = env.state, env.contacts(0)
        env.step(np.tile(c, (4, 1)))
        if chk:
            _check_forces(env, o, 0, st0, cache, c, ('grasp', t), stats)
    assert stats['touch'] >= 20, stats                # the distal skins pressed on the cube in the steps that were checked
    env.close()
    N = 34
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    o = Oracle(3, 64, 64, f32=True)
    rng = np.random.default_rng(5)
    env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
    plans = [env.get_plan(i) for i in range(N)]
    before = dict(stats)
    for t in range(760):
        chk = t >= 200 and t % 20 == 0
        if chk:
            ncs = np.array([len(env.contacts(i)) for i in range(N)])
            sel = np.argsort(-ncs)[:4]
            st0 = env.state
            cache = {int(i): env.contacts(int(i)) for i in sel}
        env.step_plan(render=False)
        if chk:
            for i in sel:
                _check_forces(env, o, int(i), st0, cache[int(i)], plans[i][t], ('push', t, int(i)), stats)
    assert stats['forces'] - before['forces'] > 200, stats
    print("normal forces / touch sensors vs the float oracle: %d loaded contacts compared, worst share of the bound %.2f" % (stats['forces'], stats['worst_force_share']))
    env.close()


def test_edge_edge_contacts_match_the_oracle():
    """Crossing edges (a tilted cube edge over the shelf's front edge, tests/test_oracle_pins.py) at several gaps, small overlaps
    and tilts: the device's contact lists -- vertex candidates followed by the edge-edge candidate -- are those of the float
    oracle bit for bit, and so is the list without the edge pass (RR_NO_EDGE_CONTACTS)."""
    from tests.test_oracle_pins import _edge_crossing_pose
    cases = [(0.002, -40.0), (0.0005, -55.0), (-0.001, -40.0), (0.012, -30.0), (-0.0003, -50.0), (0.019, -45.0)]
    N = len(cases)
    for edges in (1, 0):
        if not edges:
            os.environ['RR_NO_EDGE_CONTACTS'] = '1'
        try:
            env = BatchedREALRobotEnv(N, objects=1, width=64, height=64)
        finally:
            os.environ.pop('RR_NO_EDGE_CONTACTS', None)
        o = Oracle(1, 64, 64, f32=True, edge_contacts=edges)
        env.reset()
        for i, (gap, tilt) in enumerate(cases):
            env.set_object_pose(i, 0, _edge_crossing_pose(gap, tilt)[0].astype(np.float32))
        st0 = env.state
        env.step(None, render=False)
        n_edge = 0
        for i, (gap, tilt) in enumerate(cases):
            o.state = st0[i].astype(np.float64)
            o.step(None)
            cd, co = env.contacts(i), o.contacts()
            assert _lists_identical(cd, co), (edges, i, cd, co)
            # an edge-edge contact: its normal is the common normal of the two edges (neither a shelf nor a cube facet normal)
            n_exp = _edge_crossing_pose(gap, tilt)[1]
            n_edge += int(any(np.abs(c[6:9] - n_exp).max() < 1e-3 and abs(c[9] - gap) < 3e-4 for c in cd))
        # (an edge pair that overlaps by more than 0.5 mm beyond the deepest vertex candidate is no contact: the -1 mm case)
        assert n_edge == (N - 1 if edges else 0), (edges, n_edge)
        env.close()


def test_warm_start_follows_the_oracle_over_a_trajectory():
    """Device and oracle run side by side from reset, each with its own contact history (no hand-over): two solver sweeps
    per step only, so that the inherited impulses carry the resting objects -- without the warm start they sink by a
    millimetre (tests/test_oracle_pins.py); matching decisions that differed between the two sides would show as a
    different penetration or a different set of active contacts."""
    env = BatchedREALRobotEnv(3, objects=3, width=64, height=64, solver_iters=2)
    o = Oracle(3, 64, 64, f32=True, solver_iters=2)
    env.reset()
    o.reset()
    for t in range(250):
        env.step(None, render=False)
        o.step(None)
        if t % 50 == 49:
            cd, co = env.contacts(1), o.contacts()
            assert len(cd) == len(co) == 12, (t, len(cd), len(co))
            assert np.abs(cd[:, 9] - co[:, 9]).max() < 2e-5, (t, np.abs(cd[:, 9] - co[:, 9]).max())          # penetrations
            assert np.abs(cd[:, 10] - co[:, 10]).max() < 0.02 * co[:, 10].max() + 0.02, t                     # normal forces
    d = env.state[1].astype(np.float64) - o.state
    dp = np.abs(d[22:].reshape(3, 13)[:, :3])
    assert dp[:, 2].max() < 2e-5 and dp[:, :2].max() < 3e-4          # heights; the slow lateral creep at two sweeps per step
    c = env.contacts(0)
    assert c[:, 9].min() > -5e-4            # (a cold start at two sweeps per step: below -6e-4)
    env.close()
