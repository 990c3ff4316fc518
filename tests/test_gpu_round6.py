"""GPU tests (-m gpu) of round 6: the preparation on sixteen lanes per env (k_prep16) against the thread-per-env kernels, the
video maker through evaluate(), and the delta image records of the renderer."""
import os

import numpy as np
import pytest

from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions

pytestmark = pytest.mark.gpu

S_BR, S_BP, S_BAX, S_MINV, S_QDS, S_OR, S_OIINV, S_OVS, S_OWS, S_OP, S_TOTAL = 0, 99, 132, 165, 286, 297, 324, 351, 360, 369, 378


def _make(monkeypatch, envvars, *a, **k):
    for key, v in envvars.items():
        monkeypatch.setenv(key, v)
    try:
        return BatchedREALRobotEnv(*a, **k)
    finally:
        for key in envvars:
            monkeypatch.delenv(key)


def _rich_states(N, seed):
    """States off a run: arms anywhere in their range and moving, objects pushed around / toppled / in flight."""
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    for t in range(260):
        env.step(synthetic_actions(range(N), t, seed=seed).astype(np.float32))
    st = env.state
    env.close()
    rng = np.random.default_rng(seed)
    st[:, 11:22] += rng.normal(0, 2.0, size=(N, 11)).astype(np.float32)          # joint velocities up to several rad/s
    return st


def test_stream_order_and_run_ahead_bound_change_no_result(monkeypatch):
    """RR_SKIP_QUEUES only decides which hardware queues the side streams get (worth 50 % of the step time in the wrong
    process): states, contacts and frames of a run with it are those of a run without, bit for bit."""
    N, T = 96, 120
    cmds = [synthetic_actions(range(N), t, seed=11).astype(np.float32) for t in range(T)]
    runs = []
    # (... and the bound on the host's run-ahead, RR_RUN_AHEAD, only decides how old the list lengths are that pick a placement)
    for setting in (None, {'RR_SKIP_QUEUES': '1,1'}, {'RR_SKIP_QUEUES': '2,0'}, {'RR_RUN_AHEAD': '0'}, {'RR_RUN_AHEAD': '2'}):
        env = _make(monkeypatch, setting or {}, N, objects=3, width=64, height=64)
        for t in range(T):
            env.step(cmds[t], render=True)
        runs.append((env.state, env.host(nat.F_RGB), env.host(nat.F_DEPTH), env.host(nat.F_TOUCH), env.host(nat.F_ENV_CLASS)))
        env.close()
    assert (runs[0][4] > 0).any()                     # (heavy envs -- the side streams -- were in play)
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("N", [4096, 5])
def test_prep16_matches_the_thread_per_env_preparation(monkeypatch, N):
    """k_prep_b16 (sixteen lanes per env) against k_prep_b (a thread per env) on the same states, through the diagnostic field
    RR_F_PREP, with the look-ahead off (the step then prepares itself in line: k_prep_a + k_prep_b / k_prep_b16 on the state that
    was set): M^-1 and the unconstrained joint velocities agree to rounding (a different association of the compiler's fused
    multiply-adds: 1e-4 of the row's diagonal -- float32 Cholesky of a mass matrix with condition ~1e4 -- / 2e-4 rad/s at velocities up
    to 40 rad/s); frames and object terms -- k_prep_a in
    both -- are bit for bit the same.  Then the look-ahead form: k_prep_ab16 alone produces, bit for bit, what k_prep_a + k_prep_b16
    produce (frames through its own contraction-free chain on the row lanes, object terms on the object lanes)."""
    st = _rich_states(N, 3)
    recs = {}
    for name, envv in (('scalar', {'RR_PREP_SCALAR': '1', 'RR_NO_LOOKAHEAD': '1'}), ('p16', {'RR_NO_LOOKAHEAD': '1'})):
        env = _make(monkeypatch, envv, N, objects=3, width=64, height=64)
        env.state = st
        env.step(None)
        recs[name] = env.host(nat.F_PREP)
        env.close()
    a, b = recs['scalar'], recs['p16']
    assert a.shape == (N, nat.PREP_FLOATS)
    assert np.array_equal(a[:, :S_MINV], b[:, :S_MINV]) and np.array_equal(a[:, S_OR:], b[:, S_OR:])       # k_prep_a's part
    Ma, Mb = a[:, S_MINV:S_QDS].reshape(N, 11, 11).astype(np.float64), b[:, S_MINV:S_QDS].reshape(N, 11, 11).astype(np.float64)
    diag = np.sqrt(np.einsum('nii,njj->nij', Ma, Ma))
    rel = np.abs(Ma - Mb) / diag
    dq = np.abs(a[:, S_QDS:S_OR] - b[:, S_QDS:S_OR])
    print("N=%d: M^-1 worst |diff| / sqrt(M^-1_ii M^-1_jj) %.2e; qd* worst |diff| %.2e rad/s (|qd*| up to %.1f)" % (N, rel.max(), dq.max(), np.abs(a[:, S_QDS:S_OR]).max()))
    assert rel.max() < 1e-4 and dq.max() < 2e-4
    assert np.abs(Ma - np.swapaxes(Ma, 1, 2)).max() / np.abs(Ma).max() < 1e-5                           # (and it is an inverse mass matrix: symmetric)
    # the look-ahead form: the record a step leaves is the one an in-line preparation of the SAME state computes
    la = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    la.state = st
    la.step(None)                         # k_prep_a + k_prep_b16 in line (the state was set from outside), solve, then k_prep_ab16 ahead
    rec_la, st1 = la.host(nat.F_PREP), la.state
    la.close()
    il = _make(monkeypatch, {'RR_NO_LOOKAHEAD': '1'}, N, objects=3, width=64, height=64)
    il.state = st1
    il.step(None)
    rec_il = il.host(nat.F_PREP)
    il.close()
    parts = (('frames', 0, S_MINV), ('M^-1', S_MINV, S_QDS), ('qd*', S_QDS, S_OR), ('object terms', S_OR, S_TOTAL))
    differ = [nm for nm, lo, hi in parts if not np.array_equal(rec_la[:, lo:hi], rec_il[:, lo:hi])]
    assert not differ, (differ, [float(np.abs(rec_la[:, lo:hi].astype(np.float64) - rec_il[:, lo:hi]).max()) for nm, lo, hi in parts])


def test_runs_under_both_preparations_stay_together(monkeypatch):
    """Free-running: 64 envs x 400 full-range steps under the two preparations.  Until an env's first contact involving the robot the
    two runs differ by rounding only (1e-4 rad); afterwards they are two samples of the same chaotic system (like device vs float64
    oracle, tests/test_gpu_trajectory.py) -- what must hold is that neither flags an error and most envs stay within 1e-2 rad."""
    N, T = 64, 400
    a = _make(monkeypatch, {'RR_PREP_SCALAR': '1'}, N, objects=3, width=64, height=64)
    b = BatchedREALRobotEnv(N, objects=3, width=64, height=64)
    free = np.ones(N, bool)
    worst_free = 0.0
    for t in range(T):
        cmd = synthetic_actions(range(N), t, seed=21).astype(np.float32)
        a.step(cmd)
        b.step(cmd)
        if t % 20 == 19:
            sa, sb = a.state, b.state
            for i in np.where(free)[0]:
                c = a.contacts(int(i))
                if len(c) and ((c[:, 0] >= 0) & (c[:, 0] < 16)).any():
                    free[i] = False
            d = np.abs(sa[:, :11] - sb[:, :11]).max(1)
            worst_free = max(worst_free, float(d[free].max()) if free.any() else 0.0)
    print("two preparations, %d steps: worst joint difference before an env's first robot contact %.2e rad; %d envs never touched; "
          "%d of %d envs within 1e-2 rad at the end" % (T, worst_free, int(free.sum()), int((d < 1e-2).sum()), N))
    assert worst_free < 1e-4
    assert (d < 1e-2).sum() >= N // 2
    assert (a.host(nat.F_ERRFLAGS) == 0).all() and (b.host(nat.F_ERRFLAGS) == 0).all()
    a.close()
    b.close()


def test_evaluate_writes_videos_with_goal_and_start_insets(tmp_path, monkeypatch):
    """f3 (videomaker.py:94-129) end to end: evaluate(video=(intrinsic, extrinsic, debug)) films the debug camera of the reference's
    VideoMaker through the HIP rasteriser; a trial's frames carry the goal image top right and the start image top left."""
    import real_robots_amd as rr
    from real_robots_amd import videomaker as vm
    from real_robots_amd.generate_goals import generate_goals, save_goals
    monkeypatch.chdir(tmp_path)
    goals = generate_goals(2, 0, 0, n_obj=1, seed=5, batch=8, width=64, height=64)
    save_goals(str(tmp_path / 'goals.npy'), goals)

    class Still(rr.BasePolicy):
        def step(self, observation, reward, done):
            return {'joint_command': np.zeros(9), 'render': True}
    svc_video = (range(0, 17), {0}, False)
    from real_robots_amd.evaluate import EvaluationService
    svc = EvaluationService(Still, 'R1', 'joints', 1, 16, 16, 1, False, str(tmp_path / 'goals.npy.npz'), svc_video,
                            env_kwargs={'eye_width': 64, 'eye_height': 64})
    svc.run_intrinsic_phase()
    svc.run_extrinsic_phase()
    files = svc.videomaker.files
    assert len(files) == 2 and all(os.path.exists(f) for f in files)
    raw = open(files[1], 'rb').read()
    fb = 320 * 240 * 3
    n = (len(raw) - raw.index(b'movi') - 4) // (fb + 8)
    assert n == 2                                                   # steps 8 and 16
    frame = np.frombuffer(raw[-fb:], np.uint8).reshape(240, 320, 3)[::-1, :, ::-1]
    goal_inset = vm.make_inset(goals[0].retina, "GOAL")
    assert np.array_equal(frame[:80, 320 - 106:], goal_inset)
    assert frame[120:, :].std() > 5                                 # the debug camera sees the scene (not a blank frame)


def test_delta_image_records_from_the_renderers_lists_single_rank_self_gather():
    """The delta gather fed by the renderer (rr_pack_image_delta: records of the pixels in the fragment lists, prefix-packed by one
    cumsum over RR_F_FRAG_COUNT) and applied by rr_apply_image_delta, as a world-1 RCCL self-gather on one GPU: after every step the
    persistent copy equals the device images bit for bit -- full-range commands, objects pushed around, a masked reset and a
    teleport on the way; exact mode (payload sized from this step's totals) and sync-free mode (from the previous step's: no
    host read; a frame that outgrows its cap is flagged one step late and repaired)."""
    import torch
    import torch.distributed as dist
    from real_robots_amd.distributed import DeltaImageGather
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29000 + os.getpid() % 2000))
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        N, H, W = 96, 128, 128
        for sync_free in (False, True):
            env = BatchedREALRobotEnv(N, objects=3, width=W, height=H, want_mask=False)
            rgb = torch.as_tensor(env.device_buffer(nat.F_RGB), device='cuda:0')
            dep = torch.as_tensor(env.device_buffer(nat.F_DEPTH), device='cuda:0')
            dg = DeltaImageGather(env=env, sync_free=sync_free)
            rng = np.random.default_rng(5)
            n_delta, worst_share, inexact = 0, 0.0, []
            for t in range(120):
                if t == 60:
                    env.reset((rng.random(N) < 0.3).astype(np.uint8))
                if t == 80:
                    env.set_object_pose(3, 1, np.array([-0.1, 0.1, 0.5, 0, 0, 0, 1], np.float32))
                env.step(synthetic_actions(range(N), t, seed=8).astype(np.float32), render=True)
                slabs_before = dg.slab_steps
                g_rgb, g_dep = dg.step(rgb, dep)
                torch.cuda.synchronize()
                same = bool(torch.equal(g_rgb, rgb)) and bool(torch.equal(g_dep.view(torch.int32), dep.view(torch.int32)))
                if not same:
                    inexact.append(t)
                if dg.slab_steps == slabs_before:
                    n_delta += 1
                    worst_share = max(worst_share, dg.bytes_last / (N * H * W * 7))
            if not sync_free:
                assert not inexact, inexact
                assert n_delta >= 115 and worst_share < 0.25, (n_delta, worst_share)        # the records are a fraction of the slabs
            else:
                # (every inexact frame was flagged in the following step and repaired there)
                assert len(inexact) <= 3 and n_delta >= 100, (inexact, n_delta)
            print("delta gather, sync_free=%s: %d of 120 steps shipped records (worst %.3f of the slab bytes), %d slab steps, inexact frames %s"
                  % (sync_free, n_delta, worst_share, dg.slab_steps, inexact))
            env.close()
    finally:
        dist.destroy_process_group()


_SCHEDULE_SCRIPT = r"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, %(root)r)
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from oracle.kinematics import inverse_kinematics, quat_from_euler
N = 4096
press = inverse_kinematics(np.zeros(11), [-0.15, 0.25, 0.40], quat_from_euler(0, 3.14, -1.57))
press = np.concatenate([press[:7], [0.0, 0.0]]).astype(np.float32)
crush = inverse_kinematics(np.zeros(11), [-0.15, 0.25, 0.33], quat_from_euler(0, 3.14, -1.57))
crush = np.concatenate([crush[:7], [0.0, 0.0]]).astype(np.float32)
workloads = {'no heavy env': np.zeros((N, 9), np.float32), '30 %% heavy': np.zeros((N, 9), np.float32), '10 %% very heavy': np.zeros((N, 9), np.float32)}
workloads['30 %% heavy'][np.arange(N) %% 10 < 3] = press
workloads['10 %% very heavy'][np.arange(N) %% 10 == 0] = crush

def ms_per_step(force, cmd_dev):
    if force is None:
        os.environ.pop('RR_FORCE_HCOUNT', None)
    else:
        os.environ['RR_FORCE_HCOUNT'] = '%%d,%%d' %% force
    env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
    for _ in range(150):
        env.step(device_ptr=cmd_dev.data_ptr(), render=True)
    best = 1e9
    for _ in range(3):
        env.sync()
        t0 = time.perf_counter()
        for _ in range(120):
            env.step(device_ptr=cmd_dev.data_ptr(), render=True)
        env.sync()
        best = min(best, (time.perf_counter() - t0) / 120 * 1e3)
    cls = env.host(nat.F_ENV_CLASS)
    env.close()
    return best, int((cls == 1).sum()), int((cls == 2).sum())

out = {}
for name, cmd in workloads.items():
    cmd_dev = torch.from_numpy(cmd).cuda()
    auto, nh, nvh = ms_per_step(None, cmd_dev)
    auto = min(auto, ms_per_step(None, cmd_dev)[0])
    forced = {}
    for h in (0, 150, 600, 1300, 2600):
        for vh in (0, 30, 200, 1200):
            if vh <= max(h, 1) * 4:
                forced['%%d,%%d' %% (h, vh)] = ms_per_step((h, vh), cmd_dev)[0]
    out[name] = dict(auto=auto, heavy=nh, very_heavy=nvh, forced=forced)
print('RESULT ' + json.dumps(out))
"""


def test_schedule_choices_off_the_bench_workload():
    """The placement of a step (rr_step: which stream solves / renders which class, where the look-ahead goes, list walker or grid
    for the heavy envs' visibility pass) is chosen from lagged host copies of the two heavy-list lengths through constants that were
    tuned on the benchmark's workload (LA_VH_MAX, RENDER_LIST_WGS, split_max_pct, the walker / grid threshold).  Two workloads the
    benchmark never shows -- NO heavy env at all (arms at home, objects at rest), 30 % of the envs pressing the gripper on the
    table (heavy from the first contact on) and 10 % crushing it onto the table (very heavy) -- are timed under the automatic choice and under every forced reading of the counts
    (RR_FORCE_HCOUNT: placements only, results are bitwise the same, tests/test_gpu_round4.py): the automatic choice is within 3 % of
    the best forced one (4 % for the very heavy workload: measured 2.3-2.7 %).  (In a process of its own: the suite's other tests leave streams, RCCL threads and a warm allocator behind
    that add a millisecond-scale jitter to 0.4 ms steps.  The first run of this test moved the walker / grid threshold from a quarter
    of the batch to a third: at 1 230 heavy envs the walker was 3.6 % ahead; the macro workload, 1 486 heavy envs and up, wants the grid.
    The third workload, added later in the round, found the automatic choice 7.8 % behind: 410 very heavy envs and no other heavy one
    want placement 1 and the one-env-per-wave solve -- both rules had been fitted to the macro workload, where BOTH lists are long.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', _SCHEDULE_SCRIPT % dict(root=root)], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')][0][7:])
    for name, o in out.items():
        best = min(o['forced'], key=o['forced'].get)
        worst = max(o['forced'], key=o['forced'].get)
        print("%s (%d heavy, %d very heavy envs): automatic %.4f ms per step; forced readings: best %.4f at (%s), worst %.4f at (%s)"
              % (name, o['heavy'], o['very_heavy'], o['auto'], o['forced'][best], best, o['forced'][worst], worst))
        if name == 'no heavy env':
            assert o['heavy'] == 0 and o['very_heavy'] == 0
        elif name == '30 % heavy':
            assert o['heavy'] + o['very_heavy'] >= 0.25 * 4096
        else:
            assert o['very_heavy'] >= 0.08 * 4096
        # (the third workload sits 2.3-2.7 % behind its best forced reading in five runs out of five -- placement 2 with a heavy list
        # that is in fact empty -- a stable, small loss that is stated rather than tuned away; its bound leaves room for jitter)
        assert o['auto'] <= (1.04 if name == '10 % very heavy' else 1.03) * o['forced'][best], (name, o)
