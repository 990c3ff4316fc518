"""Development tool: one-step differences device vs float oracle (and float vs double oracle) as a function of the largest
normal force, over the cases of tests/test_gpu_contacts_fuzz.py -- the data behind the force-scaled tolerances of that test.
Writes gpurun_out/crush_stats.jsonl (one record per check)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle                              # noqa: E402
from real_robots_amd.batched import BatchedREALRobotEnv       # noqa: E402
from real_robots_amd.distributed import synthetic_actions     # noqa: E402


def run(case, seed0, out):
    rng = np.random.default_rng(seed0 * 1000 + case)
    N = int(rng.choice([1, 3, 5, 17, 34, 63, 130]))
    nobj = int(rng.integers(1, 4))
    W, H = [(64, 48), (64, 64), (128, 128), (160, 120), (320, 240)][int(rng.integers(0, 5))]
    pool = rng.choice([None, None, "0", "900", "2500"])
    if pool:
        os.environ['RR_SOLVER_POOL'] = str(pool)
    try:
        env = BatchedREALRobotEnv(N, objects=nobj, width=64, height=64)
    finally:
        os.environ.pop('RR_SOLVER_POOL', None)
    o32, o64 = Oracle(nobj, 64, 64, f32=True), Oracle(nobj, 64, 64)
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
        rng.random(); rng.random()          # (the test draws reset / teleport decisions here; keep the stream aligned loosely)
        flags = (rng.random(N) < 0.5).astype(np.uint8)
        chk = t % 10 == 9
        if chk:
            st0 = env.state
            ncs = np.array([len(env.contacts(i)) for i in range(N)])
            sel = sorted(set(list(np.argsort(-ncs)[:3])))
            cache = {int(i): env.contacts(int(i)) for i in sel}
        env.step(cmd, render=False)
        if not chk:
            continue
        st1 = env.state
        for i in sel:
            rec = dict(case=case, t=t, env=int(i), nobj=nobj)
            res = {}
            for name, o in (('f32', o32), ('f64', o64)):
                o.state = st0[i].astype(np.float64)
                o.set_contact_cache(cache[int(i)])
                o.step(cmd[i].astype(np.float64))
                res[name] = (o.state.copy(), o.contacts())
            cd = env.contacts(i)
            s32, c32 = res['f32']
            s64, c64 = res['f64']
            rec['nc'] = len(cd)
            rec['same_list'] = bool(len(cd) == len(c32) and (len(cd) == 0 or (cd[:, :10] == c32[:, :10].astype(np.float32)).all()))
            if not rec['same_list'] or len(cd) == 0:
                out.write(json.dumps(rec) + "\n")
                continue
            rec['fmax'] = float(cd[:, 10].max())
            d = st1[i].astype(np.float64) - s32
            rec['dq'] = float(np.abs(d[:11]).max()); rec['dqd'] = float(np.abs(d[11:22]).max())
            ob = d[22:22 + 13 * nobj].reshape(nobj, 13)
            rec['dpose'] = float(np.abs(ob[:, :7]).max()); rec['dvel'] = float(np.abs(ob[:, 7:]).max())
            e = s32 - s64
            rec['o_dq'] = float(np.abs(e[:11]).max()); rec['o_dqd'] = float(np.abs(e[11:22]).max())
            ob = e[22:22 + 13 * nobj].reshape(nobj, 13)
            rec['o_dpose'] = float(np.abs(ob[:, :7]).max()); rec['o_dvel'] = float(np.abs(ob[:, 7:]).max())
            rec['qd_max'] = float(np.abs(s32[11:22]).max())
            # solver-independent quantities, evaluated on the float oracle's rows (o32 holds the last step's problem)
            o32.state = st0[i].astype(np.float64); o32.set_contact_cache(cache[int(i)]); o32.step(cmd[i].astype(np.float64))
            r_dev = o32.solution_residual(st1[i], cd[:, 10].astype(np.float64), 1.0)
            r_orc = o32.solution_residual(s32, c32[:, 10], 1.0)
            for k in ('res_sum', 'res_max', 'f_sum', 'f_max', 'n_active'):
                rec['dev_' + k] = float(r_dev[k]); rec['orc_' + k] = float(r_orc[k])
            rec['active_xor'] = bin(r_dev['active'] ^ r_orc['active']).count('1')
            # forces of the contacts whose activity differs
            x = r_dev['active'] ^ r_orc['active']
            rec['xor_fmax'] = float(max([max(cd[c, 10], c32[c, 10]) for c in range(len(cd)) if (x >> c) & 1] or [0.0]))
            rec['df_max'] = float(np.abs(cd[:, 10] - c32[:, 10]).max())
            out.write(json.dumps(rec) + "\n")
    env.close()


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'crush_stats.jsonl'), 'w') as f:
        for case in range(n):
            run(case, 2, f)
            f.flush()
    print("done")
