"""One-step parity of the pushing scenario under non-default create-time parameters (solver iterations, ERP, margin, dt)."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from oracle.oracle import Oracle
N = 34
for kw_env, kw_or in (({'solver_iters': 5}, {'solver_iters': 5}), ({'erp': 0.9}, {'erp': 0.9}), ({'margin': 0.01}, {'margin': 0.01}),
                      ({'dt': 0.0025, 'solver_iters': 20}, {'dt': 0.0025, 'solver_iters': 20})):
    env = BatchedREALRobotEnv(N, objects=3, width=64, height=64, **kw_env)
    o = Oracle(3, 64, 64, f32=True, **kw_or)
    rng = np.random.default_rng(5)
    env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
    plans = [env.get_plan(i) for i in range(N)]
    wj = wo = 0.0; checked = 0; mism = 0; heavy = 0
    for t in range(600):
        chk = t >= 150 and t % 25 == 0
        if chk:
            ncs = np.array([len(env.contacts(i)) for i in range(N)]); sel = np.argsort(-ncs)[:3]; st0 = env.state
        env.step_plan(render=False)
        if chk:
            st1 = env.state
            for i in sel:
                o.state = st0[i].astype(np.float64); o.step(plans[i][t].astype(np.float64))
                mism += len(env.contacts(i)) != len(o.contacts()); heavy = max(heavy, len(env.contacts(i)))
                wj = max(wj, np.abs(st1[i][:22] - o.state[:22]).max())
                d = (st1[i][22:] - o.state[22:]).reshape(3, 13); wo = max(wo, np.abs(d[:, :3]).max()); checked += 1
    print(kw_env, "checked", checked, "max contacts", heavy, "contact-count mismatches", mism, "worst joints %.2e object pos %.2e" % (wj, wo), "errflags", int((env.host(nat.F_ERRFLAGS) != 0).sum()), flush=True)
    env.close()
