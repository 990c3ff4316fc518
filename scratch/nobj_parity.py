"""One-step parity of the pushing scenario with one and two objects."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from oracle.oracle import Oracle
N = 33
for nobj in (1, 2):
    env = BatchedREALRobotEnv(N, objects=nobj, width=64, height=64)
    o = Oracle(nobj, 64, 64, f32=True)
    rng = np.random.default_rng(6)
    env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
    plans = [env.get_plan(i) for i in range(N)]
    wj = wo = 0.0; checked = mism = heavy = 0
    ns = 22 + 13 * 3
    for t in range(600):
        chk = t >= 150 and t % 25 == 0
        if chk:
            ncs = np.array([len(env.contacts(i)) for i in range(N)]); sel = np.argsort(-ncs)[:3]; st0 = env.state
        env.step_plan(render=(t % 10 == 0))
        if chk:
            st1 = env.state
            for i in sel:
                o.state = st0[i].astype(np.float64); o.step(plans[i][t].astype(np.float64))
                mism += len(env.contacts(i)) != len(o.contacts()); heavy = max(heavy, len(env.contacts(i)))
                wj = max(wj, np.abs(st1[i][:22] - o.state[:22]).max())
                d = (st1[i][22:22 + 13 * nobj] - o.state[22:22 + 13 * nobj]).reshape(nobj, 13); wo = max(wo, np.abs(d[:, :3]).max()); checked += 1
    r, d, m = o.render()
    o.state = env.state[0].astype(np.float64); r, d, m = o.render()
    env.render()
    print("objects", nobj, "checked", checked, "max contacts", heavy, "contact-count mismatches", mism, "worst joints %.2e object pos %.2e" % (wj, wo),
          "image mask equal", bool((env.host(nat.F_MASK)[0] == m).all()), "rgb max diff", int(np.abs(env.host(nat.F_RGB)[0].astype(int) - r.astype(int)).max()), flush=True)
    env.close()
