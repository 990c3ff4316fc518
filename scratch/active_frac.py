import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd.batched import BatchedREALRobotEnv
N = 1024
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
rng = np.random.default_rng(0)
env.plan_macro(rng.uniform([-0.25, -0.5], [0.05, 0.5], size=(N, 2, 2)))
for t in range(1000):
    env.step_plan(render=False)
    if t in (250, 450, 700):
        tot = act = 0; heavy_tot = heavy_act = 0; os_tot = os_act = 0
        for i in range(N):
            c = env.contacts(i)
            if not len(c): continue
            a = c[:, 0].astype(int); b = c[:, 1].astype(int)
            rob = ((a >= 0) & (a < 16)) | ((b >= 0) & (b < 16))
            tot += rob.sum(); act += (c[rob, 10] > 0).sum()
            osm = (a >= 16) & (b < 0); os_tot += osm.sum(); os_act += (c[osm, 10] > 0).sum()
            if rob.sum() >= 20: heavy_tot += rob.sum(); heavy_act += (c[rob, 10] > 0).sum()
        print("t", t, "robot contacts", tot, "with force %.0f%%" % (100 * act / max(tot, 1)), "| in envs with >=20 robot contacts: %d, with force %.0f%%" % (heavy_tot, 100 * heavy_act / max(heavy_tot, 1)),
              "| object-static %d with force %.0f%%" % (os_tot, 100 * os_act / max(os_tot, 1)))
