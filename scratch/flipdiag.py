"""Diagnose image mismatches of test_raster_parity_many_poses_near_camera: which pixels differ, by how much."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions
from oracle.oracle import Oracle
N, W, H = 24, 128, 128
env = BatchedREALRobotEnv(N, objects=3, width=W, height=H)
o = Oracle(3, W, H)
o32 = Oracle(3, W, H, f32=True)
for t in range(180):
    act = synthetic_actions(range(N), t, seed=11) * 0.8
    env.step(act, render=(t % 60 == 59))
    if t % 60 == 59:
        st, rgb, dep, msk = env.state, env.host(nat.F_RGB), env.host(nat.F_DEPTH), env.host(nat.F_MASK)
        for i in range(N):
            o.state = st[i].astype(np.float64)
            r, d, m = o.render()
            diff = np.abs(r.astype(int) - rgb[i].astype(int)).max(-1)
            dd = np.abs(d - dep[i])
            bad = (diff > 1) | (dd > 1e-5) | (m != msk[i])
            if bad.any():
                ys, xs = np.nonzero(bad)
                print("t", t, "env", i, "bad pixels", len(ys))
                o32.state = st[i].astype(np.float64)
                r32, d32, m32 = o32.render()
                for y, x in list(zip(ys, xs))[:12]:
                    print("  f32 oracle: depth %.7f rgb %s" % (d32[y, x], r32[y, x]))
                    print("  (%3d,%3d) mask hip %d ora %d  depth hip %.7f ora %.7f  rgb hip %s ora %s" %
                          (y, x, msk[i][y, x], m[y, x], dep[i][y, x], d[y, x], rgb[i][y, x], r[y, x]))
                    y0, y1, x0, x1 = max(0, y-1), min(H, y+2), max(0, x-1), min(W, x+2)
                    print("     hip depth nbhd", np.array2string(dep[i][y0:y1, x0:x1], precision=5).replace("\n", " "))
                    print("     ora depth nbhd", np.array2string(d[y0:y1, x0:x1], precision=5).replace("\n", " "))
                    print("     mask nbhd", msk[i][y0:y1, x0:x1].ravel())
