"""-m gpu: FREE-RUNNING parity of the HIP path (f32) against the float64 oracle on the benchmark's own workload.

`north_star` asks for poses that match "on identical action sequences within a stated float tolerance".  Every other differential
test is one teacher-forced step from the device state; this one lets 64 envs of the headline workload (bench.make_commands:
full-range resample-and-hold joint commands, reference protocol env.py:314-356) run for 2 000 steps on the device and, with the
same commands, on the float64 oracle, nobody correcting anybody.  Contact dynamics with friction is not a contraction: once an arm
slides on the table or pushes an object, a rounding difference decides stick against slip and the two trajectories part for good
-- in this code as in any pair of floating-point implementations (Bullet float vs Bullet double alike).  What can be stated, and
is asserted below (numbers of round 5, MI355X):

  * up to an env's first contact of the robot (with table, shelf or object) or between two objects -- free arm motion under the
    position motors, objects at rest on the table -- the joints agree to JOINT_TOL rad and the object poses to POSE_TOL (m /
    quaternion component) over up to 2 000 steps (measured: 2.4e-6 rad, 9.0e-6);
  * the horizon over which an env stays within DIV_TOL (1e-3 rad / m) although its robot is in contact: at least half of the envs
    never leave it in 2 000 steps (measured: 43 of 64, with 42 of the 64 arms touching something on the way), the 10th percentile of
    the divergence step is bounded from below (measured: 435);
  * the error distribution at fixed horizons is printed (median env at step 1999: 3.6e-7 rad, 9.0e-6 m).
"""
import numpy as np
import pytest

from oracle.oracle import Oracle
from real_robots_amd.batched import BatchedREALRobotEnv
from real_robots_amd.distributed import synthetic_actions

pytestmark = pytest.mark.gpu

N, T = 64, 2000
JOINT_TOL = 1.0e-5          # rad, up to the env's first robot / object-object contact
POSE_TOL = 3.0e-5           # m / quaternion component, up to the env's first robot / object-object contact
DIV_TOL = 1.0e-3            # an env counts as diverged from the first step with a joint / pose error above this
P10_DIV_MIN = 200           # steps: nine envs in ten stay within DIV_TOL at least this long


def _ids(c):
    return tuple((int(r[0]), int(r[1]), int(r[2])) for r in c)


def test_free_running_trajectories_against_the_float64_oracle():
    ids = np.arange(N)
    env = BatchedREALRobotEnv(N, objects=3, width=32, height=32)
    orc = [Oracle(3, 32, 32) for _ in range(N)]
    first = np.full(N, T, np.int64)             # first step whose contact lists differ (bodies / link of every contact, in order)
    touch = np.full(N, T, np.int64)             # first step with a contact of the robot (with table, shelf or an object) or between two objects
    div = np.full(N, T, np.int64)               # first step with a joint error > DIV_TOL rad or a pose error > DIV_TOL
    ej = np.zeros((N, T)); eo = np.zeros((N, T))
    cmd = None
    for t in range(T):
        if t % 20 == 0:
            cmd = synthetic_actions(ids, t, hold_prob=0.05).astype(np.float32)       # bench.make_commands' epochs
        env.step(cmd, render=False)
        st = env.state
        for i in range(N):
            o = orc[i]
            o.step(cmd[i].astype(np.float64))
            co = o.contacts()
            if first[i] == T and _ids(env.contacts(i)) != _ids(co):
                first[i] = t
            if touch[i] == T and any(0 <= r[0] < 16 or (r[0] >= 16 and r[1] >= 16) for r in co):
                touch[i] = t
            ref = o.state
            ej[i, t] = np.abs(st[i][:11] - ref[:11]).max()
            eo[i, t] = max(np.abs(st[i][22 + 13 * k:29 + 13 * k] - ref[22 + 13 * k:29 + 13 * k]).max() for k in range(3))
            if div[i] == T and (ej[i, t] > DIV_TOL or eo[i, t] > DIV_TOL):
                div[i] = t
    env.close()

    def worst_before(stop):
        return (max(ej[i, :stop[i]].max() if stop[i] else 0.0 for i in range(N)), max(eo[i, :stop[i]].max() if stop[i] else 0.0 for i in range(N)))
    pct = lambda a: (int(np.median(a)), int(np.percentile(a, 10)), int(np.percentile(a, 90)), int((a == T).sum()))
    print("\nfree-running f32 device vs f64 oracle, %d envs x %d steps of the headline workload:" % (N, T))
    print("  first robot / object-object contact:          median step %d, p10 / p90 %d / %d, never in %d envs" % pct(touch))
    print("  first contact-list difference:                median step %d, p10 / p90 %d / %d, never in %d envs" % pct(first))
    print("  first error above %.0e (rad / m / quat):      median step %d, p10 / p90 %d / %d, never in %d envs" % ((DIV_TOL,) + pct(div)))
    stop = np.minimum(first, touch)
    wj, wo = worst_before(stop)
    print("  up to the first of {robot touches anything, two objects touch, lists differ}: joints max %.2e rad, object poses max %.2e" % (wj, wo))
    wj2, wo2 = worst_before(first)
    print("  up to the first contact-list difference:             joints max %.2e rad, object poses max %.2e" % (wj2, wo2))
    for h in (100, 500, 1000, 1999):
        print("  at step %4d: joints median / p90 / max %.1e / %.1e / %.1e rad, poses %.1e / %.1e / %.1e" % (
            h, np.median(ej[:, h]), np.percentile(ej[:, h], 90), ej[:, h].max(), np.median(eo[:, h]), np.percentile(eo[:, h], 90), eo[:, h].max()))
    assert wj <= JOINT_TOL and wo <= POSE_TOL, (wj, wo)
    assert (div == T).sum() * 2 >= N and np.percentile(div, 10) >= P10_DIV_MIN, pct(div)
