"""Goal dataset generator on the batched simulator (mirror of real_robots/generate_goals.py).

The reference draws one placement at a time, lets a single pybullet env settle (<= 1000 steps), snapshots the retina,
mask and poses, and rejects placements that break the separation / orientation / start-goal predicates
(generate_goals.py:21-68 runEnv, 79-108 generatePosition, 133-226 drawPosition, 249-272 isOnShelf/isOnTable,
229-246 checkRepeatability, 275-365 generateGoalREAL2020, 435-436 file format). Here B candidate goals are drawn per round and the B envs settle in
lock-step on the GPU; the predicates and thresholds are the reference's. Output: `np.savez_compressed(path, goals)` with
`Goal` objects (fields initial_state, final_state, retina, retina_before, mask, challenge, subtype), loadable by
`REALRobotEnv.load_goals`.
"""
import math

import numpy as np

from . import _native as nat
from .batched import BatchedREALRobotEnv, OBJECT_NAMES
from .envs.env import Goal
from .mathutil import quat_from_euler


def _orient_diff(q1, q2):
    return np.minimum(np.linalg.norm(q1 - q2, axis=-1), np.linalg.norm(q1 + q2, axis=-1))


def settle(env, max_t=1000):
    """Batched runEnv (generate_goals.py:21-68): zero command until every object moved < 1e-4 m and < 1e-3 (quaternion)
    for more than 20 consecutive steps. Returns (poses [N, n_obj, 7], failed [N])."""
    poses = env.host(nat.F_OBJ_POSE).astype(np.float64)
    stable = np.zeros(env.N, np.int64)
    done = np.zeros(env.N, bool)
    for t in range(max_t):
        old = poses
        env.step(None)
        poses = env.host(nat.F_OBJ_POSE).astype(np.float64)
        pos_diff = np.linalg.norm(old[:, :, :3] - poses[:, :, :3], axis=2).max(1)
        or_diff = _orient_diff(old[:, :, 3:], poses[:, :, 3:]).max(1)
        ok = (pos_diff < 0.0001) & (or_diff < 0.001) & (t > 10)
        stable = np.where(ok, stable + 1, 0)
        done |= stable > 20
        if done.all():
            break
    return poses, ~done


def generate_position(rng, base_orient, fixed, table_plane):
    """generate_goals.py:79-108."""
    if table_plane is None:
        min_x, max_x = -.25, .25
    elif table_plane:
        min_x, max_x = -.25, .05
    else:
        min_x, max_x = .10, .25
    x = rng.random() * (max_x - min_x) + min_x
    y = rng.random() * 0.9 - 0.45
    z = 0.40 if x <= 0.05 else 0.50
    if fixed:
        orientation = base_orient
    else:
        orientation = quat_from_euler(*(rng.random(3) * math.pi * 2))
    return np.array([x, y, z] + list(orientation))


def _min_separation(p):          # p [n_obj, >=3]
    if len(p) < 2:
        return np.inf
    d = np.linalg.norm(p[:, None, :3] - p[None, :, :3], axis=2)
    return d[d > 0].min()


def two_near_objects(p, max_dist):
    """generate_goals.py:296-313 / 325-338: some pair of objects no farther apart than max_dist (p [n_obj, >=3])."""
    if len(p) < 2:
        return False
    d = np.linalg.norm(p[:, None, :3] - p[None, :, :3], axis=2)
    return bool((d[~np.eye(len(p), dtype=bool)] <= max_dist).any())


def is_on_shelf(obj, z):          # generate_goals.py:249-259
    return z > ({'mustard': 0.545}.get(obj, 0.55) - 0.15)


def is_on_table(obj, z):          # generate_goals.py:262-272
    return z < ({'tomato': 0.49}.get(obj, 0.48) - 0.15)


class _Drawer:
    def __init__(self, env, rng, n_obj):
        self.env, self.rng, self.n_obj = env, rng, n_obj
        env.reset()
        base, _ = settle(env)
        self.base_orient = base[0, :, 3:]          # settled default orientations ("basePosition")

    def draw(self, fixed_orientation, on_table_only, min_separation):
        """Batched drawPosition (generate_goals.py:133-226): one candidate per env. Returns start poses, settled
        poses, retina, mask and a validity mask."""
        env, N, n = self.env, self.env.N, self.n_obj
        start = np.zeros((N, n, 7))
        for e in range(N):
            placed = []
            for o in self.rng.permutation(n):
                while True:
                    pose = generate_position(self.rng, self.base_orient[o], fixed_orientation, True if on_table_only else None)
                    if not placed or _min_separation(np.array([p for _, p in placed] + [pose])) >= min_separation:
                        break
                placed.append((o, pose))
            for o, pose in placed:
                start[e, o] = pose
        env.reset()
        settle(env)
        env.set_object_poses(start.astype(np.float32))           # one upload for the whole batch
        actual, failed = settle(env)
        env.render()
        retina, mask = env.host(nat.F_RGB), env.host(nat.F_MASK)
        valid = ~failed
        for e in range(N):
            if not valid[e]:
                continue
            if _min_separation(actual[e]) < min_separation:
                valid[e] = False
            elif fixed_orientation and (_orient_diff(start[e, :, 3:], actual[e, :, 3:]) > 0.041).any():
                valid[e] = False
        return start, actual, retina, mask, valid


def generate_goals(n_2d_goals=25, n_25d_goals=15, n_3d_goals=10, n_obj=3, seed=None, batch=64, width=320, height=240,
                   device=0, max_rounds=50, max_objects_dist=2.0, repeatability=False):
    """Returns the list of Goal objects (2D, then 2.5D, then 3D) like generate_goals.main (generate_goals.py:406-436).
    max_objects_dist: generateGoalREAL2020's predicate of the same name (its default, 2 m, never binds on a 0.5 x 0.9 m table).
    Resampling differs from the reference when it DOES bind: the reference redraws only the final placement of a candidate whose
    objects are too far apart (generate_goals.py:325-338), this batched form rejects the whole (initial, final) candidate -- the
    accepted goals satisfy the same predicates, their distribution is not the same for a binding max_objects_dist.
    repeatability=True also runs check_repeatability on the result, as the reference's main does (generate_goals.py:438), and
    returns (goals, (maxDiffPos, maxDiffOr))."""
    rng = np.random.default_rng(seed)
    env = BatchedREALRobotEnv(batch, objects=n_obj, width=width, height=height, device=device)
    drawer = _Drawer(env, rng, n_obj)
    names = OBJECT_NAMES[:n_obj]
    goals = []
    specs = [("2D", n_2d_goals, False, 0.2, 0.25), ("2.5D", n_25d_goals, True, 0.2, 0.25), ("3D", n_3d_goals, True, 0.2, 0.0)]
    for goal_type, count, on_shelf, min_start_goal, min_obj_dist in specs:
        fixed = goal_type != '3D'
        got, rounds = [], 0
        while len(got) < count and rounds < max_rounds:
            rounds += 1
            _, ini, ret0, _, ok0 = drawer.draw(fixed, not on_shelf, min_obj_dist)
            _, fin, ret1, msk1, ok1 = drawer.draw(fixed, not on_shelf, min_obj_dist)
            for e in range(batch):
                if len(got) >= count or not (ok0[e] and ok1[e]):
                    continue
                # generate_goals.py:311-351: one object on the shelf in the initial or final state (not for 2D),
                # every object moved by at least min_start_goal in the table plane
                shelf = goal_type == '2D' or any(is_on_shelf(names[o], ini[e, o, 2]) or is_on_shelf(names[o], fin[e, o, 2])
                                                for o in range(n_obj))
                moved = all(np.linalg.norm(fin[e, o, :2] - ini[e, o, :2]) >= min_start_goal for o in range(n_obj))
                # generate_goals.py:296-313, 325-338: (3D goals with several objects) two objects within max_objects_dist of each
                # other in the initial state, or else in the final one
                near = n_obj == 1 or goal_type != '3D' or two_near_objects(ini[e], max_objects_dist) or two_near_objects(fin[e], max_objects_dist)
                if not (shelf and moved and near):
                    continue
                g = Goal(initial_state={names[o]: ini[e, o].copy() for o in range(n_obj)},
                         final_state={names[o]: fin[e, o].copy() for o in range(n_obj)},
                         retina=ret1[e].copy(), retina_before=ret0[e].copy(), challenge=goal_type, mask=msk1[e].copy())
                g.subtype = str(n_obj)
                got.append(g)
        if len(got) < count:
            raise RuntimeError("could not generate %d %s goals in %d rounds" % (count, goal_type, max_rounds))
        goals += got
    rep = check_repeatability(env, goals, drawer) if repeatability else None
    env.close()
    return (goals, rep) if repeatability else goals


def check_repeatability(env, goals, drawer=None):
    """Batched checkRepeatability (generate_goals.py:229-246): every goal's objects are put back at its initial state
    (generateRealPosition, :111-121: reset, settle, re-pose, settle) -- env.N goals at a time -- and the settled poses are compared
    with the state they started from.  Returns (maxDiffPos, maxDiffOr) over the goals, the reference's two figures -- including
    its quirk: its maxDiffOr is max(maxDiffPos, diffOr of the last goal) (:242), kept so that the two programs print the same --
    or 1000000 when a placement does not settle within 1000 steps (:244-246)."""
    N = env.N
    names = list(goals[0].initial_state) if goals else []
    n = len(names)
    order = [OBJECT_NAMES.index(k) for k in names]
    max_pos, max_or = 0.0, 0.0
    for g0 in range(0, len(goals), N):
        chunk = goals[g0:g0 + N]
        start = np.zeros((N, env.n_objects, 7))
        env.reset()
        settle(env)
        start[:] = env.host(nat.F_OBJ_POSE).astype(np.float64)          # envs beyond the chunk keep their settled default
        for e, g in enumerate(chunk):
            for k, o in zip(names, order):
                start[e, o] = g.initial_state[k]
        env.set_object_poses(start.astype(np.float32))
        actual, failed = settle(env)
        for e, g in enumerate(chunk):
            p0 = np.vstack([g.initial_state[k] for k in names])
            p1 = np.vstack([actual[e, o] for o in order])
            diff_pos = float(np.linalg.norm(p1[:, :3] - p0[:, :3]))
            diff_or = float(min(np.linalg.norm(p1[:, 3:] - p0[:, 3:]), np.linalg.norm(p1[:, 3:] + p0[:, 3:])))
            max_pos = max(max_pos, diff_pos)
            max_or = max(max_pos, diff_or)
            print("Replicated diffPos:{} diffOr:{}".format(diff_pos, diff_or))
            if failed[e]:
                print("*****************FAILED************!!!!")
                return 1000000
    return max_pos, max_or


def save_goals(path, goals):
    """np.savez_compressed(path, list_of_goals) (generate_goals.py:435-436; env.py:143-145 reads items()[0][1])."""
    import real_robots  # noqa: F401  the alias package makes Goal pickle as real_robots.envs.env.Goal (the reference's path)
    assert Goal.__module__ == 'real_robots.envs.env'
    arr = np.empty(len(goals), dtype=object)
    for i, g in enumerate(goals):
        arr[i] = g
    np.savez_compressed(path, arr)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Generate a goals dataset for REALRobot (batched, MI355X)")
    ap.add_argument('--seed', type=int, default=None)
    ap.add_argument('--n_2d_goals', type=int, default=25)
    ap.add_argument('--n_25d_goals', type=int, default=15)
    ap.add_argument('--n_3d_goals', type=int, default=10)
    ap.add_argument('--n_obj', type=int, default=3)
    a = ap.parse_args(argv)
    # the reference's order (generate_goals.py:435-438): the dataset is on disk BEFORE the repeatability check runs -- an exception or
    # an interrupt in the check's reset / settle passes loses nothing
    goals = generate_goals(a.n_2d_goals, a.n_25d_goals, a.n_3d_goals, a.n_obj, a.seed)
    save_goals('goals-REAL2020-s{}-{}-{}-{}-{}.npy'.format(a.seed, a.n_2d_goals, a.n_25d_goals, a.n_3d_goals, a.n_obj), goals)
    env = BatchedREALRobotEnv(64, objects=a.n_obj, width=320, height=240)
    try:
        max_pos, max_or = check_repeatability(env, goals)
    finally:
        env.close()
    print("Repeatability check: maxDiffPos %g maxDiffOr %g" % (max_pos, max_or))


if __name__ == '__main__':
    main()
