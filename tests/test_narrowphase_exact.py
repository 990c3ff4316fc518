"""CPU test (-m "not gpu"): the oracle's narrow phase against EXACT polytope geometry.

What Bullet computes behind `stepSimulation` (reference real_robots/envs/env.py:340) for a pair of convex hulls is the closest
points / penetration depth of the two FULL hulls (GJK / EPA).  The oracle (and the device, bit for bit the same
lists) replaces that by vertex-in-polytope + edge-edge candidates on hulls reduced to <= 192 vertices / planes.
This is the one part of "what Bullet would compute" that can be pinned without Bullet: the exact signed distance of two convex
polytopes is the distance of the origin from their Minkowski difference, computed here with scipy's qhull on the full hull
vertex sets of the reference's OBJ files (tests/golden/full_hulls.npz, written by tools/compile_model.py with
RR_FULL_HULLS_OUT; the OBJ files themselves do not travel).

For every colliding shape-pair class -- finger / skin vs cube, tomato, mustard; each object vs table, shelf and each other; arm
links vs table -- random relative poses are drawn with the exact gap in [-4 mm, +18 mm] (the 2 cm margin is Bullet's contact
breaking threshold) and the oracle's DEEPEST contact of the pair is compared with the exact signed distance and separating
direction.  `python tests/test_narrowphase_exact.py [n]` runs n (default 2000) poses per class on all cores and writes
tests/golden/narrowphase_exact_summary.json (the numbers quoted in DESIGN.md 3); the pytest entry runs a seeded subset and
asserts the bounds stated in BOUNDS below.
"""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..'))
from oracle.oracle import Oracle  # noqa: E402

MARGIN = 0.02
TABLE, SHELF = 0, 1
CUBE, TOMATO, MUSTARD = 19, 20, 21
OBJ_NAME = {CUBE: 'cube', TOMATO: 'tomato', MUSTARD: 'mustard'}
FINGER_SHAPES = {'finger_00': 11, 'finger_01': 12, 'skin_01': 13, 'skin_00': 14, 'finger_10': 15, 'finger_11': 16, 'skin_11': 17, 'skin_10': 18}
ARM_SHAPES = {'link_3': 5, 'link_4': 6, 'link_5': 7, 'link_6': 8, 'link_7': 9, 'base': 10}

_hulls = None


def hulls():
    global _hulls
    if _hulls is None:
        H = np.load(os.path.join(HERE, 'golden', 'full_hulls.npz'))
        _hulls = [H['hull_%d' % i].astype(np.float64) for i in range(len(H['names']))]
    return _hulls


def tri_closest_to_origin(T):
    """closest points to the origin on the triangles T [n, 3, 3] (Voronoi-region tests, vectorised) -> [n, 3]"""
    a, b, c = T[:, 0], T[:, 1], T[:, 2]
    ab, ac = b - a, c - a
    d1, d2 = -(ab * a).sum(1), -(ac * a).sum(1)
    d3, d4 = -(ab * b).sum(1), -(ac * b).sum(1)
    d5, d6 = -(ab * c).sum(1), -(ac * c).sum(1)
    vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
    out = np.empty_like(a)
    done = np.zeros(len(a), bool)

    def put(m, p):
        m = m & ~done
        out[m] = p[m]
        done[m] = True
    with np.errstate(divide='ignore', invalid='ignore'):
        put((d1 <= 0) & (d2 <= 0), a)
        put((d3 >= 0) & (d4 <= d3), b)
        put((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[:, None] * ab)
        put((d6 >= 0) & (d5 <= d6), c)
        put((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac)
        put((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[:, None] * (c - b))
        den = 1.0 / (va + vb + vc)
        put(np.ones(len(a), bool), a + ab * (vb * den)[:, None] + ac * (vc * den)[:, None])
    return out


def signed_distance(A, B):
    """Exact signed distance of two convex polytopes given by their vertices (world frame) = distance of the origin from the
    Minkowski difference A - B.  d > 0: Euclidean distance of the closest points; d < 0: penetration depth (the shortest
    translation that separates them -- what EPA returns).  n: unit direction from B towards A (A moves along +n to separate)."""
    from scipy.spatial import ConvexHull
    D = (A[:, None, :] - B[None, :, :]).reshape(-1, 3)
    h = ConvexHull(D)
    d0 = h.equations[:, 3]
    if (d0 <= 0).all():
        k = int(np.argmax(d0))
        return float(d0[k]), -h.equations[k, :3]
    P = tri_closest_to_origin(D[h.simplices[d0 > 0]])       # (only facets that face the origin can hold the closest point)
    r = np.linalg.norm(P, axis=1)
    k = int(np.argmin(r))
    if r[k] < 1e-12:            # touching: the direction is the facet's
        k = int(np.argmax(d0))
        return 0.0, -h.equations[k, :3]
    return float(r[k]), P[k] / r[k]


def rand_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class Pair:
    """One shape pair in the oracle's pair order (sa, sb) with the free object `mover` (0..2) that is placed; `exact()` is
    the signed distance of the two full hulls at the oracle's own shape transforms."""

    def __init__(self, o, sa, sb):
        self.o, self.sa, self.sb = o, sa, sb

    def contacts(self):
        return self.o.pair_contacts(self.sa, self.sb)

    def exact(self):
        _, (Ra, pa), (Rb, pb) = self.contacts()
        H = hulls()
        return signed_distance(H[self.sa] @ Ra.T + pa, H[self.sb] @ Rb.T + pb)


def place_object(pair, obj, mover_is_a, gap, rng):
    """Moves object `obj` along the exact separating direction until the exact signed distance of the pair is `gap`
    (a translation by t changes the distance by at most |t|: approach from outside in steps, then push in)."""
    o = pair.o
    for _ in range(12):
        d, n = pair.exact()
        target = max(gap, 0.0)
        if abs(d - target) < 2e-5:
            break
        s = o.state
        step = (d - target) * (1.0 if d > target else 1.0)
        s[22 + 13 * obj:22 + 13 * obj + 3] += (-n if mover_is_a else n) * step          # A moves along -n to approach, B along +n
        o.state = s
    if gap < 0:
        d, n = pair.exact()
        s = o.state
        s[22 + 13 * obj:22 + 13 * obj + 3] += (-n if mover_is_a else n) * (d - gap)
        o.state = s


def sample_free(o, sa, sb, rng):
    """Object sa against a static / another object / a robot shape sb: random orientation of the object, random direction."""
    H = hulls()
    obj = sa - CUBE if sa >= CUBE else sb - CUBE
    mover_is_a = sa >= CUBE
    other = sb if mover_is_a else sa
    pair = Pair(o, sa, sb)
    _, (Ra, pa), (Rb, pb) = pair.contacts()
    Ro, po = (Rb, pb) if mover_is_a else (Ra, pa)
    Wo = H[other] @ Ro.T + po
    if other in (TABLE, SHELF):
        # above (mostly) or beside the box, over its footprint with an overhang -- faces, edges and corners of the box all occur
        lo, hi = Wo.min(0), Wo.max(0)
        c = np.array([rng.uniform(lo[0] - 0.05, hi[0] + 0.05), rng.uniform(lo[1] - 0.05, hi[1] + 0.05), hi[2] + 0.25])
        if other == TABLE:          # keep the table-top cases clear of the shelf: it is another pair
            c[0] = rng.uniform(lo[0] - 0.05, -0.08)
    else:
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        c = Wo.mean(0) + u * 0.35
    s = o.state
    s[22 + 13 * obj:22 + 13 * obj + 3] = c
    s[22 + 13 * obj + 3:22 + 13 * obj + 7] = rand_quat(rng)
    o.state = s
    # first approach: straight towards the other shape's centre (statics: straight down), then along the exact direction
    return pair, obj, mover_is_a


GRIP_Q = np.array([0.3, 0.9, -0.2, -1.2, 0.4, 0.8, 0.1, 0.0, 0.0, 0.0, 0.0])       # gripper in free space above the table


def run_case(cls, seed):
    """One random pose of class `cls` = (kind, sa, sb) -> dict(exact d, n, oracle deepest contact dist, angle, count)."""
    rng = np.random.default_rng(seed)
    kind, sa, sb = cls
    o = Oracle(3, 32, 32)
    # (a third each: overlapping by up to 4 mm, touching within 2 mm, speculative up to 18 mm)
    gap = (rng.uniform(-0.004, 0.0), rng.uniform(0.0, 0.002), rng.uniform(0.002, 0.018))[seed % 3]
    s = o.state
    for k in range(3):      # park all objects far away from everything
        s[22 + 13 * k:22 + 13 * k + 3] = [3.0 + k, 3.0, 3.0]
    if kind == 'free':
        if sa < CUBE:       # robot shape vs object: a random gripper opening in free space
            q = GRIP_Q.copy()
            q[7] = q[9] = rng.uniform(0.0, 1.5)
            q[8] = q[10] = -rng.uniform(0.0, 1.5)
            s[:11] = q
        o.state = s
        pair, obj, mover_is_a = sample_free(o, sa, sb, rng)
        place_object(pair, obj, mover_is_a, gap, rng)
    else:                   # arm link vs table: joint angles scaled from upright towards a random posture until the gap is met
        o.state = s
        pair = Pair(o, sa, sb)
        lim = np.array([2.96, 2.09, 2.96, 2.09, 2.96, 2.09, 3.05])
        for _ in range(200):
            q1 = np.zeros(11)
            q1[:7] = rng.uniform(-lim, lim)
            q1[7] = q1[9] = rng.uniform(0.0, 1.5)
            q1[8] = q1[10] = -rng.uniform(0.0, 1.5)
            s[:11] = q1
            o.state = s
            if pair.exact()[0] < gap:
                break
        else:
            return None
        lo_, hi_ = 0.0, 1.0         # d(0) > gap (upright), d(1) < gap
        for _ in range(40):
            mid = 0.5 * (lo_ + hi_)
            s[:11] = mid * q1
            o.state = s
            d = pair.exact()[0]
            if abs(d - gap) < 2e-5:
                break
            if d > gap:
                lo_ = mid
            else:
                hi_ = mid
    d, n = pair.exact()
    c, _, _ = pair.contacts()
    rec = dict(d=d, n_contacts=int(len(c)))
    if len(c):
        k = int(np.argmin(c[:, 9]))
        rec.update(dist=float(c[k, 9]), angle=float(np.degrees(np.arccos(np.clip(c[k, 6:9] @ n, -1.0, 1.0)))))
        # (round 6) the kept contact that REPRESENTS the exact closest feature pair: among the <= 4 contacts of the pair the one whose
        # distance is nearest the exact signed distance.  For separated shapes the "deepest" contact above is usually a vertex candidate
        # whose plane distance under-estimates the gap (a speculative contact that activates, correctly, when the vertex enters the
        # other hull through that face); the edge-edge candidate beside it carries the true distance and direction.
        m = int(np.argmin(np.abs(c[:, 9] - d)))
        rec.update(match_err=float(abs(c[m, 9] - d)), match_angle=float(np.degrees(np.arccos(np.clip(c[m, 6:9] @ n, -1.0, 1.0)))))
    return rec


def classes():
    out = []
    for ob in (CUBE, TOMATO, MUSTARD):
        out.append(('%s-table' % OBJ_NAME[ob], ('free', ob, TABLE)))
        out.append(('%s-shelf' % OBJ_NAME[ob], ('free', ob, SHELF)))
    out += [('cube-tomato', ('free', CUBE, TOMATO)), ('cube-mustard', ('free', CUBE, MUSTARD)), ('tomato-mustard', ('free', TOMATO, MUSTARD))]
    for nm, sh in FINGER_SHAPES.items():
        for ob in (CUBE, TOMATO, MUSTARD):
            out.append(('%s-%s' % (nm, OBJ_NAME[ob]), ('free', sh, ob)))
    for nm, sh in ARM_SHAPES.items():
        out.append(('arm:%s-table' % nm, ('arm', sh, TABLE)))
    return out


def summarise(recs):
    """Per class: how far the oracle's deepest contact is from exact geometry, by regime of the exact signed distance."""
    recs = [r for r in recs if r is not None]
    out = dict(n=len(recs))
    bins = (('penetrating', -1.0, 0.0), ('touching_0_2mm', 0.0, 0.002), ('speculative_2_18mm', 0.002, 0.0181))
    for name, lo, hi in bins:
        rr = [r for r in recs if lo <= r['d'] < hi]
        have = [r for r in rr if r['n_contacts'] > 0]
        e = np.array([r['dist'] - r['d'] for r in have]) if have else np.zeros(0)
        a = np.array([r['angle'] for r in have]) if have else np.zeros(0)
        out[name] = dict(n=len(rr), missed=len(rr) - len(have), unseen_overlap=int(sum(1 for r in have if r['d'] < -5e-4 and r['dist'] > 0.0)),
                         err_mm_max=float(np.abs(e).max() * 1e3) if len(e) else 0.0, err_mm_p99=float(np.percentile(np.abs(e), 99) * 1e3) if len(e) else 0.0,
                         err_mm_median=float(np.median(np.abs(e)) * 1e3) if len(e) else 0.0,
                         deeper_mm_max=float(max(0.0, (-e).max()) * 1e3) if len(e) else 0.0,      # oracle reports the shapes CLOSER than they are
                         farther_mm_max=float(max(0.0, e.max()) * 1e3) if len(e) else 0.0,
                         angle_deg_max=float(a.max()) if len(a) else 0.0, angle_deg_p99=float(np.percentile(a, 99)) if len(a) else 0.0,
                         angle_deg_median=float(np.median(a)) if len(a) else 0.0,
                         match_angle_deg_median=float(np.median([r['match_angle'] for r in have])) if have else 0.0,
                         match_err_mm_median=float(np.median([r['match_err'] for r in have]) * 1e3) if have else 0.0)
    return out


def _job(args):
    cls, seed = args
    return run_case(cls, seed)


def run_all(n, workers):
    import multiprocessing as mp
    res = {}
    with mp.Pool(workers) as pool:
        for ci, (name, cls) in enumerate(classes()):
            recs = pool.map(_job, [(cls, 100000 * ci + i) for i in range(n)], chunksize=16)
            res[name] = summarise(recs)
            print(name, json.dumps(res[name]), flush=True)
    return res


# Stated bounds (round 5, with the rim samples of tools/compile_model.py in the model; 2 000 poses per class in
# tests/golden/narrowphase_exact_summary.json, 78 000 poses in all):
#   * NO pair whose full hulls overlap or come within 2 mm goes without a contact, in any class;
#   * overlapping / touching pairs: the deepest contact's distance is within 3.2 mm (objects, gripper) / 4.7 mm (arm links) of the
#     exact signed distance in 99 % of the poses, the median error is below 0.45 mm; the worst cases (2-13 mm objects / gripper,
#     3-17 mm arm links) are overlaps along an EDGE-EDGE axis between an edge that is not stored (shorter than 4 mm, dihedral
#     angle below 15 degrees, beyond the 48 longest) and a smooth surface -- `unseen_overlap` counts the poses where the hulls
#     overlap by more than 0.5 mm and the deepest contact still reports a gap (0-21 of 667 per object / gripper class, 9-29 per arm link);
#   * the contact NORMAL is the exact separating direction (median 0.0 degrees) where a face is involved -- objects on table and
#     shelf, arm links on the table --, and the face normal of the nearer shape (median 4-33 degrees off the exact direction) where
#     the closest features are two edges or an edge and a vertex of random-oriented gripper / object pairs: the vertex-in-polytope
#     scheme reports a FACE normal by construction, Bullet's GJK / EPA the edge-edge direction;
#   * separated pairs (speculative contacts, 2-18 mm): the plane distance under-estimates the Euclidean distance of vertex-edge
#     and vertex-vertex features by up to 6.7 mm -- a speculative contact that acts a little early, never a missed touch.
BOUNDS = dict(p99_mm=dict(obj=3.3, arm=4.8), median_mm=0.45, subset_p90_mm=dict(obj=3.3, arm=6.0))


def _kind(name):
    return 'arm' if name.startswith('arm:') else 'obj'


def test_committed_summary_holds_the_stated_bounds():
    """The 2 000-pose-per-class run (python tests/test_narrowphase_exact.py) as committed: the numbers DESIGN.md 3 quotes."""
    d = json.load(open(os.path.join(HERE, 'golden', 'narrowphase_exact_summary.json')))
    assert d['poses_per_class'] >= 2000 and len(d['classes']) == len(classes())
    for name, s in d['classes'].items():
        for regime in ('penetrating', 'touching_0_2mm'):
            r = s[regime]
            assert r['n'] >= 600 and r['missed'] == 0, (name, regime, r)
            assert r['err_mm_p99'] <= BOUNDS['p99_mm'][_kind(name)] and r['err_mm_median'] <= BOUNDS['median_mm'], (name, regime, r)
        assert s['speculative_2_18mm']['missed'] <= 4, (name, s['speculative_2_18mm'])


@pytest.mark.parametrize('name,cls', classes())
def test_deepest_contact_matches_exact_geometry(name, cls):
    """A seeded subset of fresh poses per class against the same bounds (the percentile of a small sample: p90 instead of p99)."""
    n = 9 if cls[0] == 'arm' else 24
    recs = [r for r in (run_case(cls, 7000 + i) for i in range(n)) if r is not None]
    assert len(recs) >= n - 1
    near = [r for r in recs if r['d'] < 0.002]
    assert all(r['n_contacts'] > 0 for r in near), name          # nothing overlapping or touching goes unseen
    e = np.array([abs(r['dist'] - r['d']) for r in near]) * 1e3
    assert np.percentile(e, 90) <= BOUNDS['subset_p90_mm'][_kind(name)], (name, np.sort(e)[-4:])


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    res = run_all(n, int(os.environ.get('RR_WORKERS', '8')))
    with open(os.path.join(HERE, 'golden', 'narrowphase_exact_summary.json'), 'w') as f:
        json.dump(dict(poses_per_class=n, margin=MARGIN, classes=res), f, indent=1)
