#!/usr/bin/env python3
"""Model compiler: URDF + OBJ + PNG  ->  compact binary blob (`realrobot_model.bin`).

Runs in the build container only (it reads the reference's *data* files under
/root/reference/real_robots/data/kuka_gripper_description; no reference Python
is imported or copied).  The blob is committed under real_robots_amd/data/ and is
the single model source for both the oracle (oracle/rr_oracle.c) and the HIP
library (real_robots_amd/csrc), so neither needs the reference tree at run time.

What is derived (reference anchors):
  * kinematic tree, joint frames/axes, masses, COMs   kuka_gripper.urdf:20-546
  * robot base pose [-0.55, 0, -0.04]                  real_robots/envs/robot.py:46
  * object set / reset poses                           real_robots/envs/robot.py:19-24,49-50
  * contact coefficients                               *.urdf <contact> blocks
  * collision = convex hull of the visual OBJ          SURVEY.md A.1.3 (Bullet convex hull per OBJ)
  * robot link inertia = AABB box inertia of the hull  SURVEY.md A.1.1 (no URDF_USE_INERTIA_FROM_FILE
                                                       for the robot, robot.py:54-56; objects DO use
                                                       the URDF inertia, robot.py:222)
Blob format ("named tensor archive"), little endian:
  char magic[8] = "RRMODEL1"; u32 n_entries; u32 pad;
  n_entries x { char name[32]; u32 dtype(0=f32,1=i32,2=u8); u32 ndim; u32 shape[4]; u64 offset; u64 nbytes }
  payload (each entry 16-byte aligned, offsets from file start)
"""
import os
import struct
import sys
import xml.etree.ElementTree as ET

import numpy as np
from PIL import Image
from scipy.spatial import ConvexHull

REF = os.environ.get("RR_REFERENCE_DATA",
                     "/root/reference/real_robots/data/kuka_gripper_description")
VMAX = 192    # cap of collision vertices per shape (the greedy reduction stops earlier when HULL_TOL is met)
FMAX = 192    # cap of collision planes per shape
MARGIN = 0.001  # Bullet gUrdfDefaultCollisionMargin, used for the inertia AABB only
CLUSTER = 64    # triangles per raster cluster (= one wavefront iteration)


# ----------------------------------------------------------------------------- math
def rpy_to_mat(r, p, y):
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def mat_to_quat(R):
    """xyzw"""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w = (R[2, 1] - R[1, 2]) / s
        x = 0.25 * s
        y = (R[0, 1] + R[1, 0]) / s
        z = (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w = (R[0, 2] - R[2, 0]) / s
        x = (R[0, 1] + R[1, 0]) / s
        y = 0.25 * s
        z = (R[1, 2] + R[2, 1]) / s
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w = (R[1, 0] - R[0, 1]) / s
        x = (R[0, 2] + R[2, 0]) / s
        y = (R[1, 2] + R[2, 1]) / s
        z = 0.25 * s
    q = np.array([x, y, z, w])
    return q / np.linalg.norm(q)


def fl(s):
    return [float(x) for x in s.split()]


# ----------------------------------------------------------------------------- OBJ
def load_obj(path):
    """Returns (tri_pos[T,3,3], tri_nrm[T,3,3], tri_uv[T,3,2], verts[V,3]); polygons fan-triangulated."""
    v, vt, vn, tris = [], [], [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == 'v':
            v.append([float(x) for x in p[1:4]])
        elif p[0] == 'vt':
            vt.append([float(x) for x in p[1:3]])
        elif p[0] == 'vn':
            vn.append([float(x) for x in p[1:4]])
        elif p[0] == 'f':
            idx = []
            for tok in p[1:]:
                a = tok.split('/')
                vi = int(a[0]) - 1
                ti = int(a[1]) - 1 if len(a) > 1 and a[1] else -1
                ni = int(a[2]) - 1 if len(a) > 2 and a[2] else -1
                idx.append((vi, ti, ni))
            for k in range(1, len(idx) - 1):
                tris.append((idx[0], idx[k], idx[k + 1]))
    v = np.array(v, dtype=np.float64)
    vt = np.array(vt, dtype=np.float64) if vt else np.zeros((0, 2))
    vn = np.array(vn, dtype=np.float64) if vn else np.zeros((0, 3))
    T = len(tris)
    tp = np.zeros((T, 3, 3))
    tn = np.zeros((T, 3, 3))
    tu = np.zeros((T, 3, 2))
    for t, tri in enumerate(tris):
        for k, (vi, ti, ni) in enumerate(tri):
            tp[t, k] = v[vi]
            if ti >= 0:
                tu[t, k] = vt[ti]
            if ni >= 0:
                tn[t, k] = vn[ni]
        if tri[0][2] < 0:
            n = np.cross(tp[t, 1] - tp[t, 0], tp[t, 2] - tp[t, 0])
            n /= (np.linalg.norm(n) + 1e-30)
            tn[t, :] = n
    return tp, tn, tu, v


def mtl_info(obj_path):
    """(texture path or None, Kd rgb)"""
    mtl = obj_path[:-4] + '.mtl'
    tex, kd = None, (1.0, 1.0, 1.0)
    if os.path.exists(mtl):
        for line in open(mtl):
            p = line.split()
            if not p:
                continue
            if p[0] == 'map_Kd':
                tex = os.path.join(os.path.dirname(obj_path), p[1])
            elif p[0] == 'Kd':
                kd = tuple(float(x) for x in p[1:4])
    return tex, kd


# ----------------------------------------------------------------------------- hull simplification
# Bullet collides the full convex hull of every OBJ (link_1 has 1 407 vertices).  The device kernel works on
# a reduced vertex set (an inner approximation: a subset of the hull's vertices) and a reduced plane set (an outer
# approximation: a subset of the hull's facet planes), both chosen greedily by the error they remove, until the
# deviation from the full hull is below HULL_TOL or the caps are reached.  The remaining deviations are stored in the
# blob (`shape_dev`) and bounded by tests/test_oracle_pins.py::test_collision_hull_reduction_error.
HULL_TOL = 3.0e-4   # m
NPREF = 6           # the first NPREF planes of every shape are its extreme planes along +-x, +-y, +-z (cheap exact rejection)


def hull_facets(pts):
    """(vertices [H,3], unique facet planes [F,4] with n.x + d <= 0 inside) of the convex hull of pts."""
    hull = ConvexHull(pts)
    eq = hull.equations
    keep, seen = [], set()
    for i, e in enumerate(np.round(eq, 7)):          # coplanar triangles of one facet
        k = e.tobytes()
        if k not in seen:
            seen.add(k)
            keep.append(i)
    return pts[hull.vertices], eq[keep]


def vertex_deviation(hv, sub):
    """How far the full hull sticks out of the hull of the subset `sub` (m), and the vertex that sticks out most."""
    h = ConvexHull(sub)
    out = (hv @ h.equations[:, :3].T + h.equations[:, 3]).max(1)
    i = int(np.argmax(out))
    return max(float(out[i]), 0.0), i


def plane_deviation(hv, eq, chosen):
    """How far the polytope of the plane subset `chosen` sticks out of the full hull (m), and the facet to add."""
    from scipy.spatial import HalfspaceIntersection
    pv = HalfspaceIntersection(eq[chosen], hv.mean(0)).intersections
    viol = pv @ eq[:, :3].T + eq[:, 3]
    viol[:, chosen] = -1.0
    _, fi = np.unravel_index(np.argmax(viol), viol.shape)
    return max(float(viol.max()), 0.0), int(fi)


# Long edges of a shape's full hull: the vertex-in-polytope candidates of the narrow phase cannot see two edges that cross away
# from any vertex (a cube edge lying across a shelf edge); edges shorter than EDGE_MIN are covered by their end points.
EMAX = 48           # long edges kept per shape (the longest first)
EDGE_MIN = 0.004    # m  (1.5 cm until round 6: the cube's bevel edges and the skin pads' short edges were not stored -- tests/test_narrowphase_exact.py)
EDGE_COS = 0.9659   # only sharp edges (dihedral angle >= 15 degrees): on a rounded surface the end points of an edge are close to it


def hull_edges(pts, min_len, cos_tol=0.9995):
    """Long geometric edges of the convex hull of pts: [(p0, p1, n1, n2)] -- segments between two facets whose normals differ,
    collinear pieces with the same pair of facets merged."""
    hull = ConvexHull(pts)
    eq, simp, nb = hull.equations, hull.simplices, hull.neighbors
    # facet id per simplex: union of coplanar neighbours
    parent = list(range(len(simp)))
    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]; a = parent[a]
        return a
    for i in range(len(simp)):
        for j in nb[i]:
            if j > i and eq[i, :3] @ eq[j, :3] > cos_tol and abs(eq[i, 3] - eq[j, 3]) < 1e-4 * max(1.0, abs(eq[i, 3])) + 2e-4:
                parent[find(i)] = find(j)
    fac = np.array([find(i) for i in range(len(simp))])
    # facet normal = area-weighted mean
    fn = {}
    for f in np.unique(fac):
        idx = np.where(fac == f)[0]
        n = np.zeros(3)
        for i in idx:
            a, b, c = pts[simp[i]]
            n += 0.5 * np.linalg.norm(np.cross(b - a, c - a)) * eq[i, :3]
        fn[f] = n / np.linalg.norm(n)
    segs = {}
    for i in range(len(simp)):
        for j in nb[i]:
            if j > i and fac[i] != fac[j]:
                common = list(set(simp[i]) & set(simp[j]))
                if len(common) == 2:
                    key = (min(fac[i], fac[j]), max(fac[i], fac[j]))
                    segs.setdefault(key, []).append(common)
    out = []
    for (f1, f2), lst in segs.items():
        # merge: all segments between the same two facets are collinear (intersection line of two planes)
        P = pts[np.unique(np.array(lst).ravel())]
        d = np.cross(fn[f1], fn[f2]); d /= np.linalg.norm(d)
        t = P @ d
        # chain may have gaps in principle (not for convex facets): take the extent
        p0, p1 = P[np.argmin(t)], P[np.argmax(t)]
        if np.linalg.norm(p1 - p0) >= min_len and fn[f1] @ fn[f2] <= EDGE_COS:
            out.append((p0, p1, fn[f1], fn[f2]))
    return out


# Sample points on long sharp edges, appended to a shape's collision vertices (any boundary point is a valid "vertex" of the
# vertex-in-polytope test).  Why: a long straight edge lying across a SMOOTH convex surface -- the table's rim under an arm link,
# a cube edge under a link -- has no vertex near the touching point, and the smooth side has no sharp edge for the edge-edge
# pass: tests/test_narrowphase_exact.py measured such pairs overlapping by up to 22 mm unseen.  With samples every EDGE_SAMPLE
# metres the miss is bounded by spacing^2 / (8 R) (R: radius of the smooth surface; 1 mm at R = 5 cm).  Statics only: the rims the
# arm can reach (table: top rim without the side under the shelf; shelf: its top rim); robot links: none (their vertex sets are
# at the cap); objects: none (below).  At most EDGE_VCAP vertices per shape (two rounds of 64 lanes in k_collide).
EDGE_SAMPLE = 0.02
EDGE_VCAP = 128


def add_edge_samples(name, owner_type, verts, edges):
    if owner_type == 1 or len(verts) >= EDGE_VCAP:
        return verts
    if name == 'table_base':
        use = [e for e in edges if min(e[0][2], e[1][2]) > 0.27 and not (e[0][0] > 0.25 and e[1][0] > 0.25)]
    elif name == 'table_upper':
        use = [e for e in edges if min(e[0][2], e[1][2]) > 0.30]
    else:
        # (objects: measured and dropped -- a cube edge sample inside a finger's hull takes the normal of the finger's nearest SIDE facet
        # and wedges a grasped cube out of the gripper, tests/test_oracle_pins.py::test_touch_sensor_fires_when_gripper_closes_on_cube)
        return verts
    spacing = EDGE_SAMPLE if owner_type == 0 else 0.015
    pts = []
    for (p0, p1, _, _) in use:                   # (longest edges first; an edge whose samples do not fit the cap any more gets none)
        nseg = int(np.ceil(np.linalg.norm(p1 - p0) / spacing - 1e-9))
        if len(verts) + len(pts) + nseg - 1 > EDGE_VCAP:
            continue
        pts += [p0 + (p1 - p0) * (k / nseg) for k in range(1, nseg)]
    print('shape %-12s + %d edge samples on %d edges (spacing %.1f mm)' % (name, len(pts), len(use), spacing * 1e3))
    return np.vstack([verts, np.array(pts)]) if pts else verts


ROFF_MARGIN = 0.02   # the contact margin the offset radii below are computed for (rr_create's default; a larger one disables their use)


def offset_radius(planes, center, interior):
    """Circumradius about `center` of the polytope {x: n.x <= c + ROFF_MARGIN for all planes}: the shape grown by the contact
    margin PLANE BY PLANE -- the set a vertex must lie in to become a contact candidate (oracle verts_in_planes: largest signed
    distance to the planes below the margin).  At a sharp corner it reaches beyond margin from the shape (by margin / sin of the
    half angle), which is why radius + margin does not bound it.  k_collide: a shape whose bounding sphere of THIS radius lies
    beyond one plane of the other shape cannot take part in any candidate of the pair, in either direction."""
    from scipy.spatial import HalfspaceIntersection
    pl = np.asarray(planes, np.float32).astype(np.float64)          # the planes as the kernels read them
    hs = np.concatenate([pl[:, :3], -(pl[:, 3:4] + ROFF_MARGIN)], axis=1)
    v = HalfspaceIntersection(hs, np.asarray(interior, np.float64)).intersections
    r = float(np.linalg.norm(v - np.asarray(center, np.float64)[None], axis=1).max())
    return r * 1.0005 + 2e-5


def simplify_hull(pts):
    """pts [V,3] (already in the owner's frame). Returns verts[<=VMAX,3], planes[<=FMAX,4] (n, c: n.x<=c inside),
    (vertex deviation, plane deviation) in metres."""
    hv, eq = hull_facets(pts)
    axes = np.vstack([np.eye(3), -np.eye(3)])
    # --- vertices: extremes first, then always the hull vertex farthest outside the current subset's hull
    chosen = []
    for d in axes:
        i = int(np.argmax(hv @ d))
        if i not in chosen:
            chosen.append(i)
    while len(chosen) < 4:
        chosen.append(int(np.argmax(np.min(np.linalg.norm(hv[:, None] - hv[None, chosen], axis=2), axis=1))))
    if len(hv) <= VMAX and len(hv) <= 24:
        chosen, vdev = list(range(len(hv))), 0.0
    else:
        while True:
            try:
                vdev, i = vertex_deviation(hv, hv[chosen])
            except Exception:                            # degenerate start: spread by distance
                vdev, i = 1.0, int(np.argmax(np.min(np.linalg.norm(hv[:, None] - hv[None, chosen], axis=2), axis=1)))
            if vdev < HULL_TOL or len(chosen) >= VMAX or i in chosen:
                break
            chosen.append(i)
    verts = hv[chosen]
    # --- planes: the six extreme facets first (kernel prefilter), then always the facet that cuts off the worst corner
    pch = []
    for d in axes:
        i = int(np.argmax(eq[:, :3] @ d))
        if i not in pch:
            pch.append(i)
    k = 0
    while len(pch) < min(NPREF, len(eq)):                # boxes etc.: fewer than six distinct extremes -> fill in order
        if k not in pch:
            pch.append(k)
        k += 1
    if len(eq) <= 24:
        pch += [i for i in range(len(eq)) if i not in pch]
        pdev = 0.0
    else:
        while True:
            pdev, fi = plane_deviation(hv, eq, pch)
            if pdev < HULL_TOL or len(pch) >= FMAX:
                break
            pch.append(fi)
    planes = np.array([[eq[i, 0], eq[i, 1], eq[i, 2], -eq[i, 3]] for i in pch])
    return verts, planes, (vdev, pdev)


# ----------------------------------------------------------------------------- URDF
class Link:
    pass


def parse_urdf(path):
    root = ET.parse(path).getroot()
    materials = {}
    for m in root.findall('material'):
        c = m.find('color')
        if c is not None:
            materials[m.get('name')] = fl(c.get('rgba'))
    links, joints = {}, []
    for le in root.findall('link'):
        L = Link()
        L.name = le.get('name')
        ine = le.find('inertial')
        L.mass = float(ine.find('mass').get('value'))
        o = ine.find('origin')
        L.com = np.array(fl(o.get('xyz')))
        L.com_rpy = fl(o.get('rpy'))
        I = ine.find('inertia')
        L.inertia = np.array([float(I.get('ixx')), float(I.get('iyy')), float(I.get('izz'))])
        vis = le.find('visual')
        L.mesh = None
        if vis is not None:
            me = vis.find('geometry').find('mesh')
            L.mesh = me.get('filename').split('/')[-1]
            L.scale = np.array(fl(me.get('scale'))) if me.get('scale') else np.ones(3)
            vo = vis.find('origin')
            L.vis_xyz = np.array(fl(vo.get('xyz')))
            L.vis_R = rpy_to_mat(*fl(vo.get('rpy')))
            mat = vis.find('material')
            L.material = mat.get('name') if mat is not None else None
        ce = le.find('contact')
        L.friction, L.restitution, L.rolling, L.spinning = 0.5, 0.0, 0.0, 0.0  # Bullet defaults
        if ce is not None:
            for tag, attr in (('lateral_friction', 'friction'), ('restitution', 'restitution'),
                              ('rolling_friction', 'rolling'), ('spinning_friction', 'spinning')):
                t = ce.find(tag)
                if t is not None:
                    setattr(L, attr, float(t.get('value')))
        links[L.name] = L
    for je in root.findall('joint'):
        J = Link()
        J.name = je.get('name')
        J.type = je.get('type')
        J.parent = je.find('parent').get('link')
        J.child = je.find('child').get('link')
        o = je.find('origin')
        J.xyz = np.array(fl(o.get('xyz')))
        J.R = rpy_to_mat(*fl(o.get('rpy')))
        ax = je.find('axis')
        J.axis = np.array(fl(ax.get('xyz'))) if ax is not None else np.array([0., 0., 1.])
        d = je.find('dynamics')
        J.damping = float(d.get('damping')) if d is not None else 0.0
        lim = je.find('limit')
        J.lower = float(lim.get('lower')) if lim is not None else 0.0
        J.upper = float(lim.get('upper')) if lim is not None else 0.0
        joints.append(J)
    return links, joints, materials


def box_inertia_from_pts(mass, pts_com_frame):
    """Bullet-style AABB box inertia (btCompoundShape/btPolyhedralConvexShape::calculateLocalInertia)."""
    lo = pts_com_frame.min(0) - MARGIN
    hi = pts_com_frame.max(0) + MARGIN
    l = hi - lo
    return mass / 12.0 * np.array([l[1] ** 2 + l[2] ** 2, l[0] ** 2 + l[2] ** 2, l[0] ** 2 + l[1] ** 2])


# ----------------------------------------------------------------------------- build
class Blob:
    def __init__(self):
        self.entries = []

    def add(self, name, arr, dtype):
        np_dt = {0: np.float32, 1: np.int32, 2: np.uint8}[dtype]
        a = np.ascontiguousarray(np.asarray(arr), dtype=np_dt)
        if a.ndim == 0:
            a = a.reshape(1)
        assert a.ndim <= 4 and len(name) < 32
        self.entries.append((name, dtype, a))

    def write(self, path):
        n = len(self.entries)
        head = 16 + n * (32 + 4 + 4 + 16 + 8 + 8)
        off = (head + 15) // 16 * 16
        table = b''
        payload = []
        for name, dt, a in self.entries:
            nb = a.nbytes
            shape = list(a.shape) + [1] * (4 - a.ndim)
            table += struct.pack('<32sII4IQQ', name.encode(), dt, a.ndim, *shape, off, nb)
            payload.append((off, a.tobytes()))
            off = (off + nb + 15) // 16 * 16
        with open(path, 'wb') as f:
            f.write(struct.pack('<8sII', b'RRMODEL1', n, 0))
            f.write(table)
            for o, b in payload:
                f.seek(o)
                f.write(b)
            f.seek(off - 1)
            f.write(b'\0')
        return off


def build_meshlets(P32, max_t=CLUSTER, max_v=CLUSTER):
    """Greedy meshlets over the float32 corner positions P32[T,3,3]: groups of <= max_t triangles that reference <=
    max_v distinct vertex positions (welded by exact float32 equality), grown from a seed over shared vertices, adding
    the triangle that brings the fewest new vertices (ties: closest to the running centroid).  Under-filled meshlets are
    merged with their nearest neighbour when the limits allow.  Returns a list of triangle-index lists."""
    T = len(P32)
    keys = {}
    idx = np.zeros((T, 3), np.int64)
    for t in range(T):
        for k in range(3):
            idx[t, k] = keys.setdefault(P32[t, k].tobytes(), len(keys))
    cent = P32.astype(np.float64).mean(1)
    v2t = {}
    for t in range(T):
        for v in idx[t]:
            v2t.setdefault(int(v), []).append(t)
    used = np.zeros(T, bool)
    order = np.lexsort((cent[:, 0], cent[:, 1], cent[:, 2]))       # deterministic seed sweep (z, then y, then x)
    out, ptr = [], 0
    while True:
        while ptr < T and used[order[ptr]]:
            ptr += 1
        if ptr >= T:
            break
        seed = int(order[ptr])
        cur = [seed]
        used[seed] = True
        verts = set(int(v) for v in idx[seed])
        frontier = set()

        def push(t):
            for v in idx[t]:
                for u in v2t[int(v)]:
                    if not used[u]:
                        frontier.add(u)
        push(seed)
        c0 = cent[seed].copy()
        while len(cur) < max_t and frontier:
            best, bestscore = None, None
            for u in sorted(frontier):
                newv = sum(1 for v in idx[u] if int(v) not in verts)
                if len(verts) + newv > max_v:
                    continue
                sc = (newv, float(np.sum((cent[u] - c0) ** 2)), u)
                if bestscore is None or sc < bestscore:
                    best, bestscore = u, sc
            if best is None:
                break
            frontier.discard(best)
            cur.append(best)
            used[best] = True
            verts.update(int(v) for v in idx[best])
            push(best)
            c0 = cent[cur].mean(0)
        out.append(cur)
    changed = True
    while changed:                                             # merge under-filled meshlets
        changed = False
        out.sort(key=lambda m: (len(m), m[0]))
        for i in range(len(out)):
            if len(out[i]) >= max_t // 2:
                continue
            ci = cent[out[i]].mean(0)
            vi = set(idx[out[i]].ravel().tolist())
            best, bd = None, None
            for j in range(len(out)):
                if j == i or len(out[i]) + len(out[j]) > max_t:
                    continue
                if len(vi | set(idx[out[j]].ravel().tolist())) > max_v:
                    continue
                d = (float(np.sum((cent[out[j]].mean(0) - ci) ** 2)), j)
                if bd is None or d < bd:
                    best, bd = j, d
            if best is not None:
                out[best] = out[best] + out[i]
                del out[i]
                changed = True
                break
    out.sort(key=lambda m: min(m))
    return out


def main(out_path):
    meshes = os.path.join(REF, 'meshes')
    urdf = os.path.join(REF, 'urdf')
    links, joints, materials = parse_urdf(os.path.join(urdf, 'kuka_gripper.urdf'))
    child_joint = {j.child: j for j in joints}
    root_link = [n for n in links if n not in child_joint][0]
    assert root_link == 'lbr_iiwa_link_0'

    # ---- order links depth-first in URDF joint order (Bullet link index order)
    link_order = []

    def walk(name):
        link_order.append(name)
        for j in joints:
            if j.parent == name:
                walk(j.child)
    walk(root_link)
    # link ids: 0 = link_0 (static robot base), 1.. = the 16 children in depth-first order
    link_id = {n: i for i, n in enumerate(link_order)}

    # ---- dynamic bodies = links reached through revolute joints; fixed children merged
    body_links = []        # per body: list of (link name, R_body_link, p_body_link)
    body_of_link = {}      # link -> (body idx or -1 for static base, R, p)
    body_joint = []
    body_of_link[root_link] = (-1, np.eye(3), np.zeros(3))

    def assign(name):
        for j in joints:
            if j.parent != name:
                continue
            pb, pR, pp = body_of_link[name]
            if j.type == 'fixed':
                R = pR @ j.R
                p = pp + pR @ j.xyz
                body_of_link[j.child] = (pb, R, p)
                if pb >= 0:
                    body_links[pb].append((j.child, R, p))
            else:
                b = len(body_joint)
                body_joint.append((j, pb, pR, pp))
                body_links.append([(j.child, np.eye(3), np.zeros(3))])
                body_of_link[j.child] = (b, np.eye(3), np.zeros(3))
            assign(j.child)
    assign(root_link)
    NB = len(body_joint)
    assert NB == 11, NB
    body_names = [body_links[b][0][0] for b in range(NB)]
    print('bodies:', body_names)

    # mesh cache
    mesh_cache = {}

    def get_mesh(fn):
        if fn not in mesh_cache:
            mesh_cache[fn] = load_obj(os.path.join(meshes, fn))
        return mesh_cache[fn]

    def link_points(L):
        """visual/collision vertices in the link frame (scale, then visual origin)."""
        _, _, _, v = get_mesh(L.mesh)
        return (v * L.scale) @ L.vis_R.T + L.vis_xyz

    # ---- per body dynamics parameters
    parent = np.zeros(NB, np.int32)
    jpos = np.zeros((NB, 3))
    jrot = np.zeros((NB, 3, 3))
    axis = np.zeros((NB, 3))
    bmass = np.zeros(NB)
    bcom = np.zeros((NB, 3))
    binertia = np.zeros((NB, 6))
    binertia_urdf = np.zeros((NB, 6))
    damping = np.zeros(NB)
    jlower = np.zeros(NB)
    jupper = np.zeros(NB)
    for b, (j, pb, pR, pp) in enumerate(body_joint):
        parent[b] = pb
        jpos[b] = pp + pR @ j.xyz       # joint origin in the parent *body* frame
        jrot[b] = pR @ j.R
        axis[b] = j.axis / np.linalg.norm(j.axis)
        damping[b] = j.damping
        jlower[b], jupper[b] = j.lower, j.upper
        # merge inertial properties of the body's links (all inertial rpy are zero in this model)
        m_tot, h = 0.0, np.zeros(3)
        parts = []
        for (ln, R, p) in body_links[b]:
            L = links[ln]
            assert np.allclose(L.com_rpy, 0)
            if L.mass <= 0:
                continue
            c = p + R @ L.com
            pts = (link_points(L) - L.com)            # in the link's inertial frame
            Ibox = np.diag(box_inertia_from_pts(L.mass, pts))
            Iurdf = np.diag(L.inertia)
            parts.append((L.mass, c, R @ Ibox @ R.T, R @ Iurdf @ R.T))
            m_tot += L.mass
            h += L.mass * c
        c_tot = h / m_tot
        I1 = np.zeros((3, 3))
        I2 = np.zeros((3, 3))
        for (m, c, Ib, Iu) in parts:
            d = c - c_tot
            par = m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
            I1 += Ib + par
            I2 += Iu + par
        bmass[b] = m_tot
        bcom[b] = c_tot
        binertia[b] = [I1[0, 0], I1[1, 1], I1[2, 2], I1[0, 1], I1[0, 2], I1[1, 2]]
        binertia_urdf[b] = [I2[0, 0], I2[1, 1], I2[2, 2], I2[0, 1], I2[0, 2], I2[1, 2]]

    # ---- objects / statics
    obj_names = ['cube', 'tomato', 'mustard']            # robot.py:49-50 (after "table")
    obj_reset = {                                        # robot.py:19-24 (xyz + rpy)
        'table': [0.0, 0.0, 0.08, 0.0, 0.0, 0.0],
        'mustard': [-0.10, 0.30, 0.45, 1.57080, 3.14159, 0.0],
        'cube': [-0.10, 0.0, 0.45, 0.0, 0.0, 0.0],
        'tomato': [-0.10, -0.30, 0.45, 0.0, 0.0, 0.0]}
    robot_pos = np.array([-0.55, 0.0, -0.04])            # robot.py:46
    table_links, table_joints, _ = parse_urdf(os.path.join(urdf, 'table.urdf'))
    table_pos = np.array(obj_reset['table'][:3])

    obj_mass = np.zeros(3)
    obj_inertia = np.zeros((3, 3))
    obj_pose0 = np.zeros((3, 7))
    obj_links = []
    for i, n in enumerate(obj_names):
        ol, _, _ = parse_urdf(os.path.join(urdf, n + '.urdf'))
        L = list(ol.values())[0]
        assert np.allclose(L.com, 0) and np.allclose(L.vis_xyz, 0)
        obj_links.append(L)
        obj_mass[i] = L.mass
        obj_inertia[i] = L.inertia
        pr = obj_reset[n]
        obj_pose0[i, :3] = pr[:3]
        obj_pose0[i, 3:] = mat_to_quat(rpy_to_mat(*pr[3:]))

    # ---- collision shapes
    # owner_type: 0 static(world frame), 1 robot body, 2 object
    shapes = []
    full_hulls = []     # per shape: ALL vertices of the convex hull of the OBJ (owner frame) -- what Bullet collides (A.1.3)

    def add_shape(name, owner_type, owner_idx, lid, pts_owner_frame, L, body_uid):
        full_hulls.append(pts_owner_frame[ConvexHull(pts_owner_frame).vertices].astype(np.float32))
        verts, planes, dev = simplify_hull(pts_owner_frame)
        edges = sorted(hull_edges(pts_owner_frame, EDGE_MIN), key=lambda e: -np.linalg.norm(e[1] - e[0]))[:EMAX]
        verts = add_edge_samples(name, owner_type, verts, edges)
        c = 0.5 * (pts_owner_frame.min(0) + pts_owner_frame.max(0))
        r = float(np.max(np.linalg.norm(pts_owner_frame - c, axis=1)))
        shapes.append(dict(name=name, otype=owner_type, oidx=owner_idx, link=lid, verts=verts, planes=planes,
                           center=c, radius=r, friction=L.friction, restitution=L.restitution, uid=body_uid, dev=dev,
                           rolling=L.rolling, spinning=L.spinning, edges=edges))
        print('shape %-12s owner(%d,%2d) link %2d  V=%3d F=%3d E=%2d  deviation from the full hull: vertices %.2f mm, planes %.2f mm  r=%.3f' %
              (name, owner_type, owner_idx, lid, len(verts), len(planes), len(edges), dev[0] * 1e3, dev[1] * 1e3, r))

    # statics first: table, shelf, robot base link_0
    for ln in ('table_base', 'table_upper'):
        L = table_links[ln]
        add_shape(ln, 0, 0, -1, link_points(L) + table_pos, L, 1)
    L0 = links[root_link]
    add_shape(root_link, 0, 0, 0, link_points(L0) + robot_pos, L0, 0)
    n_static = len(shapes)
    # robot moving links
    for ln in link_order[1:]:
        L = links[ln]
        b, R, p = body_of_link[ln]
        pts = link_points(L) @ R.T + p
        add_shape(ln, 1, b, link_id[ln], pts, L, 0)
    n_robot = len(shapes) - n_static
    for i, L in enumerate(obj_links):
        add_shape(obj_names[i], 2, i, -1, link_points(L), L, 2 + i)

    NS = len(shapes)
    sh_owner = np.zeros((NS, 4), np.int32)      # otype, oidx, link id, body uid
    sh_nv = np.zeros(NS, np.int32)
    sh_nf = np.zeros(NS, np.int32)
    sh_verts = np.zeros((NS, VMAX, 3))
    sh_planes = np.zeros((NS, FMAX, 4))
    sh_planes[:, :, 3] = 1e9                    # padded planes never bind (n=0, c=+big)
    sh_sphere = np.zeros((NS, 4))
    sh_mat = np.zeros((NS, 2))
    sh_roll = np.zeros((NS, 2))                 # rolling, spinning friction coefficients (URDF <contact>)
    sh_dev = np.zeros((NS, 2))                  # deviation of the reduced vertex / plane set from the full hull (m)
    sh_ne = np.zeros(NS, np.int32)
    sh_roff = np.zeros(NS)                      # offset_radius(): circumradius of the shape grown by ROFF_MARGIN plane by plane
    sh_edges = np.zeros((NS, EMAX, 12))         # long hull edges: p0 (3), p1 - p0 (3), the two facet normals (3 + 3), owner frame
    for s, S in enumerate(shapes):
        sh_owner[s] = [S['otype'], S['oidx'], S['link'], S['uid']]
        nv, nf = len(S['verts']), len(S['planes'])
        sh_nv[s], sh_nf[s] = nv, nf
        sh_verts[s, :nv] = S['verts']
        sh_verts[s, nv:] = S['verts'][0]        # padding repeats vertex 0 (harmless duplicates are masked by nv)
        sh_planes[s, :nf] = S['planes']
        sh_sphere[s, :3] = S['center']
        sh_sphere[s, 3] = S['radius']
        sh_roff[s] = offset_radius(S['planes'], S['center'], np.mean(S['verts'], axis=0))
        assert sh_roff[s] >= S['radius'] + ROFF_MARGIN, (S['name'], sh_roff[s], S['radius'])
        sh_mat[s] = [S['friction'], S['restitution']]
        sh_roll[s] = [S['rolling'], S['spinning']]
        sh_dev[s] = S['dev']
        sh_ne[s] = len(S['edges'])
        for k, (p0, p1, n1, n2) in enumerate(S['edges']):
            sh_edges[s, k] = np.concatenate([p0, p1 - p0, n1, n2])

    # touch sensor links: skin_00, skin_01, skin_10, skin_11  (robot.py:156)
    touch_links = np.array([link_id[n] for n in ('skin_00', 'skin_01', 'skin_10', 'skin_11')], np.int32)

    # ---- link frames relative to bodies (for get_part_pos / parts[name].get_position(): link COM frame)
    NL = len(link_order)
    link_body = np.zeros(NL, np.int32)
    link_pos = np.zeros((NL, 3))
    link_rot = np.zeros((NL, 3, 3))
    for ln in link_order:
        b, R, p = body_of_link[ln]
        i = link_id[ln]
        link_body[i] = b
        link_pos[i] = p + R @ links[ln].com     # BodyPart.get_pose reports the COM frame
        link_rot[i] = R

    # ---- render instances
    textures = []       # list of HxWx3 uint8
    tex_key = {}

    def get_tex(path):
        if path is None:
            return -1
        if path not in tex_key:
            im = Image.open(path)
            a = np.array(im.convert('RGBA'))
            rgb = a[:, :, :3].copy()
            if im.mode == 'RGBA':
                # TinyRenderer ignores alpha; keep RGB as stored
                pass
            if (rgb.reshape(-1, 3) == rgb.reshape(-1, 3)[0]).all():
                rgb = rgb[:1, :1].copy()         # uniform colour image -> 1x1
            tex_key[path] = len(textures)
            textures.append(rgb)
        return tex_key[path]

    inst = []
    tri_pos, tri_nrm, tri_uv = [], [], []
    cl_verts, tri_vidx = [], []

    def add_instance(name, owner_type, owner_idx, uid, L, R, p):
        tp, tn, tu, _ = get_mesh(L.mesh)
        S = L.scale
        A = R @ L.vis_R                                  # rotation after scale
        P = (tp * S) @ A.T + (R @ L.vis_xyz + p)
        Nn = (tn / S) @ A.T                              # inverse-transpose for non-uniform scale
        Nn /= (np.linalg.norm(Nn, axis=2, keepdims=True) + 1e-30)
        texpath, kd = mtl_info(os.path.join(meshes, L.mesh))
        tid = get_tex(texpath)
        # colour: textured meshes show the texel unmodified; untextured use the MTL Kd (skin: 0.8)
        col = (1.0, 1.0, 1.0) if tid >= 0 else kd
        # Raster clusters ("meshlets"): the triangles of the instance are regrouped into clusters of <= CLUSTER
        # triangles over <= CLUSTER distinct vertices; every cluster is padded to exactly CLUSTER triangles with
        # degenerate (zero-area) ones, so a 64-triangle window of the rasteriser is one cluster of one instance: the
        # wave culls it with one sphere test and projects each of its vertices once (one lane per vertex).
        P32 = P.astype(np.float32)
        groups = build_meshlets(P32)
        Pn, Nnn, tun, cverts, vidx = [], [], [], [], []
        for g in groups:
            keys, loc = {}, np.zeros((CLUSTER, 3), np.int32)
            cv = np.zeros((CLUSTER, 3), np.float32)
            for r, t in enumerate(g):
                for k in range(3):
                    key = P32[t, k].tobytes()
                    if key not in keys:
                        keys[key] = len(keys)
                        cv[keys[key]] = P32[t, k]
                    loc[r, k] = keys[key]
            assert len(keys) <= CLUSTER and len(g) <= CLUSTER
            cv[len(keys):] = cv[0]
            padn = CLUSTER - len(g)
            Pg = np.concatenate([P[g], np.repeat(P[g[0]][None, :1, :], 3, axis=1).repeat(padn, axis=0)]) if padn else P[g]
            if padn:
                first_local = loc[0, 0]
                loc[len(g):] = first_local
            Pn.append(Pg)
            Nnn.append(np.concatenate([Nn[g], np.repeat(Nn[g[0]][None], padn, axis=0)]) if padn else Nn[g])
            tun.append(np.concatenate([tu[g], np.repeat(tu[g[0]][None], padn, axis=0)]) if padn else tu[g])
            cverts.append(cv)
            vidx.append(loc)
        P, Nn, tu = np.concatenate(Pn), np.concatenate(Nnn), np.concatenate(tun)
        start = sum(len(x) for x in tri_pos)
        tri_pos.append(P)
        tri_nrm.append(Nn)
        tri_uv.append(tu)
        cl_verts.append(np.stack(cverts))
        tri_vidx.append(np.concatenate(vidx))
        inst.append(dict(name=name, otype=owner_type, oidx=owner_idx, uid=uid, tex=tid, col=col,
                         start=start, count=len(P)))

    for ln in ('table_base', 'table_upper'):
        add_instance(ln, 0, 0, 1, table_links[ln], np.eye(3), table_pos)
    add_instance(root_link, 0, 0, 0, L0, np.eye(3), robot_pos)
    n_static_inst = len(inst)
    for ln in link_order[1:]:
        b, R, p = body_of_link[ln]
        add_instance(ln, 1, b, 0, links[ln], R, p)
    for i, L in enumerate(obj_links):
        add_instance(obj_names[i], 2, i, 2 + i, L, np.eye(3), np.zeros(3))
    tri_pos = np.concatenate(tri_pos)
    tri_nrm = np.concatenate(tri_nrm)
    tri_uv = np.concatenate(tri_uv)
    cl_verts = np.concatenate(cl_verts)                      # [ncl, CLUSTER, 3] float32, instance frame
    tri_vidx = np.concatenate(tri_vidx)                      # [NT, 3] cluster-local vertex indices
    # consistency: the cluster vertices reproduce the float32 corner positions bit for bit
    chk = cl_verts[np.arange(len(tri_vidx)) // CLUSTER][np.arange(len(tri_vidx))[:, None], tri_vidx]
    assert np.array_equal(chk, tri_pos.astype(np.float32)), 'cluster vertices do not reproduce tri_pos'
    tri_vpack = (tri_vidx[:, 0] | (tri_vidx[:, 1] << 8) | (tri_vidx[:, 2] << 16)).astype(np.int32)
    NI = len(inst)
    in_owner = np.zeros((NI, 4), np.int32)   # otype, oidx, uid, tex
    in_range = np.zeros((NI, 2), np.int32)
    in_color = np.zeros((NI, 3))
    for i, I in enumerate(inst):
        in_owner[i] = [I['otype'], I['oidx'], I['uid'], I['tex']]
        in_range[i] = [I['start'], I['count']]
        in_color[i] = I['col']
        print('inst %-16s owner(%d,%2d) uid %d tex %2d tris %5d' % (I['name'], I['otype'], I['oidx'], I['uid'],
                                                                    I['tex'], I['count']))
    # back-face culling eligibility: consistently wound (geometric normal agrees with the vertex normals), positive
    # signed volume (outward CCW) and (nearly) closed surface.  Culling such a mesh does not change what a z-buffered
    # renderer shows (back faces of a closed surface are always hidden); fingers/skins do not qualify and are
    # always rasterised two-sided, like TinyRenderer does for everything.
    from collections import Counter
    in_cull = np.zeros(NI, np.int32)
    for i, I in enumerate(inst):
        P = tri_pos[I['start']:I['start'] + I['count']]
        Nn = tri_nrm[I['start']:I['start'] + I['count']]
        real = ~(np.all(P[:, 0] == P[:, 1], axis=1) & np.all(P[:, 0] == P[:, 2], axis=1))     # drop the padding triangles
        P, Nn = P[real], Nn[real]
        gn = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
        agree = float(((gn * Nn.mean(1)).sum(1) > 0).mean())
        vol = float(np.einsum('ij,ij->i', P[:, 0], np.cross(P[:, 1], P[:, 2])).sum() / 6)
        edges = Counter()
        for t in P:
            k = [tuple(np.round(v, 6)) for v in t]
            for a, b in ((0, 1), (1, 2), (2, 0)):
                edges[tuple(sorted((k[a], k[b])))] += 1
        closed = float(np.mean([c == 2 for c in edges.values()]))
        in_cull[i] = int(agree >= 0.999 and vol > 0 and closed >= 0.9999)   # arm links are open at the joints: not culled
        print('cull %-16s agree %.3f closed %.3f vol %+.2e -> %d' % (I['name'], agree, closed, vol, in_cull[i]))
    # bounding sphere of every 64-triangle cluster in its instance's frame (centre of the AABB, max vertex distance)
    assert len(tri_pos) % CLUSTER == 0
    ncl = len(tri_pos) // CLUSTER
    cl_sphere = np.zeros((ncl, 4))
    for k in range(ncl):
        pts = tri_pos[k * CLUSTER:(k + 1) * CLUSTER].reshape(-1, 3)
        c = 0.5 * (pts.min(0) + pts.max(0))
        cl_sphere[k, :3] = c
        cl_sphere[k, 3] = np.linalg.norm(pts - c, axis=1).max()
    tri_inst = np.zeros(len(tri_pos), np.int32)
    for i, I in enumerate(inst):
        tri_inst[I['start']:I['start'] + I['count']] = i
    tex_info = np.zeros((len(textures), 3), np.int32)   # offset (texels), w, h
    off = 0
    for t, T in enumerate(textures):
        tex_info[t] = [off, T.shape[1], T.shape[0]]
        off += T.shape[0] * T.shape[1]
    # texels stored RGBX (4 bytes) so the device fetches one dword per texel
    tex_data = np.zeros((off, 4), np.uint8)
    for t, T in enumerate(textures):
        o = tex_info[t, 0]
        tex_data[o:o + T.shape[0] * T.shape[1], :3] = T.reshape(-1, 3)
        tex_data[o:o + T.shape[0] * T.shape[1], 3] = 255

    # ---- action protocol constants (robot.py:58-67, env.py:317)
    min_j = -np.ones(9) * np.pi * 0.944
    max_j = np.ones(9) * np.pi * 0.944
    min_j[0], max_j[0] = -np.pi * 0.666, np.pi * 0.666
    min_j[1:9:2] = -np.pi * 0.666
    max_j[1:9:2] = np.pi * 0.666
    min_j[6], max_j[6] = -np.pi * 0.972, np.pi * 0.972
    min_j[-2:] = 0
    max_j[-2:] = np.pi / 2
    max_diff = np.array([0.2, 0.2, 0.2, 0.2, 0.2, 0.3, 0.3, 0.1, 0.1])
    # dof order q[0..10]: joints 1..7, finger00, finger01, finger10, finger11
    dof_names = [j.name for (j, _, _, _) in body_joint]
    print('dofs:', dof_names)
    assert dof_names[7:] == ['base_to_finger00_joint', 'finger00_to_finger01_joint',
                             'base_to_finger10_joint', 'finger10_to_finger11_joint']

    B = Blob()
    F, I32, U8 = 0, 1, 2
    B.add('dims', [NB, NL, NS, NI, len(tri_pos), len(textures), n_static, n_robot, VMAX, FMAX, n_static_inst], I32)
    B.add('robot_pos', robot_pos, F)
    B.add('body_parent', parent, I32)
    B.add('body_jpos', jpos, F)
    B.add('body_jrot', jrot, F)
    B.add('body_axis', axis, F)
    B.add('body_mass', bmass, F)
    B.add('body_com', bcom, F)
    B.add('body_inertia', binertia, F)            # Bullet-style AABB inertia (default)
    B.add('body_inertia_urdf', binertia_urdf, F)  # URDF <inertia> alternative
    B.add('body_damping', damping, F)
    B.add('body_limits', np.stack([jlower, jupper], 1), F)
    B.add('obj_mass', obj_mass, F)
    B.add('obj_inertia', obj_inertia, F)
    B.add('obj_pose0', obj_pose0, F)
    B.add('table_pos', table_pos, F)
    B.add('shape_owner', sh_owner, I32)
    B.add('shape_nv', sh_nv, I32)
    B.add('shape_nf', sh_nf, I32)
    B.add('shape_verts', sh_verts, F)
    B.add('shape_planes', sh_planes, F)
    B.add('shape_sphere', sh_sphere, F)
    B.add('shape_mat', sh_mat, F)
    B.add('shape_roll', sh_roll, F)
    B.add('shape_dev', sh_dev, F)
    B.add('shape_ne', sh_ne, I32)
    B.add('shape_edges', sh_edges, F)
    B.add('shape_roff', np.concatenate([sh_roff, [ROFF_MARGIN]]), F)      # [NS] radii + the margin they hold for
    B.add('touch_links', touch_links, I32)
    B.add('link_body', link_body, I32)
    B.add('link_pos', link_pos, F)
    B.add('link_rot', link_rot, F)
    B.add('inst_owner', in_owner, I32)
    B.add('inst_range', in_range, I32)
    B.add('inst_color', in_color, F)
    B.add('inst_cull', in_cull, I32)
    B.add('tri_pos', tri_pos, F)
    B.add('tri_nrm', tri_nrm, F)
    B.add('tri_uv', tri_uv, F)
    B.add('tri_inst', tri_inst, I32)
    B.add('cluster_sphere', cl_sphere, F)
    B.add('cluster_verts', cl_verts, F)
    B.add('tri_vidx', tri_vpack, I32)
    B.add('tex_info', tex_info, I32)
    B.add('tex_data', tex_data, U8)
    B.add('act_min', min_j, F)
    B.add('act_max', max_j, F)
    B.add('act_maxdiff', max_diff, F)
    size = B.write(out_path)
    names = '\n'.join(link_order)
    with open(os.path.splitext(out_path)[0] + '_links.txt', 'w') as f:
        f.write(names + '\n')
    print('wrote %s (%.2f MB), %d tris, %d shapes, %d instances, %d textures' %
          (out_path, size / 1e6, len(tri_pos), NS, NI, len(textures)))
    # RR_FULL_HULLS_OUT=tests/golden/full_hulls.npz: the full hull vertex sets as a test fixture (tests/test_narrowphase_exact.py
    # compares the oracle's contacts with the exact signed distance of the FULL hulls; the reference's OBJ files do not travel)
    fh = os.environ.get('RR_FULL_HULLS_OUT')
    if fh:
        np.savez_compressed(fh, names=np.array([S['name'] for S in shapes]), **{'hull_%d' % i: h for i, h in enumerate(full_hulls)})
        print('wrote %s: %s hull vertices' % (fh, [len(h) for h in full_hulls]))


if __name__ == '__main__':
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(
        os.path.dirname(os.path.abspath(__file__)), '..', 'real_robots_amd', 'data', 'realrobot_model.bin')
    main(out)
