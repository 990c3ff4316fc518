"""Per render instance: is the mesh a closed, consistently oriented 2-manifold (every directed edge matched by exactly one
opposite directed edge)?  Input to the question whether back-facing triangles can ever be visible."""
import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from collections import Counter
from real_robots_amd import model
m = model.load_model()
tp = m['tri_pos']; ti = m['tri_inst']
deg = np.all(tp[:, 0] == tp[:, 1], -1) & np.all(tp[:, 0] == tp[:, 2], -1)
for i in range(len(m['inst_owner'])):
    sel = np.nonzero((ti == i) & ~deg)[0]
    if not len(sel): continue
    verts = {}
    def vid(v):
        return verts.setdefault(tuple(np.asarray(v).tolist()), len(verts))
    de = Counter(); zero = 0
    for t in sel:
        a, b, c = (vid(tp[t, k]) for k in range(3))
        if a == b or b == c or a == c: zero += 1; continue
        for e in ((a, b), (b, c), (c, a)): de[e] += 1
    unmatched = sum(1 for (a, b), n in de.items() if de.get((b, a), 0) != n)
    multi = sum(1 for n in de.values() if n > 1)
    # signed volume
    P = tp[sel].astype(np.float64)
    vol = np.einsum('ij,ij->i', P[:, 0], np.cross(P[:, 1], P[:, 2])).sum() / 6
    print("inst %2d owner %s tris %5d verts %5d zero-area %3d directed edges %6d unmatched %5d multi %4d volume %+.3e" %
          (i, m['inst_owner'][i][:2], len(sel), len(verts), zero, len(de), unmatched, multi, vol))
