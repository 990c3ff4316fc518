import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import model, kinematics as K
m = model.load_model()
W = H = 128
q = np.zeros(11)
if len(sys.argv) > 1: q[:7] = np.random.default_rng(int(sys.argv[1])).uniform(-0.8, 0.8, 7)
R, p, _ = K.forward(q)
tp = m['tri_pos'].astype(np.float64); ti = m['tri_inst']
owner = m['inst_owner']
V = K.look_at(np.array([0.01, 0, 1.2]), m['table_pos'].astype(np.float64), np.array([0, 0, 1.0]))
Pm = K.perspective(80.0, 1.0, 0.1, 100.0)
VP = Pm @ V
sx = np.zeros((len(tp), 3)); sy = np.zeros_like(sx); cwv = np.zeros_like(sx)
for i in range(len(owner)):
    ot, oi = owner[i][0], owner[i][1]
    if ot == 1: Ri, pi = R[oi], p[oi]
    elif ot == 2: Ri, pi = K.quat_to_mat(m['obj_pose0'][oi][3:]), m['obj_pose0'][oi][:3]
    else: Ri, pi = np.eye(3), np.zeros(3)
    sel = ti == i
    wp = tp[sel] @ Ri.T + pi
    c = np.concatenate([wp, np.ones(wp.shape[:2] + (1,))], -1) @ VP.T
    sx[sel] = (c[..., 0] / c[..., 3] + 1) * 0.5 * W; sy[sel] = (c[..., 1] / c[..., 3] + 1) * 0.5 * H; cwv[sel] = c[..., 3]
dyn = np.arange(len(tp)) >= 3200
x0 = np.ceil(np.maximum(sx.min(1), 0)); x1 = np.floor(np.minimum(sx.max(1), W - 1))
y0 = np.ceil(np.maximum(sy.min(1), 0)); y1 = np.floor(np.minimum(sy.max(1), H - 1))
deg = np.all(tp[:, 0] == tp[:, 1], -1)
live = dyn & (x1 >= x0) & (y1 >= y0) & ~deg
area = ((x1 - x0 + 1) * (y1 - y0 + 1))
sa = (sx[:, 1] - sx[:, 0]) * (sy[:, 2] - sy[:, 0]) - (sx[:, 2] - sx[:, 0]) * (sy[:, 1] - sy[:, 0])
print('dyn tris', dyn.sum(), 'padded/degenerate', (dyn & deg).sum(), 'live', live.sum(), 'front-facing live', (live & (sa > 0)).sum())
a = area[live]
print('bbox area hist:', {k: int(((a >= lo) & (a < hi)).sum()) for k, (lo, hi) in {'1': (1, 2), '2': (2, 3), '3-4': (3, 5), '5-8': (5, 9), '9-16': (9, 17), '17-32': (17, 33), '33-64': (33, 65), '65+': (65, 1e9)}.items()})
print('sum area small(<=32)', a[a <= 32].sum(), 'sum area big', a[a > 32].sum(), 'true covered (|sa|/2)', np.abs(sa[live]).sum() / 2)
# per-window (64 tris) max small area -> divergence cost
tot_iter = 0; tot_work = 0; nwin = 0
for w0 in range(3200, len(tp), 64):
    l = live[w0:w0 + 64]; ar = area[w0:w0 + 64]
    small = l & (ar <= 32)
    if l.any(): nwin += 1
    if small.any(): tot_iter += ar[small].max(); tot_work += ar[small].sum()
print('windows', nwin, 'small-loop iterations (sum of per-window max)', tot_iter, 'useful lane-iterations', tot_work, 'efficiency', tot_work / (64 * tot_iter))
for i in range(len(owner)):
    sel = ti == i
    print(i, owner[i], 'tris', sel.sum(), 'live', (live & sel).sum(), 'mean area', area[live & sel].mean() if (live & sel).any() else 0)
