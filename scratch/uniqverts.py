import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from real_robots_amd import model
m = model.load_model()
tp = m['tri_pos']; ir = m['inst_range']
cnt = []
for i in range(len(ir)):
    s, c = ir[i]
    for w in range(s, s + c, 64):
        P = tp[w:w + 64].reshape(-1, 3)
        cnt.append(len({tuple(np.round(v, 7)) for v in P}))
cnt = np.array(cnt)
print('windows', len(cnt), 'mean unique verts', cnt.mean(), 'max', cnt.max(), 'frac <= 64:', (cnt <= 64).mean(), 'hist', np.histogram(cnt, bins=[0, 32, 48, 64, 96, 128, 193])[0])
