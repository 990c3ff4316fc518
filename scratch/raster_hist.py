"""Sizing of the rasteriser (CPU, oracle): histogram of the moving triangles by the number of sample points in their clipped
bounding box, over states of the bench workload (full-range resample-and-hold commands).  python scratch/raster_hist.py"""
import ctypes as C
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.oracle import Oracle, _lib
from real_robots_amd.distributed import synthetic_actions

L = _lib(True)
h = np.zeros(24, np.int64)
L.rro_set_debug_hist.argtypes = [C.c_void_p]
frames = 0
for env in range(6):
    o = Oracle(3, 128, 128, f32=True)
    for t in range(240):
        o.step(synthetic_actions([env], t, seed=3)[0].astype(np.float64))
        if t % 40 == 39:
            L.rro_set_debug_hist(h.ctypes.data)
            o.render()
            L.rro_set_debug_hist(None)
            frames += 1
names = ['0', '1', '2', '3-4', '5-8', '9-16', '17-64', '>64']
print('per frame (%d frames): bucket  triangles  points tested  points covered' % frames)
for b in range(8):
    print('%6s %10.1f %10.1f %10.1f' % (names[b], h[b] / frames, h[8 + b] / frames, h[16 + b] / frames))
print('total  %10.1f %10.1f %10.1f' % (h[:8].sum() / frames, h[8:16].sum() / frames, h[16:].sum() / frames))
