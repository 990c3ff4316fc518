"""Workgroup timeline of k_collide (development build, RR_ABLATE=131072): the collision pass of a step without camera, alone on
the machine -- slots busy, gaps, the spread of the shader engines' last ends."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['RR_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so')
os.environ['RR_ABLATE'] = '131072'
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
N = 4096
cmds = bench.make_commands(torch, np, np.arange(N), 200, 1.0, 'cuda:0')
env = BatchedREALRobotEnv(N, objects=3, width=128, height=128, want_mask=False)
for t in range(172):
    env.step(device_ptr=cmds[t].data_ptr(), render=True)
env.sync()
lib = nat.load_library()
buf = np.zeros((65536, 3), np.uint64)
for rep in range(3):
    env.step(device_ptr=cmds[172 + rep].data_ptr(), render=bool(rep % 2)); env.sync()
    lib.rr_debug_collide_wgtime(buf.ctypes.data_as(ctypes.c_void_p))
    b = buf[:N]; st, en, hw = b[:, 0].astype(np.int64), b[:, 1].astype(np.int64), b[:, 2]
    t0 = st.min(); st -= t0; en -= t0; span = en.max(); dur = en - st
    xcc = ((hw >> np.uint64(32)) & np.uint64(15)).astype(np.int64); hid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    se = (hid >> 13) & 7; cu = xcc * 65536 + ((hid >> 8) & 0xff)
    keys = np.unique(cu)
    print('rep', rep, '(render %s)' % bool(rep % 2), 'span %.1f us' % (span / 100.0), 'duration us: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f' % tuple(x / 100.0 for x in (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max())))
    print('   slots busy (sum of durations / (CUs x 4 x span)): %.3f; ideal span at full slots %.1f us' % (dur.sum() / (len(keys) * 4.0 * span), dur.sum() / (len(keys) * 4.0) / 100.0))
    ends = np.array([en[(xcc == x) & (se == s)].max() for x in range(8) for s in range(4)]); work = np.array([dur[(xcc == x) & (se == s)].sum() for x in range(8) for s in range(4)])
    print('   shader engines: last end mean %.1f min %.1f max %.1f us; work max / mean %.3f' % (ends.mean() / 100.0, ends.min() / 100.0, ends.max() / 100.0, work.max() / work.mean()))
    o = np.argsort(st); q = N // 4
    print('   mean duration of the workgroups by start quarter: ' + ' '.join('%.1f' % (dur[o[i * q:(i + 1) * q]].mean() / 100.0) for i in range(4)), '; starts of the quarters (us): ' + ' '.join('%.1f' % (st[o[i * q]] / 100.0) for i in range(4)))
    print('   start of the k-th workgroup (us): ' + ' '.join('%d:%.1f' % (k, st[o[k]] / 100.0) for k in (64, 128, 256, 512, 768, 1023, 1500, 2047, 3071, 4000, 4095)))
    last = np.argsort(-en)[:16]
    print('   last to end (start, duration us): ' + ' '.join('(%.0f, %.0f)' % (st[i] / 100.0, dur[i] / 100.0) for i in last))
    busy = np.zeros(int(span) + 1); np.add.at(busy, st, 1); np.add.at(busy, en, -1); busy = np.cumsum(busy)
    print('   workgroups resident at 5 us steps: ' + ' '.join('%d' % busy[min(int(t * 100), len(busy) - 1)] for t in np.arange(2.5, span / 100.0, 5.0)))
env.close()
