"""Workgroup timeline of k_raster (development build, RR_ABLATE=16384): how many of the four workgroup slots of a CU hold a running
workgroup, the gap between a workgroup's end and the start of the next one on its CU, the spread of workgroup durations."""
import os, sys, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ['RR_LIB'] = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'real_robots_amd', 'csrc', 'librealrobot_hip_stats.so')
os.environ['RR_ABLATE'] = '0' if (len(sys.argv) > 1 and sys.argv[1] == 'shade') else '16384'
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
    env.step(device_ptr=cmds[t].data_ptr(), render=(t >= 168))
env.sync()
lib = nat.load_library()
SHADE = len(sys.argv) > 1 and sys.argv[1] == 'shade'      # timeline of k_shade's workgroups instead (two per (env, tile): 32 768 slots of 65 536)
if SHADE: lib.rr_debug_shade_ablate(0x10000)
buf = np.zeros((65536, 3), np.uint64)
for rep in range(3):
    lib.rr_debug_raster_wgtime(None, 1)
    env.render(); env.sync()
    lib.rr_debug_raster_wgtime(buf.ctypes.data_as(ctypes.c_void_p), 0)
    b = buf[:N * 8] if SHADE else buf[:N * 4]
    SL = 8 if SHADE else 4                                # workgroup slots per CU
    ok = b[:, 1] > 0
    st, en, hw = b[ok, 0].astype(np.int64), b[ok, 1].astype(np.int64), b[ok, 2]
    t0 = st.min(); st -= t0; en -= t0
    span = en.max()
    dur = en - st
    cu = ((hw >> np.uint64(32)) & np.uint64(15)).astype(np.int64) * 65536 + ((hw >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)   # xcc, (se, sh, cu)
    keys = np.unique(cu)
    print('rep', rep, 'workgroups', ok.sum(), 'CUs seen', len(keys), 'span %.1f us' % (span / 100.0),
          'duration us: mean %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f' % tuple(x / 100.0 for x in (dur.mean(), np.percentile(dur, 10), np.percentile(dur, 50), np.percentile(dur, 90), dur.max())))
    print('   slot occupancy (sum of durations / (CUs x slots x span)): %.3f' % (dur.sum() / (len(keys) * float(SL) * span)))
    # per CU: concurrency over time and the gaps between an end and the next start
    gaps = []; conc = np.zeros(10); last_end = []
    for k in keys[:256]:
        m = cu == k
        ev = sorted([(s, 1) for s in st[m]] + [(e, -1) for e in en[m]])
        c = 0; prev = 0
        for tme, d in ev:
            conc[min(c, 9)] += tme - prev; prev = tme; c += d
        conc[0] += span - prev
        ss, ee = np.sort(st[m]), np.sort(en[m])
        # the i-th start after the first four follows the (i-4)-th end
        if len(ss) > SL: gaps.extend((ss[SL:] - ee[:len(ss) - SL]).tolist())
        last_end.append(ee[-1])
    gaps = np.array(gaps)
    print('   time with c workgroups on a CU (fraction of span), c = 0..9:', np.round(conc / conc.sum(), 3))
    print('   end -> next start on the CU, us: mean %.2f p50 %.2f p90 %.2f' % (gaps.mean() / 100.0, np.percentile(gaps, 50) / 100.0, np.percentile(gaps, 90) / 100.0))
    le = np.array(last_end); print('   last end per CU relative to span: mean %.3f min %.3f' % ((le / span).mean(), le.min() / span))
    tl = (np.flatnonzero(ok) // 2) % 4 if SHADE else np.flatnonzero(ok) % 4
    for tt in range(4):
        m = tl == tt
        print('   tile %d: duration us mean %.2f p50 %.2f p90 %.2f max %.2f | start us mean %.1f min %.1f max %.1f | end max %.1f' % (tt, dur[m].mean() / 100.0, np.percentile(dur[m], 50) / 100.0,
              np.percentile(dur[m], 90) / 100.0, dur[m].max() / 100.0, st[m].mean() / 100.0, st[m].min() / 100.0, st[m].max() / 100.0, en[m].max() / 100.0))
    print('   sum of durations / all slots = %.1f us (span %.1f us); workgroups still running at 80 / 90 / 95 %% of the span: %d / %d / %d' % (dur.sum() / (256.0 * SL) / 100.0, span / 100.0,
          ((st < 0.8 * span) & (en > 0.8 * span)).sum(), ((st < 0.9 * span) & (en > 0.9 * span)).sum(), ((st < 0.95 * span) & (en > 0.95 * span)).sum()))
    late = en > 0.9 * span
    print('   of the workgroups ending in the last 10 %%: tiles', np.bincount(tl[late], minlength=4), 'their mean duration %.1f us, mean start at %.2f of the span' % (dur[late].mean() / 100.0, (st[late] / span).mean()))
    mk = np.zeros((65536, 8), np.uint64)
    if not SHADE and lib.rr_debug_raster_wgmarks(mk.ctypes.data_as(ctypes.c_void_p)) == 0:
        mk = mk[:N * 4][ok].astype(np.int64)
        names = ['flags read', 'LDS filled, instances staged', 'clusters culled', 'window loop', 'near-plane pass', 'list written']
        for tt in range(4):
            m = tl == tt
            full = mk[m][(mk[m][:, :7] > 0).all(axis=1)]          # (a workgroup that leaves after the cull -- empty tile -- has no later marks)
            d = np.diff(full[:, :7], axis=1) / 100.0
            if len(d) == 0: continue
            print('   tile %d marks of the %d workgroups that rasterise (us, mean / median): ' % (tt, len(d)) + '; '.join('%s %.2f / %.2f' % (names[i], d[:, i].mean(), np.median(d[:, i])) for i in range(6)))
    if SHADE and lib.rr_debug_raster_wgmarks(mk.ctypes.data_as(ctypes.c_void_p)) == 0:
        m6 = mk[:N * 8][ok].astype(np.int64)
        full = (m6[:, 1:7] > 0).all(axis=1)
        seq = np.concatenate([b[ok, 0].astype(np.int64)[full, None], m6[full, 1:7], b[ok, 1].astype(np.int64)[full, None]], axis=1)
        d = np.diff(seq, axis=1) / 100.0
        names = ['count known', 'constants staged', 'list entry', 'key + record', 'shaded (texel)', 'stores issued', 'rest of the loop + exit']
        print('   marks of thread 0, first trip, %d workgroups with fragments (us, mean / median / p90): ' % full.sum() + '; '.join('%s %.2f / %.2f / %.2f' % (names[i], d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)) for i in range(7)))
        print('   whole workgroup: mean %.2f us' % ((seq[:, -1] - seq[:, 0]).mean() / 100.0))
    per_cu = np.array([(cu == k).sum() for k in keys]); print('   workgroups per CU: min %d max %d' % (per_cu.min(), per_cu.max()))
os.makedirs(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out'), exist_ok=True)
np.save(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'gpurun_out', 'rwgtime_last.npy'), buf[:N * 8])
env.close()
