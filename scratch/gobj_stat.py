"""How many heavy / very heavy envs have an OBJECT in a generic contact row (robot-object, object-object, or an object's fifth+ static contact)?
Those without could hand their object lanes' rows to an object wave (as the light kernel does)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from real_robots_amd import _native as nat
from real_robots_amd.batched import BatchedREALRobotEnv
import importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
for N, nobj, T in ((4096, 3, 370), (4096, 3, 2100), (1024, 1, 2000)):
    cmds = bench.make_commands(torch, np, np.arange(N), T + 1, 1.0, 'cuda:0')
    env = BatchedREALRobotEnv(N, objects=nobj, width=32, height=32, want_mask=False)
    for t in range(T):
        env.step(device_ptr=cmds[t].data_ptr(), render=False)
    env.sync()
    cls = env.host(nat.F_ENV_CLASS)
    out = {}
    for c in (1, 2):
        ids = np.flatnonzero(cls == c)
        n_obj_gen = 0; ngen = []; nos = []; only_os = 0; only_os_le8 = 0; robot_static_only = 0; kinds = {}
        for i in ids:
            ct = env.contacts(int(i))
            a, b = ct[:, 0].astype(int), ct[:, 1].astype(int)          # bodyA, bodyB: -1 static, 0..15 robot, 16+ object
            os_pair = ((a >= 16) & (b < 0)) | ((b >= 16) & (a < 0))
            # object-lane rows: the first four static contacts of each object
            taken = np.zeros(len(ct), bool)
            for o in range(nobj):
                idx = np.flatnonzero(os_pair & ((a == 16 + o) | (b == 16 + o)))[:4]
                taken[idx] = True
            gen = ~taken
            obj_in_gen = gen & ((a >= 16) | (b >= 16))
            n_obj_gen += bool(obj_in_gen.any()); ngen.append(int(gen.sum())); nos.append(int(taken.sum()))
            gen_os = gen & os_pair                       # an object's fifth+ static contact
            robot = gen & (((a >= 0) & (a < 16)) | ((b >= 0) & (b < 16)))
            objobj = gen & (a >= 16) & (b >= 16)
            if gen.any() and (gen_os == gen).all():
                only_os += 1
                per_obj = max(int((os_pair & ((a == 16 + o) | (b == 16 + o))).sum()) for o in range(nobj))
                only_os_le8 += per_obj <= 8
            if gen.any() and not obj_in_gen.any(): robot_static_only += 1
            k = ('os+' if gen_os.any() else '') + ('robot-obj ' if (robot & ((a >= 16) | (b >= 16))).any() else '') + ('robot-static ' if (robot & ~((a >= 16) | (b >= 16))).any() else '') + ('obj-obj' if objobj.any() else '')
            kinds[k] = kinds.get(k, 0) + 1
        out[c] = (len(ids), n_obj_gen, float(np.mean(ngen)) if ngen else 0, float(np.mean(nos)) if nos else 0)
        print('   class', c, 'envs', len(ids), 'generic rows ONLY from an object\'s 5th+ static contacts:', only_os, '(of them <= 8 per object:', only_os_le8, ') robot-static only:', robot_static_only, kinds)
    print('N %d objects %d step %d: heavy %d (object in a generic row: %d; mean generic %.1f, object-lane contacts %.1f) | very heavy %d (%d; %.1f, %.1f)' % (
        (N, nobj, T) + out[1] + out[2]), flush=True)
    env.close()
