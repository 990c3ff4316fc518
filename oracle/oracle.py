"""ctypes wrapper around the CPU oracle (oracle/rr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Never imported by real_robots_amd (the product path fails loudly without its HIP library instead).
"parity unpinned" w.r.t. PyBullet -- see rr_oracle.h.
"""
import ctypes as C
import gzip
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BLOB_GZ = os.path.join(_HERE, '..', 'real_robots_amd', 'data', 'realrobot_model.bin.gz')
_blob_cache = None


def model_blob():
    global _blob_cache
    if _blob_cache is None:
        with gzip.open(_BLOB_GZ, 'rb') as f:
            _blob_cache = f.read()
    return _blob_cache


class Params(C.Structure):
    _fields_ = [('dt', C.c_double), ('gravity', C.c_double), ('solver_iters', C.c_int), ('erp', C.c_double),
                ('margin', C.c_double), ('motor_kp', C.c_double), ('motor_kd', C.c_double),
                ('motor_max_force', C.c_double), ('lin_damping', C.c_double), ('ang_damping', C.c_double),
                ('rest_threshold', C.c_double), ('use_urdf_inertia', C.c_int), ('edge_contacts', C.c_int),
                ('warmstart', C.c_double), ('no_rate_limit', C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, '_build', 'librr_oracle.so')
    if force or not os.path.exists(so) or not os.path.exists(so.replace('.so', '_f32.so')):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


_libs = {}


def _lib(f32=False):
    if f32 not in _libs:
        so = build()
        if f32:
            so = so.replace('.so', '_f32.so')
        L = C.CDLL(so)
        L.rro_create.restype = C.c_void_p
        L.rro_create.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(Params)]
        L.rro_destroy.argtypes = [C.c_void_p]
        L.rro_reset.argtypes = [C.c_void_p]
        L.rro_step.argtypes = [C.c_void_p, C.c_void_p]
        L.rro_step.restype = C.c_int
        L.rro_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rro_get_state.argtypes = [C.c_void_p, C.c_void_p]
        L.rro_set_state.argtypes = [C.c_void_p, C.c_void_p]
        L.rro_get_obs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rro_timestep.argtypes = [C.c_void_p]
        L.rro_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.rro_run.restype = C.c_int
        L.rro_link_pose.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rro_contacts.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rro_set_object_pose.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rro_set_contact_cache.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.rro_mass_matrix.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rro_set_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rro_default_params.argtypes = [C.POINTER(Params)]
        L.rro_solution_residual.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
        L.rro_solution_residual.restype = C.c_int
        L.rro_pair_contacts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.rro_pair_contacts.restype = C.c_int
        _libs[f32] = L
    return _libs[f32]


def params_from_solver(solver):
    """Oracle keyword arguments for a `solver=` dict of the product (real_robots_amd._native.SOLVER_DEFAULTS keys): the same
    constants under the oracle's names (rr_oracle.h rro_params)."""
    out = {}
    for k, v in (solver or {}).items():
        if k == 'ik_single_seed':
            continue                      # (planning only: oracle/kinematics.py generate_plan(single_seed=...))
        if k == 'rate_limit':
            out['no_rate_limit'] = 0 if v else 1
        elif k in ('lin_damping', 'ang_damping', 'erp', 'warmstart', 'motor_kp', 'motor_kd', 'motor_max_force'):
            out[k] = float(v)
        else:
            raise KeyError(k)
    return out


LINK_NAMES = open(os.path.join(_HERE, '..', 'real_robots_amd', 'data', 'realrobot_model_links.txt')).read().split()


class Oracle:
    """One REALRobot env on the CPU. f32=True uses the float build (same algorithm, fp32 arithmetic)."""

    def __init__(self, n_objects=3, width=320, height=240, f32=False, **params):
        self.L = _lib(f32)
        p = Params()
        self.L.rro_default_params(C.byref(p))
        for k, v in params.items():
            setattr(p, k, v)
        blob = model_blob()
        self.h = self.L.rro_create(blob, len(blob), n_objects, width, height, C.byref(p))
        assert self.h
        self.n_objects, self.W, self.H = n_objects, width, height

    def __del__(self):
        if getattr(self, 'h', None):
            self.L.rro_destroy(self.h)
            self.h = None

    def reset(self):
        self.L.rro_reset(self.h)

    def step(self, action9=None):
        if action9 is None:
            return self.L.rro_step(self.h, None)
        a = np.ascontiguousarray(action9, dtype=np.float64)
        assert a.shape == (9,)
        return self.L.rro_step(self.h, a.ctypes.data)

    def render(self):
        rgb = np.empty((self.H, self.W, 3), np.uint8)
        depth = np.empty((self.H, self.W), np.float32)
        mask = np.empty((self.H, self.W), np.int32)
        self.L.rro_render(self.h, rgb.ctypes.data, depth.ctypes.data, mask.ctypes.data)
        return rgb, depth, mask

    def set_camera(self, view, proj):
        """Row-major 4x4 OpenGL view / projection matrices replacing the eye camera (EnvCamera, env.py:470-513)."""
        v = np.ascontiguousarray(view, dtype=np.float32).reshape(16)
        p = np.ascontiguousarray(proj, dtype=np.float32).reshape(16)
        self.L.rro_set_camera(self.h, v.ctypes.data, p.ctypes.data)

    @property
    def state(self):
        s = np.empty(61)
        self.L.rro_get_state(self.h, s.ctypes.data)
        return s

    @state.setter
    def state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        assert s.shape == (61,)
        self.L.rro_set_state(self.h, s.ctypes.data)

    def obs(self):
        j, t, p = np.empty(9), np.empty(4), np.empty(3 * self.n_objects)
        self.L.rro_get_obs(self.h, j.ctypes.data, t.ctypes.data, p.ctypes.data)
        return j, t, p.reshape(self.n_objects, 3)

    def run(self, actions, state_every=0):
        """T steps in ONE library call (the GIL is released for all of it: oracles on a thread pool run in parallel): actions [T, 9]
        -> dict(ncontacts [T], nrobot [T] (contacts whose body A is a robot body), touch [T, 4], states [T // state_every, 61])."""
        a = np.ascontiguousarray(actions, dtype=np.float64)
        T = len(a)
        assert a.shape == (T, 9)
        nc, nr, touch = np.zeros(T, np.int32), np.zeros(T, np.int32), np.zeros((T, 4))
        states = np.zeros((T // state_every if state_every else 0, 61))
        n = self.L.rro_run(self.h, a.ctypes.data, T, nc.ctypes.data, nr.ctypes.data, touch.ctypes.data, int(state_every),
                           states.ctypes.data if state_every else None)
        assert n == len(states), n
        return dict(ncontacts=nc, nrobot=nr, touch=touch, states=states)

    @property
    def timestep(self):
        return self.L.rro_timestep(self.h)

    def link_pose(self, name_or_idx):
        i = LINK_NAMES.index(name_or_idx) if isinstance(name_or_idx, str) else name_or_idx
        p = np.empty(7)
        self.L.rro_link_pose(self.h, i, p.ctypes.data)
        return p

    def contacts(self):
        out = np.empty((48, 12))
        n = self.L.rro_contacts(self.h, out.ctypes.data, 48)
        return out[:n]

    def set_contact_cache(self, records):
        """Contact history for the warm start (records as returned by contacts()); call after setting the state."""
        r = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 12)
        self.L.rro_set_contact_cache(self.h, r.ctypes.data, len(r))

    def solution_residual(self, state_after, normal_forces, active_thresh=1.0):
        """Solver-independent check of a candidate solution of the LAST step's contact problem: post-step state (61) and the
        normal force of every contact -> dict(res_sum, res_max [N: what one more Gauss-Seidel update of a normal row would
        change], f_sum, f_max, n_active, active [bit mask of contacts with force > active_thresh])."""
        s = np.ascontiguousarray(state_after, dtype=np.float64)
        f = np.ascontiguousarray(normal_forces, dtype=np.float64)
        out, bits = np.zeros(5), np.zeros(1, np.uint64)
        n = self.L.rro_solution_residual(self.h, s.ctypes.data, f.ctypes.data, len(f), float(active_thresh), out.ctypes.data, bits.ctypes.data)
        assert n == len(f), "contact count differs from the oracle's last step"
        return dict(res_sum=out[0], res_max=out[1], f_sum=out[2], f_max=out[3], n_active=int(out[4]), active=int(bits[0]))

    def pair_contacts(self, shape_a, shape_b):
        """Narrow phase of one shape pair at the present state -> (contact records [n, 12], (Ra, pa), (Rb, pb))."""
        out, xf = np.empty((48, 12)), np.empty(24)
        n = self.L.rro_pair_contacts(self.h, shape_a, shape_b, out.ctypes.data, 48, xf.ctypes.data)
        assert n >= 0
        return out[:n], (xf[:9].reshape(3, 3), xf[9:12]), (xf[12:21].reshape(3, 3), xf[21:24])

    def set_object_pose(self, obj, pose7):
        p = np.ascontiguousarray(pose7, dtype=np.float64)
        self.L.rro_set_object_pose(self.h, obj, p.ctypes.data)

    def mass_matrix(self):
        M, b = np.empty((11, 11)), np.empty(11)
        self.L.rro_mass_matrix(self.h, M.ctypes.data, b.ctypes.data)
        return M, b
