"""ctypes binding of librealrobot_hip.so (C ABI in include/realrobot.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is present, creation fails
loudly (RuntimeError) -- nothing here imports the oracle.
"""
import ctypes as C
import gzip
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RR_LIB', os.path.join(_HERE, 'csrc', 'librealrobot_hip.so'))
BLOB_GZ = os.environ.get('RR_MODEL', os.path.join(_HERE, 'data', 'realrobot_model.bin.gz'))      # (RR_MODEL: another compiled model, A/B)
LINK_NAMES = open(os.path.join(_HERE, 'data', 'realrobot_model_links.txt')).read().split()

RR_ABI_VERSION = 6
(F_JOINTS, F_TOUCH, F_OBJ_POSE, F_RGB, F_DEPTH, F_MASK, F_TIMESTEP, F_ERRFLAGS, F_STATE, F_FRAG_COUNT, F_CONTACT_COUNT,
 F_ENV_CLASS, F_PREP) = range(13)
PREP_FLOATS = 378          # RR_F_PREP: frames 165, M^-1 121, qd* 11, object terms 81 -- in this order (realrobot.hip S_*)
NUM_KERNELS = 9
# id 5 = image set-up outside the two render kernels: the full static copy of the first frame (and the earlier schemes
# RR_FULL_COPY / RR_SEPARATE_RESTORE); it does not run in steady state
KERNEL_NAMES = ('k_prep', 'k_collide', 'k_solve', 'k_render_setup', 'k_raster', 'k_image_setup', 'k_shade',
                'k_solve_heavy', 'render_heavy')      # 7, 8: the heavy envs' solve / render (side streams in an untimed step);
                                                      # k_prep / k_collide = the look-ahead of the next step

# every symbol include/realrobot.h declares (tests check the library exports all of them)
SYMBOLS = ('rr_create', 'rr_destroy', 'rr_set_stream', 'rr_reset', 'rr_set_object_pose', 'rr_set_object_home', 'rr_step', 'rr_render',
           'rr_get_buffer', 'rr_copy_to_host', 'rr_set_state', 'rr_sync', 'rr_link_poses', 'rr_get_contacts',
           'rr_set_timing', 'rr_get_timing', 'rr_last_error', 'rr_abi_version', 'rr_ik', 'rr_plan_macro', 'rr_get_plan',
           'rr_step_plan', 'rr_set_camera', 'rr_set_object_poses', 'rr_step_plan_masked', 'rr_checkpoint_bytes',
           'rr_checkpoint_save', 'rr_checkpoint_restore', 'rr_evaluate_goals', 'rr_device_microbench', 'rr_map_observations', 'rr_map_images', 'rr_sync_observations', 'rr_select_image_mirror',
           'rr_pack_image_delta', 'rr_apply_image_delta')


class Config(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('num_envs', C.c_int32), ('n_objects', C.c_int32),
                ('width', C.c_int32), ('height', C.c_int32), ('device', C.c_int32), ('solver_iters', C.c_int32),
                ('envs_per_block', C.c_int32), ('dt', C.c_float), ('erp', C.c_float), ('margin', C.c_float),
                ('use_urdf_inertia', C.c_int32), ('flags', C.c_int32),
                # the constants the reference leaves to pybullet's defaults (include/realrobot.h: 0 -> default, < 0 -> literal zero)
                ('motor_kp', C.c_float), ('motor_kd', C.c_float), ('motor_max_force', C.c_float), ('warmstart', C.c_float),
                ('lin_damping', C.c_float), ('ang_damping', C.c_float), ('solver_flags', C.c_int32)]


FLAG_NO_MASK = 1
SOLVER_NO_RATE_LIMIT = 1
SOLVER_IK_SINGLE_SEED = 2
# `solver=` of BatchedREALRobotEnv / make(): name -> documented default (UPSTREAM, unverifiable here)
SOLVER_DEFAULTS = {'motor_kp': 0.1, 'motor_kd': 1.0, 'motor_max_force': 100000.0, 'warmstart': 0.85, 'lin_damping': 0.04,
                   'ang_damping': 0.04, 'erp': 0.2, 'rate_limit': True, 'ik_single_seed': False}


def apply_solver(cfg, solver):
    """Writes a `solver` dict (keys of SOLVER_DEFAULTS; None / missing: the default) into an rr_config.  A value of exactly 0 is
    passed as the header's "literal zero" (a negative number); unknown keys and non-finite values raise ValueError."""
    for k, v in (solver or {}).items():
        if k not in SOLVER_DEFAULTS:
            raise ValueError("unknown solver parameter %r (known: %s)" % (k, ', '.join(sorted(SOLVER_DEFAULTS))))
        if k == 'rate_limit':
            cfg.solver_flags = (cfg.solver_flags & ~SOLVER_NO_RATE_LIMIT) | (0 if v else SOLVER_NO_RATE_LIMIT)
            continue
        if k == 'ik_single_seed':
            cfg.solver_flags = (cfg.solver_flags & ~SOLVER_IK_SINGLE_SEED) | (SOLVER_IK_SINGLE_SEED if v else 0)
            continue
        v = float(v)
        if not np.isfinite(v) or v < 0:
            raise ValueError("solver parameter %s must be finite and >= 0" % k)
        if k == 'erp' and v == 0:
            raise ValueError("solver parameter erp must be > 0")
        setattr(cfg, k, v if v > 0 else -1.0)


_lib = None
_blob = None


def model_blob():
    global _blob
    if _blob is None:
        with gzip.open(BLOB_GZ, 'rb') as f:
            _blob = f.read()
    return _blob


def load_library():
    """Loads librealrobot_hip.so; raises RuntimeError with a build hint when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "real_robots_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C real_robots_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime. If torch is going to be used in this process (zero-copy
    # views of the observation buffers, torch.distributed), its runtime stack has to be loaded before this library's
    # (/opt/rocm) one, otherwise torch later reports "No HIP GPUs are available". Importing torch first is harmless
    # when it is not needed and is skipped when it is not installed.
    if os.environ.get('RR_NO_TORCH_PRELOAD') is None:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int32
    L.rr_create.argtypes = [C.POINTER(Config), C.c_char_p, C.c_size_t, vp, C.POINTER(vp)]
    L.rr_destroy.argtypes = [vp]
    L.rr_set_stream.argtypes = [vp, vp]
    L.rr_reset.argtypes = [vp, vp]
    L.rr_set_object_pose.argtypes = [vp, i32, i32, vp]
    L.rr_set_object_home.argtypes = [vp, i32, i32, vp]
    L.rr_step.argtypes = [vp, vp, i32, i32, vp]
    L.rr_render.argtypes = [vp]
    L.rr_get_buffer.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.rr_copy_to_host.argtypes = [vp, i32, vp, C.c_size_t]
    L.rr_set_state.argtypes = [vp, vp]
    L.rr_sync.argtypes = [vp]
    L.rr_link_poses.argtypes = [vp, vp]
    L.rr_get_contacts.argtypes = [vp, i32, vp, i32, C.POINTER(i32)]
    L.rr_set_timing.argtypes = [vp, i32]
    L.rr_get_timing.argtypes = [vp, vp, vp]
    L.rr_ik.argtypes = [vp, vp, vp, vp]
    L.rr_plan_macro.argtypes = [vp, vp, vp]
    L.rr_get_plan.argtypes = [vp, i32, vp]
    L.rr_step_plan.argtypes = [vp, i32, vp]
    L.rr_set_camera.argtypes = [vp, vp, vp]
    L.rr_set_object_poses.argtypes = [vp, vp, vp]
    L.rr_step_plan_masked.argtypes = [vp, vp, i32, vp]
    L.rr_evaluate_goals.argtypes = [vp, vp, vp, vp]
    L.rr_device_microbench.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    L.rr_map_observations.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.rr_sync_observations.argtypes = [vp]
    L.rr_map_images.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.rr_select_image_mirror.argtypes = [vp, C.c_int32]
    L.rr_pack_image_delta.argtypes = [vp, vp, vp, C.c_uint32]
    L.rr_apply_image_delta.argtypes = [vp, vp, i32, C.c_uint32, C.c_size_t, vp, vp, vp]
    L.rr_checkpoint_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.rr_checkpoint_save.argtypes = [vp, vp, C.c_size_t]
    L.rr_checkpoint_restore.argtypes = [vp, vp, C.c_size_t]
    L.rr_last_error.restype = C.c_char_p
    L.rr_abi_version.restype = i32
    for name in SYMBOLS:
        if name not in ('rr_last_error', 'rr_abi_version'):
            getattr(L, name).restype = i32
    if L.rr_abi_version() != RR_ABI_VERSION:
        raise RuntimeError("real_robots_amd: ABI version mismatch, rebuild the HIP library")
    _lib = L
    return L


class NativeError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise NativeError("librealrobot_hip: error %d: %s" % (rc, load_library().rr_last_error().decode()))


class DeviceBuffer:
    """Device memory view exposing __cuda_array_interface__ (zero-copy into torch: torch.as_tensor(buf, device=...))."""

    def __init__(self, ptr, shape, typestr, owner):
        self.ptr, self.shape, self.typestr, self._owner = ptr, tuple(shape), typestr, owner

    @property
    def __cuda_array_interface__(self):
        return {'shape': self.shape, 'typestr': self.typestr, 'data': (self.ptr, False), 'version': 2, 'strides': None}


# ---------------------------------------------------------------------------------------------- DLPack export
class _DLDevice(C.Structure):
    _fields_ = [('device_type', C.c_int32), ('device_id', C.c_int32)]


class _DLDataType(C.Structure):
    _fields_ = [('code', C.c_uint8), ('bits', C.c_uint8), ('lanes', C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [('data', C.c_void_p), ('device', _DLDevice), ('ndim', C.c_int32), ('dtype', _DLDataType),
                ('shape', C.POINTER(C.c_int64)), ('strides', C.POINTER(C.c_int64)), ('byte_offset', C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DLDeleter = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [('dl_tensor', _DLTensor), ('manager_ctx', C.c_void_p), ('deleter', _DLDeleter)]
KDL_ROCM = 10                       # DLDeviceType kDLROCM
_dl_alive = {}                      # address of a DLManagedTensor handed out -> (struct, shape array, owner)


@_DLDeleter
def _dl_deleter(ptr):
    _dl_alive.pop(C.addressof(ptr.contents), None)


_PyCapsuleDestructor = C.CFUNCTYPE(None, C.c_void_p)


@_PyCapsuleDestructor
def _dl_capsule_destructor(capsule):
    # a capsule nobody consumed is still called "dltensor": its tensor is ours to release
    api = C.pythonapi
    api.PyCapsule_IsValid.argtypes, api.PyCapsule_IsValid.restype = [C.c_void_p, C.c_char_p], C.c_int
    api.PyCapsule_GetPointer.argtypes, api.PyCapsule_GetPointer.restype = [C.c_void_p, C.c_char_p], C.c_void_p
    if api.PyCapsule_IsValid(capsule, b'dltensor'):
        _dl_alive.pop(api.PyCapsule_GetPointer(capsule, b'dltensor'), None)


def _dlpack_capsule(ptr, shape, np_dtype, device_id, owner):
    dt = np.dtype(np_dtype)
    code = {'f': 2, 'i': 0, 'u': 1}[dt.kind]
    m = _DLManagedTensor()
    shp = (C.c_int64 * len(shape))(*shape)
    m.dl_tensor.data = ptr
    m.dl_tensor.device = _DLDevice(KDL_ROCM, device_id)
    m.dl_tensor.ndim = len(shape)
    m.dl_tensor.dtype = _DLDataType(code, dt.itemsize * 8, 1)
    m.dl_tensor.shape = shp
    m.dl_tensor.strides = None          # compact row-major
    m.dl_tensor.byte_offset = 0
    m.manager_ctx = None
    m.deleter = _dl_deleter
    _dl_alive[C.addressof(m)] = (m, shp, owner)
    api = C.pythonapi
    api.PyCapsule_New.argtypes, api.PyCapsule_New.restype = [C.c_void_p, C.c_char_p, _PyCapsuleDestructor], C.py_object
    return api.PyCapsule_New(C.addressof(m), b'dltensor', _dl_capsule_destructor)


def _dlpack(self, stream=None, **kwargs):
    """DLPack export of a device buffer (zero-copy into torch / cupy / jax on ROCm).  The buffer belongs to the env handle
    (kept alive by the capsule) and is rewritten by every step on the library's stream: `stream` is accepted for protocol
    compatibility, ordering follows the stream contract of include/realrobot.h."""
    return _dlpack_capsule(self.ptr, self.shape, self.typestr, getattr(self._owner, 'device', 0), self._owner)


DeviceBuffer.__dlpack__ = _dlpack
DeviceBuffer.__dlpack_device__ = lambda self: (KDL_ROCM, getattr(self._owner, 'device', 0))


def device_microbench(device=0):
    """HBM copy / triad bandwidth (GB/s) and the VALU issue rate of a sample-test-like mix at eight waves per SIMD
    (G wave64-instructions/s), measured on `device` by the library (rr_device_microbench)."""
    L = load_library()
    out = {}
    for kind, name in ((0, 'hbm_copy_GBs'), (1, 'hbm_triad_GBs'), (2, 'valu_mix_G_wave_instr_s')):
        r = C.c_double()
        check(L.rr_device_microbench(int(device), kind, C.byref(r)))
        out[name] = round(r.value, 1)
    return out
