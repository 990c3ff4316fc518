"""BatchedREALRobotEnv: N independent REALRobot envs stepped in lock-step on one MI355X.

Host-side mirror of the reference's step protocol for a batch (REALRobotEnv.step_joints, env.py:326-356;
Kuka.apply_action/calc_state/get_touch_sensors, robot.py:152-211), implemented by librealrobot_hip.so.
Observations stay on the device; `obs_host()` copies what the caller asks for.
"""
import ctypes as C

import numpy as np

from . import _native as nat

OBJECT_NAMES = ['cube', 'tomato', 'mustard']       # robot.py:49-50 (after "table")


class BatchedREALRobotEnv:
    def __init__(self, num_envs, objects=3, width=320, height=240, device=0, solver_iters=50, envs_per_block=0,
                 stream=None, use_urdf_inertia=False, dt=0.0, erp=0.0, margin=0.0, want_mask=True, solver=None):
        """solver: dict overriding the constants the reference leaves to pybullet's defaults (keys of `_native.SOLVER_DEFAULTS`:
        motor_kp 0.1, motor_kd 1.0, motor_max_force 100000, warmstart 0.85, lin_damping / ang_damping 0.04, erp 0.2, rate_limit
        True -- robot.py:196-201, env.py:202-204, 314-321)."""
        self.L = nat.load_library()
        cfg = nat.Config()
        cfg.abi_version = nat.RR_ABI_VERSION
        cfg.num_envs, cfg.n_objects, cfg.width, cfg.height = int(num_envs), int(objects), int(width), int(height)
        cfg.device, cfg.solver_iters, cfg.envs_per_block = int(device), int(solver_iters), int(envs_per_block)
        cfg.dt, cfg.erp, cfg.margin, cfg.use_urdf_inertia = dt, erp, margin, int(bool(use_urdf_inertia))
        cfg.flags = 0 if want_mask else nat.FLAG_NO_MASK
        nat.apply_solver(cfg, solver)
        self.solver = dict(nat.SOLVER_DEFAULTS, **({'erp': erp} if erp > 0 else {}), **(solver or {}))
        blob = nat.model_blob()
        h = C.c_void_p()
        self.h = None
        nat.check(self.L.rr_create(C.byref(cfg), blob, len(blob), C.c_void_p(stream or 0), C.byref(h)))
        self.h = h
        self.N, self.n_objects, self.W, self.H, self.device = int(num_envs), int(objects), int(width), int(height), int(device)
        self.object_names = OBJECT_NAMES[:self.n_objects]
        self._shapes = {
            nat.F_JOINTS: ((self.N, 9), np.float32), nat.F_TOUCH: ((self.N, 4), np.float32),
            nat.F_OBJ_POSE: ((self.N, self.n_objects, 7), np.float32),
            nat.F_RGB: ((self.N, self.H, self.W, 3), np.uint8), nat.F_DEPTH: ((self.N, self.H, self.W), np.float32),
            nat.F_MASK: ((self.N, self.H, self.W), np.int32), nat.F_TIMESTEP: ((self.N,), np.int32),
            nat.F_ERRFLAGS: ((self.N,), np.uint32), nat.F_STATE: ((self.N, 61), np.float32),
            nat.F_FRAG_COUNT: ((self.N, 1), np.uint32), nat.F_CONTACT_COUNT: ((self.N,), np.int32),
            nat.F_ENV_CLASS: ((self.N,), np.int32), nat.F_PREP: ((self.N, nat.PREP_FLOATS), np.float32)}
        p_, n_ = C.c_void_p(), C.c_size_t()                  # the tile count is the library's choice: ask for it
        nat.check(self.L.rr_get_buffer(self.h, nat.F_FRAG_COUNT, C.byref(p_), C.byref(n_)))
        self._shapes[nat.F_FRAG_COUNT] = ((self.N, max(1, n_.value // (4 * self.N))), np.uint32)

    def map_images(self, mask=True):
        """Pinned host copies of the images that every rendered step refreshes (rr_map_images; a handful of envs only): numpy views
        (rgb [N, H, W, 3] u8, depth [N, H, W] f32, mask [N, H, W] i32 or None) -- valid after `sync()`."""
        key = '_img_mirror_m' if mask else '_img_mirror'
        if getattr(self, key, None) is None:
            pr, pd, pm = C.c_void_p(), C.c_void_p(), C.c_void_p()
            nat.check(self.L.rr_map_images(self.h, C.byref(pr), C.byref(pd), C.byref(pm) if mask else None))
            npx = self.N * self.H * self.W
            rgb = np.frombuffer((C.c_uint8 * (npx * 3)).from_address(pr.value), dtype=np.uint8).reshape(self.N, self.H, self.W, 3)
            dep = np.frombuffer((C.c_float * npx).from_address(pd.value), dtype=np.float32).reshape(self.N, self.H, self.W)
            msk = np.frombuffer((C.c_int32 * npx).from_address(pm.value), dtype=np.int32).reshape(self.N, self.H, self.W) if mask else None
            setattr(self, key, (rgb, dep, msk))
        # (whoever asks for the views with the mask gets a mask block that is being refreshed: a block deselected by
        # select_image_mirror is selected again here -- and brought up to date at once by the library -- so that no holder of these
        # views reads a stale mask without an error)
        sel = getattr(self, '_img_sel', 7)
        if mask and not sel & 4:
            self.select_image_mirror(rgb=bool(sel & 1), depth=bool(sel & 2), mask=True)
        return getattr(self, key)

    def select_image_mirror(self, rgb=True, depth=True, mask=True):
        """Which mapped image blocks a rendered step refreshes (rr_select_image_mirror); a block selected again is brought up to
        date at once (valid after `sync_observations()`).  The selection is tracked here: `map_images(mask=True)` selects a
        deselected mask block again."""
        sel = (1 if rgb else 0) | (2 if depth else 0) | (4 if mask else 0)
        if sel != getattr(self, '_img_sel', 7):
            nat.check(self.L.rr_select_image_mirror(self.h, sel))
            self._img_sel = sel

    def close(self):
        self._mirror = self._img_mirror = self._img_mirror_m = None      # (views into memory the library frees)
        if getattr(self, 'h', None):
            self.L.rr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ stepping
    def reset(self, env_mask=None):
        if env_mask is None:
            nat.check(self.L.rr_reset(self.h, None))
        else:
            m = np.ascontiguousarray(env_mask, dtype=np.uint8)
            assert m.shape == (self.N,)
            nat.check(self.L.rr_reset(self.h, m.ctypes.data))

    def step(self, joint_cmd=None, render=False, device_ptr=None):
        """joint_cmd: None (zeros, env.py:333-334), float array [N, 9] on the host, or `device_ptr` (int address of
        an f32 [N, 9] device buffer). render: False / True / uint8 array [N] of per-env camera flags."""
        flags = None
        if isinstance(render, np.ndarray) and render.size > 1:
            flags = np.ascontiguousarray(render, dtype=np.uint8)
            assert flags.shape == (self.N,)
            mode = 2
        else:
            mode = 1 if np.any(render) else 0
        if device_ptr is not None:
            nat.check(self.L.rr_step(self.h, C.c_void_p(device_ptr), 1, mode, flags.ctypes.data if flags is not None else None))
            return
        if joint_cmd is None:
            nat.check(self.L.rr_step(self.h, None, 0, mode, flags.ctypes.data if flags is not None else None))
            return
        a = np.ascontiguousarray(joint_cmd, dtype=np.float32)
        assert a.shape == (self.N, 9), "joint_command must have shape [N, 9]"      # robot.py:190
        assert np.isfinite(a).all(), "joint_command must be finite"               # robot.py:189
        nat.check(self.L.rr_step(self.h, a.ctypes.data, 0, mode, flags.ctypes.data if flags is not None else None))

    def render(self):
        nat.check(self.L.rr_render(self.h))

    def sync(self):
        nat.check(self.L.rr_sync(self.h))

    def sync_observations(self):
        """Waits for the mapped observation blocks of the last step only (rr_sync_observations)."""
        nat.check(self.L.rr_sync_observations(self.h))

    def map_observations(self):
        """Host mirror of the low-dimensional observations (rr_map_observations): numpy views over pinned host memory that every
        step refreshes -- valid after `sync()`.  Returns dict(joints [N, 9], touch [N, 4], obj_pose [N, n_obj, 7], timestep [N],
        errflags [N])."""
        if getattr(self, '_mirror', None) is None:
            p, n = C.c_void_p(), C.c_size_t()
            nat.check(self.L.rr_map_observations(self.h, C.byref(p), C.byref(n)))
            N, k = self.N, self.n_objects
            buf = (C.c_float * (n.value // 4)).from_address(p.value)
            f = np.frombuffer(buf, dtype=np.float32)
            o = 0
            out = {}
            for name, shape in (('joints', (N, 9)), ('touch', (N, 4)), ('obj_pose', (N, k, 7))):
                sz = int(np.prod(shape))
                out[name] = f[o:o + sz].reshape(shape)
                o += sz
            out['timestep'] = f[o:o + N].view(np.int32)
            out['errflags'] = f[o + N:o + 2 * N].view(np.uint32)
            self._mirror = out
        return self._mirror

    # ------------------------------------------------------------------ data access
    def host(self, field):
        shape, dt = self._shapes[field]
        out = np.empty(shape, dt)
        nat.check(self.L.rr_copy_to_host(self.h, field, out.ctypes.data, out.nbytes))
        return out

    def device_buffer(self, field):
        shape, dt = self._shapes[field]
        p, n = C.c_void_p(), C.c_size_t()
        nat.check(self.L.rr_get_buffer(self.h, field, C.byref(p), C.byref(n)))
        return nat.DeviceBuffer(p.value, shape, np.dtype(dt).str, self)

    @property
    def state(self):
        return self.host(nat.F_STATE)

    @state.setter
    def state(self, s):
        s = np.ascontiguousarray(s, dtype=np.float32)
        assert s.shape == (self.N, 61)
        nat.check(self.L.rr_set_state(self.h, s.ctypes.data))

    def checkpoint(self):
        """Opaque snapshot (numpy uint8 array) of everything a later `restore` needs to continue bit for bit: state with motor
        targets, contact history of the warm start, episode clocks, error flags, touch sensors, object home poses."""
        n = C.c_size_t()
        nat.check(self.L.rr_checkpoint_bytes(self.h, C.byref(n)))
        buf = np.empty(n.value, np.uint8)
        nat.check(self.L.rr_checkpoint_save(self.h, buf.ctypes.data, buf.nbytes))
        return buf

    def restore(self, ckpt):
        buf = np.ascontiguousarray(ckpt, dtype=np.uint8)
        nat.check(self.L.rr_checkpoint_restore(self.h, buf.ctypes.data, buf.nbytes))

    def set_object_pose(self, env, obj, pose7):
        p = np.ascontiguousarray(pose7, dtype=np.float32)
        assert p.shape == (7,)
        nat.check(self.L.rr_set_object_pose(self.h, int(env), int(obj), p.ctypes.data))

    def set_object_poses(self, poses, env_mask=None):
        """Teleports the objects of the masked envs (None: all): poses [N, n_objects, 7] (xyz + xyzw quaternion),
        velocities zeroed -- one upload for the whole batch."""
        p = np.ascontiguousarray(poses, dtype=np.float32)
        assert p.shape == (self.N, self.n_objects, 7)
        m = None
        if env_mask is not None:
            m = np.ascontiguousarray(env_mask, dtype=np.uint8)
            assert m.shape == (self.N,)
        nat.check(self.L.rr_set_object_poses(self.h, p.ctypes.data, m.ctypes.data if m is not None else None))

    def set_object_home(self, env, obj, pose7):
        """Pose object `obj` of env `env` (None: every env) returns to on reset / when it leaves the table
        (Kuka.object_poses, robot.py:19-24)."""
        p = np.ascontiguousarray(pose7, dtype=np.float32)
        assert p.shape == (7,)
        nat.check(self.L.rr_set_object_home(self.h, -1 if env is None else int(env), int(obj), p.ctypes.data))

    def evaluate_goals(self, goal_pos, goal_mask=None):
        """REALRobotEnv.evaluateGoal (env.py:181-200) for the whole batch, on the device: goal_pos [N, n_objects, 3], goal_mask
        [N, n_objects] (None: every object counts) -> scores float32 [N]."""
        g = np.ascontiguousarray(goal_pos, dtype=np.float32)
        assert g.shape == (self.N, self.n_objects, 3)
        m = None
        if goal_mask is not None:
            m = np.ascontiguousarray(goal_mask, dtype=np.uint8)
            assert m.shape == (self.N, self.n_objects)
        out = np.empty(self.N, np.float32)
        nat.check(self.L.rr_evaluate_goals(self.h, g.ctypes.data, m.ctypes.data if m is not None else None, out.ctypes.data))
        return out

    def link_poses(self):
        out = np.empty((self.N, len(nat.LINK_NAMES), 7), np.float32)
        nat.check(self.L.rr_link_poses(self.h, out.ctypes.data))
        return out

    def contacts(self, env):
        out = np.empty((48, 12), np.float32)
        n = C.c_int32()
        nat.check(self.L.rr_get_contacts(self.h, int(env), out.ctypes.data, 48, C.byref(n)))
        return out[:n.value]

    # ------------------------------------------------------------------ IK / macro plans (K8)
    def ik(self, targets):
        """Batched DLS IK of the gripper base from every env's current joints. targets [N, 7] (xyz + xyzw quat).
        Returns (q [N, 11], residual [N])."""
        t = np.ascontiguousarray(targets, dtype=np.float32)
        assert t.shape == (self.N, 7)
        q = np.empty((self.N, 11), np.float32)
        err = np.empty(self.N, np.float32)
        nat.check(self.L.rr_ik(self.h, t.ctypes.data, q.ctypes.data, err.ctypes.data))
        return q, err

    def plan_macro(self, macro_actions, env_mask=None):
        """Builds the 1000-step plans of macro actions [N, 2, 2] on the device (env.py:388-454)."""
        m = np.ascontiguousarray(macro_actions, dtype=np.float32).reshape(self.N, 4)
        mk = None
        if env_mask is not None:
            mk = np.ascontiguousarray(env_mask, dtype=np.uint8)
            assert mk.shape == (self.N,)
        nat.check(self.L.rr_plan_macro(self.h, m.ctypes.data, mk.ctypes.data if mk is not None else None))

    def get_plan(self, env):
        out = np.empty((1000, 9), np.float32)
        nat.check(self.L.rr_get_plan(self.h, int(env), out.ctypes.data))
        return out

    def step_plan(self, render=False, idle=None):
        """Every env consumes the next row of its plan; envs flagged in `idle` (uint8 [N]) take zeros(9) instead and
        keep their place (macro_action None, env.py:391-393)."""
        flags = None
        if isinstance(render, np.ndarray) and render.size > 1:
            flags = np.ascontiguousarray(render, dtype=np.uint8)
            mode = 2
        else:
            mode = 1 if np.any(render) else 0
        if idle is not None:
            idle = np.ascontiguousarray(idle, dtype=np.uint8)
            assert idle.shape == (self.N,)
        nat.check(self.L.rr_step_plan_masked(self.h, idle.ctypes.data if idle is not None else None, mode,
                                             flags.ctypes.data if flags is not None else None))

    def step_macro(self, macro_actions, render=False):
        """Batched REALRobotEnv.step_macro (env.py:388-412): a new macro action (or an exhausted plan) triggers
        re-planning for that env; every env then consumes the next row of its plan.  `macro_actions` is an array
        [N, 2, 2] or a sequence whose entries may be None (that env steps with zeros(9), env.py:391-393)."""
        none = np.array([a is None for a in macro_actions], dtype=bool) if not isinstance(macro_actions, np.ndarray) \
            else np.zeros(self.N, bool)
        if none.any():
            macro_actions = [np.zeros((2, 2)) if a is None else a for a in macro_actions]
        m = np.ascontiguousarray(macro_actions, dtype=np.float64).reshape(self.N, 4)
        if not hasattr(self, '_macro_req'):
            self._macro_req = np.full((self.N, 4), np.nan)
            self._macro_step = np.zeros(self.N, np.int64)
        need = ((~np.all(m == self._macro_req, axis=1)) | (self._macro_step >= 1000)) & ~none
        if need.any():
            if not hasattr(self, '_macro_any'):       # the plan buffers exist after the first plan_macro
                self._macro_any = True
            self.plan_macro(m, need.astype(np.uint8))
            self._macro_req[need] = m[need]
            self._macro_step[need] = 0
        if none.all() and not hasattr(self, '_macro_any'):
            self.step(None, render=render)             # nobody has a plan yet: plain zeros step
            return
        self.step_plan(render, idle=none.astype(np.uint8) if none.any() else None)
        self._macro_step[~none] += 1

    def set_camera(self, view, proj):
        """Row-major 4x4 OpenGL view / projection matrices replacing the eye camera of this batch (None, None: the default eye)."""
        if view is None and proj is None:
            nat.check(self.L.rr_set_camera(self.h, None, None))
            return
        v = np.ascontiguousarray(view, dtype=np.float32).reshape(16)
        p = np.ascontiguousarray(proj, dtype=np.float32).reshape(16)
        nat.check(self.L.rr_set_camera(self.h, v.ctypes.data, p.ctypes.data))

    def set_timing(self, on):
        nat.check(self.L.rr_set_timing(self.h, int(on)))

    def get_timing(self):
        ms = np.zeros(nat.NUM_KERNELS, np.float32)
        n = np.zeros(nat.NUM_KERNELS, np.int32)
        nat.check(self.L.rr_get_timing(self.h, ms.ctypes.data, n.ctypes.data))
        return dict(zip(nat.KERNEL_NAMES, zip(ms.tolist(), n.tolist())))
