/* realrobot.h -- C ABI of librealrobot_hip.so: batched REALRobot env.step() on MI355X (gfx950).
 *
 * Drop-in boundary.  The reference has no FFI for this path: its boundary is the duck-typed gym API
 * `gym.make(id) -> REALRobotEnv` with reset()/step()/render() (real_robots/__init__.py:22-28,
 * real_robots/envs/env.py:206-219,326-356) and, underneath, ~50 pybullet C-API calls per step
 * (SURVEY.md 3.3).  This header is what a maintainer binds instead of `import pybullet` for the step path;
 * every entry point cites the reference call(s) it replaces.  The ctypes binding is
 * real_robots_amd/_native.py; INTEGRATION.md shows the stub to add to the reference.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returns 0 on success or a
 * negative RR_E* code and never throws/aborts; rr_last_error() gives the message of the last failure on the
 * calling thread.  The library owns all device memory; the caller owns host buffers.  One rr_env may be
 * used from one thread at a time.  Work is enqueued on the stream given at creation (or rr_set_stream) and
 * is asynchronous until rr_sync / rr_copy_to_host.
 * Layouts: every buffer is row-major with the env index outermost ([N, ...]).
 */
#ifndef REALROBOT_H
#define REALROBOT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RR_ABI_VERSION 6   /* 3: RR_F_CONTACT_COUNT, RR_F_ENV_CLASS, rr_checkpoint_*, rr_evaluate_goals; 4: rr_map_observations, rr_map_images,
                              rr_sync_observations, rr_device_microbench; checkpoint blobs carry the step parameters (version 2 header);
                              5: rr_select_image_mirror; 6: rr_config carries the motor / solver constants the reference leaves to
                              pybullet's defaults (motor_kp .. solver_flags, in the place of reserved[7]: same struct size); checkpoint
                              header version 3 carries them too; RR_F_PREP; rr_pack_image_delta, rr_apply_image_delta */

enum {
    RR_OK = 0,
    RR_EINVAL = -1,     /* bad argument */
    RR_EDEVICE = -2,    /* HIP runtime error (no GPU, out of memory, launch failure) */
    RR_EMODEL = -3,     /* malformed model blob */
    RR_EACTION = -4     /* non-finite action (the reference asserts, robot.py:189) */
};

/* fields of rr_get_buffer / rr_copy_to_host */
enum {
    RR_F_JOINTS = 0,    /* f32 [N, 9]        Kuka.calc_state()            robot.py:203-211 */
    RR_F_TOUCH = 1,     /* f32 [N, 4]        Kuka.get_touch_sensors()     robot.py:152-163 */
    RR_F_OBJ_POSE = 2,  /* f32 [N, n_obj, 7] object_bodies[..].get_pose() env.py:236-244 (xyz + xyzw quat) */
    RR_F_RGB = 3,       /* u8  [N, H, W, 3]  retina                       env.py:249-255,560-562 */
    RR_F_DEPTH = 4,     /* f32 [N, H, W]     GL depth in [0,1]            env.py:564-565 */
    RR_F_MASK = 5,      /* i32 [N, H, W]     body unique id, -1 background  env.py:552-558 */
    RR_F_TIMESTEP = 6,  /* i32 [N]           env.timestep                 env.py:217,346 */
    RR_F_ERRFLAGS = 7,  /* u32 [N]           1: non-finite state detected (env frozen until reset / set_state); 2: this step's command was
                                             not finite (env not stepped, robot.py:189: its state, clock and observations stay as they
                                             are; its contact list is dropped -- RR_F_CONTACT_COUNT 0, rr_get_contacts empty -- and the
                                             next accepted step starts its contact solve cold, like a step after rr_set_state);
                                             4: internal consistency of the solver (never expected; the env stops stepping until reset / set_state);
                                             8: RENDER status only -- more than 2048 near-plane-crossing triangles met one raster tile of the last
                                             rendered frame and the surplus was dropped (a camera inside a mesh); the physics ignores this bit,
                                             it stays set until rr_reset / rr_set_state of the env */
    RR_F_STATE = 8,     /* f32 [N, 61]       q[11] qd[11] 3x(pos3 quat4 lin3 ang3)  (checkpoint / parity) */
    RR_F_FRAG_COUNT = 9,/* u32 [N, tiles]    diagnostic: entries of k_shade's work list in the last render (pixels won by moving geometry + pixels vacated since the frame before) */
    RR_F_CONTACT_COUNT = 10, /* i32 [N]      contacts of the last solved step (rr_get_contacts returns them one env at a time); 0 after
                                             rr_reset / rr_set_state of that env */
    RR_F_ENV_CLASS = 11,     /* i32 [N]      diagnostic: 0 light, 1 heavy, 2 very heavy -- which launch solved / rendered the env in the last
                                             step.  Both fields live in fixed buffers written by the solve kernels: a pointer
                                             from rr_get_buffer stays valid over steps like every other field's */
    RR_F_PREP = 12,          /* f32 [N, 378]  diagnostic: the preparation's record of every env, as the last preparation launch left it -- frames of the
                                             eleven bodies (R 99, position 33, joint axis 33), M^-1 (121), unconstrained joint velocities (11), the objects'
                                             rotation / world inverse inertia / unconstrained velocities / collision position (27 + 27 + 9 + 9 + 9); after a
                                             step with the look-ahead it describes the state the step LEFT (tests compare the kernel's forms through it) */
    RR_F_COUNT = 13
};

/* rr_config.flags */
#define RR_FLAG_NO_MASK 1   /* R2 environments have no `mask` observation (robot.py:99-112): do not produce RR_F_MASK */

typedef struct rr_config {
    int32_t abi_version;    /* RR_ABI_VERSION */
    int32_t num_envs;       /* N envs on this device */
    int32_t n_objects;      /* 1..3: cube, tomato, mustard      robot.py:49-50 */
    int32_t width, height;  /* eye camera; reference default 320x240 (robot.py:30-31).  width: a multiple of 4 in [4, 1024], height in
                               [1, 1024], at most 255 raster tiles of 4096 pixels (1024 x 1020 fits, 1024 x 1024 does not): RR_EINVAL otherwise */
    int32_t device;         /* HIP device ordinal */
    int32_t solver_iters;   /* PGS iterations; <=0 -> 50        SURVEY A.1.2 */
    int32_t envs_per_block; /* physics kernels: envs (threads) per workgroup; <=0 -> default */
    float dt;               /* <=0 -> 0.005                     env.py:203-204 */
    float erp;              /* <=0 -> 0.2 */
    float margin;           /* <=0 -> 0.02 */
    int32_t use_urdf_inertia; /* 0: Bullet AABB inertia for robot links (default); 1: URDF <inertia> */
    int32_t flags;          /* RR_FLAG_* */
    /* The constants below are NOT in the reference tree: robot.py:196-201 calls Joint.set_position -> setJointMotorControl2(
     * POSITION_CONTROL, targetPosition) and leaves positionGain / velocityGain / force to pybullet's defaults, env.py:202-204 leaves
     * the solver to Bullet's (unverifiable here).  They are parameters of the handle, reach every
     * kernel as scalar arguments and are part of a checkpoint's header; 0 selects the documented default, a NEGATIVE value a
     * literal zero (gain off / cold start / no damping).  The reference's own tracking script (tests/test_actions.py:62-71,
     * 147-152) is met at every check point by motor_kp >= 0.5 and not by 0.1 (tests/golden/macro_sensitivity.json). */
    float motor_kp;         /* 0 -> 0.1      positionGain: v_target = kp (target - q) / dt + (1 - kd) qd */
    float motor_kd;         /* 0 -> 1.0      velocityGain */
    float motor_max_force;  /* 0 -> 100000   |motor impulse| <= force dt */
    float warmstart;        /* 0 -> 0.85     Bullet's m_warmstartingFactor on the matched normal impulses; < 0: cold start every step */
    float lin_damping;      /* 0 -> 0.04     btMultiBody base damping of the free objects; < 0: none */
    float ang_damping;      /* 0 -> 0.04 */
    int32_t solver_flags;   /* RR_SOLVER_* */
} rr_config;
#define RR_SOLVER_NO_RATE_LIMIT 1   /* limitActionByJoint (env.py:314-321) is skipped: the clipped command itself is the motor target */
#define RR_SOLVER_IK_SINGLE_SEED 2  /* rr_ik / rr_plan_macro: ONE damped-least-squares solve per target, seeded with the env's current joints --
                                       the literal call pattern of the reference (env.py:372-375, 421-427: one calculateInverseKinematics per
                                       way point, no stepping in between).  Default (0): the best of several seeds -- current joints, an
                                       elbow-up posture, the previous way point -- by convergence, then elbow height / continuity: which branch
                                       pybullet's own solver lands on from one seed is not specified by the reference */

typedef struct rr_env rr_env;

/* Replaces REALRobotEnv.__init__ + the lazy bullet client/world creation in reset()
 * (env.py:36-122,202-219; robot.py:44-118,165-185: loadURDF of robot + table + objects).
 * `model_blob` is the compiled model (tools/compile_model.py). `stream` is a hipStream_t or NULL (default stream). */
int rr_create(const rr_config *cfg, const void *model_blob, size_t blob_bytes, void *stream, rr_env **out);
int rr_destroy(rr_env *env);
int rr_set_stream(rr_env *env, void *stream);

/* Replaces env.reset() (env.py:206-219; robot.py:120-129,165-185) for the envs whose mask byte is non-zero
 * (env_mask == NULL: all envs). Device-side state copy from the template; does not render. */
int rr_reset(rr_env *env, const uint8_t *env_mask_host);

/* Replaces BodyPart.reset_pose via robot.object_bodies[name].reset_pose (env.py:159-162; zeroes velocity). */
int rr_set_object_pose(rr_env *env, int32_t env_index, int32_t obj, const float *pose7);
/* Batched rr_set_object_pose: poses f32 [N, n_obj, 7] (host) for the envs whose mask byte is non-zero (NULL: all envs).
 * Replaces the per-object loop of REALRobotEnv.set_goal (env.py:159-162) for a whole batch with one upload. */
int rr_set_object_poses(rr_env *env, const float *poses_host, const uint8_t *env_mask_host);
/* The pose (xyz + xyzw quaternion, host) object `obj` of env `env_index` (< 0: every env) returns to on rr_reset and when the
 * out-of-bounds rule fires (env.py:257-264). Replaces in-place edits of Kuka.object_poses (robot.py:19-24; the reference's
 * tests/test_actions.py:95-98 parks the objects on the shelf that way). Defaults: the poses of the model blob. */
int rr_set_object_home(rr_env *env, int32_t env_index, int32_t obj, const float *pose7);

/* Replaces one REALRobotEnv.step_joints() (env.py:326-356) for all N envs:
 *   limitActionByJoint (env.py:314-321), control_objects_limits (env.py:257-264), Kuka.apply_action
 *   (robot.py:188-201), scene.global_step() -> stepSimulation (env.py:340), calc_state/get_touch_sensors
 *   (robot.py:203-211,152-163) and, when render_mode != 0, get_retina (env.py:249-255).
 * joint_cmd: f32 [N, 9] (device pointer if cmd_on_device, else host; NULL -> zeros as env.py:333-334).
 *   Host commands / flags are copied into a pinned staging ring before the call returns (the caller may reuse its
 *   buffer at once; the call does not wait for the device).
 *   STREAM CONTRACT for cmd_on_device: the buffer is read IN PLACE by the first kernel of the step, on the library's
 *   stream (rr_create / rr_set_stream).  The caller must (1) have produced it on that same stream, or have made that
 *   stream wait for the producer (event / synchronise), and (2) not overwrite it before the step has consumed it --
 *   i.e. not before later work on the same stream, or rr_sync.  The zero-copy views of rr_get_buffer carry no stream
 *   either: readers on another stream must order themselves after the step the same way.
 * render_mode: 0 none, 1 all envs, 2 per-env flags in render_flags_host (u8 [N]). */
int rr_step(rr_env *env, const float *joint_cmd, int32_t cmd_on_device, int32_t render_mode,
            const uint8_t *render_flags_host);

/* Replaces EyeCamera.render (env.py:536-567) for all envs at the current state (used by reset()/set_goal()). */
int rr_render(rr_env *env);

/* Replaces the camera of this env handle (row-major 4x4 OpenGL view and projection matrices, host). The default is the
 * reference's eye camera; the facade uses a second env handle with EnvCamera's matrices for render('rgb_array')
 * (computeViewMatrixFromYawPitchRoll / computeProjectionMatrixFOV, env.py:480-499).  Both pointers NULL: back to the default
 * eye camera (eye (0.01, 0, 1.2) -> table position, up (0, 0, 1), fov 80, near 0.1, far 100; env.py:136-141, 253-255, 548-551). */
int rr_set_camera(rr_env *env, const float *view16, const float *proj16);

/* Device pointer + size of an observation/state buffer (valid until rr_destroy). */
int rr_get_buffer(rr_env *env, int32_t field, void **dev_ptr, size_t *bytes);
/* The buffers are owned by the library and READ-ONLY for the caller: observations are rewritten by every step, and the
 * image buffers (RGB, DEPTH, MASK) persist from frame to frame -- a render only rewrites the pixels that differ from the
 * previous frame of that env, so a caller that scribbles into them would see its marks survive. */
/* Synchronising copy of a whole field to host memory. */
int rr_copy_to_host(rr_env *env, int32_t field, void *dst, size_t bytes);
/* Overwrites the simulation state from host memory (f32 [N, 61]); parity tests, goal set-up.  The contact history of the warm
 * start (the previous step's contact list, see rr_get_contacts) is not part of the 61 floats: the step after rr_set_state /
 * rr_reset starts cold, as after pybullet's resetSimulation.  To continue a run exactly, use rr_checkpoint_save / _restore.
 * (rr_set_object_pose(s) keeps the history, like resetBasePositionAndOrientation keeps Bullet's manifolds: cached points of a
 * teleported body are farther than the contact margin from its new contacts and match nothing.) */
int rr_set_state(rr_env *env, const float *state_host);
/* Checkpoint = everything a restore needs to continue BIT FOR BIT where the save left off: the state (with the motor targets),
 * the contact history of the warm start (contact list + normal forces of the last solved step -- Bullet's persistent manifolds
 * with their cached impulses, which pybullet.saveState / restoreState carry too), episode clocks, error flags, touch sensors and
 * the per-env object home poses.  Opaque host blob of rr_checkpoint_bytes() bytes, valid for env handles of the same num_envs /
 * n_objects.  (Macro plans in flight are host-side policy state and not part of it.)  save + restore + step == step, tested. */
int rr_checkpoint_bytes(rr_env *env, size_t *bytes);
int rr_checkpoint_save(rr_env *env, void *dst_host, size_t bytes);
int rr_checkpoint_restore(rr_env *env, const void *src_host, size_t bytes);
int rr_sync(rr_env *env);
/* Host mirror of the low-dimensional observations, for callers that read them on the host after every step (the gym facade:
 * Kuka.calc_state + get_touch_sensors, robot.py:152-163, 203-211): a pinned, device-mapped host block
 *   { f32 joints [N][9] | f32 touch [N][4] | f32 object poses [N][n_obj][7] | i32 timestep [N] | u32 errflags [N] }
 * which, once mapped, the last launch of every rr_step / rr_reset / rr_set_state / rr_set_object_pose(s) / rr_checkpoint_restore
 * on the library's stream refreshes.  Valid to read after rr_sync (one wait per step instead of one synchronising copy per
 * field); owned by the library until rr_destroy.  Repeated calls return the same block. */
int rr_map_observations(rr_env *env, void **host_ptr, size_t *bytes);
/* The same for the images of a handful of envs (EyeCamera.render returns host arrays, env.py:536-567): pinned host copies of
 * RR_F_RGB / RR_F_DEPTH / RR_F_MASK (pass NULL for what is not wanted), refreshed by asynchronous copies behind every rr_step that
 * renders and every rr_render; valid to read after rr_sync.  At most 256 MiB per step in total. */
int rr_map_images(rr_env *env, void **rgb_host, void **depth_host, void **mask_host);
/* Which of the mapped image blocks a rendered step refreshes: a mask of 1 (RGB), 2 (depth), 4 (mask); default 7.  A caller that
 * asked for the mask once (get_observation_extended, env.py:291-312) and then goes back to plain observations (env.py:266-289)
 * deselects it instead of paying its copy after every step; a field selected again is brought up to date at once (valid after
 * rr_sync_observations).  The blocks stay mapped either way. */
int rr_select_image_mirror(rr_env *env, int32_t fields);
/* Waits until the mapped blocks (rr_map_observations / rr_map_images) hold the observations of the last step -- not for the rest of
 * the stream: with a handful of envs the state part of the NEXT step is queued behind the mirror and runs while the
 * caller computes its next action.  Without a mapping it is rr_sync. */
int rr_sync_observations(rr_env *env);

/* Delta records for the observation gather of a sharded batch (the reference has one env per process and no gather):
 * the pixels in which the last rendered frame of every env may differ from the one before are the entries of the renderer's
 * fragment lists (RR_F_FRAG_COUNT of them per (env, tile) item).  rr_pack_image_delta writes one 12-byte record per entry --
 * { env * H * W + row * W + col, r | g << 8 | b << 16, depth bits }, the pixel's NEW value -- at offsets_dev[item] + i, where
 * offsets_dev (u32 [N * tiles], device) is the exclusive prefix sum of RR_F_FRAG_COUNT made by the caller; records beyond
 * `capacity` are dropped.  On the library's stream.  After a frame that rewrote whole images (the first render of a handle,
 * rr_set_camera) the lists do not describe the change: ship the slabs then.
 * rr_apply_image_delta is the receiving side, with no env handle: `world` blocks of `capacity` records of which the first
 * totals_dev[r] are valid, applied to persistent images of world * pixels_per_rank pixels (rank r's records address its block);
 * enqueued on `stream` (a hipStream_t or NULL) of the current device. */
int rr_pack_image_delta(rr_env *env, const uint32_t *offsets_dev, uint32_t *records_dev, uint32_t capacity);
int rr_apply_image_delta(const uint32_t *records_dev, const uint32_t *totals_dev, int32_t world, uint32_t capacity, size_t pixels_per_rank,
                         uint8_t *rgb_dev, float *depth_dev, void *stream);

/* Replaces robot.parts[name].get_position()/get_pose() (env.py:230-232): world pose of the COM frame of
 * every robot link, f32 [N, 17, 7] (URDF depth-first link order, see data/realrobot_model_links.txt). */
int rr_link_poses(rr_env *env, float *out_host);
/* Replaces Kuka.get_contacts (robot.py:131-150) for one env: up to max_contacts rows of 12 floats
 * {bodyA, bodyB, linkA, x,y,z, nx,ny,nz, distance, normal_force, mu}; *count receives the number written.  This list --
 * the contacts of the last step with the normal forces the solver found -- is also the contact history the next step's
 * warm start matches its contacts against (Bullet: persistent manifolds, m_warmstartingFactor 0.85). */
int rr_get_contacts(rr_env *env, int32_t env_index, float *out_host, int32_t max_contacts, int32_t *count);

/* Replaces REALRobotEnv.evaluateGoal (env.py:181-200) for all envs at once: score[i] = sum over the objects o of env i whose
 * goal_mask byte is non-zero (NULL: every object) of exp(-(ln 4 / 0.10) * |goal_pos[i][o] - position[i][o]|), computed on the
 * device from the state.  goal_pos: f32 [N, n_obj, 3] (host), goal_mask: u8 [N, n_obj] (host), score_out: f32 [N] (host). */
int rr_evaluate_goals(rr_env *env, const float *goal_pos_host, const uint8_t *goal_mask_host, float *score_out_host);

/* Batched damped-least-squares inverse kinematics for link 7 (gripper `base`), seeded with each env's current joints.
 * Replaces pybullet.calculateInverseKinematics(0, 7, pos, orn, maxNumIterations=1000, residualThreshold=0.001) in
 * step_cartesian (env.py:372-375).  targets: f32 [N, 7] (xyz + xyzw quaternion, host); q_out: f32 [N, 11] (all movable
 * dofs like pybullet; the fingers keep their current values); err_out (nullable): f32 [N] final pose residual. */
int rr_ik(rr_env *env, const float *targets_host, float *q_out_host, float *err_out_host);
/* Replaces REALRobotEnv.generate_plan (env.py:388-454) for the envs selected by the mask (NULL: all): builds the
 * 1000-step joint-space plan of a macro action [[x1, y1], [x2, y2]] on the device. macro: f32 [N, 4] host. */
int rr_plan_macro(rr_env *env, const float *macro_host, const uint8_t *env_mask_host);
/* Copies one env's plan to the host, f32 [1000, 9] (tests / debugging). */
int rr_get_plan(rr_env *env, int32_t env_index, float *plan_host);
/* Replaces step_macro's next_step() + step_joints (env.py:404-412, 463-467): every env consumes the next row of its
 * plan (the last row repeats once the plan is exhausted; the host decides when to re-plan) and steps. */
int rr_step_plan(rr_env *env, int32_t render_mode, const uint8_t *render_flags_host);
/* Same, but the envs whose idle byte is non-zero (u8 [N], host; NULL: none) take the command zeros(9) for this step and
 * keep their place in the plan: step_macro with macro_action None (env.py:391-393). */
int rr_step_plan_masked(rr_env *env, const uint8_t *idle_mask_host, int32_t render_mode, const uint8_t *render_flags_host);

/* Per-kernel device timing with HIP events on the library's stream (bench.py roofline leg).
 * After rr_set_timing(env, 1), each rr_step/rr_render records events; rr_get_timing returns accumulated
 * milliseconds and launch counts per kernel since the last call and resets them.
 * kernel ids: 0 prep (state part: forward kinematics, object terms, joint-space dynamics), 1 collide, 2 solve (command part --
 * rate limit, clipping, motor targets -- then rows, Gauss-Seidel, integration), 3 render_setup, 4 raster, 5 image set-up outside
 * the two render kernels (the full static copy of the first frame; the separate restore pass with RR_SEPARATE_RESTORE), 6 shade.
 * When the step would run its heavy envs on the side stream, the timed step runs the same launches one
 * after the other: 2 / 3 / 4 / 6 then hold what the main stream runs (the light envs), 7 the solve and 8 the render
 * (set-up + raster + shade) of the heavy envs, which run beside them in an untimed step.  0 and 1 are the LOOK-AHEAD of the
 * next step, which an untimed step runs under its render. */
#define RR_NUM_KERNELS 9
int rr_set_timing(rr_env *env, int32_t enable);
int rr_get_timing(rr_env *env, float *ms_out /*[RR_NUM_KERNELS]*/, int32_t *launches_out /*[RR_NUM_KERNELS]*/);

/* Device micro-benchmarks for the roofline of bench.py (the achievable figure is measured in the same run, not quoted):
 * kind 0: HBM copy bandwidth, 1: HBM triad bandwidth (GB/s; 256 MiB arrays); 2: VALU issue rate of a sample-test-like instruction
 * mix at eight waves per SIMD (G wave64-instructions/s).  No env handle needed; fails with RR_EDEVICE without a GPU. */
int rr_device_microbench(int32_t device, int32_t kind, double *result);

const char *rr_last_error(void);
int rr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
