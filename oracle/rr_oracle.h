/* rr_oracle.h -- CPU restatement ("oracle") of the REALRobot env.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under real_robots_amd/ may include, link or call this.
 * Allowed users: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 *
 * PARITY STATUS: "parity unpinned" w.r.t. PyBullet.  The arithmetic of the reference path lives in
 * the un-vendored, un-pinned third-party `pybullet` (setup.py:24-32 of the reference), which is not
 * installed here and cannot be built here.  This file restates Bullet's *published* algorithm
 * (articulated-body dynamics, velocity-level motor + contact rows, projected Gauss-Seidel,
 * semi-implicit Euler, TinyRenderer-style z-buffered rasteriser) and is pinned only against the
 * known answers the reference tree itself holds (tests/test_actions.py:60-71,147-152 FK/tracking,
 * generate_goals.py:249-272 rest heights); see tests/test_oracle_pins.py and DESIGN.md.
 *
 * Build:  make -C oracle          -> oracle/_build/librr_oracle.so   (real = double)
 *                                    oracle/_build/librr_oracle_f32.so (real = float)
 */
#ifndef RR_ORACLE_H
#define RR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef RR_FLOAT
typedef float real;
#else
typedef double real;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define RRO_NB 11        /* revolute dofs / moving bodies */
#define RRO_NOBJ 3       /* max free objects */
#define RRO_MAXC 48      /* contact cap per env */
#define RRO_STATE (2 * RRO_NB + 13 * RRO_NOBJ)   /* 61 */

typedef struct rro_params {
    double dt;              /* 0.005            env.py:203-204 */
    double gravity;         /* 9.81             env.py:203,208 */
    int solver_iters;       /* 50               Bullet default after resetSimulation */
    double erp;             /* 0.2              Bullet default contact ERP */
    double margin;          /* 0.02             Bullet contact breaking threshold */
    double motor_kp;        /* 0.1              pybullet setJointMotorControl2 default positionGain */
    double motor_kd;        /* 1.0              default velocityGain */
    double motor_max_force; /* 100000           default force   */
    double lin_damping;     /* 0.04             btMultiBody default base damping */
    double ang_damping;     /* 0.04 */
    double rest_threshold;  /* 0.2              m_restitutionVelocityThreshold */
    int use_urdf_inertia;   /* 0: Bullet AABB inertia for robot links (default); 1: URDF <inertia> */
    int edge_contacts;      /* 1                edge-edge candidates (0: vertex tests only) */
    double warmstart;       /* 0.85             Bullet m_warmstartingFactor on the matched normal impulses (0: cold start every step) */
    int no_rate_limit;      /* 0                diagnostics (tests/golden/make_macro_sensitivity.py): 1 skips limitActionByJoint (env.py:314-321) */
} rro_params;

typedef struct rr_oracle rr_oracle;

void rro_default_params(rro_params *p);
rr_oracle *rro_create(const void *blob, size_t nbytes, int n_objects, int width, int height,
                      const rro_params *params /* nullable */);
void rro_destroy(rr_oracle *o);
void rro_reset(rr_oracle *o);
/* One env.step_joints() (env.py:326-356) without the observation render.  action9 == NULL -> zeros(9).
 * returns 0, or -1 when the action is not finite (robot.py:189). */
int rro_step(rr_oracle *o, const double *action9);
/* top-down eye camera (env.py:249-255, 536-567). rgb u8[H,W,3], depth f32[H,W] (GL depth 0..1), mask i32[H,W] */
void rro_render(rr_oracle *o, uint8_t *rgb, float *depth, int32_t *mask);
/* camera override for the following renders (row-major 4x4 OpenGL view / projection; EnvCamera env.py:470-513) */
void rro_set_camera(rr_oracle *o, const float *view16, const float *proj16);
void rro_get_state(const rr_oracle *o, double *state61);
void rro_set_state(rr_oracle *o, const double *state61);      /* also forgets the contact history (cold start) */
/* contact history for the warm start, in rro_contacts() layout (call after rro_set_state) */
void rro_set_contact_cache(rr_oracle *o, const double *rec12, int n);
void rro_get_obs(const rr_oracle *o, double *joints9, double *touch4, double *objpos /*[nobj*3]*/);
int rro_timestep(const rr_oracle *o);
/* T steps in one call: per step the contact count, the number of contacts whose body A is a robot body, the four touch sensors;
 * the state every `state_every` steps.  Returns the number of states written (-1 - t: action t not finite). */
int rro_run(rr_oracle *o, const double *actions, int T, int *ncontacts, int *nrobot, double *touch, int state_every, double *states);
/* world pose (xyz + xyzw quaternion) of the COM frame of robot link `link` (0..16, URDF depth-first) */
void rro_link_pose(const rr_oracle *o, int link, double *pose7);
/* contacts of the last step: per contact 12 doubles {bodyA, bodyB, linkA, x,y,z, nx,ny,nz, dist, normal_force, mu};
 * returns count */
int rro_contacts(const rr_oracle *o, double *out, int max_contacts);
void rro_set_object_pose(rr_oracle *o, int obj, const double *pose7);
/* solver-independent check of a candidate solution (post-step velocities in state61 layout + one normal force per
 * contact) of the contact problem of the last rro_step: natural residual of the normal rows, force sum, active set.
 * out5 = {sum residual (N), max residual (N), sum of normal forces, largest normal force, active contacts} */
int rro_solution_residual(const rr_oracle *o, const double *state_after61, const double *normal_force, int n,
                          double active_thresh, double *out5, uint64_t *active_bits);
/* diagnostics */
/* narrow phase of one shape pair (model shape indices) at the present state; xf24 (nullable): the shapes' world transforms */
int rro_pair_contacts(rr_oracle *o, int shape_a, int shape_b, double *out, int max_contacts, double *xf24);
void rro_mass_matrix(rr_oracle *o, double *M121, double *bias11);

#ifdef __cplusplus
}
#endif
#endif
