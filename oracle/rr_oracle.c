/* rr_oracle.c -- CPU restatement of the REALRobot env.step() hot path.  See rr_oracle.h (header comment:
 * TEST INFRASTRUCTURE ONLY, "parity unpinned" w.r.t. PyBullet).
 *
 * Reference anchors (paths relative to /root/reference):
 *   step protocol          real_robots/envs/env.py:326-356  (step_joints)
 *   rate limit             real_robots/envs/env.py:314-321  (limitActionByJoint)
 *   OOB object reset       real_robots/envs/env.py:257-264  (control_objects_limits)
 *   action -> 11 motors    real_robots/envs/robot.py:188-201 (apply_action)
 *   joint read-back        real_robots/envs/robot.py:203-211 (calc_state)
 *   touch sensors          real_robots/envs/robot.py:131-163
 *   reset                  real_robots/envs/env.py:206-219, robot.py:120-129,165-185
 *   physics step           env.py:340 scene.global_step() -> pybullet.stepSimulation (UPSTREAM Bullet3,
 *                          restated from its published design: btMultiBody forward dynamics,
 *                          btMultiBodyJointMotor rows, multibody contact rows, PGS, semi-implicit Euler)
 *   camera                 env.py:136-141,249-255,516-567 (EyeCamera.renderTarget -> TinyRenderer)
 */
#include "rr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NB RRO_NB
#define NOBJ RRO_NOBJ
#define MAXC RRO_MAXC
#define NROWS (3 * NB + 6 * MAXC)
#define NDOF (NB + 6 * NOBJ)

#ifdef RR_FLOAT
#define RFMA(a, b, c) fmaf((a), (b), (c))
#define RSQRT(x) sqrtf(x)
#define RFABS(x) fabsf(x)
#define RSIN(x) sinf(x)
#define RCOS(x) cosf(x)
#else
#define RFMA(a, b, c) fma((a), (b), (c))
#define RSQRT(x) sqrt(x)
#define RFABS(x) fabs(x)
#define RSIN(x) sin(x)
#define RCOS(x) cos(x)
#endif

/* ------------------------------------------------------------------------------------------- blob */
typedef struct {
    char name[32];
    uint32_t dtype, ndim, shape[4];
    uint64_t offset, nbytes;
} blob_entry;

static const blob_entry *blob_find(const void *blob, const char *name) {
    const char *b = (const char *)blob;
    uint32_t n;
    if (memcmp(b, "RRMODEL1", 8) != 0) return NULL;
    memcpy(&n, b + 8, 4);
    const blob_entry *e = (const blob_entry *)(b + 16);
    for (uint32_t i = 0; i < n; i++)
        if (strncmp(e[i].name, name, 32) == 0) return &e[i];
    return NULL;
}
static const float *blob_f32(const void *blob, const char *name) {
    const blob_entry *e = blob_find(blob, name);
    if (!e || e->dtype != 0) { fprintf(stderr, "rr_oracle: missing f32 '%s'\n", name); abort(); }
    return (const float *)((const char *)blob + e->offset);
}
static const int32_t *blob_i32(const void *blob, const char *name) {
    const blob_entry *e = blob_find(blob, name);
    if (!e || e->dtype != 1) { fprintf(stderr, "rr_oracle: missing i32 '%s'\n", name); abort(); }
    return (const int32_t *)((const char *)blob + e->offset);
}
static const uint8_t *blob_u8(const void *blob, const char *name) {
    const blob_entry *e = blob_find(blob, name);
    if (!e || e->dtype != 2) { fprintf(stderr, "rr_oracle: missing u8 '%s'\n", name); abort(); }
    return (const uint8_t *)((const char *)blob + e->offset);
}

/* ------------------------------------------------------------------------------------------- model */
#define MAXSHAPES 32
#define EMAXC 48         /* long sharp hull edges per shape (tools/compile_model.py EMAX) */
#define VMAXC 192
#define FMAXC 192
#define MAXINST 32
#define MAXLINKS 24

typedef struct {
    int nb, nl, ns, ni, nt, ntex, n_static, n_robot, vmax, fmax;
    real robot_pos[3];
    int parent[NB];
    real jpos[NB][3], jrot[NB][9], axis[NB][3], mass[NB], com[NB][3], inertia[NB][6], damping[NB], limits[NB][2];
    real obj_mass[NOBJ], obj_inertia[NOBJ][3], obj_pose0[NOBJ][7];
    real table_pos[3];
    int sh_otype[MAXSHAPES], sh_oidx[MAXSHAPES], sh_link[MAXSHAPES], sh_uid[MAXSHAPES];
    int sh_nv[MAXSHAPES], sh_nf[MAXSHAPES];
    real sh_verts[MAXSHAPES][VMAXC][3], sh_planes[MAXSHAPES][FMAXC][4], sh_sphere[MAXSHAPES][4];
    real sh_fric[MAXSHAPES], sh_rest[MAXSHAPES], sh_roll[MAXSHAPES], sh_spin[MAXSHAPES];
    int sh_ne[MAXSHAPES];
    real sh_edges[MAXSHAPES][EMAXC][12];     /* long sharp hull edges: p0, p1 - p0, the two facet normals (owner frame) */
    int touch_links[4];
    int link_body[MAXLINKS];
    real link_pos[MAXLINKS][3], link_rot[MAXLINKS][9];
    real act_min[9], act_max[9], act_maxdiff[9];
    /* render (kept float: the rasteriser is restated in float on purpose, see raster section) */
    int in_otype[MAXINST], in_oidx[MAXINST], in_uid[MAXINST], in_tex[MAXINST], in_start[MAXINST], in_count[MAXINST];
    float in_color[MAXINST][3];
    const float *tri_pos, *tri_nrm, *tri_uv;
    const int32_t *tri_inst, *tex_info;
    const uint8_t *tex_data;
} model_t;

typedef struct {
    int bodyA, bodyB;   /* -1 static, 0..10 robot body, 16+i object i.  normal points from B to A */
    int linkA;          /* robot link id of the robot-side shape (or -1) */
    real x[3], n[3], dist, mu, rest;
    real roll, spin;    /* combined rolling / spinning friction coefficients */
    real lambda_n;      /* normal impulse the solver found */
    real lambda0;       /* warm start: warmstart factor x the normal impulse of the matched contact of the previous step */
} contact_t;

struct rr_oracle {
    model_t m;
    rro_params p;
    int nobj, W, H;
    void *blob_copy;
    /* state */
    real q[NB], qd[NB];
    real opos[NOBJ][3], oquat[NOBJ][4], ovel[NOBJ][3], oang[NOBJ][3];
    real tgt[NB];
    int timestep;
    real touch[4];
    int ncontacts;
    contact_t contacts[MAXC];
    /* camera override (EnvCamera, env.py:470-513): row-major proj * view, set by rro_set_camera */
    int cam_custom;
    float cam_VP[16];
    /* kinematics cache */
    real bR[NB][9], bp[NB][3], baxis[NB][3], bcom[NB][3], bIw[NB][9];
};

/* ------------------------------------------------------------------------------------------- small math */
static inline void v3_set(real *o, real a, real b, real c) { o[0] = a; o[1] = b; o[2] = c; }
static inline void v3_copy(real *o, const real *a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static inline void v3_add(real *o, const real *a, const real *b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3_sub(real *o, const real *a, const real *b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void v3_scale(real *o, const real *a, real s) { o[0] = a[0] * s; o[1] = a[1] * s; o[2] = a[2] * s; }
static inline void v3_madd(real *o, const real *a, real s) { o[0] += a[0] * s; o[1] += a[1] * s; o[2] += a[2] * s; }
static inline real v3_dot(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3_cross(real *o, const real *a, const real *b) {
    real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3_mulv(real *o, const real *M, const real *v) {
    real x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    real y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    real z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3_tmulv(real *o, const real *M, const real *v) { /* M^T v */
    real x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    real y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    real z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3_mul(real *o, const real *A, const real *B) {
    real t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(o, t, sizeof t);
}
#ifdef RR_FLOAT
/* The float build is the bit-level checker of the device's collision decisions (which contacts exist): joint rotations
 * use this explicit sequence of IEEE single operations -- the same one as det_sincosf in csrc/realrobot.hip -- because
 * glibc's sinf/cosf and the device's differ in the last bit.  Cody-Waite reduction by pi/2 in three parts, Cephes
 * minimax polynomials on [-pi/4, pi/4]; < 2 ulp for |x| up to a few turns (joint angles).  The float64 build (the oracle
 * proper) uses libm. */
static void det_sincosf(float x, float *sn, float *cs) {
    const float k = rintf(x * 0.63661977236758134308f);
    float r = x - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.54978995489188216e-8f;
    const float z = r * r;
    float ps = -1.9515295891e-4f * z + 8.3321608736e-3f;
    ps = ps * z - 1.6666654611e-1f;
    const float s = r + r * z * ps;
    float pc = 2.443315711809948e-5f * z - 1.388731625493765e-3f;
    pc = pc * z + 4.166664568298827e-2f;
    const float c = (1.0f - 0.5f * z) + z * z * pc;
    const int n = (int)k & 3;
    *sn = n == 0 ? s : (n == 1 ? c : (n == 2 ? -s : -c));
    *cs = n == 0 ? c : (n == 1 ? -s : (n == 2 ? -c : s));
}
#endif
/* Rodrigues: rotation about unit axis a by angle */
static void m3_axis_angle(real *R, const real *a, real ang) {
#ifdef RR_FLOAT
    real c, s;
    det_sincosf(ang, &s, &c);
    real t = 1 - c;
#else
    real c = RCOS(ang), s = RSIN(ang), t = 1 - c;
#endif
    R[0] = t * a[0] * a[0] + c;        R[1] = t * a[0] * a[1] - s * a[2]; R[2] = t * a[0] * a[2] + s * a[1];
    R[3] = t * a[0] * a[1] + s * a[2]; R[4] = t * a[1] * a[1] + c;        R[5] = t * a[1] * a[2] - s * a[0];
    R[6] = t * a[0] * a[2] - s * a[1]; R[7] = t * a[1] * a[2] + s * a[0]; R[8] = t * a[2] * a[2] + c;
}
static void quat_to_m3(real *R, const real *q) { /* xyzw */
    real x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
static void m3_to_quat(real *q, const real *R) {
    real t = R[0] + R[4] + R[8];
    if (t > 0) {
        real s = RSQRT(t + 1) * 2;
        q[3] = s / 4; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        real s = RSQRT(1 + R[0] - R[4] - R[8]) * 2;
        q[3] = (R[7] - R[5]) / s; q[0] = s / 4; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        real s = RSQRT(1 + R[4] - R[0] - R[8]) * 2;
        q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = s / 4; q[2] = (R[5] + R[7]) / s;
    } else {
        real s = RSQRT(1 + R[8] - R[0] - R[4]) * 2;
        q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = s / 4;
    }
}
/* symmetric inertia (xx,yy,zz,xy,xz,yz) rotated to world: R I R^T, full 3x3 out */
static void inertia_world(real *Iw, const real *R, const real *I6) {
    real I[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
    real T[9], Rt[9] = {R[0], R[3], R[6], R[1], R[4], R[7], R[2], R[5], R[8]};
    m3_mul(T, R, I);
    m3_mul(Iw, T, Rt);
}

/* ------------------------------------------------------------------------------------------- create */
void rro_default_params(rro_params *p) {
    p->dt = 0.005; p->gravity = 9.81; p->solver_iters = 50; p->erp = 0.2; p->margin = 0.02; p->edge_contacts = 1; p->warmstart = 0.85;
    p->motor_kp = 0.1; p->motor_kd = 1.0; p->motor_max_force = 100000.0;
    p->lin_damping = 0.04; p->ang_damping = 0.04; p->rest_threshold = 0.2; p->use_urdf_inertia = 0; p->no_rate_limit = 0;
}

static void cpy_f(real *dst, const float *src, int n) { for (int i = 0; i < n; i++) dst[i] = (real)src[i]; }

rr_oracle *rro_create(const void *blob_in, size_t nbytes, int n_objects, int width, int height, const rro_params *params) {
    if (n_objects < 1 || n_objects > NOBJ) return NULL;
    rr_oracle *o = (rr_oracle *)calloc(1, sizeof *o);
    o->blob_copy = malloc(nbytes);
    memcpy(o->blob_copy, blob_in, nbytes);
    const void *blob = o->blob_copy;
    if (params) o->p = *params; else rro_default_params(&o->p);
    o->nobj = n_objects; o->W = width; o->H = height;
    model_t *m = &o->m;
    const int32_t *dims = blob_i32(blob, "dims");
    m->nb = dims[0]; m->nl = dims[1]; m->ns = dims[2]; m->ni = dims[3]; m->nt = dims[4]; m->ntex = dims[5];
    m->n_static = dims[6]; m->n_robot = dims[7]; m->vmax = dims[8]; m->fmax = dims[9];
    if (m->nb != NB || m->ns > MAXSHAPES || m->vmax != VMAXC || m->fmax != FMAXC || m->ni > MAXINST || m->nl > MAXLINKS) {
        fprintf(stderr, "rr_oracle: blob dims mismatch\n"); abort();
    }
    cpy_f(m->robot_pos, blob_f32(blob, "robot_pos"), 3);
    memcpy(m->parent, blob_i32(blob, "body_parent"), sizeof(int) * NB);
    cpy_f(&m->jpos[0][0], blob_f32(blob, "body_jpos"), NB * 3);
    cpy_f(&m->jrot[0][0], blob_f32(blob, "body_jrot"), NB * 9);
    cpy_f(&m->axis[0][0], blob_f32(blob, "body_axis"), NB * 3);
    cpy_f(m->mass, blob_f32(blob, "body_mass"), NB);
    cpy_f(&m->com[0][0], blob_f32(blob, "body_com"), NB * 3);
    cpy_f(&m->inertia[0][0], blob_f32(blob, o->p.use_urdf_inertia ? "body_inertia_urdf" : "body_inertia"), NB * 6);
    cpy_f(m->damping, blob_f32(blob, "body_damping"), NB);
    cpy_f(&m->limits[0][0], blob_f32(blob, "body_limits"), NB * 2);
    cpy_f(m->obj_mass, blob_f32(blob, "obj_mass"), NOBJ);
    cpy_f(&m->obj_inertia[0][0], blob_f32(blob, "obj_inertia"), NOBJ * 3);
    cpy_f(&m->obj_pose0[0][0], blob_f32(blob, "obj_pose0"), NOBJ * 7);
    cpy_f(m->table_pos, blob_f32(blob, "table_pos"), 3);
    const int32_t *so = blob_i32(blob, "shape_owner");
    for (int s = 0; s < m->ns; s++) {
        m->sh_otype[s] = so[4 * s]; m->sh_oidx[s] = so[4 * s + 1]; m->sh_link[s] = so[4 * s + 2]; m->sh_uid[s] = so[4 * s + 3];
    }
    memcpy(m->sh_nv, blob_i32(blob, "shape_nv"), sizeof(int) * m->ns);
    memcpy(m->sh_nf, blob_i32(blob, "shape_nf"), sizeof(int) * m->ns);
    cpy_f(&m->sh_verts[0][0][0], blob_f32(blob, "shape_verts"), m->ns * VMAXC * 3);
    cpy_f(&m->sh_planes[0][0][0], blob_f32(blob, "shape_planes"), m->ns * FMAXC * 4);
    cpy_f(&m->sh_sphere[0][0], blob_f32(blob, "shape_sphere"), m->ns * 4);
    const float *sm = blob_f32(blob, "shape_mat");
    for (int s = 0; s < m->ns; s++) { m->sh_fric[s] = sm[2 * s]; m->sh_rest[s] = sm[2 * s + 1]; }
    const float *sr = blob_f32(blob, "shape_roll");
    for (int s = 0; s < m->ns; s++) { m->sh_roll[s] = sr[2 * s]; m->sh_spin[s] = sr[2 * s + 1]; }
    memcpy(m->sh_ne, blob_i32(blob, "shape_ne"), sizeof(int) * m->ns);
    cpy_f(&m->sh_edges[0][0][0], blob_f32(blob, "shape_edges"), m->ns * EMAXC * 12);
    memcpy(m->touch_links, blob_i32(blob, "touch_links"), sizeof(int) * 4);
    memcpy(m->link_body, blob_i32(blob, "link_body"), sizeof(int) * m->nl);
    cpy_f(&m->link_pos[0][0], blob_f32(blob, "link_pos"), m->nl * 3);
    cpy_f(&m->link_rot[0][0], blob_f32(blob, "link_rot"), m->nl * 9);
    cpy_f(m->act_min, blob_f32(blob, "act_min"), 9);
    cpy_f(m->act_max, blob_f32(blob, "act_max"), 9);
    cpy_f(m->act_maxdiff, blob_f32(blob, "act_maxdiff"), 9);
    const int32_t *io = blob_i32(blob, "inst_owner"), *ir = blob_i32(blob, "inst_range");
    const float *ic = blob_f32(blob, "inst_color");
    for (int i = 0; i < m->ni; i++) {
        m->in_otype[i] = io[4 * i]; m->in_oidx[i] = io[4 * i + 1]; m->in_uid[i] = io[4 * i + 2]; m->in_tex[i] = io[4 * i + 3];
        m->in_start[i] = ir[2 * i]; m->in_count[i] = ir[2 * i + 1];
        for (int k = 0; k < 3; k++) m->in_color[i][k] = ic[3 * i + k];
    }
    m->tri_pos = blob_f32(blob, "tri_pos"); m->tri_nrm = blob_f32(blob, "tri_nrm"); m->tri_uv = blob_f32(blob, "tri_uv");
    m->tri_inst = blob_i32(blob, "tri_inst"); m->tex_info = blob_i32(blob, "tex_info"); m->tex_data = blob_u8(blob, "tex_data");
    rro_reset(o);
    return o;
}

void rro_destroy(rr_oracle *o) {
    if (!o) return;
    free(o->blob_copy);
    free(o);
}

/* env.reset (env.py:206-219; robot.py:165-185): joints (0,0), objects at object_poses, timestep 0 */
void rro_reset(rr_oracle *o) {
    memset(o->q, 0, sizeof o->q);
    memset(o->qd, 0, sizeof o->qd);
    memset(o->tgt, 0, sizeof o->tgt);
    for (int i = 0; i < NOBJ; i++) {
        v3_copy(o->opos[i], o->m.obj_pose0[i]);
        for (int k = 0; k < 4; k++) o->oquat[i][k] = o->m.obj_pose0[i][3 + k];
        v3_set(o->ovel[i], 0, 0, 0);
        v3_set(o->oang[i], 0, 0, 0);
    }
    o->timestep = 0;
    o->ncontacts = 0;
    for (int k = 0; k < 4; k++) o->touch[k] = 0;
}

/* ------------------------------------------------------------------------------------------- kinematics */
static void forward_kinematics(rr_oracle *o) {
    const model_t *m = &o->m;
    const real I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int b = 0; b < NB; b++) {
        const real *Rp = I3, *pp = m->robot_pos;
        if (m->parent[b] >= 0) { Rp = o->bR[m->parent[b]]; pp = o->bp[m->parent[b]]; }
        real Rj[9], Rq[9], t[3];
        m3_mul(Rj, Rp, m->jrot[b]);
        m3_mulv(t, Rp, m->jpos[b]);
        v3_add(o->bp[b], pp, t);
        m3_axis_angle(Rq, m->axis[b], o->q[b]);
        m3_mul(o->bR[b], Rj, Rq);
        m3_mulv(o->baxis[b], Rj, m->axis[b]);
        m3_mulv(t, o->bR[b], m->com[b]);
        v3_add(o->bcom[b], o->bp[b], t);
        inertia_world(o->bIw[b], o->bR[b], m->inertia[b]);
    }
}

/* ------------------------------------------------------------------------------------------- dynamics */
/* Joint-space mass matrix by composite rigid bodies (COM-relative form, well conditioned in fp32) and
 * bias forces (Coriolis/centrifugal + gravity) by recursive Newton-Euler; all in world coordinates. */
static void mass_matrix_and_bias(rr_oracle *o, real M[NB][NB], real bias[NB]) {
    const model_t *m = &o->m;
    /* composite inertia: mass, COM, inertia about the composite COM */
    real cm[NB], cc[NB][3], cI[NB][9];
    for (int b = 0; b < NB; b++) {
        cm[b] = m->mass[b];
        v3_copy(cc[b], o->bcom[b]);
        memcpy(cI[b], o->bIw[b], sizeof(real) * 9);
    }
    for (int b = NB - 1; b >= 0; b--) {
        int p = m->parent[b];
        if (p < 0) continue;
        /* merge composite b into composite p */
        real mt = cm[p] + cm[b], c[3], d1[3], d2[3];
        for (int k = 0; k < 3; k++) c[k] = (cm[p] * cc[p][k] + cm[b] * cc[b][k]) / mt;
        v3_sub(d1, cc[p], c);
        v3_sub(d2, cc[b], c);
        real s1 = v3_dot(d1, d1), s2 = v3_dot(d2, d2);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                real e = (i == j) ? 1 : 0;
                cI[p][3 * i + j] = cI[p][3 * i + j] + cI[b][3 * i + j] + cm[p] * (s1 * e - d1[i] * d1[j]) + cm[b] * (s2 * e - d2[i] * d2[j]);
            }
        cm[p] = mt;
        v3_copy(cc[p], c);
    }
    for (int i = 0; i < NB; i++)
        for (int j = 0; j < NB; j++) M[i][j] = 0;
    for (int j = 0; j < NB; j++) {
        /* momentum of composite j under unit velocity of joint j */
        real Ia[3], rj[3], vj[3], f[3];
        m3_mulv(Ia, cI[j], o->baxis[j]);
        v3_sub(rj, cc[j], o->bp[j]);
        v3_cross(vj, o->baxis[j], rj);
        v3_scale(f, vj, cm[j]);
        int i = j;
        while (i >= 0) {
            real ri[3], t[3], nn[3];
            v3_sub(ri, cc[j], o->bp[i]);
            v3_cross(t, ri, f);
            v3_add(nn, Ia, t);
            real v = v3_dot(o->baxis[i], nn);
            M[i][j] = v;
            M[j][i] = v;
            i = m->parent[i];
        }
    }
    /* RNEA with qdd = 0, base acceleration +g (gravity trick) */
    real w[NB][3], al[NB][3], ap[NB][3], F[NB][3], N[NB][3];
    for (int b = 0; b < NB; b++) {
        int p = m->parent[b];
        real wp[3] = {0, 0, 0}, alp[3] = {0, 0, 0}, app[3] = {0, 0, (real)o->p.gravity}, pp[3];
        if (p >= 0) { v3_copy(wp, w[p]); v3_copy(alp, al[p]); v3_copy(app, ap[p]); v3_copy(pp, o->bp[p]); }
        else v3_copy(pp, m->robot_pos);
        real t[3], t2[3], d[3];
        v3_copy(w[b], wp);
        v3_madd(w[b], o->baxis[b], o->qd[b]);
        v3_cross(t, wp, o->baxis[b]);
        v3_copy(al[b], alp);
        v3_madd(al[b], t, o->qd[b]);
        v3_sub(d, o->bp[b], pp);
        v3_cross(t, alp, d);
        v3_cross(t2, wp, d);
        v3_cross(t2, wp, t2);
        v3_add(ap[b], app, t);
        v3_add(ap[b], ap[b], t2);
        /* COM acceleration */
        real r[3], ac[3];
        v3_sub(r, o->bcom[b], o->bp[b]);
        v3_cross(t, al[b], r);
        v3_cross(t2, w[b], r);
        v3_cross(t2, w[b], t2);
        v3_add(ac, ap[b], t);
        v3_add(ac, ac, t2);
        v3_scale(F[b], ac, m->mass[b]);
        real Iw[3], Ial[3];
        m3_mulv(Iw, o->bIw[b], w[b]);
        m3_mulv(Ial, o->bIw[b], al[b]);
        v3_cross(t, w[b], Iw);
        v3_add(N[b], Ial, t);
        v3_cross(t, r, F[b]);
        v3_add(N[b], N[b], t); /* torque about the body origin p_b */
    }
    for (int b = NB - 1; b >= 0; b--) {
        bias[b] = v3_dot(o->baxis[b], N[b]);
        int p = m->parent[b];
        if (p >= 0) {
            real d[3], t[3];
            v3_sub(d, o->bp[b], o->bp[p]);
            v3_cross(t, d, F[b]);
            v3_add(N[p], N[p], N[b]);
            v3_add(N[p], N[p], t);
            v3_add(F[p], F[p], F[b]);
        }
    }
}

/* dense Cholesky M = L L^T (lower), in place on a copy; returns L; then Minv */
static void cholesky_inverse(const real M[NB][NB], real Minv[NB][NB]) {
    real L[NB][NB];
    memset(L, 0, sizeof L);
    for (int i = 0; i < NB; i++) {
        for (int j = 0; j <= i; j++) {
            real s = M[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) L[i][i] = RSQRT(s);
            else L[i][j] = s / L[j][j];
        }
    }
    /* solve for each unit vector */
    for (int c = 0; c < NB; c++) {
        real y[NB];
        for (int i = 0; i < NB; i++) {
            real s = (i == c) ? 1 : 0;
            for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
            y[i] = s / L[i][i];
        }
        for (int i = NB - 1; i >= 0; i--) {
            real s = y[i];
            for (int k = i + 1; k < NB; k++) s -= L[k][i] * Minv[k][c];
            Minv[i][c] = s / L[i][i];
        }
    }
}

/* ------------------------------------------------------------------------------------------- collision */
typedef struct { real R[9], p[3]; } xform_t;

static void shape_xform(const rr_oracle *o, int s, xform_t *X) {
    const model_t *m = &o->m;
    static const real I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (m->sh_otype[s] == 0) { memcpy(X->R, I3, sizeof I3); v3_set(X->p, 0, 0, 0); }
    else if (m->sh_otype[s] == 1) { memcpy(X->R, o->bR[m->sh_oidx[s]], sizeof I3); v3_copy(X->p, o->bp[m->sh_oidx[s]]); }
    else { quat_to_m3(X->R, o->oquat[m->sh_oidx[s]]); v3_copy(X->p, o->opos[m->sh_oidx[s]]); }
}
static int shape_body(const model_t *m, int s) {
    if (m->sh_otype[s] == 0) return -1;
    if (m->sh_otype[s] == 1) return m->sh_oidx[s];
    return 16 + m->sh_oidx[s];
}

typedef struct { real x[3], n[3], s; } cand_t;
#define CAND_MAX 128   /* candidates kept per pair, in candidate order (the device kernel has the same cap) */

/* vertices of shape sa tested against the planes of shape sb. Appends candidates with normal pointing
 * from sb towards sa, multiplied by `sign` (so callers can keep the B->A convention). */
static int verts_in_planes(const rr_oracle *o, int sa, const xform_t *Xa, int sb, const xform_t *Xb, real sign,
                           cand_t *out, int n) {
    const model_t *m = &o->m;
    real margin = (real)o->p.margin;
    for (int v = 0; v < m->sh_nv[sa]; v++) {
        /* The inner loops of the narrow phase are written with explicit fused multiply-adds, in a fixed association that the
         * device kernel repeats instruction for instruction (a C fma() and v_fma_f32 round identically): half the operations
         * of the unfused form, and still bit-identical on both sides.  (Build with -mfma; without it libm emulates fma exactly.) */
        real xw[3], d[3], xl[3];
        const real *R = Xa->R, *Q = Xb->R, *vv = m->sh_verts[sa][v];
        xw[0] = RFMA(R[0], vv[0], RFMA(R[1], vv[1], RFMA(R[2], vv[2], Xa->p[0])));
        xw[1] = RFMA(R[3], vv[0], RFMA(R[4], vv[1], RFMA(R[5], vv[2], Xa->p[1])));
        xw[2] = RFMA(R[6], vv[0], RFMA(R[7], vv[1], RFMA(R[8], vv[2], Xa->p[2])));
        v3_sub(d, xw, Xb->p);
        xl[0] = RFMA(Q[0], d[0], RFMA(Q[3], d[1], Q[6] * d[2]));
        xl[1] = RFMA(Q[1], d[0], RFMA(Q[4], d[1], Q[7] * d[2]));
        xl[2] = RFMA(Q[2], d[0], RFMA(Q[5], d[1], Q[8] * d[2]));
        real best = -1e30f;
        int bf = 0;
        for (int f = 0; f < m->sh_nf[sb]; f++) {
            const real *pl = m->sh_planes[sb][f];
            real s = RFMA(pl[0], xl[0], RFMA(pl[1], xl[1], RFMA(pl[2], xl[2], -pl[3])));
            if (s > best) { best = s; bf = f; }
        }
        if (best < margin && n < CAND_MAX) {
            real nw[3];
            m3_mulv(nw, Xb->R, m->sh_planes[sb][bf]);
            cand_t *c = &out[n++];
            /* contact point: midway between the vertex and its projection on the plane */
            c->x[0] = xw[0] - (real)0.5 * best * nw[0];
            c->x[1] = xw[1] - (real)0.5 * best * nw[1];
            c->x[2] = xw[2] - (real)0.5 * best * nw[2];
            v3_scale(c->n, nw, sign);
            c->s = best;
        }
    }
    return n;
}

/* Edge-edge candidates.  The vertex tests above cannot see two edges that cross away from any vertex (a cube edge lying
 * across a shelf edge: GJK/EPA in Bullet reports the closest points of the two edges).  For every pair of long sharp hull
 * edges (tools/compile_model.py: >= 4 mm, dihedral angle >= 15 degrees, the 48 longest of a shape) whose lines' closest
 * points lie strictly inside both segments: the common normal n = +-(d1 x d2)/|d1 x d2| is a contact normal iff it is a
 * face of the Minkowski difference, i.e. the direction u from A towards B lies in the fan between the two facet normals
 * of A's edge and -u in the fan of B's edge.  With the closest points inside both segments such a pair is the closest
 * feature pair of the two convex shapes when they are apart (signed distance along the normal B -> A in [0, margin)); for
 * overlapping shapes it is one separating-axis candidate among others, accepted only while shallow (> -EDGE_DEPTH): the
 * true minimum translation is then no deeper than that. */
#define EDGE_DEPTH ((real)0.005)
#define EDGE_SLOP ((real)0.0005)
static int edge_edge(const rr_oracle *o, int sa, const xform_t *Xa, int sb, const xform_t *Xb, cand_t *out, int n) {
    const model_t *m = &o->m;
    real margin = (real)o->p.margin;
    /* overlapping shapes: an edge axis is only the contact normal if it is not deeper than the face axes -- the deepest
     * vertex candidate measures those (a gripper pad pressed on a cube face overlaps it by a fraction of a millimetre; the
     * pad's edges cross the cube's edges with much larger overlaps along their common normals, which are no contacts) */
    real smin = 0;
    for (int i = 0; i < n; i++) if (out[i].s < smin) smin = out[i].s;
    const real lo = o->p.edge_contacts == 2 ? (real)-1 : smin - EDGE_SLOP;      /* (2: diagnostic, no such guard) */
    for (int i = 0; i < m->sh_ne[sa]; i++) {
        const real *ea = m->sh_edges[sa][i];
        real P0[3], D1[3], a1[3], a2[3];
        m3_mulv(P0, Xa->R, ea); v3_add(P0, P0, Xa->p);
        m3_mulv(D1, Xa->R, ea + 3); m3_mulv(a1, Xa->R, ea + 6); m3_mulv(a2, Xa->R, ea + 9);
        for (int j = 0; j < m->sh_ne[sb]; j++) {
            const real *eb = m->sh_edges[sb][j];
            real Q0[3], D2[3], b1[3], b2[3];
            m3_mulv(Q0, Xb->R, eb); v3_add(Q0, Q0, Xb->p);
            m3_mulv(D2, Xb->R, eb + 3); m3_mulv(b1, Xb->R, eb + 6); m3_mulv(b2, Xb->R, eb + 9);
            real r[3];
            v3_sub(r, P0, Q0);
            real a = v3_dot(D1, D1), e = v3_dot(D2, D2), b = v3_dot(D1, D2), c = v3_dot(D1, r), f = v3_dot(D2, r);
            real ae = a * e, den = ae - b * b;
            if (!(den > (real)1e-4 * ae)) continue;           /* (nearly) parallel: left to the end points */
            real s = (b * f - c * e) / den, t = (a * f - b * c) / den;
            if (!(s > 0 && s < 1 && t > 0 && t < 1)) continue;
            real p[3], q[3], nv[3], u[3], w[3], ma[3], mb[3], c1[3], c2[3], pq[3];
            for (int k = 0; k < 3; k++) { p[k] = P0[k] + s * D1[k]; q[k] = Q0[k] + t * D2[k]; }
            v3_cross(nv, D1, D2);
            real il = (real)1 / RSQRT(v3_dot(nv, nv));
            v3_add(ma, a1, a2);
            real sg = v3_dot(nv, ma) > 0 ? il : -il;           /* u: unit, from A towards B */
            v3_scale(u, nv, sg);
            v3_cross(c1, a1, u); v3_cross(c2, u, a2);
            if (!(v3_dot(c1, c2) >= 0)) continue;
            v3_scale(w, u, (real)-1);
            v3_add(mb, b1, b2);
            if (!(v3_dot(w, mb) > 0)) continue;
            v3_cross(c1, b1, w); v3_cross(c2, w, b2);
            if (!(v3_dot(c1, c2) >= 0)) continue;
            v3_sub(pq, p, q);
            real dist = v3_dot(w, pq);
            if (!(dist < margin && dist > -EDGE_DEPTH && dist > lo)) continue;
            if (n < CAND_MAX) {
                cand_t *cd = &out[n++];
                for (int k = 0; k < 3; k++) cd->x[k] = q[k] + (real)0.5 * pq[k];
                v3_copy(cd->n, w);
                cd->s = dist;
            }
        }
    }
    return n;
}

/* Manifold reduction to <= 4 points: deepest point first, then the three points that spread the manifold most
 * (farthest from the first, farthest from that line, farthest on the other side of it). "Deepest" is taken with a
 * tolerance (TIE_TOL, see below). Candidates within
 * TIER_TOL of the deepest penetration ("tier 1": the features actually touching) are preferred at every pick;
 * the remaining speculative candidates are only used when tier 1 has no admissible point. */
#define TIER_TOL ((real)0.001)
#define TIE_TOL ((real)0.0005)
static int reduce4(const cand_t *c, int n, int *sel) {
    if (n <= 4) { for (int i = 0; i < n; i++) sel[i] = i; return n; }
    int k0 = 0;
    for (int i = 1; i < n; i++) if (c[i].s < c[k0].s) k0 = i;
    real lim = c[k0].s + TIER_TOL;
    /* the anchor of the manifold: the deepest candidate, with a tolerance TIE_TOL on "deepest" -- an
     * object resting flat has a whole face within a fraction of a millimetre of the deepest vertex; anchoring on "the"
     * deepest one would pick another quadruple of the face's vertices whenever the object rocks by a micro-radian, and
     * without persistent manifolds that alone keeps a bevelled cube rocking for ever */
    {   /* among the tied candidates the extreme one along a fixed skew direction: a corner of the face, so that the
         * farthest-point picks below return the face's other corners */
        real tl = c[k0].s + TIE_TOL, bestf = (real)-3.0e38;
        for (int i = 0; i < n; i++) {
            if (!(c[i].s < tl)) continue;
            real f = c[i].x[0] + (real)0.618 * c[i].x[1] + (real)0.382 * c[i].x[2];
            if (f > bestf) { bestf = f; k0 = i; }
        }
    }
    int k1 = -1, k2 = -1, k3 = -1;
    real e[3] = {0, 0, 0}, cr2[3] = {0, 0, 0};
    for (int tier = 0; tier < 2 && k1 < 0; tier++) {
        real best = -1;
        for (int i = 0; i < n; i++) {
            if (i == k0 || (tier == 0 && !(c[i].s < lim))) continue;
            real d[3]; v3_sub(d, c[i].x, c[k0].x);
            real v = v3_dot(d, d);
            if (v > best) { best = v; k1 = i; }
        }
    }
    v3_sub(e, c[k1].x, c[k0].x);
    for (int tier = 0; tier < 2 && k2 < 0; tier++) {
        real best = -1;
        for (int i = 0; i < n; i++) {
            if (i == k0 || i == k1 || (tier == 0 && !(c[i].s < lim))) continue;
            real d[3], cr[3]; v3_sub(d, c[i].x, c[k0].x);
            v3_cross(cr, d, e);
            real v = v3_dot(cr, cr);
            if (v > best) { best = v; k2 = i; v3_copy(cr2, cr); }
        }
    }
    sel[0] = k0; sel[1] = k1; sel[2] = k2;
    for (int tier = 0; tier < 2 && k3 < 0; tier++) {
        real best = 0;
        for (int i = 0; i < n; i++) {
            if (i == k0 || i == k1 || i == k2 || (tier == 0 && !(c[i].s < lim))) continue;
            real d[3], cr[3]; v3_sub(d, c[i].x, c[k0].x);
            v3_cross(cr, d, e);
            real v = -v3_dot(cr, cr2);
            if (v > best) { best = v; k3 = i; }
        }
    }
    if (k3 >= 0) { sel[3] = k3; return 4; }
    return 3;
}

static void collide_pair(rr_oracle *o, int sa, int sb, const xform_t *X) {
    const model_t *m = &o->m;
    const xform_t *Xa = &X[sa], *Xb = &X[sb];
    real ca[3], cb[3], d[3];
    m3_mulv(ca, Xa->R, m->sh_sphere[sa]); v3_add(ca, ca, Xa->p);
    m3_mulv(cb, Xb->R, m->sh_sphere[sb]); v3_add(cb, cb, Xb->p);
    v3_sub(d, ca, cb);
    real rr = m->sh_sphere[sa][3] + m->sh_sphere[sb][3] + (real)o->p.margin;
    if (v3_dot(d, d) > rr * rr) return;
    cand_t cand[CAND_MAX];
    int n = 0;
    n = verts_in_planes(o, sa, Xa, sb, Xb, (real)1, cand, n);   /* A's vertices in B: normal B->A */
    n = verts_in_planes(o, sb, Xb, sa, Xa, (real)-1, cand, n);  /* B's vertices in A: normal A->B, flipped */
    if (o->p.edge_contacts) n = edge_edge(o, sa, Xa, sb, Xb, cand, n);
    if (n == 0) return;
    int sel[4];
    int k = reduce4(cand, n, sel);
    for (int i = 0; i < k && o->ncontacts < MAXC; i++) {
        contact_t *c = &o->contacts[o->ncontacts++];
        c->bodyA = shape_body(m, sa);
        c->bodyB = shape_body(m, sb);
        c->linkA = m->sh_link[sa];
        v3_copy(c->x, cand[sel[i]].x);
        v3_copy(c->n, cand[sel[i]].n);
        c->dist = cand[sel[i]].s;
        c->mu = m->sh_fric[sa] * m->sh_fric[sb];
        c->rest = m->sh_rest[sa] * m->sh_rest[sb];
        /* btManifoldResult::calculateCombinedRollingFriction / SpinningFriction: r_a mu_b + r_b mu_a, clamped to 10
         * (URDF <rolling_friction>, <spinning_friction>: cube.urdf:6-7, tomato.urdf:6-7, mustard.urdf:6-7,
         * kuka_gripper.urdf:292-296 ...) */
        c->roll = m->sh_roll[sa] * m->sh_fric[sb] + m->sh_roll[sb] * m->sh_fric[sa];
        c->spin = m->sh_spin[sa] * m->sh_fric[sb] + m->sh_spin[sb] * m->sh_fric[sa];
        if (c->roll > 10) c->roll = 10;
        if (c->spin > 10) c->spin = 10;
        c->lambda_n = 0;
        c->lambda0 = 0;
    }
}

/* Warm starting (btPersistentManifold + SOLVER_USE_WARMSTARTING, m_warmstartingFactor 0.85).  Bullet keeps
 * up to four points per collision-object pair alive from step to step, matches a new point to the cached one nearest to it
 * within the contact breaking threshold (getCacheEntry) and starts the solver from 0.85 x its normal impulse
 * (btMultiBodyConstraintSolver::setupMultiBodyContactConstraint: friction and torsional rows start from zero).  Contacts
 * are regenerated from scratch here, so the cache is the previous step's contact list: a new contact inherits from the
 * previous contact of the same bodies (bodyA, bodyB, linkA) nearest to it in world space within the margin, provided no
 * other new contact of those bodies is nearer to that previous contact (mutual nearest neighbours: one heir per cached
 * point, independent of the order of evaluation). */
#define WARM_DIST2 ((real)0.0004)      /* (0.02 m)^2 */
static void warm_start_match(rr_oracle *o, const contact_t *prev, int nprev) {
    const real factor = (real)o->p.warmstart;
    const int n = o->ncontacts;
    for (int i = 0; i < n; i++) o->contacts[i].lambda0 = 0;
    if (!(factor > 0)) return;
    for (int i = 0; i < n; i++) {
        contact_t *c = &o->contacts[i];
        int bj = -1;
        real bd = WARM_DIST2;
        for (int j = 0; j < nprev; j++) {
            const contact_t *pc = &prev[j];
            if (pc->bodyA != c->bodyA || pc->bodyB != c->bodyB || pc->linkA != c->linkA) continue;
            real d[3];
            v3_sub(d, c->x, pc->x);
            real d2 = v3_dot(d, d);
            if (d2 < bd) { bd = d2; bj = j; }
        }
        if (bj < 0) continue;
        int heir = 1;
        for (int k = 0; k < n && heir; k++) {
            const contact_t *ck = &o->contacts[k];
            if (k == i || ck->bodyA != c->bodyA || ck->bodyB != c->bodyB || ck->linkA != c->linkA) continue;
            real d[3];
            v3_sub(d, ck->x, prev[bj].x);
            real d2 = v3_dot(d, d);
            if (d2 < bd || (d2 == bd && k < i)) heir = 0;
        }
        if (heir) c->lambda0 = factor * prev[bj].lambda_n;
    }
}

static void collide(rr_oracle *o) {
    const model_t *m = &o->m;
    xform_t X[MAXSHAPES];
    for (int s = 0; s < m->ns; s++) shape_xform(o, s, &X[s]);
    o->ncontacts = 0;
    int s_obj0 = m->n_static + m->n_robot;
    /* (A) object x static */
    for (int i = 0; i < o->nobj; i++)
        for (int s = 0; s < m->n_static; s++) collide_pair(o, s_obj0 + i, s, X);
    /* (B) object x object */
    for (int i = 0; i < o->nobj; i++)
        for (int j = i + 1; j < o->nobj; j++) collide_pair(o, s_obj0 + i, s_obj0 + j, X);
    /* (C) robot moving shapes x {table, shelf}  (robot self-collision incl. its own base link_0 is off) */
    for (int r = 0; r < m->n_robot; r++)
        for (int s = 0; s < 2; s++) collide_pair(o, m->n_static + r, s, X);
    /* (D) robot moving shapes x objects */
    for (int r = 0; r < m->n_robot; r++)
        for (int i = 0; i < o->nobj; i++) collide_pair(o, m->n_static + r, s_obj0 + i, X);
}

/* Narrow phase of ONE shape pair at the present state (tests/test_narrowphase_exact.py: the contacts the oracle emits for a
 * pair against the exact signed distance of the two full hulls).  Does not touch the env's contact list.  out: rro_contacts()
 * records; xf24: the two shapes' world transforms {R (9, row major), p (3)} x 2.  Returns the number of contacts. */
int rro_pair_contacts(rr_oracle *o, int sa, int sb, double *out, int maxc, double *xf24) {
    const model_t *m = &o->m;
    if (sa < 0 || sb < 0 || sa >= m->ns || sb >= m->ns) return -1;
    static __thread contact_t keep[MAXC];
    const int nkeep = o->ncontacts;
    memcpy(keep, o->contacts, sizeof(contact_t) * (size_t)nkeep);
    forward_kinematics(o);
    xform_t X[MAXSHAPES];
    for (int s = 0; s < m->ns; s++) shape_xform(o, s, &X[s]);
    o->ncontacts = 0;
    collide_pair(o, sa, sb, X);
    const int n = rro_contacts(o, out, maxc);
    memcpy(o->contacts, keep, sizeof(contact_t) * (size_t)nkeep);
    o->ncontacts = nkeep;
    if (xf24)
        for (int side = 0; side < 2; side++) {
            const xform_t *x = &X[side == 0 ? sa : sb];
            for (int k = 0; k < 9; k++) xf24[12 * side + k] = x->R[k];
            for (int k = 0; k < 3; k++) xf24[12 * side + 9 + k] = x->p[k];
        }
    return n;
}

/* ------------------------------------------------------------------------------------------- solver */
typedef struct {
    int bodyA, bodyB;
    real Ja[NB], MJa[NB];               /* robot part (either A or B side; at most one side is the robot) */
    real la[3], aa[3], mla[3], maa[3];  /* object A: linear/angular jacobian and M^-1 J^T */
    real lb[3], ab[3], mlb[3], mab[3];  /* object B */
    real rhs, dinv, lo, hi, lambda;
    int normal_row;                     /* friction rows: index of the normal row; -1 otherwise */
    real mu;
} row_t;

static void plane_space(const real *n, real *p, real *q) { /* btPlaneSpace1 */
    if (RFABS(n[2]) > (real)0.7071067811865475244) {
        real a = n[1] * n[1] + n[2] * n[2];
        real k = 1 / RSQRT(a);
        p[0] = 0; p[1] = -n[2] * k; p[2] = n[1] * k;
        q[0] = a * k; q[1] = -n[0] * p[2]; q[2] = n[0] * p[1];
    } else {
        real a = n[0] * n[0] + n[1] * n[1];
        real k = 1 / RSQRT(a);
        p[0] = -n[1] * k; p[1] = n[0] * k; p[2] = 0;
        q[0] = -n[2] * p[1]; q[1] = n[2] * p[0]; q[2] = a * k;
    }
}

/* Builds one contact row along direction `dir` at point x. Returns the relative velocity A-B along dir. */
static real build_row(const rr_oracle *o, row_t *r, const contact_t *c, const real *dir, const real Minv[NB][NB],
                      const real *qdstar, real ovs[NOBJ][3], real ows[NOBJ][3], real oIinv[NOBJ][9]) {
    const model_t *m = &o->m;
    memset(r, 0, sizeof *r);
    r->bodyA = c->bodyA; r->bodyB = c->bodyB;
    r->normal_row = -1;
    real diag = 0, rel = 0;
    for (int side = 0; side < 2; side++) {
        int body = side == 0 ? c->bodyA : c->bodyB;
        real sg = side == 0 ? (real)1 : (real)-1;
        if (body < 0) continue;
        if (body < 16) {
            int k = body;
            while (k >= 0) {
                real d[3], t[3];
                v3_sub(d, c->x, o->bp[k]);
                v3_cross(t, o->baxis[k], d);
                r->Ja[k] = sg * v3_dot(dir, t);
                k = m->parent[k];
            }
            for (int i = 0; i < NB; i++) {
                real s = 0;
                for (int j = 0; j < NB; j++) s += Minv[i][j] * r->Ja[j];
                r->MJa[i] = s;
            }
            for (int i = 0; i < NB; i++) { diag += r->Ja[i] * r->MJa[i]; rel += r->Ja[i] * qdstar[i]; }
        } else {
            int ob = body - 16;
            real rr[3], ang[3], lin[3], mang[3];
            v3_sub(rr, c->x, o->opos[ob]);
            v3_scale(lin, dir, sg);
            v3_cross(ang, rr, lin);
            m3_mulv(mang, oIinv[ob], ang);
            real *L = side == 0 ? r->la : r->lb, *A = side == 0 ? r->aa : r->ab;
            real *ML = side == 0 ? r->mla : r->mlb, *MA = side == 0 ? r->maa : r->mab;
            v3_copy(L, lin); v3_copy(A, ang);
            v3_scale(ML, lin, 1 / m->obj_mass[ob]);
            v3_copy(MA, mang);
            diag += v3_dot(L, ML) + v3_dot(A, MA);
            rel += v3_dot(L, ovs[ob]) + v3_dot(A, ows[ob]);
        }
    }
    r->dinv = diag > 0 ? 1 / diag : 0;
    return rel;
}

/* Torsional friction row (btMultiBodyConstraintSolver::setupMultiBodyTorsionalFrictionConstraint): a purely angular
 * Jacobian about `axis` -- joint j contributes axis_j . axis for the ancestors of the robot link, a free object
 * (0, +-axis).  Returns the relative angular velocity A - B about the axis. */
static real build_row_torsional(const rr_oracle *o, row_t *r, const contact_t *c, const real *axis, const real Minv[NB][NB],
                                const real *qdstar, real ows[NOBJ][3], real oIinv[NOBJ][9]) {
    const model_t *m = &o->m;
    memset(r, 0, sizeof *r);
    r->bodyA = c->bodyA; r->bodyB = c->bodyB;
    r->normal_row = -1;
    real diag = 0, rel = 0;
    for (int side = 0; side < 2; side++) {
        int body = side == 0 ? c->bodyA : c->bodyB;
        real sg = side == 0 ? (real)1 : (real)-1;
        if (body < 0) continue;
        if (body < 16) {
            int k = body;
            while (k >= 0) {
                r->Ja[k] = sg * v3_dot(axis, o->baxis[k]);
                k = m->parent[k];
            }
            for (int i = 0; i < NB; i++) {
                real s = 0;
                for (int j = 0; j < NB; j++) s += Minv[i][j] * r->Ja[j];
                r->MJa[i] = s;
            }
            for (int i = 0; i < NB; i++) { diag += r->Ja[i] * r->MJa[i]; rel += r->Ja[i] * qdstar[i]; }
        } else {
            int ob = body - 16;
            real ang[3], mang[3];
            v3_scale(ang, axis, sg);
            m3_mulv(mang, oIinv[ob], ang);
            real *A = side == 0 ? r->aa : r->ab, *MA = side == 0 ? r->maa : r->mab;
            v3_copy(A, ang);
            v3_copy(MA, mang);
            diag += v3_dot(A, MA);
            rel += v3_dot(A, ows[ob]);
        }
    }
    r->dinv = diag > 0 ? 1 / diag : 0;
    return rel;
}

/* the contact problem of the last rro_step (any oracle instance): rows, where the normal rows start, the unconstrained
 * velocities -- kept for rro_solution_residual() */
/* (thread-local: oracles stepped on a thread pool -- rro_run, tests/test_gpu_trajectory.py -- must not share work areas; the
 * residual check runs on the thread that stepped) */
static __thread row_t g_rows[NROWS];
static __thread int g_nr, g_first_normal, g_ncontacts;
static __thread real g_qds[NB], g_ovs[NOBJ][3], g_ows[NOBJ][3];

static void solve_and_integrate(rr_oracle *o) {
    const model_t *m = &o->m;
    const rro_params *P = &o->p;
    real dt = (real)P->dt;
    real M[NB][NB], bias[NB], Minv[NB][NB];
    mass_matrix_and_bias(o, M, bias);
    cholesky_inverse(M, Minv);
    /* unconstrained velocities */
    real qds[NB];
    for (int i = 0; i < NB; i++) {
        real s = 0;
        for (int j = 0; j < NB; j++) s += Minv[i][j] * (-bias[j] - m->damping[j] * o->qd[j]);
        qds[i] = o->qd[i] + dt * s;
    }
    real ovs[NOBJ][3], ows[NOBJ][3], oIinv[NOBJ][9];
    for (int i = 0; i < o->nobj; i++) {
        real R[9], Iw[9], I6[6] = {m->obj_inertia[i][0], m->obj_inertia[i][1], m->obj_inertia[i][2], 0, 0, 0};
        real Ii6[6] = {1 / m->obj_inertia[i][0], 1 / m->obj_inertia[i][1], 1 / m->obj_inertia[i][2], 0, 0, 0};
        quat_to_m3(R, o->oquat[i]);
        inertia_world(Iw, R, I6);
        inertia_world(oIinv[i], R, Ii6);
        real vn = RSQRT(v3_dot(o->ovel[i], o->ovel[i])), wn = RSQRT(v3_dot(o->oang[i], o->oang[i]));
        real kl = (real)P->lin_damping, ka = (real)P->ang_damping;
        for (int k = 0; k < 3; k++) ovs[i][k] = o->ovel[i][k] + dt * (-(o->ovel[i][k]) * (kl + kl * vn));
        ovs[i][2] -= dt * (real)P->gravity;
        real Iwv[3], g[3], al[3];
        m3_mulv(Iwv, Iw, o->oang[i]);
        v3_cross(g, o->oang[i], Iwv);
        m3_mulv(al, oIinv[i], g);
        for (int k = 0; k < 3; k++) ows[i][k] = o->oang[i][k] + dt * (-al[k] - o->oang[i][k] * (ka + ka * wn));
    }
    /* rows (file scope: rro_solution_residual() looks at the last step's problem) */
    row_t *rows = g_rows;
    int nr = 0;
    for (int j = 0; j < NB; j++) { /* btMultiBodyJointMotor position control */
        row_t *r = &rows[nr++];
        memset(r, 0, sizeof *r);
        r->bodyA = j; r->bodyB = -1; r->normal_row = -1;
        r->Ja[j] = 1;
        for (int i = 0; i < NB; i++) r->MJa[i] = Minv[i][j];
        real diag = Minv[j][j];
        r->dinv = 1 / diag;
        real vt = (real)P->motor_kp * (o->tgt[j] - o->q[j]) / dt + qds[j] + (real)P->motor_kd * (0 - qds[j]);
        r->rhs = (vt - qds[j]) * r->dinv;
        r->hi = (real)(P->motor_max_force * P->dt);
        r->lo = -r->hi;
    }
    /* joint limit rows (btMultiBodyJointLimitConstraint): only joints with lower < upper (the second finger
     * joints have lower 0 > upper -pi/2 in the URDF, kuka_gripper.urdf:322,397 -> no limit). A row is only
     * materialised when the limit is closer than RRO_LIMIT_WINDOW (it cannot become active otherwise). */
    for (int j = 0; j < NB; j++) {
        if (!(m->limits[j][0] < m->limits[j][1])) continue;
        for (int side = 0; side < 2; side++) {
            real dist = side == 0 ? o->q[j] - m->limits[j][0] : m->limits[j][1] - o->q[j];
            if (dist >= (real)0.5) continue;
            real sg = side == 0 ? (real)1 : (real)-1;
            row_t *r = &rows[nr++];
            memset(r, 0, sizeof *r);
            r->bodyA = j; r->bodyB = -1; r->normal_row = -1;
            r->Ja[j] = sg;
            for (int i = 0; i < NB; i++) r->MJa[i] = sg * Minv[i][j];
            r->dinv = 1 / Minv[j][j];
            real rel = sg * qds[j];
            real verr = -rel, perr = 0;
            if (dist > 0) verr -= dist / dt;
            else perr = -dist * (real)P->erp / dt;
            r->rhs = (perr + verr) * r->dinv;
            r->lo = 0; r->hi = (real)100;   /* btMultiBodyConstraint default m_maxAppliedImpulse */
        }
    }
    int first_normal = nr;
    for (int c = 0; c < o->ncontacts; c++) {
        contact_t *ct = &o->contacts[c];
        row_t *r = &rows[nr++];
        real rel = build_row(o, r, ct, ct->n, Minv, qds, ovs, ows, oIinv);
        real rest = 0;
        if (RFABS(rel) >= (real)P->rest_threshold) { rest = ct->rest * -rel; if (rest < 0) rest = 0; }
        real verr = rest - rel, perr = 0;
        if (ct->dist > 0) verr -= ct->dist / dt;
        else perr = -ct->dist * (real)P->erp / dt;
        r->rhs = (perr + verr) * r->dinv;
        r->lo = 0; r->hi = (real)1e10;
    }
    int first_fric = nr;
    for (int c = 0; c < o->ncontacts; c++) {
        contact_t *ct = &o->contacts[c];
        real t1[3], t2[3];
        plane_space(ct->n, t1, t2);
        for (int k = 0; k < 2; k++) {
            row_t *r = &rows[nr++];
            real rel = build_row(o, r, ct, k == 0 ? t1 : t2, Minv, qds, ovs, ows, oIinv);
            r->rhs = -rel * r->dinv;
            r->normal_row = first_normal + c;
            r->mu = ct->mu;
        }
    }
    (void)first_fric;
    /* torsional friction rows, after the lateral friction rows (Bullet keeps them in a list of their own, swept after the
     * friction list): per contact one spinning row about the normal when the combined spinning coefficient is positive
     * and two rolling rows about the tangents when the combined rolling coefficient is; bounds +- coefficient x normal
     * impulse, velocity target zero */
    for (int c = 0; c < o->ncontacts; c++) {
        contact_t *ct = &o->contacts[c];
        real t1[3], t2[3];
        plane_space(ct->n, t1, t2);
        for (int k = 0; k < 3; k++) {
            real coef = k == 0 ? ct->spin : ct->roll;
            if (!(coef > 0)) continue;
            row_t *r = &rows[nr++];
            real rel = build_row_torsional(o, r, ct, k == 0 ? ct->n : (k == 1 ? t1 : t2), Minv, qds, ows, oIinv);
            r->rhs = -rel * r->dinv;
            r->normal_row = first_normal + c;
            r->mu = coef;
        }
    }
    /* projected Gauss-Seidel on velocity deltas */
    real dq[NB], dv[NOBJ][3], dw[NOBJ][3];
    memset(dq, 0, sizeof dq); memset(dv, 0, sizeof dv); memset(dw, 0, sizeof dw);
    for (int c = 0; c < o->ncontacts; c++) {     /* warm start: the normal rows start from the inherited impulses, already applied */
        row_t *r = &rows[first_normal + c];
        real l0 = o->contacts[c].lambda0;
        if (!(l0 > 0)) continue;
        r->lambda = l0;
        int robot = (r->bodyA >= 0 && r->bodyA < 16) || (r->bodyB >= 0 && r->bodyB < 16);
        if (robot) for (int i = 0; i < NB; i++) dq[i] += r->MJa[i] * l0;
        if (r->bodyA >= 16) { int ob = r->bodyA - 16; v3_madd(dv[ob], r->mla, l0); v3_madd(dw[ob], r->maa, l0); }
        if (r->bodyB >= 16) { int ob = r->bodyB - 16; v3_madd(dv[ob], r->mlb, l0); v3_madd(dw[ob], r->mab, l0); }
    }
    for (int it = 0; it < P->solver_iters; it++) {
        for (int k = 0; k < nr; k++) {
            row_t *r = &rows[k];
            if (r->normal_row >= 0) {
                real ln = rows[r->normal_row].lambda;
                r->hi = r->mu * ln; r->lo = -r->hi;
            }
            real jv = 0;
            int robot = (r->bodyA >= 0 && r->bodyA < 16) || (r->bodyB >= 0 && r->bodyB < 16);
            if (robot) for (int i = 0; i < NB; i++) jv += r->Ja[i] * dq[i];
            if (r->bodyA >= 16) { int ob = r->bodyA - 16; jv += v3_dot(r->la, dv[ob]) + v3_dot(r->aa, dw[ob]); }
            if (r->bodyB >= 16) { int ob = r->bodyB - 16; jv += v3_dot(r->lb, dv[ob]) + v3_dot(r->ab, dw[ob]); }
            real dl = r->rhs - jv * r->dinv;
            real sum = r->lambda + dl;
            if (sum < r->lo) { dl = r->lo - r->lambda; sum = r->lo; }
            else if (sum > r->hi) { dl = r->hi - r->lambda; sum = r->hi; }
            r->lambda = sum;
            if (robot) for (int i = 0; i < NB; i++) dq[i] += r->MJa[i] * dl;
            if (r->bodyA >= 16) { int ob = r->bodyA - 16; v3_madd(dv[ob], r->mla, dl); v3_madd(dw[ob], r->maa, dl); }
            if (r->bodyB >= 16) { int ob = r->bodyB - 16; v3_madd(dv[ob], r->mlb, dl); v3_madd(dw[ob], r->mab, dl); }
        }
    }
    for (int c = 0; c < o->ncontacts; c++) o->contacts[c].lambda_n = rows[first_normal + c].lambda;
    g_nr = nr; g_first_normal = first_normal; g_ncontacts = o->ncontacts;
    memcpy(g_qds, qds, sizeof g_qds); memcpy(g_ovs, ovs, sizeof g_ovs); memcpy(g_ows, ows, sizeof g_ows);
    /* integrate (semi-implicit Euler) */
    for (int i = 0; i < NB; i++) {
        o->qd[i] = qds[i] + dq[i];
        o->q[i] += dt * o->qd[i];
    }
    for (int i = 0; i < o->nobj; i++) {
        for (int k = 0; k < 3; k++) {
            o->ovel[i][k] = ovs[i][k] + dv[i][k];
            o->oang[i][k] = ows[i][k] + dw[i][k];
            o->opos[i][k] += dt * o->ovel[i][k];
        }
        /* q <- exp(w dt) * q */
        real w = RSQRT(v3_dot(o->oang[i], o->oang[i]));
        real ang = w * dt;
        real dqv[4];
        if (ang > (real)1e-12) {
            real s = RSIN(ang / 2) / w;
            dqv[0] = o->oang[i][0] * s; dqv[1] = o->oang[i][1] * s; dqv[2] = o->oang[i][2] * s; dqv[3] = RCOS(ang / 2);
        } else {
            dqv[0] = o->oang[i][0] * dt / 2; dqv[1] = o->oang[i][1] * dt / 2; dqv[2] = o->oang[i][2] * dt / 2; dqv[3] = 1;
        }
        real *q = o->oquat[i], r[4];
        r[0] = dqv[3] * q[0] + dqv[0] * q[3] + dqv[1] * q[2] - dqv[2] * q[1];
        r[1] = dqv[3] * q[1] - dqv[0] * q[2] + dqv[1] * q[3] + dqv[2] * q[0];
        r[2] = dqv[3] * q[2] + dqv[0] * q[1] - dqv[1] * q[0] + dqv[2] * q[3];
        r[3] = dqv[3] * q[3] - dqv[0] * q[0] - dqv[1] * q[1] - dqv[2] * q[2];
        real nrm = RSQRT(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
        for (int k = 0; k < 4; k++) q[k] = r[k] / nrm;
    }
}

/* ------------------------------------------------------------------------------------------- step */
static void calc_state9(const rr_oracle *o, real *j9) { /* robot.py:203-211 */
    for (int i = 0; i < 7; i++) j9[i] = o->q[i];
    j9[7] = o->q[7];
    j9[8] = -o->q[8];
}

int rro_step(rr_oracle *o, const double *action9) {
    const model_t *m = &o->m;
    real a[9], cur[9];
    for (int i = 0; i < 9; i++) {
        double v = action9 ? action9[i] : 0.0;           /* env.py:333-334 */
        if (!isfinite(v)) return -1;                      /* robot.py:189 */
        a[i] = (real)v;
    }
    /* limitActionByJoint env.py:314-321 */
    calc_state9(o, cur);
    for (int i = 0; i < 9 && !o->p.no_rate_limit; i++) {
        real d = a[i] - cur[i];
        if (d > m->act_maxdiff[i]) d = m->act_maxdiff[i];
        if (d < -m->act_maxdiff[i]) d = -m->act_maxdiff[i];
        a[i] = cur[i] + d;
    }
    /* control_objects_limits env.py:257-264 (the table itself never triggers: z == 0.08, x == 0) */
    for (int i = 0; i < o->nobj; i++) {
        real x = o->opos[i][0], z = o->opos[i][2];
        if (z < m->table_pos[2] || (x > (real)0.11 && z < (real)0.29)) {
            v3_copy(o->opos[i], m->obj_pose0[i]);
            for (int k = 0; k < 4; k++) o->oquat[i][k] = m->obj_pose0[i][3 + k];
            v3_set(o->ovel[i], 0, 0, 0);   /* resetBasePositionAndOrientation zeroes the velocity */
            v3_set(o->oang[i], 0, 0, 0);
        }
    }
    /* apply_action robot.py:188-201 */
    for (int i = 0; i < 9; i++) {
        if (a[i] > m->act_max[i]) a[i] = m->act_max[i];
        if (a[i] < m->act_min[i]) a[i] = m->act_min[i];
    }
    {
        real hi = 2 * a[7], v = a[8];
        if (v > hi) v = hi;
        if (v < 0) v = 0;
        a[8] = v;
    }
    for (int i = 0; i < 7; i++) o->tgt[i] = a[i];
    o->tgt[7] = a[7]; o->tgt[9] = a[7];
    o->tgt[8] = -a[8]; o->tgt[10] = -a[8];
    /* scene.global_step() env.py:340 */
    forward_kinematics(o);
    {
        static __thread contact_t prev[MAXC];
        const int nprev = o->ncontacts;
        memcpy(prev, o->contacts, sizeof(contact_t) * (size_t)nprev);
        collide(o);
        warm_start_match(o, prev, nprev);
    }
    solve_and_integrate(o);
    /* touch sensors robot.py:152-163: max normal force over contacts of each skin link, |distance| < 0.1 */
    for (int k = 0; k < 4; k++) o->touch[k] = 0;
    for (int c = 0; c < o->ncontacts; c++) {
        const contact_t *ct = &o->contacts[c];
        if (ct->bodyA < 0 || ct->bodyA >= 16) continue;
        if (RFABS(ct->dist) >= (real)0.1) continue;
        for (int k = 0; k < 4; k++)
            if (ct->linkA == m->touch_links[k]) {
                real f = ct->lambda_n / (real)o->p.dt;
                if (f > o->touch[k]) o->touch[k] = f;
            }
    }
    o->timestep += 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------- accessors */
void rro_get_state(const rr_oracle *o, double *s) {
    int k = 0;
    for (int i = 0; i < NB; i++) s[k++] = o->q[i];
    for (int i = 0; i < NB; i++) s[k++] = o->qd[i];
    for (int i = 0; i < NOBJ; i++) {
        for (int j = 0; j < 3; j++) s[k++] = o->opos[i][j];
        for (int j = 0; j < 4; j++) s[k++] = o->oquat[i][j];
        for (int j = 0; j < 3; j++) s[k++] = o->ovel[i][j];
        for (int j = 0; j < 3; j++) s[k++] = o->oang[i][j];
    }
}
void rro_set_state(rr_oracle *o, const double *s) {
    int k = 0;
    for (int i = 0; i < NB; i++) o->q[i] = (real)s[k++];
    for (int i = 0; i < NB; i++) o->qd[i] = (real)s[k++];
    for (int i = 0; i < NOBJ; i++) {
        for (int j = 0; j < 3; j++) o->opos[i][j] = (real)s[k++];
        for (int j = 0; j < 4; j++) o->oquat[i][j] = (real)s[k++];
        for (int j = 0; j < 3; j++) o->ovel[i][j] = (real)s[k++];
        for (int j = 0; j < 3; j++) o->oang[i][j] = (real)s[k++];
    }
    o->ncontacts = 0;        /* a state set from outside has no contact history: the next step starts cold */
}
/* The contact cache of the warm start = the contact list of the previous step, in rro_contacts() layout
 * {bodyA, bodyB, linkA, x (3), n (3), dist, normal force, mu}: lets a differential test continue from another
 * simulator's state *and* history. */
void rro_set_contact_cache(rr_oracle *o, const double *rec, int n) {
    if (n > MAXC) n = MAXC;
    if (n < 0) n = 0;
    for (int c = 0; c < n; c++) {
        const double *r = rec + 12 * c;
        contact_t *ct = &o->contacts[c];
        memset(ct, 0, sizeof *ct);
        ct->bodyA = (int)r[0]; ct->bodyB = (int)r[1]; ct->linkA = (int)r[2];
        for (int k = 0; k < 3; k++) { ct->x[k] = (real)r[3 + k]; ct->n[k] = (real)r[6 + k]; }
        ct->dist = (real)r[9];
        ct->lambda_n = (real)r[10] * (real)o->p.dt;
        ct->mu = (real)r[11];
    }
    o->ncontacts = n;
}
void rro_get_obs(const rr_oracle *o, double *joints9, double *touch4, double *objpos) {
    real j9[9];
    calc_state9(o, j9);
    for (int i = 0; i < 9; i++) joints9[i] = j9[i];
    for (int i = 0; i < 4; i++) touch4[i] = o->touch[i];
    for (int i = 0; i < o->nobj; i++)
        for (int k = 0; k < 3; k++) objpos[3 * i + k] = o->opos[i][k];
}
int rro_timestep(const rr_oracle *o) { return o->timestep; }

/* T steps in one call (free-running statistics, tests/test_gpu_trajectory.py: one ctypes call per env, so that a thread pool scales):
 * actions [T][9]; per step the contact count, the robot-involving contact count and the four touch sensors; every `state_every`-th
 * step (t + 1 divisible by it) the state.  Returns the number of states written, or -1 - t when action t is not finite. */
int rro_run(rr_oracle *o, const double *actions, int T, int *ncontacts, int *nrobot, double *touch, int state_every, double *states) {
    int ns = 0;
    for (int t = 0; t < T; t++) {
        if (rro_step(o, actions + 9 * t) != 0) return -1 - t;
        int nr = 0;
        for (int c = 0; c < o->ncontacts; c++) nr += o->contacts[c].bodyA >= 0 && o->contacts[c].bodyA < 16;
        if (ncontacts) ncontacts[t] = o->ncontacts;
        if (nrobot) nrobot[t] = nr;
        if (touch) for (int i = 0; i < 4; i++) touch[4 * t + i] = o->touch[i];
        if (states && state_every > 0 && (t + 1) % state_every == 0) rro_get_state(o, states + (size_t)RRO_STATE * ns++);
    }
    return ns;
}

void rro_link_pose(const rr_oracle *oc, int link, double *pose7) {
    rr_oracle *o = (rr_oracle *)oc;
    const model_t *m = &o->m;
    forward_kinematics(o);
    int b = m->link_body[link];
    real R[9], p[3], q[4];
    if (b < 0) {
        memcpy(R, m->link_rot[link], sizeof R);
        v3_add(p, m->robot_pos, m->link_pos[link]);
    } else {
        m3_mul(R, o->bR[b], m->link_rot[link]);
        m3_mulv(p, o->bR[b], m->link_pos[link]);
        v3_add(p, p, o->bp[b]);
    }
    m3_to_quat(q, R);
    for (int k = 0; k < 3; k++) pose7[k] = p[k];
    for (int k = 0; k < 4; k++) pose7[3 + k] = q[k];
}

int rro_contacts(const rr_oracle *o, double *out, int maxc) {
    int n = o->ncontacts < maxc ? o->ncontacts : maxc;
    for (int c = 0; c < n; c++) {
        const contact_t *ct = &o->contacts[c];
        double *r = out + 12 * c;
        r[0] = ct->bodyA; r[1] = ct->bodyB; r[2] = ct->linkA;
        for (int k = 0; k < 3; k++) { r[3 + k] = ct->x[k]; r[6 + k] = ct->n[k]; }
        r[9] = ct->dist; r[10] = ct->lambda_n / o->p.dt; r[11] = ct->mu;
    }
    return n;
}

/* Solver-independent check of a candidate solution of the LAST step's contact problem (differential tests): given the
 * velocities after the step (state61 layout) and the normal force of every contact, the natural residual of the normal
 * rows  r_c = | max(lambda_c + rhs_c - dinv_c J_c (v - v*), 0) - lambda_c |  -- zero for an exact solution of the
 * complementarity problem, and what one more Gauss-Seidel update of the row would change -- plus the force sum and the
 * active set.  out[0] sum r_c / dt (N), out[1] max r_c / dt, out[2] sum of normal forces (N), out[3] largest normal
 * force, out[4] number of contacts with force > active_thresh; active_bits: bit c set for those. Returns the number of
 * contacts of the last step (-1: n does not match). */
int rro_solution_residual(const rr_oracle *o, const double *state_after61, const double *normal_force, int n,
                          double active_thresh, double *out5, uint64_t *active_bits) {
    if (n != g_ncontacts) return -1;
    real dq[NB], dv[NOBJ][3], dw[NOBJ][3];
    for (int i = 0; i < NB; i++) dq[i] = (real)state_after61[NB + i] - g_qds[i];
    for (int i = 0; i < NOBJ; i++)
        for (int k = 0; k < 3; k++) {
            dv[i][k] = i < o->nobj ? (real)state_after61[2 * NB + 13 * i + 7 + k] - g_ovs[i][k] : 0;
            dw[i][k] = i < o->nobj ? (real)state_after61[2 * NB + 13 * i + 10 + k] - g_ows[i][k] : 0;
        }
    double sum = 0, mx = 0, fsum = 0, fmax = 0;
    int nact = 0;
    uint64_t bits = 0;
    const double dt = o->p.dt;
    for (int c = 0; c < n; c++) {
        const row_t *r = &g_rows[g_first_normal + c];
        double jv = 0;
        int robot = (r->bodyA >= 0 && r->bodyA < 16) || (r->bodyB >= 0 && r->bodyB < 16);
        if (robot) for (int i = 0; i < NB; i++) jv += (double)r->Ja[i] * dq[i];
        if (r->bodyA >= 16) { int ob = r->bodyA - 16; jv += v3_dot(r->la, dv[ob]) + v3_dot(r->aa, dw[ob]); }
        if (r->bodyB >= 16) { int ob = r->bodyB - 16; jv += v3_dot(r->lb, dv[ob]) + v3_dot(r->ab, dw[ob]); }
        const double lam = normal_force[c] * dt;
        double nxt = lam + (double)r->rhs - jv * (double)r->dinv;
        if (nxt < 0) nxt = 0;
        const double res = fabs(nxt - lam) / dt;
        sum += res; if (res > mx) mx = res;
        fsum += normal_force[c]; if (normal_force[c] > fmax) fmax = normal_force[c];
        if (normal_force[c] > active_thresh) { nact++; bits |= 1ull << c; }
    }
    out5[0] = sum; out5[1] = mx; out5[2] = fsum; out5[3] = fmax; out5[4] = nact;
    if (active_bits) *active_bits = bits;
    return n;
}

void rro_set_object_pose(rr_oracle *o, int obj, const double *pose7) {
    for (int k = 0; k < 3; k++) o->opos[obj][k] = (real)pose7[k];
    for (int k = 0; k < 4; k++) o->oquat[obj][k] = (real)pose7[3 + k];
    v3_set(o->ovel[obj], 0, 0, 0);
    v3_set(o->oang[obj], 0, 0, 0);
}

void rro_mass_matrix(rr_oracle *o, double *M121, double *bias11) {
    real M[NB][NB], bias[NB];
    forward_kinematics(o);
    mass_matrix_and_bias(o, M, bias);
    for (int i = 0; i < NB; i++) {
        bias11[i] = bias[i];
        for (int j = 0; j < NB; j++) M121[i * NB + j] = M[i][j];
    }
}

/* ------------------------------------------------------------------------------------------- raster */
/* TinyRenderer-style software rasteriser, restated in FLOAT regardless of `real` (coverage decisions are
 * discontinuous, so the checker uses the arithmetic width of the thing it checks).
 *   camera   : look-at eye (0.01,0,1.2) -> table position, up (0,0,1); GL perspective fov 80 deg (vertical),
 *              aspect W/H, near 0.1, far 100                                  env.py:136-141,249-255,543-551
 *   sampling : integer pixel coordinates in a (x+1)*W/2 viewport, image row = H-1-y   (TinyRenderer viewport())
 *   shading  : ambient 0.6 + diffuse 0.35 max(0,n.l) + specular 0.05 max(r.z,0)^2, light dir (-50,30,100),
 *              nearest texel x instance colour, background white, depth = GL depth in [0,1], mask = body uid / -1
 *   ties     : equal depth -> lowest triangle id wins
 *   clipping : triangles that cross the near plane (w = 0.1) are clipped against it (clip_near), fragments outside the depth
 *              range are discarded
 */
typedef struct { float m[16]; } mat4;

static void camera_matrices(const rr_oracle *o, float *VP /*16 row-major*/) {
    if (o->cam_custom) { memcpy(VP, o->cam_VP, sizeof o->cam_VP); return; }
    float eye[3] = {0.01f, 0.0f, 1.2f};
    float tgt[3] = {(float)o->m.table_pos[0], (float)o->m.table_pos[1], (float)o->m.table_pos[2]};
    float up[3] = {0, 0, 1};
    float f[3] = {tgt[0] - eye[0], tgt[1] - eye[1], tgt[2] - eye[2]};
    float fl = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    f[0] /= fl; f[1] /= fl; f[2] /= fl;
    float s[3] = {f[1] * up[2] - f[2] * up[1], f[2] * up[0] - f[0] * up[2], f[0] * up[1] - f[1] * up[0]};
    float sl = sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    s[0] /= sl; s[1] /= sl; s[2] /= sl;
    float u[3] = {s[1] * f[2] - s[2] * f[1], s[2] * f[0] - s[0] * f[2], s[0] * f[1] - s[1] * f[0]};
    float V[16] = {s[0], s[1], s[2], -(s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2]),
                   u[0], u[1], u[2], -(u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2]),
                   -f[0], -f[1], -f[2], (f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2]),
                   0, 0, 0, 1};
    float fov = 80.0f, nearv = 0.1f, farv = 100.0f;
    float aspect = (float)o->W / (float)o->H;
    float yscale = 1.0f / tanf(fov * 3.14159265358979323846f / 360.0f);
    float xscale = yscale / aspect;
    float P[16] = {xscale, 0, 0, 0,
                   0, yscale, 0, 0,
                   0, 0, (nearv + farv) / (nearv - farv), 2 * nearv * farv / (nearv - farv),
                   0, 0, -1, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float a = 0;
            for (int k = 0; k < 4; k++) a += P[4 * i + k] * V[4 * k + j];
            VP[4 * i + j] = a;
        }
}

/* Replaces the eye camera by an arbitrary one: row-major 4x4 OpenGL view and projection matrices (float), VP = proj * view
 * with the same summation order as the product of the default camera above.  EnvCamera of render('rgb_array'):
 * computeViewMatrixFromYawPitchRoll / computeProjectionMatrixFOV, env.py:480-499. */
void rro_set_camera(rr_oracle *o, const float *view16, const float *proj16) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float a = 0;
            for (int k = 0; k < 4; k++) a += proj16[4 * i + k] * view16[4 * k + j];
            o->cam_VP[4 * i + j] = a;
        }
    o->cam_custom = 1;
}

static void instance_xform(const rr_oracle *o, int inst, float *R, float *p) {
    const model_t *m = &o->m;
    real Rr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pr[3] = {0, 0, 0};
    if (m->in_otype[inst] == 1) { memcpy(Rr, o->bR[m->in_oidx[inst]], sizeof Rr); v3_copy(pr, o->bp[m->in_oidx[inst]]); }
    else if (m->in_otype[inst] == 2) { quat_to_m3(Rr, o->oquat[m->in_oidx[inst]]); v3_copy(pr, o->opos[m->in_oidx[inst]]); }
    for (int k = 0; k < 9; k++) R[k] = (float)Rr[k];
    for (int k = 0; k < 3; k++) p[k] = (float)pr[k];
}

typedef struct { float sx[3], sy[3], sz[3], w[3]; int ok; } stri_t;
#define NEAR_W 0.1f      /* clip-space w of the near plane (env.py:548-551 nearVal 0.1) */

/* clip coordinates of the three corners: c[k] = {x, y, z, w} */
static void clip_coords(const float *MVP, const float *tp, float c[3][4]) {
    for (int k = 0; k < 3; k++) {
        const float *v = tp + 3 * k;
        c[k][0] = fmaf(MVP[0], v[0], fmaf(MVP[1], v[1], fmaf(MVP[2], v[2], MVP[3])));
        c[k][1] = fmaf(MVP[4], v[0], fmaf(MVP[5], v[1], fmaf(MVP[6], v[2], MVP[7])));
        c[k][2] = fmaf(MVP[8], v[0], fmaf(MVP[9], v[1], fmaf(MVP[10], v[2], MVP[11])));
        c[k][3] = fmaf(MVP[12], v[0], fmaf(MVP[13], v[1], fmaf(MVP[14], v[2], MVP[15])));
    }
}
/* perspective division + viewport of one clip-space vertex (TinyRenderer viewport(): integer sample points in a
 * (x+1)*W/2 window) */
static void to_screen(const float *c, int W, int H, float *sx, float *sy, float *sz, float *w) {
    float iw = 1.0f / c[3];
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    *sx = fmaf(c[0] * iw, hw, hw);          /* (x / w + 1) * W / 2 */
    *sy = fmaf(c[1] * iw, hh, hh);
    *sz = c[2] * iw;
    *w = c[3];
}
static void project_tri(const float *MVP, const float *tp, int W, int H, stri_t *s) {
    float c[3][4];
    clip_coords(MVP, tp, c);
    s->ok = 1;
    for (int k = 0; k < 3; k++) {
        if (c[k][3] < NEAR_W) { s->ok = 0; return; }
        to_screen(c[k], W, H, &s->sx[k], &s->sy[k], &s->sz[k], &s->w[k]);
    }
}
/* Sutherland-Hodgman against the near plane w >= NEAR_W: a triangle with one or two corners nearer than the plane becomes
 * a triangle or a quadrilateral (a fan of two triangles) -- TinyRenderer clips the triangles that cross the eye plane and
 * discards the fragments nearer than the near plane, which is the same coverage.  An intersection is always computed from
 * the inside corner a towards the outside corner b, t = (near - w_a) / (w_b - w_a), and gets w = near exactly.
 * Returns the number of polygon vertices (0, 3 or 4) in out[][4]. */
static int clip_near(float c[3][4], float out[4][4]) {
    int n = 0;
    for (int i = 0; i < 3; i++) {
        int j = (i + 1) % 3;
        int in_i = c[i][3] >= NEAR_W, in_j = c[j][3] >= NEAR_W;
        if (in_i) { for (int k = 0; k < 4; k++) out[n][k] = c[i][k]; n++; }
        if (in_i != in_j) {
            const float *a = in_i ? c[i] : c[j], *b = in_i ? c[j] : c[i];
            float t = (NEAR_W - a[3]) / (b[3] - a[3]);
            out[n][0] = fmaf(t, b[0] - a[0], a[0]);
            out[n][1] = fmaf(t, b[1] - a[1], a[1]);
            out[n][2] = fmaf(t, b[2] - a[2], a[2]);
            out[n][3] = NEAR_W;
            n++;
        }
    }
    return n;
}

static inline int bary(const stri_t *s, float px, float py, float *b) {
    float x0 = s->sx[0], y0 = s->sy[0], x1 = s->sx[1], y1 = s->sy[1], x2 = s->sx[2], y2 = s->sy[2];
    /* every product-difference is one multiplication and one fused multiply-add (a C fmaf and the GPU's v_fma_f32 round
     * identically, so the HIP path reproduces these bits with half the instructions of the unfused form) */
    float area = fmaf(x1 - x0, y2 - y0, -((x2 - x0) * (y1 - y0)));
    if (fabsf(area) < 1e-12f) return 0;
    float ia = 1.0f / area;
    b[0] = fmaf(x1 - px, y2 - py, -((x2 - px) * (y1 - py))) * ia;
    b[1] = fmaf(x2 - px, y0 - py, -((x0 - px) * (y2 - py))) * ia;
    b[2] = 1.0f - b[0] - b[1];
    return b[0] >= 0 && b[1] >= 0 && b[2] >= 0;
}

int32_t *rro_debug_tri = NULL;   /* optional [H*W] buffer receiving the winning triangle id (tests/debug) */
void rro_set_debug_tri(int32_t *p) { rro_debug_tri = p; }
/* optional work histogram of the visibility pass (sizing of the rasteriser, scratch/raster_hist.py): for triangles of moving
 * instances, by the number n of sample points in the clipped bounding box -- bucket 0: n = 0, 1: 1, 2: 2, 3: 3..4, 4: 5..8,
 * 5: 9..16, 6: 17..64, 7: > 64 -- h[b] triangles, h[8 + b] sample points tested, h[16 + b] sample points covered */
int64_t *rro_debug_hist = NULL;
void rro_set_debug_hist(int64_t *p) { rro_debug_hist = p; }
static int hist_bucket(int n) { return n <= 0 ? 0 : n == 1 ? 1 : n == 2 ? 2 : n <= 4 ? 3 : n <= 8 ? 4 : n <= 16 ? 5 : n <= 64 ? 6 : 7; }
void rro_render(rr_oracle *o, uint8_t *rgb, float *depth, int32_t *mask) {
    const model_t *m = &o->m;
    int W = o->W, H = o->H;
    forward_kinematics(o);
    float VP[16];
    camera_matrices(o, VP);
    uint64_t *vis = (uint64_t *)malloc(sizeof(uint64_t) * W * H);
    for (int i = 0; i < W * H; i++) vis[i] = ~0ull;
    static __thread float MVPs[MAXINST][16], Rs[MAXINST][9];
    for (int i = 0; i < m->ni; i++) {
        float p[3];
        instance_xform(o, i, Rs[i], p);
        float Mm[16] = {Rs[i][0], Rs[i][1], Rs[i][2], p[0], Rs[i][3], Rs[i][4], Rs[i][5], p[1],
                        Rs[i][6], Rs[i][7], Rs[i][8], p[2], 0, 0, 0, 1};
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++) {
                float a = 0;
                for (int k = 0; k < 4; k++) a += VP[4 * r + k] * Mm[4 * k + c];
                MVPs[i][4 * r + c] = a;
            }
    }
    int n_inst_used = m->ni - (NOBJ - o->nobj);
    for (int t = 0; t < m->nt; t++) {
        int inst = m->tri_inst[t];
        if (inst >= n_inst_used) continue;   /* unused objects are the trailing instances */
        float cc[3][4], poly[4][4];
        clip_coords(MVPs[inst], m->tri_pos + 9 * t, cc);
        int n_in = (cc[0][3] >= NEAR_W) + (cc[1][3] >= NEAR_W) + (cc[2][3] >= NEAR_W);
        if (n_in == 0) continue;
        int npoly = 3;
        if (n_in == 3) memcpy(poly, cc, sizeof cc);
        else npoly = clip_near(cc, poly);
        for (int sub = 0; sub + 2 < npoly; sub++) {      /* fan: (0, 1, 2), (0, 2, 3) */
            stri_t s;
            const int vi[3] = {0, sub + 1, sub + 2};
            for (int k = 0; k < 3; k++) to_screen(poly[vi[k]], W, H, &s.sx[k], &s.sy[k], &s.sz[k], &s.w[k]);
            float xmin = fminf(s.sx[0], fminf(s.sx[1], s.sx[2])), xmax = fmaxf(s.sx[0], fmaxf(s.sx[1], s.sx[2]));
            float ymin = fminf(s.sy[0], fminf(s.sy[1], s.sy[2])), ymax = fmaxf(s.sy[0], fmaxf(s.sy[1], s.sy[2]));
            if (xmax < 0 || ymax < 0 || xmin > (float)(W - 1) || ymin > (float)(H - 1)) continue;
            int x0 = (int)ceilf(fmaxf(xmin, 0.0f)), x1 = (int)floorf(fminf(xmax, (float)(W - 1)));
            int y0 = (int)ceilf(fmaxf(ymin, 0.0f)), y1 = (int)floorf(fminf(ymax, (float)(H - 1)));
            int hb = -1;
            if (rro_debug_hist && inst >= 3) {
                int n = (x1 >= x0 && y1 >= y0) ? (x1 - x0 + 1) * (y1 - y0 + 1) : 0;
                hb = hist_bucket(n);
                rro_debug_hist[hb]++; rro_debug_hist[8 + hb] += n;
            }
            for (int py = y0; py <= y1; py++)
                for (int px = x0; px <= x1; px++) {
                    float b[3];
                    if (!bary(&s, (float)px, (float)py, b)) continue;
                    if (hb >= 0) rro_debug_hist[16 + hb]++;
                    float z = fmaf(b[0], s.sz[0], fmaf(b[1], s.sz[1], b[2] * s.sz[2]));
                    float d = fmaf(0.5f, z, 0.5f);
                    if (!(d >= 0.0f && d <= 1.0f)) continue;
                    uint32_t db;
                    memcpy(&db, &d, 4);
                    uint64_t key = ((uint64_t)db << 32) | (uint32_t)t;
                    int idx = (H - 1 - py) * W + px;
                    if (key < vis[idx]) vis[idx] = key;
                }
        }
    }
    /* resolve / shade */
    float L[3] = {-50.0f, 30.0f, 100.0f};
    float ll = sqrtf(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
    L[0] /= ll; L[1] /= ll; L[2] /= ll;
    for (int row = 0; row < H; row++)
        for (int px = 0; px < W; px++) {
            int idx = row * W + px;
            uint64_t key = vis[idx];
            if (rro_debug_tri) rro_debug_tri[idx] = key == ~0ull ? -1 : (int32_t)(key & 0xffffffffu);
            if (key == ~0ull) {
                rgb[3 * idx] = rgb[3 * idx + 1] = rgb[3 * idx + 2] = 255;
                depth[idx] = 1.0f;
                mask[idx] = -1;
                continue;
            }
            uint32_t db = (uint32_t)(key >> 32);
            int t = (int)(key & 0xffffffffu);
            float d;
            memcpy(&d, &db, 4);
            int inst = m->tri_inst[t];
            stri_t s;
            project_tri(MVPs[inst], m->tri_pos + 9 * t, W, H, &s);
            float c0, c1, c2;
            if (s.ok) {
                float b[3] = {0, 0, 0};
                bary(&s, (float)px, (float)(H - 1 - row), b);
                c0 = b[0] / s.w[0]; c1 = b[1] / s.w[1]; c2 = b[2] / s.w[2];
            } else {
                /* a corner is nearer than the near plane (the triangle was clipped): perspective-correct weights straight
                 * from the clip coordinates -- the point sum c_i V_i projects onto the sample iff c is orthogonal to
                 * u_i = x_i - xn w_i and v_i = y_i - yn w_i, i.e. c ~ u x v */
                float cc[3][4];
                clip_coords(MVPs[inst], m->tri_pos + 9 * t, cc);
                float xn = (float)px * (2.0f / (float)W) - 1.0f, yn = (float)(H - 1 - row) * (2.0f / (float)H) - 1.0f;
                float u0 = cc[0][0] - xn * cc[0][3], u1 = cc[1][0] - xn * cc[1][3], u2 = cc[2][0] - xn * cc[2][3];
                float v0 = cc[0][1] - yn * cc[0][3], v1 = cc[1][1] - yn * cc[1][3], v2 = cc[2][1] - yn * cc[2][3];
                c0 = u1 * v2 - u2 * v1; c1 = u2 * v0 - u0 * v2; c2 = u0 * v1 - u1 * v0;
            }
            float cs = 1.0f / (c0 + c1 + c2);
            c0 *= cs; c1 *= cs; c2 *= cs;
            const float *nn = m->tri_nrm + 9 * t, *uv = m->tri_uv + 6 * t;
            float nl[3] = {c0 * nn[0] + c1 * nn[3] + c2 * nn[6], c0 * nn[1] + c1 * nn[4] + c2 * nn[7], c0 * nn[2] + c1 * nn[5] + c2 * nn[8]};
            const float *R = Rs[inst];
            float nw[3] = {R[0] * nl[0] + R[1] * nl[1] + R[2] * nl[2], R[3] * nl[0] + R[4] * nl[1] + R[5] * nl[2],
                           R[6] * nl[0] + R[7] * nl[1] + R[8] * nl[2]};
            float nlen = sqrtf(nw[0] * nw[0] + nw[1] * nw[1] + nw[2] * nw[2]);
            if (nlen > 0) { nw[0] /= nlen; nw[1] /= nlen; nw[2] /= nlen; }
            float ndl = nw[0] * L[0] + nw[1] * L[1] + nw[2] * L[2];
            float diff = fmaxf(ndl, 0.0f);
            float rv[3] = {nw[0] * (2 * ndl) - L[0], nw[1] * (2 * ndl) - L[1], nw[2] * (2 * ndl) - L[2]};
            float rl = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
            float rz = rl > 0 ? fmaxf(rv[2] / rl, 0.0f) : 0.0f;
            float spec = rz * rz;
            float tex[3] = {255.0f, 255.0f, 255.0f};
            int tid = m->in_tex[inst];
            if (tid >= 0) {
                float u = c0 * uv[0] + c1 * uv[2] + c2 * uv[4], v = c0 * uv[1] + c1 * uv[3] + c2 * uv[5];
                u = u - floorf(u); v = v - floorf(v);
                int tw = m->tex_info[3 * tid + 1], th = m->tex_info[3 * tid + 2];
                int tx = (int)(u * (float)tw), ty = (int)(v * (float)th);
                if (tx > tw - 1) tx = tw - 1;
                if (ty > th - 1) ty = th - 1;
                const uint8_t *px4 = m->tex_data + 4 * ((size_t)m->tex_info[3 * tid] + (size_t)(th - 1 - ty) * tw + tx);
                tex[0] = px4[0]; tex[1] = px4[1]; tex[2] = px4[2];
            }
            float shade = 0.6f + 0.35f * diff + 0.05f * spec;
            for (int k = 0; k < 3; k++) {
                float c = tex[k] * m->in_color[inst][k] * shade;
                int ci = (int)c;
                if (ci > 255) ci = 255;
                rgb[3 * idx + k] = (uint8_t)ci;
            }
            depth[idx] = d;
            mask[idx] = m->in_uid[inst];
        }
    free(vis);
}
