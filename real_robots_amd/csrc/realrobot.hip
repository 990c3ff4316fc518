// realrobot.hip -- batched REALRobot env.step() for MI355X (gfx950): HIP kernels + C ABI (include/realrobot.h).
//
// One process per GPU; N independent envs per device.  State lives in HBM as SoA [field][env] fp32 so that a
// wavefront whose lanes are consecutive envs reads/writes 256-byte contiguous lines.  A step is these kernels
// (main stream unless noted; DESIGN.md 5 has the work-item maps and what bounds each of them):
//   k_prep_a        1 thread / env : action protocol (env.py:314-321, 257-264; robot.py:188-201), forward kinematics,
//                                    object terms (rotation, world inverse inertia, unconstrained velocities); launch order
//                                    of k_collide (last step's heavy envs first)
//   k_prep_b        1 thread / env : joint-space mass matrix (composite rigid bodies), bias (RNEA), Cholesky, M^-1,
//                                    unconstrained joint velocities -- side stream, beside k_collide
//   k_collide       4 wavefronts / env: bounding spheres -> close pairs, dealt out among the waves -> lane-per-vertex convex
//                                    tests + edge-edge pass -> <= 4 points / pair, listed in pair order; warm-start matching
//                                    against the previous step's list; classifies the env (light / heavy / very heavy)
//   k_solve         16 lanes / env : row assembly (motors, joint limits, contact normal + 2 friction + 3 torsional rows),
//                                    PGS with the object-vs-static rows in registers and the generic rows in a slot layout
//                                    streamed from global memory, semi-implicit Euler, touch sensors, observation pack
//                                    (robot.py:152-163,203-211); three launches in a rendering step: light envs (main
//                                    stream), heavy and very heavy envs (two side streams)
//   k_render_setup  1 thread / (env, instance): FK of the ancestor chain -> model-view-projection + shading constants
//   k_raster        1 workgroup / (env, tile): visibility only -- 64-bit atomic-min buffer (depth | triangle id) in LDS,
//                                    meshlet clusters, fragment list out (winners + pixels vacated since the last frame)
//   k_shade         deferred shading of the fragment lists; vacated / occluded pixels go back to the static layer
//                                    (the images persist in HBM: k_static_copy, the full copy, only runs for the first
//                                    frame; k_restore is the separate-pass variant kept for RR_SEPARATE_RESTORE)
//   k_render_list / k_raster_list  the same three stages for the envs of a heavy list (side streams): one list-walking
//                                    launch for a short list, list-walking visibility + the grid kernels for a long one
// The arithmetic restates what the reference delegates to pybullet.stepSimulation / getCameraImage
// (env.py:340, 536-567); the algorithm and its constants are specified in DESIGN.md and checked against
// oracle/rr_oracle.c by tests/ (never linked here).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/realrobot.h"

#define NB 11
#define NOBJ 3
#define MAXC 48
#define NLINK_MAX 24
#define MAXSHAPES 32
#define VMAXC 192        // caps of collision vertices / planes per shape (tools/compile_model.py stops earlier at 0.3 mm)
#define FMAXC 192
#define EMAXC 48         // long sharp hull edges per shape (tools/compile_model.py EMAX)
#define MAXINST 32
#define MAXPAIRS 96
#define NSTATE 61

// ---------------------------------------------------------------------------------------------- error handling
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIPCHK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) return fail(RR_EDEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------------------------------------- model (device view)
struct BodyParams {   // passed by value as kernel argument -> scalar loads, uniform across the wave
    int parent[NB];
    float jpos[NB][3], jrot[NB][9], axis[NB][3], mass[NB], com[NB][3], inertia[NB][6], damping[NB], limits[NB][2];
    float robot_pos[3];
    float obj_mass[NOBJ], obj_inertia[NOBJ][3], obj_pose0[NOBJ][7];
    float table_z;
    float act_min[9], act_max[9], act_maxdiff[9];
    int touch_links[4];
};

struct SimParams {
    int N, nobj, iters, npairs, ablate, small_area, os_cap, edge_contacts, heavy_min, heavy2_min, coop_build;
    float warmstart;   // Bullet's m_warmstartingFactor (0.85); 0: cold start every step   // os_cap: object-vs-static contacts per env with rows in LDS (<= OS_CAP)
    float dt, gravity, erp, margin, kp, kd, max_impulse, lin_damp, ang_damp, rest_thresh;
};

struct ShapeData {    // global memory, read uniformly
    int otype[MAXSHAPES], oidx[MAXSHAPES], link[MAXSHAPES], nv[MAXSHAPES], nf[MAXSHAPES];
    float verts[MAXSHAPES][VMAXC][3];
    float planes[MAXSHAPES][FMAXC][4];
    float sphere[MAXSHAPES][4];
    float fric[MAXSHAPES], rest[MAXSHAPES], roll[MAXSHAPES], spin[MAXSHAPES];
    int ne[MAXSHAPES];
    float edges[MAXSHAPES][EMAXC][12];    // long sharp hull edges: p0, p1 - p0, the two facet normals (owner frame)
    int pair_a[MAXPAIRS], pair_b[MAXPAIRS];
    int pair_meta[MAXPAIRS][4];      // {bodyA, bodyB, link of shape a, 0}: body = -1 static, 0..15 robot body, 16+i object i
    float pair_mat[MAXPAIRS][4];     // {friction, restitution} products of the two shapes, combined {rolling, spinning} friction
};

struct RenderModel {
    int ni, nt, W, H, tile_h, ntiles, first_dynamic_tri;
    int tile_w, ntx;         // raster tiles are tile_w x tile_h pixels, ntx of them across: full-width strips up to 128 columns, 64 x 64 squares above (rr_create)
    int in_otype[MAXINST], in_oidx[MAXINST], in_uid[MAXINST], in_tex[MAXINST], in_cull[MAXINST];
    int tile_xbits;          // bits of a column within a tile (2^tile_xbits >= tile_w); a row within a tile then fits 14 - tile_xbits bits (tile_w * tile_h <= 4096)
    unsigned w_magic;        // ceil(2^32 / tile_w): row of a pixel-in-tile index = __umulhi(index, w_magic), exact for index < 2^20 and tile_w <= 1024
    int any_cull;            // some in_cull is set (RR_CULL): the window loop looks the flag of its instance up only then
    float in_color[MAXINST][3];
    int tex_off[16], tex_w[16], tex_h[16];
    int link_body[NLINK_MAX];
    float link_pos[NLINK_MAX][3], link_rot[NLINK_MAX][9];
    int nl;
    float VP[16];
    float plane_norm[5];     // |xyz| of the frustum planes w+x, w-x, w+y, w-y, w (near) -- invariant under the rigid model matrices
    float tile_plane[256][8];  // per raster tile: NDC y of its first / last sample row and |xyz| of those two planes {ndc_a, nrm_a, ndc_b, nrm_b}, then the same for its first / last sample column
                             // (host side, frustum_plane_norms: two square roots and twenty multiply-adds per WAVE of the visibility pass otherwise)
};

// scratch slots (floats per env), one record per env [N][S_TOTAL]
enum {
    S_BR = 0,                    // 11*9
    S_BP = S_BR + 99,            // 11*3
    S_BAX = S_BP + 33,           // 11*3
    S_MINV = S_BAX + 33,         // 121
    S_QDS = S_MINV + 121,        // 11
    S_OR = S_QDS + 11,           // 3*9
    S_OIINV = S_OR + 27,         // 3*9
    S_OVS = S_OIINV + 27,        // 9
    S_OWS = S_OVS + 9,           // 9
    S_OP = S_OWS + 9,            // 9: object positions the collision pass uses (the home position when the out-of-bounds rule fires)
    S_TOTAL = S_OP + 9
};

// state slots (floats per env), SoA [slot][N]
enum {
    ST_Q = 0, ST_QD = 11, ST_OPOS = 22, ST_OQUAT = 31, ST_OVEL = 43, ST_OANG = 52, ST_TGT = 61, ST_TOTAL = 72
};

struct DevPtrs {
    float *state;      // [ST_TOTAL][N]
    float *scratch;    // [N][S_TOTAL]
    // Contact frame of the step being solved ("current") -- the list k_solve reads, the classes the three solve / render
    // launches select by -- and the frame the collision pass of the NEXT step fills ("next").  The host swaps the two when a
    // step starts (rr_step); the current list with the normal forces (cforce) is also the contact history of the warm start.
    float4 *clist;     // [N][MAXC][3]  the env's contacts in pair order {x, y, z, nx | ny, nz, distance, meta | mu, restitution, rolling,
                       // spinning}; meta = bodyA | bodyB << 8 | linkA << 16 (bytes; -1 static, 0..15 robot body, 16+i object i).
                       // Written by the env's k_collide workgroup (as clist_next), read by its k_solve group in one round trip.
    int *ccount;       // [N] number of contacts in clist
    int *ccount_pub, *class_pub;   // [N] RR_F_CONTACT_COUNT / RR_F_ENV_CLASS: the count and class of the last SOLVED step in fixed storage (the
                                   // frames change roles every step; a pointer handed out by rr_get_buffer must not) -- written by the solve kernels
    float *cwarm;      // [N][MAXC] initial normal impulse of every contact of clist (k_collide: 0.85 x the matched previous one)
    int *hgflag;       // [N] 0: light; 1: this env has generic contact rows this step -- "heavy"; 2: more than P.heavy2_min of them -- "very heavy"
    int *hlist2;       // [N] the very heavy envs of this step (a handful: an arm crushed onto the table at the contact cap)
    int *hcount2;      // [0] their number, [1] work counter of their render
    int *hlist;        // [N] the heavy envs of this step (in arrival order: placement only, never a result)
    int *hcount;       // [0] their number, [1] work counter of k_raster_list / k_render_list
    float4 *clist_next; int *ccount_next; float *cwarm_next; int *hgflag_next, *hlist2_next, *hcount2_next, *hlist_next, *hcount_next;
    int *hpos, *hpos_next;     // [N] place of a heavy / very heavy env in its list (k_collide's launch order, collide_launch_env)
    float *cforce;     // [N][MAXC] normal force of every contact of the last solved step (rr_get_contacts, touch sensors, warm start)
    int *hcount_host;  // device address of the pinned host word that receives the current number of heavy envs (or nullptr)
    int *timestep;     // [N]
    unsigned *errflags;// [N]
    float *obj_home;   // [NOBJ*7][N] per-env pose an object is put back to by reset / the out-of-bounds rule (robot.py:19-24, mutable there)
    float4 *grows;     // [N * GP_RECS][16] generic solver rows in the slot layout, as canonical normal rows and as the blocks the sweeps stream (GP_*)
    float *cmd;        // [N][9]
    const float *cmd_in; // [N][9] the command buffer of this step: cmd, or the caller's device buffer (read in place)
    float *joints;     // [N][9]
    float *touch;      // [N][4]
    float *objpose;    // [N][nobj][7]
    float *inst_xf;    // [N][MAXINST][32]  per-instance render constants: mvp (16), then R (9), colour (3), tex_off, tex_w, tex_h, uid
    unsigned char *render_flags; // [N]
    unsigned char *rgb; float *depth; int *mask;
    const float *tri_pos;   // SoA [9][NT]
    const float4 *tri_rec;  // AoS [NT][8]: one 128-byte shading record per triangle {pos[9], nrm[9], uv[6], inst, pad}
    const int *tri_inst;    // [NT]
    const float4 *cluster_sphere; // [NT/64] bounding sphere (instance frame) of each 64-triangle raster cluster
    const float *cluster_verts;   // [NT/64][3][64] the cluster's distinct vertex positions (x row, y row, z row)
    const int *tri_vidx;          // [NT] cluster-local vertex indices of the triangle's corners as ds_bpermute byte addresses: 4 v0 | 4 v1 << 8 | 4 v2 << 16
    const unsigned *tex;    // RGBX texels
    const ShapeData *shapes;
    const unsigned long long *static_vis;   // [H*W] visibility keys of the never-moving instances (or nullptr)
    unsigned long long *static_vis_out;
    unsigned char *static_rgb; float *static_depth; int *static_mask;   // [H*W] shaded static layer (shared by all envs)
    uint2 *frag_list;       // [N*ntiles][TILE_PIX] pixels won by moving triangles: {depth bits, pixel-in-tile << 18 | triangle}
    unsigned *frag_count;   // [N*ntiles]
    // dispatch order of k_raster's workgroups (raster_order_class): the (env, tile) items by falling cost of the previous frame
    unsigned *item_cost;        // [N*ntiles] duration of the item's workgroup in the last frame that rasterised it (100 MHz ticks)
    unsigned *item_bin;         // [N*ntiles] raster_order_class: the bin an item was counted in (its second pass reads it back, not the cost again)
    const unsigned *item_perm;  // [N*ntiles] env << 8 | tile, costly items first; nullptr: env-major grid (envs, tiles)
};

// Which envs a launch handles: 0 all; 1 the light envs; 2 the heavy ones; 3 the very heavy ones (rr_step runs the few heavy
// envs -- an arm pressed on the table, a gripper pushing objects: dozens of generic contact rows -- and their render on the
// side stream, beside the render of the others; the handful at the contact cap, whose solve takes half as long again, on
// a side stream of their own, so that the render of the other heavy envs does not wait for them).
__device__ __forceinline__ bool env_selected(const int *hgflag, int env, int sel) {
    if (sel == 0) return true;
    const int cls = hgflag[env];
    return sel == 1 ? cls == 0 : (sel == 2 ? cls == 1 : cls == 2);
}

// Kinematic tree of the 11 moving bodies (lbr_iiwa_link_1..7 [+gripper base], finger_00, finger_01, finger_10,
// finger_11); compile-time so that per-body register arrays are statically indexed. rr_create verifies the blob.
__device__ constexpr int PARENT[NB] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 6, 9};
static const int PARENT_HOST[NB] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 6, 9};

// ---------------------------------------------------------------------------------------------- device math
struct v3 { float x, y, z; };
__device__ __forceinline__ v3 mk(float x, float y, float z) { v3 r = {x, y, z}; return r; }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
struct m3 { float m[9]; };
__device__ __forceinline__ v3 mulv(const m3 &M, v3 v) {
    return mk(M.m[0] * v.x + M.m[1] * v.y + M.m[2] * v.z, M.m[3] * v.x + M.m[4] * v.y + M.m[5] * v.z,
              M.m[6] * v.x + M.m[7] * v.y + M.m[8] * v.z);
}
__device__ __forceinline__ v3 tmulv(const m3 &M, v3 v) {
    return mk(M.m[0] * v.x + M.m[3] * v.y + M.m[6] * v.z, M.m[1] * v.x + M.m[4] * v.y + M.m[7] * v.z,
              M.m[2] * v.x + M.m[5] * v.y + M.m[8] * v.z);
}
__device__ __forceinline__ m3 mul(const m3 &A, const m3 &B) {
    m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return r;
}
__device__ __forceinline__ m3 transpose(const m3 &A) {
    m3 r = {{A.m[0], A.m[3], A.m[6], A.m[1], A.m[4], A.m[7], A.m[2], A.m[5], A.m[8]}};
    return r;
}
__device__ __forceinline__ m3 axis_angle(v3 a, float ang) {
    float s, c;
    sincosf(ang, &s, &c);
    float t = 1.0f - c;
    m3 R = {{t * a.x * a.x + c, t * a.x * a.y - s * a.z, t * a.x * a.z + s * a.y,
             t * a.x * a.y + s * a.z, t * a.y * a.y + c, t * a.y * a.z - s * a.x,
             t * a.x * a.z - s * a.y, t * a.y * a.z + s * a.x, t * a.z * a.z + c}};
    return R;
}
__device__ __forceinline__ m3 quat_to_m3(float x, float y, float z, float w) {
    m3 R = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    return R;
}
__device__ __forceinline__ void m3_to_quat(const m3 &R, float *q) {
    const float *r = R.m;
    float t = r[0] + r[4] + r[8];
    if (t > 0) {
        float s = sqrtf(t + 1) * 2;
        q[3] = s / 4; q[0] = (r[7] - r[5]) / s; q[1] = (r[2] - r[6]) / s; q[2] = (r[3] - r[1]) / s;
    } else if (r[0] > r[4] && r[0] > r[8]) {
        float s = sqrtf(1 + r[0] - r[4] - r[8]) * 2;
        q[3] = (r[7] - r[5]) / s; q[0] = s / 4; q[1] = (r[1] + r[3]) / s; q[2] = (r[2] + r[6]) / s;
    } else if (r[4] > r[8]) {
        float s = sqrtf(1 + r[4] - r[0] - r[8]) * 2;
        q[3] = (r[2] - r[6]) / s; q[0] = (r[1] + r[3]) / s; q[1] = s / 4; q[2] = (r[5] + r[7]) / s;
    } else {
        float s = sqrtf(1 + r[8] - r[0] - r[4]) * 2;
        q[3] = (r[3] - r[1]) / s; q[0] = (r[2] + r[6]) / s; q[1] = (r[5] + r[7]) / s; q[2] = s / 4;
    }
}
// R diag(I6 as xx,yy,zz,xy,xz,yz) R^T
__device__ __forceinline__ m3 inertia_world(const m3 &R, const float *I6) {
    m3 I = {{I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]}};
    return mul(mul(R, I), transpose(R));
}


// ---- contraction-free twins ------------------------------------------------------------------------------------------
// Everything that decides WHICH contacts exist (forward kinematics, shape transforms, the sphere cull, vertex-in-polytope
// distances, the manifold reduction) is evaluated without FMA contraction and with the operation order of
// oracle/rr_oracle.c, and the joint rotations use det_sincosf below instead of the vendor sincosf: with the same state
// in, the candidate set `best < margin`, the arg-max plane and the "farthest point" picks of symmetric configurations (a
// cube flat on the table has four equally good corners) come out bit-identical to the oracle's float build.  clang
// attaches the `contract` flag to each fmul / fadd where it is written, so helpers inlined from outside a pragma region
// could still fuse: these twins are used instead of the operators above wherever that matters.
#pragma clang fp contract(off)
namespace nc {
__device__ __forceinline__ v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 scale(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ v3 cross(v3 a, v3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ v3 mulv(const m3 &M, v3 v) {
    return mk(M.m[0] * v.x + M.m[1] * v.y + M.m[2] * v.z, M.m[3] * v.x + M.m[4] * v.y + M.m[5] * v.z,
              M.m[6] * v.x + M.m[7] * v.y + M.m[8] * v.z);
}
__device__ __forceinline__ v3 tmulv(const m3 &M, v3 v) {
    return mk(M.m[0] * v.x + M.m[3] * v.y + M.m[6] * v.z, M.m[1] * v.x + M.m[4] * v.y + M.m[7] * v.z,
              M.m[2] * v.x + M.m[5] * v.y + M.m[8] * v.z);
}
__device__ __forceinline__ m3 mul(const m3 &A, const m3 &B) {
    m3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return r;
}
// The narrow phase's inner loops use EXPLICIT fused multiply-adds in a fixed association, mirrored by fma()/fmaf() calls in
// the oracle (verts_in_planes): v_fma_f32 and a C fmaf round identically, so the results stay bit-identical with half the
// instructions of the unfused form.
__device__ __forceinline__ v3 mulv_add_fma(const m3 &M, v3 v, v3 p) {       // M v + p
    return mk(__builtin_fmaf(M.m[0], v.x, __builtin_fmaf(M.m[1], v.y, __builtin_fmaf(M.m[2], v.z, p.x))),
              __builtin_fmaf(M.m[3], v.x, __builtin_fmaf(M.m[4], v.y, __builtin_fmaf(M.m[5], v.z, p.y))),
              __builtin_fmaf(M.m[6], v.x, __builtin_fmaf(M.m[7], v.y, __builtin_fmaf(M.m[8], v.z, p.z))));
}
__device__ __forceinline__ v3 tmulv_fma(const m3 &M, v3 d) {               // M^T d
    return mk(__builtin_fmaf(M.m[0], d.x, __builtin_fmaf(M.m[3], d.y, M.m[6] * d.z)),
              __builtin_fmaf(M.m[1], d.x, __builtin_fmaf(M.m[4], d.y, M.m[7] * d.z)),
              __builtin_fmaf(M.m[2], d.x, __builtin_fmaf(M.m[5], d.y, M.m[8] * d.z)));
}
__device__ __forceinline__ float plane_dist_fma(float4 pl, v3 x) {           // n . x - c
    return __builtin_fmaf(pl.x, x.x, __builtin_fmaf(pl.y, x.y, __builtin_fmaf(pl.z, x.z, -pl.w)));
}
// sin and cos of x (|x| up to a few turns: joint angles) from explicit IEEE single operations only -- the same sequence
// as det_sincosf in oracle/rr_oracle.c, hence bit-identical results on both sides (the vendor sincosf and glibc's differ
// in the last bit).  Cody-Waite reduction by pi/2 in three parts, Cephes minimax polynomials on [-pi/4, pi/4]; < 2 ulp.
__device__ __forceinline__ void det_sincosf(float x, float *sn, float *cs) {
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
__device__ __forceinline__ m3 axis_angle(v3 a, float ang) {
    float s, c;
    det_sincosf(ang, &s, &c);
    float t = 1.0f - c;
    m3 R = {{t * a.x * a.x + c, t * a.x * a.y - s * a.z, t * a.x * a.z + s * a.y,
             t * a.x * a.y + s * a.z, t * a.y * a.y + c, t * a.y * a.z - s * a.x,
             t * a.x * a.z - s * a.y, t * a.y * a.z + s * a.x, t * a.z * a.z + c}};
    return R;
}
__device__ __forceinline__ m3 quat_to_m3(float x, float y, float z, float w) {
    m3 R = {{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
             2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
             2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}};
    return R;
}
}  // namespace nc
#pragma clang fp contract(fast)

#define SCR(slot) scratch[(size_t)env * S_TOTAL + (slot)]      // per-env record [N][S_TOTAL]: the workgroup-per-env kernels (k_collide, k_solve) read it with contiguous lanes
#define STT(slot) state[(size_t)(slot) * N + env]

// Forward kinematics for all 11 bodies; results kept in registers/local arrays.  Contraction-free (see nc above): the
// transforms feed the collision tests, whose accept / reject decisions are compared bit for bit with the oracle.
__device__ void fk_all(const BodyParams &bp_, const float *q, m3 *bR, v3 *bp, v3 *bax) {
#pragma unroll
    for (int b = 0; b < NB; b++) {
        m3 Rp = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
        v3 pp = mk(bp_.robot_pos[0], bp_.robot_pos[1], bp_.robot_pos[2]);
        const int p = PARENT[b];
        if (p >= 0) { Rp = bR[p >= 0 ? p : 0]; pp = bp[p >= 0 ? p : 0]; }
        m3 jr;
#pragma unroll
        for (int k = 0; k < 9; k++) jr.m[k] = bp_.jrot[b][k];
        m3 Rj = nc::mul(Rp, jr);
        v3 ax = mk(bp_.axis[b][0], bp_.axis[b][1], bp_.axis[b][2]);
        bp[b] = nc::add(pp, nc::mulv(Rp, mk(bp_.jpos[b][0], bp_.jpos[b][1], bp_.jpos[b][2])));
        bR[b] = nc::mul(Rj, nc::axis_angle(ax, q[b]));
        bax[b] = nc::mulv(Rj, ax);
    }
}

// ---------------------------------------------------------------------------------------------- k_prep
// Which env a work item of a per-class launch handles: sel 0 all envs (idx = env); 1 the light envs (idx = env, others are
// skipped); 2 / 3 entry idx of the heavy / very heavy list of the CURRENT contact frame.  -1: nothing to do.
__device__ __forceinline__ int pick_env(const DevPtrs &D, int sel, int idx, int N) {
    if (idx >= N) return -1;
    if (sel == 0) return idx;
    if (sel == 1) return D.hgflag[idx] == 0 ? idx : -1;
    if (sel == 2) return idx < D.hcount[0] ? D.hlist[idx] : -1;
    return idx < D.hcount2[0] ? D.hlist2[idx] : -1;
}

// The out-of-bounds rule of control_objects_limits (env.py:257-264) as a function of the object's position.
__device__ __forceinline__ bool object_out_of_bounds(float x, float z, float table_z) { return z < table_z || (x > 0.11f && z < 0.29f); }

// The state part of the preparation -- everything of a step that does NOT need the action, i.e. a function of the state the
// previous step left: PHASE 1 = forward kinematics and object terms (rotation, world inverse inertia, unconstrained
// velocities; the out-of-bounds rule decides which object pose counts) -- all the collision kernel needs -- and PHASE 2 =
// joint-space dynamics (mass matrix, bias forces, Cholesky, M^-1, unconstrained velocities), which only the solver needs.
// Together with k_collide they form the LOOK-AHEAD of a step: rr_step launches them for step t+1 behind the solve of step t,
// class by class (sel, pick_env), under the render of step t; only when the state was changed from outside in between
// (reset, set_state, teleports) do they run in line at the start of the step.  (When the kernels are timed one by one the
// two phases run back to back and are reported together as "k_prep".)
template <int PHASE>
__device__ __forceinline__ void prep_body(const BodyParams &B, const SimParams &P, const DevPtrs &D, int env) {
    const int N = P.N;
    const float *state = D.state;
    float *scratch = D.scratch;
    if (D.errflags[env] & 1u) return;   // frozen env
    float q[NB], qd[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) { q[i] = STT(ST_Q + i); qd[i] = STT(ST_QD + i); }
    if (PHASE == 1) {
        // ---- kinematics only, streamed: a body's frame goes to the scratch record as soon as it exists and only the frame at hand
        // and the one the chain forks from are kept (fk_all's operations in fk_all's order: same bits).  The kernel then needs a
        // third of the registers (69 instead of 194 VGPRs).
        m3 Rc = {{1, 0, 0, 0, 1, 0, 0, 0, 1}}, Rf = Rc;
        v3 pc = mk(B.robot_pos[0], B.robot_pos[1], B.robot_pos[2]), pf = pc;
        constexpr int FORK = 6;                    // PARENT: a chain 0..8 with the second finger pair 9, 10 hanging off body 6
        static_assert(PARENT[9] == FORK && PARENT[10] == 9 && PARENT[8] == 7 && PARENT[7] == 6, "streamed forward kinematics");
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (b == 9) { Rc = Rf; pc = pf; }      // (bodies 0..8: the parent is the previous body; 9: body 6; 10: body 9)
            m3 jr;
#pragma unroll
            for (int k = 0; k < 9; k++) jr.m[k] = B.jrot[b][k];
            const m3 Rj = nc::mul(Rc, jr);
            const v3 ax = mk(B.axis[b][0], B.axis[b][1], B.axis[b][2]);
            const v3 pb = nc::add(pc, nc::mulv(Rc, mk(B.jpos[b][0], B.jpos[b][1], B.jpos[b][2])));
            const m3 Rb = nc::mul(Rj, nc::axis_angle(ax, q[b]));
            const v3 axw = nc::mulv(Rj, ax);
#pragma unroll
            for (int k = 0; k < 9; k++) SCR(S_BR + 9 * b + k) = Rb.m[k];
            SCR(S_BP + 3 * b) = pb.x; SCR(S_BP + 3 * b + 1) = pb.y; SCR(S_BP + 3 * b + 2) = pb.z;
            SCR(S_BAX + 3 * b) = axw.x; SCR(S_BAX + 3 * b + 1) = axw.y; SCR(S_BAX + 3 * b + 2) = axw.z;
            Rc = Rb; pc = pb;
            if (b == FORK) { Rf = Rb; pf = pb; }
        }
    }
    // ---- kinematics
    m3 bR[NB]; v3 bp[NB], bax[NB], bcom[NB]; m3 bI[NB];
    if (PHASE != 1) fk_all(B, q, bR, bp, bax);
#pragma unroll
    for (int b = 0; b < (PHASE == 1 ? 0 : NB); b++) {
        bcom[b] = bp[b] + mulv(bR[b], mk(B.com[b][0], B.com[b][1], B.com[b][2]));
        bI[b] = inertia_world(bR[b], B.inertia[b]);
        if (PHASE != 2) {
#pragma unroll
            for (int k = 0; k < 9; k++) SCR(S_BR + 9 * b + k) = bR[b].m[k];
            SCR(S_BP + 3 * b) = bp[b].x; SCR(S_BP + 3 * b + 1) = bp[b].y; SCR(S_BP + 3 * b + 2) = bp[b].z;
            SCR(S_BAX + 3 * b) = bax[b].x; SCR(S_BAX + 3 * b + 1) = bax[b].y; SCR(S_BAX + 3 * b + 2) = bax[b].z;
        }
    }
    if (PHASE != 1) {
    // ---- composite rigid body mass matrix
    float cm[NB]; v3 cc[NB]; m3 cI[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) { cm[b] = B.mass[b]; cc[b] = bcom[b]; cI[b] = bI[b]; }
#pragma unroll
    for (int b = NB - 1; b >= 1; b--) {
        const int p = PARENT[b];
        float mt = cm[p] + cm[b];
        v3 c = (cc[p] * cm[p] + cc[b] * cm[b]) * (1.0f / mt);
        v3 d1 = cc[p] - c, d2 = cc[b] - c;
        float s1 = dot(d1, d1), s2 = dot(d2, d2);
        float d1a[3] = {d1.x, d1.y, d1.z}, d2a[3] = {d2.x, d2.y, d2.z};
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                float e = (i == j) ? 1.0f : 0.0f;
                cI[p].m[3 * i + j] = cI[p].m[3 * i + j] + cI[b].m[3 * i + j] + cm[p] * (s1 * e - d1a[i] * d1a[j]) + cm[b] * (s2 * e - d2a[i] * d2a[j]);
            }
        cm[p] = mt;
        cc[p] = c;
    }
    float M[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
        for (int j = 0; j < NB; j++) M[i][j] = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        v3 Ia = mulv(cI[j], bax[j]);
        v3 f = cross(bax[j], cc[j] - bp[j]) * cm[j];
#pragma unroll
        for (int i = j; i >= 0; i--) {
            // ancestors-or-self of j: chain 0..6 is linear; fingers branch at body 6
            bool anc = (i == j) || (i <= 6 && j <= 6) || (i <= 6 && j >= 7) || (i == 7 && j == 8) || (i == 9 && j == 10);
            if (!anc) continue;
            v3 nn = Ia + cross(cc[j] - bp[i], f);
            float v = dot(bax[i], nn);
            M[i][j] = v; M[j][i] = v;
        }
    }
    // ---- RNEA bias (qdd = 0, base acceleration +g)
    v3 w[NB], al[NB], ap[NB], F[NB], Nn[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int p = PARENT[b];
        v3 wp = mk(0, 0, 0), alp = mk(0, 0, 0), app = mk(0, 0, P.gravity), pp = mk(B.robot_pos[0], B.robot_pos[1], B.robot_pos[2]);
        if (p >= 0) { wp = w[p >= 0 ? p : 0]; alp = al[p >= 0 ? p : 0]; app = ap[p >= 0 ? p : 0]; pp = bp[p >= 0 ? p : 0]; }
        w[b] = wp + bax[b] * qd[b];
        al[b] = alp + cross(wp, bax[b]) * qd[b];
        v3 d = bp[b] - pp;
        ap[b] = app + cross(alp, d) + cross(wp, cross(wp, d));
        v3 r = bcom[b] - bp[b];
        v3 ac = ap[b] + cross(al[b], r) + cross(w[b], cross(w[b], r));
        F[b] = ac * B.mass[b];
        Nn[b] = mulv(bI[b], al[b]) + cross(w[b], mulv(bI[b], w[b])) + cross(r, F[b]);
    }
    float bias[NB];
#pragma unroll
    for (int b = NB - 1; b >= 0; b--) {
        bias[b] = dot(bax[b], Nn[b]);
        const int p = PARENT[b];
        if (p >= 0) {
            const int pi = p >= 0 ? p : 0;
            Nn[pi] = Nn[pi] + Nn[b] + cross(bp[b] - bp[pi], F[b]);
            F[pi] = F[pi] + F[b];
        }
    }
    // ---- Cholesky + inverse
    // (divisions by the diagonal are multiplications by its reciprocal: 11 IEEE divisions instead of ~300)
    float L[NB][NB], iL[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) {
            float s = M[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            if (i == j) { L[i][i] = sqrtf(s); iL[i] = 1.0f / L[i][i]; }
            else L[i][j] = s * iL[j];
        }
    }
    float rhs[NB], qdd[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) { rhs[i] = -bias[i] - B.damping[i] * qd[i]; qdd[i] = 0; }
#pragma unroll
    for (int c = 0; c < NB; c++) {
        float y[NB], xcol[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) {
            float s = (i == c) ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
            y[i] = s * iL[i];
        }
#pragma unroll
        for (int i = NB - 1; i >= 0; i--) {
            float s = y[i];
#pragma unroll
            for (int k = i + 1; k < NB; k++) s -= L[k][i] * xcol[k];
            xcol[i] = s * iL[i];
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            SCR(S_MINV + i * NB + c) = xcol[i];
            qdd[i] += xcol[i] * rhs[c];
        }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) SCR(S_QDS + i) = qd[i] + P.dt * qdd[i];
    }
    if (PHASE == 2) return;
    // ---- objects: rotation, inverse inertia, unconstrained velocities -- of the pose that counts: an object the out-of-bounds
    // rule (env.py:257-264) sends home is taken at its home pose, at rest (the solve kernel writes that pose into the state when the step starts)
    for (int i = 0; i < P.nobj; i++) {
        float px = STT(ST_OPOS + 3 * i), py = STT(ST_OPOS + 3 * i + 1), pz = STT(ST_OPOS + 3 * i + 2);
        float qx = STT(ST_OQUAT + 4 * i), qy = STT(ST_OQUAT + 4 * i + 1), qz = STT(ST_OQUAT + 4 * i + 2), qw = STT(ST_OQUAT + 4 * i + 3);
        v3 v = mk(STT(ST_OVEL + 3 * i), STT(ST_OVEL + 3 * i + 1), STT(ST_OVEL + 3 * i + 2));
        v3 om = mk(STT(ST_OANG + 3 * i), STT(ST_OANG + 3 * i + 1), STT(ST_OANG + 3 * i + 2));
        if (object_out_of_bounds(px, pz, B.table_z)) {
            px = D.obj_home[(size_t)(7 * i) * N + env]; py = D.obj_home[(size_t)(7 * i + 1) * N + env]; pz = D.obj_home[(size_t)(7 * i + 2) * N + env];
            qx = D.obj_home[(size_t)(7 * i + 3) * N + env]; qy = D.obj_home[(size_t)(7 * i + 4) * N + env];
            qz = D.obj_home[(size_t)(7 * i + 5) * N + env]; qw = D.obj_home[(size_t)(7 * i + 6) * N + env];
            v = mk(0, 0, 0); om = mk(0, 0, 0);
        }
        m3 R = nc::quat_to_m3(qx, qy, qz, qw);
        float I6[6] = {B.obj_inertia[i][0], B.obj_inertia[i][1], B.obj_inertia[i][2], 0, 0, 0};
        float Ii6[6] = {1.0f / B.obj_inertia[i][0], 1.0f / B.obj_inertia[i][1], 1.0f / B.obj_inertia[i][2], 0, 0, 0};
        m3 Iw = inertia_world(R, I6), Iinv = inertia_world(R, Ii6);
        float vn = sqrtf(dot(v, v)), wn = sqrtf(dot(om, om));
        v3 vs = v + (v * (-(P.lin_damp + P.lin_damp * vn))) * P.dt;
        vs.z -= P.dt * P.gravity;
        v3 alo = mulv(Iinv, cross(om, mulv(Iw, om)));
        v3 ws = om + (alo * -1.0f - om * (P.ang_damp + P.ang_damp * wn)) * P.dt;
#pragma unroll
        for (int k = 0; k < 9; k++) { SCR(S_OR + 9 * i + k) = R.m[k]; SCR(S_OIINV + 9 * i + k) = Iinv.m[k]; }
        SCR(S_OVS + 3 * i) = vs.x; SCR(S_OVS + 3 * i + 1) = vs.y; SCR(S_OVS + 3 * i + 2) = vs.z;
        SCR(S_OWS + 3 * i) = ws.x; SCR(S_OWS + 3 * i + 1) = ws.y; SCR(S_OWS + 3 * i + 2) = ws.z;
        SCR(S_OP + 3 * i) = px; SCR(S_OP + 3 * i + 1) = py; SCR(S_OP + 3 * i + 2) = pz;
    }
}
// One thread per env of the class `sel` (pick_env); the launches cover N work items whatever the class (a heavy list's length
// is known on the device only; work items past its end exit at once).
template <int PHASE>
__device__ __forceinline__ void prep_class(const BodyParams &B, const SimParams &P, const DevPtrs &D, int sel) {
    const int env = pick_env(D, sel, blockIdx.x * blockDim.x + threadIdx.x, P.N);
    if (env < 0) return;
    prep_body<PHASE>(B, P, D, env);
}
__global__ void __launch_bounds__(64) k_prep_a(BodyParams B, SimParams P, DevPtrs D, int sel, int zero_counts) {
    if (zero_counts && blockIdx.x == 0 && threadIdx.x == 0) {       // in-line pass: the frame k_collide is about to fill holds stale counts
        D.hcount_next[0] = 0; D.hcount_next[1] = 0; D.hcount2_next[0] = 0; D.hcount2_next[1] = 0;
    }
    prep_class<1>(B, P, D, sel);
}
__global__ void __launch_bounds__(64) k_prep_b(BodyParams B, SimParams P, DevPtrs D, int sel) { prep_class<2>(B, P, D, sel); }
// both phases in one launch (the look-ahead behind a class' solve: one kernel instead of two on that stream)
__global__ void __launch_bounds__(64) k_prep_ab(BodyParams B, SimParams P, DevPtrs D, int sel) { prep_class<0>(B, P, D, sel); }

// ---------------------------------------------------------------------------------------------- k_collide
struct Xf { m3 R; v3 p; };

// World transform of the owner of shape s.  All loads are unconditional and from always-valid addresses (the values are
// selected afterwards): with loads under the owner-type branches the compiler merges the branches into a select of base
// pointers whose value is undefined on the static path and may still issue the load -- a fault at a garbage address.
__device__ __forceinline__ Xf load_xf(const ShapeData *S, int s, const float *scratch, int env) {
    Xf X;
    const int ot = S->otype[s];
    const int oi = ot == 0 ? 0 : S->oidx[s];
    const int rslot = (ot == 1 ? S_BR : S_OR) + 9 * oi;
    float r[9];
    r[0] = SCR(rslot); r[1] = SCR(rslot + 1); r[2] = SCR(rslot + 2); r[3] = SCR(rslot + 3); r[4] = SCR(rslot + 4);
    r[5] = SCR(rslot + 5); r[6] = SCR(rslot + 6); r[7] = SCR(rslot + 7); r[8] = SCR(rslot + 8);
    const int ob = ot == 2 ? oi : 0, bb = ot == 1 ? oi : 0;
    const float pbx = SCR(S_BP + 3 * bb), pby = SCR(S_BP + 3 * bb + 1), pbz = SCR(S_BP + 3 * bb + 2);
    const float pox = SCR(S_OP + 3 * ob), poy = SCR(S_OP + 3 * ob + 1), poz = SCR(S_OP + 3 * ob + 2);      // (the pose that counts: prep_body<1>)
    const bool st = ot == 0, body = ot == 1;
    X.R.m[0] = st ? 1.0f : r[0]; X.R.m[1] = st ? 0.0f : r[1]; X.R.m[2] = st ? 0.0f : r[2];
    X.R.m[3] = st ? 0.0f : r[3]; X.R.m[4] = st ? 1.0f : r[4]; X.R.m[5] = st ? 0.0f : r[5];
    X.R.m[6] = st ? 0.0f : r[6]; X.R.m[7] = st ? 0.0f : r[7]; X.R.m[8] = st ? 1.0f : r[8];
    X.p = mk(st ? 0.0f : (body ? pbx : pox), st ? 0.0f : (body ? pby : poy), st ? 0.0f : (body ? pbz : poz));
    return X;
}

#ifdef RR_RASTER_STATS
// development build only: cycle stamps of kernel phases (lane 0 of every block), see scratch/sprof.py
__device__ unsigned long long g_sprof[16];
#define SPROF(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); sp_acc[i] += (unsigned)(now_ - sp_t0); sp_t0 = now_; } while (0)   /* cheap phase clock: s_memtime into per-wave accumulators, flushed once (SBLK_END); RR_ABLATE = block << 16 | 0x4000: one block only */
#define SPROF_INIT unsigned long long sp_t0 = __builtin_readcyclecounter(); unsigned sp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SPROF_FLUSH do { if ((threadIdx.x & 63) == 0 && (threadIdx.x >> 6) == ((OW && LIGHT_OW_THREADS == 384) ? 1 : 0) && (!(P.ablate & 0x4000) || (int)blockIdx.x == (P.ablate >> 16))) { _Pragma("unroll") for (int i_ = 0; i_ < 12; i_++) atomicAdd(&g_sprof[i_], (unsigned long long)sp_acc[i_]); } } while (0)
// per solver workgroup: {total cycles, cycles up to the end of the row build, cycles of the PGS loop, -, then per env
// nc | generic contacts << 8 | leading object-vs-static contacts << 16 | last F-list length << 24}
__device__ unsigned g_sblk[4096 * 8];
#define SBLK_BEGIN const unsigned long long sb_t0 = __builtin_readcyclecounter(); unsigned long long sb_t1 = sb_t0, sb_t2 = sb_t0;
#define SBLK_MARK(v) v = __builtin_readcyclecounter();
#define SBLK_END(ncv, gv, osv, nfv) do { if (unit < 4096) { if (l == 0) g_sblk[unit * 8 + 4 + (grp & 3)] = (unsigned)(ncv) | ((unsigned)(gv) << 8) | ((unsigned)(osv) << 16) | ((unsigned)(nfv) << 24); \
    if ((threadIdx.x & 63) == 0) { g_sblk[unit * 8] = (unsigned)(__builtin_readcyclecounter() - sb_t0); g_sblk[unit * 8 + 1] = (unsigned)(sb_t1 - sb_t0); g_sblk[unit * 8 + 2] = (unsigned)(sb_t2 - sb_t1); } } } while (0)
extern "C" int rr_debug_solver_blocks(unsigned *out, int nblocks) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sblk), sizeof(unsigned) * 8 * (size_t)nblocks) == hipSuccess ? 0 : -1;
}
extern "C" int rr_debug_solver_prof(unsigned long long *out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sprof), sizeof(g_sprof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_sprof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#else
#define SPROF(i)
#define SPROF_INIT
#define SPROF_FLUSH
#define SBLK_BEGIN
#define SBLK_MARK(v)
#define SBLK_END(ncv, gv, osv, nfv)
#endif

// ---- collision: one wavefront per env ------------------------------------------------------------------------------
// Every collision shape is the convex hull of its OBJ reduced to <= 192 vertices and <= 192 facet planes
// (deviation from the full hull < 1 mm, tests/test_oracle_pins.py).  A contact candidate is a vertex of one shape whose
// largest signed distance to the planes of the other is below the margin; <= 4 points per pair are kept.  A wavefront
// owns an env:
//   1. lanes 0..ns-1 load the transform of "their" shape and its world bounding-sphere centre into LDS (one round trip);
//   2. the pairs are sphere-tested 64 at a time, survivors are collected with ballots;
//   3. every surviving pair is handled by the whole wave, direction 0 (vertices of a against the planes of b) and then
//      direction 1 (b against a):
//        a. exact cull -- the bounding sphere of "mine" beyond one plane of "other" by more than the margin: nothing;
//        b. the planes of "other" are staged in LDS; lane = vertex, 64 at a time: a vertex that is beyond one of the first
//           six planes (the shape's extreme facets along +-x, +-y, +-z) by the margin cannot be a candidate and is dropped
//           (exact: the candidate test is a maximum over ALL planes of the very same values); survivors are compacted,
//           in vertex order, into an LDS list;
//        c. lane = survivor: maximum over all planes (first maximum wins), candidates appended to the pair's list.
//   4. manifold reduction over the candidate list (direction 0 by vertex index, then direction 1 -- the oracle's order):
//      the "first best wins" selections of the oracle's reduce4() are wave reductions with lowest-index tie-breaks.
// The whole kernel is contraction-free (see "contraction-free twins"): results are bit-identical to the oracle's float
// build.  No atomics, no work list: results do not depend on scheduling.
#define COLLIDE_WAVES COLLIDE_WAVES_   // waves per env: the close pairs are dealt out among them (the slowest env sets the kernel's duration)
#define COLLIDE_THREADS (64 * COLLIDE_WAVES)
#define CCHUNK 64         // close pairs whose results are held in LDS at a time (COLLIDE_THREADS / 4 record writers)
#define NPREF 6          // leading planes of every shape used by the prefilter (tools/compile_model.py orders them)
#define CAND_MAX 128     // candidates kept per pair (the oracle applies the same cap)
#define CSHAPES 24       // collision shapes staged in LDS (rr_create checks the model: 22)
#define COLLIDE_WAVES_ 4
#define VH_MAX 1024         // cap of the very heavy list (k_collide)
#define COOP_MAX 256        // (1024 measured on the macro workload, 388 very heavy envs: no gain) lists up to this long (lagged host count) are solved one env per wave (coop row build); longer ones four to a wave
#ifdef RR_RASTER_STATS
#define CABL(bit) (P.ablate & (bit))      // development build: phase ablations (256 stage only, 512 no pairs, 1024 cull only)
// per-env phase cycles of k_collide (scratch/cprof.py): 0 stage, 1 sphere tests, 2 loads + cull, 3 prefilter, 4 all-plane pass,
// 5 manifold reduction, 6 record write, 7 close pairs
__device__ unsigned g_cprof[8192 * COLLIDE_WAVES_][8];
extern "C" int rr_debug_collide_prof(unsigned *out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cprof), sizeof(unsigned) * 8 * COLLIDE_WAVES_ * (size_t)n) == hipSuccess ? 0 : -1;   // [n][waves][8]
}
#define CPROF_INIT unsigned cp_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long cp_t0 = __builtin_readcyclecounter();
#define CPROF(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); cp_[i] += (unsigned)(now_ - cp_t0); cp_t0 = now_; } while (0)
#define CPROF_COUNT(i) cp_[i]++
#define CPROF_END do { if (lane == 0 && env < 8192) for (int i_ = 0; i_ < 8; i_++) g_cprof[env * COLLIDE_WAVES_ + wv][i_] = cp_[i_]; } while (0)
#else
#define CABL(bit) false
#define CPROF_INIT
#define CPROF(i)
#define CPROF_COUNT(i)
#define CPROF_END
#endif

// wave-wide "first lane holding the maximum of v among lanes with ok" (returns -1 when no lane is ok or none exceeds
// floor).  The maximum is reduced with DPP row rotations inside each 16-lane row and four v_readlane across the rows
// (an LDS-crossbar butterfly of six dependent ds_bpermute costs ten times as much on a chain this short).
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_max(float m) {
    m = fmaxf(m, dpp_mov<0x128>(m));     // row_ror:8
    m = fmaxf(m, dpp_mov<0x124>(m));     // row_ror:4
    m = fmaxf(m, dpp_mov<0x122>(m));     // row_ror:2
    m = fmaxf(m, dpp_mov<0x121>(m));     // row_ror:1  -> every lane holds its row's maximum
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// value of lane src (wave-uniform) in every lane
__device__ __forceinline__ float lane_f(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// exclusive prefix sum over the 64 lanes of a wave (DPP row_shr scan inside each 16-lane row + the row totals)
template <int SHR> __device__ __forceinline__ int dpp_shr0(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + SHR, 0xf, 0xf, true); }
__device__ __forceinline__ int wave_excl_scan(int v, int lane, int &total) {
    int x = v;
    x += dpp_shr0<1>(x); x += dpp_shr0<2>(x); x += dpp_shr0<4>(x); x += dpp_shr0<8>(x);
    const int t0 = __builtin_amdgcn_readlane(x, 15), t1 = __builtin_amdgcn_readlane(x, 31), t2 = __builtin_amdgcn_readlane(x, 47),
              t3 = __builtin_amdgcn_readlane(x, 63);
    const int row = lane >> 4;
    const int off = row == 0 ? 0 : (row == 1 ? t0 : (row == 2 ? t0 + t1 : t0 + t1 + t2));
    total = t0 + t1 + t2 + t3;
    return x + off - v;
}

// the candidate / plane / survivor arrays are private to a wave: LDS traffic is ordered per wave, so a compiler-level fence replaces s_barrier
#define CSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
static_assert(CCHUNK * 4 == COLLIDE_THREADS && CCHUNK <= 64 && MAXPAIRS < 128 && EMAXC <= 64 && MAXC <= 64, "one record writer per (slot, r); pair ids in 7 bits, edge ids in 6, the contact history staged by one wave");
static_assert(VMAXC % 64 == 0 && FMAXC % 64 == 0 && VMAXC < 256 && FMAXC < 256, "vertex / plane passes of 64, counts packed in bytes");
// (the whole kernel is contraction-free and uses the nc:: helpers: see "contraction-free twins" above)
#pragma clang fp contract(off)
// An edge of a shape in the world frame, and the edge-edge candidate test -- the oracle's edge_edge(), operation for operation.
struct EdgeW { v3 p0, d, n1, n2; };
__device__ __forceinline__ EdgeW edge_world(const float *e, const float *x /* R (9), p (3) */) {
    m3 R;
#pragma unroll
    for (int k = 0; k < 9; k++) R.m[k] = x[k];
    EdgeW E;
    E.p0 = nc::add(nc::mulv(R, mk(e[0], e[1], e[2])), mk(x[9], x[10], x[11]));
    E.d = nc::mulv(R, mk(e[3], e[4], e[5]));
    E.n1 = nc::mulv(R, mk(e[6], e[7], e[8]));
    E.n2 = nc::mulv(R, mk(e[9], e[10], e[11]));
    return E;
}
// true: the closest points of the two edge lines lie strictly inside both segments, the common normal is a face of the
// Minkowski difference and the signed distance along it is inside (-EDGE_DEPTH, margin); cnd = (contact point, distance),
// nrm = the normal B -> A
__device__ __forceinline__ bool edge_pair(const EdgeW &A, const EdgeW &B, float margin, float lo, float4 &cnd, v3 &nrm) {
    const v3 r = nc::sub(A.p0, B.p0);
    const float a = nc::dot(A.d, A.d), e = nc::dot(B.d, B.d), b = nc::dot(A.d, B.d), c = nc::dot(A.d, r), f = nc::dot(B.d, r);
    const float ae = a * e, den = ae - b * b;
    nrm = mk(0, 0, 0); cnd = make_float4(0, 0, 0, 0);
    if (!(den > 1e-4f * ae)) return false;
    const float s = (b * f - c * e) / den, t = (a * f - b * c) / den;
    if (!(s > 0 && s < 1 && t > 0 && t < 1)) return false;
    const v3 p = nc::add(A.p0, nc::scale(A.d, s)), q = nc::add(B.p0, nc::scale(B.d, t));
    const v3 nv = nc::cross(A.d, B.d);
    const float il = 1.0f / sqrtf(nc::dot(nv, nv));
    const float sg = nc::dot(nv, nc::add(A.n1, A.n2)) > 0 ? il : -il;
    const v3 u = nc::scale(nv, sg);
    if (!(nc::dot(nc::cross(A.n1, u), nc::cross(u, A.n2)) >= 0)) return false;
    const v3 w = nc::scale(u, -1.0f);
    if (!(nc::dot(w, nc::add(B.n1, B.n2)) > 0)) return false;
    if (!(nc::dot(nc::cross(B.n1, w), nc::cross(w, B.n2)) >= 0)) return false;
    const v3 pq = nc::sub(p, q);
    const float dist = nc::dot(w, pq);
    if (!(dist < margin && dist > -0.005f && dist > lo)) return false;      // (oracle EDGE_DEPTH; lo: not deeper than the face axes)
    const v3 x = nc::add(q, nc::scale(pq, 0.5f));
    cnd = make_float4(x.x, x.y, x.z, dist);
    nrm = w;
    return true;
}
// index of the first maximum of val(i) over the candidates i < n with ok(i), or -1 when there is none above `floor_`;
// candidates are visited 64 at a time, a later chunk only replaces the winner when it is strictly greater
#define CAND_ARGMAX(RESULT, FLOOR, OKEXPR, VALEXPR)                                                   \
    {                                                                                                 \
        float bestv_ = (FLOOR); int besti_ = -1;                                                      \
        for (int c0_ = 0; c0_ < ncand; c0_ += 64) {                                                   \
            const int ci = c0_ + lane;                                                                \
            const bool in_ = ci < ncand;                                                              \
            const float4 ca = cand_a[in_ ? ci : 0];                                                   \
            const bool ok_ = in_ && (OKEXPR);                                                         \
            const float val_ = (VALEXPR);                                                             \
            const float m_ = wave_max(ok_ ? val_ : -3.0e38f);                                         \
            if (m_ > bestv_) {                                                                        \
                const unsigned long long bb_ = __ballot(ok_ && val_ == m_);                           \
                if (bb_) { bestv_ = m_; besti_ = c0_ + __ffsll((long long)bb_) - 1; }                 \
            }                                                                                         \
        }                                                                                             \
        RESULT = besti_;                                                                              \
    }
// The collision pass of one env: reads the contact history (current frame: D.clist, D.ccount, D.cforce) and the collision
// inputs prep_body<1> left in the scratch slab, fills the NEXT frame (D.clist_next, D.ccount_next, D.cwarm_next, class + lists).
__device__ __forceinline__ void collide_env(const SimParams &P, const DevPtrs &D, int ns, int env) {
    const int N = P.N;
    const float *state = D.state;
    float *scratch = D.scratch;
    if (D.errflags[env] & 1u) {         // frozen env: no contacts, light
        if (threadIdx.x == 0) { D.ccount_next[env] = 0; D.hgflag_next[env] = 0; }
        return;
    }
    const ShapeData *S = D.shapes;
    // 38.9 KB of LDS: four workgroups = sixteen waves per CU
    __shared__ float xf[CSHAPES][12];          // R (row-major 9), p (3) of every shape's owner
    __shared__ float4 sph[CSHAPES];            // world bounding sphere
    __shared__ int pair_ab[MAXPAIRS];          // shape a | shape b << 8 of every pair   } staged once: no global load of
    __shared__ int shape_n[CSHAPES];           // vertex count | plane count << 8 | edge count << 16   } metadata inside the pair loop
    __shared__ int pair_key[MAXPAIRS];         // bodyA | bodyB << 8 | link of shape a << 16 (the meta word of the pair's contacts)
    __shared__ unsigned char edge_id[COLLIDE_WAVES][2][EMAXC];   // per wave: the original indices of the staged edges
    __shared__ unsigned char close_pair[MAXPAIRS];   // the pairs that pass the sphere test, in pair order
    __shared__ int n_close, next_item, nct_sh;
    __shared__ float4 res_a[CCHUNK][4];        // per close pair of the chunk: the kept candidates (contact point, signed distance)
    __shared__ int res_b[CCHUNK][4];           //                              plane of "other" | direction << 8
    __shared__ int res_k[CCHUNK];              //                              their number; after the scan: offset in the env's list | kept << 8
    __shared__ int res_key[CCHUNK];            //                              bodyA | bodyB << 8 | linkA << 16 of the pair
    // per wave:
    __shared__ float4 planes_w[COLLIDE_WAVES][FMAXC];      // planes of "other" in the current direction
    __shared__ float surv_w[COLLIDE_WAVES][VMAXC][3];      // world position of the vertices that survive the prefilter, in vertex order
    __shared__ float4 cand_a_w[COLLIDE_WAVES][CAND_MAX];   // candidates of the pair: contact point, signed distance
    __shared__ unsigned short cand_b_w[COLLIDE_WAVES][CAND_MAX];   //                 plane of "other" it is nearest to | direction << 8; edge candidates: edge of a | edge of b << 6 | 1 << 15
    __shared__ float4 pv[MAXC];                // contact history (warm start): {x, y, z, normal force} of the previous step's contacts,
    __shared__ int pv_key[MAXC];               //   their bodyA | bodyB << 8 | linkA << 16 | pair << 24,
    __shared__ unsigned char run_s[MAXPAIRS], run_e[MAXPAIRS];   //   and per pair its run [start, end) of that list
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float4 *planes = planes_w[wv];
    float (*surv)[3] = surv_w[wv];
    float4 *cand_a = cand_a_w[wv];
    unsigned short *cand_b = cand_b_w[wv];
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    CPROF_INIT
    const int nprev = P.warmstart > 0.0f ? min(D.ccount[env], MAXC) : 0;      // contacts of the step before (the current frame; this pass fills the next one)
    for (int pr = tid; pr < P.npairs; pr += COLLIDE_THREADS) {
        const int ba = S->pair_meta[pr][0], bb = S->pair_meta[pr][1];
        const int cls = (((ba >= 0 && ba < 16) || (bb >= 0 && bb < 16)) ? 1 : 0) | ((ba >= 16 && bb >= 16) ? 2 : 0);
        pair_key[pr] = (ba & 255) | ((bb & 255) << 8) | ((S->pair_meta[pr][2] & 255) << 16);
        pair_ab[pr] = S->pair_a[pr] | (S->pair_b[pr] << 8) | (cls << 16) | (bb < 0 ? 1 << 18 : 0);      // + robot involved (bit 16), object-object (bit 17), b static (bit 18)
    }
    if (wv == 1) {       // the contact history comes in with the same round trip as the shape transforms (it has left the L2 since k_solve read it)
        for (int q = lane; q < MAXPAIRS; q += 64) { run_s[q] = 0; run_e[q] = 0; }
        if (lane < nprev) {
            const float4 *pr_ = D.clist + ((size_t)env * MAXC + lane) * 3;
            const float4 p0 = pr_[0], p1 = pr_[1];
            pv[lane] = make_float4(p0.x, p0.y, p0.z, D.cforce[(size_t)env * MAXC + lane]);
            pv_key[lane] = __float_as_int(p1.w);
        }
        CSYNC();
        if (lane < nprev) {       // the list is in pair order: the contacts of a pair form one run
            const int pj = pv_key[lane] >> 24;
            if (lane == 0 || (pv_key[lane - 1] >> 24) != pj) run_s[pj] = (unsigned char)lane;
            if (lane + 1 >= nprev || (pv_key[lane + 1] >> 24) != pj) run_e[pj] = (unsigned char)(lane + 1);
        }
    }
    if (tid < ns) {
        shape_n[tid] = S->nv[tid] | (S->nf[tid] << 8) | (S->ne[tid] << 16);
        const Xf X = load_xf(S, tid, scratch, env);
#pragma unroll
        for (int k = 0; k < 9; k++) xf[tid][k] = X.R.m[k];
        xf[tid][9] = X.p.x; xf[tid][10] = X.p.y; xf[tid][11] = X.p.z;
        const v3 c = nc::add(nc::mulv(X.R, mk(S->sphere[tid][0], S->sphere[tid][1], S->sphere[tid][2])), X.p);
        sph[tid] = make_float4(c.x, c.y, c.z, S->sphere[tid][3]);
    }
    __syncthreads();
    CPROF(0);
    if (CABL(256)) return;
    // ---- the pairs whose bounding spheres come within the margin, in pair order (wave 0)
    if (wv == 0) {
        int nc_ = 0;
        for (int p0 = 0; p0 < P.npairs; p0 += 64) {
            const int pr = p0 + lane;
            bool close = false;
            if (pr < P.npairs) {
                const float4 a = sph[pair_ab[pr] & 255], b = sph[(pair_ab[pr] >> 8) & 255];
                const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z, rr = a.w + b.w + P.margin;
                close = !(dx * dx + dy * dy + dz * dz > rr * rr);
            }
            const unsigned long long cm_ = CABL(512) ? 0ull : __ballot(close);
            if (close && !CABL(512)) close_pair[nc_ + __popcll(cm_ & lt_mask)] = (unsigned char)pr;
            nc_ += __popcll(cm_);
        }
        if (lane == 0) { n_close = nc_; next_item = 0; }
    }
    __syncthreads();
    CPROF(1);
    const int ncl = n_close;
    int nct = 0;                                // contacts of this env so far        }
    int ngen = 0;                               // contacts that are not object-vs-static ones the object lanes take: generic rows    } tracked by wave 0
    int oscnt0 = 0, oscnt1 = 0, oscnt2 = 0;     // object-vs-static contacts per object }
    int nchunk = 0;
    for (int base = 0; base < ncl && nct < MAXC; base += nchunk) {
        nchunk = min(CCHUNK, ncl - base);
        // (more than CCHUNK close pairs: rare) the pairs of one moving shape with the statics share their bodies -- the warm
        // start below looks at all contacts of the same bodies, so a chunk never ends inside such a run (<= 3 pairs)
        while (base + nchunk < ncl) {
            const int pa_ = pair_ab[close_pair[base + nchunk - 1]], pb_ = pair_ab[close_pair[base + nchunk]];
            if ((pa_ & 255) == (pb_ & 255) && (pa_ & pb_ & (1 << 18))) nchunk--; else break;
        }
        // ---- every wave takes the next unprocessed close pair of the chunk (the result goes to the pair's slot: the
        // outcome does not depend on which wave took it)
        for (;;) {
            int item = 0;
            if (lane == 0) item = atomicAdd(&next_item, 1);
            item = __builtin_amdgcn_readfirstlane(item);
            if (item >= base + nchunk) break;
            CPROF_COUNT(7);
            const int slot = item - base;
            const int pair = close_pair[item];
            const int sa = pair_ab[pair] & 255, sb = (pair_ab[pair] >> 8) & 255;
            int ncand = 0;                      // wave-uniform
            bool apart = false;                 // the exact cull of a direction found the shapes more than the margin apart
            for (int dirflag = 0; dirflag < 2; dirflag++) {
                // "mine" = the shape whose vertices are tested, "other" = the shape whose planes they are tested against
                const int sm = dirflag ? sb : sa, so = dirflag ? sa : sb;
                Xf Xm, Xo;
#pragma unroll
                for (int kk = 0; kk < 9; kk++) { Xm.R.m[kk] = xf[sm][kk]; Xo.R.m[kk] = xf[so][kk]; }
                Xm.p = mk(xf[sm][9], xf[sm][10], xf[sm][11]);
                Xo.p = mk(xf[so][9], xf[so][10], xf[so][11]);
                const int nv = shape_n[sm] & 255, nf = (shape_n[so] >> 8) & 255;
                // every global load of this direction is issued here, before the first wait: the planes of "other" (lane =
                // plane, three per lane) and the vertices of "mine" (lane = vertex, three passes) -- one round trip
                float4 plr[FMAXC / 64];
                float vxr[VMAXC / 64], vyr[VMAXC / 64], vzr[VMAXC / 64];
#pragma unroll
                for (int i = 0; i < FMAXC / 64; i++) plr[i] = *(const float4 *)S->planes[so][min(lane + 64 * i, FMAXC - 1)];
#pragma unroll
                for (int i = 0; i < VMAXC / 64; i++) {
                    const float *vp = S->verts[sm][min(lane + 64 * i, VMAXC - 1)];
                    vxr[i] = vp[0]; vyr[i] = vp[1]; vzr[i] = vp[2];
                }
                CSYNC();            // the previous direction's reads of planes / surv are done
                bool sep = false;
                {   // a. exact cull + b. plane staging (lane = plane)
                    const float4 cm = sph[sm];
                    const v3 cl = nc::tmulv(Xo.R, nc::sub(mk(cm.x, cm.y, cm.z), Xo.p));
#pragma unroll
                    for (int i = 0; i < FMAXC / 64; i++) {
                        const int f = lane + 64 * i;
                        if (f < nf) {
                            const float4 pl = plr[i];
                            planes[f] = pl;
                            sep = sep || (pl.x * cl.x + pl.y * cl.y + pl.z * cl.z - pl.w > cm.w + P.margin);
                        }
                    }
                }
                CPROF(2);
                if (__ballot(sep)) { apart = true; continue; }
                if (CABL(1024)) continue;
                CSYNC();
                const int npre = min(NPREF, nf);
                int nsurv = 0;                  // wave-uniform
#pragma unroll
                for (int i = 0; i < VMAXC / 64; i++) {
                    if (64 * i >= nv) break;
                    const int v = 64 * i + lane;
                    bool keep = false;
                    v3 xw = mk(0, 0, 0);
                    if (v < nv) {
                        xw = nc::mulv_add_fma(Xm.R, mk(vxr[i], vyr[i], vzr[i]), Xm.p);
                        const v3 xl = nc::tmulv_fma(Xo.R, nc::sub(xw, Xo.p));
                        float best = -1e30f;
                        for (int f = 0; f < npre; f++) best = fmaxf(best, nc::plane_dist_fma(planes[f], xl));
                        keep = best < P.margin;
                    }
                    const unsigned long long km = __ballot(keep);
                    if (keep) {
                        const int pos = nsurv + __popcll(km & lt_mask);
                        surv[pos][0] = xw.x; surv[pos][1] = xw.y; surv[pos][2] = xw.z;
                    }
                    nsurv += __popcll(km);
                }
                CPROF(3);
                if (nsurv == 0) continue;
                CSYNC();
                for (int k0 = 0; k0 < nsurv; k0 += 64) {
                    const int k = k0 + lane;
                    bool hit = false;
                    float cx = 0, cy = 0, cz = 0, cs = 0;
                    int bfk = 0;
                    if (k < nsurv) {
                        const v3 xw = mk(surv[k][0], surv[k][1], surv[k][2]);
                        const v3 xl = nc::tmulv_fma(Xo.R, nc::sub(xw, Xo.p));
                        float best = -1e30f;
                        int bf = 0;
                        for (int f = 0; f < nf; f++) {
                            const float sd = nc::plane_dist_fma(planes[f], xl);
                            if (sd > best) { best = sd; bf = f; }
                        }
                        if (best < P.margin) {
                            const float4 pl = planes[bf];
                            const v3 nw = nc::mulv(Xo.R, mk(pl.x, pl.y, pl.z));
                            cx = xw.x - 0.5f * best * nw.x; cy = xw.y - 0.5f * best * nw.y; cz = xw.z - 0.5f * best * nw.z;
                            cs = best;
                            bfk = bf | (dirflag << 8);
                            hit = true;
                        }
                    }
                    const unsigned long long hm = __ballot(hit);
                    const int pos = ncand + __popcll(hm & lt_mask);
                    if (hit && pos < CAND_MAX) {
                        cand_a[pos] = make_float4(cx, cy, cz, cs);
                        cand_b[pos] = (unsigned short)bfk;
                    }
                    ncand = min(ncand + __popcll(hm), CAND_MAX);
                }
                CPROF(4);
            }
            CSYNC();
            // ---- edge-edge candidates (oracle edge_edge()): two edges crossing away from any vertex.  Lane = edge: the long sharp edges of both shapes go to the world frame, those that come
            // within the margin of the other shape's bounding sphere are staged in LDS (planes: A's, surv: B's; in index
            // order); then lane = edge pair, in the oracle's (i, j) order.
            const int ne_a = shape_n[sa] >> 16, ne_b = shape_n[sb] >> 16;
            if (!apart && ne_a > 0 && ne_b > 0 && P.edge_contacts && !CABL(1024)) {
                int nst[2];
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    const int sm = side ? sb : sa, so = side ? sa : sb, ne = side ? ne_b : ne_a;
                    bool keep = false;
                    EdgeW E;
                    if (lane < ne) {
                        E = edge_world(S->edges[sm][lane], xf[sm]);
                        // conservative: a contact needs a point of this edge within radius + margin of the other sphere's centre
                        const float4 co = sph[so];
                        const v3 r0 = nc::sub(mk(co.x, co.y, co.z), E.p0);
                        const float dd = nc::dot(E.d, E.d);
                        float tt = nc::dot(r0, E.d) / dd;
                        tt = fminf(fmaxf(tt, 0.0f), 1.0f);
                        const v3 rr = nc::sub(r0, nc::scale(E.d, tt));
                        const float lim = co.w + P.margin;
                        keep = nc::dot(rr, rr) <= lim * lim * 1.01f + 1e-6f;
                    }
                    const unsigned long long km = __ballot(keep);
                    if (keep) {
                        const int pos = __popcll(km & lt_mask);
                        float4 *dst = side ? (float4 *)&surv[0][0] + 3 * pos : planes + 3 * pos;
                        dst[0] = make_float4(E.p0.x, E.p0.y, E.p0.z, E.d.x);
                        dst[1] = make_float4(E.d.y, E.d.z, E.n1.x, E.n1.y);
                        dst[2] = make_float4(E.n1.z, E.n2.x, E.n2.y, E.n2.z);
                        edge_id[wv][side][pos] = (unsigned char)lane;
                    }
                    nst[side] = __popcll(km);
                }
                CSYNC();
                // overlapping shapes: an edge axis deeper than the deepest vertex candidate (the face axes) is no contact
                float smin = 0.0f;
                for (int c0_ = 0; c0_ < ncand; c0_ += 64) smin = fminf(smin, -wave_max(c0_ + lane < ncand ? -cand_a[c0_ + lane].w : -3.0e38f));
                const float lo = smin - 0.0005f;                 // (oracle EDGE_SLOP)
                const int na = nst[0], nb = nst[1], npairs_e = na * nb;
                const float inb = 1.0f / (float)nb;
                for (int k0 = 0; k0 < npairs_e; k0 += 64) {
                    const int kk = k0 + lane;
                    bool hit = false;
                    float4 cnd = make_float4(0, 0, 0, 0);
                    int ids = 0;
                    if (kk < npairs_e) {
                        const int i = (int)(((float)kk + 0.5f) * inb), j = kk - i * nb;      // exact for these small integers
                        const float4 *pa = planes + 3 * i, *pb = (const float4 *)&surv[0][0] + 3 * j;
                        const float4 a0 = pa[0], a1_ = pa[1], a2_ = pa[2], b0 = pb[0], b1_ = pb[1], b2_ = pb[2];
                        EdgeW A, B;
                        A.p0 = mk(a0.x, a0.y, a0.z); A.d = mk(a0.w, a1_.x, a1_.y); A.n1 = mk(a1_.z, a1_.w, a2_.x); A.n2 = mk(a2_.y, a2_.z, a2_.w);
                        B.p0 = mk(b0.x, b0.y, b0.z); B.d = mk(b0.w, b1_.x, b1_.y); B.n1 = mk(b1_.z, b1_.w, b2_.x); B.n2 = mk(b2_.y, b2_.z, b2_.w);
                        v3 nrm;
                        hit = edge_pair(A, B, P.margin, lo, cnd, nrm);
                        ids = edge_id[wv][0][i] | (edge_id[wv][1][j] << 6) | (1 << 15);
                    }
                    const unsigned long long hm = __ballot(hit);
                    const int pos = ncand + __popcll(hm & lt_mask);
                    if (hit && pos < CAND_MAX) { cand_a[pos] = cnd; cand_b[pos] = (unsigned short)ids; }
                    ncand = min(ncand + __popcll(hm), CAND_MAX);
                }
                CSYNC();
            }
            // manifold reduction, same rule as the oracle's reduce4(): deepest first, then maximal spread, preferring the
            // candidates within TIER_TOL (1 mm) of the deepest penetration at every pick; ties go to the first candidate
            int sel0 = -1, sel1 = -1, sel2 = -1, sel3 = -1, k = 0;
            if (CABL(2048)) ncand = 0;
            if (ncand <= 4) {
                sel0 = ncand > 0 ? 0 : -1; sel1 = ncand > 1 ? 1 : -1; sel2 = ncand > 2 ? 2 : -1; sel3 = ncand > 3 ? 3 : -1;
                k = ncand;
            } else {
                CAND_ARGMAX(sel0, -3.0e38f, true, -ca.w)                       // smallest s, first one
                const float smin = cand_a[sel0].w, lim = smin + 0.001f, tie = smin + 0.0005f;
                // anchor = a candidate within TIE_TOL (0.5 mm) of the deepest (oracle reduce4(): a resting face must give
                // the same quadruple whichever of its vertices happens to be deepest by a micrometre)
                // -- and among those the extreme one along a fixed skew direction, i.e. a corner of the face
                CAND_ARGMAX(sel0, -3.0e38f, ca.w < tie, ca.x + 0.618f * ca.y + 0.382f * ca.z)
                const float4 c0 = cand_a[sel0];
                const v3 x0 = mk(c0.x, c0.y, c0.z);
#define V1_ nc::dot(nc::sub(mk(ca.x, ca.y, ca.z), x0), nc::sub(mk(ca.x, ca.y, ca.z), x0))
                CAND_ARGMAX(sel1, -1.0f, ci != sel0 && ca.w < lim, V1_)
                if (sel1 < 0) CAND_ARGMAX(sel1, -1.0f, ci != sel0, V1_)
                const float4 c1 = cand_a[sel1];
                const v3 e = nc::sub(mk(c1.x, c1.y, c1.z), x0);
#define CR_ nc::cross(nc::sub(mk(ca.x, ca.y, ca.z), x0), e)
#define V2_ nc::dot(CR_, CR_)
                CAND_ARGMAX(sel2, -1.0f, ci != sel0 && ci != sel1 && ca.w < lim, V2_)
                if (sel2 < 0) CAND_ARGMAX(sel2, -1.0f, ci != sel0 && ci != sel1, V2_)
                const float4 c2 = cand_a[sel2];
                const v3 cr2 = nc::cross(nc::sub(mk(c2.x, c2.y, c2.z), x0), e);
#define V3_ (-nc::dot(CR_, cr2))
                CAND_ARGMAX(sel3, 0.0f, ci != sel0 && ci != sel1 && ci != sel2 && ca.w < lim, V3_)
                if (sel3 < 0) CAND_ARGMAX(sel3, 0.0f, ci != sel0 && ci != sel1 && ci != sel2, V3_)
#undef V1_
#undef CR_
#undef V2_
#undef V3_
                k = sel3 >= 0 ? 4 : 3;
            }
            if (lane < k) {
                const int ci = lane == 0 ? sel0 : (lane == 1 ? sel1 : (lane == 2 ? sel2 : sel3));
                res_a[slot][lane] = cand_a[ci];
                res_b[slot][lane] = cand_b[ci];
            }
            if (lane == 0) res_k[slot] = k;
            CSYNC();            // the candidate list is reused by the next pair
            CPROF(5);
        }
        __syncthreads();
        // ---- the chunk's contacts go to the env's list in pair order, up to MAXC (the oracle stops there too): wave 0
        // numbers them and classifies the env's solver group, then thread (slot, r) writes the slot's r-th record
        if (wv == 0) {
            const bool in_ = lane < nchunk;
            const int kj = in_ ? res_k[lane] : 0;
            int total;
            const int off = min(nct + wave_excl_scan(kj, lane, total), MAXC);
            const int kk = min(kj, MAXC - off);
            if (in_) {
                res_k[lane] = off | (kk << 8);
                res_key[lane] = pair_key[close_pair[base + lane]];
            }
            // the solver's object lanes take up to four object-vs-static contacts per object; anything else is a generic row
            const int pab = pair_ab[close_pair[base + (in_ ? lane : 0)]];
            const int cls = (pab >> 16) & 3;
            const int ob = (pab & 255) - (ns - NOBJ);                   // object index of shape a for the object-vs-static pairs
            const bool os = cls == 0 && ob >= 0;
            { int tg_; wave_excl_scan(os ? 0 : kk, lane, tg_); ngen += tg_; }
            int t0_, t1_, t2_;
            wave_excl_scan(os && ob == 0 ? kk : 0, lane, t0_);
            wave_excl_scan(os && ob == 1 ? kk : 0, lane, t1_);
            wave_excl_scan(os && ob == 2 ? kk : 0, lane, t2_);
            oscnt0 += t0_; oscnt1 += t1_; oscnt2 += t2_;
            nct = min(nct + total, MAXC);
        }
        __syncthreads();
        {
            const int slot = tid >> 2, r = tid & 3;
            const int ok_ = slot < nchunk ? res_k[slot] : 0;
            if (r < (ok_ >> 8)) {
                const int pair = close_pair[base + slot];
                const int sa = pair_ab[pair] & 255, sb = (pair_ab[pair] >> 8) & 255;
                const float4 a = res_a[slot][r];
                // the normal (B -> A) of a kept candidate: its plane of "other", rotated to the world, as in the candidate test
                const int kb = res_b[slot][r];
                v3 nb;
                if (kb >> 15) {       // edge-edge candidate: the common normal of edge (kb & 63) of a and edge (kb >> 6 & 63) of b, as in the test
                    float4 dummy;
                    edge_pair(edge_world(S->edges[sa][kb & 63], xf[sa]), edge_world(S->edges[sb][(kb >> 6) & 63], xf[sb]), P.margin, -1.0f, dummy, nb);
                } else {
                    const int so_ = (kb >> 8) ? sa : sb;
                    m3 Ro;
#pragma unroll
                    for (int kk = 0; kk < 9; kk++) Ro.m[kk] = xf[so_][kk];
                    const float4 pl = *(const float4 *)S->planes[so_][kb & 255];
                    nb = nc::mulv(Ro, mk(pl.x, pl.y, pl.z));
                    if (kb >> 8) nb = nc::scale(nb, -1.0f);
                }
                const float4 b = make_float4(nb.x, nb.y, nb.z, 0.0f);
                const int meta = pair_key[pair];
                float4 *rec = D.clist_next + ((size_t)env * MAXC + (ok_ & 255) + r) * 3;
                rec[0] = make_float4(a.x, a.y, a.z, b.x);
                rec[1] = make_float4(b.y, b.z, a.w, __int_as_float(meta | (pair << 24)));      // (bits 24..30: the pair, for the next step's matching)
                rec[2] = *(const float4 *)S->pair_mat[pair];
                // warm start (oracle warm_start_match()): the previous contact of the same bodies nearest to this one within the
                // margin, unless another new contact of those bodies is nearer to it (or as near and earlier in the list).
                // Pairs with the same bodies -- a moving shape against the statics -- are at most three consecutive ones, so
                // their contacts form one range of the previous list and one range of slots; all LDS reads of a stage are
                // issued together (a chain of dependent LDS round trips at the kernel's tail costs more than the arithmetic).
                float lam0 = 0.0f;
                {
                    const v3 x = mk(a.x, a.y, a.z);
                    const int key = meta, me = (ok_ & 255) + r;
                    int J0 = MAXC, J1 = 0;
                    {
                        int kq[5], rs[5], re[5];
#pragma unroll
                        for (int u = 0; u < 5; u++) {
                            const int q = min(max(pair - 2 + u, 0), P.npairs - 1);
                            kq[u] = pair_key[q]; rs[u] = run_s[q]; re[u] = run_e[q];
                        }
#pragma unroll
                        for (int u = 0; u < 5; u++) if (kq[u] == key && re[u] > rs[u]) { J0 = min(J0, rs[u]); J1 = max(J1, re[u]); }
                    }
                    int bj = -1;
                    float bd = 0.0004f;
                    for (int j0_ = J0; j0_ < J1; j0_ += 4) {
                        float4 pj[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) pj[u] = pv[min(j0_ + u, MAXC - 1)];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const v3 d = nc::sub(x, mk(pj[u].x, pj[u].y, pj[u].z));
                            const float d2 = nc::dot(d, d);
                            if (j0_ + u < J1 && d2 < bd) { bd = d2; bj = j0_ + u; }
                        }
                    }
                    if (bj >= 0) {
                        const float4 pb = pv[bj];
                        int k2[5], o2[5];
#pragma unroll
                        for (int u = 0; u < 5; u++) {
                            const int s2 = min(max(slot - 2 + u, 0), nchunk - 1);
                            k2[u] = res_key[s2]; o2[u] = res_k[s2];
                        }
                        bool heir = true;
#pragma unroll
                        for (int u = 0; u < 5; u++) {
                            const int s2 = slot - 2 + u;
                            if (s2 < 0 || s2 >= nchunk || k2[u] != key) continue;       // (wave-divergent, but the body's reads are issued together)
                            float4 a2[4];
#pragma unroll
                            for (int r2 = 0; r2 < 4; r2++) a2[r2] = res_a[s2][r2];
#pragma unroll
                            for (int r2 = 0; r2 < 4; r2++) {
                                const int other = (o2[u] & 255) + r2;
                                const v3 d = nc::sub(mk(a2[r2].x, a2[r2].y, a2[r2].z), mk(pb.x, pb.y, pb.z));
                                const float d2 = nc::dot(d, d);
                                if (r2 < (o2[u] >> 8) && other != me && (d2 < bd || (d2 == bd && other < me))) heir = false;
                            }
                        }
                        if (heir) lam0 = P.warmstart * (pb.w * P.dt);
                    }
                }
                D.cwarm_next[(size_t)env * MAXC + (ok_ & 255) + r] = lam0;
            }
        }
        CPROF(6);
        if (base + nchunk < ncl) {  // (more than CCHUNK close pairs: rare) the result slots are reused; every wave needs the count
            if (tid == 0) { nct_sh = nct; next_item = base + nchunk; }
            __syncthreads();
            nct = nct_sh;
        }
    }
    CPROF_END;
    if (tid == 0) {
        // "heavy": more generic contacts than P.heavy_min (an env with a few generic rows lengthens its wave's chain by a
        // quarter, an arm pressed on the table fourfold -- only the latter are worth the side stream when many envs have some)
        // generic contacts = all but the object-vs-static ones the solver's object lanes take: up to KOS = 4 per object and
        // P.os_cap in all, in list order (those pairs come first) -- the row builder's rule, so that a "light" env has none
        ngen = nct - min(P.os_cap, min(oscnt0, 4) + min(oscnt1, 4) + min(oscnt2, 4));
        const bool heavy = ngen > P.heavy_min;
        D.ccount_next[env] = nct;
        // (the very heavy list is capped at VH_MAX entries -- its solve is launched with one wave per entry; beyond that an env is
        // just "heavy": classes are scheduling, never a result)
        bool vh = heavy && ngen > P.heavy2_min;
        if (vh) {
            const int sl = atomicAdd(D.hcount2_next, 1);
            if (sl < VH_MAX) { D.hgflag_next[env] = 2; D.hlist2_next[sl] = env; D.hpos_next[env] = sl; }
            else { atomicSub(D.hcount2_next, 1); vh = false; }
        }
        if (heavy && !vh) { const int sl = atomicAdd(D.hcount_next, 1); D.hgflag_next[env] = 1; D.hlist_next[sl] = env; D.hpos_next[env] = sl; }
        else if (!heavy) D.hgflag_next[env] = 0;
    }
}
// One workgroup per env of the class `sel` (pick_env): N workgroups whatever the class (a heavy list's length is known on the
// device only; workgroups past its end, or whose env is not of the class, exit before touching anything).
// All envs (sel 0) in the order "longest first": the kernel lasts four rounds of workgroups and an env pressed on the table
// takes four times the mean (scratch/cprof.py), so the envs that were very heavy / heavy in the step just solved -- the lists of
// the CURRENT contact frame -- take the first h_first workgroups (a lagged host count + a margin; list entries beyond it and
// everybody else follow in env order: an env's place in its list, hpos, says on which side it is).  Every env exactly once.
#ifdef RR_RASTER_STATS
__device__ unsigned long long g_cwg_time[65536][3];
extern "C" int rr_debug_collide_wgtime(unsigned long long *out /*[65536][3]*/) { return (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_cwg_time), sizeof(g_cwg_time)) == hipSuccess) ? 0 : -1; }
#endif
__device__ __forceinline__ int collide_launch_env(const DevPtrs &D, int idx, int N, int h_first) {
    if (idx < h_first) {
        const int n2 = D.hcount2[0], n1 = D.hcount[0];
        return idx < n2 ? D.hlist2[idx] : (idx - n2 < n1 ? D.hlist[idx - n2] : -1);
    }
    const int env = idx - h_first;
    if (env >= N) return -1;
    const int g = D.hgflag[env];
    if (g != 0 && (g == 2 ? D.hpos[env] : D.hcount2[0] + D.hpos[env]) < h_first) return -1;      // (it was among the first)
    return env;
}
__global__ void __launch_bounds__(COLLIDE_THREADS, 4) k_collide(SimParams P, DevPtrs D, int ns, int sel, int h_first) {
    const int env = (sel == 0 && h_first > 0) ? collide_launch_env(D, blockIdx.x, P.N, h_first) : pick_env(D, sel, blockIdx.x, P.N);
    if (env < 0) return;
#ifdef RR_RASTER_STATS
    unsigned long long wt0_ = 0;      // (workgroup timeline of the collision pass: RR_ABLATE=131072, scratch/rwgtime.py collide)
    if (P.ablate & 0x20000) wt0_ = __builtin_amdgcn_s_memrealtime();
#endif
    collide_env(P, D, ns, env);
#ifdef RR_RASTER_STATS
    if ((P.ablate & 0x20000) && threadIdx.x == 0) {
        unsigned long long *g = g_cwg_time[env & 65535];
        g[0] = wt0_; g[1] = __builtin_amdgcn_s_memrealtime();
        g[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32);
    }
#endif
}
#undef CAND_ARGMAX
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------------- k_solve
__device__ __forceinline__ void plane_space(v3 n, v3 &p, v3 &q) {   // btPlaneSpace1
    if (fabsf(n.z) > 0.70710678118654752440f) {
        float a = n.y * n.y + n.z * n.z;
        float k = 1.0f / sqrtf(a);
        p = mk(0, -n.z * k, n.y * k);
        q = mk(a * k, -n.x * p.z, n.x * p.y);
    } else {
        float a = n.x * n.x + n.y * n.y;
        float k = 1.0f / sqrtf(a);
        p = mk(-n.y * k, n.x * k, 0);
        q = mk(-n.z * p.y, n.z * p.x, a * k);
    }
}

// ---- solver ------------------------------------------------------------------------------------------------------
// One env per 16-lane group (4 envs per wavefront, SGRP groups per workgroup).  The PGS chain of an env is sequential
// (Gauss-Seidel, Bullet's order: motors, limits, all normals, all lateral frictions, all torsional frictions), each SIMD
// runs a single wave, so the kernel lasts as long as the longest chain: what counts is the number of (dependent)
// instructions per row step.  Three kinds of rows:
//   * motor / joint-limit rows: lane j < 11 owns joint j (dq), rows in registers (MOTOR_STEP, LIMIT_STEP);
//   * object-vs-static contacts (objects resting on table / shelf): lane 11 + o owns the
//     velocity change (dv, dw) of object o and sweeps that object's rows on its own -- rows of different objects touch
//     disjoint variables, so the three object lanes run side by side and no cross-lane sum is needed; up to KOS contacts
//     per object, all six rows of each in registers (an object's further contacts with statics take the generic path);
//   * generic contacts (robot involved, two objects, object-static beyond os_cap): SLOT LAYOUT.  Every scalar velocity
//     variable of the env has a fixed (lane, slot):
//         slot A: lanes 0..10 joint velocities dq (the same register as the motor rows'), lanes 11..15 object 2 (v.xyz, w.xy)
//         slot B: lanes 0..5 object 0 (v.xyz, w.xyz), lanes 6..11 object 1, lane 12 object 2 (w.z)
//     and a row is one float4 per lane {J_A, (M^-1 J^T)_A, J_B, (M^-1 J^T)_B} (zeros where the row does not act): a row
//     step is one 16-byte load, two multiply-adds, a four-step DPP sum, the clamp, one DPP broadcast and two
//     multiply-adds -- no role logic, no per-row scalars in memory: rows are swept in blocks of 16, lane k of the group
//     holds {rhs, 1/diag, lambda, bounds} of the block's row k in registers, computes the clamp of "its" row from the
//     common J.v, and the impulse change of row k is lane k's value (DPP row_newbcast k).  The row data (256 B per row,
//     up to 6 rows per contact) does not fit LDS for an env with dozens of contacts; it lives in global memory
//     (D.grows, L2 resident), written once by the row builder and streamed through a register queue eight rows ahead.
//     Object velocities are moved between the object lanes and the slots around each generic sweep, only for the objects
//     that some generic contact of the wave touches.
//   Rows of contacts without normal impulse whose friction impulses are zero have the bounds [-0, 0] and cannot move
//   anything: the friction and torsional sweeps run over compacted lists of the other rows, rebuilt after every normal
//   sweep (94 % of the robot contacts of a pushing gripper are speculative: inside the margin, not touching).
// LDS per env (LF_TOTAL floats): Minv (121), motor rows (11 x {rhs, dinv, lambda}), joint-limit rows (22 x {rhs, lambda}),
// contact meta (48 ints), friction / spinning / rolling coefficients of the object-lane contacts (3 x 12), the objects' data (3 x 20),
// a staging area for 16 contact records, object-lane rows (12 x (3 x 12 linear + 3 x 8 torsional)), generic row scalars (288 x {rhs, dinv, bound coefficient, lambda}), the two row lists.
#define SGRP 4           // envs per workgroup (64 threads)
#define OS_CAP 12        // object-vs-static contacts per env on the object lanes: KOS per object, rows staged in LDS
#define GROWS (6 * MAXC) // generic row ids: 6 j + k for generic contact j; k = 0 normal, 1 2 lateral, 3 spinning, 4 5 rolling
// LF_ZPAD: the scalars of the DUMMY generic contacts MAXC .. MAXC + 3 (all zero: a block step on them is a no-op) -- what a group
// sweeps when another env of its wave has more blocks.  LF_LISTA / LF_LISTT: the active contacts of the friction / torsional
// passes, one byte each, padded with MAXC.
#define LIST_CAP (MAXC + 4)
enum { LF_MINV = 0, LF_MOT = LF_MINV + 124, LF_LIM = LF_MOT + 36, LF_META = LF_LIM + 44, LF_MU = LF_META + MAXC, LF_SPIN = LF_MU + OS_CAP,
       LF_ROLL = LF_SPIN + OS_CAP, LF_OBJ = LF_ROLL + OS_CAP, LF_CST = LF_OBJ + NOBJ * 20, LF_OSL = LF_CST + 16 * 12, LF_OST = LF_OSL + OS_CAP * 36,
       LF_GSC = LF_OST + OS_CAP * 24, LF_ZPAD = LF_GSC + 4 * GROWS, LF_LISTA = LF_ZPAD + 4 * 24, LF_LISTT = LF_LISTA + 16, LF_ENVW = LF_LISTT + 16, LF_TOTAL = LF_ENVW + 4 };
static_assert(LIST_CAP <= 64, "a list of LIST_CAP bytes must fit its 16 floats");
// Global store of the generic rows (D.grows), per env GP_RECS records of 16 lanes x float4:
//   GP_C + j       the normal row of generic contact j in the canonical layout {J_A, MJ_A, J_B, MJ_B} (block repack, warm start)
//   GP_N + 5 b     block of the normal rows of contacts 4 b .. 4 b + 3  {J pair, J pair, MJ pair, MJ pair, cross terms}
//   GP_T + 5 j     block of the torsional rows of contact j (spinning, rolling, rolling, -)
//   GP_F + 3 j     block of the lateral friction rows of contact j {J pair, MJ pair, cross term}
// Block index NBLK4 / contact index MAXC are the dummies: all-zero records nobody writes.
#define NBLK4 (MAXC / 4)
enum { GP_C = 0, GP_N = GP_C + MAXC, GP_T = GP_N + 5 * (NBLK4 + 1), GP_F = GP_T + 5 * (MAXC + 1), GP_RECS = GP_F + 3 * (MAXC + 1) };
#define GP_ENV_BYTES (GP_RECS * 256)
#define SLDS_FLOATS (SGRP * LF_TOTAL)
static_assert(SLDS_FLOATS * 4 * 4 <= 163840, "sixteen envs (four 64-thread workgroups or one 256-thread workgroup) must fit the 160 KiB LDS of a CU");
static_assert(LF_TOTAL % 4 == 0 && LF_OSL % 4 == 0 && LF_OST % 4 == 0 && LF_GSC % 4 == 0 && LF_OBJ % 4 == 0 && LF_CST % 4 == 0, "row parts must be 16-byte aligned");
extern __shared__ __attribute__((aligned(16))) float g_slds[];      // (blockDim.x / 16) x LF_TOTAL floats
#define LD(slot) g_slds[(slot)]

// meta word: bodyA (8) | bodyB (8) | linkA (8) | swept by the object lane (1) | object-lane slot or generic contact index (6)
__device__ __forceinline__ int meta_bodyA(int m) { return (signed char)(m & 255); }
__device__ __forceinline__ int meta_bodyB(int m) { return (signed char)((m >> 8) & 255); }
__device__ __forceinline__ int meta_link(int m) { return (signed char)((m >> 16) & 255); }
__device__ __forceinline__ bool meta_fast(int m) { return (m >> 24) & 1; }
__device__ __forceinline__ int meta_slot(int m) { return (m >> 25) & 63; }
__device__ __forceinline__ bool meta_near(int m) { return m < 0; }     // |distance| < 0.1 (robot.py:136)

// Sum over the 16 lanes of a group (= one DPP row), result in every lane: four rotate-and-add steps on the VALU
// (v_add_f32 with row_ror:8/4/2/1), no LDS crossbar traffic. Every lane performs the same commutative pairings, so
// all 16 lanes hold bitwise the same sum.
template <int ROR>
__device__ __forceinline__ float dpp_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + ROR, 0xf, 0xf, false));
}
// value of lane J of each 16-lane row, in every lane of that row (gfx90a+ DPP row_newbcast)
template <int J> __device__ __forceinline__ float row_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x150 + J, 0xf, 0xf, false));
}
template <int J> __device__ __forceinline__ int row_bcast_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xf, 0xf, false); }
// value of an arbitrary lane of the wavefront (ds_bpermute; the source lane must be active)
__device__ __forceinline__ float lane_gather(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int lane_gather_i(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ float group_sum(float v) {
    v += dpp_ror<8>(v);
    v += dpp_ror<4>(v);
    v += dpp_ror<2>(v);
    v += dpp_ror<1>(v);
    return v;
}

// ancestors-or-self of body b as a bit mask over joints (kinematic tree PARENT)
__device__ constexpr unsigned ANC[NB] = {0x001, 0x003, 0x007, 0x00f, 0x01f, 0x03f, 0x07f, 0x0ff, 0x1ff, 0x27f, 0x67f};

// Per-object data a contact row needs (pose, world inverse inertia, unconstrained velocities), fetched once per pair.
struct ObjData { v3 op, vs, ws; m3 Iinv; float imass; };

// What a lane owns in the slot layout: its object (or -1) and the unit vectors that pick its component out of a
// (linear, angular) pair -- J = lin . el + ang . ea.
struct SlotOwner { int objA, objB; v3 elA, eaA, elB, eaB; };
__device__ __forceinline__ v3 unit3(int c) { return mk(c == 0 ? 1.0f : 0.0f, c == 1 ? 1.0f : 0.0f, c == 2 ? 1.0f : 0.0f); }
__device__ __forceinline__ SlotOwner slot_owner(int l) {
    SlotOwner s;
    const int cA = l - NB;                                   // slot A, lanes 11..15: object 2, components 0..4
    s.objA = l >= NB ? 2 : -1;
    s.elA = unit3(cA < 3 ? cA : -1); s.eaA = unit3(cA >= 3 ? cA - 3 : -1);
    const int cB = l < 6 ? l : (l < 12 ? l - 6 : 5);         // slot B: object 0 | object 1 | object 2 component 5
    s.objB = l < 6 ? 0 : (l < 12 ? 1 : (l == 12 ? 2 : -1));
    s.elB = unit3(cB < 3 ? cB : -1); s.eaB = unit3(cB >= 3 ? cB - 3 : -1);
    return s;
}

// Generic rows in the slot layout, built per contact in two stages so that a row costs a handful of dot products:
//   stage 1 (once per contact): for this lane's slot-A and slot-B variable the vectors g, h with
//       J = dir . g,   (M^-1 J^T) = dir . h        for a linear row along `dir` at the contact point
//     and gt, ht likewise for a torsional row about `dir`.  Joint lane l (slot A): g = +-(a_l x (x - p_l)) for the ancestors
//     of the contact's robot body, gt = +-a_l, M^-1 J^T by the row of Minv; an object component lane: with e the unit vector
//     of its component, r = x - object position, s = +1 (body A) / -1 (body B):
//       linear component    g = s e,            h = s e / m
//       angular component   g = s (e x r),      h = s ((I^-1 e) x r)      [(r x d) . e = d . (e x r), I^-1 symmetric]
//     torsional row:        gt = s e (angular components only), ht = s I^-1 e
//   stage 2 (per row): four dot products, the 11-step Minv product for the joint lanes, two group sums, one store.
// Branch-free: the four envs of a wave have different body types.
struct GrowCtx { v3 gA, hA, gtA, htA, gB, hB, gtB, htB; };
__device__ __forceinline__ void grow_slot(int myobj, v3 el, v3 ea, int bodyA, int bodyB, v3 x, const ObjData &oA, const ObjData &oB,
                                          v3 &g, v3 &h, v3 &gt, v3 &ht) {
    const bool isA = myobj >= 0 && bodyA == 16 + myobj, isB = myobj >= 0 && bodyB == 16 + myobj;
    const float s = isA ? 1.0f : (isB ? -1.0f : 0.0f);
    const v3 op = isA ? oA.op : oB.op;
    const float im = isA ? oA.imass : oB.imass;
    m3 Ii;
#pragma unroll
    for (int k = 0; k < 9; k++) Ii.m[k] = isA ? oA.Iinv.m[k] : oB.Iinv.m[k];
    const v3 r = x - op;
    const v3 Iea = mulv(Ii, ea);
    g = (el + cross(ea, r)) * s;
    h = (el * im + cross(Iea, r)) * s;
    gt = ea * s;
    ht = Iea * s;
}

// Row step of an object-vs-static contact row held in registers, executed by the lane that owns the object only:
// b0,b1,b2 = the row (dir.xy dir.z ang.x | ang.yz dir.z mang.x | mang.yz rhs dinv), lambda in `lam`, bounds [lo, hi]; the
// object's velocity change lives in three register pairs V01 = (v.x v.y), V23 = (v.z w.x), V45 = (w.y w.z): J.v is three
// packed multiply-adds and one add, the update three packed FMAs (v_pk_fma_f32: two lanes' worth of fp32 per issue slot).
// An all-zero row with lam = 0 is a
// no-op (dl = 0), which is how absent rows are represented -- no predicates, no scalar mask registers.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#define P2(a, b) (v2f{(a), (b)})
#define REG_ROW_STEP(b0, b1, b2, lam, lo, hi)                                                                     \
    do {                                                                                                          \
        v2f p_ = P2((b0).x, (b0).y) * V01;                                                                        \
        p_ = pk_fma(P2((b0).z, (b0).w), V23, p_);                                                                 \
        p_ = pk_fma(P2((b1).x, (b1).y), V45, p_);                                                                 \
        const float jv_ = p_.x + p_.y;                                                                            \
        const float s0_ = fmaf(-jv_, (b2).w, (lam) + (b2).z);       /* (lambda + rhs) - dinv * J.v */             \
        const float sum_ = __builtin_amdgcn_fmed3f(s0_, (lo), (hi));                                              \
        const float dl_ = sum_ - (lam);                                                                           \
        (lam) = sum_;                                                                                             \
        const float sm_ = dl_ * inv_mass;                                                                         \
        V01 = pk_fma(P2((b0).x, (b0).y), P2(sm_, sm_), V01);                                                      \
        V23 = pk_fma(P2((b1).z, (b1).w), P2(sm_, dl_), V23);                                                      \
        V45 = pk_fma(P2((b2).x, (b2).y), P2(dl_, dl_), V45);                                                      \
    } while (0)
// (the components of the pairs by name)
#define DVX V01.x
#define DVY V01.y
#define DVZ V23.x
#define DWX V23.y
#define DWY V45.x
#define DWZ V45.y
#define KLIM 2           // joint-limit rows kept in registers (usually two: the finger lower limits)
#define KOS 4            // object-vs-static contacts per object kept in registers (a resting object has <= 4)

__device__ __forceinline__ float4 sel4(bool has, float4 v) {
    return make_float4(has ? v.x : 0.0f, has ? v.y : 0.0f, has ? v.z : 0.0f, has ? v.w : 0.0f);
}

// sel 0: wave b handles the envs 4 b .. 4 b + 3; 1: the same, but heavy envs are left out (their 16 lanes run along as no-ops);
// 2: wave w of the launch handles the heavy envs 4 w .. 4 w + 3 of D.hlist -- packed four to a wave whichever groups they
// come from (256-thread workgroups: sixteen heavy envs fill a CU's LDS and leave the other CUs to the render of the light
// envs).  No result depends on which envs share a wave: only trip counts and the choice between equivalent code paths do.
// One object-vs-static contact row pair along / about `dir` (k = 0: the normal, 1 2: the tangents): the linear row
// {dir, ang, I^-1 ang, rhs, 1/diag} and the torsional row about the same axis {I^-1 dir, rhs, 1/diag}, written to the LDS
// row areas of slot `slot`.  Used by the sequential row builder (lanes 0..2 of a group take k = lane) and by the
// lane-per-contact builder of the light form (one lane, k = 0..2): the same expressions, hence the same bits.
__device__ __forceinline__ void os_row_pair(int k, v3 dir, v3 x, const ObjData &oA, float dist, float rest, float spin, float roll, float lam0,
                                            float dt, float erp, float rest_thresh, int row_osl, int row_ost) {
    const v3 ang = cross(x - oA.op, dir);
    const v3 mang = mulv(oA.Iinv, ang);
    const float diag = dot(dir, dir) * oA.imass + dot(ang, mang);
    const float rel = dot(dir, oA.vs) + dot(ang, oA.ws);
    const float dinv = diag > 0 ? 1.0f / diag : 0.0f;
    float rr = 0;
    if (fabsf(rel) >= rest_thresh) { rr = rest * -rel; if (rr < 0) rr = 0; }
    float verr = rr - rel, perr = 0;
    if (dist > 0) verr -= dist / dt;
    else perr = -dist * erp / dt;
    const float rhs = k == 0 ? (perr + verr) * dinv : -rel * dinv;
    // torsional: pure rotation about dir; absent (coefficient 0) rows are all-zero
    const float coef = k == 0 ? spin : roll;
    const v3 tm = mulv(oA.Iinv, dir);
    const float tdiag = dot(dir, tm);
    const float tdinv = (coef > 0 && tdiag > 0) ? 1.0f / tdiag : 0.0f;
    const float trhs = -dot(dir, oA.ws) * tdinv;
    // (layout: REG_ROW_STEP / REG_TORS_STEP take the row in aligned pairs -- packed fp32 operations on (v.x v.y) (v.z w.x) (w.y w.z))
    float4 *bp4 = (float4 *)&LD(row_osl);
    bp4[0] = make_float4(dir.x, dir.y, dir.z, ang.x);
    bp4[1] = make_float4(ang.y, ang.z, dir.z, mang.x);
    bp4[2] = make_float4(mang.y, mang.z, rhs, dinv);
    float4 *tp4 = (float4 *)&LD(row_ost);
    tp4[0] = make_float4(tm.x, trhs, tm.y, tm.z);
    tp4[1] = make_float4(tdinv, k == 0 ? lam0 : 0.0f, 0.0f, 0.0f);      // (.y of the normal row: warm-start impulse)
}

// ---------------------------------------------------------------------------------------------- render setup
// Instance transforms for the rasteriser, one thread per (env, instance): a robot-link thread composes the joint
// transforms of its body's ancestors only (same operations, in the same order, as fk_all() for that chain), an object
// thread converts the object's quaternion; the 12 floats of an instance are stored as three 16-byte words, so a wave
// writes a contiguous span.  (One thread per env needed 264 stores with a 1.5 KB stride between lanes.)
// instance i of env: FK of its owner's ancestor chain -> model-view-projection matrix and shading constants (D.inst_xf).
// qj: the robot's joint angles; op7: pose {x y z, quaternion} of the instance's object (read for an object instance only).
// Two callers: k_render_setup (values from the state) and the light solve, which runs it on the values it has just
// integrated (registers / LDS) -- the same operations on the same floats, hence the same bits.
__device__ __forceinline__ void instance_setup_core(const BodyParams &B, const RenderModel &RM, const DevPtrs &D, int env, int i, int ot, int oi,
                                                    const float (&qj)[NB], const float (&op7)[7]) {
    m3 R = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    v3 p = mk(0, 0, 0);
    if (ot == 1) {
        unsigned anc = 0;
#pragma unroll
        for (int b = 0; b < NB; b++) if (oi == b) anc = ANC[b];
        p = mk(B.robot_pos[0], B.robot_pos[1], B.robot_pos[2]);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            if (!((anc >> b) & 1u)) continue;
            m3 jr;
#pragma unroll
            for (int k = 0; k < 9; k++) jr.m[k] = B.jrot[b][k];
            const m3 Rj = nc::mul(R, jr);
            const v3 ax = mk(B.axis[b][0], B.axis[b][1], B.axis[b][2]);
            p = nc::add(p, nc::mulv(R, mk(B.jpos[b][0], B.jpos[b][1], B.jpos[b][2])));
            R = nc::mul(Rj, nc::axis_angle(ax, qj[b]));
        }
    } else if (ot == 2) {
        R = nc::quat_to_m3(op7[3], op7[4], op7[5], op7[6]);
        p = mk(op7[0], op7[1], op7[2]);
    }
    // mvp = VP * [R p; 0 1] (same summation order as the oracle's 4x4 product, no FMA contraction), then the shading
    // constants: the raster and shading workgroups just copy these 128 bytes per instance into LDS
    float mvp[16];
    {
#pragma clang fp contract(off)
        const float xf[12] = {R.m[0], R.m[1], R.m[2], R.m[3], R.m[4], R.m[5], R.m[6], R.m[7], R.m[8], p.x, p.y, p.z};
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int r = e >> 2, c = e & 3;
            float a = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) a += RM.VP[4 * r + k] * (c < 3 ? xf[3 * k + c] : xf[9 + k]);
            a += RM.VP[4 * r + 3] * (c == 3 ? 1.0f : 0.0f);
            mvp[e] = a;
        }
    }
    const int tidx = RM.in_tex[i];
    float4 *o = (float4 *)(D.inst_xf + ((size_t)env * MAXINST + i) * 32);
    o[0] = make_float4(mvp[0], mvp[1], mvp[2], mvp[3]);
    o[1] = make_float4(mvp[4], mvp[5], mvp[6], mvp[7]);
    o[2] = make_float4(mvp[8], mvp[9], mvp[10], mvp[11]);
    o[3] = make_float4(mvp[12], mvp[13], mvp[14], mvp[15]);
    o[4] = make_float4(R.m[0], R.m[1], R.m[2], R.m[3]);
    o[5] = make_float4(R.m[4], R.m[5], R.m[6], R.m[7]);
    o[6] = make_float4(R.m[8], RM.in_color[i][0], RM.in_color[i][1], RM.in_color[i][2]);
    o[7] = make_float4(__int_as_float(tidx >= 0 ? RM.tex_off[tidx] : 0), __int_as_float(tidx >= 0 ? RM.tex_w[tidx] : 0),
                       __int_as_float(tidx >= 0 ? RM.tex_h[tidx] : 0), __int_as_float(RM.in_uid[i]));
}
__device__ __forceinline__ void instance_setup(const BodyParams &B, const SimParams &P, const RenderModel &RM, const DevPtrs &D, int env, int i) {
    const int N = P.N;
    const float *state = D.state;
    const int ot = RM.in_otype[i], oi = RM.in_oidx[i];
    float qj[NB] = {0}, op7[7] = {0};
    if (ot == 1) {                 // all joint angles requested up front: one round trip instead of one per ancestor
#pragma unroll
        for (int b = 0; b < NB; b++) qj[b] = STT(ST_Q + b);
    } else if (ot == 2) {
#pragma unroll
        for (int k = 0; k < 3; k++) op7[k] = STT(ST_OPOS + 3 * oi + k);
#pragma unroll
        for (int k = 0; k < 4; k++) op7[3 + k] = STT(ST_OQUAT + 4 * oi + k);
    }
    instance_setup_core(B, RM, D, env, i, ot, oi, qj, op7);
}

// GEN = false is the form for the light envs (sel 1): no generic contact row can occur there (k_collide classifies by the
// very rule the row builder uses), so everything of the generic path -- the row builder, the streamed sweeps, their register
// queue -- folds away at compile time and the sweep of motors, limits and object-lane rows runs without the register
// spills (v_accvgpr_read: a third of the torsional steps' instructions) the full kernel needs.  Same source, same
// arithmetic: results do not depend on which form solved an env (split-equivalence tests, bitwise).
// OW ("object wave", light form only): workgroups of five waves solve sixteen envs -- waves 0..3 are the envs' 16-lane groups
// (command part, row build, the robot's rows, joint integration), wave 4 runs the object chains of all sixteen envs, one LANE per
// (env, object) (light_object_wave below).  In a light env the robot's rows and each object's rows are separate problems;
// with the object rows on lanes 11..13 of every group 62 % of a sweep's instructions ran with 3 of 16 lanes live.
__device__ void light_object_wave(const BodyParams &B, const SimParams &P, const DevPtrs &D, const RenderModel *RMp);
#ifndef LIGHT_OW_THREADS
#define LIGHT_OW_THREADS 384
#endif
#define OW_FLAGS 0       // L_GSC words of an env that its group publishes for the object wave: flags (1 light env of this launch, 2 does not step)
#define OW_MASK 1        // .. +3: the contacts (bit = list index) of object 0 / 1 / 2
#define OW_QUAT 4        // .. +12: the objects' orientations at the start of the step (after the out-of-bounds rule)
template <bool GEN, bool OW = false>
__device__ __forceinline__ void solve_body(const BodyParams &B, const SimParams &P, const DevPtrs &D, int sel, int coop_launch, const RenderModel *RMp = nullptr) {
    static_assert(!(GEN && OW), "the object wave belongs to the light form");
    const int N = P.N;
    if (sel <= 1 && blockIdx.x == 0 && threadIdx.x == 0) {
        // (once per step, by the launch every step has: the bookkeeping of the contact frame this step's look-ahead will fill;
        // the number of heavy / very heavy envs of this step goes to pinned host memory on the way -- a posted write)
        if (D.hcount_host) { D.hcount_host[0] = D.hcount[0]; D.hcount_host[1] = D.hcount2[0]; }
        D.hcount_next[0] = 0; D.hcount_next[1] = 0; D.hcount2_next[0] = 0; D.hcount2_next[1] = 0;
    }
#if LIGHT_OW_THREADS == 384
    // wave roles O J J J - J: waves w and w + 4 of a workgroup land on the same SIMD (tools/ubench/simd_map.hip), so the object
    // wave has its SIMD to itself -- wave 4 only keeps the workgroup's barrier company and ends -- and two of the four group
    // waves share one (together as many issue slots as the object wave).  Object wave 53 -> 37 us (scratch/sprof_light.py), the
    // kernel 56.7 -> 54.3 us, the step 0.691 -> 0.685 ms.  (LIGHT_OW_THREADS 320: J J J J O, the object wave shares with a group wave.)
    int jwave = 0;
    if (OW) {
        const int wv = threadIdx.x >> 6;
        if (wv == 0) { light_object_wave(B, P, D, RMp); return; }
        if (wv == 4) { __syncthreads(); return; }
        jwave = wv < 4 ? wv - 1 : 3;
    }
    const int grp = OW ? 4 * jwave + ((threadIdx.x >> 4) & 3) : threadIdx.x >> 4, l = threadIdx.x & 15;
#else
    if (OW && (threadIdx.x >> 6) == 4) { light_object_wave(B, P, D, RMp); return; }
    const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
#endif
    const int unit = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // the wave's index in the launch
    // coop (a heavy / very heavy list of at most COOP_MAX envs -- the host's lagged count decides, any actual count is handled):
    // ONE env per wave -- its four 16-lane groups build the rows of four contacts at a time (the row build of an env at the
    // contact cap is a fifth of its chain with 16 lanes); group 0 then sweeps, groups 1..3 run along as no-ops.  The wave has
    // one LDS region (the launch asks for a quarter of the LDS of the packed form: the same sixteen envs per CU).  Same
    // arithmetic per row whichever group builds it: results do not depend on the mode (tested bitwise).
    const bool coop = GEN && coop_launch && (sel >= 2 || sel == 0);    // (sel 0: every env its own wave -- batches of at most one wave per SIMD, rr_step)
    const int cg = coop ? (grp & 3) : 0;                              // this group's place among the builders of its env
    int env_raw = OW ? 16 * (int)blockIdx.x + grp : (coop ? unit : 4 * unit + (grp & 3));
    bool mine = true;                                                 // this 16-lane group has an env to solve in this launch
    if (sel == 2) { mine = env_raw < *D.hcount; env_raw = mine ? D.hlist[env_raw] : N; }
    else if (sel == 3) { mine = env_raw < *D.hcount2; env_raw = mine ? D.hlist2[env_raw] : N; }
    else if (sel == 1) mine = env_raw < N && D.hgflag[env_raw] == 0;
    if (!OW && __ballot(mine) == 0ull) return;                        // (wave-uniform; OW: every wave goes to the workgroup's barrier)
    int env = env_raw < N ? env_raw : N - 1;                          // groups without an env run along as no-ops
    float *state = D.state, *scratch = D.scratch;
    const int lj = l < NB ? l : 0;               // joint owned by this lane (lanes >= 11 alias joint 0, masked)
    const int lo_ = (l >= NB && l < NB + NOBJ) ? l - NB : -1;   // object owned by this lane
    const ShapeData *S = D.shapes;
    const int fix = (coop ? (grp >> 2) : grp) * LF_TOTAL;            // (coop: the four groups of a wave share its one LDS region)
    const int L_MINV = fix + LF_MINV, L_MOT = fix + LF_MOT, L_LIM = fix + LF_LIM, L_META = fix + LF_META, L_MU = fix + LF_MU,
              L_SPIN = fix + LF_SPIN, L_ROLL = fix + LF_ROLL, L_OSL = fix + LF_OSL, L_OST = fix + LF_OST, L_GSC = fix + LF_GSC,
              L_OBJ = fix + LF_OBJ, L_CST = fix + LF_CST;
    unsigned char *listA = (unsigned char *)&LD(fix + LF_LISTA), *listT = (unsigned char *)&LD(fix + LF_LISTT);
    const float dt = P.dt, inv_dt = 1.0f / P.dt;
    SPROF_INIT
    SBLK_BEGIN
    // ---- stage-in: EVERY global load of the env's inputs is issued here, before anything is computed from them or stored (a
    // global round trip costs ~7000 cycles at one wave per SIMD, and a store in between would hold the later loads back):
    // error flags, command, the lane's joint state and frame, M^-1 (staged in LDS below), the contact count, the first 16
    // contact records with their inherited impulses, the object lanes' data.
    const unsigned ef0 = (mine && env_raw < N) ? D.errflags[env] : 1u;
    const int lc = l < 9 ? l : 0;
    const float cmd_v = D.cmd_in[(size_t)env * 9 + lc];
    const float act_md = B.act_maxdiff[lc], act_lo = B.act_min[lc], act_hi = B.act_max[lc];
    const float q_l = STT(ST_Q + lj), qds_l = SCR(S_QDS + lj);
    float minv_stage[8];
#pragma unroll
    for (int i = 0; i < 8; i++) minv_stage[i] = SCR(S_MINV + min(l + 16 * i, NB * NB - 1));
    const v3 pk_l = mk(SCR(S_BP + 3 * lj), SCR(S_BP + 3 * lj + 1), SCR(S_BP + 3 * lj + 2));
    const v3 ak_l = mk(SCR(S_BAX + 3 * lj), SCR(S_BAX + 3 * lj + 1), SCR(S_BAX + 3 * lj + 2));
    const int ccount_in = D.ccount[env];
    const int cls_in = sel == 0 ? D.hgflag[env] : sel - 1;           // (sel 1 / 2 / 3: the launch's class)
    float cw0, cw1, cw2;              // warm-start impulses of contacts l, 16 + l, 32 + l
    float4 cr0, cr1, cr2, cr3, cr4, cr5, cr6, cr7, cr8;   // lane l's float4 #(l + 16 i) of the list: cr0..2 = the first 16 records
    {
        const float4 *cl = D.clist + (size_t)env * MAXC * 3;
        cr0 = cl[l]; cr1 = cl[16 + l]; cr2 = cl[32 + l];
        cr3 = cr4 = cr5 = cr6 = cr7 = cr8 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (GEN && __ballot(ccount_in > 16) != 0ull) {  // (a light env has at most 12 contacts, all on the object lanes; a heavy one
            cr3 = cl[48 + l]; cr4 = cl[64 + l]; cr5 = cl[80 + l];     //  with more than 16 pays a second round trip for the other 32 records)
            cr6 = cl[96 + l]; cr7 = cl[112 + l]; cr8 = cl[128 + l];
        }
        const float *cwp = D.cwarm + (size_t)env * MAXC;
        cw0 = cwp[l]; cw1 = cwp[16 + l]; cw2 = cwp[32 + l];       // (unconditional: selected by the count below)
    }
    // the object lanes: the pose that counts (k_prep: the home pose when the out-of-bounds rule fires), inverse inertia,
    // unconstrained velocities; the raw position decides whether the rule fires (then this kernel writes the home pose into the state)
    ObjData myobj;
    const int ob_l = (l >= NB && l < NB + NOBJ && l - NB < P.nobj) ? l - NB : 0;
    const float raw_x = STT(ST_OPOS + 3 * ob_l), raw_z = STT(ST_OPOS + 3 * ob_l + 2);
    float oquat[4];                   // orientation at the start of the step (replaced by the home orientation below if the rule fires)
#pragma unroll
    for (int k = 0; k < 4; k++) oquat[k] = STT(ST_OQUAT + 4 * ob_l + k);
    myobj.op = mk(SCR(S_OP + 3 * ob_l), SCR(S_OP + 3 * ob_l + 1), SCR(S_OP + 3 * ob_l + 2));
#pragma unroll
    for (int kk = 0; kk < 9; kk++) myobj.Iinv.m[kk] = SCR(S_OIINV + 9 * ob_l + kk);
    myobj.vs = mk(SCR(S_OVS + 3 * ob_l), SCR(S_OVS + 3 * ob_l + 1), SCR(S_OVS + 3 * ob_l + 2));
    myobj.ws = mk(SCR(S_OWS + 3 * ob_l), SCR(S_OWS + 3 * ob_l + 1), SCR(S_OWS + 3 * ob_l + 2));
    myobj.imass = 1.0f / (ob_l == 0 ? B.obj_mass[0] : (ob_l == 1 ? B.obj_mass[1] : B.obj_mass[2]));
    // ---- the command part of the step -- everything that needs the action: limitActionByJoint (env.py:314-321), the clipping
    // and gripper coupling of Kuka.apply_action (robot.py:188-201) -> the motor target of this lane's joint; a non-finite
    // command flags the env (robot.py:189 asserts) and the env does not step.  It also APPLIES the out-of-bounds rule
    // (env.py:257-264) to the state: the state part of the preparation only derived the collision inputs from the re-posed object.
    float tgt_l = 0.0f;
    bool rejected = false;
    {
        const float cur = l == 8 ? -q_l : q_l;                           // robot.py:203-211 (lanes 0..8: q_l is joint l)
        float d = cmd_v - cur;                                           // env.py:314-321
        d = fminf(d, act_md);
        d = fmaxf(d, -act_md);
        float a = cur + d;
        a = fmaxf(act_lo, fminf(a, act_hi));                             // robot.py:192
        const float a7 = row_bcast<7>(a);
        if (l == 8) a = fmaxf(0.0f, fminf(2.0f * a7, a));                // robot.py:193
        const float a8 = row_bcast<8>(a);
        tgt_l = l < 7 ? a : ((l == 7 || l == 9) ? a7 : -a8);             // robot.py:195-201 (lanes 8, 10: -a8)
        rejected = ((unsigned)(__ballot(l < 9 && !isfinite(cmd_v)) >> (16 * (grp & 3))) & 0xffffu) != 0u;
    }
    const bool frozen = (ef0 & 1u) != 0u;
    bool dead = frozen || (ef0 & ~11u) != 0u || rejected;      // (bit 8 is a render status: it never stops the physics)
    const int nct = dead ? 0 : min(ccount_in, MAXC);
    cw0 = l < nct ? cw0 : 0.0f; cw1 = 16 + l < nct ? cw1 : 0.0f; cw2 = 32 + l < nct ? cw2 : 0.0f;
    // ---- and only now the stores of the command part
    if (!frozen && l == 0) D.errflags[env] = rejected ? ((ef0 & ~2u) | 2u) : (ef0 & ~2u);
    // an env whose command was rejected does not step; the list the look-ahead made for this step is dropped with it, so that
    // the normal forces of the last solved step (cforce) are never matched against a list they do not belong to
    if (!frozen && rejected && l == 0) D.ccount[env] = 0;
    if (l == 0 && mine && env_raw < N) { D.ccount_pub[env] = (!frozen && rejected) ? 0 : ccount_in; D.class_pub[env] = cls_in; }
    if (!dead) {
        if (l < NB) STT(ST_TGT + l) = tgt_l;
        if (lo_ >= 0 && lo_ < P.nobj && object_out_of_bounds(raw_x, raw_z, B.table_z)) {       // env.py:257-264
            const int i = lo_;
            for (int k = 0; k < 3; k++) { STT(ST_OPOS + 3 * i + k) = D.obj_home[(size_t)(7 * i + k) * N + env]; STT(ST_OVEL + 3 * i + k) = 0; STT(ST_OANG + 3 * i + k) = 0; }
#pragma unroll
            for (int k = 0; k < 4; k++) { oquat[k] = D.obj_home[(size_t)(7 * i + 3 + k) * N + env]; STT(ST_OQUAT + 4 * i + k) = oquat[k]; }
        }
    }
    // M^-1 goes to LDS (row builders, limit rows)
#pragma unroll
    for (int i = 0; i < 8; i++) if (l + 16 * i < NB * NB) LD(L_MINV + l + 16 * i) = minv_stage[i];
    float wsA = 0.0f, wsB = 0.0f;     // slot-layout velocity change of the warm-start impulses of the generic normal rows
    // lanes 11..13 publish "their" object's data in LDS: the row builder reads the data of a contact's objects from there
    // (20 floats per object: position, 1/mass, I^-1, v*, w*)
    if (lo_ >= 0) {
        float4 *od = (float4 *)&LD(L_OBJ + 20 * lo_);
        od[0] = make_float4(myobj.op.x, myobj.op.y, myobj.op.z, myobj.imass);
        od[1] = make_float4(myobj.Iinv.m[0], myobj.Iinv.m[1], myobj.Iinv.m[2], myobj.Iinv.m[3]);
        od[2] = make_float4(myobj.Iinv.m[4], myobj.Iinv.m[5], myobj.Iinv.m[6], myobj.Iinv.m[7]);
        od[3] = make_float4(myobj.Iinv.m[8], myobj.vs.x, myobj.vs.y, myobj.vs.z);
        od[4] = make_float4(myobj.ws.x, myobj.ws.y, myobj.ws.z, 0.0f);
    }
    // row l of Minv in registers (motor rows, and M^-1 J^T of the generic rows)
    float minv_l[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) minv_l[j] = l < NB ? LD(L_MINV + lj * NB + j) : 0.0f;
    // the unconstrained velocities in the slot layout (relative velocity of a generic row = sum over lanes of J . u)
    const SlotOwner so = slot_owner(l);
    float ua = l < NB ? qds_l : 0.0f, ub = 0.0f;
    {
#define U_FROM(O)                                                                                                  \
        {                                                                                                          \
            const v3 vs_ = mk(row_bcast<NB + (O)>(myobj.vs.x), row_bcast<NB + (O)>(myobj.vs.y), row_bcast<NB + (O)>(myobj.vs.z));     \
            const v3 ws_ = mk(row_bcast<NB + (O)>(myobj.ws.x), row_bcast<NB + (O)>(myobj.ws.y), row_bcast<NB + (O)>(myobj.ws.z));     \
            ua = so.objA == (O) ? dot(vs_, so.elA) + dot(ws_, so.eaA) : ua;                                        \
            ub = so.objB == (O) ? dot(vs_, so.elB) + dot(ws_, so.eaB) : ub;                                        \
        }
        U_FROM(0) U_FROM(1) U_FROM(2)
#undef U_FROM
    }
    SPROF(0);
    // ---- walk the contact list in order, build rows (all lanes of the group run the control flow redundantly): 16 records at
    // a time go through the LDS staging area, from where every lane reads the record of the contact at hand
    int n_os = 0;                     // object-vs-static contacts swept by the object lanes (slots 0 .. n_os-1 of the LDS staging area)
    unsigned own_os = 0;              // this lane's share of them (bit = slot)
    unsigned fcnt = 0;                // ... per object, 4 bits each
    unsigned gobj = 0;                // objects touched by generic contacts of this env (bit o)
    int nc = 0, ng = 0;               // contacts; generic contacts
    bool class_mismatch = false;      // GEN = false only: a contact that needs generic rows
    const int nct_max = max(max(__builtin_amdgcn_readlane(nct, 0), __builtin_amdgcn_readlane(nct, 16)),
                            max(__builtin_amdgcn_readlane(nct, 32), __builtin_amdgcn_readlane(nct, 48)));
    if (!GEN) {
        // ---- light form: every contact is an object-vs-static one on the object lanes (<= 12 of them), so there is no slot
        // to hand out in list order -- contact c is slot c -- and LANE c builds all six rows of contact c: one trip instead of
        // one per contact (each a chain of LDS round trips at one wave per SIMD).  Same expressions as the walk below
        // (os_row_pair), same bits.
        {
            float4 *st = (float4 *)&LD(L_CST);
            st[l] = cr0; st[16 + l] = cr1; st[32 + l] = cr2;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool have = l < nct;
        const int ci = l;
        const float4 ra = *(const float4 *)&LD(L_CST + 12 * ci), rb_ = *(const float4 *)&LD(L_CST + 12 * ci + 4), rc = *(const float4 *)&LD(L_CST + 12 * ci + 8);
        const v3 x = mk(ra.x, ra.y, ra.z), n = mk(ra.w, rb_.x, rb_.y);
        const float dist = rb_.z;
        const int cm = __float_as_int(rb_.w);
        const int bodyA = (signed char)(cm & 255), bodyB = (signed char)((cm >> 8) & 255), linkA = (signed char)((cm >> 16) & 255);
        const float mu = rc.x, rest = rc.y, roll = rc.z, spin = rc.w;
        const bool ospair = bodyA >= 16 && bodyB < 0;
        const int obA = bodyA >= 16 ? min(bodyA - 16, NOBJ - 1) : 0;
        ObjData oA;
        {
            const float4 *od = (const float4 *)&LD(L_OBJ + 20 * obA);
            const float4 d0 = od[0], d1 = od[1], d2 = od[2], d3 = od[3], d4 = od[4];
            oA.op = mk(d0.x, d0.y, d0.z); oA.imass = d0.w;
            oA.Iinv.m[0] = d1.x; oA.Iinv.m[1] = d1.y; oA.Iinv.m[2] = d1.z; oA.Iinv.m[3] = d1.w;
            oA.Iinv.m[4] = d2.x; oA.Iinv.m[5] = d2.y; oA.Iinv.m[6] = d2.z; oA.Iinv.m[7] = d2.w;
            oA.Iinv.m[8] = d3.x; oA.vs = mk(d3.y, d3.z, d3.w); oA.ws = mk(d4.x, d4.y, d4.z);
        }
        // which contacts belong to which object: the masks the object lanes own, and this contact's place among its object's
        const int gsh = 16 * (grp & 3);
        const unsigned m0 = (unsigned)(__ballot(have && ospair && obA == 0) >> gsh) & 0xffffu, m1 = (unsigned)(__ballot(have && ospair && obA == 1) >> gsh) & 0xffffu,
                       m2 = (unsigned)(__ballot(have && ospair && obA == 2) >> gsh) & 0xffffu;
        const unsigned mine_m = obA == 0 ? m0 : (obA == 1 ? m1 : m2);
        const int before = __popc(mine_m & ((1u << l) - 1u));
        // (the row builder's rule: an object-vs-static pair, among the first KOS of its object and the first P.os_cap in all)
        class_mismatch = have && !(ospair && before < KOS && l < P.os_cap);
        own_os = lo_ == 0 ? m0 : (lo_ == 1 ? m1 : (lo_ == 2 ? m2 : 0u));
        n_os = nct; nc = nct;
        if (have) {
            const int meta = (bodyA & 255) | ((bodyB & 255) << 8) | ((linkA & 255) << 16) | (1 << 24) | (ci << 25) |
                             (fabsf(dist) < 0.1f ? (int)0x80000000u : 0);        // robot.py:136 contact_threshold
            *(int *)&LD(L_META + ci) = meta;
            LD(L_MU + ci) = mu; LD(L_SPIN + ci) = spin; LD(L_ROLL + ci) = roll;
            v3 t1, t2;
            plane_space(n, t1, t2);
            const float lam0 = cw0;           // (this lane holds the warm-start impulse of contact l)
#pragma unroll
            for (int k = 0; k < 3; k++)
                os_row_pair(k, k == 0 ? n : (k == 1 ? t1 : t2), x, oA, dist, rest, spin, roll, lam0, dt, P.erp, P.rest_thresh,
                            L_OSL + (3 * ci + k) * 12, L_OST + (3 * ci + k) * 8);
        }
        if (OW) {
            // what the object wave needs besides the rows and the objects' data (L_OBJ): who steps, whose contacts, orientations
            if (l == 0) {
                *(int *)&LD(L_GSC + OW_FLAGS) = ((mine && env_raw < N) ? 1 : 0) | (dead ? 2 : 0);
                *(unsigned *)&LD(L_GSC + OW_MASK) = m0; *(unsigned *)&LD(L_GSC + OW_MASK + 1) = m1; *(unsigned *)&LD(L_GSC + OW_MASK + 2) = m2;
            }
            if (lo_ >= 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) LD(L_GSC + OW_QUAT + 4 * lo_ + k) = oquat[k];
            }
            own_os = 0;
            __syncthreads();
        } else {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        }
    } else
#pragma unroll 1        // one copy of the (large) row-building body: the kernel must stay inside the instruction cache
    for (int bt = 0; 16 * bt < nct_max; bt++) {
        {   // batch bt = records 16 bt .. 16 bt + 15 = float4 48 bt .. 48 bt + 47 of the list = cr(3 bt) .. cr(3 bt + 2) of the lanes
            float4 *st = (float4 *)&LD(L_CST);
#define STAGE_REC(I, A_, B_, C_)                                                                                        \
            st[16 * (I) + l] = make_float4(bt == 0 ? A_.x : (bt == 1 ? B_.x : C_.x), bt == 0 ? A_.y : (bt == 1 ? B_.y : C_.y), \
                                           bt == 0 ? A_.z : (bt == 1 ? B_.z : C_.z), bt == 0 ? A_.w : (bt == 1 ? B_.w : C_.w));
            STAGE_REC(0, cr0, cr3, cr6) STAGE_REC(1, cr1, cr4, cr7) STAGE_REC(2, cr2, cr5, cr8)
#undef STAGE_REC
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (coop: group cg takes contact ci0 + cg of every quadruple; otherwise one contact per trip)
        for (int ci0 = 0; ci0 < 16 && 16 * bt + ci0 < nct; ci0 += coop ? 4 : 1) {
            const int ci = min(ci0 + cg, 15);
            const bool have = 16 * bt + ci0 + cg < nct;             // (coop: the last quadruple may be short -- the group runs along, stores nothing)
            const float4 ra = *(const float4 *)&LD(L_CST + 12 * ci), rb_ = *(const float4 *)&LD(L_CST + 12 * ci + 4), rc = *(const float4 *)&LD(L_CST + 12 * ci + 8);
            const v3 x = mk(ra.x, ra.y, ra.z), n = mk(ra.w, rb_.x, rb_.y);
            const float dist = rb_.z;
            const int cm = __float_as_int(rb_.w);
            const int bodyA = (signed char)(cm & 255), bodyB = (signed char)((cm >> 8) & 255), linkA = (signed char)((cm >> 16) & 255);
            const float mu = rc.x, rest = rc.y, roll = rc.z, spin = rc.w;
            // warm start: this contact's initial normal impulse is held by lane ci of the group
            const float lam0 = lane_gather(bt == 0 ? cw0 : (bt == 1 ? cw1 : cw2), (threadIdx.x & 48) + ci);
            const bool ospair = bodyA >= 16 && bodyB < 0;
            ObjData oA, oB;
#pragma unroll
            for (int side = 0; side < 2; side++) {
                ObjData &o = side == 0 ? oA : oB;
                const int body = side == 0 ? bodyA : bodyB;
                const float4 *od = (const float4 *)&LD(L_OBJ + 20 * (body >= 16 ? body - 16 : 0));
                const float4 d0 = od[0], d1 = od[1], d2 = od[2], d3 = od[3], d4 = od[4];
                o.op = mk(d0.x, d0.y, d0.z); o.imass = d0.w;
                o.Iinv.m[0] = d1.x; o.Iinv.m[1] = d1.y; o.Iinv.m[2] = d1.z; o.Iinv.m[3] = d1.w;
                o.Iinv.m[4] = d2.x; o.Iinv.m[5] = d2.y; o.Iinv.m[6] = d2.z; o.Iinv.m[7] = d2.w;
                o.Iinv.m[8] = d3.x; o.vs = mk(d3.y, d3.z, d3.w); o.ws = mk(d4.x, d4.y, d4.z);
            }
            // the first KOS object-vs-static contacts of an object are swept by the object's lane, any further ones take the
            // generic path (rows of different objects commute, and an object's own rows keep their order: its pairs with
            // the statics precede every other pair).  The slots are handed out in list order: with four builders each group
            // replays the decisions of the quadruple's contacts (codes exchanged with v_readlane) and keeps the one of its own.
            const int obA = bodyA >= 16 ? bodyA - 16 : 0;
            const int code = have ? (1 | (ospair ? 2 : 0) | (obA << 2)) : 0;
            bool fast = false;
            int slot = 0, mync = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (j > 0 && !coop) break;
                const int cj = coop ? __builtin_amdgcn_readlane(code, 16 * j) : code;
                if (!(cj & 1)) continue;
                const int ob_j = (cj >> 2) & 3;
                const bool fast_j = (cj & 2) && ((fcnt >> (4 * ob_j)) & 15u) < KOS && n_os < P.os_cap;
                const int slot_j = fast_j ? n_os : ng;
                if (j == cg) { fast = fast_j; slot = slot_j; mync = nc; }
                if (fast_j) {
                    n_os++;
                    fcnt += 1u << (4 * ob_j);
                    if (lo_ >= 0 && ob_j == lo_) own_os |= 1u << slot_j;
                } else if (GEN) ng++;
                else class_mismatch = true;       // (cannot happen: k_collide counts generic contacts by this very rule)
                nc++;
            }
            const int meta = (bodyA & 255) | ((bodyB & 255) << 8) | ((linkA & 255) << 16) | ((fast ? 1 : 0) << 24) | (slot << 25) |
                             (fabsf(dist) < 0.1f ? (int)0x80000000u : 0);        // robot.py:136 contact_threshold
            if (l == 0 && have) {
                *(int *)&LD(L_META + mync) = meta;
                if (fast) { LD(L_MU + slot) = mu; LD(L_SPIN + slot) = spin; LD(L_ROLL + slot) = roll; }      // (generic rows carry their coefficient)
            }
            v3 t1, t2;
            plane_space(n, t1, t2);
            if (fast) {
                // the three linear rows (n, t1, t2) and the three torsional rows about the same axes are built by lanes 0, 1, 2
                if (l < 3 && have)
                    os_row_pair(l, l == 0 ? n : (l == 1 ? t1 : t2), x, oA, dist, rest, spin, roll, lam0, dt, P.erp, P.rest_thresh,
                                L_OSL + (3 * slot + l) * 12, L_OST + (3 * slot + l) * 8);
                continue;
            }
            if (!GEN || !have) continue;
            // generic contact j = slot: six rows in the slot layout
            const int r0 = 6 * slot;
            if (bodyA >= 16) gobj |= 1u << (bodyA - 16);
            if (bodyB >= 16) gobj |= 1u << (bodyB - 16);
            GrowCtx gc;
            {
                const int rb = bodyA >= 0 && bodyA < 16 ? bodyA : (bodyB >= 0 && bodyB < 16 ? bodyB : -1);
                const float sgr = rb < 0 ? 0.0f : (rb == bodyA ? 1.0f : -1.0f);
                const float jm = (l < NB && rb >= 0 && ((ANC[rb >= 0 ? rb : 0] >> l) & 1u)) ? sgr : 0.0f;     // joint lanes: sign x ancestor mask
                v3 gA, hA, gtA, htA;
                grow_slot(so.objA, so.elA, so.eaA, bodyA, bodyB, x, oA, oB, gA, hA, gtA, htA);
                grow_slot(so.objB, so.elB, so.eaB, bodyA, bodyB, x, oA, oB, gc.gB, gc.hB, gc.gtB, gc.htB);
                const v3 cj = cross(ak_l, x - pk_l) * jm, aj = ak_l * jm;
                gc.gA = l < NB ? cj : gA; gc.gtA = l < NB ? aj : gtA; gc.hA = hA; gc.htA = htA;
            }
            // the three linear rows (n, t1, t2), then the three torsional rows about the same axes: the rows of a triple are
            // independent, so their chains (the 11-step M^-1 product, the group sums) are evaluated side by side
#pragma unroll 1
            for (int tq = 0; tq < 2; tq++) {
                const bool tors = tq == 1;
                // torsional rows without a coefficient (robot link against table / shelf: most contacts of a crushed arm) are
                // never swept: only their (zero) scalars are read, when the sweep lists are built
                if (tors && __ballot(spin > 0 || roll > 0) == 0ull) {
                    if (l == 0) {
                        *(float4 *)&LD(L_GSC + 4 * (r0 + 3)) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        *(float4 *)&LD(L_GSC + 4 * (r0 + 4)) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        *(float4 *)&LD(L_GSC + 4 * (r0 + 5)) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    }
                    continue;
                }
                const v3 gA_ = tors ? gc.gtA : gc.gA, gB_ = tors ? gc.gtB : gc.gB, hA_ = tors ? gc.htA : gc.hA, hB_ = tors ? gc.htB : gc.hB;
                float ja3[3], jb3[3], mja3[3], mjb3[3];
#pragma unroll
                for (int ka = 0; ka < 3; ka++) {
                    const v3 d = ka == 0 ? n : (ka == 1 ? t1 : t2);
                    ja3[ka] = dot(d, gA_); jb3[ka] = dot(d, gB_); mjb3[ka] = dot(d, hB_);
                    mja3[ka] = 0.0f;
                }
#define MJA_STEP(J) { mja3[0] += minv_l[J] * row_bcast<J>(ja3[0]); mja3[1] += minv_l[J] * row_bcast<J>(ja3[1]); mja3[2] += minv_l[J] * row_bcast<J>(ja3[2]); }
                MJA_STEP(0) MJA_STEP(1) MJA_STEP(2) MJA_STEP(3) MJA_STEP(4) MJA_STEP(5) MJA_STEP(6) MJA_STEP(7) MJA_STEP(8) MJA_STEP(9) MJA_STEP(10)
#undef MJA_STEP
                float mjf3[3], dinv3[3];          // M^-1 J^T of this lane's slot-A variable; 1 / diagonal (0: row absent)
#pragma unroll
                for (int ka = 0; ka < 3; ka++) {
                    const int kr = 3 * tq + ka;
                    const v3 d = ka == 0 ? n : (ka == 1 ? t1 : t2);
                    const bool present = !tors || (ka == 0 ? spin > 0 : roll > 0);
                    const float ja = ja3[ka], jb = jb3[ka], mjb = mjb3[ka];
                    const float mja = l < NB ? mja3[ka] : dot(d, hA_);
                    mjf3[ka] = mja;
                    const float diag = group_sum(ja * mja + jb * mjb);
                    const float rel = group_sum(ja * ua + jb * ub);
                    // (the normal row in the canonical layout: the block repack below and the coop warm start read it back)
                    if (kr == 0) D.grows[((size_t)env * GP_RECS + GP_C + slot) * 16 + l] = make_float4(ja, mja, jb, mjb);
                    if (kr == 0 && !coop) { wsA = fmaf(mja, lam0, wsA); wsB = fmaf(mjb, lam0, wsB); }      // (coop: replayed in list order below)
                    float rhsn;
                    if (kr == 0) {
                        float r = 0;
                        if (fabsf(rel) >= P.rest_thresh) { r = rest * -rel; if (r < 0) r = 0; }
                        float verr = r - rel, perr = 0;
                        if (dist > 0) verr -= dist * inv_dt;
                        else perr = -dist * P.erp * inv_dt;
                        rhsn = perr + verr;
                    } else rhsn = -rel;
                    const float dinv = (present && diag > 0) ? 1.0f / diag : 0.0f;
                    dinv3[ka] = dinv;
                    if (l == 0)
                        *(float4 *)&LD(L_GSC + 4 * (r0 + kr)) = make_float4(rhsn * dinv, dinv, kr == 0 ? 0.0f : (kr < 3 ? mu : (kr == 3 ? spin : roll)), kr == 0 ? lam0 : 0.0f);
                }
                // ---- the rows as the sweeps stream them (block Gauss-Seidel, see the sweep): lane (h, g) = (bit 3, bit 2) of the group
                const bool hh_ = (l & 8) != 0, gg_ = (l & 4) != 0;
                if (!tors) {
                    // lateral friction pair (t1, t2): J pair in the lane's order (its own row first), M^-1 J^T pair, cross term of row t2
                    const float c21 = group_sum(ja3[2] * mjf3[1] + jb3[2] * mjb3[1]);
                    float4 *fp = D.grows + ((size_t)env * GP_RECS + GP_F + 3 * slot) * 16 + l;
                    fp[0] = make_float4(hh_ ? ja3[2] : ja3[1], hh_ ? ja3[1] : ja3[2], hh_ ? jb3[2] : jb3[1], hh_ ? jb3[1] : jb3[2]);
                    fp[16] = make_float4(mjf3[1], mjb3[1], mjf3[2], mjb3[2]);
                    fp[32] = make_float4(hh_ ? -(dinv3[2] * c21) : 0.0f, 0.0f, 0.0f, 0.0f);
                } else {
                    // torsional triple (spinning about n, rolling about t1, t2) + an absent fourth row; absent rows are all zero
                    const bool ps = spin > 0, pr = roll > 0;
                    const float a0 = ps ? ja3[0] : 0.0f, b0 = ps ? jb3[0] : 0.0f, ma0 = ps ? mjf3[0] : 0.0f, mb0 = ps ? mjb3[0] : 0.0f;
                    const float a1 = pr ? ja3[1] : 0.0f, b1 = pr ? jb3[1] : 0.0f, ma1 = pr ? mjf3[1] : 0.0f, mb1 = pr ? mjb3[1] : 0.0f;
                    const float a2 = pr ? ja3[2] : 0.0f, b2 = pr ? jb3[2] : 0.0f, ma2 = pr ? mjf3[2] : 0.0f, mb2 = pr ? mjb3[2] : 0.0f;
                    const float c10 = group_sum(a1 * ma0 + b1 * mb0), c20 = group_sum(a2 * ma0 + b2 * mb0), c21 = group_sum(a2 * ma1 + b2 * mb1);
                    const float4 pa = make_float4(hh_ ? a1 : a0, hh_ ? a0 : a1, hh_ ? b1 : b0, hh_ ? b0 : b1);      // pair (row 0, row 1)
                    const float4 pb = make_float4(hh_ ? 0.0f : a2, hh_ ? a2 : 0.0f, hh_ ? 0.0f : b2, hh_ ? b2 : 0.0f);  // pair (row 2, -)
                    float4 *tp = D.grows + ((size_t)env * GP_RECS + GP_T + 5 * slot) * 16 + l;
                    tp[0] = make_float4(gg_ ? pb.x : pa.x, gg_ ? pb.y : pa.y, gg_ ? pb.z : pa.z, gg_ ? pb.w : pa.w);     // (component selects: a
                    tp[16] = make_float4(gg_ ? pa.x : pb.x, gg_ ? pa.y : pb.y, gg_ ? pa.z : pb.z, gg_ ? pa.w : pb.w);    //  select of structs goes through scratch)
                    tp[32] = make_float4(ma0, mb0, ma1, mb1);
                    tp[48] = make_float4(ma2, mb2, 0.0f, 0.0f);
                    // (the lane's row: 2 g + h -- cross terms with the rows in front of it, times -1/diag)
                    const float x0 = gg_ ? (hh_ ? 0.0f : -(dinv3[2] * c20)) : (hh_ ? -(dinv3[1] * c10) : 0.0f);
                    const float x1 = (gg_ && !hh_) ? -(dinv3[2] * c21) : 0.0f;
                    tp[64] = make_float4(x0, x1, 0.0f, 0.0f);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");        // the staging area is rewritten by the next batch
        __builtin_amdgcn_wave_barrier();
    }
    if (!GEN) {
        const unsigned mm = (unsigned)(__ballot(class_mismatch) >> (16 * (grp & 3))) & 0xffffu;
        if (mm != 0u && l == 0 && !dead) atomicOr(&D.errflags[env], 4u);      // (internal consistency: never seen)
    }
    if (GEN) {
        // ---- block repack of the generic NORMAL rows: contacts 4 b .. 4 b + 3 as one block {J pair, J pair, M^-1 J^T pair, M^-1 J^T
        // pair, cross terms} in the lanes' order (see the sweep), from the canonical rows the builders have just stored.  coop:
        // group cg takes the blocks cg, cg + 4, ..  Also: the scalars of the dummy contacts and of the last block's absent rows.
        const int ngw = max(max(__builtin_amdgcn_readlane(ng, 0), __builtin_amdgcn_readlane(ng, 16)),
                            max(__builtin_amdgcn_readlane(ng, 32), __builtin_amdgcn_readlane(ng, 48)));
        if (ngw > 0) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            {
                float4 *zp = (float4 *)&LD(fix + LF_ZPAD);
                zp[l] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (l < 8) zp[16 + l] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            const bool hh_ = (l & 8) != 0, gg_ = (l & 4) != 0;
            const int nbw = (ngw + 3) >> 2;
#pragma unroll 1
            for (int b = cg; b < nbw; b += coop ? 4 : 1) {
                const int c0 = 4 * b;
                const float4 *cp = D.grows + ((size_t)env * GP_RECS + GP_C) * 16 + l;
                // (unconditional loads from always-valid addresses, values selected: DESIGN.md 7)
                const float4 r0_ = sel4(c0 < ng, cp[16 * (c0 < ng ? c0 : 0)]), r1_ = sel4(c0 + 1 < ng, cp[16 * (c0 + 1 < ng ? c0 + 1 : 0)]),
                             r2_ = sel4(c0 + 2 < ng, cp[16 * (c0 + 2 < ng ? c0 + 2 : 0)]), r3_ = sel4(c0 + 3 < ng, cp[16 * (c0 + 3 < ng ? c0 + 3 : 0)]);
                const float d1 = c0 + 1 < ng ? LD(L_GSC + 24 * (c0 + 1 < ng ? c0 + 1 : 0) + 1) : 0.0f, d2 = c0 + 2 < ng ? LD(L_GSC + 24 * (c0 + 2 < ng ? c0 + 2 : 0) + 1) : 0.0f,
                            d3 = c0 + 3 < ng ? LD(L_GSC + 24 * (c0 + 3 < ng ? c0 + 3 : 0) + 1) : 0.0f;
                // cross terms J_k . M^-1 J_i^T (i < k): what row i's impulse change adds to row k's J . v
                const float c10 = group_sum(r1_.x * r0_.y + r1_.z * r0_.w);
                const float c20 = group_sum(r2_.x * r0_.y + r2_.z * r0_.w), c21 = group_sum(r2_.x * r1_.y + r2_.z * r1_.w);
                const float c30 = group_sum(r3_.x * r0_.y + r3_.z * r0_.w), c31 = group_sum(r3_.x * r1_.y + r3_.z * r1_.w), c32 = group_sum(r3_.x * r2_.y + r3_.z * r2_.w);
                const float4 pa = make_float4(hh_ ? r1_.x : r0_.x, hh_ ? r0_.x : r1_.x, hh_ ? r1_.z : r0_.z, hh_ ? r0_.z : r1_.z);
                const float4 pb = make_float4(hh_ ? r3_.x : r2_.x, hh_ ? r2_.x : r3_.x, hh_ ? r3_.z : r2_.z, hh_ ? r2_.z : r3_.z);
                const float dk = gg_ ? (hh_ ? d3 : d2) : d1;          // (the lane's row 2 g + h; row 0 has no cross term)
                const float x0 = gg_ ? (hh_ ? c30 : c20) : (hh_ ? c10 : 0.0f), x1 = gg_ ? (hh_ ? c31 : c21) : 0.0f, x2 = (gg_ && hh_) ? c32 : 0.0f;
                if (c0 < ng) {
                    float4 *np = D.grows + ((size_t)env * GP_RECS + GP_N + 5 * b) * 16 + l;
                    np[0] = make_float4(gg_ ? pb.x : pa.x, gg_ ? pb.y : pa.y, gg_ ? pb.z : pa.z, gg_ ? pb.w : pa.w);
                    np[16] = make_float4(gg_ ? pa.x : pb.x, gg_ ? pa.y : pb.y, gg_ ? pa.z : pb.z, gg_ ? pa.w : pb.w);
                    np[32] = make_float4(r0_.y, r0_.w, r1_.y, r1_.w);
                    np[48] = make_float4(r2_.y, r2_.w, r3_.y, r3_.w);
                    np[64] = make_float4(-(dk * x0), -(dk * x1), -(dk * x2), 0.0f);
                    if (l >= 1 && l <= 3 && c0 + l >= ng) *(float4 *)&LD(L_GSC + 24 * (c0 + l)) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                } else if (!coop) {
                    // (four envs to a wave: this env has fewer blocks than another one of its wave -- it sweeps all-zero blocks meanwhile)
                    float4 *np = D.grows + ((size_t)env * GP_RECS + GP_N + 5 * b) * 16 + l;
                    const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    np[0] = z4; np[16] = z4; np[32] = z4; np[48] = z4; np[64] = z4;
                    if (l < 4) *(float4 *)&LD(L_GSC + 24 * (c0 + l)) = z4;
                }
            }
        }
    }
    if (coop) {
        // the warm-start velocity change of the generic normal rows, in list order, from the stored rows: the very fma sequence
        // a single builder runs in line (each of the four builders holds only its own contacts' share)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        for (int j = 0; j < ng; j++) {
            const float4 rw = D.grows[((size_t)env * GP_RECS + GP_C + j) * 16 + l];
            const float l0 = LD(L_GSC + 24 * j + 3);
            wsA = fmaf(rw.y, l0, wsA); wsB = fmaf(rw.w, l0, wsB);
        }
        // groups 1..3 have done their part: from here on they run along without rows of their own and store nothing
        if (cg != 0) { ng = 0; own_os = 0; n_os = 0; wsA = 0.0f; wsB = 0.0f; dead = true; }
    }
    SPROF(1);
    // ---- motor + limit rows: lane j < 11 builds the rows of joint j
    if (l < NB) {
        float dinv = 1.0f / LD(L_MINV + l * NB + l);
        float vt = P.kp * (tgt_l - q_l) / dt + qds_l + P.kd * (0.0f - qds_l);
        LD(L_MOT + 3 * l) = (vt - qds_l) * dinv;
        LD(L_MOT + 3 * l + 1) = dinv;
        LD(L_MOT + 3 * l + 2) = 0.0f;
        float lo = B.limits[l][0], hi = B.limits[l][1];
#pragma unroll
        for (int side = 0; side < 2; side++) {
            float dist = side == 0 ? q_l - lo : hi - q_l;
            bool on = (lo < hi) && (dist < 0.5f);
            float sg = side == 0 ? 1.0f : -1.0f;
            float rel = sg * qds_l;
            float verr = -rel, perr = 0;
            if (dist > 0) verr -= dist / dt;
            else perr = -dist * P.erp / dt;
            LD(L_LIM + 2 * (2 * l + side)) = on ? (perr + verr) * dinv : -1e30f;   // -1e30: row absent
            LD(L_LIM + 2 * (2 * l + side) + 1) = 0.0f;
        }
    }
    SPROF(2);
    // compact list of the limit rows that exist (usually the two finger lower limits), in row order
    unsigned limmask = 0;
#pragma unroll
    for (int js = 0; js < 2 * NB; js++) if (LD(L_LIM + 2 * js) > -1e29f) limmask |= 1u << js;    // 22 independent LDS reads
    if (coop && cg != 0) limmask = 0;            // (the limit rows live in the LDS region group 0 sweeps)
    // ---- PGS.  Lane state: dq (slot A: lanes 0..10 joints, lanes 11..15 object 2 during generic sweeps), vb (slot B), and
    // (dv, dw) of object lane-11 on lanes 11..13
    float dq = wsA, vb = wsB;          // (warm start: the inherited impulses of the generic normal rows are already applied)
    v2f V01 = P2(0.0f, 0.0f), V23 = P2(0.0f, 0.0f), V45 = P2(0.0f, 0.0f);      // (dv.xy) (dv.z dw.x) (dw.yz), REG_ROW_STEP
    const float inv_mass = lo_ >= 0 ? 1.0f / B.obj_mass[lo_ >= 0 ? lo_ : 0] : 0.0f;
    const float max_imp = P.max_impulse;
    const float m_rhs = l < NB ? LD(L_MOT + 3 * lj) : 0.0f, m_dinv = l < NB ? LD(L_MOT + 3 * lj + 1) : 0.0f;
    float m_lam = 0.0f, m_c = m_rhs;        // m_c = lambda + rhs of this lane's motor row, kept up to date off the critical chain
#define LDB4(r, off) (*(const float4 *)&LD(L_OSL + (r) * 12 + (off)))
#define LDT4(r, off) (*(const float4 *)&LD(L_OST + (r) * 8 + (off)))
#define LDZ4(has, r, off) sel4((has), LDB4((r), (off)))
    int lim_j[KLIM]; float lim_sg[KLIM], lim_rhs[KLIM], lim_dinv[KLIM], lim_col[KLIM], lim_lam[KLIM];
    {
        unsigned rem = limmask;
#pragma unroll
        for (int k = 0; k < KLIM; k++) {
            const bool has = rem != 0;
            const int js = has ? __ffs(rem) - 1 : 0;
            rem &= rem - 1;
            lim_j[k] = js >> 1; lim_sg[k] = (js & 1) == 0 ? 1.0f : -1.0f;
            lim_rhs[k] = has ? LD(L_LIM + 2 * js) : 0.0f; lim_dinv[k] = has ? LD(L_MOT + 3 * (js >> 1) + 1) : 0.0f;
            lim_col[k] = (has && l < NB) ? LD(L_MINV + lj * NB + (js >> 1)) : 0.0f;
            lim_lam[k] = 0.0f;
        }
        limmask = rem;                                   // rows left for the LDS loop
    }
    unsigned os_cs = 0;                                  // contact indices of the register rows, one byte each
    float os_mu[KOS], os_ln[KOS], os_l1[KOS], os_l2[KOS], os_l0[KOS];      // (os_l0: the inherited normal impulses)
    float4 os_n0[KOS], os_n1[KOS], os_n2[KOS], os_a0[KOS], os_a1[KOS], os_a2[KOS], os_b0[KOS], os_b1[KOS], os_b2[KOS];
    {
        unsigned rem = own_os;
#pragma unroll
        for (int i = 0; i < KOS; i++) {
            const bool has = rem != 0;
            const int c = has ? __ffs(rem) - 1 : 0;
            rem &= rem - 1;
            os_cs |= (has ? (unsigned)c : 255u) << (8 * i);
            os_mu[i] = has ? LD(L_MU + c) : 0.0f; os_ln[i] = 0.0f; os_l1[i] = 0.0f; os_l2[i] = 0.0f;
            os_l0[i] = has ? LD(L_OST + (3 * c) * 8 + 5) : 0.0f;
            os_n0[i] = LDZ4(has, 3 * c, 0); os_n1[i] = LDZ4(has, 3 * c, 4); os_n2[i] = LDZ4(has, 3 * c, 8);
            os_a0[i] = LDZ4(has, 3 * c + 1, 0); os_a1[i] = LDZ4(has, 3 * c + 1, 4); os_a2[i] = LDZ4(has, 3 * c + 1, 8);
            os_b0[i] = LDZ4(has, 3 * c + 2, 0); os_b1[i] = LDZ4(has, 3 * c + 2, 4); os_b2[i] = LDZ4(has, 3 * c + 2, 8);
        }
    }
    // torsional rows of the same contacts: rotation about the axis of linear row k (its dir), {m.x, rhs, m.y, m.z} (m = M^-1 J^T), 1/diag, lambda
    float4 ot_m[KOS][3]; float ot_d[KOS][3], ot_l[KOS][3], os_sp[KOS], os_ro[KOS];
#pragma unroll
    for (int i = 0; i < KOS; i++) {
        const int c = (os_cs >> (8 * i)) & 255;
        const bool has = c != 255;
        const int cc = has ? c : 0;
        os_sp[i] = has ? LD(L_SPIN + cc) : 0.0f; os_ro[i] = has ? LD(L_ROLL + cc) : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            ot_m[i][k] = sel4(has, LDT4(3 * cc + k, 0));
            ot_d[i][k] = has ? LD(L_OST + (3 * cc + k) * 8 + 4) : 0.0f;
            ot_l[i][k] = 0.0f;
        }
    }
    // motors: every lane evaluates the step of "its" row from its own dq; row J's impulse change is lane J's value,
    // broadcast to the 16 lanes of the env with one DPP row_newbcast (branch-free clamp, same values as if/else)
#define MOTOR_STEP(J)                                                                     \
        {                                                                                 \
            const float s0_ = fmaf(-dq, m_dinv, m_c);          /* (lambda + rhs) - dinv * dq */ \
            const float sum_ = fminf(fmaxf(s0_, -max_imp), max_imp);                      \
            const float dl_ = sum_ - m_lam;                                               \
            m_lam = (l == (J)) ? sum_ : m_lam;                                            \
            m_c = (l == (J)) ? sum_ + m_rhs : m_c;                                        \
            dq += minv_l[J] * row_bcast<J>(dl_);                                          \
        }
#define SWEEP_MOTORS                                                                                  \
        MOTOR_STEP(0) MOTOR_STEP(1) MOTOR_STEP(2) MOTOR_STEP(3) MOTOR_STEP(4) MOTOR_STEP(5)           \
        MOTOR_STEP(6) MOTOR_STEP(7) MOTOR_STEP(8) MOTOR_STEP(9) MOTOR_STEP(10)
#define LIMIT_STEP(k)   /* joint limit held in registers; absent rows are all-zero: dl = 0 */        \
        {                                                                                             \
            const float dqj_ = group_sum(l == lim_j[k] ? dq : 0.0f);                                  \
            const float s0_ = fmaf(-(lim_sg[k] * dqj_), lim_dinv[k], lim_lam[k] + lim_rhs[k]);         \
            const float sum_ = fminf(fmaxf(s0_, 0.0f), 100.0f);                                       \
            const float dl_ = sum_ - lim_lam[k];                                                      \
            lim_lam[k] = sum_;                                                                        \
            dq += lim_col[k] * (lim_sg[k] * dl_);                                                     \
        }
#define OSN_STEP(i) REG_ROW_STEP(os_n0[i], os_n1[i], os_n2[i], os_ln[i], 0.0f, 1e10f);
#define OSF_STEP(i)                                                                                   \
        {                                                                                             \
            const float hi_ = os_mu[i] * os_ln[i];                                                    \
            REG_ROW_STEP(os_a0[i], os_a1[i], os_a2[i], os_l1[i], -hi_, hi_);                          \
            REG_ROW_STEP(os_b0[i], os_b1[i], os_b2[i], os_l2[i], -hi_, hi_);                          \
        }
#define REG_TORS_STEP(AX, i, k, HI)                                                                  \
        {                                                                                             \
            const float jv_ = fmaf((AX).x, DWX, fmaf((AX).y, DWY, (AX).z * DWZ));                     \
            const float s0_ = fmaf(-jv_, ot_d[i][k], ot_l[i][k] + ot_m[i][k].y);                      \
            const float sum_ = __builtin_amdgcn_fmed3f(s0_, -(HI), (HI));                             \
            const float dl_ = sum_ - ot_l[i][k];                                                      \
            ot_l[i][k] = sum_;                                                                        \
            DWX = fmaf(ot_m[i][k].x, dl_, DWX);                                                       \
            V45 = pk_fma(P2(ot_m[i][k].z, ot_m[i][k].w), P2(dl_, dl_), V45);                          \
        }
#define OST_STEP(i)                                                                                   \
        {                                                                                             \
            const float hs_ = os_sp[i] * os_ln[i], hr_ = os_ro[i] * os_ln[i];                         \
            REG_TORS_STEP(os_n0[i], i, 0, hs_) REG_TORS_STEP(os_a0[i], i, 1, hr_) REG_TORS_STEP(os_b0[i], i, 2, hr_) \
        }
    // joint limits beyond the register rows (GEN: all of them), existing rows only; the expressions of LIMIT_STEP
#define LIMIT_LOOP                                                                                    \
        _Pragma("unroll 1")                                                                           \
        for (unsigned rem = limmask; rem; rem &= rem - 1) {                                           \
            const int js = __ffs(rem) - 1;                                                            \
            const int j = js >> 1;                                                                    \
            const float lr = LD(L_LIM + 2 * js), ll = LD(L_LIM + 2 * js + 1), di = LD(L_MOT + 3 * j + 1); \
            const float col = l < NB ? LD(L_MINV + lj * NB + j) : 0.0f;                               \
            const float sg = (js & 1) == 0 ? 1.0f : -1.0f;                                            \
            const float dqj_ = group_sum(l == j ? dq : 0.0f);                                         \
            const float s0_ = fmaf(-(sg * dqj_), di, ll + lr);                                        \
            const float sum_ = fminf(fmaxf(s0_, 0.0f), 100.0f);                                       \
            const float dl_ = sum_ - ll;                                                              \
            LD(L_LIM + 2 * js + 1) = sum_;                                                            \
            dq += col * (sg * dl_);                                                                   \
        }
    SPROF(3);
    SBLK_MARK(sb_t1)
    static_assert(KLIM == 2 && KOS == 4, "the sweeps below are written out for KLIM = 2, KOS = 4");
    // When no env of this wave has a row outside the registers (no generic contact, no further limit rows), an iteration is one straight-line block: the robot chain (motors, limits) and the object chain
    // (normals, frictions, torsional) are independent and the scheduler overlaps them.
    const bool simple = __ballot(!(ng == 0 && limmask == 0)) == 0ull;
    // generic sweeps: trip counts and the objects to move between the object lanes and the slots, over the whole wave
    const int ng_max = max(max(__builtin_amdgcn_readlane(ng, 0), __builtin_amdgcn_readlane(ng, 16)),
                           max(__builtin_amdgcn_readlane(ng, 32), __builtin_amdgcn_readlane(ng, 48)));
    const unsigned gobj_w = (unsigned)__builtin_amdgcn_readlane((int)gobj, 0) | (unsigned)__builtin_amdgcn_readlane((int)gobj, 16) |
                            (unsigned)__builtin_amdgcn_readlane((int)gobj, 32) | (unsigned)__builtin_amdgcn_readlane((int)gobj, 48);
    // the generic rows written to global memory by the builder are read back by this same wave: its stores must have
    // completed (workgroup scope: the CU's vector L1 is write-through and shared by the workgroup -- no cache maintenance;
    // an agent-scope fence would write back the XCD's L2)
    if (ng_max > 0) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // ---- generic rows: BLOCK Gauss-Seidel.  Rows are swept in blocks -- the normal rows of four consecutive contacts, the two
    // lateral rows of a contact, its (up to) three torsional rows -- whose J . v are formed SIDE BY SIDE from the velocities in
    // front of the block; row k's value then takes the impulse changes of the block's rows i < k through the cross terms
    // J_k . M^-1 J_i^T (stored with the block, times -1/diag_k).  Algebraically the very Gauss-Seidel sequence (same order, same
    // clamps); at one wave per SIMD a kernel lasts as long as its instruction count, and a block of four costs 29 VALU
    // instructions instead of 4 x 14 (two rows: 18 instead of 28):
    //   * lane (h, g) = (bit 3, bit 2) of the group ends up with the sum of row 2 g + h (two-row block: row h).  The J pairs are
    //     STORED in the order each lane needs ({own row, partner's row} x {slot A, slot B}, builder / repack above), so that the
    //     products are two packed multiply-adds per pair and the 16-lane sums of all four rows take five DPP adds: across the
    //     halves (row_ror 8: my second value is my partner's first), across the quads (row_half_mirror), inside the quad;
    //   * the clamp of row k is evaluated by the lanes of row k (their scalars rhs, 1/diag, bound, lambda come from LDS), its impulse
    //     change goes to all lanes with one DPP row_newbcast and updates the running sums s (one FMA) and the velocities
    //     (dq, vb) (one packed FMA with the row's (M^-1 J^T)_A, (M^-1 J^T)_B pair);
    //   * a lane's s stops changing once its own row is done (its later cross terms are zero): `sum` after the last step is the
    //     row's new lambda in every lane of the row.
    // The blocks are streamed from global memory (L2 / L1 resident: the wave wrote them) through ONE register set: every part of
    // the next block is requested as soon as the registers it lands in are free (BLK4_STEP), its scalars from LDS behind the
    // step.  Measured (profiles/r05_*): two sets (a block further ahead) were slower everywhere -- the 24 registers they take
    // push the object lanes' rows into AGPR traffic (+17 % on a sweep without generic rows) -- and three sets spilled.  A normal
    // block takes ~210 shader cycles (52 per row; the row-by-row step of round 4: 82), a lateral pair ~140.  Dummy blocks (contact
    // MAXC: all-zero records and scalars; all-zero normal blocks behind an env's own) are what a group sweeps while another env of
    // its wave has blocks left.  Byte offsets are 32 bits (rr_create checks N).
    const unsigned gp_l = (unsigned)env * (unsigned)GP_ENV_BYTES + ((unsigned)l << 4);
    // (LDS is addressed through 32-bit address-space-3 pointers made from byte addresses: the generic-pointer form costs an add per access)
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) const v4f_ lds_cf4;
    typedef __attribute__((address_space(3))) float lds_f;
    typedef __attribute__((address_space(3))) const unsigned char lds_cu8;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)g_slds;
#define LDS_F4(BYTE) (*(lds_cf4 *)(size_t)(unsigned)(BYTE))
#define LDS_F(BYTE) (*(lds_f *)(size_t)(unsigned)(BYTE))
    const bool hh = (l & 8) != 0, gg = (l & 4) != 0;
    const int rho = (gg ? 2 : 0) + (hh ? 1 : 0);
    const unsigned gsc_b = lds0 + 4u * (unsigned)L_GSC;             // byte address of the generic row scalars in LDS
    const int nb_g = (ng + 3) >> 2;                                 // normal blocks of this group's env
    const int nb_max = (ng_max + 3) >> 2;
    const unsigned list_b = lds0 + 4u * (unsigned)(fix + LF_LISTA); // byte address of listA (listT: + 64)
    // object lane (11 + O) -> slots, and back; comps 0..5 = dv.xyz, dw.xyz
#define TO_SLOT(O, C, REG, SLOTREG, LANE) { const float t_ = row_bcast<NB + (O)>(REG); SLOTREG = (l == (LANE)) ? t_ : SLOTREG; }
#define FROM_SLOT(O, C, REG, SLOTREG, LANE) { const float t_ = row_bcast<LANE>(SLOTREG); REG = (l == NB + (O)) ? t_ : REG; }
#define OBJ_SLOTS(OP)                                                                                                  \
    if (gobj_w & 1u) { OP(0, 0, DVX, vb, 0) OP(0, 1, DVY, vb, 1) OP(0, 2, DVZ, vb, 2) OP(0, 3, DWX, vb, 3) OP(0, 4, DWY, vb, 4) OP(0, 5, DWZ, vb, 5) } \
    if (gobj_w & 2u) { OP(1, 0, DVX, vb, 6) OP(1, 1, DVY, vb, 7) OP(1, 2, DVZ, vb, 8) OP(1, 3, DWX, vb, 9) OP(1, 4, DWY, vb, 10) OP(1, 5, DWZ, vb, 11) } \
    if (gobj_w & 4u) { OP(2, 0, DVX, dq, 11) OP(2, 1, DVY, dq, 12) OP(2, 2, DVZ, dq, 13) OP(2, 3, DWX, dq, 14) OP(2, 4, DWY, dq, 15) OP(2, 5, DWZ, vb, 12) }
    // warm start: the object components of the generic rows' initial velocity change go from the slots to the object lanes,
    // then the register rows add theirs (b2.w of a normal row = its inherited impulse; absent rows are all-zero)
    if (ng_max > 0) { OBJ_SLOTS(FROM_SLOT) }
#pragma unroll
    for (int i = 0; i < (OW ? 0 : KOS); i++) {
        const float l0_ = os_l0[i], sm_ = l0_ * inv_mass;
        os_ln[i] = l0_;
        V01 = pk_fma(P2(os_n0[i].x, os_n0[i].y), P2(sm_, sm_), V01);
        V23 = pk_fma(P2(os_n1[i].z, os_n1[i].w), P2(sm_, l0_), V23);
        V45 = pk_fma(P2(os_n2[i].x, os_n2[i].y), P2(l0_, l0_), V45);
    }
    int nA = 0, nT = 0;                                   // entries of this env's lists of active contacts (lateral / torsional pass)
    if (ng_max > 0) {                                     // (both lists start as all dummies)
#pragma unroll
        for (int k = 0; k < 64; k += 16) { listA[k + l] = (unsigned char)MAXC; listT[k + l] = (unsigned char)MAXC; }
    }
    // the register set of the block in flight: records {j0, j1, m0, m1, x}, scalars {rhs, 1/diag, lambda, upper bound}, the LDS
    // byte address of the scalars (lambda goes back there)
#define BLK_DECL(R) float4 R##j0 = make_float4(0, 0, 0, 0), R##j1 = R##j0, R##m0 = R##j0, R##m1 = R##j0; float R##x0 = 0.0f, R##x1 = 0.0f, R##x2 = 0.0f, R##rhs = 0.0f, R##di = 0.0f, R##lam = 0.0f, R##hi = 0.0f; int R##a = 0;
    BLK_DECL(A)
    // where block I of the pass lives: N pass: block I itself (a group whose env has fewer blocks than its wave sweeps the all-zero
    // blocks the repack left behind its own); F / T pass: the contact in entry I of the pass' list (dummy MAXC: all-zero records)
#define BLK_ENTRY(I) ((int)*(lds_cu8 *)(size_t)(lst_b + (unsigned)(I)))
#define GP_PTR(OFF) ((const char *)D.grows + (unsigned)(OFF))
#define BLK4_FETCH(R, OFF) { const char *bp__ = GP_PTR(OFF); R##j0 = *(const float4 *)bp__; R##j1 = *(const float4 *)(bp__ + 256); R##m0 = *(const float4 *)(bp__ + 512);   \
                             R##m1 = *(const float4 *)(bp__ + 768); R##x0 = *(const float *)(bp__ + 1024); R##x1 = *(const float *)(bp__ + 1028); R##x2 = *(const float *)(bp__ + 1032); }
#define BLK2_FETCH(R, OFF) { const char *bp__ = GP_PTR(OFF); R##j0 = *(const float4 *)bp__; R##m0 = *(const float4 *)(bp__ + 256); R##x0 = *(const float *)(bp__ + 512); }
#define SC_SET(R, HIEXPR) { const v4f_ sc__ = LDS_F4(R##a); R##rhs = sc__.x; R##di = sc__.y; R##lam = sc__.w; R##hi = (HIEXPR); }
    // N pass: the lane's row is the normal row of contact 4 b + rho
    const unsigned gpn_l = gp_l + GP_N * 256, nsc_l = gsc_b + (unsigned)rho * 96u;
#define N_OFF(I) (gpn_l + (unsigned)(I) * 1280u)
    typedef __attribute__((address_space(3))) const v2f lds_cf2;
    // (the bound coefficient of a normal row is not read: a 16-byte read whose third word is dead makes the next writer of that register wait for the LDS)
#define N_SC(R, I) { R##a = (int)(nsc_l + (unsigned)(I) * 384u); const v2f rd__ = *(lds_cf2 *)(size_t)(unsigned)R##a; R##rhs = rd__.x; R##di = rd__.y; R##lam = LDS_F(R##a + 12); R##hi = 1e10f; }
    // F pass: the lateral rows 6 j + 1 + h of contact j; T pass: the torsional rows 6 j + 3 + rho (rho 3: absent -> dummy contact)
    const unsigned gpf_l = gp_l + GP_F * 256, fsc_l = gsc_b + (hh ? 32u : 16u), gpt_l = gp_l + GP_T * 256, tsc_l = gsc_b + (rho == 3 ? MAXC * 96u : 48u + 16u * (unsigned)rho);
#define F_OFF(J) (gpf_l + (unsigned)(J) * 768u)
#define F_SC(R, J) { const unsigned t__ = (unsigned)(J) * 96u; R##a = (int)(fsc_l + t__); const float ln__ = LDS_F(gsc_b + t__ + 12u); SC_SET(R, sc__.z * ln__) }
#define T_OFF(J) (gpt_l + (unsigned)(J) * 1280u)
#define T_SC(R, J) { const unsigned t__ = (unsigned)(J) * 96u; R##a = (int)(tsc_l + (rho == 3 ? 0u : t__)); const float ln__ = LDS_F(gsc_b + t__ + 12u); SC_SET(R, sc__.z * ln__) }
#define DPPF(CTRL, V) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(V), (CTRL), 0xf, 0xf, false))
#define BLK_ROW(R, LANE, X, M0, M1)                                                                                    \
                {                                                                                                      \
                    sum_ = __builtin_amdgcn_fmed3f(s_, lo_, hi_);                                                      \
                    const float b_ = row_bcast<LANE>(sum_ - lam_);                                                     \
                    s_ = fmaf((X), b_, s_);                                                                            \
                    V_ = pk_fma(P2((M0), (M1)), P2(b_, b_), V_);                                                       \
                }
#define BLK_LAST(R, LANE, M0, M1)                                                                                      \
                {                                                                                                      \
                    sum_ = __builtin_amdgcn_fmed3f(s_, lo_, hi_);                                                      \
                    const float b_ = row_bcast<LANE>(sum_ - lam_);                                                     \
                    V_ = pk_fma(P2((M0), (M1)), P2(b_, b_), V_);                                                       \
                }
    // one block of four rows (HI: upper bound of the lane's row; lower bound LO).  NEXT: byte offset of the block the register
    // set sweeps next: each part of it is requested as soon as the registers it lands in are free -- the J pairs right behind
    // the products, the first M^-1 J^T pair behind row 1, the rest behind the last row
#define SCHED_FENCE __builtin_amdgcn_sched_barrier(0);
#define BLK4_STEP(R, LO, HI, NEXT)                                                                                     \
                {                                                                                                      \
                    const char *bp__ = GP_PTR(NEXT);                                                                   \
                    v2f p01_ = P2(R##j0.x, R##j0.y) * P2(dq, dq);                                                      \
                    v2f p23_ = P2(R##j1.x, R##j1.y) * P2(dq, dq);                                                      \
                    p01_ = pk_fma(P2(R##j0.z, R##j0.w), P2(vb, vb), p01_);                                             \
                    p23_ = pk_fma(P2(R##j1.z, R##j1.w), P2(vb, vb), p23_);                                             \
                    SCHED_FENCE                                                                                        \
                    R##j0 = *(const float4 *)bp__; R##j1 = *(const float4 *)(bp__ + 256);                             \
                    SCHED_FENCE                                                                                        \
                    const float s0_ = p01_.x + dpp_ror<8>(p01_.y);                                                     \
                    const float s1_ = p23_.x + dpp_ror<8>(p23_.y);                                                     \
                    float u_ = s0_ + DPPF(0x141, s1_);                                                                 \
                    u_ += DPPF(0xB1, u_);                                                                              \
                    u_ += DPPF(0x4E, u_);                                                                              \
                    const float lam_ = R##lam, lo_ = (LO), hi_ = (HI);                                                 \
                    float s_ = fmaf(-u_, R##di, lam_ + R##rhs), sum_;                                                  \
                    v2f V_ = P2(dq, vb);                                                                               \
                    BLK_ROW(R, 0, R##x0, R##m0.x, R##m0.y)                                                             \
                    BLK_ROW(R, 8, R##x1, R##m0.z, R##m0.w)                                                             \
                    SCHED_FENCE                                                                                        \
                    R##m0 = *(const float4 *)(bp__ + 512);                                                             \
                    SCHED_FENCE                                                                                        \
                    BLK_ROW(R, 4, R##x2, R##m1.x, R##m1.y)                                                             \
                    BLK_LAST(R, 12, R##m1.z, R##m1.w)                                                                  \
                    dq = V_.x; vb = V_.y;                                                                              \
                    LDS_F(R##a + 12) = sum_;                                                                           \
                    R##m1 = *(const float4 *)(bp__ + 768); R##x0 = *(const float *)(bp__ + 1024); R##x1 = *(const float *)(bp__ + 1028); R##x2 = *(const float *)(bp__ + 1032); \
                }
    // one block of two rows (lanes 0..7: the first row, 8..15: the second)
#define BLK2_STEP(R, LO, HI, NEXT)                                                                                     \
                {                                                                                                      \
                    const char *bp__ = GP_PTR(NEXT);                                                                   \
                    v2f p_ = P2(R##j0.x, R##j0.y) * P2(dq, dq);                                                        \
                    p_ = pk_fma(P2(R##j0.z, R##j0.w), P2(vb, vb), p_);                                                 \
                    SCHED_FENCE                                                                                        \
                    R##j0 = *(const float4 *)bp__;                                                                     \
                    SCHED_FENCE                                                                                        \
                    float u_ = p_.x + dpp_ror<8>(p_.y);                                                                \
                    u_ += DPPF(0x141, u_);                                                                             \
                    u_ += DPPF(0xB1, u_);                                                                              \
                    u_ += DPPF(0x4E, u_);                                                                              \
                    const float lam_ = R##lam, lo_ = (LO), hi_ = (HI);                                                 \
                    float s_ = fmaf(-u_, R##di, lam_ + R##rhs), sum_;                                                  \
                    v2f V_ = P2(dq, vb);                                                                               \
                    BLK_ROW(R, 0, R##x0, R##m0.x, R##m0.y)                                                             \
                    BLK_LAST(R, 8, R##m0.z, R##m0.w)                                                                   \
                    dq = V_.x; vb = V_.y;                                                                              \
                    LDS_F(R##a + 12) = sum_;                                                                           \
                    R##m0 = *(const float4 *)(bp__ + 256); R##x0 = *(const float *)(bp__ + 512);                       \
                }
    if (GEN) *(volatile int *)&LD(fix + LF_ENVW) = env;
    // (coop: groups 1..3 of the wave have no rows of their own -- their lanes are switched off for the sweeps)
    if (!(coop && cg != 0))
    for (int it = 0; it < P.iters; it++) {
        // compiler barrier: the row data in LDS / global memory is loop invariant, but hoisting hundreds of such loads out
        // of the sweep loop exhausts the register file (everything that should live in registers is held explicitly)
        asm volatile("" ::: "memory");
        if (simple) {
            SWEEP_MOTORS
            LIMIT_STEP(0) LIMIT_STEP(1)
            if (!OW) {
                OSN_STEP(0) OSN_STEP(1) OSN_STEP(2) OSN_STEP(3)
                OSF_STEP(0) OSF_STEP(1) OSF_STEP(2) OSF_STEP(3)
                OST_STEP(0) OST_STEP(1) OST_STEP(2) OST_STEP(3)
            }
            continue;
        }
        SPROF(4);
        SWEEP_MOTORS
        LIMIT_STEP(0) LIMIT_STEP(1)
        LIMIT_LOOP
        SPROF(7);
        if (OW) continue;                         // (the object chains are the object wave's)
#pragma unroll 1
        for (int pass = 0; pass < 3; pass++) {    // all normals, then all lateral frictions, then all torsional frictions
            // ---- the generic sweep of this pass: its first blocks are requested before the object lanes' own work
            const int cnt = pass == 0 ? nb_g : (pass == 1 ? nA : nT);
#define WAVE_MAX4(V) (coop ? __builtin_amdgcn_readlane((V), 0) : max(max(__builtin_amdgcn_readlane((V), 0), __builtin_amdgcn_readlane((V), 16)), max(__builtin_amdgcn_readlane((V), 32), __builtin_amdgcn_readlane((V), 48))))
            const int cmax = pass == 0 ? nb_max : WAVE_MAX4(cnt);
            const unsigned lst_b = list_b + (pass == 2 ? 64u : 0u);
            int jn_ = MAXC;                                   // (F / T pass: the contact of the block after the next one)
            if (cmax > 0) {
                if (pass == 0) {
                    BLK4_FETCH(A, N_OFF(0))
                } else if (pass == 1) {
                    BLK2_FETCH(A, F_OFF(BLK_ENTRY(0)))
                } else {
                    BLK4_FETCH(A, T_OFF(BLK_ENTRY(0)))
                }
            }
            // ---- object-vs-static rows of this pass (object lanes, side by side; all register resident)
            if (pass == 0) { OSN_STEP(0) OSN_STEP(1) OSN_STEP(2) OSN_STEP(3) }
            else if (pass == 1) { OSF_STEP(0) OSF_STEP(1) OSF_STEP(2) OSF_STEP(3) }
            else { OST_STEP(0) OST_STEP(1) OST_STEP(2) OST_STEP(3) }
            SPROF(8);
            if (cmax == 0) continue;
            OBJ_SLOTS(TO_SLOT)
            SPROF(9);
            // (one block per trip: the records of block i + 1 are requested inside the step of block i, its scalars behind it)
            // (the scalars of the pass' first block: requested only now -- the object lanes' rows need the registers)
            if (pass == 0) { N_SC(A, 0) }
            else if (pass == 1) { const int ja_ = BLK_ENTRY(0); jn_ = BLK_ENTRY(1); F_SC(A, ja_) }
            else { const int ja_ = BLK_ENTRY(0); jn_ = BLK_ENTRY(1); T_SC(A, ja_) }
            if (pass == 0) {
#define N_ITER(R, R2, I)                                                                                               \
                {                                                                                                      \
                    BLK4_STEP(R, 0.0f, R##hi, N_OFF((I) + 1))                                                          \
                    N_SC(R2, (I) + 1)                                                                                  \
                }
                for (int i0 = 0; i0 < cmax; i0++) N_ITER(A, A, i0)
#undef N_ITER
            } else if (pass == 1) {
#define F_ITER(R, R2, I)                                                                                               \
                {                                                                                                      \
                    const int j1_ = jn_;                                                                               \
                    jn_ = BLK_ENTRY((I) + 2);                                                                          \
                    BLK2_STEP(R, -R##hi, R##hi, F_OFF(j1_))                                                            \
                    F_SC(R2, j1_)                                                                                      \
                }
                for (int i0 = 0; i0 < cmax; i0++) F_ITER(A, A, i0)
#undef F_ITER
            } else {
#define T_ITER(R, R2, I)                                                                                               \
                {                                                                                                      \
                    const int j1_ = jn_;                                                                               \
                    jn_ = BLK_ENTRY((I) + 2);                                                                          \
                    BLK4_STEP(R, -R##hi, R##hi, T_OFF(j1_))                                                            \
                    T_SC(R2, j1_)                                                                                      \
                }
                for (int i0 = 0; i0 < cmax; i0++) T_ITER(A, A, i0)
#undef T_ITER
            }
            if (pass == 0) SPROF(10); else if (pass == 1) SPROF(5); else SPROF(6);      // (generic blocks: normal / lateral / torsional pass)
            OBJ_SLOTS(FROM_SLOT)
            if (pass == 0) {
                // ---- the contacts whose lateral and torsional rows can move something: those with a normal impulse, or with a
                //      lateral / torsional impulse left from the previous sweep.  The lists keep their dummy padding.
                const int nA_old = nA, nT_old = nT;
                nA = 0; nT = 0;
                for (int j0 = 0; j0 < ng_max; j0 += 16) {
                    const int j = j0 + l;
                    bool act = false, ht = false;
                    if (j < ng) {
                        // (all eight LDS reads issued together: with `||` they would form a chain of dependent round trips)
                        const int r = L_GSC + 24 * j;
                        const float l0 = LD(r + 3), l1 = LD(r + 7), l2 = LD(r + 11), l3 = LD(r + 15), l4 = LD(r + 19), l5 = LD(r + 23);
                        const float cs_ = LD(r + 14), cr_ = LD(r + 18);      // coefficients of the spinning / rolling rows
                        act = (l0 > 0.0f) | (l1 != 0.0f) | (l2 != 0.0f) | (l3 != 0.0f) | (l4 != 0.0f) | (l5 != 0.0f);
                        ht = act & ((cs_ > 0.0f) | (cr_ > 0.0f));
                    }
                    const unsigned lt = (1u << l) - 1u;
                    const unsigned ma = (unsigned)(__ballot(act) >> (16 * (grp & 3))) & 0xffffu, mt = (unsigned)(__ballot(ht) >> (16 * (grp & 3))) & 0xffffu;
                    if (act) listA[nA + __popc(ma & lt)] = (unsigned char)j;
                    if (ht) listT[nT + __popc(mt & lt)] = (unsigned char)j;
                    nA += __popc(ma); nT += __popc(mt);
                }
                // (entries a shrinking list leaves behind become dummies again: sixteen behind the new end at once, more only
                // when a list lost more than that in one sweep)
                listA[nA + l] = (unsigned char)MAXC; listT[nT + l] = (unsigned char)MAXC;
                if (__ballot((nA_old - nA > 16) | (nT_old - nT > 16)) != 0ull) {
                    for (int k = 16 + l; nA + k < nA_old; k += 16) listA[nA + k] = (unsigned char)MAXC;
                    for (int k = 16 + l; nT + k < nT_old; k += 16) listT[nT + k] = (unsigned char)MAXC;
                }
            }
            SPROF(11);
            SPROF(3);           // (calibration: two stamps back to back -- the cost of a stamp, once per pass)
        }
    }
    SPROF(4);
    // (GEN: the env index is parked in LDS over the sweeps -- the kernel with the generic rows has no register to spare there)
    if (GEN) env = *(volatile int *)&LD(fix + LF_ENVW);
    SBLK_MARK(sb_t2)
    SBLK_END(nc, ng, n_os, nA);
    SPROF_FLUSH;
    // normal impulses of the register-resident contact rows go back to their LDS slots (contact forces below)
#pragma unroll
    for (int i = 0; i < (OW ? 0 : KOS); i++) {
        const int c = (os_cs >> (8 * i)) & 255;
        if (c != 255) LD(L_OSL + (3 * c) * 12 + 11) = os_ln[i];
    }
    // (the light solve of a step with camera also sets up the render instances of its envs -- below; an env that does not
    // step is drawn as it stands)
    const bool setup = !GEN && RMp != nullptr && mine && env_raw < N;
    if (dead && !setup) return;
    // (the lane predicates of the tail are taken from a copy of the lane index the compiler cannot see through: lane masks of
    // the stage-in kept alive over the sweeps cost scalar registers the solver does not have)
    int lt = l;
    asm volatile("" : "+v"(lt));
    float q_fin = q_l;                 // joint angle / object pose after this step: what the render instances are set up from
    float o_fin[7] = {0, 0, 0, 0, 0, 0, 1};
    if (dead) {
        if (!OW && lt >= NB && lt - NB < P.nobj) {
#pragma unroll
            for (int k = 0; k < 3; k++) o_fin[k] = STT(ST_OPOS + 3 * (lt - NB) + k);
#pragma unroll
            for (int k = 0; k < 4; k++) o_fin[3 + k] = STT(ST_OQUAT + 4 * (lt - NB) + k);
        }
    } else {
    // ---- integrate: lanes 0..10 joints, lanes 11..13 objects
    bool finite = true;
    if (lt < NB) {
        // (GEN: the kernel with the generic rows reads the two values again instead of holding them over the sweeps, like the objects' below)
        const float qds_t = GEN ? SCR(S_QDS + (lt < NB ? lt : 0)) : qds_l, q_t = GEN ? STT(ST_Q + (lt < NB ? lt : 0)) : q_l;
        float v = qds_t + dq;
        float qn = fmaf(dt, v, q_t);            // (the integration is written out in fused operations: both forms of the kernel must round alike)
        finite = isfinite(qn);
        q_fin = qn;
        STT(ST_QD + lt) = v;
        STT(ST_Q + lt) = qn;
        if (lt < 7) D.joints[(size_t)env * 9 + lt] = qn;               // robot.py:203-211
        else if (lt == 7) D.joints[(size_t)env * 9 + 7] = qn;
        else if (lt == 8) D.joints[(size_t)env * 9 + 8] = -qn;
    }
    if (!OW && lt >= NB && lt - NB < P.nobj) {
        // (position, orientation and unconstrained velocities are the registers of the stage-in: no load at the tail of the chain)
        const int i = lt - NB;
        float v[3], w[3], pn[3];
        const float dvi[3] = {DVX, DVY, DVZ}, dwi[3] = {DWX, DWY, DWZ};
        // (GEN: the solver with the generic rows has no 13 registers to spare over the sweeps -- it reads them again)
        float vs3[3] = {myobj.vs.x, myobj.vs.y, myobj.vs.z}, ws3[3] = {myobj.ws.x, myobj.ws.y, myobj.ws.z};
        float op3[3] = {myobj.op.x, myobj.op.y, myobj.op.z};
        if (GEN) {
#pragma unroll
            for (int k = 0; k < 3; k++) { vs3[k] = SCR(S_OVS + 3 * i + k); ws3[k] = SCR(S_OWS + 3 * i + k); op3[k] = SCR(S_OP + 3 * i + k); }
#pragma unroll
            for (int k = 0; k < 4; k++) oquat[k] = STT(ST_OQUAT + 4 * i + k);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
            v[k] = vs3[k] + dvi[k];
            w[k] = ws3[k] + dwi[k];
            pn[k] = fmaf(dt, v[k], op3[k]);
            finite = finite && isfinite(pn[k]);
            STT(ST_OVEL + 3 * i + k) = v[k];
            STT(ST_OANG + 3 * i + k) = w[k];
            STT(ST_OPOS + 3 * i + k) = pn[k];
        }
        float wn = sqrtf(fmaf(w[0], w[0], fmaf(w[1], w[1], w[2] * w[2])));
        float ang = wn * dt, d0, d1, d2, d3;
        if (ang > 1e-12f) {
            float sn, cs;
            sincosf(ang * 0.5f, &sn, &cs);
            float s = sn / wn;
            d0 = w[0] * s; d1 = w[1] * s; d2 = w[2] * s; d3 = cs;
        } else {
            d0 = w[0] * dt * 0.5f; d1 = w[1] * dt * 0.5f; d2 = w[2] * dt * 0.5f; d3 = 1.0f;
        }
        float q0 = oquat[0], q1 = oquat[1], q2 = oquat[2], q3 = oquat[3];
        float r0 = fmaf(d3, q0, fmaf(d0, q3, fmaf(d1, q2, -(d2 * q1))));
        float r1 = fmaf(d3, q1, fmaf(-d0, q2, fmaf(d1, q3, d2 * q0)));
        float r2 = fmaf(d3, q2, fmaf(d0, q1, fmaf(-d1, q0, d2 * q3)));
        float r3 = fmaf(d3, q3, fmaf(-d0, q0, fmaf(-d1, q1, -(d2 * q2))));
        float inv = 1.0f / sqrtf(fmaf(r0, r0, fmaf(r1, r1, fmaf(r2, r2, r3 * r3))));
        STT(ST_OQUAT + 4 * i) = r0 * inv; STT(ST_OQUAT + 4 * i + 1) = r1 * inv;
        STT(ST_OQUAT + 4 * i + 2) = r2 * inv; STT(ST_OQUAT + 4 * i + 3) = r3 * inv;
        float *op = D.objpose + ((size_t)env * P.nobj + i) * 7;
        for (int k = 0; k < 3; k++) op[k] = pn[k];
        op[3] = r0 * inv; op[4] = r1 * inv; op[5] = r2 * inv; op[6] = r3 * inv;
        o_fin[0] = pn[0]; o_fin[1] = pn[1]; o_fin[2] = pn[2];
        o_fin[3] = r0 * inv; o_fin[4] = r1 * inv; o_fin[5] = r2 * inv; o_fin[6] = r3 * inv;
    }
    SPROF(5);
    if (!finite) atomicOr(&D.errflags[env], 1u);
    // ---- touch sensors (robot.py:152-163) + contact forces: lane c takes contacts c, c + 16, c + 32; the four maxima go round
    // the group (a maximum does not depend on the order it is taken in)
    if (OW) {
        // (a light env has no contact of the robot: the touch sensors read zero; the contact forces are the object wave's)
        if (l < 4) D.touch[(size_t)env * 4 + l] = 0.0f;
        if (l == 0) D.timestep[env] += 1;
    } else {
        float touch[4] = {0, 0, 0, 0};
        for (int c0 = 0; c0 < nc; c0 += 16) {
            const int c = c0 + l;
            if (c < nc) {
                const int meta = *(const int *)&LD(L_META + c);
                const int bodyA = meta_bodyA(meta), link = meta_link(meta);
                const float lam = meta_fast(meta) ? LD(L_OSL + (3 * meta_slot(meta)) * 12 + 11) : LD(L_GSC + 24 * meta_slot(meta) + 3);
                const float f = lam / dt;
                D.cforce[(size_t)env * MAXC + c] = f;
                if (bodyA >= 0 && bodyA < 16 && meta_near(meta)) {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (link == B.touch_links[k]) touch[k] = fmaxf(touch[k], f);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float m = touch[k];
            m = fmaxf(m, dpp_ror<8>(m)); m = fmaxf(m, dpp_ror<4>(m)); m = fmaxf(m, dpp_ror<2>(m)); m = fmaxf(m, dpp_ror<1>(m));
            touch[k] = m;
        }
        if (l == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) D.touch[(size_t)env * 4 + k] = touch[k];
            D.timestep[env] += 1;
        }
    }
    }       // (!dead)
    SPROF(6);
    // ---- render set-up of this env's instances (k_render_setup's work, from the values just integrated: the joint angles
    // go round the group by DPP, the object poses through LDS; lane l takes instances l and 16 + l)
    if (!GEN) {
        if (!setup) return;
        const RenderModel &RM = *RMp;
        if (!OW && lt >= NB && lt < NB + NOBJ) {
#pragma unroll
            for (int k = 0; k < 7; k++) LD(L_OBJ + 20 * (lt - NB) + k) = o_fin[k];
        }
        float qj[NB];
        static_assert(NB == 11, "joint angles round the group");
        qj[0] = row_bcast<0>(q_fin); qj[1] = row_bcast<1>(q_fin); qj[2] = row_bcast<2>(q_fin); qj[3] = row_bcast<3>(q_fin);
        qj[4] = row_bcast<4>(q_fin); qj[5] = row_bcast<5>(q_fin); qj[6] = row_bcast<6>(q_fin); qj[7] = row_bcast<7>(q_fin);
        qj[8] = row_bcast<8>(q_fin); qj[9] = row_bcast<9>(q_fin); qj[10] = row_bcast<10>(q_fin);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int i = lt; i < RM.ni; i += 16) {
            const int ot = RM.in_otype[i], oi = RM.in_oidx[i];
            if (OW && ot == 2) continue;          // (an object's instances are set up by the object wave, from what it integrated)
            float op7[7];
            const int ob = ot == 2 ? min(max(oi, 0), NOBJ - 1) : 0;
#pragma unroll
            for (int k = 0; k < 7; k++) op7[k] = LD(L_OBJ + 20 * ob + k);
            instance_setup_core(B, RM, D, env, i, ot, oi, qj, op7);
        }
    }
}
// Wave 4 of a k_solve_light workgroup (solve_body<false, true>): the object chains of the workgroup's sixteen envs, lane 3 e + o =
// object o of env e.  Everything it needs was staged in LDS by the env's group before the workgroup's barrier: the objects'
// data (L_OBJ), the rows of the contacts (L_OSL / L_OST, coefficients), flags, contact masks and orientations (OW_*).  Same
// row steps, same order within a chain, same integration as the object lanes of the one-kernel form: same bits.
__device__ __forceinline__ void light_object_wave(const BodyParams &B, const SimParams &P, const DevPtrs &D, const RenderModel *RMp) {
    const int N = P.N;
    const int t = threadIdx.x & 63;
    const int e = min(t / 3, 15), o = t - 3 * (t / 3);
    const int env_raw = 16 * (int)blockIdx.x + e;
    const bool lane_on = t < 48 && o < P.nobj && env_raw < N;
    const int env = env_raw < N ? env_raw : N - 1;
    const int fix = e * LF_TOTAL;
    const int L_MU = fix + LF_MU, L_SPIN = fix + LF_SPIN, L_ROLL = fix + LF_ROLL, L_OSL = fix + LF_OSL, L_OST = fix + LF_OST,
              L_GSC = fix + LF_GSC, L_OBJ = fix + LF_OBJ;
    float *state = D.state;
    const float dt = P.dt;
#ifdef RR_RASTER_STATS
    unsigned long long ow_t0 = __builtin_amdgcn_s_memrealtime();      // (100 MHz: the shader clock counter runs with the DVFS state)
#define OWPROF(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); if (t == 0 && (!(P.ablate & 0x4000) || (int)blockIdx.x == (P.ablate >> 16))) atomicAdd(&g_sprof[i], now_ - ow_t0); ow_t0 = now_; } while (0)
#else
#define OWPROF(i)
#endif
    // (while the env groups stage in and build rows: which render instance is this object's -- uniform loads, one round trip)
    int my_inst = -1, my_inst2 = -1;
    if (RMp) {
        const RenderModel &RM = *RMp;
        const int ni = RM.ni;
#pragma unroll
        for (int i = 0; i < MAXINST; i++) {
            const bool m_ = i < ni && RM.in_otype[i] == 2 && RM.in_oidx[i] == o;
            my_inst2 = (m_ && my_inst >= 0 && my_inst2 < 0) ? i : my_inst2;
            my_inst = (m_ && my_inst < 0) ? i : my_inst;
        }
    }
#ifdef RR_RASTER_STATS
    if (t == 0 && blockIdx.x < 4096) g_sblk[(2048 + blockIdx.x) * 8] = (unsigned)ow_t0;       // (scratch/ow_blocks.py: when did this workgroup start / end?)
#endif
    __syncthreads();
    OWPROF(8);           // waiting for the env groups (stage-in, command part, row build)
    const int flags = *(const int *)&LD(L_GSC + OW_FLAGS);
    if (!lane_on || !(flags & 1)) return;          // (per lane: nothing below crosses lanes)
    const bool dead = (flags & 2) != 0;
    const unsigned own_os = dead ? 0u : *(const unsigned *)&LD(L_GSC + OW_MASK + o);
    const float inv_mass = LD(L_OBJ + 20 * o + 3);
    float o_fin[7] = {0, 0, 0, 0, 0, 0, 1};
    if (dead) {
        if (!RMp) return;
#pragma unroll
        for (int k = 0; k < 3; k++) o_fin[k] = STT(ST_OPOS + 3 * o + k);
#pragma unroll
        for (int k = 0; k < 4; k++) o_fin[3 + k] = STT(ST_OQUAT + 4 * o + k);
    } else {
    // ---- this object's contacts: all six rows of each in registers (as the object lanes of solve_body hold them)
    unsigned os_cs = 0;
    float os_mu[KOS], os_ln[KOS], os_l1[KOS], os_l2[KOS], os_l0[KOS];
    float4 os_n0[KOS], os_n1[KOS], os_n2[KOS], os_a0[KOS], os_a1[KOS], os_a2[KOS], os_b0[KOS], os_b1[KOS], os_b2[KOS];
    {
        unsigned rem = own_os;
#pragma unroll
        for (int i = 0; i < KOS; i++) {
            const bool has = rem != 0;
            const int c = has ? __ffs(rem) - 1 : 0;
            rem &= rem - 1;
            os_cs |= (has ? (unsigned)c : 255u) << (8 * i);
            os_mu[i] = has ? LD(L_MU + c) : 0.0f; os_ln[i] = 0.0f; os_l1[i] = 0.0f; os_l2[i] = 0.0f;
            os_l0[i] = has ? LD(L_OST + (3 * c) * 8 + 5) : 0.0f;
            os_n0[i] = LDZ4(has, 3 * c, 0); os_n1[i] = LDZ4(has, 3 * c, 4); os_n2[i] = LDZ4(has, 3 * c, 8);
            os_a0[i] = LDZ4(has, 3 * c + 1, 0); os_a1[i] = LDZ4(has, 3 * c + 1, 4); os_a2[i] = LDZ4(has, 3 * c + 1, 8);
            os_b0[i] = LDZ4(has, 3 * c + 2, 0); os_b1[i] = LDZ4(has, 3 * c + 2, 4); os_b2[i] = LDZ4(has, 3 * c + 2, 8);
        }
    }
    float4 ot_m[KOS][3]; float ot_d[KOS][3], ot_l[KOS][3], os_sp[KOS], os_ro[KOS];
#pragma unroll
    for (int i = 0; i < KOS; i++) {
        const int c = (os_cs >> (8 * i)) & 255;
        const bool has = c != 255;
        const int cc = has ? c : 0;
        os_sp[i] = has ? LD(L_SPIN + cc) : 0.0f; os_ro[i] = has ? LD(L_ROLL + cc) : 0.0f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            ot_m[i][k] = sel4(has, LDT4(3 * cc + k, 0));
            ot_d[i][k] = has ? LD(L_OST + (3 * cc + k) * 8 + 4) : 0.0f;
            ot_l[i][k] = 0.0f;
        }
    }
    OWPROF(9);           // rows from LDS to registers
    v2f V01 = P2(0.0f, 0.0f), V23 = P2(0.0f, 0.0f), V45 = P2(0.0f, 0.0f);
    // warm start: the inherited normal impulses
#pragma unroll
    for (int i = 0; i < KOS; i++) {
        const float l0_ = os_l0[i], sm_ = l0_ * inv_mass;
        os_ln[i] = l0_;
        V01 = pk_fma(P2(os_n0[i].x, os_n0[i].y), P2(sm_, sm_), V01);
        V23.x = fmaf(os_n0[i].z, sm_, V23.x); V23.y = fmaf(os_n1[i].w, l0_, V23.y);
        V45 = pk_fma(P2(os_n2[i].x, os_n2[i].y), P2(l0_, l0_), V45);
    }
    // (the middle pair's update as two scalar FMAs on dir.z of the first pair: the same two operations as the packed one, and the
    // twelve registers of the rows' second copy of dir.z go to row data the wave would otherwise spill -- it shares its SIMD)
#pragma push_macro("REG_ROW_STEP")
#undef REG_ROW_STEP
#define REG_ROW_STEP(b0, b1, b2, lam, lo, hi)                                                                     \
    do {                                                                                                          \
        v2f p_ = P2((b0).x, (b0).y) * V01;                                                                        \
        p_ = pk_fma(P2((b0).z, (b0).w), V23, p_);                                                                 \
        p_ = pk_fma(P2((b1).x, (b1).y), V45, p_);                                                                 \
        const float jv_ = p_.x + p_.y;                                                                            \
        const float s0_ = fmaf(-jv_, (b2).w, (lam) + (b2).z);                                                     \
        const float sum_ = __builtin_amdgcn_fmed3f(s0_, (lo), (hi));                                              \
        const float dl_ = sum_ - (lam);                                                                           \
        (lam) = sum_;                                                                                             \
        const float sm_ = dl_ * inv_mass;                                                                         \
        V01 = pk_fma(P2((b0).x, (b0).y), P2(sm_, sm_), V01);                                                      \
        V23.x = fmaf((b0).z, sm_, V23.x); V23.y = fmaf((b1).w, dl_, V23.y);                                       \
        V45 = pk_fma(P2((b2).x, (b2).y), P2(dl_, dl_), V45);                                                      \
    } while (0)
    for (int it = 0; it < P.iters; it++) {
        asm volatile("" ::: "memory");
        OSN_STEP(0) OSN_STEP(1) OSN_STEP(2) OSN_STEP(3)
        OSF_STEP(0) OSF_STEP(1) OSF_STEP(2) OSF_STEP(3)
        OST_STEP(0) OST_STEP(1) OST_STEP(2) OST_STEP(3)
    }
#pragma pop_macro("REG_ROW_STEP")
    OWPROF(10);          // sweeps
    // ---- contact forces of this object's contacts (robot.py:131-150 reads them)
#pragma unroll
    for (int i = 0; i < KOS; i++) {
        const int c = (os_cs >> (8 * i)) & 255;
        if (c != 255) D.cforce[(size_t)env * MAXC + c] = os_ln[i] / dt;
    }
    // ---- integrate (the object lanes' code of solve_body, same operations); the object's pose and unconstrained velocities are
    // fetched from LDS only now: thirteen registers the sweeps need for row data
    bool finite = true;
    {
        const int i = o;
        float op3[3], vs3[3], ws3[3], oquat[4];
        {
            asm volatile("" ::: "memory");
            const float4 *od = (const float4 *)&LD(L_OBJ + 20 * o);
            const float4 d0 = od[0], d3 = od[3], d4 = od[4];
            op3[0] = d0.x; op3[1] = d0.y; op3[2] = d0.z;
            vs3[0] = d3.y; vs3[1] = d3.z; vs3[2] = d3.w; ws3[0] = d4.x; ws3[1] = d4.y; ws3[2] = d4.z;
#pragma unroll
            for (int k = 0; k < 4; k++) oquat[k] = LD(L_GSC + OW_QUAT + 4 * o + k);
        }
        float v[3], w[3], pn[3];
        const float dvi[3] = {DVX, DVY, DVZ}, dwi[3] = {DWX, DWY, DWZ};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            v[k] = vs3[k] + dvi[k];
            w[k] = ws3[k] + dwi[k];
            pn[k] = fmaf(dt, v[k], op3[k]);
            finite = finite && isfinite(pn[k]);
            STT(ST_OVEL + 3 * i + k) = v[k];
            STT(ST_OANG + 3 * i + k) = w[k];
            STT(ST_OPOS + 3 * i + k) = pn[k];
        }
        float wn = sqrtf(fmaf(w[0], w[0], fmaf(w[1], w[1], w[2] * w[2])));
        float ang = wn * dt, d0, d1, d2, d3;
        if (ang > 1e-12f) {
            float sn, cs;
            sincosf(ang * 0.5f, &sn, &cs);
            float s_ = sn / wn;
            d0 = w[0] * s_; d1 = w[1] * s_; d2 = w[2] * s_; d3 = cs;
        } else {
            d0 = w[0] * dt * 0.5f; d1 = w[1] * dt * 0.5f; d2 = w[2] * dt * 0.5f; d3 = 1.0f;
        }
        float q0 = oquat[0], q1 = oquat[1], q2 = oquat[2], q3 = oquat[3];
        float r0 = fmaf(d3, q0, fmaf(d0, q3, fmaf(d1, q2, -(d2 * q1))));
        float r1 = fmaf(d3, q1, fmaf(-d0, q2, fmaf(d1, q3, d2 * q0)));
        float r2 = fmaf(d3, q2, fmaf(d0, q1, fmaf(-d1, q0, d2 * q3)));
        float r3 = fmaf(d3, q3, fmaf(-d0, q0, fmaf(-d1, q1, -(d2 * q2))));
        float inv = 1.0f / sqrtf(fmaf(r0, r0, fmaf(r1, r1, fmaf(r2, r2, r3 * r3))));
        STT(ST_OQUAT + 4 * i) = r0 * inv; STT(ST_OQUAT + 4 * i + 1) = r1 * inv;
        STT(ST_OQUAT + 4 * i + 2) = r2 * inv; STT(ST_OQUAT + 4 * i + 3) = r3 * inv;
        float *op = D.objpose + ((size_t)env * P.nobj + i) * 7;
        for (int k = 0; k < 3; k++) op[k] = pn[k];
        op[3] = r0 * inv; op[4] = r1 * inv; op[5] = r2 * inv; op[6] = r3 * inv;
        o_fin[0] = pn[0]; o_fin[1] = pn[1]; o_fin[2] = pn[2];
        o_fin[3] = r0 * inv; o_fin[4] = r1 * inv; o_fin[5] = r2 * inv; o_fin[6] = r3 * inv;
    }
    if (!finite) atomicOr(&D.errflags[env], 1u);
    OWPROF(11);          // forces, integration
    }       // (!dead)
    // ---- the render instances of this object
    if (RMp) {
        const RenderModel &RM = *RMp;
        const float qj[NB] = {0};
        if (my_inst >= 0) instance_setup_core(B, RM, D, env, my_inst, 2, o, qj, o_fin);
        if (my_inst2 >= 0) {          // (an object drawn as more than two instances: the rest in a loop)
            instance_setup_core(B, RM, D, env, my_inst2, 2, o, qj, o_fin);
#pragma unroll 1
            for (int i = my_inst2 + 1; i < RM.ni; i++)
                if (RM.in_otype[i] == 2 && RM.in_oidx[i] == o) instance_setup_core(B, RM, D, env, i, 2, o, qj, o_fin);
        }
    }
    OWPROF(12);          // render instances
#ifdef RR_RASTER_STATS
    if (t == 0 && blockIdx.x < 4096) g_sblk[(2048 + blockIdx.x) * 8 + 1] = (unsigned)__builtin_amdgcn_s_memrealtime();
#endif
}
__global__ void __launch_bounds__(256) k_solve(BodyParams B, SimParams P, DevPtrs D, int sel, int coop_launch) { solve_body<true>(B, P, D, sel, coop_launch); }
// the light envs of a split step (sel 1), 64-thread workgroups
// (RMp: the render model when the step draws -- the kernel then sets up the render instances of its envs; else nullptr)
__global__ void __launch_bounds__(64) k_solve_light(BodyParams B, SimParams P, DevPtrs D, const RenderModel *RMp) { solve_body<false>(B, P, D, 1, 0, RMp); }
// ... in workgroups for sixteen envs with one wave running the object chains (OW); the bookkeeping of solve_body's first lines
// is thread 0's whatever its wave's role
__global__ void __launch_bounds__(LIGHT_OW_THREADS) k_solve_light_ow(BodyParams B, SimParams P, DevPtrs D, const RenderModel *RMp) { solve_body<false, true>(B, P, D, 1, 0, RMp); }

// obs pack without stepping (after reset / set_state)
__global__ void k_obs(SimParams P, DevPtrs D) {
    const int N = P.N;
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    const float *state = D.state;
    for (int i = 0; i < 7; i++) D.joints[(size_t)env * 9 + i] = STT(ST_Q + i);
    D.joints[(size_t)env * 9 + 7] = STT(ST_Q + 7);
    D.joints[(size_t)env * 9 + 8] = -STT(ST_Q + 8);
    for (int i = 0; i < P.nobj; i++) {
        float *op = D.objpose + ((size_t)env * P.nobj + i) * 7;
        for (int k = 0; k < 3; k++) op[k] = STT(ST_OPOS + 3 * i + k);
        for (int k = 0; k < 4; k++) op[3 + k] = STT(ST_OQUAT + 4 * i + k);
    }
}

// Host mirror of the low-dimensional observations (rr_map_observations): device-mapped pinned host memory that a step's last
// launch on the main stream fills -- the single-env facade then needs ONE wait per step and no device-to-host copy calls.
struct ObsMirror { float *joints, *touch, *objpose; int *timestep; unsigned *errflags; };
__global__ void k_obs_mirror(SimParams P, DevPtrs D, ObsMirror M) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= P.N) return;
    for (int i = 0; i < 9; i++) M.joints[(size_t)env * 9 + i] = D.joints[(size_t)env * 9 + i];
    for (int i = 0; i < 4; i++) M.touch[(size_t)env * 4 + i] = D.touch[(size_t)env * 4 + i];
    for (int i = 0; i < P.nobj * 7; i++) M.objpose[(size_t)env * P.nobj * 7 + i] = D.objpose[(size_t)env * P.nobj * 7 + i];
    M.timestep[env] = D.timestep[env];
    M.errflags[env] = D.errflags[env];
}

// ---------------------------------------------------------------------------------------------- reset / state io
__global__ void k_reset(BodyParams B, SimParams P, DevPtrs D, const unsigned char *mask) {
    const int N = P.N;
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    if (mask && !mask[env]) return;
    float *state = D.state;
    for (int i = 0; i < ST_TOTAL; i++) STT(i) = 0;
    for (int i = 0; i < NOBJ; i++) {
        for (int k = 0; k < 3; k++) STT(ST_OPOS + 3 * i + k) = D.obj_home[(size_t)(7 * i + k) * N + env];
        for (int k = 0; k < 4; k++) STT(ST_OQUAT + 4 * i + k) = D.obj_home[(size_t)(7 * i + 3 + k) * N + env];
    }
    D.timestep[env] = 0;
    D.errflags[env] = 0;
    for (int k = 0; k < 4; k++) D.touch[(size_t)env * 4 + k] = 0;
    D.ccount[env] = 0;
    D.ccount_pub[env] = 0; D.class_pub[env] = 0;
}

__global__ void k_state_io(SimParams P, DevPtrs D, float *aos /*[N][61]*/, int to_aos) {
    const int N = P.N;
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    float *state = D.state;
    // AoS order (same as the oracle's 61-vector): q[11] qd[11] then per object pos3 quat4 lin3 ang3
    float *a = aos + (size_t)env * NSTATE;
    for (int i = 0; i < 22; i++) {
        if (to_aos) a[i] = STT(i); else STT(i) = a[i];
    }
    for (int o = 0; o < NOBJ; o++) {
        float *b = a + 22 + 13 * o;
        for (int k = 0; k < 3; k++) {
            if (to_aos) { b[k] = STT(ST_OPOS + 3 * o + k); b[7 + k] = STT(ST_OVEL + 3 * o + k); b[10 + k] = STT(ST_OANG + 3 * o + k); }
            else { STT(ST_OPOS + 3 * o + k) = b[k]; STT(ST_OVEL + 3 * o + k) = b[7 + k]; STT(ST_OANG + 3 * o + k) = b[10 + k]; }
        }
        for (int k = 0; k < 4; k++) {
            if (to_aos) b[3 + k] = STT(ST_OQUAT + 4 * o + k); else STT(ST_OQUAT + 4 * o + k) = b[3 + k];
        }
    }
    if (!to_aos) {
        // a restored state is a fresh start for everything derived from the previous step: the NaN guard's freeze bit, the
        // contact list behind rr_get_contacts and the touch sensors (as k_reset does)
        D.errflags[env] = 0;
        for (int k = 0; k < 4; k++) D.touch[(size_t)env * 4 + k] = 0;
        D.ccount[env] = 0;
        D.ccount_pub[env] = 0; D.class_pub[env] = 0;
    }
}

// rr_set_object_poses: poses [N][nobj][7] for the envs whose mask byte is set (nullptr: all); zeroes the velocities like
// BodyPart.reset_pose -> resetBasePositionAndOrientation (env.py:159-162)
__global__ void k_set_object_poses(SimParams P, DevPtrs D, const float *poses, const unsigned char *mask) {
    const int N = P.N;
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    if (mask && !mask[env]) return;
    float *state = D.state;
    for (int i = 0; i < P.nobj; i++) {
        const float *p7 = poses + ((size_t)env * P.nobj + i) * 7;
        for (int k = 0; k < 3; k++) { STT(ST_OPOS + 3 * i + k) = p7[k]; STT(ST_OVEL + 3 * i + k) = 0; STT(ST_OANG + 3 * i + k) = 0; }
        for (int k = 0; k < 4; k++) STT(ST_OQUAT + 4 * i + k) = p7[3 + k];
    }
}

// REALRobotEnv.evaluateGoal (env.py:181-200) for every env: sum over the goal's objects of exp(-(ln 4 / 0.10) |p_goal - p|)
__global__ void k_goal_score(SimParams P, DevPtrs D, const float *goal_pos /*[N][nobj][3]*/, const unsigned char *goal_mask /*[N][nobj] or nullptr*/, float *score) {
    const int N = P.N;
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    const float *state = D.state;
    const float pos_const = 13.862943611198906f;        // -log(0.25) / 0.10: the score falls to 0.25 within 10 cm
    float sc = 0.0f;
    for (int i = 0; i < P.nobj; i++) {
        if (goal_mask && !goal_mask[(size_t)env * P.nobj + i]) continue;
        const float *g = goal_pos + ((size_t)env * P.nobj + i) * 3;
        const float dx = g[0] - STT(ST_OPOS + 3 * i), dy = g[1] - STT(ST_OPOS + 3 * i + 1), dz = g[2] - STT(ST_OPOS + 3 * i + 2);
        sc += expf(-pos_const * sqrtf(dx * dx + dy * dy + dz * dz));
    }
    score[env] = sc;
}

// ---------------------------------------------------------------------------------------------- render setup
__global__ void __launch_bounds__(64) k_render_setup(BodyParams B, SimParams P, const RenderModel *RMp, DevPtrs D, int sel) {
    const RenderModel &RM = *RMp;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int env = gid >> 5, i = gid & (MAXINST - 1);
    static_assert(MAXINST == 32, "thread -> (env, instance) mapping");
    if (env >= P.N || i >= RM.ni || !env_selected(D.hgflag, env, sel)) return;
    instance_setup(B, P, RM, D, env, i);
}

// link poses (COM frame) for rr_link_poses
__global__ void __launch_bounds__(64) k_link_poses(BodyParams B, SimParams P, const RenderModel *RMp, DevPtrs D, float *out /*[N][nl][7]*/) {
    const RenderModel &RM = *RMp;
    const int N = P.N;
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    const float *state = D.state;
    float q[NB];
#pragma unroll
    for (int i = 0; i < NB; i++) q[i] = STT(ST_Q + i);
    m3 bR[NB]; v3 bp[NB], bax[NB];
    fk_all(B, q, bR, bp, bax);
    for (int l = 0; l < RM.nl; l++) {
        int b = RM.link_body[l];
        m3 lr;
        for (int k = 0; k < 9; k++) lr.m[k] = RM.link_rot[l][k];
        v3 lp = mk(RM.link_pos[l][0], RM.link_pos[l][1], RM.link_pos[l][2]);
        m3 R = lr;
        v3 p = mk(B.robot_pos[0], B.robot_pos[1], B.robot_pos[2]) + lp;
        if (b >= 0) {
            m3 Rb = bR[0]; v3 pb = bp[0];
#pragma unroll
            for (int bb = 0; bb < NB; bb++) if (bb == b) { Rb = bR[bb]; pb = bp[bb]; }
            R = mul(Rb, lr);
            p = mulv(Rb, lp) + pb;
        }
        float qq[4];
        m3_to_quat(R, qq);
        float *o = out + ((size_t)env * RM.nl + l) * 7;
        o[0] = p.x; o[1] = p.y; o[2] = p.z; o[3] = qq[0]; o[4] = qq[1]; o[5] = qq[2]; o[6] = qq[3];
    }
}

// ---------------------------------------------------------------------------------------------- inverse kinematics (K8)
// Batched damped-least-squares IK for link 7 (gripper `base`): the device restatement of what step_cartesian /
// generate_plan ask pybullet for (env.py:372-375, 422-427: calculateInverseKinematics(0, 7, pos, orn,
// maxNumIterations=1000, residualThreshold=0.001)); same algorithm as oracle/kinematics.py (the numpy checker used by
// tests): J^T (J J^T + 0.01 I)^-1 e steps clamped to 0.5 rad, wrapped to (-pi, pi], two or three seeds, branch choice
// by (converged, continuity with the previous way-point | elbow height).
struct IkModel {          // arm chain constants, passed by value
    float jpos[7][3], jrot[7][9], axis[7][3], robot_pos[3];
    float ee_pos[3], ee_rot[9];        // gripper base frame in body 6
};

__device__ void ik_fk(const IkModel &M, const float *q, v3 *p, v3 *ax, m3 &Re, v3 &pe, float &elbow_z) {
    m3 R = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
    v3 pp = mk(M.robot_pos[0], M.robot_pos[1], M.robot_pos[2]);
#pragma unroll
    for (int b = 0; b < 7; b++) {
        m3 jr;
#pragma unroll
        for (int k = 0; k < 9; k++) jr.m[k] = M.jrot[b][k];
        m3 Rj = mul(R, jr);
        v3 a = mk(M.axis[b][0], M.axis[b][1], M.axis[b][2]);
        pp = pp + mulv(R, mk(M.jpos[b][0], M.jpos[b][1], M.jpos[b][2]));
        p[b] = pp;
        ax[b] = mulv(Rj, a);
        R = mul(Rj, axis_angle(a, q[b]));
        if (b == 3) elbow_z = pp.z;
    }
    m3 er;
#pragma unroll
    for (int k = 0; k < 9; k++) er.m[k] = M.ee_rot[k];
    Re = mul(R, er);
    pe = pp + mulv(R, mk(M.ee_pos[0], M.ee_pos[1], M.ee_pos[2]));
}

// one DLS run from q (in/out); returns the final residual norm
__device__ float ik_dls(const IkModel &M, float *q, v3 tp, const m3 &Rt, int max_iters) {
    float err = 1e30f;
    for (int it = 0; it < max_iters; it++) {
        v3 p[7], ax[7], pe; m3 Re; float ez;
        ik_fk(M, q, p, ax, Re, pe, ez);
        v3 dp = tp - pe;
        m3 Rerr = mul(Rt, transpose(Re));
        v3 w = mk(0.5f * (Rerr.m[7] - Rerr.m[5]), 0.5f * (Rerr.m[2] - Rerr.m[6]), 0.5f * (Rerr.m[3] - Rerr.m[1]));
        float e[6] = {dp.x, dp.y, dp.z, w.x, w.y, w.z};
        err = sqrtf(dot(dp, dp) + dot(w, w));
        if (err < 1e-3f) break;
        float J[6][7];
#pragma unroll
        for (int k = 0; k < 7; k++) {
            v3 jl = cross(ax[k], pe - p[k]);
            J[0][k] = jl.x; J[1][k] = jl.y; J[2][k] = jl.z; J[3][k] = ax[k].x; J[4][k] = ax[k].y; J[5][k] = ax[k].z;
        }
        float A[6][6];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) {
                float s = (i == j) ? 0.01f : 0.0f;
#pragma unroll
                for (int k = 0; k < 7; k++) s += J[i][k] * J[j][k];
                A[i][j] = s;
            }
        // Cholesky solve A x = e
        float L[6][6], y[6], x[6];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j <= i; j++) {
                float s = A[i][j];
#pragma unroll
                for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
                if (i == j) L[i][i] = sqrtf(s);
                else L[i][j] = s / L[j][j];
            }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            float s = e[i];
#pragma unroll
            for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
            y[i] = s / L[i][i];
        }
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            float s = y[i];
#pragma unroll
            for (int k = i + 1; k < 6; k++) s -= L[k][i] * x[k];
            x[i] = s / L[i][i];
        }
        float dq[7], mx = 0;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            float s = 0;
#pragma unroll
            for (int i = 0; i < 6; i++) s += J[i][k] * x[i];
            dq[k] = s;
            mx = fmaxf(mx, fabsf(s));
        }
        float sc = mx > 0.5f ? 0.5f / mx : 1.0f;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            float v = q[k] + dq[k] * sc;
            v = v + 3.14159265358979f;
            v = v - 6.28318530717959f * floorf(v * 0.159154943091895f);
            q[k] = v - 3.14159265358979f;
        }
    }
    return err;
}

// best of {current joints, elbow-up seed, previous way-point}; out7 receives the solution
__device__ float ik_solve(const IkModel &M, const float *q_cur, const float *prefer /*nullable*/, v3 tp, const m3 &Rt, float *out7) {
    const float elbow_up[7] = {0.0f, 0.6f, 0.0f, -1.3f, 0.0f, 1.2f, 0.0f};
    bool have = false, best_conv = false;
    float best_key = 0, best_err = 0;
    const int nseeds = prefer ? 3 : 2;
    for (int sidx = 0; sidx < nseeds; sidx++) {
        float q[7];
#pragma unroll
        for (int k = 0; k < 7; k++) q[k] = sidx == 0 ? q_cur[k] : (sidx == 1 ? elbow_up[k] : prefer[k]);
        float err = ik_dls(M, q, tp, Rt, 1000);
        bool conv = err < 1e-2f;
        float key;
        if (prefer) {
            float mxd = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) mxd = fmaxf(mxd, fabsf(q[k] - prefer[k]));
            key = -mxd;
        } else {
            v3 p[7], ax[7], pe; m3 Re; float ez;
            ik_fk(M, q, p, ax, Re, pe, ez);
            key = ez;
        }
        bool better = !have || (conv && !best_conv) || (conv == best_conv && key > best_key);
        if (better) {
            have = true; best_conv = conv; best_key = key; best_err = err;
#pragma unroll
            for (int k = 0; k < 7; k++) out7[k] = q[k];
        }
    }
    return best_err;
}

__device__ __forceinline__ m3 quat_target(const float *qt) {
    float n = rsqrtf(qt[0] * qt[0] + qt[1] * qt[1] + qt[2] * qt[2] + qt[3] * qt[3]);
    return quat_to_m3(qt[0] * n, qt[1] * n, qt[2] * n, qt[3] * n);
}

// single IK per env from the current joints: targets [N][7] (pos, quat xyzw) -> out [N][11] (fingers keep their values)
__global__ void __launch_bounds__(64) k_ik(IkModel M, SimParams P, DevPtrs D, const float *targets, float *out_q, float *out_err) {
    const int N = P.N;
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    const float *state = D.state;
    float qc[7], o7[7];
#pragma unroll
    for (int k = 0; k < 7; k++) qc[k] = STT(ST_Q + k);
    const float *t = targets + (size_t)env * 7;
    m3 Rt = quat_target(t + 3);
    float err = ik_solve(M, qc, nullptr, mk(t[0], t[1], t[2]), Rt, o7);
    for (int k = 0; k < 7; k++) out_q[(size_t)env * 11 + k] = o7[k];
    for (int k = 7; k < 11; k++) out_q[(size_t)env * 11 + k] = STT(ST_Q + k);
    out_err[env] = err;
}

// 1000-step macro plan per env (env.py:388-454): rows 0-99 home2, 100-199 above p1 (z 0.6), 200-249 at p1 (z 0.46),
// 250-749 p1 -> p2 at z 0.46 in <= 5 cm IK segments, 750-799 above p2, 800-899 home2, 900-999 home.
#define PLAN_LEN 1000
__global__ void __launch_bounds__(64) k_plan_macro(IkModel M, SimParams P, DevPtrs D, const float *macro /*[N][4]*/,
                                                   const unsigned char *mask, float *plan /*[N][1000][9]*/, int *plan_step) {
    const int N = P.N;
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    if (mask && !mask[env]) return;
    const float *state = D.state;
    float qc[7];
#pragma unroll
    for (int k = 0; k < 7; k++) qc[k] = STT(ST_Q + k);
    const float f7 = STT(ST_Q + 7), f8 = STT(ST_Q + 8);       // env.py:427: first 9 of the 11 dofs
    const float p1x = macro[(size_t)env * 4], p1y = macro[(size_t)env * 4 + 1], p2x = macro[(size_t)env * 4 + 2], p2y = macro[(size_t)env * 4 + 3];
    // getQuaternionFromEuler([0, 3.14, -1.57])  (env.py:422)
    float cr = 1.0f, sr = 0.0f, sp, cp, sy, cy;
    sincosf(3.14f * 0.5f, &sp, &cp);
    sincosf(-1.57f * 0.5f, &sy, &cy);
    float qt[4] = {sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
    m3 Rt = quat_target(qt);
    float *pl = plan + (size_t)env * PLAN_LEN * 9;
    auto fill = [&](int r0, int r1, const float *q7, bool ik) {
        for (int r = r0; r < r1; r++) {
            for (int k = 0; k < 7; k++) pl[r * 9 + k] = q7[k];
            pl[r * 9 + 7] = ik ? f7 : q7[7];
            pl[r * 9 + 8] = ik ? f8 : q7[8];
        }
    };
    float home2[9] = {0, 0, 0, 0, 0, 1.57079632679f, 1.57079632679f, 0, 0}, home[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    fill(0, 100, home2, false);
    fill(800, 900, home2, false);
    fill(900, 1000, home, false);
    float last[7], cur[7];
    ik_solve(M, qc, nullptr, mk(p1x, p1y, 0.6f), Rt, last);
    fill(100, 200, last, true);
    ik_solve(M, qc, last, mk(p1x, p1y, 0.46f), Rt, cur);
    for (int k = 0; k < 7; k++) last[k] = cur[k];
    fill(200, 250, last, true);
    // interpolate3D(p1, p2, 500) (env.py:430-441)
    float dx = p2x - p1x, dy = p2y - p1y;
    float dist = sqrtf(dx * dx + dy * dy);
    int pieces = (int)(dist / 0.05f) + 1;
    if (pieces > 500) pieces = 500;
    int chunk = 500 / pieces;
    for (int i = 0; i < pieces; i++) {
        float f = (float)(i + 1) / (float)pieces;
        ik_solve(M, qc, last, mk(p1x + dx * f, p1y + dy * f, 0.46f), Rt, cur);
        for (int k = 0; k < 7; k++) last[k] = cur[k];
        fill(250 + i * chunk, 750, last, true);
    }
    ik_solve(M, qc, last, mk(p2x, p2y, 0.6f), Rt, cur);
    fill(750, 800, cur, true);
    plan_step[env] = 0;
}

// command of this step = plan row plan_step (clamped to the last row); advances plan_step
__global__ void k_plan_fetch(SimParams P, DevPtrs D, const float *plan, int *plan_step, const unsigned char *idle) {
    const int N = P.N;
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= N) return;
    if (idle && idle[env]) {       // macro_action None: zeros(9), the plan keeps its position (env.py:391-393)
        for (int k = 0; k < 9; k++) D.cmd[(size_t)env * 9 + k] = 0.0f;
        return;
    }
    int st = plan_step[env];
    int r = st < PLAN_LEN ? st : PLAN_LEN - 1;
    for (int k = 0; k < 9; k++) D.cmd[(size_t)env * 9 + k] = plan[((size_t)env * PLAN_LEN + r) * 9 + k];
    plan_step[env] = st + 1;
}

// ---------------------------------------------------------------------------------------------- rasteriser
// One workgroup per (env, tile). Tile = full-width strip of tile_h rows (<= TILE_PIX pixels, 8 bytes of LDS key each).
#ifndef RASTER_THREADS
#define RASTER_THREADS 512       // with TILE_PIX 4096: 39.5 KB of LDS -> four workgroups = eight waves per SIMD (A/B below)
#endif
#define VIS_WAS_DYNAMIC (~0ull - 1)   // visibility key of a pixel that was dynamic in the previous frame and is not (yet) now
#define FRAG_VACATED 0x3ffffu         // triangle field of a fragment-list entry for such a pixel: back to the static layer
// A/B at 4096 envs, 128x128 (k_raster ms): 16384 px x 1024 threads (one workgroup per CU) 0.564; 8192 x 512 (two) 0.521;
// 4096 x 512 (three) 0.449; 4096 x 256 0.629; 2048 x 256 0.546.  At 320x240: 0.460 / 0.432 / 0.442.
#ifndef TILE_PIX
#define TILE_PIX 4096
#endif
// LDS of a raster workgroup: 32 KB of keys + 1 KB window list + 1 KB running ends + 1.5 KB matrices + 4 KB clip queue = 39.5 KB:
// four workgroups per CU = eight waves per SIMD, which also needs <= 64 VGPRs (A/B: three workgroups 0.434 ms, four 0.390 ms)
#ifndef MAXWIN
#define MAXWIN 512       // 64-triangle windows per model (rr_create checks nt)
#endif
#define RASTER_INST 24   // instances whose matrices a raster workgroup stages (rr_create checks the model)
#define RASTER_ATTR __attribute__((amdgpu_waves_per_eu(8, 8)))

static_assert(RASTER_INST <= MAXINST && MAXWIN <= 512, "clip queue entries: window position (9 bits) << 6 | lane");
#ifndef CLIPQ
#define CLIPQ 2048       // triangles crossing the near plane per (env, tile) that are clipped (a link cut by the plane has a few hundred;
#endif                   // an overflow raises error flag 4 of the env instead of dropping triangles silently)
#ifndef INLINE_PIX
#define INLINE_PIX 2     // sample points of a small triangle walked by its own lane; the rest is redistributed over the wave
#endif
#define SMALL_AREA 64    // bbox area (sample points) up to which a triangle takes the per-lane + redistribution path (A/B: 16 0.60, 32 0.54, 64 0.53, 128 0.54 ms)

struct STri { float sx[3], sy[3], sz[3], w[3]; };

// Coverage decisions are discontinuous: projection, barycentrics and depth must round exactly like the oracle's --
// geometry a few cm from the near plane projects to huge screen coordinates where one rounding flips whole pixel bands.
// So nothing here is left to the compiler's contraction: the fused operations are written out (a C fmaf in the oracle and
// v_fma_f32 round identically), everything else is compiled contraction-free.
#define FMA3(a0, x0, a1, x1, a2, x2, c) __builtin_fmaf((a0), (x0), __builtin_fmaf((a1), (x1), __builtin_fmaf((a2), (x2), (c))))   // a0 x0 + (a1 x1 + (a2 x2 + c))
#define PDIFF(a, b, c, d) __builtin_fmaf((a), (b), -((c) * (d)))                                                                  // a b - c d: one product, one fused step
// Correctly rounded 1 / x for a normal x whose reciprocal is normal too (|x| within 2^-96 .. 2^96: clip w >= 0.1, signed areas
// 1e-12 .. 1e12): the compiler's IEEE division expands to v_div_scale x 2, v_rcp, a Newton step, two residual corrections,
// v_div_fmas, v_div_fixup -- eleven instructions; inside that range the scaling and the fix-up are the identity and what remains is
// this sequence, operation for operation (so the bits are those of `1.0f / x` and of the oracle's division): eight instructions.
__device__ __forceinline__ float recip_exact(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e0 = __builtin_fmaf(-x, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float q0 = r1;                                   // numerator 1
    const float e1 = __builtin_fmaf(-x, q0, 1.0f);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-x, q1, 1.0f);
    return __builtin_fmaf(e2, r1, q1);
}
__device__ __forceinline__ bool project_tri(const float *mvp, const float *tp /*9 floats*/, int W, int H, STri &s) {
#pragma clang fp contract(off)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float vx = tp[3 * k], vy = tp[3 * k + 1], vz = tp[3 * k + 2];
        float cx = FMA3(mvp[0], vx, mvp[1], vy, mvp[2], vz, mvp[3]);
        float cy = FMA3(mvp[4], vx, mvp[5], vy, mvp[6], vz, mvp[7]);
        float cz = FMA3(mvp[8], vx, mvp[9], vy, mvp[10], vz, mvp[11]);
        float cw = FMA3(mvp[12], vx, mvp[13], vy, mvp[14], vz, mvp[15]);
        if (cw < 0.1f) return false;
        float iw = recip_exact(cw);
        s.sx[k] = __builtin_fmaf(cx * iw, 0.5f * (float)W, 0.5f * (float)W);
        s.sy[k] = __builtin_fmaf(cy * iw, 0.5f * (float)H, 0.5f * (float)H);
        s.sz[k] = cz * iw;
        s.w[k] = iw;             // reciprocal clip w (deferred shading)
    }
    return true;
}

// One vertex of project_tri (identical arithmetic); a vertex nearer than the near plane yields sx = NaN.
__device__ __forceinline__ void project_vertex(const float *mvp, float vx, float vy, float vz, int W, int H, float &sx, float &sy, float &sz) {
#pragma clang fp contract(off)
    const float cx = FMA3(mvp[0], vx, mvp[1], vy, mvp[2], vz, mvp[3]);
    const float cy = FMA3(mvp[4], vx, mvp[5], vy, mvp[6], vz, mvp[7]);
    const float cz = FMA3(mvp[8], vx, mvp[9], vy, mvp[10], vz, mvp[11]);
    const float cw = FMA3(mvp[12], vx, mvp[13], vy, mvp[14], vz, mvp[15]);
    const float iw = recip_exact(cw);                         // (a vertex behind the near plane is flagged below; its values are never used)
    sx = (cw < 0.1f) ? __int_as_float(0x7fc00000) : __builtin_fmaf(cx * iw, 0.5f * (float)W, 0.5f * (float)W);
    sy = __builtin_fmaf(cy * iw, 0.5f * (float)H, 0.5f * (float)H);
    sz = cz * iw;
}
// Near-plane clipping (oracle clip_near / to_screen, identical arithmetic): clip coordinates of one vertex, the screen
// position of a clip-space point, and the intersection of an edge with w = NEAR_W computed from the inside end a towards
// the outside end b.
#define NEAR_W 0.1f
__device__ __forceinline__ void clip_vertex(const float *mvp, float vx, float vy, float vz, float *c) {
#pragma clang fp contract(off)
    c[0] = FMA3(mvp[0], vx, mvp[1], vy, mvp[2], vz, mvp[3]);
    c[1] = FMA3(mvp[4], vx, mvp[5], vy, mvp[6], vz, mvp[7]);
    c[2] = FMA3(mvp[8], vx, mvp[9], vy, mvp[10], vz, mvp[11]);
    c[3] = FMA3(mvp[12], vx, mvp[13], vy, mvp[14], vz, mvp[15]);
}
__device__ __forceinline__ void clip_to_screen(const float *c, int W, int H, float &sx, float &sy, float &sz) {
#pragma clang fp contract(off)
    const float iw = 1.0f / c[3];
    sx = __builtin_fmaf(c[0] * iw, 0.5f * (float)W, 0.5f * (float)W);
    sy = __builtin_fmaf(c[1] * iw, 0.5f * (float)H, 0.5f * (float)H);
    sz = c[2] * iw;
}
__device__ __forceinline__ void clip_edge(const float *a, const float *b, float *o) {
#pragma clang fp contract(off)
    const float t = (NEAR_W - a[3]) / (b[3] - a[3]);
    o[0] = __builtin_fmaf(t, b[0] - a[0], a[0]);
    o[1] = __builtin_fmaf(t, b[1] - a[1], a[1]);
    o[2] = __builtin_fmaf(t, b[2] - a[2], a[2]);
    o[3] = NEAR_W;
}

__device__ __forceinline__ bool bary(const STri &s, float px, float py, float *b) {
#pragma clang fp contract(off)
    float x0 = s.sx[0], y0 = s.sy[0], x1 = s.sx[1], y1 = s.sy[1], x2 = s.sx[2], y2 = s.sy[2];
    float area = PDIFF(x1 - x0, y2 - y0, x2 - x0, y1 - y0);
    if (fabsf(area) < 1e-12f) return false;
    float ia = recip_exact(area);
    b[0] = PDIFF(x1 - px, y2 - py, x2 - px, y1 - py) * ia;
    b[1] = PDIFF(x2 - px, y0 - py, x0 - px, y2 - py) * ia;
    b[2] = 1.0f - b[0] - b[1];
    return b[0] >= 0 && b[1] >= 0 && b[2] >= 0;
}

__device__ __forceinline__ void raster_pixel(const STri &s, int t, int px, int py, int H, int W, int row0, int rows,
                                             unsigned long long *vis) {
#pragma clang fp contract(off)
    int row = H - 1 - py;
    if (row < row0 || row >= row0 + rows) return;
    float b[3];
    if (!bary(s, (float)px, (float)py, b)) return;
    float z = __builtin_fmaf(b[0], s.sz[0], __builtin_fmaf(b[1], s.sz[1], b[2] * s.sz[2]));
    float d = __builtin_fmaf(0.5f, z, 0.5f);
    if (!(d >= 0.0f && d <= 1.0f)) return;
    unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)t;
    atomicMin(&vis[(row - row0) * W + px], key);
}

// Same arithmetic as bary()/raster_pixel() with the per-triangle part (signed area, its reciprocal) hoisted out of the
// pixel loop; (px, py) must already lie inside the tile.
struct TriEdge { float ia; bool ok; };
__device__ __forceinline__ TriEdge tri_edge(const STri &s) {
#pragma clang fp contract(off)
    float area = PDIFF(s.sx[1] - s.sx[0], s.sy[2] - s.sy[0], s.sx[2] - s.sx[0], s.sy[1] - s.sy[0]);
    TriEdge e;
    e.ok = !(fabsf(area) < 1e-12f);
    e.ia = recip_exact(area);                               // (|area| >= 1e-12 when ok; a degenerate triangle is not drawn and its value never used)
    return e;
}
__device__ __forceinline__ void raster_pixel_hoisted(const STri &s, float ia, int t, int px, int py, int H, int W /* row stride of the tile's buffer */, int row0,
                                                     unsigned long long *vis, int xoff = 0 /* first column of the tile */) {
#pragma clang fp contract(off)
    const float fx = (float)px, fy = (float)py;
    const float b0 = PDIFF(s.sx[1] - fx, s.sy[2] - fy, s.sx[2] - fx, s.sy[1] - fy) * ia;
    const float b1 = PDIFF(s.sx[2] - fx, s.sy[0] - fy, s.sx[0] - fx, s.sy[2] - fy) * ia;
    const float b2 = 1.0f - b0 - b1;
    if (!(b0 >= 0 && b1 >= 0 && b2 >= 0)) return;
    const float z = __builtin_fmaf(b0, s.sz[0], __builtin_fmaf(b1, s.sz[1], b2 * s.sz[2]));
    const float d = __builtin_fmaf(0.5f, z, 0.5f);
    if (!(d >= 0.0f && d <= 1.0f)) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)t;
    atomicMin(&vis[(H - 1 - py - row0) * W + px - xoff], key);
}

// The same for a sample point given as floats (integer valued: exact) with its index in the tile's visibility buffer already known.
__device__ __forceinline__ void raster_pixel_at(const STri &s, float ia, int t, float fx, float fy, int vidx, unsigned long long *vis) {
#pragma clang fp contract(off)
    const float b0 = PDIFF(s.sx[1] - fx, s.sy[2] - fy, s.sx[2] - fx, s.sy[1] - fy) * ia;
    const float b1 = PDIFF(s.sx[2] - fx, s.sy[0] - fy, s.sx[0] - fx, s.sy[2] - fy) * ia;
    const float b2 = 1.0f - b0 - b1;
    if (!(b0 >= 0 && b1 >= 0 && b2 >= 0)) return;
    const float z = __builtin_fmaf(b0, s.sz[0], __builtin_fmaf(b1, s.sz[1], b2 * s.sz[2]));
    const float d = __builtin_fmaf(0.5f, z, 0.5f);
    if (!(d >= 0.0f && d <= 1.0f)) return;
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)t;
    atomicMin(&vis[vidx], key);
}

// Conservative test "can any sample point of the pixel rectangle [px0,px1]x[py0,py1] pass raster_pixel_hoisted's
// inside test?".  Exact barycentrics are affine in (px, py), so their maximum over the rectangle is at a corner; the
// float evaluation of raster_pixel_hoisted differs from the exact value by at most delta (see the caller), hence a
// rectangle whose four corner values of some barycentric are all < -2*delta cannot contain a covered sample point.
__device__ __forceinline__ bool block_may_overlap(const STri &s, float ia, int px0, int px1, int py0, int py1, float delta2) {
#pragma clang fp contract(off)
    float m0 = -3.0e38f, m1 = -3.0e38f, m2 = -3.0e38f;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float fx = (float)((c & 1) ? px1 : px0), fy = (float)((c & 2) ? py1 : py0);
        const float b0 = PDIFF(s.sx[1] - fx, s.sy[2] - fy, s.sx[2] - fx, s.sy[1] - fy) * ia;
        const float b1 = PDIFF(s.sx[2] - fx, s.sy[0] - fy, s.sx[0] - fx, s.sy[2] - fy) * ia;
        const float b2 = 1.0f - b0 - b1;
        m0 = fmaxf(m0, b0); m1 = fmaxf(m1, b1); m2 = fmaxf(m2, b2);
    }
    return !(m0 < -delta2 || m1 < -delta2 || m2 < -delta2);
}

// Deferred shading of one pixel from its visibility key (depth bits | triangle id): re-projects the winning triangle,
// perspective-correct barycentrics, interpolated normal -> Phong-like TinyRenderer shading, nearest texel.
// mvp / sinst: per-instance constants of this env staged in LDS by stage_instances(); sinst holds 16 floats per instance
// {R[9], colour[3], tex_off, tex_w (0: untextured), tex_h, uid}, which keeps the chain of dependent global loads of a
// shaded pixel at two (triangle record, texel).
struct ShadeCtx { const DevPtrs *D; const float *mvp; const float *sinst; int W, H; };
// one 128-byte record per triangle: 7 x 16-byte loads from a single cache line instead of 26 scattered dwords
// (every lane shades a different triangle, so the cost of a load is its number of distinct lines)
struct TriRec { float4 v[7]; };
__device__ __forceinline__ TriRec load_tri_rec(const DevPtrs &D, int t) {
    TriRec r;
    const float4 *rp = D.tri_rec + (size_t)8 * t;
#pragma unroll
    for (int k = 0; k < 7; k++) r.v[k] = rp[k];
    return r;
}
__device__ __forceinline__ void shade_pixel(const ShadeCtx &c, const TriRec &tr, int px, int row, unsigned char *rgb3, int &mask) {
    const DevPtrs &D = *c.D;
    const int W = c.W, H = c.H;
    const float Lx = -50.0f, Ly = 30.0f, Lz = 100.0f;
    const float linv = 1.0f / sqrtf(Lx * Lx + Ly * Ly + Lz * Lz);
    const float L0 = Lx * linv, L1 = Ly * linv, L2 = Lz * linv;
    float rec[28];
#pragma unroll
    for (int k = 0; k < 7; k++) { rec[4 * k] = tr.v[k].x; rec[4 * k + 1] = tr.v[k].y; rec[4 * k + 2] = tr.v[k].z; rec[4 * k + 3] = tr.v[k].w; }
    const float *tp = rec, *nn = rec + 9, *uv = rec + 18;
    const int inst = __float_as_int(rec[24]);
    STri s;
    float c0, c1, c2;
    if (project_tri(c.mvp + inst * 16, tp, W, H, s)) {
        float b[3] = {0, 0, 0};
        bary(s, (float)px, (float)(H - 1 - row), b);
        // perspective-correct weights b_k / w_k: s.w holds 1 / w_k (the reciprocal the projection computed anyway)
        c0 = b[0] * s.w[0]; c1 = b[1] * s.w[1]; c2 = b[2] * s.w[2];
    } else {
        // a corner is nearer than the near plane (the triangle was clipped by k_raster): perspective-correct weights straight
        // from the clip coordinates, c ~ u x v with u_i = x_i - xn w_i, v_i = y_i - yn w_i (oracle rro_render, same arithmetic)
#pragma clang fp contract(off)
        float k0[4], k1[4], k2[4];
        clip_vertex(c.mvp + inst * 16, tp[0], tp[1], tp[2], k0);
        clip_vertex(c.mvp + inst * 16, tp[3], tp[4], tp[5], k1);
        clip_vertex(c.mvp + inst * 16, tp[6], tp[7], tp[8], k2);
        const float xn = (float)px * (2.0f / (float)W) - 1.0f, yn = (float)(H - 1 - row) * (2.0f / (float)H) - 1.0f;
        const float u0 = k0[0] - xn * k0[3], u1 = k1[0] - xn * k1[3], u2 = k2[0] - xn * k2[3];
        const float v0 = k0[1] - yn * k0[3], v1 = k1[1] - yn * k1[3], v2 = k2[1] - yn * k2[3];
        c0 = u1 * v2 - u2 * v1; c1 = u2 * v0 - u0 * v2; c2 = u0 * v1 - u1 * v0;
    }
    float cs = recip_exact(c0 + c1 + c2);              // (sum of the perspective weights: of the order of 1 / w, far inside the range)
    c0 *= cs; c1 *= cs; c2 *= cs;
    float n0 = c0 * nn[0] + c1 * nn[3] + c2 * nn[6], n1 = c0 * nn[1] + c1 * nn[4] + c2 * nn[7], n2 = c0 * nn[2] + c1 * nn[5] + c2 * nn[8];
    const float *xf = c.sinst + inst * 16;
    float w0 = xf[0] * n0 + xf[1] * n1 + xf[2] * n2, w1 = xf[3] * n0 + xf[4] * n1 + xf[5] * n2, w2 = xf[6] * n0 + xf[7] * n1 + xf[8] * n2;
    float nlen = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    if (nlen > 0) { const float inl = recip_exact(nlen); w0 *= inl; w1 *= inl; w2 *= inl; }      // (a rotated unit normal: length ~1)
    float ndl = w0 * L0 + w1 * L1 + w2 * L2;
    float diff = fmaxf(ndl, 0.0f);
    float r0 = w0 * (2 * ndl) - L0, r1 = w1 * (2 * ndl) - L1, r2 = w2 * (2 * ndl) - L2;
    float rl = sqrtf(r0 * r0 + r1 * r1 + r2 * r2);
    float rz = rl > 0 ? fmaxf(r2 / rl, 0.0f) : 0.0f;
    float spec = rz * rz;
    float tex0 = 255.0f, tex1 = 255.0f, tex2 = 255.0f;
    const int tw = __float_as_int(xf[13]);
    if (tw > 0) {
        float u = c0 * uv[0] + c1 * uv[2] + c2 * uv[4], v = c0 * uv[1] + c1 * uv[3] + c2 * uv[5];
        u = u - floorf(u); v = v - floorf(v);
        const int th = __float_as_int(xf[14]);
        int tx = min((int)(u * (float)tw), tw - 1), ty = min((int)(v * (float)th), th - 1);
#if defined(RR_SHADE_PROBE) && (RR_SHADE_PROBE & 2)
        unsigned px4 = D.tex[(size_t)__float_as_int(xf[12]) + (size_t)((tx + ty) & 63)];      // (probe: texels of one cache-line set)
#else
        unsigned px4 = D.tex[(size_t)__float_as_int(xf[12]) + (size_t)(th - 1 - ty) * tw + tx];
#endif
        tex0 = (float)(px4 & 255); tex1 = (float)((px4 >> 8) & 255); tex2 = (float)((px4 >> 16) & 255);
    }
    float shade = 0.6f + 0.35f * diff + 0.05f * spec;
    rgb3[0] = (unsigned char)min((int)(tex0 * xf[9] * shade), 255);
    rgb3[1] = (unsigned char)min((int)(tex1 * xf[10] * shade), 255);
    rgb3[2] = (unsigned char)min((int)(tex2 * xf[11] * shade), 255);
    mask = __float_as_int(xf[15]);
}

// Stages the per-instance constants of one env (computed by k_render_setup) in LDS: mvp and, when sinst != nullptr, the
// shading constants. A plain copy, 16 bytes per thread. Caller synchronises.
__device__ __forceinline__ void stage_instances(const RenderModel &RM, const DevPtrs &D, int env, int tid, int nthreads,
                                                float (*mvp)[16], float (*sinst)[16]) {
    const float4 *src = (const float4 *)(D.inst_xf + (size_t)env * MAXINST * 32);
    for (int i = tid; i < RM.ni * 8; i += nthreads) {
        const int inst = i >> 3, q = i & 7;
        if (q < 4) *(float4 *)&mvp[inst][4 * q] = src[i];
        else if (sinst) *(float4 *)&sinst[inst][4 * (q - 4)] = src[i];
    }
}

#ifdef RR_RASTER_STATS
// Development-only work counters (librealrobot_hip_stats.so, `make stats`; never part of the shipped library).
__device__ unsigned long long g_rstats[16];
#define RSTAT(i, v) do { if (P.ablate & 0x8000) atomicAdd(&g_rstats[i], (unsigned long long)(v)); } while (0)   /* RR_ABLATE=32768 */
extern "C" int rr_debug_raster_stats(unsigned long long *out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_rstats), sizeof(g_rstats)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_rstats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
// phase clock of the window loop (RR_ABLATE=4096; scratch/rphase.py): wave cycles between phase marks; every workgroup sums its
// waves in LDS and writes its ten totals to a slot of its own (global atomics from every wave would be what the clock measures)
__device__ unsigned g_rphase_wg[65536][10];
#define PH_DECL unsigned long long ph_t_ = __builtin_readcyclecounter(); unsigned pa0_ = 0, pa1_ = 0, pa2_ = 0, pa3_ = 0, pa4_ = 0, pa5_ = 0, pa6_ = 0, pa7_ = 0, pa8_ = 0, pa9_ = 0; \
    __shared__ unsigned ph_lds_[10]; if (tid < 10) ph_lds_[tid] = 0;
#define PH(i) do { if (P.ablate & 0x1000) { const unsigned long long n_ = __builtin_readcyclecounter(); pa##i##_ += (unsigned)(n_ - ph_t_); ph_t_ = n_; } } while (0)
#define PH_FLUSH do { if (P.ablate & 0x1000) { if (lane == 0) { atomicAdd(&ph_lds_[0], pa0_); atomicAdd(&ph_lds_[1], pa1_); atomicAdd(&ph_lds_[2], pa2_); atomicAdd(&ph_lds_[3], pa3_); atomicAdd(&ph_lds_[4], pa4_); \
    atomicAdd(&ph_lds_[5], pa5_); atomicAdd(&ph_lds_[6], pa6_); atomicAdd(&ph_lds_[7], pa7_); atomicAdd(&ph_lds_[8], pa8_); atomicAdd(&ph_lds_[9], pa9_); } __syncthreads(); \
    if (tid < 10) g_rphase_wg[(env * RM.ntiles + tile) & 65535][tid] = ph_lds_[tid]; } } while (0)
extern "C" int rr_debug_raster_phase(unsigned long long *out16, int reset) {
    static unsigned h[65536][10];
    if (out16) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rphase_wg), sizeof(h)) != hipSuccess) return -1;
        for (int i = 0; i < 16; i++) out16[i] = 0;
        for (int w = 0; w < 65536; w++) for (int i = 0; i < 10; i++) out16[i] += h[w][i];
    }
    if (reset) { memset(h, 0, sizeof(h)); if (hipMemcpyToSymbol(HIP_SYMBOL(g_rphase_wg), h, sizeof(h)) != hipSuccess) return -1; }
    return 0;
}
// workgroup timeline of k_raster (RR_ABLATE=16384; scratch/rwgtime.py): start / end on the 100 MHz clock and the hardware slot
// (HW_ID, XCC_ID) of every workgroup of the last launch -- how full the four workgroup slots of a CU are kept
__device__ unsigned long long g_rwg_time[65536][3];
__device__ int g_shade_ablate;      // (k_shade has no SimParams: rr_debug_shade_ablate sets what it records)
extern "C" int rr_debug_shade_ablate(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_shade_ablate), &v, sizeof(v)) == hipSuccess ? 0 : -1; }
extern "C" int rr_debug_raster_wgtime(unsigned long long *out /*[65536][3]*/, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rwg_time), sizeof(g_rwg_time)) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[65536][3]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_rwg_time), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
// ... and marks inside a workgroup (thread 0, 100 MHz clock): 0 kernel entry, 1 selection flags there, 2 LDS filled + instances staged,
// 3 clusters culled, 4 window loop done, 5 near-plane pass done, 6 list written
__device__ unsigned long long g_rwg_marks[65536][8];
#define WGM(i) do { if ((P.ablate & 0x4000) && threadIdx.x == 0) g_rwg_marks[(env * RM.ntiles + tile) & 65535][i] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int rr_debug_raster_wgmarks(unsigned long long *out /*[65536][8]*/) {
    return (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rwg_marks), sizeof(g_rwg_marks)) == hipSuccess) ? 0 : -1;
}
// phase ablations of k_raster for the time breakdown in DESIGN.md (RR_ABLATE bits: 1 no rasterisation after projection,
// 2 no wave-cooperative path, 8 no triangles at all); compiled out of the shipped library
#define ABL(bit) (P.ablate & (bit))
#else
#define RSTAT(i, v)
#define WGM(i)
#define ABL(bit) false
#define PH_DECL
#define PH(i)
#define PH_FLUSH
#endif

// The cluster cull of raster_tile: clip-space centre of the cluster's bounding sphere against the five
// frustum planes, then against the planes of a tile's first / last sample row and column (RM.tile_plane), radius enlarged -- conservative.
struct ClusterClip { float cx, cy, cw, r; };
__device__ __forceinline__ bool cluster_outside_frustum(const RenderModel &RM, const float *m, const float4 cs, ClusterClip &c) {
    c.cx = m[0] * cs.x + m[1] * cs.y + m[2] * cs.z + m[3]; c.cy = m[4] * cs.x + m[5] * cs.y + m[6] * cs.z + m[7];
    c.cw = m[12] * cs.x + m[13] * cs.y + m[14] * cs.z + m[15];
    c.r = cs.w * 1.001f + 1e-4f;
    return (c.cw + c.cx) < -c.r * RM.plane_norm[0] || (c.cw - c.cx) < -c.r * RM.plane_norm[1] || (c.cw + c.cy) < -c.r * RM.plane_norm[2] ||
           (c.cw - c.cy) < -c.r * RM.plane_norm[3] || (c.cw - 0.1f) < -c.r * RM.plane_norm[4];
}
__device__ __forceinline__ bool cluster_outside_tile(const ClusterClip &c, const float *tp /*RM.tile_plane[tile]*/, bool tiled, bool xtiled) {
    return (tiled && ((tp[0] * c.cw - c.cy) > c.r * tp[1] * 1.001f + 1e-4f * c.cw || (c.cy - tp[2] * c.cw) > c.r * tp[3] * 1.001f + 1e-4f * c.cw)) ||
           (xtiled && ((tp[4] * c.cw - c.cx) > c.r * tp[5] * 1.001f + 1e-4f * c.cw || (c.cx - tp[6] * c.cw) > c.r * tp[7] * 1.001f + 1e-4f * c.cw));
}

// Visibility pass of one (env, tile).  pass 0 = per-env frame: starts from the static layer's keys when D.static_vis !=
// nullptr and rasterises only the triangles of moving instances; pass 1 = static layer (instances that never move: table,
// shelf, robot base link_0; one launch at creation, result shared by all envs).  Output: the list of pixels won by a
// rasterised triangle {depth bits, pixel-in-tile << 18 | triangle}, shaded by k_shade; everything else in the image is the
// static layer, copied by k_static_copy.
struct ImageOut { unsigned char *rgb; float *depth; int *mask; size_t env_stride; /* pixels between envs */ };
// the tile's visibility buffer (file scope: the list-walking render kernels stage their shading constants in it once the
// fragment list is out -- their workgroups then need no more LDS than a raster workgroup)
__shared__ __attribute__((aligned(16))) unsigned long long g_vis[TILE_PIX];
// PRE: the caller has fetched this thread's two words of the previous frame's list (count, first entry) beside its own loads.
template <int NT_, bool CARRY = true, bool LOOPED = false, bool PRE = false>
__device__ __forceinline__ void raster_tile(const SimParams &P, const RenderModel &RM, const DevPtrs &D, int n_inst_used, int pass, int env, int tile, int restore,
                                            unsigned n_old_pre = 0u, unsigned en_first_pre = 0u) {
    unsigned long long *vis = g_vis;
    __shared__ __attribute__((aligned(16))) float mvp[RASTER_INST][16];
    __shared__ unsigned nlist, wcount, wnext;
    __shared__ unsigned short wlist[MAXWIN];
    __shared__ unsigned short wends[NT_ / 64][64];      // per wave: running ends of the lanes' left-over points (<= 64 x SMALL_AREA)
    __shared__ unsigned short clipq[CLIPQ];             // triangles that cross the near plane (rare), clipped after the window loop: position in wlist << 6 | lane
    __shared__ unsigned nclipq;
    const int W = RM.W, H = RM.H;
    // the tile: rows [row0, row0 + rows) x columns [tx0i, tx0i + cols); its visibility buffer has TW pixels per row
    const int TW = RM.tile_w, tyi = tile / RM.ntx, tx0i = (tile - tyi * RM.ntx) * TW;
    const int row0 = tyi * RM.tile_h;
    const int rows = min(RM.tile_h, H - row0), cols = min(TW, W - tx0i);
    const int npix = rows * TW;
    // (k_raster's instance -- PRE -- leaves the duration of the workgroup for the next frame's dispatch order, raster_order_class; the
    // start time waits in LDS: scalar registers held across the whole pass were spilled)
    __shared__ unsigned s_tc0;
    if (PRE && threadIdx.x == 0) s_tc0 = (unsigned)__builtin_amdgcn_s_memrealtime();
    int tid_ = threadIdx.x;
    if (LOOPED) asm volatile("" : "+v"(tid_));      // (inside an item loop: nothing derived from the thread index is kept live across items)
    const int tid = tid_;
    const bool layered = (pass == 0) && (D.static_vis != nullptr);
    // (incremental image update, see below: this thread's entry of the previous frame's list is fetched first -- its
    // round trip hides behind the LDS fill and the instance staging)
    // (a workgroup of an empty tile is nothing but a chain of round trips to memory -- 5 us, and two thirds of the tiles of the
    // benchmark camera are nearly empty, scratch/rwgtime.py: every load whose address is known is issued before the first wait)
    const unsigned n_old = restore ? (PRE ? n_old_pre : D.frag_count[(size_t)env * RM.ntiles + tile]) : 0u;
    const uint2 *old_lst = D.frag_list + ((size_t)env * RM.ntiles + tile) * TILE_PIX;
    const unsigned en_first = (unsigned)tid < n_old ? (PRE ? en_first_pre : old_lst[tid].y) : 0xffffffffu;
    // the cull's inputs of this thread's cluster (one cluster per thread: MAXWIN <= NT_), fetched ahead of the LDS fill and the barrier
    const int t_begin_ = ((pass == 0) && (D.static_vis != nullptr)) ? RM.first_dynamic_tri : 0;
    const int nwin_ = ((ABL(8) ? 0 : ((pass == 1) ? RM.first_dynamic_tri : RM.nt)) - t_begin_ + 63) >> 6;
    const bool cull_mine = tid < nwin_;
    const int cull_tb = min(t_begin_ + (min(tid, max(nwin_ - 1, 0)) << 6), RM.nt - 64);      // (clamped: the loads are unconditional, straight-line code)
    const int cull_inst = D.tri_inst[cull_tb];
    const float4 cull_cs = D.cluster_sphere[cull_tb >> 6];
    // The tile starts empty (an LDS-only fill); the static layer's keys are compared at compaction time and only for the
    // few pixels a moving triangle reached (min is associative) -- no 128 KB read of the static keys per env.
    for (int i = tid; i < npix; i += NT_) vis[i] = ~0ull;
    if (tid == 0) { nlist = 0; wcount = 0; wnext = 0; nclipq = 0; }
    stage_instances(RM, D, env, tid, NT_, mvp, nullptr);
    __syncthreads();
    WGM(2);
    // tile bounds in screen y (py = H-1-row)
    const float ty0 = (float)(H - 1 - (row0 + rows - 1)), ty1 = (float)(H - 1 - row0);
    const float txlo = (float)tx0i, txhi = (float)(tx0i + cols - 1);      // ... and in screen x
    const int NT = RM.nt;
    const int t_begin = layered ? RM.first_dynamic_tri : 0;
    const int t_end = (pass == 1) ? RM.first_dynamic_tri : NT;
    // Window queue: one thread per 64-triangle cluster ("window"; never spans two instances) runs the frustum test of the
    // cluster's bounding sphere and appends survivors to an LDS list; the waves then pull windows from that list with an
    // LDS counter, so culled windows cost nothing in the wave loops and waves that drew cheap windows take more of them.
    const int t_stop = ABL(8) ? 0 : t_end;
    const int lane = tid & 63, lx = lane & 7, ly = lane >> 3;
    const int nwin = (t_stop - t_begin + 63) >> 6;
    // tile bounds as two more planes of the cull (multi-tile images): NDC y of the tile's first and last sample rows
    const bool xtiled = RM.ntx > 1;
    const bool tiled = RM.ntiles > 1;
    // Incremental image update (do_render): the env's image in HBM still holds its previous frame.  The pixels of that
    // frame's fragment list are marked in the (still empty) visibility buffer with a key above every real one; those that
    // no triangle reaches this time leave the compaction as "vacated" entries, which k_shade puts back to the static layer.
    if (restore) {
        if (en_first != 0xffffffffu && (en_first & 0x3ffffu) != FRAG_VACATED) vis[en_first >> 18] = VIS_WAS_DYNAMIC;
        for (unsigned i = tid + NT_; i < n_old; i += NT_) {          // more than 1024 entries: rare
            const unsigned en = old_lst[i].y;
            if ((en & 0x3ffffu) != FRAG_VACATED) vis[en >> 18] = VIS_WAS_DYNAMIC;     // (a vacated entry was put back last time)
        }
    }
    static_assert(MAXWIN <= NT_, "one cluster per thread in the cull");
    if (cull_mine) {
        const int wi = tid;
        int inst = cull_inst;
        float4 cs = cull_cs;
        asm("" : "+v"(inst), "+v"(cs.x), "+v"(cs.y), "+v"(cs.z), "+v"(cs.w));      // (nothing computed from them ahead of the barrier: the wait for the two loads stays down here)
        ClusterClip cc;
        const bool out = cluster_outside_frustum(RM, mvp[inst], cs, cc) || cluster_outside_tile(cc, RM.tile_plane[tile], tiled, xtiled);
        RSTAT(0, 1);                                // windows
        if (!out && inst < n_inst_used) wlist[atomicAdd(&wcount, 1u)] = (unsigned short)wi;
    }
    __syncthreads();
    WGM(3);
    // Every wave takes 64 consecutive triangles per window.  A triangle whose clipped bounding box holds <= small_area
    // sample points is rasterised by its own lane; bigger ones are handed to the whole wave (ballot, v_readlane broadcast
    // of the projected triangle, 64 sample points per step in 8x8 blocks) so that one large triangle does not make 63
    // lanes wait.
    const unsigned nw = wcount;
    // Nothing reaches this tile and nothing was drawn in it last time (more than half of the lower tiles of the benchmark camera):
    // its list stays empty -- no window loop, no near-plane pass, no compaction of an untouched buffer (3 of the 6.6 us such a
    // workgroup lasts, scratch/rwgtime.py)
    if (pass == 0 && nw == 0 && n_old == 0) {
        if (tid == 0) { D.frag_count[(size_t)env * RM.ntiles + tile] = 0; RSTAT(13, 1); if (PRE && D.item_cost) D.item_cost[(size_t)env * RM.ntiles + tile] = (unsigned)__builtin_amdgcn_s_memrealtime() - s_tc0 + 1u; }
        return;
    }
#ifdef RR_RASTER_STATS
    const unsigned long long t_loop0_ = __builtin_readcyclecounter();
#endif
    PH_DECL
    for (;;) {
        unsigned k = 0;
        if (lane == 0) k = atomicAdd(&wnext, 1u);
        k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
        if (k >= nw) break;
        PH(0);                                      // window fetch
        const int tb = t_begin + ((int)wlist[k] << 6);
        if (lane == 0) RSTAT(1, 1);                 // windows that pass the cluster test
        const int t = tb + lane;
        bool live = t < t_stop;
        const int inst = D.tri_inst[tb];
        STri s;
        int x0 = 0, y0 = 0, x1 = -1, y1 = -1, area = 0;
        bool needs_clip = false;
        float ia = 0.0f;
        {
            // The window is one cluster: lane l projects the cluster's vertex l (<= 64 distinct positions for its 64
            // triangles, so a vertex is transformed once instead of once per incident corner); every triangle lane then
            // gathers its three corners from the owning lanes (ds_bpermute).  Same arithmetic as project_tri.
            const float *cv = D.cluster_verts + (size_t)(tb >> 6) * 192;
            float psx, psy, psz;
            project_vertex(mvp[inst], cv[lane], cv[64 + lane], cv[128 + lane], W, H, psx, psy, psz);
            const int vi = D.tri_vidx[t];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int adr = (vi >> (8 * k)) & 255;               // (byte address of the owning lane: v_bfe_u32)
                s.sx[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(adr, __float_as_int(psx)));
                s.sy[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(adr, __float_as_int(psy)));
                s.sz[k] = __int_as_float(__builtin_amdgcn_ds_bpermute(adr, __float_as_int(psz)));
                s.w[k] = 1.0f;
            }
#if defined(RR_PROBE_BPERM) || defined(RR_PROBE_VALU) || defined(RR_PROBE_DSMIN)
            // (sensitivity probes, development variants only: extra LDS permutes / VALU work / LDS atomics per window -- which pipe
            // does the kernel's duration follow?  scratch/variants)
            {
                float pv0 = psx, pv1 = psy, pv2 = psz, pv3 = psx + psy;
#ifdef RR_PROBE_BPERM
                for (int i_ = 0; i_ < RR_PROBE_BPERM / 4; i_++) { pv0 = lane_gather(pv0, (lane + 1) & 63); pv1 = lane_gather(pv1, (lane + 3) & 63); pv2 = lane_gather(pv2, (lane + 5) & 63); pv3 = lane_gather(pv3, (lane + 7) & 63); }
#endif
#ifdef RR_PROBE_VALU
#pragma unroll
                for (int i_ = 0; i_ < RR_PROBE_VALU / 4; i_++) { pv0 = __builtin_fmaf(pv0, 1.0001f, 0.5f); pv1 = __builtin_fmaf(pv1, 1.0001f, 0.5f); pv2 = __builtin_fmaf(pv2, 1.0001f, 0.5f); pv3 = __builtin_fmaf(pv3, 1.0001f, 0.5f); }
#endif
#ifdef RR_PROBE_DSMIN
                for (int i_ = 0; i_ < RR_PROBE_DSMIN; i_++) atomicMin(&vis[(lane * 61 + i_ * 7) & (TILE_PIX - 1)], ~0ull);      // (never changes a key)
#endif
                if (pv0 + pv1 + pv2 + pv3 == 12345.678f) s.sz[0] = pv0;
            }
#endif
            // corners nearer than the near plane (sx = NaN): none -> the ordinary paths below; all -> nothing to draw;
            // one or two -> the triangle is clipped against the plane by the whole wave (rare, see the end of the loop body)
            // (a NaN among the three sx: their sum is NaN; all three: their minimum is NaN too -- v_min3 returns a number when it has one)
            const float sxsum = s.sx[0] + s.sx[1] + s.sx[2], sxmin = fminf(s.sx[0], fminf(s.sx[1], s.sx[2]));
            const bool any_near = sxsum != sxsum, all_near = sxmin != sxmin;
            needs_clip = live && any_near && !all_near;
            live = live && !any_near;
        }
        PH(1);                                      // loads, projection, corner gather
        {
            // Straight-line set-up for all 64 lanes (nested `if (live)` blocks save nothing in lock-step and cost exec-mask bookkeeping
            // and re-initialisation on every path): box, clamps as one v_med3 each -- the visibility test in front of them
            // guarantees xmax >= the tile's first column, xmin <= its last, ... so med3(x, lo, hi) is max(x, lo) resp. min(x, hi): the
            // oracle's box clipped to the tile --,
            // signed area and its reciprocal; a lane that is not live ends with area 0.
            const float xmin = fminf(s.sx[0], fminf(s.sx[1], s.sx[2])), xmax = fmaxf(s.sx[0], fmaxf(s.sx[1], s.sx[2]));
            const float ymin = fminf(s.sy[0], fminf(s.sy[1], s.sy[2])), ymax = fmaxf(s.sy[0], fmaxf(s.sy[1], s.sy[2]));
            bool on = live && !(xmax < txlo || ymax < ty0 || xmin > txhi || ymin > ty1);
            const TriEdge te = tri_edge(s);
            // back faces of closed, consistently wound meshes can never win the depth test (opt-in, RR_CULL; wave-uniform per window)
            if (RM.any_cull && RM.in_cull[inst]) on = on && !(PDIFF(s.sx[1] - s.sx[0], s.sy[2] - s.sy[0], s.sx[2] - s.sx[0], s.sy[1] - s.sy[0]) <= 0.0f);
            x0 = (int)ceilf(__builtin_amdgcn_fmed3f(xmin, txlo, txhi)); x1 = (int)floorf(__builtin_amdgcn_fmed3f(xmax, txlo, txhi));
            y0 = (int)ceilf(__builtin_amdgcn_fmed3f(ymin, ty0, ty1)); y1 = (int)floorf(__builtin_amdgcn_fmed3f(ymax, ty0, ty1));
            on = on && !(x1 < x0 || y1 < y0) && te.ok;
            ia = te.ia;
            area = on ? (x1 - x0 + 1) * (y1 - y0 + 1) : 0;
            live = on;
        }
        PH(2);                                      // bounding box, set-up
        if (ABL(1)) continue;
        const bool big = live && area > P.small_area;
#ifdef RR_RASTER_STATS
        {
            const unsigned long long lm = __ballot(live), bm = __ballot(big);
            int sa_ = (live && !big) ? area : 0, mx = sa_, sm = sa_;
            for (int o = 32; o; o >>= 1) { mx = max(mx, __shfl_xor(mx, o)); sm += __shfl_xor(sm, o); }
            if (lane == 0) {
                RSTAT(2, __popcll(lm)); RSTAT(3, __popcll(bm)); RSTAT(4, sm); RSTAT(5, mx);
                if (lm) RSTAT(6, 1);
                if (lm & ~bm) RSTAT(7, 1);
            }
            if (big) RSTAT(8, area);
        }
#endif
        // Small triangles.  Most cover one or two sample points, a few up to small_area: a per-lane walk of the whole
        // bounding box makes the wave wait for its largest triangle (~20 % lane utilisation).  So every lane walks only
        // the first INLINE_PIX points of its box; the remaining points of all lanes are dealt out evenly: a wave prefix
        // sum numbers them, each lane takes points lane, lane+64, ..., finds the owner with a 6-step binary search over
        // the running ends (64 ints of LDS per wave), fetches the owner's triangle with ds_bpermute and tests its point.
        // Same per-point arithmetic, and ds_min is order independent.
        {
            const bool small = live && !big;
            const int bw = x1 - x0 + 1;
            const int ninl = small ? min(area, INLINE_PIX) : 0;
#if INLINE_PIX == 2
            // (written out: the first point of the box, then its right neighbour -- or the point below it when the box is one
            // column wide; a loop with a per-lane trip count pays for its bookkeeping in every iteration)
            if (!CARRY) {        // (the list-walking kernels: three registers fewer around their item loop)
                if (ninl > 0) raster_pixel_hoisted(s, ia, t, x0, y0, H, TW, row0, vis, tx0i);
                if (ninl > 1) raster_pixel_hoisted(s, ia, t, bw > 1 ? x0 + 1 : x0, bw > 1 ? y0 : y0 + 1, H, TW, row0, vis, tx0i);
            } else {
                // (coordinates as floats and the buffer index carried from the first point to the second: one conversion pair and
                // one integer multiply per triangle instead of per point)
                const float fx0 = (float)x0, fy0 = (float)y0;
                const int vi0 = (H - 1 - y0 - row0) * TW + x0 - tx0i;
                const bool wide = bw > 1;
                if (ninl > 0) raster_pixel_at(s, ia, t, fx0, fy0, vi0, vis);
                if (ninl > 1) raster_pixel_at(s, ia, t, wide ? fx0 + 1.0f : fx0, wide ? fy0 : fy0 + 1.0f, wide ? vi0 + 1 : vi0 - TW, vis);
            }
#else
            int px = x0, py = y0;
            for (int i = 0; i < ninl; i++) {
                raster_pixel_hoisted(s, ia, t, px, py, H, TW, row0, vis, tx0i);
                if (++px > x1) { px = x0; py++; }
            }
#endif
            PH(3);                                  // in-lane sample points
            const int rem = small ? area - ninl : 0;
            int total;
            const int pre = wave_excl_scan(rem, lane, total);
            PH(4);                                  // scan
            if (total > 0) {
                unsigned short *we = wends[tid >> 6];
                we[lane] = (unsigned short)(pre + rem);                                // inclusive ends, non-decreasing over the lanes
                // (box origin relative to the tile, box width and the lane's first point number in one word: one ds_bpermute per round
                // instead of four -- an LDS permute costs the CU six cycles, three times a 4-byte LDS read (tools/ubench/valu_issue.hip).
                // 14 bits of origin: column in tile_xbits, row above it -- a tile has <= 4096 pixels --; 6 of width - 1: a small box is at
                // most 64 wide; 12 of point number: at most 64 x 62 left-over points)
                const int ty0i = H - row0 - rows, xb = RM.tile_xbits;                  // (screen y of the tile's lowest row)
                const int packed = (x0 - tx0i) | ((y0 - ty0i) << xb) | ((bw - 1) << 14) | (pre << 20);
                for (int w0 = 0; w0 < total; w0 += 64) {             // wave-uniform trip count: ds_bpermute needs the owner lanes active
                    const int w = w0 + lane;
                    const bool valid = w < total;
                    int src = 0;                                     // owner = number of lanes whose points end at or before w
#pragma unroll
                    for (int step = 32; step; step >>= 1) if ((int)we[src + step - 1] <= w) src += step;
                    src = valid ? src : 0;
                    STri bs;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        bs.sx[k] = lane_gather(s.sx[k], src); bs.sy[k] = lane_gather(s.sy[k], src); bs.sz[k] = lane_gather(s.sz[k], src);
                        bs.w[k] = 1.0f;
                    }
                    const float bia = lane_gather(ia, src);
                    const int pk = lane_gather_i(packed, src);
                    const int sx0 = (pk & ((1 << xb) - 1)) + tx0i, sy0 = ((pk & 0x3fff) >> xb) + ty0i, sbw = ((pk >> 14) & 63) + 1;
                    const int idx = INLINE_PIX + w - (int)((unsigned)pk >> 20);
                    // row of point idx in a box sbw wide: (idx + 0.5) / sbw lies at least 0.5 / 64 from an integer, far more than the
                    // error of the approximate reciprocal on these small integers (idx < 64 + INLINE_PIX, sbw <= 64)
                    const int ry = (int)(((float)idx + 0.5f) * __builtin_amdgcn_rcpf((float)sbw));
                    if (valid) raster_pixel_hoisted(bs, bia, tb + src, sx0 + idx - ry * sbw, sy0 + ry, H, TW, row0, vis, tx0i);
                }
            }
            PH(5);                                  // redistribution rounds
        }
        unsigned long long todo = ABL(2) ? 0ull : __ballot(big);
        while (todo) {        // wave-cooperative: all 64 lanes rasterise the triangle of lane `src`
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            STri bs;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                bs.sx[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.sx[k]), src));
                bs.sy[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.sy[k]), src));
                bs.sz[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s.sz[k]), src));
                bs.w[k] = 1.0f;
            }
            const float bia = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ia), src));
            const int bt = tb + src;
            const int bx0 = __builtin_amdgcn_readlane(x0, src), by0 = __builtin_amdgcn_readlane(y0, src);
            const int bw = __builtin_amdgcn_readlane(x1, src) - bx0 + 1, bh = __builtin_amdgcn_readlane(y1, src) - by0 + 1;
            const int nbx = (bw + 7) >> 3, nby = (bh + 7) >> 3, nblk = nbx * nby;
            if (nblk <= 4) {
                for (int by = 0; by < bh; by += 8)
                    for (int bx = 0; bx < bw; bx += 8) {
                        const int ox = bx + lx, oy = by + ly;
                        if (ox < bw && oy < bh) raster_pixel_hoisted(bs, bia, bt, bx0 + ox, by0 + oy, H, TW, row0, vis, tx0i);
                    }
                continue;
            }
            // Hierarchical: each lane first classifies one 8x8 block (64 blocks per step) with the conservative corner
            // test; only blocks that may hold covered sample points are rasterised.  Triangles close to the near plane
            // project to slivers whose clipped bounding box is the whole image -- without this they dominate the kernel.
            // delta bounds |float - exact| of a barycentric anywhere in the tile: each edge function is a difference of
            // two products of magnitude <= X*Y (X = max|sx|+W, Y = max|sy|+H), evaluated with <= 4 roundings.
            const float X = fmaxf(fabsf(bs.sx[0]), fmaxf(fabsf(bs.sx[1]), fabsf(bs.sx[2]))) + (float)W;
            const float Y = fmaxf(fabsf(bs.sy[0]), fmaxf(fabsf(bs.sy[1]), fabsf(bs.sy[2]))) + (float)H;
            const float delta2 = 2.0f * (8.0e-6f * X * Y * fabsf(bia) + 1.0e-6f);
            const float inbx = 1.0f / (float)nbx;
            if (lane == 0) { RSTAT(10, 1); }
            for (int c0 = 0; c0 < nblk; c0 += 64) {
                const int bi = c0 + lane;
                bool keep = false;
                if (bi < nblk) {
                    const int byi = (int)(((float)bi + 0.5f) * inbx), bxi = bi - byi * nbx;
                    const int px0 = bx0 + 8 * bxi, py0 = by0 + 8 * byi;
                    keep = block_may_overlap(bs, bia, px0, min(px0 + 7, bx0 + bw - 1), py0, min(py0 + 7, by0 + bh - 1), delta2);
                }
                unsigned long long km = __ballot(keep);
                while (km) {
                    const int j = c0 + __ffsll((long long)km) - 1;
                    km &= km - 1;
                    const int byi = (int)(((float)j + 0.5f) * inbx), bxi = j - byi * nbx;
                    const int ox = 8 * bxi + lx, oy = 8 * byi + ly;
                    if (lane == 0) RSTAT(9, 1);     // 8x8 blocks rasterised by the hierarchical path
                    if (ox < bw && oy < bh) raster_pixel_hoisted(bs, bia, bt, bx0 + ox, by0 + oy, H, TW, row0, vis, tx0i);
                }
            }
        }
        PH(6);                                      // wave-cooperative large triangles
        // triangles that cross the near plane are rare: they are queued and clipped after the window loop
        if (needs_clip && !ABL(16)) { const unsigned qi = atomicAdd(&nclipq, 1u); if (qi < CLIPQ) clipq[qi] = (unsigned short)((k << 6) | lane); }
    }
    PH(7);                                          // last fetch + (below) waiting for the other waves
    // ---- triangles that cross the near plane (queued above): clipped against w = NEAR_W (Sutherland-Hodgman, the oracle's
    // clip_near()) into a triangle or a fan of two, which a whole wave rasterises under the original triangle id
    __syncthreads();
    PH(8);
    WGM(4);
    for (unsigned qi = tid >> 6; qi < min(nclipq, (unsigned)CLIPQ); qi += NT_ / 64) {
        const int bt = t_begin + ((int)wlist[clipq[qi] >> 6] << 6) + (clipq[qi] & 63);
        const int tb = bt & ~63;
        const int inst = D.tri_inst[tb];
        STri ta, tb2;
        int nsub = 0;
        {
            const int cvi = D.tri_vidx[bt];
                const float *cv = D.cluster_verts + (size_t)(tb >> 6) * 192;
            float c0[4], c1[4], c2[4];
            { const int i0 = (cvi & 255) >> 2, i1 = ((cvi >> 8) & 255) >> 2, i2 = ((cvi >> 16) & 255) >> 2;
              clip_vertex(mvp[inst], cv[i0], cv[64 + i0], cv[128 + i0], c0);
              clip_vertex(mvp[inst], cv[i1], cv[64 + i1], cv[128 + i1], c1);
              clip_vertex(mvp[inst], cv[i2], cv[64 + i2], cv[128 + i2], c2); }
            const bool in0 = c0[3] >= NEAR_W, in1 = c1[3] >= NEAR_W, in2 = c2[3] >= NEAR_W;
            float e01[4] = {0, 0, 0, 1}, e12[4] = {0, 0, 0, 1}, e20[4] = {0, 0, 0, 1};
            if (in0 != in1) { if (in0) clip_edge(c0, c1, e01); else clip_edge(c1, c0, e01); }
            if (in1 != in2) { if (in1) clip_edge(c1, c2, e12); else clip_edge(c2, c1, e12); }
            if (in2 != in0) { if (in2) clip_edge(c2, c0, e20); else clip_edge(c0, c2, e20); }
            // polygon in Sutherland-Hodgman order: [v0] [x01] [v1] [x12] [v2] [x20] with the absent ones left out (values
            // are copied, not pointed to: private arrays behind pointers would live in scratch memory)
            float p0[4], p1[4], p2[4], p3[4];
#define CP4(D_, S_) { D_[0] = S_[0]; D_[1] = S_[1]; D_[2] = S_[2]; D_[3] = S_[3]; }
            CP4(p0, c0) CP4(p1, c0) CP4(p2, c0) CP4(p3, c0)
            const int msk = (in0 ? 1 : 0) | (in1 ? 2 : 0) | (in2 ? 4 : 0);       // wave-uniform
            if (msk == 1) { CP4(p0, c0) CP4(p1, e01) CP4(p2, e20) nsub = 1; }
            else if (msk == 2) { CP4(p0, e01) CP4(p1, c1) CP4(p2, e12) nsub = 1; }
            else if (msk == 4) { CP4(p0, e12) CP4(p1, c2) CP4(p2, e20) nsub = 1; }
            else if (msk == 3) { CP4(p0, c0) CP4(p1, c1) CP4(p2, e12) CP4(p3, e20) nsub = 2; }
            else if (msk == 6) { CP4(p0, e01) CP4(p1, c1) CP4(p2, c2) CP4(p3, e20) nsub = 2; }
            else if (msk == 5) { CP4(p0, c0) CP4(p1, e01) CP4(p2, e12) CP4(p3, c2) nsub = 2; }
#undef CP4
            clip_to_screen(p0, W, H, ta.sx[0], ta.sy[0], ta.sz[0]);          // fan (0, 1, 2), (0, 2, 3)
            clip_to_screen(p1, W, H, ta.sx[1], ta.sy[1], ta.sz[1]);
            clip_to_screen(p2, W, H, ta.sx[2], ta.sy[2], ta.sz[2]);
            tb2.sx[0] = ta.sx[0]; tb2.sy[0] = ta.sy[0]; tb2.sz[0] = ta.sz[0];
            tb2.sx[1] = ta.sx[2]; tb2.sy[1] = ta.sy[2]; tb2.sz[1] = ta.sz[2];
            clip_to_screen(p3, W, H, tb2.sx[2], tb2.sy[2], tb2.sz[2]);
            ta.w[0] = ta.w[1] = ta.w[2] = tb2.w[0] = tb2.w[1] = tb2.w[2] = 1.0f;
        }
        for (int sub = 0; sub < nsub; sub++) {
            STri bs;
#pragma unroll
            for (int k = 0; k < 3; k++) { bs.sx[k] = sub ? tb2.sx[k] : ta.sx[k]; bs.sy[k] = sub ? tb2.sy[k] : ta.sy[k]; bs.sz[k] = sub ? tb2.sz[k] : ta.sz[k]; bs.w[k] = 1.0f; }
            // clipped bounding box and reciprocal area: the same arithmetic as the per-lane set-up above
            const float xmin = fminf(bs.sx[0], fminf(bs.sx[1], bs.sx[2])), xmax = fmaxf(bs.sx[0], fmaxf(bs.sx[1], bs.sx[2]));
            const float ymin = fminf(bs.sy[0], fminf(bs.sy[1], bs.sy[2])), ymax = fmaxf(bs.sy[0], fmaxf(bs.sy[1], bs.sy[2]));
            if (xmax < txlo || ymax < ty0 || xmin > txhi || ymin > ty1) continue;
            const int bx0 = (int)ceilf(fmaxf(xmin, txlo)), bx1 = (int)floorf(fminf(xmax, txhi));
            const int by0 = (int)ceilf(fmaxf(ymin, ty0)), by1 = (int)floorf(fminf(ymax, ty1));
            if (bx1 < bx0 || by1 < by0) continue;
            const TriEdge te = tri_edge(bs);
            if (!te.ok) continue;
            const float bia = te.ia;
            const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
            const int nbx = (bw + 7) >> 3, nby = (bh + 7) >> 3, nblk = nbx * nby;
            if (nblk <= 4) {
                for (int by = 0; by < bh; by += 8)
                    for (int bx = 0; bx < bw; bx += 8) {
                        const int ox = bx + lx, oy = by + ly;
                        if (ox < bw && oy < bh) raster_pixel_hoisted(bs, bia, bt, bx0 + ox, by0 + oy, H, TW, row0, vis, tx0i);
                    }
                continue;
            }
            // Hierarchical: each lane first classifies one 8x8 block (64 blocks per step) with the conservative corner
            // test; only blocks that may hold covered sample points are rasterised.  Triangles close to the near plane
            // project to slivers whose clipped bounding box is the whole image -- without this they dominate the kernel.
            // delta bounds |float - exact| of a barycentric anywhere in the tile: each edge function is a difference of
            // two products of magnitude <= X*Y (X = max|sx|+W, Y = max|sy|+H), evaluated with <= 4 roundings.
            const float X = fmaxf(fabsf(bs.sx[0]), fmaxf(fabsf(bs.sx[1]), fabsf(bs.sx[2]))) + (float)W;
            const float Y = fmaxf(fabsf(bs.sy[0]), fmaxf(fabsf(bs.sy[1]), fabsf(bs.sy[2]))) + (float)H;
            const float delta2 = 2.0f * (8.0e-6f * X * Y * fabsf(bia) + 1.0e-6f);
            const float inbx = 1.0f / (float)nbx;
            if (lane == 0) { RSTAT(10, 1); }
            for (int c0 = 0; c0 < nblk; c0 += 64) {
                const int bi = c0 + lane;
                bool keep = false;
                if (bi < nblk) {
                    const int byi = (int)(((float)bi + 0.5f) * inbx), bxi = bi - byi * nbx;
                    const int px0 = bx0 + 8 * bxi, py0 = by0 + 8 * byi;
                    keep = block_may_overlap(bs, bia, px0, min(px0 + 7, bx0 + bw - 1), py0, min(py0 + 7, by0 + bh - 1), delta2);
                }
                unsigned long long km = __ballot(keep);
                while (km) {
                    const int j = c0 + __ffsll((long long)km) - 1;
                    km &= km - 1;
                    const int byi = (int)(((float)j + 0.5f) * inbx), bxi = j - byi * nbx;
                    const int ox = 8 * bxi + lx, oy = 8 * byi + ly;
                    if (lane == 0) RSTAT(9, 1);     // 8x8 blocks rasterised by the hierarchical path
                    if (ox < bw && oy < bh) raster_pixel_hoisted(bs, bia, bt, bx0 + ox, by0 + oy, H, TW, row0, vis, tx0i);
                }
            }
        }
    }
    PH(9);                                          // near-plane pass
    PH_FLUSH;
    WGM(5);
#ifdef RR_RASTER_STATS
    const unsigned long long t_exit_ = __builtin_readcyclecounter();
#endif
    __syncthreads();
#ifdef RR_RASTER_STATS
    if (lane == 0 && (P.ablate & 0x2000)) {   // RR_ABLATE=8192: only these two (idle at the barrier / busy in the loop)
        atomicAdd(&g_rstats[14], (unsigned long long)(__builtin_readcyclecounter() - t_exit_)); atomicAdd(&g_rstats[15], (unsigned long long)(t_exit_ - t_loop0_));
    }
#endif
    if (pass == 1) {   // publish the static layer's keys
        unsigned long long *sv = D.static_vis_out + (size_t)row0 * W + tx0i;
        for (int i = tid; i < npix; i += NT_) { const int lr = i / TW, lx = i - lr * TW; if (lx < cols) sv[(size_t)lr * W + lx] = vis[i]; }
    }
    // ---- compaction: pixels owned by a triangle rasterised in this pass go to the fragment list of this (env, tile)
    // (the comparison with the static layer's key is left to k_shade: a dependent global read at the tail of this
    // LDS-limited workgroup is exposed latency, in the high-occupancy shading kernel it is not)
    uint2 *lst = D.frag_list + ((size_t)env * RM.ntiles + tile) * TILE_PIX;
#ifdef RR_RASTER_STATS
    __shared__ unsigned win_bits[MAXWIN / 32];      // clusters that own a pixel of the tile at the end
    for (int i = tid; i < MAXWIN / 32; i += NT_) win_bits[i] = 0;
    __syncthreads();
    for (int i = tid; i < npix; i += NT_) {
        const unsigned long long key = vis[i];
        if (key != ~0ull && key != VIS_WAS_DYNAMIC) { const unsigned w_ = ((unsigned)(key & 0xffffffffu) - (unsigned)t_begin) >> 6; atomicOr(&win_bits[w_ >> 5], 1u << (w_ & 31)); }
    }
    __syncthreads();
    if (tid < MAXWIN / 32) RSTAT(11, __popc(win_bits[tid]));      // (slot 11 reused: clusters with at least one winning pixel)
#endif
    // (one LDS atomic per wave and trip -- ballot, popcount, v_mbcnt -- instead of one per listed pixel: ~300 same-address atomics
    // per tile serialise in the LDS; the order of the list is immaterial)
    for (int i0 = 0; i0 < npix; i0 += NT_) {
        const int i = i0 + tid;
        const unsigned long long key = i < npix ? vis[i] : ~0ull;
        const unsigned tri = key == VIS_WAS_DYNAMIC ? FRAG_VACATED : (unsigned)(key & 0xffffffffu);
        const bool has = key != ~0ull;
        const unsigned long long hm = __ballot(has);
        if (hm == 0ull) continue;                       // (wave-uniform)
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&nlist, (unsigned)__popcll(hm));
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (has) {
            const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
            lst[slot] = make_uint2((unsigned)(key >> 32), ((unsigned)i << 18) | tri);
        }
    }
    __syncthreads();
    if (tid == 0 && nclipq > CLIPQ) atomicOr(&D.errflags[env], 8u);      // (clip queue overflow: triangles were dropped; a RENDER status -- the solve's dead mask ignores bit 8)
    WGM(6);
    if (PRE && pass == 0 && tid == 0 && D.item_cost) D.item_cost[(size_t)env * RM.ntiles + tile] = (unsigned)__builtin_amdgcn_s_memrealtime() - s_tc0 + 1u;      // (100 MHz ticks)
    if (tid == 0) { D.frag_count[(size_t)env * RM.ntiles + tile] = nlist; RSTAT(12, nlist); RSTAT(13, 1); RSTAT(14, nclipq); RSTAT(15, nclipq > 0 ? 1 : 0); }
}

// One workgroup per (env, tile); sel (env_selected): all envs, or only those of light solver groups.
// Grid (envs, tiles) -- or, with D.item_perm (pass 0), one dimension of envs * tiles workgroups that take the items in the order of
// raster_order_class: the hardware deals workgroups out by their linear index -- XCD = index mod 8, shader engine = (index / 8) mod 4,
// strictly in turn: a shader engine whose four slots per CU are taken holds up its XCD's queue -- so the index order decides how evenly
// the 32 engines are loaded (scratch/rwgtime.py, scratch/rorder.py: env order 0.708 of the slots busy, cost order 0.842).
__global__ void __launch_bounds__(RASTER_THREADS) RASTER_ATTR k_raster(SimParams P, const RenderModel *RMp, DevPtrs D, int n_inst_used, int pass, int env0, int restore, int sel) {
    int env, tile;
    if (pass == 0 && D.item_perm) { const unsigned it = D.item_perm[blockIdx.x]; if (it == 0xffffffffu) return; env = (int)(it >> 8); tile = (int)(it & 255u); }
    else { env = blockIdx.x + env0; tile = blockIdx.y; }
#ifdef RR_RASTER_STATS
    { const RenderModel &RM = *RMp; WGM(0); }
#endif
    // (the two selection flags and this thread's words of the previous frame's list in ONE round trip, then the branches)
    const size_t item = (size_t)env * RMp->ntiles + tile;
    const bool flagged = pass == 0 && D.render_flags != nullptr;
    const unsigned char rflag = (flagged ? D.render_flags : (const unsigned char *)D.hgflag)[env];
    const int cls = D.hgflag[env];
    const unsigned n_old_pre = D.frag_count[item];
    const unsigned en_first_pre = D.frag_list[item * TILE_PIX + threadIdx.x].y;
    // (one branch on all four: none of the loads can be moved behind it.  The last term is never true -- N > 0 --, whatever the
    // two words hold)
    const int selected = (int)(sel == 0) | ((int)(sel == 1) & (int)(cls == 0)) | ((int)(sel == 2) & (int)(cls == 1)) | ((int)(sel == 3) & (int)(cls == 2));
    if (((int)flagged & (int)(rflag == 0)) | (selected ^ 1) | ((int)(P.N < 0) & ((int)(n_old_pre > 0x7fffffffu) | (int)((en_first_pre >> 30) == 2u)))) return;
#ifdef RR_RASTER_STATS
    unsigned long long wt0_ = 0;
    if (P.ablate & 0x4000) wt0_ = __builtin_amdgcn_s_memrealtime();
    { const RenderModel &RM = *RMp; WGM(1); }
#endif
    raster_tile<RASTER_THREADS, true, false, true>(P, *RMp, D, n_inst_used, pass, env, tile, restore, n_old_pre, en_first_pre);
#ifdef RR_RASTER_STATS
    if ((P.ablate & 0x4000) && threadIdx.x == 0) {
        unsigned long long *g = g_rwg_time[(env * RMp->ntiles + tile) & 65535];
        g[0] = wt0_; g[1] = __builtin_amdgcn_s_memrealtime();
        g[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32);
    }
#endif
}

// Dispatch order of the next frame's k_raster: the (env, tile) items by falling cost (counting sort over 1024 linear bins of the
// durations k_raster has just measured), by a few extra workgroups of the k_shade launch that follows it (a stream of its own was
// measured: a fifth stream shares a hardware queue with one of the step's four and serialises it, 0.67 -> 0.83 ms).  One class per
// XCD -- workgroup index mod 8 is the XCD --: the envs = x (mod 8), so every XCD keeps an eighth of the envs with all their tiles
// and its queue holds items of falling cost.  (Which XCD rasterises a tile does not matter to k_shade: with the tiles of an env
// dealt to different XCDs it takes 0.089 instead of 0.088 ms.)  perm[8 * j + x] = the j-th costliest item of class x,
// env << 8 | tile, or ~0 behind the last one.
// Costs change little from frame to frame (5 ms of motion); an order that is off only loads the shader engines less evenly, the
// images do not depend on it.
#define ORDER_BINS 1024
template <int NT_>
__device__ __forceinline__ void raster_order_class(int x, int N, int ntiles, int per_class, const unsigned *cost, unsigned *bins, unsigned *perm) {
    static_assert(ORDER_BINS % NT_ == 0 && NT_ % 64 == 0 && NT_ <= 1024, "bins per thread");
    constexpr int BPT = ORDER_BINS / NT_;
    __shared__ unsigned hist[ORDER_BINS];
    __shared__ unsigned s_max, wtot[NT_ / 64];
    const int tid = threadIdx.x;
    const int n_items = x < N ? ntiles * ((N - x + 7) >> 3) : 0;          // items of this class: envs x, x + 8, ...
    __syncthreads();                                                       // (a workgroup may take several classes in turn)
    if (tid == 0) s_max = 1u;
    for (int b = tid; b < ORDER_BINS; b += NT_) hist[b] = 0u;
    __syncthreads();
#define ORDER_ITEM(k) ((size_t)(x + 8 * ((k) / ntiles)) * ntiles + (size_t)((k) % ntiles))
#define ORDER_BIN(c) (ORDER_BINS - 1 - min((unsigned)((float)(c) * scale), (unsigned)(ORDER_BINS - 1)))      /* bin 0 = the costliest items */
    unsigned mx = 0u;
    for (int k = tid; k < n_items; k += NT_) mx = max(mx, cost[ORDER_ITEM(k)]);
    for (int o = 32; o; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    if ((tid & 63) == 0) atomicMax(&s_max, mx);
    __syncthreads();
    const float scale = (float)(ORDER_BINS - 1) / (float)s_max;
    // (an item's bin is computed ONCE and kept: were a cost rewritten between the two passes -- a visibility pass running beside
    // this launch, which the launch order rules out today -- the scatter would still be a permutation)
    for (int k = tid; k < n_items; k += NT_) { const size_t i = ORDER_ITEM(k); const unsigned b = ORDER_BIN(cost[i]); bins[i] = b; atomicAdd(&hist[b], 1u); }
    __syncthreads();
    // exclusive prefix over the bins: BPT consecutive bins per thread, wave scan of the threads' sums, the waves' totals
    unsigned hb[BPT], sum = 0u;
#pragma unroll
    for (int j = 0; j < BPT; j++) { hb[j] = hist[tid * BPT + j]; sum += hb[j]; }
    unsigned inc = sum;
    for (int o = 1; o < 64; o <<= 1) { const unsigned v = (unsigned)__shfl_up((int)inc, o); if ((tid & 63) >= o) inc += v; }
    if ((tid & 63) == 63) wtot[tid >> 6] = inc;
    __syncthreads();
    unsigned base = inc - sum;
    for (int w = 0; w < (tid >> 6); w++) base += wtot[w];
#pragma unroll
    for (int j = 0; j < BPT; j++) { hist[tid * BPT + j] = base; base += hb[j]; }
    __syncthreads();
    for (int k = tid; k < n_items; k += NT_) {
        const size_t i = ORDER_ITEM(k);
        const unsigned pos = atomicAdd(&hist[min(bins[i], (unsigned)(ORDER_BINS - 1))], 1u);
        if (pos < (unsigned)n_items) perm[(size_t)8 * pos + x] = ((unsigned)(i / ntiles) << 8) | (unsigned)(i % ntiles);
    }
    for (int k = n_items + tid; k < per_class; k += NT_) perm[(size_t)8 * k + x] = 0xffffffffu;
#undef ORDER_ITEM
#undef ORDER_BIN
}

#ifndef LIST_CARRY
#define LIST_CARRY false
#endif
#ifndef LIST_WAVES
#define LIST_WAVES 6         // (79 VGPRs: three workgroups per CU -- since nothing derived from the thread index stays live across the items of the loop)
#endif
#define RASTER_LIST_WGS 768      // three per CU: the item loop around the tile needs more than the 64 VGPRs of four (a long list of heavy
                                 // envs must not be rendered at a fraction of the occupancy)
// The heavy envs (D.hlist, D.hcount -- known on the device only): a fixed number of workgroups walk the
// list, so that no LDS-filling workgroup is launched just to find that its env is not on it.
__global__ void __launch_bounds__(RASTER_THREADS) __attribute__((amdgpu_waves_per_eu(LIST_WAVES, 8))) k_raster_list(SimParams P, const RenderModel *RMp, DevPtrs D, int n_inst_used, int restore, int which) {
    const RenderModel &RM = *RMp;
    int *hcount = which ? D.hcount2 : D.hcount;
    const int *hlist = which ? D.hlist2 : D.hlist;
    const int nitems = hcount[0] * RM.ntiles;
    __shared__ int s_item;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&hcount[1], 1);      // dynamic assignment: tiles differ a lot in cost
        __syncthreads();
        const int it = s_item;
        if (it >= nitems) break;
        const int tile = it % RM.ntiles, ge = it / RM.ntiles;
        const int env = hlist[ge];
        if (env < P.N && !(D.render_flags && !D.render_flags[env])) raster_tile<RASTER_THREADS, LIST_CARRY, true>(P, RM, D, n_inst_used, 0, env, tile, restore);
        __syncthreads();        // the LDS of the tile is reused
    }
}

// Image targets of k_static_copy / k_shade: the per-env observation buffers (pass 0) or the shared static layer (pass 1).

// Copies the static layer (or the background when there is none) into the images of every rendered env: 4 pixels per
// thread (W % 4 == 0 enforced at create).  Pure streaming: reads hit L2, writes are the obs bytes of SURVEY 8(d).
#define COPY_THREADS 256
__global__ void __launch_bounds__(COPY_THREADS) k_static_copy(const RenderModel *RMp, DevPtrs D, ImageOut out, int use_flags, int N) {
    const int ngroups = (RMp->W * RMp->H) >> 2;
    for (int env = blockIdx.y; env < N; env += gridDim.y) {
    if (use_flags && D.render_flags && !D.render_flags[env]) continue;
    const size_t base = (size_t)env * out.env_stride;
    for (int g = blockIdx.x * COPY_THREADS + threadIdx.x; g < ngroups; g += gridDim.x * COPY_THREADS) {
        const unsigned *srgb = (const unsigned *)(D.static_rgb) + (size_t)3 * g;
        const unsigned r0 = srgb[0], r1 = srgb[1], r2 = srgb[2];
        const float4 dv = *(const float4 *)(D.static_depth + (size_t)4 * g);
        unsigned *rgbp = (unsigned *)(out.rgb + (base + (size_t)4 * g) * 3);
        rgbp[0] = r0; rgbp[1] = r1; rgbp[2] = r2;
        *(float4 *)(out.depth + base + (size_t)4 * g) = dv;
        if (out.mask) *(int4 *)(out.mask + base + (size_t)4 * g) = *(const int4 *)(D.static_mask + (size_t)4 * g);
    }
    }
}

// Background fill of the shared static images (before the static layer is shaded, and when there is no static layer).
// Incremental image update: an env's image in HBM still holds its previous frame, and only the pixels of that frame's
// fragment list differ from the static layer.  Putting the static values back at exactly those pixels (before k_raster
// overwrites the list) leaves the same image as a full copy of the static layer -- ~1 000 pixels instead of 16 384 per env.
#define RESTORE_THREADS 256
__global__ void __launch_bounds__(RESTORE_THREADS) k_restore(const RenderModel *RMp, DevPtrs D, ImageOut out, int use_flags) {
    const RenderModel &RM = *RMp;
    const int env = blockIdx.x, tile = blockIdx.y;
    if (use_flags && D.render_flags && !D.render_flags[env]) return;
    const unsigned n = D.frag_count[(size_t)env * RM.ntiles + tile];          // still the previous frame's count
    const uint2 *lst = D.frag_list + ((size_t)env * RM.ntiles + tile) * TILE_PIX;
    const int tyi_ = tile / RM.ntx, row0_ = tyi_ * RM.tile_h, tx0_ = (tile - tyi_ * RM.ntx) * RM.tile_w;
    const size_t ebase = (size_t)env * out.env_stride;
    // (pixel-in-tile index -> pixel of the image)
#define RESTORE_GP(PI) ((size_t)(row0_ + (int)__umulhi((unsigned)(PI), RM.w_magic)) * RM.W + (size_t)(tx0_ + (int)(PI) - (int)__umulhi((unsigned)(PI), RM.w_magic) * RM.tile_w))
    // four fragments per thread and trip, every load of the trip issued before the first store (the kernel is a chain of
    // dependent round trips: list entry -> static pixel -> store)
    for (unsigned i0 = 0; i0 < n; i0 += 4 * RESTORE_THREADS) {
        unsigned pi[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const unsigned i = i0 + k * RESTORE_THREADS + threadIdx.x; pi[k] = i < n ? lst[i].y >> 18 : 0xffffffffu; }
        unsigned char r[4][3]; float d[4]; int m[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t so = RESTORE_GP(pi[k] != 0xffffffffu ? pi[k] : 0u);
            r[k][0] = D.static_rgb[so * 3]; r[k][1] = D.static_rgb[so * 3 + 1]; r[k][2] = D.static_rgb[so * 3 + 2];
            d[k] = D.static_depth[so];
            m[k] = out.mask ? D.static_mask[so] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (pi[k] == 0xffffffffu) continue;
            const size_t o = ebase + RESTORE_GP(pi[k]);
            out.rgb[o * 3] = r[k][0]; out.rgb[o * 3 + 1] = r[k][1]; out.rgb[o * 3 + 2] = r[k][2];
            out.depth[o] = d[k];
            if (out.mask) out.mask[o] = m[k];
        }
    }
#undef RESTORE_GP
}

__global__ void k_background(const RenderModel *RMp, DevPtrs D) {
    const int npx = RMp->W * RMp->H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += gridDim.x * blockDim.x) {
        D.static_rgb[3 * i] = 255; D.static_rgb[3 * i + 1] = 255; D.static_rgb[3 * i + 2] = 255;
        D.static_depth[i] = 1.0f; D.static_mask[i] = -1;
    }
}

// Deferred shading of the fragment lists, one lane per listed pixel.  Every (env, tile) list is dealt out in
// SHADE_THREADS-entry chunks to SHADE_SPLIT workgroups (blocks whose first chunk lies beyond the list exit at once).
// A/B (k_shade, ms): 64x16 0.175, 128x8 0.145, 256x8 0.131, 256x2 0.122, 512x2 0.119, 1024x1 0.120 -- the per-block
// staging of the instance constants outweighs the tail of long lists.
#ifndef SHADE_THREADS
#define SHADE_THREADS 256
#endif
#ifndef SHADE_SPLIT
#define SHADE_SPLIT 2       // with four tiles per env a list holds ~300 entries: 256 x 8 0.113 ms, x 4 0.101, x 2 0.098, 512 x 1 0.106, 128 x 3 0.100
#endif
// chunk z of nz of the fragment list of (env, tile); mvp / sinst: the workgroup's staging arrays
// Deferred shading of chunk z of nz of the fragment list of (env, tile) by the calling workgroup.
template <int NTHREADS>
__device__ __forceinline__ void shade_block(const RenderModel &RM, const DevPtrs &D, const ImageOut &out, int env, int tile, int z, int nz,
                                            float (*mvp)[16], float (*sinst)[16]) {
    const unsigned n = D.frag_count[(size_t)env * RM.ntiles + tile];
    if ((unsigned)z * NTHREADS >= n) return;                    // (workgroup-uniform)
    stage_instances(RM, D, env, threadIdx.x, NTHREADS, mvp, sinst);
    __syncthreads();
#if defined(RR_SHADE_PROBE) && (RR_SHADE_PROBE & 8)
    if (mvp[0][0] != 12345.678f) return;           // (probe: the prologue only -- count, instance constants, barrier)
#endif
    ShadeCtx ctx;
    ctx.D = &D; ctx.mvp = &mvp[0][0]; ctx.sinst = &sinst[0][0]; ctx.W = RM.W; ctx.H = RM.H;
    const uint2 *lst = D.frag_list + ((size_t)env * RM.ntiles + tile) * TILE_PIX;
    const int tyi = tile / RM.ntx, row0 = tyi * RM.tile_h, tx0i = (tile - tyi * RM.ntx) * RM.tile_w;
    const size_t ebase = (size_t)env * out.env_stride;
    const unsigned long long *sv = D.static_vis;                 // null while the static layer itself is built
    for (unsigned i = z * NTHREADS + threadIdx.x; i < n; i += nz * NTHREADS) {
        const uint2 f = lst[i];
        const int pi = (int)(f.y >> 18), t = (int)(f.y & 0x3ffffu);
        // pixel-in-tile index -> row and column of the image (tile_w pixels per buffer row; multiply-high instead of a division)
        const int lrow = (int)__umulhi((unsigned)pi, RM.w_magic), px = tx0i + pi - lrow * RM.tile_w;
        const size_t gp = (size_t)(row0 + lrow) * RM.W + (size_t)px;        // pixel of the image / of the static layer
        // a moving triangle only shows where it beats the static layer (depth, then triangle id; static ids are lower);
        // where it does not, and where the previous frame's fragment has gone, the pixel goes back to the static layer
        // (the image persists in HBM from frame to frame, do_render)
        if (t == (int)FRAG_VACATED || (sv && !((((unsigned long long)f.x << 32) | (unsigned)t) < sv[gp]))) {
            // (vacated entries only exist in env frames; with RR_NO_STATIC_LAYER the static buffers hold the background)
            const size_t so = gp, o = ebase + gp;
            out.rgb[o * 3] = D.static_rgb[so * 3]; out.rgb[o * 3 + 1] = D.static_rgb[so * 3 + 1]; out.rgb[o * 3 + 2] = D.static_rgb[so * 3 + 2];
            out.depth[o] = D.static_depth[so];
            if (out.mask) out.mask[o] = D.static_mask[so];
            continue;
        }
        unsigned char c3[3]; int m;
#if defined(RR_SHADE_PROBE) && (RR_SHADE_PROBE & 4)
        shade_pixel(ctx, load_tri_rec(D, t & 63), px, row0 + lrow, c3, m);      // (probe: 64 records for everybody -- one cache line set)
#else
        shade_pixel(ctx, load_tri_rec(D, t), px, row0 + lrow, c3, m);
#endif
        const size_t o = ebase + gp;
#if defined(RR_SHADE_PROBE) && (RR_SHADE_PROBE & 1)
        if (c3[0] == 1 && c3[1] == 2 && c3[2] == 3 && m == 12345) out.depth[o] = 0.0f;      // (probe: no image stores)
#else
        out.rgb[o * 3] = c3[0]; out.rgb[o * 3 + 1] = c3[1]; out.rgb[o * 3 + 2] = c3[2];
        out.depth[o] = __uint_as_float(f.x);
        if (out.mask) out.mask[o] = m;
#endif
    }
}
// Measured in round 4 and dropped (k_shade alone, 4096 envs, 0.103 ms as it stands): the first list entry requested together with the
// count and the triangle record together with the instance constants (three dependent round trips instead of five): 0.104 ms;
// cooperative record loads -- eight lanes per 128-byte record, one load instruction covering eight records in eight cache
// lines instead of 64, transposed through 14 KB of LDS: 0.101 ms; one 512-thread workgroup per env walking the concatenation
// of its tile lists (4 096 workgroups instead of 32 768, instance constants staged once per env): 0.120 ms.
// order_n > 0: the launch has one more column of workgroups (blockIdx.x == order_n = the number of envs), which make the dispatch
// order of the next frame's k_raster from the costs the last one has left (raster_order_class, eight classes).
__global__ void __launch_bounds__(SHADE_THREADS) k_shade(const RenderModel *RMp, DevPtrs D, ImageOut out, int use_flags, int env0, int sel, int order_n, unsigned *order_perm) {
    __shared__ __attribute__((aligned(16))) float mvp[MAXINST][16];
    __shared__ __attribute__((aligned(16))) float sinst[MAXINST][16];
    if (order_n > 0 && (int)blockIdx.x == order_n) {
        const int nt = RMp->ntiles, per_class = ((order_n + 7) >> 3) * nt;
        for (int x = blockIdx.z * gridDim.y + blockIdx.y; x < 8; x += gridDim.y * gridDim.z) raster_order_class<SHADE_THREADS>(x, order_n, nt, per_class, D.item_cost, D.item_bin, order_perm);
        return;
    }
    const int env = blockIdx.x + env0, tile = blockIdx.y;
    if (use_flags && D.render_flags && !D.render_flags[env]) return;
    if (!env_selected(D.hgflag, env, sel)) return;
#ifdef RR_RASTER_STATS
    unsigned long long wt0_ = 0;      // (workgroup timeline of the shading: rr_debug_shade_ablate(0x10000), scratch/rwgtime.py shade)
    if (g_shade_ablate & 0x10000) wt0_ = __builtin_amdgcn_s_memrealtime();
#endif
    shade_block<SHADE_THREADS>(*RMp, D, out, env, tile, blockIdx.z, gridDim.z, mvp, sinst);
#ifdef RR_RASTER_STATS
    if ((g_shade_ablate & 0x10000) && threadIdx.x == 0) {
        unsigned long long *g = g_rwg_time[((env * RMp->ntiles + tile) * gridDim.z + blockIdx.z) & 65535];
        g[0] = wt0_; g[1] = __builtin_amdgcn_s_memrealtime();
        g[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32);
    }
#endif
}
// (Visibility pass and shading of a tile in one workgroup -- k_render_list's body for every env, 64 VGPRs, same LDS -- was
// measured: 0.470 ms instead of 0.383 + 0.097 alone, but 0.739 instead of 0.725 ms for the step: the heavy lists' render
// only gets onto the machine when this launch drains, 100 us later than behind the plain visibility pass.)
// The render of the heavy envs (D.hlist, D.hcount -- known on the device only), which follows their solve on the side stream
// and is the tail of the step's longest chain: ONE launch instead of three.  A fixed number of workgroups walk the list of
// (env, tile) items (dynamic assignment: tiles differ a lot in cost; no LDS-filling workgroup is launched just to find that
// its env is not on the list); for each item: the env's instance matrices (22 threads; the four tiles of an env write the
// same values), the visibility pass of the tile, the shading of its fragment list.
#define RENDER_LIST_WGS 768      // (two resident per CU: set-up, visibility and shading in one body need 128 VGPRs -- capped at 80 it spilled)
template <int NT_>
__device__ __forceinline__ void render_list_body(const BodyParams &B, const SimParams &P, const RenderModel *RMp, const DevPtrs &D, const ImageOut &out, int n_inst_used, int restore, int which) {
    const RenderModel &RM = *RMp;
    int *hcount = which ? D.hcount2 : D.hcount;
    const int *hlist = which ? D.hlist2 : D.hlist;
    static_assert(TILE_PIX * 8 >= 2 * MAXINST * 16 * 4, "the shading constants are staged in the visibility buffer");
    float (*smvp)[16] = (float (*)[16])g_vis;                      // (the buffer is dead once raster_tile has written the list)
    float (*sinst)[16] = (float (*)[16])((float *)g_vis + MAXINST * 16);
    const int nitems = hcount[0] * RM.ntiles;
    __shared__ int s_item;
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&hcount[1], 1);
        __syncthreads();
        const int it = s_item;
        if (it >= nitems) break;
        const int tile = it % RM.ntiles, env = hlist[it / RM.ntiles];
        if (env < P.N && !(D.render_flags && !D.render_flags[env])) {
            if ((int)threadIdx.x < RM.ni) instance_setup(B, P, RM, D, env, threadIdx.x);
            __threadfence_block();
            __syncthreads();
            raster_tile<NT_, true, true>(P, RM, D, n_inst_used, 0, env, tile, restore);
            __threadfence_block();      // the fragment list and its count, written by this workgroup, are read back below
            __syncthreads();
            shade_block<NT_>(RM, D, out, env, tile, 0, 1, smvp, sinst);
        }
        __syncthreads();        // the LDS of the tile and the staging arrays are reused
    }
}
#ifndef RENDER_LIST_WAVES
#define RENDER_LIST_WAVES 4
#endif
__global__ void __launch_bounds__(RASTER_THREADS) __attribute__((amdgpu_waves_per_eu(RENDER_LIST_WAVES, 8)))
k_render_list(BodyParams B, SimParams P, const RenderModel *RMp, DevPtrs D, ImageOut out, int n_inst_used, int restore, int which) {
    render_list_body<RASTER_THREADS>(B, P, RMp, D, out, n_inst_used, restore, which);
}
// (The same with 256 threads -- the footprint of one raster workgroup, so that it moves into the hole a retiring raster
// workgroup leaves -- was measured: 0.762 ms with the heavy list, 0.788 with both, instead of 0.727: an item takes twice as
// long and the visibility pass of the light envs loses as much as the lists gain.)

// ---------------------------------------------------------------------------------------------- host side
struct BlobEntry {
    char name[32];
    uint32_t dtype, ndim, shape[4];
    uint64_t offset, nbytes;
};

struct Blob {
    const char *base; size_t size; uint32_t n; const BlobEntry *e;
    bool init(const void *p, size_t sz) {
        base = (const char *)p; size = sz;
        if (sz < 16 || memcmp(base, "RRMODEL1", 8) != 0) return false;
        memcpy(&n, base + 8, 4);
        if (16 + (size_t)n * sizeof(BlobEntry) > sz) return false;
        e = (const BlobEntry *)(base + 16);
        for (uint32_t i = 0; i < n; i++) if (e[i].offset + e[i].nbytes > sz) return false;
        return true;
    }
    const BlobEntry *find(const char *name, uint32_t dtype) const {
        for (uint32_t i = 0; i < n; i++) if (strncmp(e[i].name, name, 32) == 0 && e[i].dtype == dtype) return &e[i];
        return nullptr;
    }
    const float *f32(const char *name, size_t min_count = 0) const {
        const BlobEntry *x = find(name, 0);
        if (!x || x->nbytes < min_count * 4) return nullptr;
        return (const float *)(base + x->offset);
    }
    const int32_t *i32(const char *name, size_t min_count = 0) const {
        const BlobEntry *x = find(name, 1);
        if (!x || x->nbytes < min_count * 4) return nullptr;
        return (const int32_t *)(base + x->offset);
    }
    const uint8_t *u8(const char *name, size_t *nbytes) const {
        const BlobEntry *x = find(name, 2);
        if (!x) return nullptr;
        *nbytes = x->nbytes;
        return (const uint8_t *)(base + x->offset);
    }
};

struct rr_env {
    rr_config cfg;
    BodyParams B;
    SimParams P;
    RenderModel RM;
    RenderModel *RM_dev;
    DevPtrs D;
    hipStream_t stream;
    int epb;                 // envs per block for physics kernels
    int n_inst_used;
    size_t field_bytes[RR_F_COUNT];
    void *field_ptr[RR_F_COUNT];
    float *state_aos;        // [N][61] staging for RR_F_STATE
    unsigned char *mask_dev; // [N]
    float *link_out;         // [N][nl][7]
    IkModel IK;
    float *plan; int *plan_step; float *ik_in; float *ik_out; float *ik_err;   // lazily allocated (macro / cartesian adapters)
    float *score_out; unsigned char *score_mask;                               // lazily allocated (rr_evaluate_goals)
    std::vector<void *> allocs;
    bool timing;
    bool full_copy, sep_restore;   // RR_FULL_COPY / RR_SEPARATE_RESTORE at create: the two earlier image-update schemes (tests, A/B)
    int split_max_pct;             // the heavy / light split is used while at most this share of the solver groups is heavy (RR_SPLIT_MAX_PCT)
    int *h_hcount;                 // pinned host copy of D.hcount[0] (device-mapped: written by k_prep_a of the following step)
    bool light_ow;                 // k_solve_light_ow available and wanted (RR_NO_OBJECT_WAVE=1: k_solve_light; A/B, tests)
    bool split_heavy;              // heavy solver groups + their render on the side stream (RR_NO_SPLIT=1 turns it off: A/B, tests)
    bool images_valid;       // every env's image holds its previous frame (static layer + the pixels of its fragment list)
    hipEvent_t ev[2 * RR_NUM_KERNELS];
    hipStream_t aux;         // side stream: the HBM-bound static-layer copy runs beside the VALU-bound physics / visibility kernels
    hipEvent_t ev_fork, ev_join, ev_dyn, ev_join2, ev_vsolved, ev_hsolved;
    hipStream_t aux2;              // the very heavy envs' solve + render (RR_HEAVY2_MIN)
    // Look-ahead: the state part of step t+1 (k_prep_ab, k_collide) runs on the side streams behind the render
    // of the heavy / very heavy envs of step t, beside the main stream's shading.
    struct Frame { float4 *clist; int *ccount; float *cwarm; int *hgflag, *hlist, *hcount, *hlist2, *hcount2, *hpos; } fr[2];
    int cur;                       // fr[cur]: the frame of the last solved step (rr_get_contacts, contact history); fr[cur ^ 1]: the look-ahead's
    bool la_valid;                 // fr[cur ^ 1] and the scratch slab hold the collision pass / dynamics of the next step for the present state
    bool lookahead;                // RR_NO_LOOKAHEAD=1: never ahead, every step prepares itself in line (A/B, tests)
    // Placement knobs of rr_step's schedule, read from the environment at rr_create so that a test can force every
    // branch the lagged heavy counters would pick (tests/test_gpu_round4.py): none of them may change a result.
    int la_vh_max;                 // RR_LA_VH_MAX (64): look-ahead behind the very heavy envs' solve while their lagged count is <= this; -1: never
    int vh_main;                   // RR_VH_ON_MAIN (-1: by the heavy list's length; 0 never; 1 always): very heavy envs' render at the main stream's tail
    bool macro_la_side;            // RR_MACRO_LA=0: with many very heavy envs the look-ahead stays at the tail of the main stream
    bool la_inline;                // RR_UNSPLIT_LA_INLINE: the unsplit step's look-ahead behind its render instead of beside it
    // cost-ordered dispatch of k_raster (RR_NO_RASTER_ORDER=1: env-major grid)
    unsigned *item_perm;           // [8 * ceil(N / 8) * ntiles] the order, written by the extra workgroups of k_shade
    bool ord_valid, ord_pending;   // item_perm holds an order; a k_raster has left costs that the next k_shade launch turns into one
    bool no_fused_setup;           // RR_NO_FUSED_SETUP: separate k_render_setup launch for the light envs
    bool collide_ordered;          // RR_COLLIDE_ORDER=0: k_collide in env order (default: last step's heavy envs first)
    void *obs_host;                // rr_map_observations: mapped pinned block {joints [N][9], touch [N][4], poses [N][nobj][7], timestep [N], errflags [N]} or nullptr
    ObsMirror obs_dev;             // its device-visible addresses
    hipEvent_t ev_obs;             // recorded behind the mirror's launches: rr_sync_observations waits for it alone
    bool ev_obs_set;
    void *img_host[3];             // rr_map_images: pinned host copies of RGB / depth / mask that every rendered step refreshes (or nullptr)
    bool coop_all;                 // RR_COOP_ALL=0: the one-launch solve of a small batch four envs to a wave (A/B, tests)
    int force_hcount[2];           // RR_FORCE_HCOUNT="h,vh": what the host-side decisions read instead of the lagged counters (tests; -1: the counters)
    int n_shapes;
    float table_pos[3];            // target of the default eye camera (env.py:253-255)
    float t_ms[RR_NUM_KERNELS];
    int t_n[RR_NUM_KERNELS];
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> pending;
    // pinned staging ring for per-step host inputs (commands, render flags): a hipMemcpyAsync from pageable memory blocks
    // the host until the copy is done; from these slots it is asynchronous, and a slot is reused only after the event
    // recorded behind its copy has completed
    char *pin_buf[4]; hipEvent_t pin_ev[4]; bool pin_used[4]; int pin_next; size_t pin_bytes;
};

// The lagged host copy of the heavy (which 0) / very heavy (which 1) list length: written to mapped pinned memory by a recent
// step's kernels, read here without any synchronisation -- it only ever selects a launch shape or a placement, never a result
// (every placement is forced and compared bitwise in tests/test_gpu_round4.py; RR_FORCE_HCOUNT pins what is read).
static inline int lagged_count(const rr_env *e, int which, int fallback) {
    if (e->force_hcount[which] >= 0) return e->force_hcount[which];
    return e->h_hcount ? ((volatile int *)e->h_hcount)[which] : fallback;
}

// k_obs (joint angles and object poses of the state -> observation buffers) and, when mapped, the host mirror behind it
static int launch_mirror(rr_env *e, bool rendered = false) {
    if (e->obs_host) hipLaunchKernelGGL(k_obs_mirror, dim3((e->P.N + 63) / 64), dim3(64), 0, e->stream, e->P, e->D, e->obs_dev);
    if (rendered) {
        const int f[3] = {RR_F_RGB, RR_F_DEPTH, RR_F_MASK};
        for (int i = 0; i < 3; i++)
            if (e->img_host[i] && e->field_ptr[f[i]]) HIPCHK(hipMemcpyAsync(e->img_host[i], e->field_ptr[f[i]], e->field_bytes[f[i]], hipMemcpyDeviceToHost, e->stream));
    }
    if (e->obs_host || e->img_host[0] || e->img_host[1] || e->img_host[2]) {
        if (!e->ev_obs) HIPCHK(hipEventCreateWithFlags(&e->ev_obs, hipEventDisableTiming));
        HIPCHK(hipEventRecord(e->ev_obs, e->stream));
        e->ev_obs_set = true;
    }
    HIPCHK(hipGetLastError());
    return RR_OK;
}
static void launch_obs(rr_env *e) {
    hipLaunchKernelGGL(k_obs, dim3((e->P.N + 255) / 256), dim3(256), 0, e->stream, e->P, e->D);
    launch_mirror(e);
}

template <typename T>
static int dev_alloc(rr_env *e, T **p, size_t count, bool zero = true) {
    void *q = nullptr;
    HIPCHK(hipMalloc(&q, count * sizeof(T) + 16));
    if (zero) HIPCHK(hipMemset(q, 0, count * sizeof(T) + 16));
    e->allocs.push_back(q);
    *p = (T *)q;
    return RR_OK;
}

static void look_at_persp(float *VP, const float *table_pos, int W, int H) {
    float eye[3] = {0.01f, 0.0f, 1.2f};                       // env.py:136
    float tgt[3] = {table_pos[0], table_pos[1], table_pos[2]};  // env.py:253-255 (table position)
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
    float fov = 80.0f, nearv = 0.1f, farv = 100.0f;           // env.py:518,548-551
    float aspect = (float)W / (float)H;
    float yscale = 1.0f / tanf(fov * 3.14159265358979323846f / 360.0f);
    float xscale = yscale / aspect;
    float Pm[16] = {xscale, 0, 0, 0, 0, yscale, 0, 0, 0, 0, (nearv + farv) / (nearv - farv), 2 * nearv * farv / (nearv - farv), 0, 0, -1, 0};
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float a = 0;
            for (int k = 0; k < 4; k++) a += Pm[4 * i + k] * V[4 * k + j];
            VP[4 * i + j] = a;
        }
}

static void frustum_plane_norms(RenderModel &RM) {
    const float *V = RM.VP;
    const float sg[4] = {1, -1, 1, -1};
    const int row[4] = {0, 0, 1, 1};
    for (int k = 0; k < 4; k++) {
        float a = V[12] + sg[k] * V[4 * row[k]], b = V[13] + sg[k] * V[4 * row[k] + 1], c = V[14] + sg[k] * V[4 * row[k] + 2];
        RM.plane_norm[k] = sqrtf(a * a + b * b + c * c);
    }
    RM.plane_norm[4] = sqrtf(V[12] * V[12] + V[13] * V[13] + V[14] * V[14]);
    // the two tile-boundary planes of every raster tile (conservative cull of a cluster against the tile's sample rows)
    for (int tile = 0; tile < RM.ntiles && tile < 256; tile++) {
        const int tyi = tile / RM.ntx, tx0 = (tile - tyi * RM.ntx) * RM.tile_w, cols = std::min(RM.tile_w, RM.W - tx0);
        const int row0 = tyi * RM.tile_h, rows = std::min(RM.tile_h, RM.H - row0);
        const float ty0 = (float)(RM.H - 1 - (row0 + rows - 1)), ty1 = (float)(RM.H - 1 - row0);
        for (int k = 0; k < 2; k++) {
            const float ndc = 2.0f * (k ? ty1 : ty0) / (float)RM.H - 1.0f;
            const float a = V[4] - ndc * V[12], b = V[5] - ndc * V[13], c = V[6] - ndc * V[14];
            RM.tile_plane[tile][2 * k] = ndc; RM.tile_plane[tile][2 * k + 1] = sqrtf(a * a + b * b + c * c);
            // (sample column px sits at NDC x = 2 px / W - 1: the viewport maps (x + 1) W / 2)
            const float ndx = 2.0f * (float)(k ? tx0 + cols - 1 : tx0) / (float)RM.W - 1.0f;
            const float ax = V[0] - ndx * V[12], bx = V[1] - ndx * V[13], cx = V[2] - ndx * V[14];
            RM.tile_plane[tile][4 + 2 * k] = ndx; RM.tile_plane[tile][5 + 2 * k] = sqrtf(ax * ax + bx * bx + cx * cx);
        }
    }
}

extern "C" {

const char *rr_last_error(void) { return g_err.c_str(); }
int rr_abi_version(void) { return RR_ABI_VERSION; }

int rr_destroy(rr_env *e) {
    if (!e) return RR_OK;
    hipSetDevice(e->cfg.device);
    hipStreamSynchronize(e->stream);
    for (void *p : e->allocs) hipFree(p);
    for (int i = 0; i < 2 * RR_NUM_KERNELS; i++) if (e->ev[i]) hipEventDestroy(e->ev[i]);
    if (e->aux) { hipStreamSynchronize(e->aux); hipStreamDestroy(e->aux); }
    if (e->aux2) { hipStreamSynchronize(e->aux2); hipStreamDestroy(e->aux2); }

    if (e->ev_join2) hipEventDestroy(e->ev_join2);
    if (e->ev_vsolved) hipEventDestroy(e->ev_vsolved);
    if (e->ev_hsolved) hipEventDestroy(e->ev_hsolved);
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->ev_join) hipEventDestroy(e->ev_join);
    if (e->ev_dyn) hipEventDestroy(e->ev_dyn);
    for (int i = 0; i < 4; i++) { if (e->pin_buf[i]) hipHostFree(e->pin_buf[i]); if (e->pin_ev[i]) hipEventDestroy(e->pin_ev[i]); }
    if (e->h_hcount) hipHostFree(e->h_hcount);
    if (e->obs_host) hipHostFree(e->obs_host);
    for (int i = 0; i < 3; i++) if (e->img_host[i]) hipHostFree(e->img_host[i]);
    if (e->ev_obs) hipEventDestroy(e->ev_obs);
    delete e;
    return RR_OK;
}

// points the device view at the contact frames: current = fr[cur], next = fr[cur ^ 1]
static void bind_frames(rr_env *e) {
    const rr_env::Frame &C = e->fr[e->cur], &X = e->fr[e->cur ^ 1];
    DevPtrs &D = e->D;
    D.clist = C.clist; D.ccount = C.ccount; D.cwarm = C.cwarm; D.hgflag = C.hgflag; D.hlist = C.hlist; D.hcount = C.hcount; D.hlist2 = C.hlist2; D.hcount2 = C.hcount2;
    D.clist_next = X.clist; D.ccount_next = X.ccount; D.cwarm_next = X.cwarm; D.hgflag_next = X.hgflag; D.hlist_next = X.hlist; D.hcount_next = X.hcount;
    D.hlist2_next = X.hlist2; D.hcount2_next = X.hcount2;
    D.hpos = C.hpos; D.hpos_next = X.hpos;
}

// The contact count and the class of every env belong to the contact frame of the last solved step, which changes place every
// step: the solve kernels publish both into fixed [N] buffers, so that a pointer from rr_get_buffer stays valid (realrobot.h).
static void refresh_frame_fields(rr_env *e) {
    e->field_ptr[RR_F_CONTACT_COUNT] = e->D.ccount_pub;
    e->field_ptr[RR_F_ENV_CLASS] = e->D.class_pub;
}

static ImageOut env_images(const rr_env *e) {
    ImageOut o;
    o.rgb = e->D.rgb; o.depth = e->D.depth; o.mask = e->D.mask; o.env_stride = (size_t)e->RM.W * e->RM.H;
    return o;
}

// (Re)builds the shared static layer for the current camera: background everywhere, then -- unless disabled with
// RR_NO_STATIC_LAYER -- the never-moving instances (table, shelf, robot base; the eye camera is fixed, env.py:136-141,
// 253-255) are rasterised and shaded once; their visibility keys seed every env's frame.
static int build_static_layer(rr_env *e) {
    hipLaunchKernelGGL(k_background, dim3(64), dim3(256), 0, e->stream, e->RM_dev, e->D);
    if (e->D.static_vis_out) {
        ImageOut so;
        so.rgb = e->D.static_rgb; so.depth = e->D.static_depth; so.mask = e->D.static_mask; so.env_stride = 0;
        e->D.static_vis = nullptr;
        hipLaunchKernelGGL(k_render_setup, dim3((e->P.N * MAXINST + 63) / 64), dim3(64), 0, e->stream, e->B, e->P, e->RM_dev, e->D, 0);
        hipLaunchKernelGGL(k_raster, dim3(1, e->RM.ntiles), dim3(RASTER_THREADS), 0, e->stream, e->P, e->RM_dev, e->D, e->n_inst_used, 1, 0, 0, 0);
        hipLaunchKernelGGL(k_shade, dim3(1, e->RM.ntiles, SHADE_SPLIT), dim3(SHADE_THREADS), 0, e->stream, e->RM_dev, e->D, so, 0, 0, 0, 0, (unsigned *)nullptr);
    }
    // the pass above used env 0's fragment list; from here on the lists describe what differs from the static layer
    if (hipMemsetAsync(e->D.frag_count, 0, (size_t)e->P.N * e->RM.ntiles * sizeof(unsigned), e->stream) != hipSuccess ||
        hipStreamSynchronize(e->stream) != hipSuccess) return fail(RR_EDEVICE, "static layer pass failed");
    e->D.static_vis = e->D.static_vis_out;
    e->images_valid = false;         // the next render starts from a full copy of the new static layer
    return RR_OK;
}

int rr_create(const rr_config *cfg, const void *model_blob, size_t blob_bytes, void *stream, rr_env **out) {
    if (!cfg || !model_blob || !out) return fail(RR_EINVAL, "rr_create: null argument");
    if (cfg->abi_version != RR_ABI_VERSION) return fail(RR_EINVAL, "rr_create: abi_version mismatch");
    if (cfg->num_envs < 1) return fail(RR_EINVAL, "rr_create: num_envs < 1");
    if ((unsigned long long)cfg->num_envs * GP_ENV_BYTES + 4096ull >= (1ull << 32)) return fail(RR_EINVAL, "rr_create: more than 33222 envs per rr_env (32-bit byte offsets of the solver's row store)");
    if (cfg->n_objects < 1 || cfg->n_objects > NOBJ) return fail(RR_EINVAL, "rr_create: n_objects must be 1..3");
    *out = nullptr;
    if (cfg->width < 4 || cfg->height < 1 || cfg->width % 4 != 0 || cfg->width > 1024 || cfg->height > 1024)
        return fail(RR_EINVAL, "rr_create: width must be a multiple of 4 in [4,1024], height in [1,1024] (10-bit box origins in the rasteriser's records)");
    {   // (the raster tiles of this image: rr_create lays them out below by the same rule; the tile index is 8 bits with one sentinel)
        int tw_ = cfg->width <= 128 ? cfg->width : 64;
        if (getenv("RR_TILE_W")) { const int t_ = atoi(getenv("RR_TILE_W")); if (t_ >= 4 && t_ <= cfg->width && t_ <= TILE_PIX) tw_ = t_; }
        int th_ = TILE_PIX / tw_; if (th_ > cfg->height) th_ = cfg->height;
        const int nt_ = ((cfg->width + tw_ - 1) / tw_) * ((cfg->height + th_ - 1) / th_);
        if (nt_ > 255) return fail(RR_EINVAL, "rr_create: image too large: more than 255 raster tiles of 4096 pixels (e.g. 1024 x 1020 fits, 1024 x 1024 does not)");
    }
    Blob b;
    if (!b.init(model_blob, blob_bytes)) return fail(RR_EMODEL, "rr_create: bad model blob header");
    const int32_t *dims = b.i32("dims", 11);
    if (!dims) return fail(RR_EMODEL, "rr_create: blob has no dims");
    int nb = dims[0], nl = dims[1], ns = dims[2], ni = dims[3], nt = dims[4], ntex = dims[5], n_static = dims[6], n_robot = dims[7];
    if (nb != NB || ns > MAXSHAPES || dims[8] != VMAXC || dims[9] != FMAXC || ni > MAXINST || ni > RASTER_INST || nl > NLINK_MAX || ntex > 16 ||
        n_static != 3 || n_robot != 16)
        return fail(RR_EMODEL, "rr_create: blob dims do not match this build");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(RR_EDEVICE, "rr_create: no HIP device available (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(RR_EINVAL, "rr_create: bad device ordinal");
    HIPCHK(hipSetDevice(cfg->device));

    rr_env *e = new rr_env();
    memset(&e->B, 0, sizeof e->B); memset(&e->RM, 0, sizeof e->RM); memset(&e->D, 0, sizeof e->D);
    memset(e->ev, 0, sizeof e->ev); memset(e->t_ms, 0, sizeof e->t_ms); memset(e->t_n, 0, sizeof e->t_n);
    e->timing = false;
    memset(e->pin_buf, 0, sizeof e->pin_buf); memset(e->pin_ev, 0, sizeof e->pin_ev); memset(e->pin_used, 0, sizeof e->pin_used); e->pin_next = 0; e->pin_bytes = 0;
    e->full_copy = getenv("RR_FULL_COPY") != nullptr;
    e->sep_restore = getenv("RR_SEPARATE_RESTORE") != nullptr;
    e->split_heavy = getenv("RR_NO_SPLIT") == nullptr;
    e->lookahead = getenv("RR_NO_LOOKAHEAD") == nullptr;
    e->la_vh_max = getenv("RR_LA_VH_MAX") ? atoi(getenv("RR_LA_VH_MAX")) : 64;
    e->vh_main = getenv("RR_VH_ON_MAIN") ? atoi(getenv("RR_VH_ON_MAIN")) : -1;
    e->macro_la_side = !(getenv("RR_MACRO_LA") && atoi(getenv("RR_MACRO_LA")) == 0);
    e->la_inline = getenv("RR_UNSPLIT_LA_INLINE") != nullptr;
    e->item_perm = nullptr; e->ord_valid = e->ord_pending = false;
    e->no_fused_setup = getenv("RR_NO_FUSED_SETUP") != nullptr;
    e->collide_ordered = !(getenv("RR_COLLIDE_ORDER") && atoi(getenv("RR_COLLIDE_ORDER")) == 0);
    e->force_hcount[0] = e->force_hcount[1] = -1;
    e->coop_all = !(getenv("RR_COOP_ALL") && atoi(getenv("RR_COOP_ALL")) == 0) && getenv("RR_NO_COOP") == nullptr;
    e->obs_host = nullptr; memset(&e->obs_dev, 0, sizeof e->obs_dev);
    e->img_host[0] = e->img_host[1] = e->img_host[2] = nullptr;
    e->ev_obs = nullptr; e->ev_obs_set = false;
    if (getenv("RR_FORCE_HCOUNT")) sscanf(getenv("RR_FORCE_HCOUNT"), "%d,%d", &e->force_hcount[0], &e->force_hcount[1]);

    e->h_hcount = nullptr;
    e->split_max_pct = getenv("RR_SPLIT_MAX_PCT") ? atoi(getenv("RR_SPLIT_MAX_PCT")) : 60;
    if (hipHostMalloc((void **)&e->h_hcount, 4 * sizeof(int), hipHostMallocMapped) == hipSuccess) { e->h_hcount[0] = 0; e->h_hcount[1] = 0; e->h_hcount[2] = -1; e->h_hcount[3] = 0; } else e->h_hcount = nullptr;
    e->plan = nullptr; e->plan_step = nullptr; e->ik_in = nullptr; e->ik_out = nullptr; e->ik_err = nullptr;
    e->score_out = nullptr; e->score_mask = nullptr;
    e->cfg = *cfg;
    e->stream = (hipStream_t)stream;
    const int N = cfg->num_envs;
#define NEED(p) if (!(p)) { rr_destroy(e); return fail(RR_EMODEL, "rr_create: blob entry missing/short: " #p); }
    const float *f; const int32_t *ip;
    BodyParams &B = e->B;
    NEED(ip = b.i32("body_parent", NB)); memcpy(B.parent, ip, sizeof B.parent);
    if (memcmp(B.parent, PARENT_HOST, sizeof PARENT_HOST) != 0) { rr_destroy(e); return fail(RR_EMODEL, "rr_create: kinematic tree differs from the compiled-in one"); }
    NEED(f = b.f32("body_jpos", NB * 3)); memcpy(B.jpos, f, sizeof B.jpos);
    NEED(f = b.f32("body_jrot", NB * 9)); memcpy(B.jrot, f, sizeof B.jrot);
    NEED(f = b.f32("body_axis", NB * 3)); memcpy(B.axis, f, sizeof B.axis);
    NEED(f = b.f32("body_mass", NB)); memcpy(B.mass, f, sizeof B.mass);
    NEED(f = b.f32("body_com", NB * 3)); memcpy(B.com, f, sizeof B.com);
    NEED(f = b.f32(cfg->use_urdf_inertia ? "body_inertia_urdf" : "body_inertia", NB * 6)); memcpy(B.inertia, f, sizeof B.inertia);
    NEED(f = b.f32("body_damping", NB)); memcpy(B.damping, f, sizeof B.damping);
    NEED(f = b.f32("body_limits", NB * 2)); memcpy(B.limits, f, sizeof B.limits);
    NEED(f = b.f32("robot_pos", 3)); memcpy(B.robot_pos, f, sizeof B.robot_pos);
    NEED(f = b.f32("obj_mass", NOBJ)); memcpy(B.obj_mass, f, sizeof B.obj_mass);
    NEED(f = b.f32("obj_inertia", NOBJ * 3)); memcpy(B.obj_inertia, f, sizeof B.obj_inertia);
    NEED(f = b.f32("obj_pose0", NOBJ * 7)); memcpy(B.obj_pose0, f, sizeof B.obj_pose0);
    const float *table_pos;
    NEED(table_pos = b.f32("table_pos", 3)); B.table_z = table_pos[2];
    NEED(f = b.f32("act_min", 9)); memcpy(B.act_min, f, sizeof B.act_min);
    NEED(f = b.f32("act_max", 9)); memcpy(B.act_max, f, sizeof B.act_max);
    NEED(f = b.f32("act_maxdiff", 9)); memcpy(B.act_maxdiff, f, sizeof B.act_maxdiff);
    NEED(ip = b.i32("touch_links", 4)); memcpy(B.touch_links, ip, sizeof B.touch_links);
    {   // arm chain + gripper base frame (link id 8 = `base`, rigidly attached to body 6) for the IK kernels
        IkModel &K = e->IK;
        memset(&K, 0, sizeof K);
        for (int j = 0; j < 7; j++) { memcpy(K.jpos[j], B.jpos[j], 12); memcpy(K.jrot[j], B.jrot[j], 36); memcpy(K.axis[j], B.axis[j], 12); }
        memcpy(K.robot_pos, B.robot_pos, 12);
        const float *lp, *lr; const int32_t *lb;
        NEED(lb = b.i32("link_body", nl)); NEED(lp = b.f32("link_pos", nl * 3)); NEED(lr = b.f32("link_rot", nl * 9));
        const int ee = 8;       // URDF depth-first id of the gripper `base` link (pybullet link index 7)
        if (nl <= ee || lb[ee] != 6) { rr_destroy(e); return fail(RR_EMODEL, "rr_create: gripper base link not found"); }
        memcpy(K.ee_pos, lp + 3 * ee, 12); memcpy(K.ee_rot, lr + 9 * ee, 36);
    }

    SimParams &P = e->P;
    P.N = N; P.nobj = cfg->n_objects; P.iters = cfg->solver_iters > 0 ? cfg->solver_iters : 50;
    P.dt = cfg->dt > 0 ? cfg->dt : 0.005f; P.gravity = 9.81f; P.erp = cfg->erp > 0 ? cfg->erp : 0.2f;
    P.margin = cfg->margin > 0 ? cfg->margin : 0.02f; P.kp = 0.1f; P.kd = 1.0f; P.max_impulse = 100000.0f * P.dt;
    P.ablate = getenv("RR_ABLATE") ? atoi(getenv("RR_ABLATE")) : 0;
    P.small_area = std::min(64, getenv("RR_SMALL_AREA") ? atoi(getenv("RR_SMALL_AREA")) : SMALL_AREA);      // (<= 64: a record's box width has 6 bits)
    // RR_SOLVER_POOL (tests): LDS floats for object-vs-static rows, 60 per contact; contacts beyond it take the generic (slot layout) path
    P.heavy_min = getenv("RR_HEAVY_MIN") ? atoi(getenv("RR_HEAVY_MIN")) : 0;
    P.heavy2_min = getenv("RR_HEAVY2_MIN") ? atoi(getenv("RR_HEAVY2_MIN")) : 16;      // generic contacts above which an env is "very heavy" (1000: never; A/B 6..30: 13-16 best)
    P.warmstart = getenv("RR_NO_WARMSTART") ? 0.0f : 0.85f;       // (diagnostics: cold start every step)
    P.coop_build = getenv("RR_NO_COOP") ? 0 : 1;                  // (A/B, tests: very heavy envs four to a wave like the heavy ones)
    P.edge_contacts = getenv("RR_NO_EDGE_CONTACTS") ? 0 : 1;       // (diagnostics: vertex candidates only)
    P.os_cap = getenv("RR_SOLVER_POOL") ? std::max(0, std::min(atoi(getenv("RR_SOLVER_POOL")) / 60, (int)OS_CAP)) : OS_CAP;
    P.lin_damp = 0.04f; P.ang_damp = 0.04f; P.rest_thresh = 0.2f;
    e->epb = cfg->envs_per_block > 0 ? cfg->envs_per_block : 64;
    if (e->epb > 64) e->epb = 64;   // physics kernels are compiled with __launch_bounds__(64)

    // shapes + pair table (same order as the oracle's collide())
    std::vector<ShapeData> sdv(1);
    ShapeData &S = sdv[0];
    memset(&S, 0, sizeof S);
    NEED(ip = b.i32("shape_owner", ns * 4));
    for (int s = 0; s < ns; s++) { S.otype[s] = ip[4 * s]; S.oidx[s] = ip[4 * s + 1]; S.link[s] = ip[4 * s + 2]; }
    NEED(ip = b.i32("shape_nv", ns)); memcpy(S.nv, ip, ns * 4);
    NEED(ip = b.i32("shape_nf", ns)); memcpy(S.nf, ip, ns * 4);
    NEED(f = b.f32("shape_verts", ns * VMAXC * 3)); memcpy(S.verts, f, (size_t)ns * VMAXC * 3 * 4);
    NEED(f = b.f32("shape_planes", ns * FMAXC * 4)); memcpy(S.planes, f, (size_t)ns * FMAXC * 4 * 4);
    NEED(f = b.f32("shape_sphere", ns * 4)); memcpy(S.sphere, f, (size_t)ns * 4 * 4);
    NEED(f = b.f32("shape_mat", ns * 2));
    for (int s = 0; s < ns; s++) { S.fric[s] = f[2 * s]; S.rest[s] = f[2 * s + 1]; }
    NEED(f = b.f32("shape_roll", ns * 2));       // URDF <rolling_friction>, <spinning_friction> (cube.urdf:6-7, kuka_gripper.urdf:292-296 ...)
    for (int s = 0; s < ns; s++) { S.roll[s] = f[2 * s]; S.spin[s] = f[2 * s + 1]; }
    NEED(ip = b.i32("shape_ne", ns)); memcpy(S.ne, ip, ns * 4);
    NEED(f = b.f32("shape_edges", ns * EMAXC * 12)); memcpy(S.edges, f, (size_t)ns * EMAXC * 12 * 4);
    for (int s = 0; s < ns; s++) if (S.ne[s] < 0 || S.ne[s] > EMAXC) { rr_destroy(e); return fail(RR_EMODEL, "rr_create: bad shape_ne"); }
    int np = 0, s_obj0 = n_static + n_robot;
    for (int i = 0; i < P.nobj; i++) for (int s = 0; s < n_static; s++) { S.pair_a[np] = s_obj0 + i; S.pair_b[np++] = s; }
    for (int i = 0; i < P.nobj; i++) for (int j = i + 1; j < P.nobj; j++) { S.pair_a[np] = s_obj0 + i; S.pair_b[np++] = s_obj0 + j; }
    for (int r = 0; r < n_robot; r++) for (int s = 0; s < 2; s++) { S.pair_a[np] = n_static + r; S.pair_b[np++] = s; }
    for (int r = 0; r < n_robot; r++) for (int i = 0; i < P.nobj; i++) { S.pair_a[np] = n_static + r; S.pair_b[np++] = s_obj0 + i; }
    P.npairs = np;
    e->n_shapes = ns;
    if (ns > CSHAPES) { rr_destroy(e); return fail(RR_EMODEL, "rr_create: more collision shapes than k_collide stages in LDS"); }
    for (int k = 0; k < np; k++) {
        const int sa = S.pair_a[k], sb = S.pair_b[k];
        S.pair_meta[k][0] = S.otype[sa] == 0 ? -1 : (S.otype[sa] == 1 ? S.oidx[sa] : 16 + S.oidx[sa]);
        S.pair_meta[k][1] = S.otype[sb] == 0 ? -1 : (S.otype[sb] == 1 ? S.oidx[sb] : 16 + S.oidx[sb]);
        S.pair_meta[k][2] = S.link[sa]; S.pair_meta[k][3] = 0;
        S.pair_mat[k][0] = S.fric[sa] * S.fric[sb]; S.pair_mat[k][1] = S.rest[sa] * S.rest[sb];
        // btManifoldResult::calculateCombinedRollingFriction / SpinningFriction: r_a mu_b + r_b mu_a, at most 10
        S.pair_mat[k][2] = std::min(S.roll[sa] * S.fric[sb] + S.roll[sb] * S.fric[sa], 10.0f);
        S.pair_mat[k][3] = std::min(S.spin[sa] * S.fric[sb] + S.spin[sb] * S.fric[sa], 10.0f);
    }

    // k_collide's warm-start matching looks for the previous contacts of the same bodies (bodyA, bodyB, linkA) among the
    // pairs pair-2 .. pair+2 only: pairs with equal keys must form runs of at most three consecutive pairs (one collision shape
    // per robot link against table / shelf, one per object against the three statics)
    for (int k = 0; k < np; k++)
        for (int j = 0; j < np; j++) {
            const bool same = S.pair_meta[k][0] == S.pair_meta[j][0] && S.pair_meta[k][1] == S.pair_meta[j][1] && S.pair_meta[k][2] == S.pair_meta[j][2];
            if (same && std::abs(k - j) > 2) { rr_destroy(e); return fail(RR_EMODEL, "rr_create: collision pairs of the same bodies are more than two apart in the pair table (warm-start matching window)"); }
        }

    // render model
    RenderModel &RM = e->RM;
    RM.ni = ni; RM.nt = nt; RM.W = cfg->width; RM.H = cfg->height; RM.nl = nl;
    // Raster tiles of <= TILE_PIX pixels: full-width strips up to 128 columns (the 128 x 128 benchmark camera: four strips of 32
    // rows); 64 x 64 squares for wider images -- a cluster of the arm is ~25 pixels across at 320 x 240 and met three of the
    // 12-row strips a full-width tile would be there (set-up work x 3.1; squares: x 1.9).  RR_TILE_W overrides (A/B, tests).
    RM.tile_w = RM.W <= 128 ? RM.W : 64;
    if (getenv("RR_TILE_W")) { const int tw_ = atoi(getenv("RR_TILE_W")); if (tw_ >= 4 && tw_ <= RM.W && tw_ <= TILE_PIX) RM.tile_w = tw_; }
    RM.ntx = (RM.W + RM.tile_w - 1) / RM.tile_w;
    RM.tile_h = TILE_PIX / RM.tile_w; if (RM.tile_h > RM.H) RM.tile_h = RM.H;
    RM.ntiles = RM.ntx * ((RM.H + RM.tile_h - 1) / RM.tile_h);
    RM.tile_xbits = 0; while ((1 << RM.tile_xbits) < RM.tile_w) RM.tile_xbits++;
    RM.w_magic = (unsigned)((0x100000000ull + (unsigned long long)RM.tile_w - 1) / (unsigned long long)RM.tile_w);
    if (RM.W > 1024 || RM.H > 1024 || RM.ntiles > 255) { rr_destroy(e); *out = nullptr; return fail(RR_EINVAL, "rr_create: image too large (checked on entry)"); }
    NEED(ip = b.i32("inst_owner", ni * 4));
    for (int i = 0; i < ni; i++) { RM.in_otype[i] = ip[4 * i]; RM.in_oidx[i] = ip[4 * i + 1]; RM.in_uid[i] = ip[4 * i + 2]; RM.in_tex[i] = ip[4 * i + 3]; }
    NEED(f = b.f32("inst_color", ni * 3)); memcpy(RM.in_color, f, (size_t)ni * 12);
    // Back-face culling of closed meshes is opt-in (RR_CULL=1): it is invisible unless the near plane cuts through a
    // mesh (then TinyRenderer shows the inside faces), so the default keeps exact parity with the two-sided oracle.
    if (getenv("RR_CULL")) { NEED(ip = b.i32("inst_cull", ni)); memcpy(RM.in_cull, ip, (size_t)ni * 4); for (int i = 0; i < ni; i++) RM.any_cull |= ip[i] != 0; }
    NEED(ip = b.i32("tex_info", ntex * 3));
    for (int t = 0; t < ntex; t++) { RM.tex_off[t] = ip[3 * t]; RM.tex_w[t] = ip[3 * t + 1]; RM.tex_h[t] = ip[3 * t + 2]; }
    NEED(ip = b.i32("link_body", nl)); memcpy(RM.link_body, ip, nl * 4);
    NEED(f = b.f32("link_pos", nl * 3)); memcpy(RM.link_pos, f, (size_t)nl * 12);
    NEED(f = b.f32("link_rot", nl * 9)); memcpy(RM.link_rot, f, (size_t)nl * 36);
    {
        const int32_t *ir;
        NEED(ir = b.i32("inst_range", ni * 2));
        int n_static_inst = dims[10];
        RM.first_dynamic_tri = (n_static_inst < ni) ? ir[2 * n_static_inst] : nt;
    }
    memcpy(e->table_pos, table_pos, sizeof e->table_pos);
    look_at_persp(RM.VP, table_pos, RM.W, RM.H);
    frustum_plane_norms(RM);
    e->n_inst_used = ni - (NOBJ - P.nobj);

    // device allocations
    DevPtrs &D = e->D;
    int rc;
#define ALLOC(ptr, count) if ((rc = dev_alloc(e, &(ptr), (count))) != RR_OK) { rr_destroy(e); return rc; }
    ALLOC(D.state, (size_t)ST_TOTAL * N);
    ALLOC(D.scratch, (size_t)S_TOTAL * N);
    for (int f = 0; f < 2; f++) {
        rr_env::Frame &F = e->fr[f];
        ALLOC(F.clist, (size_t)N * MAXC * 3); ALLOC(F.ccount, (size_t)N); ALLOC(F.cwarm, (size_t)N * MAXC);
        ALLOC(F.hgflag, (size_t)N); ALLOC(F.hpos, (size_t)N); ALLOC(F.hlist, (size_t)N); ALLOC(F.hcount, (size_t)4); ALLOC(F.hlist2, (size_t)N); ALLOC(F.hcount2, (size_t)4);
    }
    e->cur = 0; e->la_valid = false;
    bind_frames(e);
    ALLOC(D.cforce, (size_t)N * MAXC);
    ALLOC(D.ccount_pub, (size_t)N); ALLOC(D.class_pub, (size_t)N);
    e->D.hcount_host = nullptr;
    if (e->h_hcount && hipHostGetDevicePointer((void **)&e->D.hcount_host, e->h_hcount, 0) != hipSuccess) e->D.hcount_host = nullptr;
    ALLOC(D.timestep, (size_t)N);
    ALLOC(D.errflags, (size_t)N);
    ALLOC(D.obj_home, (size_t)NOBJ * 7 * N);
    ALLOC(D.grows, (size_t)N * GP_RECS * 16);        // (zeroed: the dummy block / contact records stay all zero)
    ALLOC(D.cmd, (size_t)N * 9);
    ALLOC(D.joints, (size_t)N * 9);
    ALLOC(D.touch, (size_t)N * 4);
    ALLOC(D.objpose, (size_t)N * P.nobj * 7);
    ALLOC(D.inst_xf, (size_t)N * MAXINST * 32);
    ALLOC(D.render_flags, (size_t)N);
    const size_t npx = (size_t)N * RM.W * RM.H;
    ALLOC(D.rgb, npx * 3);
    ALLOC(D.depth, npx);
    if (!(cfg->flags & RR_FLAG_NO_MASK)) ALLOC(D.mask, npx);
    ALLOC(e->state_aos, (size_t)N * NSTATE);
    ALLOC(e->mask_dev, (size_t)N);
    ALLOC(e->link_out, (size_t)N * nl * 7);
    {   // geometry: positions SoA [9][NT], normals/uv AoS
        const float *tp, *tn, *tu; const int32_t *ti;
        NEED(tp = b.f32("tri_pos", (size_t)nt * 9)); NEED(tn = b.f32("tri_nrm", (size_t)nt * 9));
        NEED(tu = b.f32("tri_uv", (size_t)nt * 6)); NEED(ti = b.i32("tri_inst", nt));
        std::vector<float> soa((size_t)nt * 9);
        for (int t = 0; t < nt; t++) for (int k = 0; k < 9; k++) soa[(size_t)k * nt + t] = tp[(size_t)t * 9 + k];
        std::vector<float> rec((size_t)nt * 32, 0.0f);
        for (int t = 0; t < nt; t++) {
            float *r = &rec[(size_t)t * 32];
            memcpy(r, tp + (size_t)t * 9, 36); memcpy(r + 9, tn + (size_t)t * 9, 36); memcpy(r + 18, tu + (size_t)t * 6, 24);
            memcpy(r + 24, ti + t, 4);
        }
        float *dp; float4 *drec; int *di; unsigned *dt_; ShapeData *ds;
        ALLOC(dp, (size_t)nt * 9); ALLOC(drec, (size_t)nt * 8); ALLOC(di, (size_t)nt);
        hipMemcpy(dp, soa.data(), (size_t)nt * 36, hipMemcpyHostToDevice);
        hipMemcpy(drec, rec.data(), (size_t)nt * 128, hipMemcpyHostToDevice);
        hipMemcpy(di, ti, (size_t)nt * 4, hipMemcpyHostToDevice);
        size_t texbytes = 0;
        const uint8_t *tex = b.u8("tex_data", &texbytes);
        NEED(tex);
        ALLOC(dt_, texbytes / 4 + 1);
        hipMemcpy(dt_, tex, texbytes, hipMemcpyHostToDevice);
        ALLOC(ds, 1);
        hipMemcpy(ds, &S, sizeof S, hipMemcpyHostToDevice);
        ALLOC(e->RM_dev, 1);
        hipMemcpy(e->RM_dev, &e->RM, sizeof e->RM, hipMemcpyHostToDevice);
        D.tri_pos = dp; D.tri_rec = drec; D.tri_inst = di; D.tex = dt_; D.shapes = ds;
        {
            const float *cs;
            if (nt % 64 != 0 || nt / 64 > MAXWIN || nt >= (1 << 18)) { rr_destroy(e); return fail(RR_EMODEL, "rr_create: triangle count must be a multiple of the cluster size 64 and below 65536"); }
            NEED(cs = b.f32("cluster_sphere", (size_t)(nt / 64) * 4));
            {
                const float *cvb; const int32_t *tv;
                NEED(cvb = b.f32("cluster_verts", (size_t)nt * 3)); NEED(tv = b.i32("tri_vidx", nt));
                std::vector<float> cvs((size_t)nt * 3);          // [cluster][64][3] -> [cluster][3][64]
                for (int c = 0; c < nt / 64; c++)
                    for (int v = 0; v < 64; v++)
                        for (int k = 0; k < 3; k++) cvs[((size_t)c * 3 + k) * 64 + v] = cvb[((size_t)c * 64 + v) * 3 + k];
                float *dcv; int *dtv;
                ALLOC(dcv, (size_t)nt * 3); ALLOC(dtv, (size_t)nt);
                hipMemcpy(dcv, cvs.data(), (size_t)nt * 12, hipMemcpyHostToDevice);
                // (the corner indices go to the device as ds_bpermute byte addresses, index x 4 in each byte: one bit-field extract per
                // corner in the window loop instead of a shift and a mask)
                std::vector<int32_t> tv4((size_t)nt);
                for (int t_ = 0; t_ < nt; t_++) {
                    const int a0 = tv[t_] & 63, a1 = (tv[t_] >> 8) & 63, a2 = (tv[t_] >> 16) & 63;
                    tv4[t_] = (a0 << 2) | (a1 << 10) | (a2 << 18);
                }
                hipMemcpy(dtv, tv4.data(), (size_t)nt * 4, hipMemcpyHostToDevice);
                D.cluster_verts = dcv; D.tri_vidx = dtv;
            }
            float4 *dcs;
            ALLOC(dcs, (size_t)nt / 64);
            hipMemcpy(dcs, cs, (size_t)(nt / 64) * 16, hipMemcpyHostToDevice);
            D.cluster_sphere = dcs;
        }
    }
    for (int i = 0; i < 2 * RR_NUM_KERNELS; i++) hipEventCreate(&e->ev[i]);
    if (!getenv("RR_NO_AUX_STREAM")) {
        // fork / join events order two streams of the same device: no timing, no system-scope fence (the cache writeback
        // and invalidation a default event performs when it is recorded costs ~6 us on the stream that records it)
        const unsigned evf = getenv("RR_EVENT_FLAGS") ? (unsigned)strtoul(getenv("RR_EVENT_FLAGS"), nullptr, 0) : (hipEventDisableTiming | hipEventDisableSystemFence);
        // the side stream carries the step's longest chain (the heavy envs' solve, then their render): with a higher
        // priority its few workgroups are dispatched ahead of the main stream's render when both are ready (RR_AUX_PRIORITY=0: same)
        int prio_lo = 0, prio_hi = 0;
        hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        const int prio = getenv("RR_AUX_PRIORITY") && atoi(getenv("RR_AUX_PRIORITY")) == 0 ? prio_lo : prio_hi;
        // (a runtime without stream priorities: plain non-blocking streams)
        auto side_stream = [&](hipStream_t *st) { return hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio) == hipSuccess || hipStreamCreateWithFlags(st, hipStreamNonBlocking) == hipSuccess; };
        if (!side_stream(&e->aux) || hipEventCreateWithFlags(&e->ev_fork, evf) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_join, evf) != hipSuccess || hipEventCreateWithFlags(&e->ev_dyn, evf) != hipSuccess ||
            !side_stream(&e->aux2) || hipEventCreateWithFlags(&e->ev_join2, evf) != hipSuccess ||
            hipEventCreateWithFlags(&e->ev_vsolved, evf) != hipSuccess || hipEventCreateWithFlags(&e->ev_hsolved, evf) != hipSuccess) { rr_destroy(e); return fail(RR_EDEVICE, "rr_create: side stream"); }
    }
    e->field_ptr[RR_F_JOINTS] = D.joints; e->field_bytes[RR_F_JOINTS] = (size_t)N * 9 * 4;
    e->field_ptr[RR_F_TOUCH] = D.touch; e->field_bytes[RR_F_TOUCH] = (size_t)N * 4 * 4;
    e->field_ptr[RR_F_OBJ_POSE] = D.objpose; e->field_bytes[RR_F_OBJ_POSE] = (size_t)N * P.nobj * 7 * 4;
    e->field_ptr[RR_F_RGB] = D.rgb; e->field_bytes[RR_F_RGB] = npx * 3;
    e->field_ptr[RR_F_DEPTH] = D.depth; e->field_bytes[RR_F_DEPTH] = npx * 4;
    e->field_ptr[RR_F_MASK] = D.mask; e->field_bytes[RR_F_MASK] = D.mask ? npx * 4 : 0;
    e->field_ptr[RR_F_TIMESTEP] = D.timestep; e->field_bytes[RR_F_TIMESTEP] = (size_t)N * 4;
    e->field_ptr[RR_F_ERRFLAGS] = D.errflags; e->field_bytes[RR_F_ERRFLAGS] = (size_t)N * 4;
    e->field_ptr[RR_F_STATE] = e->state_aos; e->field_bytes[RR_F_STATE] = (size_t)N * NSTATE * 4;
    e->field_bytes[RR_F_FRAG_COUNT] = (size_t)N * RM.ntiles * 4;     // pointer set once the list is allocated
    e->field_bytes[RR_F_CONTACT_COUNT] = (size_t)N * 4; e->field_bytes[RR_F_ENV_CLASS] = (size_t)N * 4;
    refresh_frame_fields(e);
    *out = e;
    for (int i = 0; i < NOBJ; i++) { const int rh = rr_set_object_home(e, -1, i, e->B.obj_pose0[i]); if (rh != RR_OK) { rr_destroy(e); *out = nullptr; return rh; } }
    int r = rr_reset(e, nullptr);
    if (r != RR_OK) { rr_destroy(e); *out = nullptr; return r; }
    {
        const size_t spx = (size_t)RM.W * RM.H;
        if ((r = dev_alloc(e, &e->D.static_rgb, spx * 3)) != RR_OK || (r = dev_alloc(e, &e->D.static_depth, spx)) != RR_OK ||
            (r = dev_alloc(e, &e->D.static_mask, spx)) != RR_OK || (r = dev_alloc(e, &e->D.frag_count, (size_t)N * RM.ntiles)) != RR_OK ||
            (r = dev_alloc(e, &e->D.frag_list, (size_t)N * RM.ntiles * TILE_PIX, false)) != RR_OK) { rr_destroy(e); *out = nullptr; return r; }
        if (!getenv("RR_NO_RASTER_ORDER") && (long long)N * RM.ntiles >= 2048 && (long long)N * RM.ntiles <= (1 << 20) && N < (1 << 24)) {
            if ((r = dev_alloc(e, &e->D.item_cost, (size_t)N * RM.ntiles)) != RR_OK || (r = dev_alloc(e, &e->D.item_bin, (size_t)N * RM.ntiles)) != RR_OK || (r = dev_alloc(e, &e->item_perm, (size_t)8 * ((N + 7) / 8) * RM.ntiles)) != RR_OK) { rr_destroy(e); *out = nullptr; return r; }
        }
        if (!getenv("RR_NO_STATIC_LAYER")) {
            unsigned long long *sv = nullptr;
            if ((r = dev_alloc(e, &sv, spx)) != RR_OK) { rr_destroy(e); *out = nullptr; return r; }
            e->D.static_vis_out = sv;
        }
        e->field_ptr[RR_F_FRAG_COUNT] = e->D.frag_count;
        if ((r = build_static_layer(e)) != RR_OK) { rr_destroy(e); *out = nullptr; return r; }
    }
    // the 256-thread form of k_solve (heavy solver groups, four per workgroup) asks for 158 KiB of dynamic LDS: only the
    // heavy / light split launches it, and a device that cannot grant it runs without the split (same results, one launch)
    if (e->split_heavy && hipFuncSetAttribute((const void *)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * SGRP * LF_TOTAL * sizeof(float))) != hipSuccess) {
        (void)hipGetLastError();
        e->split_heavy = false;
    }
    // the light envs' solve with an object wave (five waves, sixteen envs and the same 158 KiB per workgroup); without the
    // attribute (or with RR_NO_OBJECT_WAVE) the one-group-per-env form in 64-thread workgroups -- same results
    e->light_ow = e->split_heavy && !getenv("RR_NO_OBJECT_WAVE");
    if (e->light_ow && hipFuncSetAttribute((const void *)k_solve_light_ow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(4 * SGRP * LF_TOTAL * sizeof(float))) != hipSuccess) {
        (void)hipGetLastError();
        e->light_ow = false;
    }
    if (hipGetLastError() != hipSuccess) { rr_destroy(e); *out = nullptr; return fail(RR_EDEVICE, "rr_create: device error during set-up"); }
    return RR_OK;
}

int rr_set_stream(rr_env *e, void *stream) {
    if (!e) return fail(RR_EINVAL, "null env");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->stream = (hipStream_t)stream;
    return RR_OK;
}

static inline dim3 env_grid(const rr_env *e) { return dim3((e->P.N + e->epb - 1) / e->epb); }

int rr_reset(rr_env *e, const uint8_t *mask_host) {
    if (!e) return fail(RR_EINVAL, "null env");
    HIPCHK(hipSetDevice(e->cfg.device));
    const unsigned char *m = nullptr;
    if (mask_host) {
        HIPCHK(hipMemcpyAsync(e->mask_dev, mask_host, e->P.N, hipMemcpyHostToDevice, e->stream));
        m = e->mask_dev;
    }
    e->la_valid = false;          // the state changes from outside: the next step prepares itself in line
    hipLaunchKernelGGL(k_reset, dim3((e->P.N + 255) / 256), dim3(256), 0, e->stream, e->B, e->P, e->D, m);
    launch_obs(e);
    HIPCHK(hipGetLastError());
    return RR_OK;
}

// The pose an object returns to on reset and when it leaves the table (robot.py:19-24 `object_poses`, which callers of the
// reference edit in place, e.g. tests/test_actions.py:95-98). env_index < 0: every env.
int rr_set_object_home(rr_env *e, int32_t env_index, int32_t obj, const float *pose7) {
    if (!e || !pose7) return fail(RR_EINVAL, "null argument");
    if (env_index >= e->P.N || obj < 0 || obj >= NOBJ) return fail(RR_EINVAL, "rr_set_object_home: index out of range");
    HIPCHK(hipSetDevice(e->cfg.device));
    e->la_valid = false;          // the out-of-bounds rule of the look-ahead used the old home pose
    const size_t N = e->P.N;
    if (env_index >= 0) {
        for (int k = 0; k < 7; k++)
            HIPCHK(hipMemcpyAsync(e->D.obj_home + (size_t)(7 * obj + k) * N + env_index, pose7 + k, 4, hipMemcpyHostToDevice, e->stream));
    } else {
        std::vector<float> col(N);
        for (int k = 0; k < 7; k++) {
            std::fill(col.begin(), col.end(), pose7[k]);
            HIPCHK(hipMemcpyAsync(e->D.obj_home + (size_t)(7 * obj + k) * N, col.data(), N * 4, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
        }
    }
    HIPCHK(hipStreamSynchronize(e->stream));   // pose7 is host memory
    return RR_OK;
}

int rr_set_object_pose(rr_env *e, int32_t env_index, int32_t obj, const float *pose7) {
    if (!e || !pose7) return fail(RR_EINVAL, "null argument");
    if (env_index < 0 || env_index >= e->P.N || obj < 0 || obj >= e->P.nobj) return fail(RR_EINVAL, "rr_set_object_pose: index out of range");
    HIPCHK(hipSetDevice(e->cfg.device));
    e->la_valid = false;
    const size_t N = e->P.N;
    float zero = 0.0f;
    for (int k = 0; k < 3; k++) {
        HIPCHK(hipMemcpyAsync(e->D.state + (ST_OPOS + 3 * obj + k) * N + env_index, pose7 + k, 4, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->D.state + (ST_OVEL + 3 * obj + k) * N + env_index, &zero, 4, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->D.state + (ST_OANG + 3 * obj + k) * N + env_index, &zero, 4, hipMemcpyHostToDevice, e->stream));
    }
    for (int k = 0; k < 4; k++)
        HIPCHK(hipMemcpyAsync(e->D.state + (ST_OQUAT + 4 * obj + k) * N + env_index, pose7 + 3 + k, 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));   // pose7/zero are stack/host memory
    launch_obs(e);
    return RR_OK;
}

// Batched form of rr_set_object_pose: one upload and one kernel for the whole batch (set_goal of N envs, env.py:151-166).
int rr_set_object_poses(rr_env *e, const float *poses_host, const uint8_t *env_mask_host) {
    if (!e || !poses_host) return fail(RR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    e->la_valid = false;
    const int N = e->P.N;
    static_assert(NSTATE >= NOBJ * 7, "the state staging buffer doubles as pose staging");
    HIPCHK(hipMemcpyAsync(e->state_aos, poses_host, (size_t)N * e->P.nobj * 28, hipMemcpyHostToDevice, e->stream));
    const unsigned char *m = nullptr;
    if (env_mask_host) { HIPCHK(hipMemcpyAsync(e->mask_dev, env_mask_host, N, hipMemcpyHostToDevice, e->stream)); m = e->mask_dev; }
    hipLaunchKernelGGL(k_set_object_poses, dim3((N + 255) / 256), dim3(256), 0, e->stream, e->P, e->D, e->state_aos, m);
    launch_obs(e);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));   // the arguments are host memory
    return RR_OK;
}

static bool g_debug_sync = getenv("RR_DEBUG_SYNC") != nullptr;
static int g_skip = getenv("RR_SKIP") ? atoi(getenv("RR_SKIP")) : 0;
#define TIMED(id, launch)                                                   \
    do {                                                                    \
        if (e->timing) hipEventRecord(e->ev[2 * (id)], e->stream);          \
        if (!((g_skip >> (id)) & 1)) launch;                                \
        if (g_debug_sync) { hipError_t e__ = hipStreamSynchronize(e->stream); fprintf(stderr, "[rr] kernel %d done: %s\n", (id), hipGetErrorString(e__)); } \
        if (e->timing) {                                                    \
            hipEventRecord(e->ev[2 * (id) + 1], e->stream);                 \
            hipEventSynchronize(e->ev[2 * (id) + 1]);                       \
            float ms_ = 0;                                                  \
            hipEventElapsedTime(&ms_, e->ev[2 * (id)], e->ev[2 * (id) + 1]); \
            e->t_ms[id] += ms_; e->t_n[id] += 1;                            \
        }                                                                   \
    } while (0)

// Image set-up that precedes the first frame after create / a new static layer: full copy of the static layer (all envs).
static int ensure_images(rr_env *e, DevPtrs &D) {
    const int N = e->P.N;
    const ImageOut io = env_images(e);
    if (!e->images_valid || e->full_copy) {
        const int copy_blocks = std::min(16, (e->RM.W * e->RM.H / 4 + COPY_THREADS - 1) / COPY_THREADS);
        DevPtrs Dall = D;
        if (!e->images_valid) Dall.render_flags = nullptr;      // first frame: every env, flagged or not -- all images become valid
        TIMED(5, hipLaunchKernelGGL(k_static_copy, dim3(copy_blocks, std::min(N, 65535)), dim3(COPY_THREADS), 0, e->stream, e->RM_dev, Dall, io, 1, N));
        if (!e->images_valid) HIPCHK(hipMemsetAsync(e->D.frag_count, 0, (size_t)N * e->RM.ntiles * sizeof(unsigned), e->stream));
        e->images_valid = true;
        return 0;           // the lists are empty: nothing to restore
    }
    if (e->sep_restore) {
        TIMED(5, hipLaunchKernelGGL(k_restore, dim3(N, e->RM.ntiles), dim3(RESTORE_THREADS), 0, e->stream, e->RM_dev, D, io, 1));
        return 0;
    }
    return 1;
}

// The three render kernels for the envs selected by `sel` (env_selected) on `st`.  The images persist in HBM from frame to
// frame: only the pixels of the previous frame's fragment lists are put back to the static layer (`restore`), DESIGN.md 5.
// `setup_done`: the instances of these envs are set up already (by the light solve).
// Visibility pass of the envs selected by `sel` (all, or the light ones): one workgroup per (env, tile), dispatched in the order of the
// last frame's costs when the batch is large enough to matter; the shading launch behind it makes the next order.
static void launch_raster(rr_env *e, const DevPtrs &D, int restore, int sel, hipStream_t st) {
    const int N = e->P.N, nt = e->RM.ntiles;
    DevPtrs Do = D;
    Do.item_perm = e->item_perm && e->ord_valid ? e->item_perm : nullptr;
    hipLaunchKernelGGL(k_raster, Do.item_perm ? dim3((unsigned)(8 * ((N + 7) / 8) * nt)) : dim3(N, nt), dim3(RASTER_THREADS), 0, st, e->P, e->RM_dev, Do, e->n_inst_used, 0, 0, restore, sel);
    e->ord_pending = e->item_perm != nullptr;
}
static void launch_shade(rr_env *e, const DevPtrs &D, const ImageOut &io, int sel, hipStream_t st) {
    const int N = e->P.N;
    const bool ord = e->ord_pending && sel <= 1;       // (behind launch_raster on the same stream)
    hipLaunchKernelGGL(k_shade, dim3(N + (ord ? 1 : 0), e->RM.ntiles, SHADE_SPLIT), dim3(SHADE_THREADS), 0, st, e->RM_dev, D, io, 1, 0, sel, ord ? N : 0, e->item_perm);
    if (ord) { e->ord_pending = false; e->ord_valid = true; }
}
static void launch_render(rr_env *e, const DevPtrs &D, int restore, int sel, hipStream_t st, bool timed, bool setup_done = false) {
    const int N = e->P.N;
    const ImageOut io = env_images(e);
    if (timed) {
        if (!setup_done) TIMED(3, hipLaunchKernelGGL(k_render_setup, dim3((N * MAXINST + 63) / 64), dim3(64), 0, st, e->B, e->P, e->RM_dev, D, sel));
        TIMED(4, launch_raster(e, D, restore, sel, st));
        TIMED(6, launch_shade(e, D, io, sel, st));
    } else {
        // the heavy envs, a few (lagged host copy of their number: at most one item per workgroup): one list-walking launch for
        // set-up, visibility and shading -- the tail of the step's longest chain; many: the three kernels (the fused one needs
        // 128 VGPRs: two workgroups per CU, which a long list pays for)
        if (sel == 3 || (sel == 2 && (long long)lagged_count(e, 0, e->P.N) * e->RM.ntiles <= RENDER_LIST_WGS)) {
            hipLaunchKernelGGL(k_render_list, dim3(std::min(N * e->RM.ntiles, sel == 3 ? 256 : RENDER_LIST_WGS)), dim3(RASTER_THREADS), 0, st, e->B, e->P, e->RM_dev, D, io, e->n_inst_used, restore, sel == 3 ? 1 : 0);
            return;
        }
        if (!setup_done) hipLaunchKernelGGL(k_render_setup, dim3((N * MAXINST + 63) / 64), dim3(64), 0, st, e->B, e->P, e->RM_dev, D, sel);
        if (sel == 2) hipLaunchKernelGGL(k_raster_list, dim3(std::min(N * e->RM.ntiles, RASTER_LIST_WGS)), dim3(RASTER_THREADS), 0, st, e->P, e->RM_dev, D, e->n_inst_used, restore, 0);
        else launch_raster(e, D, restore, sel, st);
        launch_shade(e, D, io, sel, st);
    }
}

static int do_render(rr_env *e, bool use_flags) {
    DevPtrs D = e->D;
    if (!use_flags) D.render_flags = nullptr;
    const int restore = ensure_images(e, D);
    launch_render(e, D, restore, 0, e->stream, true);
    HIPCHK(hipGetLastError());
    return RR_OK;
}

// ---- look-ahead: the state part of the NEXT step (k_prep_a -> k_collide, k_prep_b beside them) ------------------------------
// Per-class launches (sel: pick_env) cover N work items whatever the class.
#define COOP_ALL_MAX 1024   // up to this many envs a step that solves all envs in one launch gives every env its own wave (RR_COOP_ALL=0: four to a wave)
#define SMALL_N_MAX 64      // up to this many envs a step without the three-stream split runs as one chain on the main stream (rr_step)
static void launch_prep_a(rr_env *e, int sel, int zero_counts, hipStream_t st) {
    hipLaunchKernelGGL(k_prep_a, env_grid(e), dim3(e->epb), 0, st, e->B, e->P, e->D, sel, zero_counts);
}
static void launch_prep_b(rr_env *e, int sel, hipStream_t st) {
    hipLaunchKernelGGL(k_prep_b, env_grid(e), dim3(e->epb), 0, st, e->B, e->P, e->D, sel);
}
static void launch_collide(rr_env *e, int sel, hipStream_t st) {
    // (h_first: the lagged host copy of the two list lengths + a margin; RR_COLLIDE_ORDER=0: env order)
    int h_first = 0;
    if (sel == 0 && e->collide_ordered && e->h_hcount) {
        const int lag = lagged_count(e, 0, 0) + lagged_count(e, 1, 0);
        if (lag > 0) h_first = std::min(e->P.N, lag + 64);
    }
    hipLaunchKernelGGL(k_collide, dim3(e->P.N + h_first), dim3(COLLIDE_THREADS), 0, st, e->P, e->D, e->n_shapes, sel, h_first);
}
static void launch_prep_ab(rr_env *e, int sel, hipStream_t st) {
    hipLaunchKernelGGL(k_prep_ab, env_grid(e), dim3(e->epb), 0, st, e->B, e->P, e->D, sel);
}
static void launch_prep_serial(rr_env *e, int sel, int zero_counts) {
    launch_prep_a(e, sel, zero_counts, e->stream);
    launch_prep_b(e, sel, e->stream);
}

// The solve of the heavy (sel 2) / very heavy (sel 3) envs on `st`.  A list the lagged host count puts at <= COOP_MAX entries is
// launched in the coop form: one env per wave, four waves per workgroup with one LDS region each (N waves: whatever the
// list's actual length, every entry has its wave; the others exit at once); a longer one four envs to a wave.
static void launch_solve_class(rr_env *e, int sel, hipStream_t st) {
    const int N = e->P.N;
    const int ngroups = (N + SGRP - 1) / SGRP;
    const size_t lds64 = (size_t)SGRP * LF_TOTAL * sizeof(float);
    const int lagged = lagged_count(e, sel == 2 ? 0 : 1, N);
    const bool coop = e->P.coop_build && lagged <= COOP_MAX;
    if (coop) hipLaunchKernelGGL(k_solve, dim3((N + 3) / 4), dim3(256), lds64, st, e->B, e->P, e->D, sel, 1);
    else hipLaunchKernelGGL(k_solve, dim3((ngroups + 3) / 4), dim3(256), 4 * lds64, st, e->B, e->P, e->D, sel, 0);
}

// The state part of a step for all envs on the main stream (k_prep_b beside k_collide on the side stream when `overlap`): at
// the start of a step whose look-ahead is missing or stale, or at the end of a step that has a single class.
static void state_part_all(rr_env *e, bool overlap) {
    if (overlap) {
        launch_prep_a(e, 0, 1, e->stream);
        hipEventRecord(e->ev_fork, e->stream);
        hipStreamWaitEvent(e->aux, e->ev_fork, 0);
        launch_prep_b(e, 0, e->aux);
        hipEventRecord(e->ev_dyn, e->aux);
        launch_collide(e, 0, e->stream);
        hipStreamWaitEvent(e->stream, e->ev_dyn, 0);
    } else {
        TIMED(0, launch_prep_serial(e, 0, 1));
        TIMED(1, launch_collide(e, 0, e->stream));
    }
}

// Next slot of the pinned ring (allocated on first use: N * 37 bytes per slot = commands + render flags).
static int pin_acquire(rr_env *e, char **slot, int *idx) {
    const int i = e->pin_next;
    e->pin_next = (i + 1) & 3;
    if (!e->pin_buf[i]) {
        e->pin_bytes = (size_t)e->P.N * 37;
        HIPCHK(hipHostMalloc((void **)&e->pin_buf[i], e->pin_bytes, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&e->pin_ev[i], hipEventDisableTiming));
    }
    if (e->pin_used[i]) HIPCHK(hipEventSynchronize(e->pin_ev[i]));
    *slot = e->pin_buf[i]; *idx = i;
    return RR_OK;
}

int rr_step(rr_env *e, const float *joint_cmd, int32_t cmd_on_device, int32_t render_mode, const uint8_t *render_flags_host) {
    if (!e) return fail(RR_EINVAL, "null env");
    if (render_mode < 0 || render_mode > 2 || (render_mode == 2 && !render_flags_host)) return fail(RR_EINVAL, "rr_step: bad render_mode");
    HIPCHK(hipSetDevice(e->cfg.device));
    const int N = e->P.N;
    const bool host_cmd = joint_cmd && !cmd_on_device, host_flags = render_mode == 2;
    char *pin = nullptr; int pin_idx = -1;
    if (host_cmd || host_flags) { const int rc = pin_acquire(e, &pin, &pin_idx); if (rc != RR_OK) return rc; }
    if (!joint_cmd) HIPCHK(hipMemsetAsync(e->D.cmd, 0, (size_t)N * 36, e->stream));      // env.py:333-334
    else if (host_cmd) {
        memcpy(pin, joint_cmd, (size_t)N * 36);
        HIPCHK(hipMemcpyAsync(e->D.cmd, pin, (size_t)N * 36, hipMemcpyHostToDevice, e->stream));
    }
    if (host_flags) {
        memcpy(pin + (size_t)N * 36, render_flags_host, N);
        HIPCHK(hipMemcpyAsync(e->D.render_flags, pin + (size_t)N * 36, N, hipMemcpyHostToDevice, e->stream));
    }
    if (pin_idx >= 0) { HIPCHK(hipEventRecord(e->pin_ev[pin_idx], e->stream)); e->pin_used[pin_idx] = true; }
    const bool overlap = e->aux && !e->timing && !g_skip;      // side streams in use (timing leg / diagnostics: everything on the main stream)
    // ---- the state part of this step, unless the previous step already computed it (look-ahead) for exactly this state
    if (!e->la_valid) state_part_all(e, overlap);
    e->cur ^= 1; e->la_valid = false;           // the frame the collision pass filled is the one this step solves
    bind_frames(e);
    // a device-resident command buffer is read in place by the solve kernels (stream order protects it like a copy would)
    e->D.cmd_in = (joint_cmd && cmd_on_device) ? joint_cmd : e->D.cmd;
    const int ngroups = (N + SGRP - 1) / SGRP;
    const size_t lds64 = (size_t)SGRP * LF_TOTAL * sizeof(float);
    // (the number of heavy envs of a recent step, written to pinned host memory by the solve kernel without anybody waiting for it: when most are
    // heavy -- macro actions, every gripper pushing -- there is nothing to gain from the split)
    const bool mostly_heavy = (long long)lagged_count(e, 0, 0) * 100 > (long long)N * e->split_max_pct;
    const bool ahead = e->lookahead;            // this step ends with the state part of the next one
    // (a step without camera runs all envs in one launch: its classes side by side were measured -- config 2: 0.525 instead of 0.452 ms)
    // (ONE env -- the gym facade -- renders in its chain on the main stream too: solve -> set-up, visibility, shading -> mirror and
    // image copies -> look-ahead; the three-stream split has nothing to overlap there and costs six events: 119 -> 95 us per step)
    if (e->aux && !g_skip && render_mode && e->split_heavy && !mostly_heavy && !(N == 1 && !e->timing)) {
        // The few envs with generic contact rows take several times as long as the others (the kernel lasts as long
        // as its longest Gauss-Seidel chain).  They are solved and rendered on the side streams -- four groups per 256-thread
        // workgroup, so that they fill the LDS of a few CUs and leave the rest to the raster workgroups of the light envs --
        // while the main stream solves and renders everybody else.
        DevPtrs D = e->D;
        if (render_mode != 2) D.render_flags = nullptr;
        const int restore = ensure_images(e, D);
        // (the light solve sets up the render instances of its envs itself: one launch and a 20 us kernel less at the head of
        // the step's main chain; RR_NO_FUSED_SETUP: the separate k_render_setup launch -- same bits, tested)
        const RenderModel *fused_rm = e->no_fused_setup ? nullptr : e->RM_dev;
        const bool light_ow = e->light_ow;
        if (e->timing) {
            // timing leg: the very same launches, one after the other on the main stream, each under its timer -- 2 / 3 / 4 / 6
            // what the main stream runs in an untimed step (the light envs), 7 / 8 what the side streams run beside it, 0 / 1 the
            // look-ahead of the next step, which an untimed step runs on the heavy stream behind the heavy envs' render
            if (light_ow) TIMED(2, hipLaunchKernelGGL(k_solve_light_ow, dim3((N + 15) / 16), dim3(LIGHT_OW_THREADS), 4 * lds64, e->stream, e->B, e->P, e->D, fused_rm));
            else TIMED(2, hipLaunchKernelGGL(k_solve_light, dim3(ngroups), dim3(SGRP * 16), lds64, e->stream, e->B, e->P, e->D, fused_rm));
            TIMED(7, { launch_solve_class(e, 2, e->stream); launch_solve_class(e, 3, e->stream); });
            launch_render(e, D, restore, 1, e->stream, true, fused_rm != nullptr);
            TIMED(8, { launch_render(e, D, restore, 2, e->stream, false); launch_render(e, D, restore, 3, e->stream, false); });
            if (ahead) {
                TIMED(0, launch_prep_ab(e, 0, e->stream));
                TIMED(1, launch_collide(e, 0, e->stream));
                e->la_valid = true;
            }
            launch_mirror(e, render_mode != 0);
            HIPCHK(hipGetLastError());
            return RR_OK;
        }
        // Look-ahead: the state part of the NEXT step for all envs, once every solve of this step has finished.
        // Where it goes (all measured, profiles/README.md):
        //  * a handful of very heavy envs (lagged count <= 64): on their stream right behind their solve -- the last solve of
        //    the step to finish -- while their render moves to the heavy stream, behind the heavy envs' render.  The collision
        //    pass then runs beside the end of the visibility pass and the shading instead of after them (0.728 instead of
        //    0.758 ms; behind the very heavy envs' render it was the tail of the step's longest chain);
        //  * hundreds of very heavy envs (macro actions): their solve + render is the longest chain; the look-ahead goes to the
        //    tail of the main stream.
        // (both heavy lists rendered by one launch once both solves are done: 0.751 ms -- the heavy envs' render then waits
        // for the very heavy envs' solve.)
        // (per-class look-aheads beside the visibility pass only queue up behind the LDS-filling raster workgroups; a fourth
        // stream for it slows every kernel of the step down -- one more hardware queue: 1.19 instead of 0.81 ms)
        const bool la_on_vh = e->la_vh_max >= 0 && lagged_count(e, 1, 0) <= e->la_vh_max;       // (RR_LA_VH_MAX=-1 forces the other placement)
        const bool vh_render_on_h = ahead && la_on_vh;
        hipEventRecord(e->ev_fork, e->stream);
        hipStreamWaitEvent(e->aux, e->ev_fork, 0);
        launch_solve_class(e, 2, e->aux);
        if (ahead) hipEventRecord(e->ev_hsolved, e->aux);
        launch_render(e, D, restore, 2, e->aux, false);
        hipStreamWaitEvent(e->aux2, e->ev_fork, 0);
        launch_solve_class(e, 3, e->aux2);
        if (ahead) hipEventRecord(e->ev_vsolved, e->aux2);
        // (the very heavy envs' render: behind the heavy envs' on their stream -- unless that list is a long one (three launches, the
        // longest chain of the step in the late window of the benchmark workload): then at the tail of the main stream, which is
        // done with the shading by then.  With a short heavy list the main stream's tail measured 0.733 instead of 0.727 ms.)
        const bool h_long = (long long)lagged_count(e, 0, 0) * e->RM.ntiles > RENDER_LIST_WGS;
        const bool vh_render_on_main = vh_render_on_h && (e->vh_main < 0 ? h_long : e->vh_main == 1);     // RR_VH_ON_MAIN: -1 by the list's length, 0 never, 1 always
        if (vh_render_on_h && !vh_render_on_main) {
            hipStreamWaitEvent(e->aux, e->ev_vsolved, 0);
            launch_render(e, D, restore, 3, e->aux, false);
        } else if (!vh_render_on_h) launch_render(e, D, restore, 3, e->aux2, false);
        // (hundreds of very heavy envs -- macro actions: RR_MACRO_LA=0 keeps the look-ahead at the tail of the main stream)
        const bool la_split_side = ahead && !la_on_vh && e->macro_la_side;
        if (!la_split_side) hipEventRecord(e->ev_join, e->aux);
        if (light_ow) hipLaunchKernelGGL(k_solve_light_ow, dim3((N + 15) / 16), dim3(LIGHT_OW_THREADS), 4 * lds64, e->stream, e->B, e->P, e->D, fused_rm);
        else hipLaunchKernelGGL(k_solve_light, dim3(ngroups), dim3(SGRP * 16), lds64, e->stream, e->B, e->P, e->D, fused_rm);
        if (ahead && la_on_vh) {
            hipEventRecord(e->ev_dyn, e->stream);             // the light envs' solve
            hipStreamWaitEvent(e->aux2, e->ev_dyn, 0);
            hipStreamWaitEvent(e->aux2, e->ev_hsolved, 0);    // the heavy envs' solve
            // (k_prep_ab needs a whole free SIMD for each of its 64 waves -- 256 VGPRs + AGPRs -- and sits in its queue until the
            // shading's grid is exhausted: 55-100 us between the last solve and the collision pass.  Measured against it: the
            // kinematics half alone (69 VGPRs: it moves into the holes retiring render workgroups leave, 21 us) in front of the
            // collision pass and the dynamics half behind it (0.685 ms) or on the main stream behind the shading (0.671)
            // instead of 0.672 -- the collision pass then runs beside the shading and takes as much longer.)
            hipLaunchKernelGGL(k_prep_ab, env_grid(e), dim3(e->epb), 0, e->aux2, e->B, e->P, e->D, 0);
            launch_collide(e, 0, e->aux2);
        }
        if (la_split_side) {
            // the heavy stream is done with its envs' render long before the very heavy envs' is (their solve lasts twice as long):
            // the kinematics half of the next step's preparation (69 VGPRs: it gets onto the machine beside the renders) and the
            // collision pass go there, once every solve is done; the dynamics half (a whole SIMD per wave) behind the very heavy
            // envs' render
            hipEventRecord(e->ev_dyn, e->stream);             // the light envs' solve
            hipStreamWaitEvent(e->aux, e->ev_dyn, 0);
            hipStreamWaitEvent(e->aux, e->ev_vsolved, 0);
            launch_prep_a(e, 0, 0, e->aux);
            launch_collide(e, 0, e->aux);
            hipEventRecord(e->ev_join, e->aux);
            hipStreamWaitEvent(e->aux2, e->ev_dyn, 0);
            hipStreamWaitEvent(e->aux2, e->ev_hsolved, 0);
            launch_prep_b(e, 0, e->aux2);
        }
        hipEventRecord(e->ev_join2, e->aux2);
        launch_render(e, D, restore, 1, e->stream, false, fused_rm != nullptr);
        if (vh_render_on_main) {
            hipStreamWaitEvent(e->stream, e->ev_vsolved, 0);
            launch_render(e, D, restore, 3, e->stream, false);
        }
        if (ahead && !la_on_vh && !la_split_side) {
            hipStreamWaitEvent(e->stream, e->ev_hsolved, 0);
            hipStreamWaitEvent(e->stream, e->ev_vsolved, 0);
            hipLaunchKernelGGL(k_prep_ab, env_grid(e), dim3(e->epb), 0, e->stream, e->B, e->P, e->D, 0);
            launch_collide(e, 0, e->stream);
        }
        if (ahead) e->la_valid = true;
        hipStreamWaitEvent(e->stream, e->ev_join, 0);
        hipStreamWaitEvent(e->stream, e->ev_join2, 0);
        launch_mirror(e, render_mode != 0);
        HIPCHK(hipGetLastError());
        return RR_OK;
    }
    // A handful of envs (the single-env gym facade, BASELINE config 1): the step is one latency chain -- solve -> preparation ->
    // collision pass, 72 + 34 + 13 us at N = 1 (scratch/small_n.py) -- and nothing overlaps with anything.  The classes are then
    // solved one after the other on the main stream by their own kernels (the light form is 41 us instead of the generic kernel's
    // 72; a launch whose list is empty ends at once: a single env is in exactly one class), and up to SMALL_N_MAX envs the look-ahead
    // runs as two launches on the same stream, no events.
    const bool small_n = N <= SMALL_N_MAX && e->split_heavy && e->aux && !e->timing && !g_skip;
    if (small_n && N == 1) {     // (one env is in exactly one class; with several, the classes side by side in the generic kernel are faster: 16 envs 0.67 vs 0.81 ms)
        if (e->light_ow) hipLaunchKernelGGL(k_solve_light_ow, dim3((N + 15) / 16), dim3(LIGHT_OW_THREADS), 4 * lds64, e->stream, e->B, e->P, e->D, (const RenderModel *)nullptr);
        else hipLaunchKernelGGL(k_solve_light, dim3(ngroups), dim3(SGRP * 16), lds64, e->stream, e->B, e->P, e->D, (const RenderModel *)nullptr);
        launch_solve_class(e, 2, e->stream);
        launch_solve_class(e, 3, e->stream);
    } else if (e->coop_all && N <= COOP_ALL_MAX && e->split_heavy)
        // (up to one wave per SIMD of the chip -- BASELINE config 2's 1024 envs: every env gets a wave of its own, whose four groups
        // build its rows side by side; four envs to a wave is the form for batches that would not fit the machine otherwise)
        TIMED(2, hipLaunchKernelGGL(k_solve, dim3((N + 3) / 4), dim3(256), lds64, e->stream, e->B, e->P, e->D, 0, 1));
    else
        TIMED(2, hipLaunchKernelGGL(k_solve, dim3(ngroups), dim3(SGRP * 16), lds64, e->stream, e->B, e->P, e->D, 0, 0));
    HIPCHK(hipGetLastError());
    int rc = RR_OK;
    // (one class.  With a camera the state part of the next step runs on the side stream beside the render of this one;
    // RR_UNSPLIT_LA_INLINE=1: behind it on the main stream.)
    const bool la_beside = ahead && render_mode && overlap && !e->la_inline && !small_n;
    if (la_beside) {
        hipEventRecord(e->ev_fork, e->stream);
        hipStreamWaitEvent(e->aux, e->ev_fork, 0);
        launch_prep_ab(e, 0, e->aux);
        launch_collide(e, 0, e->aux);
        hipEventRecord(e->ev_join, e->aux);
    }
    if (render_mode) rc = do_render(e, render_mode == 2);
    if (la_beside) {
        hipStreamWaitEvent(e->stream, e->ev_join, 0);
        e->la_valid = true;
    } else if (ahead) {
        // (a handful of envs: the observations of this step are complete here -- the mirror goes in front of the look-ahead, so that
        // a caller waiting for the observations alone (rr_sync_observations) gets them 45 us earlier and the state part of the next
        // step runs while the host computes its next action)
        if (small_n) { launch_mirror(e, render_mode != 0); launch_prep_ab(e, 0, e->stream); launch_collide(e, 0, e->stream); }
        else state_part_all(e, overlap);
        e->la_valid = true;
    }
    if (!(small_n && ahead && !la_beside)) launch_mirror(e, render_mode != 0);
    return rc;
}

int rr_render(rr_env *e) {
    if (!e) return fail(RR_EINVAL, "null env");
    HIPCHK(hipSetDevice(e->cfg.device));
    const int rc = do_render(e, false);
    launch_mirror(e, true);
    return rc;
}

int rr_get_buffer(rr_env *e, int32_t field, void **dev_ptr, size_t *bytes) {
    if (!e || field < 0 || field >= RR_F_COUNT) return fail(RR_EINVAL, "rr_get_buffer: bad field");
    refresh_frame_fields(e);
    if (dev_ptr) *dev_ptr = e->field_ptr[field];
    if (bytes) *bytes = e->field_bytes[field];
    return RR_OK;
}

int rr_copy_to_host(rr_env *e, int32_t field, void *dst, size_t bytes) {
    if (!e || !dst || field < 0 || field >= RR_F_COUNT) return fail(RR_EINVAL, "rr_copy_to_host: bad argument");
    refresh_frame_fields(e);
    if (!e->field_ptr[field]) return fail(RR_EINVAL, "rr_copy_to_host: field not available (RR_FLAG_NO_MASK)");
    if (bytes != e->field_bytes[field]) return fail(RR_EINVAL, "rr_copy_to_host: size mismatch");
    HIPCHK(hipSetDevice(e->cfg.device));
    if (field == RR_F_STATE)
        hipLaunchKernelGGL(k_state_io, dim3((e->P.N + 255) / 256), dim3(256), 0, e->stream, e->P, e->D, e->state_aos, 1);
    HIPCHK(hipMemcpyAsync(dst, e->field_ptr[field], bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}

int rr_set_state(rr_env *e, const float *state_host) {
    if (!e || !state_host) return fail(RR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    e->la_valid = false;
    HIPCHK(hipMemcpyAsync(e->state_aos, state_host, e->field_bytes[RR_F_STATE], hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_state_io, dim3((e->P.N + 255) / 256), dim3(256), 0, e->stream, e->P, e->D, e->state_aos, 0);
    launch_obs(e);
    HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}

// ---- checkpoint: everything a later restore needs to continue bit for bit ----------------------------------------------------
// {header, state slab [72][N] (incl. motor targets), contact count [N], contact list [N][48][3] float4, normal forces [N][48],
//  timestep [N], errflags [N], touch [N][4], object home poses [21][N]}: the 61-float state of RR_F_STATE plus the contact
// history of the warm start (Bullet: the persistent manifolds with their cached impulses) and the episode clocks.
// The header also carries every parameter the continuation depends on: a blob restored into a handle that steps differently
// (other dt / ERP / margin / sweeps / warm-start factor / object-lane capacity / inertia source / edge contacts) is rejected
// instead of silently diverging.
struct CkptHeader { char magic[8]; int32_t version, N, nobj, iters; float dt, erp, margin, warmstart; int32_t os_cap, edge_contacts, urdf_inertia, reserved; };
static CkptHeader ckpt_header(const rr_env *e) {
    CkptHeader hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, "RRCKPT02", 8); hd.version = 2; hd.N = e->P.N; hd.nobj = e->P.nobj; hd.iters = e->P.iters;
    hd.dt = e->P.dt; hd.erp = e->P.erp; hd.margin = e->P.margin; hd.warmstart = e->P.warmstart;
    hd.os_cap = e->P.os_cap; hd.edge_contacts = e->P.edge_contacts; hd.urdf_inertia = e->cfg.use_urdf_inertia;
    return hd;
}
static size_t ckpt_bytes(const rr_env *e) {
    const size_t N = e->P.N;
    return sizeof(CkptHeader) + 4 * (ST_TOTAL * N + N + N * MAXC * 12 + N * MAXC + N + N + N * 4 + NOBJ * 7 * N);
}
int rr_checkpoint_bytes(rr_env *e, size_t *bytes) {
    if (!e || !bytes) return fail(RR_EINVAL, "null argument");
    *bytes = ckpt_bytes(e);
    return RR_OK;
}
static int ckpt_copy(rr_env *e, char *host, bool save) {
    const size_t N = e->P.N;
    struct Part { void *dev; size_t bytes; } parts[] = {
        {e->D.state, 4 * ST_TOTAL * N}, {e->D.ccount, 4 * N}, {e->D.clist, 4 * N * MAXC * 12}, {e->D.cforce, 4 * N * MAXC},
        {e->D.timestep, 4 * N}, {e->D.errflags, 4 * N}, {e->D.touch, 4 * N * 4}, {e->D.obj_home, 4 * NOBJ * 7 * N}};
    char *h = host + sizeof(CkptHeader);
    for (const Part &p : parts) {
        if (save) HIPCHK(hipMemcpyAsync(h, p.dev, p.bytes, hipMemcpyDeviceToHost, e->stream));
        else HIPCHK(hipMemcpyAsync(p.dev, h, p.bytes, hipMemcpyHostToDevice, e->stream));
        h += p.bytes;
    }
    return RR_OK;
}
int rr_checkpoint_save(rr_env *e, void *dst_host, size_t bytes) {
    if (!e || !dst_host) return fail(RR_EINVAL, "null argument");
    if (bytes != ckpt_bytes(e)) return fail(RR_EINVAL, "rr_checkpoint_save: size mismatch (rr_checkpoint_bytes)");
    HIPCHK(hipSetDevice(e->cfg.device));
    const CkptHeader hd = ckpt_header(e);
    memcpy(dst_host, &hd, sizeof hd);
    const int rc = ckpt_copy(e, (char *)dst_host, true);
    if (rc != RR_OK) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}
int rr_checkpoint_restore(rr_env *e, const void *src_host, size_t bytes) {
    if (!e || !src_host) return fail(RR_EINVAL, "null argument");
    if (bytes != ckpt_bytes(e)) return fail(RR_EINVAL, "rr_checkpoint_restore: size mismatch (rr_checkpoint_bytes)");
    CkptHeader hd;
    memcpy(&hd, src_host, sizeof hd);
    const CkptHeader mine = ckpt_header(e);
    if (memcmp(hd.magic, mine.magic, 8) != 0 || hd.version != mine.version || hd.N != mine.N || hd.nobj != mine.nobj)
        return fail(RR_EINVAL, "rr_checkpoint_restore: not a checkpoint of an env handle of this shape");
    if (memcmp(&hd, &mine, sizeof hd) != 0)
        return fail(RR_EINVAL, "rr_checkpoint_restore: the checkpoint was taken with other step parameters (dt / erp / margin / solver_iters / warm start / object-lane capacity / inertia source / edge contacts)");
    HIPCHK(hipSetDevice(e->cfg.device));
    e->la_valid = false;
    const int rc = ckpt_copy(e, (char *)const_cast<void *>(src_host), false);
    if (rc != RR_OK) return rc;
    // the published count follows the restored list; the class diagnostic and k_collide's launch order start clean (the lists of
    // the handle's own run say nothing about the restored one)
    const size_t N = e->P.N;
    HIPCHK(hipMemcpyAsync(e->D.ccount_pub, e->D.ccount, 4 * N, hipMemcpyDeviceToDevice, e->stream));
    HIPCHK(hipMemsetAsync(e->D.class_pub, 0, 4 * N, e->stream));
    HIPCHK(hipMemsetAsync(e->D.hgflag, 0, 4 * N, e->stream));
    HIPCHK(hipMemsetAsync(e->D.hcount, 0, 16, e->stream)); HIPCHK(hipMemsetAsync(e->D.hcount2, 0, 16, e->stream));
    launch_obs(e);
    HIPCHK(hipStreamSynchronize(e->stream));   // the source is host memory
    return RR_OK;
}


// ---- device micro-benchmarks for bench.py's roofline ("achievable" measured in the same run) -------------------
// kind 0: HBM copy (read + write), 1: HBM triad a = b + s c (2 reads + 1 write), 256 MiB per array, float4 per lane, grid-stride;
// kind 2: VALU issue rate of a sample-test-like mix (sub, mul, fma, cmp, cndmask with real dependencies) at the raster kernel's
// shape -- 512-thread workgroups, 39 KB of LDS each, four per CU = eight waves per SIMD (tools/ubench/valu_issue.hip sweeps
// waves per SIMD and instruction kinds; profiles/r04_valu_issue.txt).  Result: GB/s (0, 1) or G wave64-instructions/s (2).
__global__ void __launch_bounds__(256) k_ub_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ void __launch_bounds__(256) k_ub_triad(float4 *__restrict__ a, const float4 *__restrict__ b, const float4 *__restrict__ c, float s, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 x = b[i], y = c[i];
        a[i] = make_float4(fmaf(s, y.x, x.x), fmaf(s, y.y, x.y), fmaf(s, y.z, x.z), fmaf(s, y.w, x.w));
    }
}
#define UB_S8(x) x x x x x x x x
__global__ void __launch_bounds__(512) k_ub_valu(float *out, int iters, float a, float b) {
    extern __shared__ float ub_lds[];
    float x0 = threadIdx.x * a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int it = 0; it < iters; it++) {
        asm volatile(UB_S8("v_sub_f32 %0, %1, %8\n v_mul_f32 %2, %0, %9\n v_fma_f32 %3, %2, %8, %0\n v_cmp_lt_f32 vcc, %3, %9\n"
                           "v_cndmask_b32 %4, %5, %6, vcc\n v_sub_f32 %5, %7, %9\n v_fma_f32 %6, %4, %8, %5\n v_mul_f32 %7, %6, %8\n")
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b) : "vcc");
    }
    const float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (r == 12345.678f) ub_lds[threadIdx.x] = r;
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = r;
}
int rr_device_microbench(int32_t device, int32_t kind, double *result) {
    if (!result || kind < 0 || kind > 2) return fail(RR_EINVAL, "rr_device_microbench: bad argument");
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return fail(RR_EDEVICE, "rr_device_microbench: no such HIP device (no CPU fallback)"); }
    // (everything the measurement allocates is released on every path, it runs on a stream of its own, and the caller's current
    // device is put back)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t st = nullptr;
    void *buf[3] = {nullptr, nullptr, nullptr};
    hipError_t err = hipSuccess;
    float best_ms = 1e30f;
    double units = 0.0;
#define UB(call) do { if (err == hipSuccess) err = (call); } while (0)
    hipDeviceProp_t prop;
    UB(hipGetDeviceProperties(&prop, device));
    const int ncu = err == hipSuccess ? prop.multiProcessorCount : 1;
    UB(hipEventCreate(&e0)); UB(hipEventCreate(&e1)); UB(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (kind <= 1) {
        const size_t bytes = (size_t)256 << 20, n4 = bytes / 16;
        for (int i = 0; i < (kind == 0 ? 2 : 3); i++) { UB(hipMalloc(&buf[i], bytes)); UB(hipMemsetAsync(buf[i], 0, bytes, st)); }
        for (int rep = 0; rep < 6 && err == hipSuccess; rep++) {
            UB(hipEventRecord(e0, st));
            if (kind == 0) hipLaunchKernelGGL(k_ub_copy, dim3(ncu * 16), dim3(256), 0, st, (const float4 *)buf[0], (float4 *)buf[1], n4);
            else hipLaunchKernelGGL(k_ub_triad, dim3(ncu * 16), dim3(256), 0, st, (float4 *)buf[0], (const float4 *)buf[1], (const float4 *)buf[2], 0.5f, n4);
            UB(hipEventRecord(e1, st));
            UB(hipEventSynchronize(e1));
            float ms = 0; UB(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && err == hipSuccess) best_ms = std::min(best_ms, ms);
        }
        units = (double)bytes * (kind == 0 ? 2 : 3) / 1e9;                 // GB moved per launch
    } else {
        const int iters = 800, blocks = 4 * ncu;
        UB(hipMalloc(&buf[0], (size_t)blocks * 512 * 4));
        UB(hipFuncSetAttribute((const void *)k_ub_valu, hipFuncAttributeMaxDynamicSharedMemorySize, 39 * 1024));
        for (int rep = 0; rep < 4 && err == hipSuccess; rep++) {
            UB(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_ub_valu, dim3(blocks), dim3(512), 39 * 1024, st, (float *)buf[0], iters, 1.0001f, 0.5f);
            UB(hipEventRecord(e1, st));
            UB(hipEventSynchronize(e1));
            float ms = 0; UB(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && err == hipSuccess) best_ms = std::min(best_ms, ms);
        }
        units = (double)iters * 64 * blocks * 8 / 1e9;                     // G wave-instructions per launch
    }
    UB(hipGetLastError());
#undef UB
    for (int i = 0; i < 3; i++) if (buf[i]) (void)hipFree(buf[i]);
    if (st) (void)hipStreamDestroy(st);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
    if (err != hipSuccess) { (void)hipGetLastError(); return fail(RR_EDEVICE, std::string("rr_device_microbench: ") + hipGetErrorString(err)); }
    *result = units / (best_ms * 1e-3);
    return RR_OK;
}

int rr_map_observations(rr_env *e, void **host_ptr, size_t *bytes) {
    if (!e || !host_ptr) return fail(RR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t N = e->P.N, nobj = e->P.nobj;
    const size_t nb = 4 * (N * 9 + N * 4 + N * nobj * 7 + N + N);
    if (!e->obs_host) {
        void *h = nullptr, *d = nullptr;
        HIPCHK(hipHostMalloc(&h, nb, hipHostMallocMapped));
        memset(h, 0, nb);
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { hipHostFree(h); (void)hipGetLastError(); return fail(RR_EDEVICE, "rr_map_observations: pinned host memory is not device-mapped on this system"); }
        float *f = (float *)d;
        e->obs_dev.joints = f; e->obs_dev.touch = f + N * 9; e->obs_dev.objpose = f + N * 13;
        e->obs_dev.timestep = (int *)(f + N * (13 + nobj * 7)); e->obs_dev.errflags = (unsigned *)(f + N * (14 + nobj * 7));
        e->obs_host = h;
        launch_mirror(e);
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    *host_ptr = e->obs_host;
    if (bytes) *bytes = nb;
    return RR_OK;
}

int rr_sync_observations(rr_env *e) {
    if (!e) return fail(RR_EINVAL, "null env");
    HIPCHK(hipSetDevice(e->cfg.device));
    if (e->ev_obs_set) HIPCHK(hipEventSynchronize(e->ev_obs)); else HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}

int rr_map_images(rr_env *e, void **rgb_host, void **depth_host, void **mask_host) {
    if (!e) return fail(RR_EINVAL, "null env");
    HIPCHK(hipSetDevice(e->cfg.device));
    void **out[3] = {rgb_host, depth_host, mask_host};
    const int f[3] = {RR_F_RGB, RR_F_DEPTH, RR_F_MASK};
    size_t total = 0;
    for (int i = 0; i < 3; i++) if (out[i]) total += e->field_bytes[f[i]];
    if (total > ((size_t)256 << 20)) return fail(RR_EINVAL, "rr_map_images: more than 256 MiB of images per step (meant for a handful of envs; batches read the device buffers)");
    for (int i = 0; i < 3; i++) {
        if (!out[i]) continue;
        if (!e->field_ptr[f[i]]) return fail(RR_EINVAL, "rr_map_images: field not available (RR_FLAG_NO_MASK)");
        if (!e->img_host[i]) {
            HIPCHK(hipHostMalloc(&e->img_host[i], e->field_bytes[f[i]], hipHostMallocDefault));
            HIPCHK(hipMemcpyAsync(e->img_host[i], e->field_ptr[f[i]], e->field_bytes[f[i]], hipMemcpyDeviceToHost, e->stream));
        }
        *out[i] = e->img_host[i];
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}

int rr_sync(rr_env *e) {
    if (!e) return fail(RR_EINVAL, "null env");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}

int rr_link_poses(rr_env *e, float *out_host) {
    if (!e || !out_host) return fail(RR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    hipLaunchKernelGGL(k_link_poses, dim3((e->P.N + 63) / 64), dim3(64), 0, e->stream, e->B, e->P, e->RM_dev, e->D, e->link_out);
    HIPCHK(hipMemcpyAsync(out_host, e->link_out, (size_t)e->P.N * e->RM.nl * 28, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}

int rr_get_contacts(rr_env *e, int32_t env_index, float *out_host, int32_t max_contacts, int32_t *count) {
    if (!e || !out_host || !count) return fail(RR_EINVAL, "null argument");
    if (env_index < 0 || env_index >= e->P.N) return fail(RR_EINVAL, "rr_get_contacts: env out of range");
    HIPCHK(hipSetDevice(e->cfg.device));
    int nc = 0;
    float rec[MAXC * 12], force[MAXC];
    HIPCHK(hipMemcpyAsync(&nc, e->D.ccount + env_index, 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(rec, e->D.clist + (size_t)env_index * MAXC * 3, sizeof rec, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(force, e->D.cforce + (size_t)env_index * MAXC, sizeof force, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    nc = std::max(0, std::min(nc, std::min((int)MAXC, (int)max_contacts)));
    for (int c = 0; c < nc; c++) {      // device record {x y z nx | ny nz dist meta | mu rest roll spin} -> {bodyA, bodyB, linkA, x, n, dist, force, mu}
        const float *r = rec + 12 * c;
        int meta;
        memcpy(&meta, r + 7, 4);
        float *o = out_host + 12 * c;
        o[0] = (float)(signed char)(meta & 255); o[1] = (float)(signed char)((meta >> 8) & 255); o[2] = (float)(signed char)((meta >> 16) & 255);
        o[3] = r[0]; o[4] = r[1]; o[5] = r[2]; o[6] = r[3]; o[7] = r[4]; o[8] = r[5]; o[9] = r[6]; o[10] = force[c]; o[11] = r[8];
    }
    *count = nc;
    return RR_OK;
}

int rr_evaluate_goals(rr_env *e, const float *goal_pos_host, const uint8_t *goal_mask_host, float *score_out_host) {
    if (!e || !goal_pos_host || !score_out_host) return fail(RR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    const size_t N = e->P.N, nobj = e->P.nobj;
    int rc;
    if (!e->score_out && ((rc = dev_alloc(e, &e->score_out, N)) != RR_OK || (rc = dev_alloc(e, &e->score_mask, N * NOBJ)) != RR_OK)) return rc;
    static_assert(NSTATE >= NOBJ * 3, "the state staging buffer doubles as goal-position staging");
    HIPCHK(hipMemcpyAsync(e->state_aos, goal_pos_host, N * nobj * 12, hipMemcpyHostToDevice, e->stream));
    if (goal_mask_host) HIPCHK(hipMemcpyAsync(e->score_mask, goal_mask_host, N * nobj, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_goal_score, dim3((e->P.N + 255) / 256), dim3(256), 0, e->stream, e->P, e->D, e->state_aos, goal_mask_host ? e->score_mask : nullptr, e->score_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(score_out_host, e->score_out, N * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}

static int ensure_plan_buffers(rr_env *e) {
    if (e->plan) return RR_OK;
    int rc;
    const size_t N = e->P.N;
    if ((rc = dev_alloc(e, &e->plan, N * PLAN_LEN * 9)) != RR_OK) return rc;
    if ((rc = dev_alloc(e, &e->plan_step, N)) != RR_OK) return rc;
    if ((rc = dev_alloc(e, &e->ik_in, N * 7)) != RR_OK) return rc;
    if ((rc = dev_alloc(e, &e->ik_out, N * 11)) != RR_OK) return rc;
    if ((rc = dev_alloc(e, &e->ik_err, N)) != RR_OK) return rc;
    return RR_OK;
}

int rr_ik(rr_env *e, const float *targets_host, float *q_out_host, float *err_out_host) {
    if (!e || !targets_host || !q_out_host) return fail(RR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    int rc = ensure_plan_buffers(e);
    if (rc != RR_OK) return rc;
    const int N = e->P.N;
    HIPCHK(hipMemcpyAsync(e->ik_in, targets_host, (size_t)N * 28, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(k_ik, dim3((N + 63) / 64), dim3(64), 0, e->stream, e->IK, e->P, e->D, e->ik_in, e->ik_out, e->ik_err);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(q_out_host, e->ik_out, (size_t)N * 44, hipMemcpyDeviceToHost, e->stream));
    if (err_out_host) HIPCHK(hipMemcpyAsync(err_out_host, e->ik_err, (size_t)N * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}

int rr_plan_macro(rr_env *e, const float *macro_host, const uint8_t *env_mask_host) {
    if (!e || !macro_host) return fail(RR_EINVAL, "null argument");
    HIPCHK(hipSetDevice(e->cfg.device));
    int rc = ensure_plan_buffers(e);
    if (rc != RR_OK) return rc;
    const int N = e->P.N;
    // the macro targets travel through the (otherwise idle) ik_out staging buffer: [N][4]
    HIPCHK(hipMemcpyAsync(e->ik_out, macro_host, (size_t)N * 16, hipMemcpyHostToDevice, e->stream));
    const unsigned char *m = nullptr;
    if (env_mask_host) { HIPCHK(hipMemcpyAsync(e->mask_dev, env_mask_host, N, hipMemcpyHostToDevice, e->stream)); m = e->mask_dev; }
    hipLaunchKernelGGL(k_plan_macro, dim3((N + 63) / 64), dim3(64), 0, e->stream, e->IK, e->P, e->D, e->ik_out, m, e->plan, e->plan_step);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}

int rr_get_plan(rr_env *e, int32_t env_index, float *plan_host) {
    if (!e || !plan_host) return fail(RR_EINVAL, "null argument");
    if (!e->plan) return fail(RR_EINVAL, "rr_get_plan: no plan was generated");
    if (env_index < 0 || env_index >= e->P.N) return fail(RR_EINVAL, "rr_get_plan: env out of range");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipMemcpyAsync(plan_host, e->plan + (size_t)env_index * PLAN_LEN * 9, PLAN_LEN * 36, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    return RR_OK;
}

int rr_step_plan_masked(rr_env *e, const uint8_t *idle_mask_host, int32_t render_mode, const uint8_t *render_flags_host) {
    if (!e) return fail(RR_EINVAL, "null env");
    if (!e->plan) return fail(RR_EINVAL, "rr_step_plan: call rr_plan_macro first");
    // (checked here as well: k_plan_fetch advances the plan positions, which must not happen for a step rr_step then rejects)
    if (render_mode < 0 || render_mode > 2 || (render_mode == 2 && !render_flags_host)) return fail(RR_EINVAL, "rr_step_plan: bad render_mode");
    HIPCHK(hipSetDevice(e->cfg.device));
    const unsigned char *idle = nullptr;
    if (idle_mask_host) { HIPCHK(hipMemcpyAsync(e->mask_dev, idle_mask_host, e->P.N, hipMemcpyHostToDevice, e->stream)); idle = e->mask_dev; }
    hipLaunchKernelGGL(k_plan_fetch, dim3((e->P.N + 255) / 256), dim3(256), 0, e->stream, e->P, e->D, e->plan, e->plan_step, idle);
    return rr_step(e, e->D.cmd, 1, render_mode, render_flags_host);
}

int rr_step_plan(rr_env *e, int32_t render_mode, const uint8_t *render_flags_host) {
    return rr_step_plan_masked(e, nullptr, render_mode, render_flags_host);
}

// Replaces the fixed eye camera by an arbitrary one (row-major 4x4 view and projection, OpenGL conventions) and rebuilds
// the static layer. Used for the debug camera of render('rgb_array') (EnvCamera, env.py:470-513).
int rr_set_camera(rr_env *e, const float *view16, const float *proj16) {
    if (!e || (!view16) != (!proj16)) return fail(RR_EINVAL, "rr_set_camera: null argument (both matrices, or neither for the default eye)");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipStreamSynchronize(e->stream));
    if (!view16) look_at_persp(e->RM.VP, e->table_pos, e->RM.W, e->RM.H);       // back to the reference's eye camera (env.py:136-141, 253-255)
    else
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            float a = 0;
            for (int k = 0; k < 4; k++) a += proj16[4 * i + k] * view16[4 * k + j];
            e->RM.VP[4 * i + j] = a;
        }
    frustum_plane_norms(e->RM);
    HIPCHK(hipMemcpy(e->RM_dev, &e->RM, sizeof e->RM, hipMemcpyHostToDevice));
    int rc = build_static_layer(e);
    if (rc != RR_OK) return rc;
    return RR_OK;
}

int rr_set_timing(rr_env *e, int32_t enable) {
    if (!e) return fail(RR_EINVAL, "null env");
    e->timing = enable != 0;
    memset(e->t_ms, 0, sizeof e->t_ms); memset(e->t_n, 0, sizeof e->t_n);
    return RR_OK;
}

int rr_get_timing(rr_env *e, float *ms_out, int32_t *launches_out) {
    if (!e || !ms_out || !launches_out) return fail(RR_EINVAL, "null argument");
    for (int i = 0; i < RR_NUM_KERNELS; i++) { ms_out[i] = e->t_ms[i]; launches_out[i] = e->t_n[i]; e->t_ms[i] = 0; e->t_n[i] = 0; }
    return RR_OK;
}

}  // extern "C"

