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
#include <hip/hip_ext.h>

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
    float roff[MAXSHAPES];           // circumradius about the sphere centre of the shape grown by the contact margin plane by plane (tools/compile_model.py
                                     // offset_radius; +inf when the blob has none or was compiled for a smaller margin): k_collide's pair cull
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
    // dispatch order of k_collide's workgroups (dispatch_order_class with one item per env): the envs by falling duration of their last collision pass
    unsigned *collide_cost, *collide_bin;   // [N] that duration (100 MHz ticks) / the bin it was counted in
    unsigned *collide_perm;                 // [8 * ceil(N / 8)] env << 8, costly envs first, ~0 behind the last one of a class
    float *cforce;     // [N][MAXC] normal force of every contact of the last solved step (rr_get_contacts, touch sensors, warm start)
    int *hcount_host;  // device address of the pinned host word that receives the current number of heavy envs (or nullptr)
    int *timestep;     // [N]
    unsigned *errflags;// [N]
    const float *body_tab; // [NB][16] per-body constants of k_prep16's body lanes (BT_*: com, inertia, mass, joint damping, axis)
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
    const float4 *tri_rec;  // AoS [NT][8]: one 128-byte shading record per triangle {pos[9], nrm[9], inst, -, uv[6], pad}
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
    // dispatch order of k_raster's workgroups (dispatch_order_class): the (env, tile) items by falling cost of the previous frame
    unsigned *item_cost;        // [N*ntiles] duration of the item's workgroup in the last frame that rasterised it (100 MHz ticks)
    unsigned *item_bin;         // [N*ntiles] dispatch_order_class: the bin an item was counted in (its second pass reads it back, not the cost again)
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

// Dispatch order of the next frame's k_raster: the (env, tile) items by falling cost (counting sort over 1024 linear bins of the
// durations k_raster has just measured), by a few extra workgroups of the k_shade launch that follows it (a stream of its own was
// measured: a fifth stream shares a hardware queue with one of the step's four and serialises it, 0.67 -> 0.83 ms).  One class per
// XCD -- workgroup index mod 8 is the XCD --: the envs = x (mod 8), so every XCD keeps an eighth of the envs with all their tiles
// and its queue holds items of falling cost.  (Which XCD rasterises a tile does not matter to k_shade: with the tiles of an env
// dealt to different XCDs it takes 0.089 instead of 0.088 ms.)  perm[8 * j + x] = the j-th costliest item of class x,
// env << 8 | tile, or ~0 behind the last one.
// Costs change little from frame to frame (5 ms of motion); an order that is off only loads the shader engines less evenly, the
// images do not depend on it.
// The collision pass uses the same sort with one item per env (ntiles = 1, extra workgroups of the k_prep_a / k_prep_ab launch in
// front of it, NT_ = 64): its workgroups last 14 us +- 20 % with a tail up to 70 us, and in env order a quarter of the slots stay empty.
#define ORDER_BINS 1024
template <int NT_>
__device__ __forceinline__ void dispatch_order_class(int x, int N, int ntiles, int per_class, const unsigned *cost, unsigned *bins, unsigned *perm) {
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


// The translation unit is kept in parts (one library, one compilation: the kernels share structs, device math and macros):
#include "rr_prep.inc"    // k_prep: the state part of a step (forward kinematics, object terms, joint-space dynamics)
#include "rr_collide.inc"    // k_collide: the collision pass (contact lists, warm-start matching, env classes)
#include "rr_solve.inc"    // k_solve / k_solve_light / k_solve_light_ow: command part, row build, projected Gauss-Seidel, integration
#include "rr_state.inc"    // reset / state io / observation kernels, render set-up
#include "rr_ik.inc"    // inverse kinematics and macro plans (K8)
#include "rr_render.inc"    // rasteriser: k_raster, k_shade, the list-walking render kernels
#include "rr_gather.inc"    // delta image records for the observation gather (k_pack_delta, k_apply_delta)
#include "rr_host.inc"    // host side: model blob, rr_env, the C ABI
