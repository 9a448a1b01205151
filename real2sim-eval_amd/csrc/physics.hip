// PhysTwin spring-mass soft-body stepper for MI355X (gfx950, wave64).
//
// Built from scratch against the behaviour of the reference operator
//   sim/physics/spring_mass_warp.py  (SpringMassSystemWarp: 14 NVIDIA-Warp kernels, 9 launches per substep,
//   667 substeps per env step replayed as a ~6000-node CUDA graph).
//
// MI355X design:
//   * ONE fused kernel per substep for a whole batch of environments.  Spring forces are GATHERED per
//     particle from a sliced-ELL adjacency (64 particles = one wavefront per slice, neighbour slots stored
//     slot-major so every load is a coalesced 256-byte run) instead of scattered with float atomics
//     (eval_springs, :61-104) — deterministic and atomic-free.  Velocity update (:107-129), self collision
//     (:132-268), mesh collision (:295-421) and ground integration (:424-474) run in the same thread.
//   * Self collision needs the post-force velocity of the contact partner (object_collision reads
//     v_before_collision[j]); instead of a second launch per substep the few particles that have contact
//     candidates recompute their partners' velocity update (same code path, bit-identical result).
//   * State is ping-ponged between two [env][particle]{x,v} buffers of 32-byte records (two 16-byte
//     gathers per neighbour), topology is shared by all environments of the batch and stays cache-resident.
//   * The num_substeps launches are captured once in a hipGraph.
//   * Resting pairs: per-environment N x N bitset instead of the reference's N x N byte matrix (:715).
//   * Hash grid (wp.HashGrid 128^3, cell = 5 * collision_dist): cell keys sorted with rocPRIM radix sort,
//     cells located by binary search; same cell arithmetic and traversal order as warp-lang 1.7 (unpinned,
//     see oracle/physics_oracle_impl.inc).
//   * Mesh queries: exact closest point + exact solid-angle winding number over the (small) gripper /
//     obstacle meshes, culled by per-substep mesh AABBs.  No BVH yet: the 25k-face pusher mesh is a
//     "next" row.

#include "r2s_common.h"
#include <rocprim/rocprim.hpp>
#include "../../include/r2s_physics.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace {

constexpr int SLICE = 64;
constexpr int BLOCK = 256;
constexpr int GRID_DIM = 128;          // wp.HashGrid(128,128,128), spring_mass_warp.py:541
constexpr int GRID_CELL_BITS = 21;     // 128^3 cells
constexpr float MESH_MAX_DIST = 0.02f; // :323
constexpr float WIND_THRESHOLD = 0.6f; // :323

struct PhysDev {
    int N, E, n_sub;
    // topology (shared by all envs)
    const int* slice_off;      // [n_slices]
    const int* slice_deg;      // [n_slices]
    const int* adj_j;          // sliced ELL, slot-major inside a slice
    const float* adj_inv_rest;
    const float* adj_k;
    const float* masses;       // [N]
    const int* masks;          // [N]
    // scalars
    float dt, dashpot, drag_factor, rf, cd;
    float ce, cf, cee, cef, cse, csf; // clamped restitution / friction
    int self_collision, use_pusher;
    // self-collision candidates
    const int* coll_num;       // [E,N]
    const int* coll_idx;       // [E,N,cap]
    int coll_cap;
    // meshes
    int n_mesh, n_dyn_mesh, nF, nV, n_dyn_pts;
    const int* faces;          // [nF,3] global vertex ids
    const int* mesh_map;       // [nF]
    const int* face_map;       // [nF]
    const int* mesh_face_off;  // [n_mesh+1]
    const float* mesh_pts;     // [E,nV,3] (static part is live; dynamic part = positions at t=0)
    const float* interp_pts;   // [E,n_sub,n_dyn_pts,3]
    const float* interp_center;// [E,n_sub,3]
    const float* dyn_vel;      // [E,2,3]
    const float* dyn_omega;    // [E,3]
    const float* aabb_dyn;     // [E,n_sub,n_dyn_mesh,6]
    const float* aabb_static;  // [E,n_mesh-n_dyn_mesh,6]
    float* coll_forces;        // [E,nF,3]
};

// Everything from here to the spring gather is compiled WITHOUT fused multiply-add contraction: the collision
// and mesh-query arithmetic then rounds exactly like the formulas read (and like the CPU oracle), so discrete
// decisions — which of two equidistant faces is "closest", which side of a margin a particle is on — do not
// depend on the compiler's FMA choices.  None of this code is hot.
#pragma clang fp contract(off)

struct f3 {
    float x, y, z;
};
__device__ __forceinline__ f3 mk(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float len(f3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ f3 normalize0(f3 a) // wp.normalize
{
    float l = len(a);
    return l > 0.f ? a / l : mk(0.f, 0.f, 0.f);
}
__device__ __forceinline__ f3 ld3(const float* p, size_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }
__device__ __forceinline__ f3 xyz(float4 a) { return {a.x, a.y, a.z}; }

// update_vel_from_force, :107-129
__device__ __forceinline__ f3 vel_update(const PhysDev& p, f3 v0, f3 f0, float m0)
{
    const f3 grav = mk(0.f, 0.f, -9.8f) * m0 * p.rf;
    const f3 a = (f0 + grav) / m0;
    const f3 v1 = v0 + a * p.dt;
    return v1 * p.drag_factor;
}

// ---- mesh queries ---------------------------------------------------------------------------------
// Closest point on triangle (a,b,c) to q as barycentrics (u of a, v of b) — Ericson, RTCD 5.1.5.
__device__ __forceinline__ void closest_bary(f3 a, f3 b, f3 c, f3 q, float& u, float& v)
{
    const f3 ab = b - a, ac = c - a, ap = q - a;
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) { u = 1.f; v = 0.f; return; }
    const f3 bp = q - b;
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) { u = 0.f; v = 1.f; return; }
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { const float t = d1 / (d1 - d3); u = 1.f - t; v = t; return; }
    const f3 cp = q - c;
    const float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) { u = 0.f; v = 0.f; return; }
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { const float w = d2 / (d2 - d6); u = 1.f - w; v = 0.f; return; }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        u = 0.f; v = 1.f - w; return;
    }
    const float denom = 1.f / (va + vb + vc);
    const float vv = vb * denom, ww = vc * denom;
    u = 1.f - vv - ww; v = vv;
}

struct MeshQ {
    bool result;
    float sign;
    int face;
    float u, v;
};

__device__ __forceinline__ f3 mesh_vertex(const PhysDev& p, int e, int step, int vid)
{
    if (vid < p.n_dyn_pts) return ld3(p.interp_pts, ((size_t)e * p.n_sub + step) * p.n_dyn_pts + vid);
    return ld3(p.mesh_pts, (size_t)e * p.nV + vid);
}

__device__ __forceinline__ f3 mesh_eval(const PhysDev& p, int e, int step, int face, float u, float v)
{
    const f3 a = mesh_vertex(p, e, step, p.faces[3 * face]), b = mesh_vertex(p, e, step, p.faces[3 * face + 1]),
             c = mesh_vertex(p, e, step, p.faces[3 * face + 2]);
    return a * u + b * v + c * (1.f - u - v);
}

__device__ __forceinline__ float box_dist2(f3 q, const float* bb)
{
    const float dx = fmaxf(fmaxf(bb[0] - q.x, q.x - bb[3]), 0.f);
    const float dy = fmaxf(fmaxf(bb[1] - q.y, q.y - bb[4]), 0.f);
    const float dz = fmaxf(fmaxf(bb[2] - q.z, q.z - bb[5]), 0.f);
    return dx * dx + dy * dy + dz * dz;
}

// wp.mesh_query_point_sign_winding_number(mesh, q, max_dist=0.02, accuracy=3.0, threshold=0.6) restated:
// first strict-minimum closest face with squared distance < max_dist^2; sign from the exact winding number.
// Meshes whose AABB is farther than max_dist cannot contain such a face and are skipped in the search.
__device__ MeshQ mesh_query(const PhysDev& p, int e, int step, f3 q)
{
    MeshQ r = {false, 0.f, 0, 0.f, 0.f};
    float best = MESH_MAX_DIST * MESH_MAX_DIST;
    const float cull = best * 1.0001f + 1e-12f;
    for (int m = 0; m < p.n_mesh; ++m) {
        const float* bb = m < p.n_dyn_mesh ? p.aabb_dyn + (((size_t)e * p.n_sub + step) * p.n_dyn_mesh + m) * 6
                                           : p.aabb_static + ((size_t)e * (p.n_mesh - p.n_dyn_mesh) + (m - p.n_dyn_mesh)) * 6;
        if (box_dist2(q, bb) > cull) continue;
        for (int f = p.mesh_face_off[m]; f < p.mesh_face_off[m + 1]; ++f) {
            const f3 a = mesh_vertex(p, e, step, p.faces[3 * f]), b = mesh_vertex(p, e, step, p.faces[3 * f + 1]),
                     c = mesh_vertex(p, e, step, p.faces[3 * f + 2]);
            float u, v;
            closest_bary(a, b, c, q, u, v);
            const f3 cp = a * u + b * v + c * (1.f - u - v);
            const f3 d = cp - q;
            const float d2 = dot(d, d);
            if (d2 < best) { best = d2; r.result = true; r.face = f; r.u = u; r.v = v; }
        }
    }
    if (!r.result) return r;
    float wsum = 0.f;
    for (int f = 0; f < p.nF; ++f) {
        const f3 a = mesh_vertex(p, e, step, p.faces[3 * f]) - q, b = mesh_vertex(p, e, step, p.faces[3 * f + 1]) - q,
                 c = mesh_vertex(p, e, step, p.faces[3 * f + 2]) - q;
        const float la = len(a), lb = len(b), lc = len(c);
        const float det = dot(a, cross(b, c));
        const float den = la * lb * lc + dot(a, b) * lc + dot(b, c) * la + dot(c, a) * lb;
        wsum += 2.f * atan2f(det, den);
    }
    const float wn = wsum / (float)(4.0 * 3.14159265358979323846);
    r.sign = wn > WIND_THRESHOLD ? -1.f : 1.f;
    return r;
}

// ---- spring forces: gather form of eval_springs (:61-104) ----------------------------------------
// Force on particle i from neighbour j:  [k (L/rest - 1) + c ((vj - vi) . d)] d,  d = (xj - xi) / max(L, 1e-6).
// This is exactly the reference's +F on springs[s][0] and -F on springs[s][1] (the sign flips cancel), summed
// in adjacency order instead of atomic order.  The hot loop: FMA contraction allowed, 1-ulp rsq instead of
// sqrt + three divides (the reference's own float atomics reorder sums far more than this perturbs them).
#pragma clang fp contract(fast)
__device__ __forceinline__ f3 spring_force(const PhysDev& p, const float4* __restrict__ xv, size_t env_base, int i, f3 xi, f3 vi)
{
    const int sl = i >> 6, ln = i & 63;
    const int base = p.slice_off[sl] + ln;
    const int deg = p.slice_deg[sl];
    float fx = 0.f, fy = 0.f, fz = 0.f;
#pragma unroll 4
    for (int n = 0; n < deg; ++n) {
        const int idx = base + n * SLICE;
        const int j = p.adj_j[idx];
        const float inv_rest = p.adj_inv_rest[idx];
        const float k = p.adj_k[idx];
        const float4 xj = xv[(env_base + j) * 2];
        const float4 vj = xv[(env_base + j) * 2 + 1];
        const float dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        const float rinv = __builtin_amdgcn_rsqf(fmaxf(d2, 1e-12f)); // 1 / max(L, 1e-6)
        const float L = d2 * rinv;
        const float ux = dx * rinv, uy = dy * rinv, uz = dz * rinv;
        const float v_rel = (vj.x - vi.x) * ux + (vj.y - vi.y) * uy + (vj.z - vi.z) * uz;
        const float mag = k * (L * inv_rest - 1.0f) + p.dashpot * v_rel;
        fx += ux * mag; fy += uy * mag; fz += uz * mag;
    }
    return {fx, fy, fz};
}
#pragma clang fp contract(off)

// ---- the fused substep ------------------------------------------------------------------------------
// grid = (ceil(N/256), E); one thread per (particle, environment).
template <bool SELF, bool MESH>
__global__ void __launch_bounds__(BLOCK) k_substep(const PhysDev p, const float4* __restrict__ xv_in, float4* __restrict__ xv_out,
                                                   int step, int write_forces)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= p.N) return;
    const size_t eb = (size_t)e * p.N;
    const f3 x0 = xyz(xv_in[(eb + i) * 2]);
    const f3 v0 = xyz(xv_in[(eb + i) * 2 + 1]);
    const float m1 = p.masses[i];

    // eval_springs + update_vel_from_force
    f3 v = vel_update(p, v0, spring_force(p, xv_in, eb, i, x0, v0), m1);

    // object_collision + loop, :132-193, :230-268
    if (SELF) {
        const int cnt = p.coll_num[eb + i];
        if (cnt > 0) {
            const int mask1 = p.masks[i];
            float valid = 0.f;
            f3 Jsum = mk(0.f, 0.f, 0.f);
            for (int k = 0; k < cnt; ++k) {
                const int j = p.coll_idx[(eb + i) * (size_t)p.coll_cap + k];
                const f3 x2 = xyz(xv_in[(eb + j) * 2]);
                const f3 dis = x2 - x0;
                const float dis_len = len(dis);
                if (mask1 == p.masks[j] || !(dis_len < p.cd)) continue;
                // partner's v_before_collision: the same spring gather + velocity update its own thread performs
                const f3 vj0 = xyz(xv_in[(eb + j) * 2 + 1]);
                const float m2 = p.masses[j];
                const f3 v2 = vel_update(p, vj0, spring_force(p, xv_in, eb, j, x2, vj0), m2);
                const f3 rv = v2 - v;
                if (dot(dis, rv) < -1e-4f) {
                    valid += 1.f;
                    const f3 nrm = dis / fmaxf(dis_len, 1e-6f);
                    const f3 v_rel_n = nrm * dot(rv, nrm);
                    const float inv = 1.f / m1 + 1.f / m2;
                    const f3 impulse_n = (v_rel_n * (-(1.f + p.cse))) / inv;
                    const float vnl = len(v_rel_n);
                    const f3 v_rel_t = rv - v_rel_n;
                    const float vtl = fmaxf(len(v_rel_t), 1e-6f);
                    const float a = fmaxf(0.f, 1.f - p.csf * (1.f + p.cse) * vnl / vtl);
                    const f3 impulse_t = (v_rel_t * (a - 1.f)) / inv;
                    Jsum = Jsum + (impulse_n + impulse_t);
                }
            }
            if (valid > 0.f) v = v - (Jsum / valid) / m1;
        }
    }

    f3 x = x0;
    // mesh_collision, :295-421 — advances x by v*dt for EVERY particle (:321, :420)
    if (MESH) {
        f3 vin = v;
        f3 next_x = x0 + vin * p.dt;
        f3 next_v = vin;
        MeshQ q = mesh_query(p, e, step, next_x);
        if (q.result) {
            int is_gripper;
            const int mm = p.mesh_map[q.face];
            if (!p.use_pusher) is_gripper = mm == 0 ? 1 : (mm == 1 ? 2 : 0);
            else is_gripper = mm >= 0 ? 1 : 0;
            f3 pt = mesh_eval(p, e, step, q.face, q.u, q.v);
            f3 delta = next_x - pt;
            float dist = len(delta) * q.sign;
            const float margin = (is_gripper >= 1 && !p.use_pusher) ? 0.005f : 0.001f;
            float err = dist - margin;
            if (err < 0.f) {
                f3 normal = normalize0(delta) * q.sign;
                f3 rdv = mk(0.f, 0.f, 0.f);
                float ce, cf;
                if (is_gripper >= 1) {
                    const f3 ctr = ld3(p.interp_center, (size_t)e * p.n_sub + step);
                    const f3 om = ld3(p.dyn_omega, e);
                    const f3 dv = ld3(p.dyn_vel, (size_t)e * 2 + (is_gripper == 1 ? 0 : 1));
                    rdv = dv + cross(om, x0 - ctr);
                    vin = vin - rdv;
                    ce = p.cee; cf = p.cef;
                } else {
                    ce = p.ce; cf = p.cf;
                }
                const f3 v_normal = normal * dot(vin, normal);
                const f3 v_tao = vin - v_normal;
                const float vnl = len(v_normal);
                const float vtl = fmaxf(len(v_tao), 1e-6f);
                const f3 v_normal_new = v_normal * (-ce);
                const float a = fmaxf(0.f, 1.f - cf * (1.f + ce) * vnl / vtl);
                next_v = v_normal_new + v_tao * a;
                if (is_gripper >= 1) {
                    next_v = next_v + rdv;
                    next_x = x0 + next_v * p.dt;
                    q = mesh_query(p, e, step, next_x); // the reference rebinds `query` (:397)
                    if (q.result) {
                        pt = mesh_eval(p, e, step, q.face, q.u, q.v);
                        delta = next_x - pt;
                        dist = len(delta) * q.sign;
                        err = dist - margin;
                        if (err < 0.f) {
                            normal = normalize0(delta) * q.sign;
                            next_x = next_x - normal * err;
                        }
                    }
                } else {
                    next_x = next_x - normal * err;
                }
                if (write_forces) {
                    const f3 fo = (v_normal_new - v_normal) / p.dt;
                    float* cf3 = p.coll_forces + ((size_t)e * p.nF + p.face_map[q.face]) * 3;
                    atomicAdd(cf3, fo.x);
                    atomicAdd(cf3 + 1, fo.y);
                    atomicAdd(cf3 + 2, fo.z);
                }
            }
        }
        x = next_x;
        v = next_v;
    }

    // integrate_ground_collision, :424-474
    {
        const f3 normal = mk(0.f, 0.f, 1.f) * p.rf;
        const float x_z = x.z, v_z = v.z;
        const float next_x_z = (x_z + v_z * p.dt) * p.rf;
        f3 v1;
        float toi;
        if (next_x_z < 0.f && v_z * p.rf < -1e-4f) {
            const f3 v_normal = normal * dot(v, normal);
            const f3 v_tao = v - v_normal;
            const float vnl = len(v_normal);
            const float vtl = fmaxf(len(v_tao), 1e-6f);
            const f3 v_normal_new = v_normal * (-p.ce);
            const float a = fmaxf(0.f, 1.f - p.cf * (1.f + p.ce) * vnl / vtl);
            v1 = v_normal_new + v_tao * a;
            toi = -(x_z - 0.f) / v_z;
        } else {
            v1 = v;
            toi = 0.f;
        }
        const f3 xn = x + v * toi + v1 * (p.dt - toi);
        xv_out[(eb + i) * 2] = make_float4(xn.x, xn.y, xn.z, 0.f);
        xv_out[(eb + i) * 2 + 1] = make_float4(v1.x, v1.y, v1.z, 0.f);
    }
}

// ---- state pack / unpack ----------------------------------------------------------------------------
__global__ void k_pack(int total, const float* __restrict__ x, const float* __restrict__ v, float4* __restrict__ xv)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    if (x) xv[2 * (size_t)t] = make_float4(x[3 * (size_t)t], x[3 * (size_t)t + 1], x[3 * (size_t)t + 2], 0.f);
    if (v) xv[2 * (size_t)t + 1] = make_float4(v[3 * (size_t)t], v[3 * (size_t)t + 1], v[3 * (size_t)t + 2], 0.f);
}
__global__ void k_unpack(int total, const float4* __restrict__ xv, float* __restrict__ x, float* __restrict__ v)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    if (x) { const float4 a = xv[2 * (size_t)t]; x[3 * (size_t)t] = a.x; x[3 * (size_t)t + 1] = a.y; x[3 * (size_t)t + 2] = a.z; }
    if (v) { const float4 a = xv[2 * (size_t)t + 1]; v[3 * (size_t)t] = a.x; v[3 * (size_t)t + 1] = a.y; v[3 * (size_t)t + 2] = a.z; }
}

// ---- mesh AABBs per (env, substep, dynamic mesh) and per (env, static mesh) ----------------------------
__global__ void k_mesh_aabb_dyn(int E, int n_sub, int n_dyn_mesh, int n_dyn_pts, const int* __restrict__ mesh_vert_off,
                                const float* __restrict__ interp, float* __restrict__ aabb)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * n_sub * n_dyn_mesh) return;
    const int m = t % n_dyn_mesh;
    const size_t es = t / n_dyn_mesh;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int vtx = mesh_vert_off[m]; vtx < mesh_vert_off[m + 1]; ++vtx) {
        const float* q = interp + (es * n_dyn_pts + vtx) * 3;
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], q[k]); hi[k] = fmaxf(hi[k], q[k]); }
    }
    float* o = aabb + (size_t)t * 6;
    for (int k = 0; k < 3; ++k) { o[k] = lo[k]; o[3 + k] = hi[k]; }
}
__global__ void k_mesh_aabb_static(int E, int n_static, int n_dyn_mesh, int nV, const int* __restrict__ mesh_vert_off,
                                   const float* __restrict__ pts, float* __restrict__ aabb)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * n_static) return;
    const int m = t % n_static, e = t / n_static;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int vtx = mesh_vert_off[n_dyn_mesh + m]; vtx < mesh_vert_off[n_dyn_mesh + m + 1]; ++vtx) {
        const float* q = pts + ((size_t)e * nV + vtx) * 3;
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], q[k]); hi[k] = fmaxf(hi[k], q[k]); }
    }
    float* o = aabb + (size_t)t * 6;
    for (int k = 0; k < 3; ++k) { o[k] = lo[k]; o[3 + k] = hi[k]; }
}

// ---- warp-style hash grid -----------------------------------------------------------------------------
__device__ __forceinline__ int grid_cell(int x, int y, int z)
{
    const int origin = 1 << 20;
    x = max(0, x + origin); y = max(0, y + origin); z = max(0, z + origin);
    return (z % GRID_DIM) * (GRID_DIM * GRID_DIM) + (y % GRID_DIM) * GRID_DIM + (x % GRID_DIM);
}

__global__ void k_grid_keys(int N, int E, const float4* __restrict__ xv, float cell_inv, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const float4 q = xv[((size_t)e * N + i) * 2];
    const int c = grid_cell((int)(q.x * cell_inv), (int)(q.y * cell_inv), (int)(q.z * cell_inv));
    keys[(size_t)e * N + i] = ((uint32_t)e << GRID_CELL_BITS) | (uint32_t)c;
    vals[(size_t)e * N + i] = (uint32_t)i;
}

__device__ __forceinline__ void cell_range(const uint32_t* __restrict__ keys, int lo0, int hi0, uint32_t key, int& b, int& en)
{
    int lo = lo0, hi = hi0;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
    b = lo;
    hi = hi0;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] <= key) lo = mid + 1; else hi = mid; }
    en = lo;
}

struct QBox { int xs, ys, zs, xe, ye, ze; };
__device__ __forceinline__ QBox query_box(float4 q, float r, float cell_inv)
{
    QBox b;
    b.xs = (int)((q.x - r) * cell_inv); b.ys = (int)((q.y - r) * cell_inv); b.zs = (int)((q.z - r) * cell_inv);
    b.xe = min((int)((q.x + r) * cell_inv), b.xs + GRID_DIM - 1);
    b.ye = min((int)((q.y + r) * cell_inv), b.ys + GRID_DIM - 1);
    b.ze = min((int)((q.z + r) * cell_inv), b.zs + GRID_DIM - 1);
    return b;
}

// build_resting_collision_pairs, :272-291 (bitset instead of N x N bytes)
__global__ void k_build_resting(int N, int E, int words, const float4* __restrict__ xv, float radius, float cell_inv,
                                const uint32_t* __restrict__ keys, const uint32_t* __restrict__ ids, uint32_t* __restrict__ bits)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const float4 q = xv[((size_t)e * N + i) * 2];
    const QBox b = query_box(q, radius, cell_inv);
    uint32_t* my = bits + ((size_t)e * N) * words;
    for (int z = b.zs; z <= b.ze; ++z)
        for (int y = b.ys; y <= b.ye; ++y)
            for (int x = b.xs; x <= b.xe; ++x) {
                int s, t;
                cell_range(keys, e * N, (e + 1) * N, ((uint32_t)e << GRID_CELL_BITS) | (uint32_t)grid_cell(x, y, z), s, t);
                for (int k = s; k < t; ++k) {
                    const int j = (int)ids[k];
                    if (j < i) {
                        atomicOr(&my[(size_t)i * words + (j >> 5)], 1u << (j & 31));
                        atomicOr(&my[(size_t)j * words + (i >> 5)], 1u << (i & 31));
                    }
                }
            }
}

// update_potential_collision, :196-227 (same candidate order: cells x-fastest, ids ascending inside a cell)
__global__ void k_candidates(int N, int E, int words, int cap, const float4* __restrict__ xv, const int* __restrict__ masks, float cd,
                             float radius, float cell_inv, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ ids,
                             const uint32_t* __restrict__ bits, int* __restrict__ coll_idx, int* __restrict__ coll_num, int* __restrict__ max_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const size_t eb = (size_t)e * N;
    const float4 q = xv[(eb + i) * 2];
    const f3 x1 = xyz(q);
    const int mask1 = masks[i];
    const QBox b = query_box(q, radius, cell_inv);
    const uint32_t* row = bits + (eb + i) * words;
    int cnt = 0;
    for (int z = b.zs; z <= b.ze; ++z)
        for (int y = b.ys; y <= b.ye; ++y)
            for (int x = b.xs; x <= b.xe; ++x) {
                int s, t;
                cell_range(keys, e * N, (e + 1) * N, ((uint32_t)e << GRID_CELL_BITS) | (uint32_t)grid_cell(x, y, z), s, t);
                for (int k = s; k < t; ++k) {
                    const int j = (int)ids[k];
                    if (j == i) continue;
                    const f3 dis = xyz(xv[(eb + j) * 2]) - x1;
                    if (!(len(dis) < cd)) continue;          // cheap test first; same set as the reference order
                    if (row[j >> 5] & (1u << (j & 31))) continue; // resting pair (stored symmetrically)
                    if (mask1 == masks[j]) continue;
                    if (cnt < cap) coll_idx[(eb + i) * (size_t)cap + cnt] = j;
                    cnt++;
                }
            }
    coll_num[eb + i] = min(cnt, cap);
    if (cnt > 0) atomicMax(max_count, cnt);
}

} // namespace

// =========================================================================================================
struct R2SPhys {
    R2SPhysParams prm{};
    int E = 0, N = 0, S = 0, n_slices = 0, ell_len = 0;
    int coll_cap = 500;
    int words = 0;
    int n_mesh = 0, n_dyn_mesh = 0, nF = 0, nV = 0, n_dyn_pts = 0;
    // host copies needed to rebuild stiffness
    std::vector<int> h_springs;
    std::vector<float> h_rest;
    std::vector<int> h_adj_spring; // ELL slot -> spring id (or -1 for padding)
    std::vector<int> h_adj_nbr;    // ELL slot -> neighbour particle
    std::vector<int> h_adj_self;   // ELL slot -> owning particle (padding target)
    std::vector<int> h_mesh_map, h_face_map;
    // device
    float4* xv[2] = {nullptr, nullptr};
    int cur = 0;
    int *d_slice_off = nullptr, *d_slice_deg = nullptr, *d_adj_j = nullptr;
    float *d_adj_inv_rest = nullptr, *d_adj_k = nullptr, *d_masses = nullptr;
    int* d_masks = nullptr;
    int *d_coll_num = nullptr, *d_coll_idx = nullptr, *d_max_count = nullptr;
    uint32_t *d_bits = nullptr, *d_keys[2] = {nullptr, nullptr}, *d_ids[2] = {nullptr, nullptr};
    char* d_sort_tmp = nullptr;
    size_t sort_bytes = 0;
    int *d_faces = nullptr, *d_mesh_map = nullptr, *d_face_map = nullptr, *d_mesh_face_off = nullptr, *d_mesh_vert_off = nullptr;
    float *d_mesh_pts = nullptr, *d_interp = nullptr, *d_center = nullptr, *d_dyn_vel = nullptr, *d_dyn_omega = nullptr;
    float *d_aabb_dyn = nullptr, *d_aabb_static = nullptr, *d_coll_forces = nullptr;
    // graph
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    int graph_start_buf = -1;
    // timing
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int last_kernels = 0;
    bool ev_pending = false;

    PhysDev dev() const
    {
        PhysDev p{};
        p.N = N; p.E = E; p.n_sub = prm.num_substeps;
        p.slice_off = d_slice_off; p.slice_deg = d_slice_deg; p.adj_j = d_adj_j; p.adj_inv_rest = d_adj_inv_rest; p.adj_k = d_adj_k;
        p.masses = d_masses; p.masks = d_masks;
        p.dt = prm.dt; p.dashpot = prm.dashpot_damping; p.drag_factor = expf(-prm.dt * prm.drag_damping);
        p.rf = prm.reverse_z ? -1.f : 1.f; p.cd = prm.collision_dist;
        auto cl = [](float x, float lo, float hi) { return std::min(std::max(x, lo), hi); };
        p.ce = cl(prm.collide_elas, 0.f, 1.f); p.cf = cl(prm.collide_fric, 0.f, 2.f);
        p.cee = cl(prm.collide_eef_elas, 0.f, 1.f); p.cef = cl(prm.collide_eef_fric, 0.f, 2.f);
        p.cse = cl(prm.collide_self_elas, 0.f, 1.f); p.csf = cl(prm.collide_self_fric, 0.f, 2.f);
        p.self_collision = prm.self_collision; p.use_pusher = prm.use_pusher;
        p.coll_num = d_coll_num; p.coll_idx = d_coll_idx; p.coll_cap = coll_cap;
        p.n_mesh = n_mesh; p.n_dyn_mesh = n_dyn_mesh; p.nF = nF; p.nV = nV; p.n_dyn_pts = n_dyn_pts;
        p.faces = d_faces; p.mesh_map = d_mesh_map; p.face_map = d_face_map; p.mesh_face_off = d_mesh_face_off;
        p.mesh_pts = d_mesh_pts; p.interp_pts = d_interp; p.interp_center = d_center; p.dyn_vel = d_dyn_vel; p.dyn_omega = d_dyn_omega;
        p.aabb_dyn = d_aabb_dyn; p.aabb_static = d_aabb_static; p.coll_forces = d_coll_forces;
        return p;
    }
};

namespace {

template <typename T>
int dev_alloc(T** p, size_t count)
{
    R2S_HIP_TRY(hipMalloc((void**)p, sizeof(T) * (count ? count : 1)));
    return R2S_OK;
}
template <typename T>
int upload(T* d, const T* h, size_t count, hipStream_t s)
{
    if (count == 0) return R2S_OK;
    R2S_HIP_TRY(hipMemcpyAsync(d, h, sizeof(T) * count, hipMemcpyHostToDevice, s));
    R2S_HIP_TRY(hipStreamSynchronize(s)); // host staging vectors may die right after the call
    return R2S_OK;
}

// Per-spring stiffness with the reference's gate and clamp (:75, :93); 0 => spring inactive.
void stiffness_from_log(const R2SPhys* h, const float* log_Y, std::vector<float>& k, std::vector<char>& active)
{
    k.resize(h->S); active.resize(h->S);
    for (int s = 0; s < h->S; ++s) {
        const float ey = expf(log_Y[s]);
        active[s] = ey > h->prm.spring_Y_min;
        k[s] = std::min(std::max(ey, h->prm.spring_Y_min), h->prm.spring_Y_max);
    }
}

int upload_stiffness(R2SPhys* h, const float* log_Y, hipStream_t s)
{
    std::vector<float> k; std::vector<char> act;
    stiffness_from_log(h, log_Y, k, act);
    // Slots of inactive springs (gate exp(logY) > Ymin fails, :75) and padding slots point at the particle
    // itself with k = 0: then d = 0 and dv = 0, so neither the spring nor the dashpot term contributes.
    std::vector<float> ell_k(h->ell_len, 0.f), ell_ir(h->ell_len, 0.f);
    std::vector<int> ell_j(h->ell_len);
    for (int t = 0; t < h->ell_len; ++t) {
        const int sp = h->h_adj_spring[t];
        if (sp >= 0 && act[sp]) { ell_k[t] = k[sp]; ell_ir[t] = 1.0f / h->h_rest[sp]; ell_j[t] = h->h_adj_nbr[t]; }
        else ell_j[t] = h->h_adj_self[t];
    }
    int rc = upload(h->d_adj_j, ell_j.data(), ell_j.size(), s);
    if (rc) return rc;
    rc = upload(h->d_adj_k, ell_k.data(), ell_k.size(), s);
    if (rc) return rc;
    return upload(h->d_adj_inv_rest, ell_ir.data(), ell_ir.size(), s);
}

int launch_substep(R2SPhys* h, const PhysDev& p, int in_buf, int step, int write_forces, hipStream_t s)
{
    dim3 grid((h->N + BLOCK - 1) / BLOCK, h->E);
    const bool self = h->prm.self_collision != 0, mesh = h->nF > 0;
    const float4* in = h->xv[in_buf];
    float4* out = h->xv[in_buf ^ 1];
    if (self && mesh) hipLaunchKernelGGL((k_substep<true, true>), grid, dim3(BLOCK), 0, s, p, in, out, step, write_forces);
    else if (self) hipLaunchKernelGGL((k_substep<true, false>), grid, dim3(BLOCK), 0, s, p, in, out, step, write_forces);
    else if (mesh) hipLaunchKernelGGL((k_substep<false, true>), grid, dim3(BLOCK), 0, s, p, in, out, step, write_forces);
    else hipLaunchKernelGGL((k_substep<false, false>), grid, dim3(BLOCK), 0, s, p, in, out, step, write_forces);
    return R2S_OK;
}

// Enqueue substeps [first, first+n) starting from buffer `start_buf`; the final state is left in buffer
// start_buf ^ (n & 1).
int enqueue_steps(R2SPhys* h, int first, int n, int start_buf, hipStream_t s)
{
    const PhysDev p = h->dev();
    int buf = start_buf;
    for (int k = 0; k < n; ++k) {
        const int last = (k == n - 1);
        if (last && h->nF > 0) R2S_HIP_TRY(hipMemsetAsync(h->d_coll_forces, 0, sizeof(float) * 3 * (size_t)h->E * h->nF, s));
        int rc = launch_substep(h, p, buf, first + k, last, s);
        if (rc) return rc;
        buf ^= 1;
    }
    return R2S_OK;
}

void drop_graph(R2SPhys* h)
{
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    h->graph_exec = nullptr; h->graph = nullptr; h->graph_start_buf = -1;
}

int capture_graph(R2SPhys* h, int start_buf)
{
    drop_graph(h);
    hipStream_t cs;
    R2S_HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    R2S_HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_steps(h, 0, h->prm.num_substeps, start_buf, cs);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(cs, &g);
    (void)hipStreamDestroy(cs);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    R2S_HIP_TRY(e);
    h->graph = g;
    R2S_HIP_TRY(hipGraphInstantiate(&h->graph_exec, g, nullptr, nullptr, 0));
    h->graph_start_buf = start_buf;
    return R2S_OK;
}

int grid_sort(R2SPhys* h, hipStream_t s, const uint32_t** keys, const uint32_t** ids)
{
    const float cell = h->prm.collision_dist * 5.0f;
    const float cell_inv = 1.0f / cell;
    dim3 grid((h->N + BLOCK - 1) / BLOCK, h->E);
    hipLaunchKernelGGL(k_grid_keys, grid, dim3(BLOCK), 0, s, h->N, h->E, h->xv[h->cur], cell_inv, h->d_keys[0], h->d_ids[0]);
    rocprim::double_buffer<uint32_t> dk(h->d_keys[0], h->d_keys[1]);
    rocprim::double_buffer<uint32_t> dv(h->d_ids[0], h->d_ids[1]);
    unsigned bits = GRID_CELL_BITS;
    while ((1u << (bits - GRID_CELL_BITS)) < (unsigned)h->E) ++bits;
    size_t need = h->sort_bytes;
    R2S_HIP_TRY(rocprim::radix_sort_pairs(h->d_sort_tmp, need, dk, dv, (size_t)h->E * h->N, 0u, bits, s));
    *keys = dk.current();
    *ids = dv.current();
    return R2S_OK;
}

} // namespace

extern "C" {

int r2s_phys_create(const R2SPhysDesc* d, R2SPhys** out, r2s_stream_t stream_)
{
    hipStream_t s = (hipStream_t)stream_;
    if (!d || !out || d->n_env <= 0 || d->num_object_points <= 0 || d->num_springs < 0 || !d->init_vertices || !d->init_masses ||
        (d->num_springs > 0 && (!d->init_springs || !d->init_rest_lengths || !d->init_spring_Y)) || d->params.num_substeps <= 0)
        return R2S_ERR_INVALID;
    if ((uint64_t)d->n_env >= (1u << (32 - GRID_CELL_BITS))) return R2S_ERR_INVALID;
    R2SPhys* h = new (std::nothrow) R2SPhys();
    if (!h) return R2S_ERR_ALLOC;
    h->prm = d->params;
    h->E = d->n_env; h->N = d->num_object_points; h->S = d->num_springs;
    h->coll_cap = d->collision_capacity > 0 ? d->collision_capacity : 500;
    const int N = h->N, E = h->E, S = h->S;
    for (int sp = 0; sp < S; ++sp)
        if (d->init_springs[2 * sp] < 0 || d->init_springs[2 * sp] >= N || d->init_springs[2 * sp + 1] < 0 || d->init_springs[2 * sp + 1] >= N) {
            delete h; return R2S_ERR_INVALID;
        }
    h->h_springs.assign(d->init_springs, d->init_springs + 2 * (size_t)S);
    h->h_rest.assign(d->init_rest_lengths, d->init_rest_lengths + S);
    int rc = R2S_OK;
#define TRY(x) do { rc = (x); if (rc != R2S_OK) { r2s_phys_destroy(h); return rc; } } while (0)

    // ---- sliced-ELL adjacency (gather form) ----
    std::vector<std::vector<std::pair<int, int>>> adj(N); // (neighbour, spring id) in spring order
    for (int sp = 0; sp < S; ++sp) {
        const int a = h->h_springs[2 * sp], b = h->h_springs[2 * sp + 1];
        adj[a].push_back({b, sp});
        adj[b].push_back({a, sp});
    }
    h->n_slices = (N + SLICE - 1) / SLICE;
    std::vector<int> slice_off(h->n_slices), slice_deg(h->n_slices);
    int total = 0;
    for (int sl = 0; sl < h->n_slices; ++sl) {
        int deg = 0;
        for (int i = sl * SLICE; i < std::min(N, (sl + 1) * SLICE); ++i) deg = std::max(deg, (int)adj[i].size());
        slice_off[sl] = total; slice_deg[sl] = deg;
        total += deg * SLICE;
    }
    h->ell_len = total;
    h->h_adj_spring.assign(total, -1);
    h->h_adj_nbr.assign(total, 0);
    h->h_adj_self.assign(total, 0);
    for (int sl = 0; sl < h->n_slices; ++sl)
        for (int ln = 0; ln < SLICE; ++ln) {
            const int i = sl * SLICE + ln;
            for (int n = 0; n < slice_deg[sl]; ++n) {
                const int t = slice_off[sl] + n * SLICE + ln;
                h->h_adj_self[t] = i < N ? i : 0;
                if (i < N && n < (int)adj[i].size()) { h->h_adj_nbr[t] = adj[i][n].first; h->h_adj_spring[t] = adj[i][n].second; }
            }
        }
    TRY(dev_alloc(&h->d_slice_off, h->n_slices)); TRY(dev_alloc(&h->d_slice_deg, h->n_slices));
    TRY(dev_alloc(&h->d_adj_j, total)); TRY(dev_alloc(&h->d_adj_inv_rest, total)); TRY(dev_alloc(&h->d_adj_k, total));
    TRY(dev_alloc(&h->d_masses, N)); TRY(dev_alloc(&h->d_masks, N));
    TRY(upload(h->d_slice_off, slice_off.data(), slice_off.size(), s)); TRY(upload(h->d_slice_deg, slice_deg.data(), slice_deg.size(), s));
    {
        std::vector<float> zero_logy(std::max(S, 1), 0.f);
        TRY(upload_stiffness(h, S > 0 ? d->init_spring_Y : zero_logy.data(), s));
    }
    TRY(upload(h->d_masses, d->init_masses, N, s));
    {
        std::vector<int> masks(N);
        for (int i = 0; i < N; ++i) masks[i] = d->init_collision_mask ? d->init_collision_mask[i] : i;
        TRY(upload(h->d_masks, masks.data(), N, s));
    }
    // ---- state ----
    for (int b = 0; b < 2; ++b) {
        TRY(dev_alloc(&h->xv[b], (size_t)E * N * 2));
        R2S_HIP_TRY(hipMemsetAsync(h->xv[b], 0, sizeof(float4) * (size_t)E * N * 2, s));
    }
    {
        std::vector<float4> pk((size_t)E * N * 2);
        for (size_t t = 0; t < (size_t)E * N; ++t) {
            pk[2 * t] = make_float4(d->init_vertices[3 * t], d->init_vertices[3 * t + 1], d->init_vertices[3 * t + 2], 0.f);
            pk[2 * t + 1] = d->init_velocities ? make_float4(d->init_velocities[3 * t], d->init_velocities[3 * t + 1], d->init_velocities[3 * t + 2], 0.f)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        TRY(upload(h->xv[0], pk.data(), pk.size(), s));
    }
    h->cur = 0;

    // ---- meshes (spring_mass_warp.py:626-711) ----
    h->n_dyn_mesh = d->n_dynamic_meshes; h->n_mesh = d->n_dynamic_meshes + d->n_static_meshes;
    const int n_sub = h->prm.num_substeps;
    if (h->n_mesh > 0) {
        std::vector<int> voff(h->n_mesh + 1, 0), foff(h->n_mesh + 1, 0);
        for (int m = 0; m < h->n_mesh; ++m) { voff[m + 1] = voff[m] + d->mesh_num_vertices[m]; foff[m + 1] = foff[m] + d->mesh_num_faces[m]; }
        h->nV = voff[h->n_mesh]; h->nF = foff[h->n_mesh]; h->n_dyn_pts = voff[h->n_dyn_mesh];
        std::vector<int> faces(3 * (size_t)h->nF);
        h->h_mesh_map.resize(h->nF); h->h_face_map.resize(h->nF);
        for (int m = 0; m < h->n_mesh; ++m)
            for (int f = foff[m]; f < foff[m + 1]; ++f) {
                for (int k = 0; k < 3; ++k) faces[3 * f + k] = d->mesh_triangles[3 * f + k] + voff[m];
                h->h_mesh_map[f] = m < h->n_dyn_mesh ? m : -(m - h->n_dyn_mesh) - 1;
                h->h_face_map[f] = f;
            }
        TRY(dev_alloc(&h->d_faces, faces.size())); TRY(dev_alloc(&h->d_mesh_map, h->nF)); TRY(dev_alloc(&h->d_face_map, h->nF));
        TRY(dev_alloc(&h->d_mesh_face_off, h->n_mesh + 1)); TRY(dev_alloc(&h->d_mesh_vert_off, h->n_mesh + 1));
        TRY(upload(h->d_faces, faces.data(), faces.size(), s)); TRY(upload(h->d_mesh_map, h->h_mesh_map.data(), h->nF, s));
        TRY(upload(h->d_face_map, h->h_face_map.data(), h->nF, s));
        TRY(upload(h->d_mesh_face_off, foff.data(), foff.size(), s)); TRY(upload(h->d_mesh_vert_off, voff.data(), voff.size(), s));
        // vertices replicated per env; interpolated dynamic points = initial points repeated (:699-703)
        std::vector<float> pts((size_t)E * h->nV * 3);
        for (int e = 0; e < E; ++e) std::copy(d->mesh_vertices, d->mesh_vertices + 3 * (size_t)h->nV, pts.begin() + (size_t)e * h->nV * 3);
        TRY(dev_alloc(&h->d_mesh_pts, pts.size())); TRY(upload(h->d_mesh_pts, pts.data(), pts.size(), s));
        std::vector<float> interp((size_t)E * n_sub * h->n_dyn_pts * 3), center((size_t)E * n_sub * 3, 0.f);
        double c[3] = {0, 0, 0};
        for (int v = 0; v < h->n_dyn_pts; ++v) for (int k = 0; k < 3; ++k) c[k] += d->mesh_vertices[3 * v + k];
        for (size_t es = 0; es < (size_t)E * n_sub; ++es) {
            std::copy(d->mesh_vertices, d->mesh_vertices + 3 * (size_t)h->n_dyn_pts, interp.begin() + es * h->n_dyn_pts * 3);
            for (int k = 0; k < 3; ++k) center[es * 3 + k] = h->n_dyn_pts ? (float)(c[k] / h->n_dyn_pts) : 0.f; // mean of dynamic points (:704-708)
        }
        TRY(dev_alloc(&h->d_interp, interp.size())); TRY(upload(h->d_interp, interp.data(), interp.size(), s));
        TRY(dev_alloc(&h->d_center, center.size())); TRY(upload(h->d_center, center.data(), center.size(), s));
        TRY(dev_alloc(&h->d_dyn_vel, (size_t)E * 6)); TRY(dev_alloc(&h->d_dyn_omega, (size_t)E * 3));
        R2S_HIP_TRY(hipMemsetAsync(h->d_dyn_vel, 0, sizeof(float) * E * 6, s));
        R2S_HIP_TRY(hipMemsetAsync(h->d_dyn_omega, 0, sizeof(float) * E * 3, s));
        TRY(dev_alloc(&h->d_aabb_dyn, (size_t)E * n_sub * std::max(1, h->n_dyn_mesh) * 6));
        TRY(dev_alloc(&h->d_aabb_static, (size_t)E * std::max(1, h->n_mesh - h->n_dyn_mesh) * 6));
        TRY(dev_alloc(&h->d_coll_forces, (size_t)E * h->nF * 3));
        R2S_HIP_TRY(hipMemsetAsync(h->d_coll_forces, 0, sizeof(float) * 3 * (size_t)E * h->nF, s));
        if (h->n_dyn_mesh > 0) {
            const int tot = E * n_sub * h->n_dyn_mesh;
            hipLaunchKernelGGL(k_mesh_aabb_dyn, dim3((tot + 255) / 256), dim3(256), 0, s, E, n_sub, h->n_dyn_mesh, h->n_dyn_pts, h->d_mesh_vert_off, h->d_interp, h->d_aabb_dyn);
        }
        if (h->n_mesh > h->n_dyn_mesh) {
            const int ns = h->n_mesh - h->n_dyn_mesh, tot = E * ns;
            hipLaunchKernelGGL(k_mesh_aabb_static, dim3((tot + 255) / 256), dim3(256), 0, s, E, ns, h->n_dyn_mesh, h->nV, h->d_mesh_vert_off, h->d_mesh_pts, h->d_aabb_static);
        }
    }

    // ---- self collision (:528-552, :714-721) ----
    TRY(dev_alloc(&h->d_coll_num, (size_t)E * N));
    R2S_HIP_TRY(hipMemsetAsync(h->d_coll_num, 0, sizeof(int) * (size_t)E * N, s));
    TRY(dev_alloc(&h->d_max_count, 4));
    R2S_HIP_TRY(hipMemsetAsync(h->d_max_count, 0, sizeof(int) * 4, s));
    if (h->prm.self_collision) {
        h->words = (N + 31) / 32;
        TRY(dev_alloc(&h->d_coll_idx, (size_t)E * N * h->coll_cap));
        TRY(dev_alloc(&h->d_bits, (size_t)E * N * h->words));
        for (int b = 0; b < 2; ++b) { TRY(dev_alloc(&h->d_keys[b], (size_t)E * N)); TRY(dev_alloc(&h->d_ids[b], (size_t)E * N)); }
        rocprim::double_buffer<uint32_t> dk((uint32_t*)nullptr, (uint32_t*)nullptr), dv((uint32_t*)nullptr, (uint32_t*)nullptr);
        R2S_HIP_TRY(rocprim::radix_sort_pairs(nullptr, h->sort_bytes, dk, dv, (size_t)E * N, 0u, 32u, s));
        TRY(dev_alloc(&h->d_sort_tmp, h->sort_bytes));
        TRY(r2s_phys_create_resting_case(h, stream_));
    }
    R2S_HIP_TRY(hipStreamSynchronize(s));
    TRY(capture_graph(h, 0));
#undef TRY
    *out = h;
    return R2S_OK;
}

void r2s_phys_destroy(R2SPhys* h)
{
    if (!h) return;
    (void)hipDeviceSynchronize();
    drop_graph(h);
    void* ptrs[] = {h->xv[0], h->xv[1], h->d_slice_off, h->d_slice_deg, h->d_adj_j, h->d_adj_inv_rest, h->d_adj_k, h->d_masses, h->d_masks,
                    h->d_coll_num, h->d_coll_idx, h->d_max_count, h->d_bits, h->d_keys[0], h->d_keys[1], h->d_ids[0], h->d_ids[1], h->d_sort_tmp,
                    h->d_faces, h->d_mesh_map, h->d_face_map, h->d_mesh_face_off, h->d_mesh_vert_off, h->d_mesh_pts, h->d_interp, h->d_center,
                    h->d_dyn_vel, h->d_dyn_omega, h->d_aabb_dyn, h->d_aabb_static, h->d_coll_forces};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    delete h;
}

int r2s_phys_set_state(R2SPhys* h, const float* x, const float* v, r2s_stream_t stream_)
{
    if (!h) return R2S_ERR_INVALID;
    const int total = h->E * h->N;
    hipLaunchKernelGGL(k_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream_, total, x, v, h->xv[h->cur]);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_get_state(R2SPhys* h, float* x, float* v, r2s_stream_t stream_)
{
    if (!h) return R2S_ERR_INVALID;
    const int total = h->E * h->N;
    hipLaunchKernelGGL(k_unpack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream_, total, h->xv[h->cur], x, v);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_create_resting_case(R2SPhys* h, r2s_stream_t stream_)
{
    if (!h || !h->prm.self_collision) return R2S_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream_;
    const uint32_t *keys, *ids;
    int rc = grid_sort(h, s, &keys, &ids);
    if (rc) return rc;
    R2S_HIP_TRY(hipMemsetAsync(h->d_bits, 0, sizeof(uint32_t) * (size_t)h->E * h->N * h->words, s));
    const float r = h->prm.collision_dist * 5.0f;
    dim3 grid((h->N + BLOCK - 1) / BLOCK, h->E);
    hipLaunchKernelGGL(k_build_resting, grid, dim3(BLOCK), 0, s, h->N, h->E, h->words, h->xv[h->cur], r, 1.0f / r, keys, ids, h->d_bits);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_update_collision_graph(R2SPhys* h, r2s_stream_t stream_)
{
    if (!h || !h->prm.self_collision) return R2S_ERR_INVALID; // `assert self.self_collision`, :807
    hipStream_t s = (hipStream_t)stream_;
    const uint32_t *keys, *ids;
    int rc = grid_sort(h, s, &keys, &ids);
    if (rc) return rc;
    R2S_HIP_TRY(hipMemsetAsync(h->d_max_count, 0, sizeof(int), s));
    const float r = h->prm.collision_dist * 5.0f;
    dim3 grid((h->N + BLOCK - 1) / BLOCK, h->E);
    hipLaunchKernelGGL(k_candidates, grid, dim3(BLOCK), 0, s, h->N, h->E, h->words, h->coll_cap, h->xv[h->cur], h->d_masks, h->prm.collision_dist, r,
                       1.0f / r, keys, ids, h->d_bits, h->d_coll_idx, h->d_coll_num, h->d_max_count);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_set_mesh_interactive(R2SPhys* h, const float* interp_points, const float* interp_center, const float* dynamic_velocity,
                                  const float* dynamic_omega, r2s_stream_t stream_)
{
    if (!h || h->n_mesh == 0 || !interp_points || !interp_center || !dynamic_velocity || !dynamic_omega) return R2S_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream_;
    const int E = h->E, n_sub = h->prm.num_substeps;
    const int n_dyn_vel = h->prm.use_pusher ? 1 : 2;
    R2S_HIP_TRY(hipMemcpyAsync(h->d_interp, interp_points, sizeof(float) * 3 * (size_t)E * n_sub * h->n_dyn_pts, hipMemcpyDeviceToDevice, s));
    R2S_HIP_TRY(hipMemcpyAsync(h->d_center, interp_center, sizeof(float) * 3 * (size_t)E * n_sub, hipMemcpyDeviceToDevice, s));
    R2S_HIP_TRY(hipMemcpy2DAsync(h->d_dyn_vel, sizeof(float) * 6, dynamic_velocity, sizeof(float) * 3 * n_dyn_vel, sizeof(float) * 3 * n_dyn_vel, E,
                                 hipMemcpyDeviceToDevice, s));
    R2S_HIP_TRY(hipMemcpyAsync(h->d_dyn_omega, dynamic_omega, sizeof(float) * 3 * (size_t)E, hipMemcpyDeviceToDevice, s));
    if (h->n_dyn_mesh > 0) {
        const int tot = E * n_sub * h->n_dyn_mesh;
        hipLaunchKernelGGL(k_mesh_aabb_dyn, dim3((tot + 255) / 256), dim3(256), 0, s, E, n_sub, h->n_dyn_mesh, h->n_dyn_pts, h->d_mesh_vert_off, h->d_interp, h->d_aabb_dyn);
    }
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_step(R2SPhys* h, int n_substeps, int first_substep, r2s_stream_t stream_)
{
    if (!h) return R2S_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream_;
    const int full = h->prm.num_substeps;
    const bool use_graph = (n_substeps <= 0 || n_substeps == full) && first_substep == 0;
    const int n = use_graph ? full : n_substeps;
    if (first_substep < 0 || first_substep + n > full) return R2S_ERR_INVALID;
    if (h->timing) {
        if (!h->ev0) { R2S_HIP_TRY(hipEventCreate(&h->ev0)); R2S_HIP_TRY(hipEventCreate(&h->ev1)); }
        R2S_HIP_TRY(hipEventRecord(h->ev0, s));
    }
    if (use_graph) {
        if (!h->graph_exec || h->graph_start_buf != h->cur) {
            int rc = capture_graph(h, h->cur);
            if (rc) return rc;
        }
        R2S_HIP_TRY(hipGraphLaunch(h->graph_exec, s));
    } else {
        int rc = enqueue_steps(h, first_substep, n, h->cur, s);
        if (rc) return rc;
    }
    h->cur ^= (n & 1);
    if (h->timing) { R2S_HIP_TRY(hipEventRecord(h->ev1, s)); h->ev_pending = true; h->last_kernels = n; }
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_collision_forces(R2SPhys* h, float** device_ptr, int32_t* n_faces)
{
    if (!h) return R2S_ERR_INVALID;
    if (device_ptr) *device_ptr = h->d_coll_forces;
    if (n_faces) *n_faces = h->nF;
    return R2S_OK;
}

int r2s_phys_mesh_maps(R2SPhys* h, int32_t* mesh_map, int32_t* face_map)
{
    if (!h) return R2S_ERR_INVALID;
    if (mesh_map) std::copy(h->h_mesh_map.begin(), h->h_mesh_map.end(), mesh_map);
    if (face_map) std::copy(h->h_face_map.begin(), h->h_face_map.end(), face_map);
    return R2S_OK;
}

int r2s_phys_collision_lists(R2SPhys* h, int32_t** number, int32_t** indices, int32_t* capacity)
{
    if (!h) return R2S_ERR_INVALID;
    if (number) *number = h->d_coll_num;
    if (indices) *indices = h->d_coll_idx;
    if (capacity) *capacity = h->coll_cap;
    return R2S_OK;
}

int r2s_phys_collision_max_count(R2SPhys* h, int32_t* max_count, r2s_stream_t stream_)
{
    if (!h || !max_count) return R2S_ERR_INVALID;
    R2S_HIP_TRY(hipMemcpyAsync(max_count, h->d_max_count, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    R2S_HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    return R2S_OK;
}

int r2s_phys_set_spring_Y(R2SPhys* h, const float* log_Y, r2s_stream_t stream_)
{
    if (!h || !log_Y) return R2S_ERR_INVALID;
    return upload_stiffness(h, log_Y, (hipStream_t)stream_);
}

int r2s_phys_set_params(R2SPhys* h, const R2SPhysParams* p, r2s_stream_t)
{
    if (!h || !p) return R2S_ERR_INVALID;
    if (p->num_substeps != h->prm.num_substeps || p->self_collision != h->prm.self_collision || p->use_pusher != h->prm.use_pusher)
        return R2S_ERR_INVALID; // structural fields are fixed at construction
    h->prm = *p;
    drop_graph(h); // kernel arguments are baked into the graph; re-captured lazily by the next step
    return R2S_OK;
}

void r2s_phys_set_timing(R2SPhys* h, int enable) { if (h) h->timing = enable != 0; }

int r2s_phys_last_step_ms(R2SPhys* h, float* ms, int32_t* kernels)
{
    if (!h || !ms || !h->ev_pending) return R2S_ERR_INVALID;
    R2S_HIP_TRY(hipEventSynchronize(h->ev1));
    R2S_HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    if (kernels) *kernels = h->last_kernels;
    return R2S_OK;
}

} // extern "C"
