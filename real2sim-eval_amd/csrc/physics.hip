// PhysTwin spring-mass soft-body stepper for MI355X (gfx950, wave64).
//
// Built from scratch against the behaviour of the reference operator
//   sim/physics/spring_mass_warp.py  (SpringMassSystemWarp: 14 NVIDIA-Warp kernels, 9 launches per substep,
//   667 substeps per env step replayed as a ~6000-node CUDA graph).
//
// MI355X design:
//   * ONE fused kernel per substep for a whole batch of environments.  Spring forces are GATHERED per
//     particle from a sliced-ELL adjacency (64 particles = one wavefront per slice, 4-slot groups stored
//     group-major so every load is a coalesced run) instead of scattered with float atomics
//     (eval_springs, :61-104) — deterministic and atomic-free.  Velocity update (:107-129), mesh collision
//     (:295-421) and ground integration (:424-474) run in the same thread.
//   * Particles are re-ordered along a Morton curve at construction (the caller never sees it: set/get_state
//     permute).  A workgroup owns 256 consecutive particles of one environment and stages their state plus
//     a precomputed HALO (every outside neighbour, up to the window capacity) into an LDS window of three
//     8-byte planes xy | (z, vz) | vxy; the ~31 neighbour gathers per particle are then three ds_read_b64
//     off one address register.  Neighbours that do not fit the window fall back to a global gather, so any
//     topology works.
//   * Workgroups are numbered so that each XCD (private 4 MiB L2) owns a contiguous chunk of particle blocks
//     for ALL environments, environment index fastest: its slice of the adjacency and its particles' state
//     stay resident in that XCD's L2 and most halo records were written by the same XCD.
//   * Self collision needs the post-force velocity of the contact partner (object_collision reads
//     v_before_collision[j]): particles that appear in a candidate list only PUBLISH their post-force velocity
//     in the fused kernel and are finished by the substep's ONE finishing launch — k_contact_finish (part 2; part 1 are
//     the deferred mesh queries) while something is near a mesh, k_self_finish otherwise; only the graph flavours used
//     while candidates exist carry it.
//   * State is ping-ponged between two buffers of 24-byte records [env][particle]{xy | (z, vz) | vxy} — the three
//     8-byte words of the LDS window's planes, no padding, nothing to repack when staging; topology is shared by all
//     environments and stays cache-resident.
//   * The num_substeps launches are captured once per flavour in a hipGraph.
//   * Resting pairs: per-environment N x N bitset instead of the reference's N x N byte matrix (:715).
//   * Hash grid (wp.HashGrid 128^3, cell = 5 * collision_dist) for create_resting_case; the per-step candidate
//     rebuild bins on a fine grid (cell = collision_dist) and restores the reference's traversal order
//     (unpinned against warp-lang 1.7 itself, see oracle/physics_oracle_impl.inc).
//   * Mesh queries (wp.mesh_query_point_sign_winding_number, :322-324).  While nothing is near a mesh the fused kernel
//     answers the rare query of a small scene itself (per-lane brute force, exact solid-angle winding number).  While
//     something is near — and always in a scene with a large mesh (> 256 faces: the ~25k-face pusher) — the fused
//     kernel only LISTS the particles that need a query and k_contact_finish finishes them, one workgroup per
//     particle: two wavefronts with the scene's <= 128 triangles in registers (one per lane), or four wavefronts over
//     a two-level box hierarchy of Morton-sorted 64-face clusters in the mesh's rest frame (sign from pseudonormals
//     for closed manifolds, the exact winding number for anything else).  What a listed particle costs is the
//     instruction stream of a lone wavefront (~3 ns per instruction), not memory latency: see mesh_query_block.

#include "r2s_common.h"
#include <rocprim/rocprim.hpp>
#include "../../include/r2s_physics.h"
#include <algorithm>
#include <array>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {

constexpr int SLICE = 64;

constexpr int GROUP = 4;               // adjacency slots per software-pipeline group (one 8/16/16-byte load each)
constexpr int TPB = 256;               // threads per workgroup of the auxiliary kernels
constexpr int GRID_DIM = 128;          // wp.HashGrid(128,128,128), spring_mass_warp.py:541
constexpr int GRID_CELL_BITS = 21;     // 128^3 cells
constexpr float MESH_MAX_DIST = 0.02f; // :323
constexpr float WIND_THRESHOLD = 0.6f; // :323
constexpr float NEAR_PAD = 0.03f;      // "contact is likely next step": within margin + 3 cm of a mesh box (0.9 m/s of approach per env step)

struct PhysDev {
    int N, E, n_sub;
    // topology (shared by all envs); all particle indices are INTERNAL (Morton order)
    int nb, cb;                // particle blocks; (block, env) work items per XCD
    int e0, ne;                // environments [e0, e0 + ne) handled by this launch (an env-step may run as parallel chains)
    const int* slice_off;      // [n_slices] first slot of a 64-particle slice (a multiple of 64 * GROUP)
    const int* slice_deg;      // [n_slices] slots per particle of the slice (a multiple of GROUP)
    const int* slice_int;      // [n_slices] 64-particle layout only: the first slice_int slots (a multiple of GROUP) of every particle of the slice
                               // point into the block's OWN records, the rest into its halo (else 0: neighbours in index order)
    // sliced ELL, 10 B per slot in three planes, GROUP-major: the 4 slots n = 4g..4g+3 of lane l of a slice sit together
    // at element (slice_off / 4 + g * 64 + l), so a group costs one 8-byte and two 16-byte coalesced loads per lane
    // (the adjacency stream is shared by every environment and is the largest L2 consumer of the kernel):
    const uint2* adj_idx;      // 4 x u16: BYTE offset (record * 8) of the neighbour in the block's LDS window; padding /
                               // inactive slots point at the owner itself (zero force)
    const float4* adj_k;       // clamp(exp(logY), Ymin, Ymax)
    const float4* adj_ir;      // a = k / rest length
    const int* rslice_off;     // [n_slices] second sliced ELL: neighbours NOT in the LDS window, gathered from global memory
    const int* rslice_deg;     // [n_slices]
    const int4* radj;          // {global particle id, bits(k), bits(k/rest), 0}; padding points at the owner
    const int* halo_off;       // [nb+1]
    const int* halo_ids;       // halo particle ids per block (LDS records B.. in this order)
    const int* perm;           // internal -> user index
    const int* inv;            // user -> internal index
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
    // What a substep hands to its finishing code is kept per substep PARITY ([2][...], par_off(p, step)): with the finishing code of
    // substep k at the head of substep k + 1's launch (p.pf, below) the fused blocks of k + 1 publish while the finishers of k still read.
    size_t par_stride;         // E * N: element offset of the odd substeps' half of vbc / xbc / vdef / cand_mark / mesh_list (mesh_rec: 2 x)
    float4* vbc;               // [2][E,N] v_before_collision published by particles that have candidates; also the velocity of
                               // particles whose large-mesh query is deferred to k_contact_finish
    float4* xbc;               // [2][E,N] position at the top of the substep of every particle the fused kernel leaves to the finishing code
                               // (candidates AND deferred mesh queries): the finishers read positions from here, never from the state arrays
    float4* vdef;              // [2][E,N] velocity of the particles whose mesh query is deferred to k_contact_finish (vbc must keep the
                               // pre-impulse value while other particles' self-collision loops still read it)
    int2* mesh_list;           // [2][...] (env, particle) of the particles deferred in this substep (one list per chain and parity)
    int* mesh_cnt;             // [n_sub + 1] entries of mesh_list per substep; [n_sub] = particles NEAR a mesh over the whole env step
                               // (margin + NEAR_PAD: what the host picks the next step's flavour from); zeroed once per env step
    int mesh_cap, mesh_defer;  // defer = 1: needy particles go to the list; 0: they are queried in place
    int4* mesh_rec;            // large-mesh scenes (MESH 2) list per ENVIRONMENT instead: [E][N][2] self-contained 32-byte records
                               // {candidate count, particle | tag << 31, x0.x, x0.y} {x0.z, v}, counted in rec_cnt [E][n_sub] — the
                               // finishing workgroup knows its environment from blockIdx, so count + record + the mesh's rigid
                               // transform are ONE round trip before the query (the chain-wide list costs two: entry, then state)
    int* rec_cnt;
    int* mq_hint;              // large-mesh scenes: [E,N] the cluster of a particle's closest face in its last query (-1 none; null: no hints) — where its
                               // next substep's first query looks first (round 5).  Plain loads and stores: a stale hint costs time, never the answer
    int* cand_mark;            // [E,N] = substep + 1 when a particle with candidates was handed to the mesh list in that substep
    const int2* cand_list;     // [E][N] per ENVIRONMENT: (env | candidate count << 12, particle) of its particles with candidates, cand_cnt_env[e] of them
    const int* cand_count;     // all of them
    const int* cand_cnt_env;   // [E]
    // meshes
    int n_mesh, n_dyn_mesh, nF, nV, n_dyn_pts;
    const int* faces;          // [nF,3] global vertex ids, in STORED order (large meshes: Morton-sorted clusters)
    const int* face_orig;      // [nF] stored face -> original (caller) face id
    const int* mesh_map;       // [nF] by ORIGINAL face id
    const int* face_map;       // [nF] by ORIGINAL face id
    const int* mesh_face_off;  // [n_mesh+1]
    int n_cl, n_xf;            // face clusters; large dynamic meshes (one rigid transform each per env and substep)
    const float* cl_box;       // [6][n_cl] rest-frame boxes (clusters of large meshes), component-major: one lane per cluster loads coalesced
    const int* mesh_kind;      // [n_mesh] bit 0: large (> 256 faces: clusters, wave-cooperative); bit 1: not a closed manifold (sign by
                               // exact winding number; closed large meshes use pseudonormals, small meshes always the winding number)
    const int* mesh_xf;        // [n_mesh] transform slot of a large dynamic mesh, else -1
    const int* xf_mesh;        // [n_xf] mesh of a transform slot
    const float* xf;           // [E,n_sub,n_xf,12]
    const float* rest_pts;     // [nV,3] vertices at construction (= rest frame of rigid meshes)
    const float* tri_rest;     // rest-frame corners of every stored face in blocks of 64 faces, [nF/64][9][64]: one lane per face loads
                               // coalesced (nine 36-byte-strided dword loads per lane cost the texture path 18 cache lines each)
    const int4* cl_info;       // [n_cl] {mesh, kind | (transform slot + 1) << 2 | faces << 8, transform slot of the mesh (-1 none), first stored face}: the whole cluster record in one load
    int n_sup, n_small;        // super-clusters (eight consecutive clusters of a large mesh); small meshes of a scene that has a large one
    // the FIRST large mesh's clusters [lm_c0, lm_c0 + lm_nc) are runs of 64 stored faces from lm_f0 (the last one shorter: up to lm_f1): their
    // cl_info record is arithmetic, not a load — the round trip between a hint and its cluster's triangles (cl_info_of)
    int lm_c0, lm_nc, lm_f0, lm_f1, lm_y;  // lm_y: the record's .y without the face count (mesh kind | (transform slot + 1) << 2), lm_nc = 0: none
    int lm_mesh, lm_slot;
    const float* sup_box;      // [6][n_sup] rest-frame boxes, component-major
    const int4* sup_info;      // [n_sup] {first cluster, clusters, transform slot (-1 none), mesh kind}
    const int* small_mesh;     // [n_small] mesh ids
    const float* pnorm;        // [nF,7,3] pseudonormals of stored faces of large meshes: face, a, b, c, ab, bc, ca
    const float* mesh_pts;     // [E,nV,3] (static part is live; dynamic part = positions at t=0)
    const float* interp_pts;   // [E,n_sub,n_dyn_pts,3]
    const float* interp_center;// [E,n_sub,3]
    const float* dyn_vel;      // [E,2,3]
    const float* dyn_omega;    // [E,3]
    const float* aabb_dyn;     // [E,n_sub,n_dyn_mesh,6]
    const float* aabb_static;  // [E,n_mesh-n_dyn_mesh,6]
    float* coll_forces;        // [E,nF,3]
    int* hit_cnt;              // [E] particles that reacted to a mesh in the LAST substep (zeroed with coll_forces)
    int* fault;                // [0] sticky: 1 = a self-collision impulse exceeded the bound the "no mesh in reach" decision relies on; 2 = a
                               // hand-off of the resident stepper timed out; [1] a particle needed a mesh query since the host last looked
    void* xch;                 // resident stepper: exchange array [E][2 buffers][3 planes][N] x 16 B {value, tag, value, tag}
    // resident stepper, mesh-query SERVERS (small scenes; see k_steps_resident): workgroups of the same launch beyond the blocks' own,
    // two wavefronts per served particle
    int srv_slots;             // server wavefront pairs of this launch (0: none — queries in place)
    void* srv_claim;           // [srv_slots] x 128 B, first granule {env * N + particle, 1, first substep of the launch it is served from, 1}; then control words and fault-report state (SRV_CTL_OFF, SRV_DBG_OFF)
    void* srv_rr;              // [E][N] x 256 B of tagged 16-byte granules: line 0 the REQUEST (x0.x x0.y | x0.z v.x | v.y v.z), line 1 the RESULT (xy | z vz | vxy)
    int* srv_ctl;              // [0] next free slot, [1] blocks that have left the launch
    int srv_quad;              // server units are quads of wavefronts (two particles per server workgroup) instead of pairs (four)
    int srv_own;               // servers OWN their particle from the claim on: spring forces from the neighbours' exchange records, velocity update, mesh response, ground (0: one request per substep, round 4's first protocol)
    unsigned spin_limit;       // poll passes before a workgroup of the resident launch gives up (RES_SPIN_LIMIT; R2S_RES_SPIN_LIMIT at create: diagnostics)
    // large batches, contact flavours (round 5): the finishing code of substep k runs at the HEAD of substep k + 1's launch (k_substep_pf)
    int pf;                    // this launch sequence runs that way
    int pf_nfin;               // finishing workgroups at the head of a launch (a multiple of 8: the fused blocks behind them keep their XCDs)
    void* pf_res;              // [E][N] x 128 B: a particle's finished state of substep k as three 16-byte granules {value, tag, value, tag},
                               // tag = k + 1, written through (sc1) by its finisher: ONE writer per 128-byte line (see SRV_LINE)
};
__device__ __forceinline__ size_t par_off(const PhysDev& p, int step) { return (size_t)(step & 1) * p.par_stride; }

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

// the resident stepper's form: the reciprocal mass is formed once per launch (1 ulp off the division above, like its force sum)
__device__ __forceinline__ f3 vel_update_rcp(const PhysDev& p, f3 v0, f3 f0, float m0, float inv_m0)
{
    const f3 grav = mk(0.f, 0.f, -9.8f) * m0 * p.rf;
    const f3 a = (f0 + grav) * inv_m0;
    const f3 v1 = v0 + a * p.dt;
    return v1 * p.drag_factor;
}

// ---- mesh queries ---------------------------------------------------------------------------------
// Closest point on triangle (a,b,c) to q as barycentrics (u of a, v of b) — Ericson, RTCD 5.1.5.
// `region`: 0 face interior, 1/2/3 vertex a/b/c, 4/5/6 edge ab/bc/ca (selects the pseudonormal of large meshes).
__device__ __forceinline__ void closest_bary(f3 a, f3 b, f3 c, f3 q, float& u, float& v, int& region)
{
    const f3 ab = b - a, ac = c - a, ap = q - a;
    const float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) { u = 1.f; v = 0.f; region = 1; return; }
    const f3 bp = q - b;
    const float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) { u = 0.f; v = 1.f; region = 2; return; }
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { const float t = d1 / (d1 - d3); u = 1.f - t; v = t; region = 4; return; }
    const f3 cp = q - c;
    const float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) { u = 0.f; v = 0.f; region = 3; return; }
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { const float w = d2 / (d2 - d6); u = 1.f - w; v = 0.f; region = 6; return; }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        u = 0.f; v = 1.f - w; region = 5; return;
    }
    const float denom = 1.f / (va + vb + vc);
    const float vv = vb * denom, ww = vc * denom;
    u = 1.f - vv - ww; v = vv; region = 0;
}

// Particle state of all environments: 24 bytes per particle as three 8-byte words xy | (z, vz) | vxy — the words of the LDS
// window's three planes, so the fused kernel stages a record with three 8-byte loads and three 8-byte LDS writes, nothing
// to repack and no padding.  Kept as one 24-byte RECORD per particle ([n][3]), not as three planes ([3][n]): a halo gather
// then touches one or two 64-byte sectors instead of three (counter passes: 49 vs 45 MB fetched per launch; measured
// 21.5 vs 22.3 us per batched substep, against 22.3 for the 32-byte {x, v} float4 pairs of round 1).
typedef float v2f __attribute__((ext_vector_type(2)));
struct StateC { const v2f* p; size_t n; };
struct StateM { v2f* p; size_t n; operator StateC() const { return {p, n}; } };
#ifdef R2S_STATE_PLANES // [3][n] planes
__device__ __forceinline__ size_t st_at(size_t n, size_t i, int k) { return (size_t)k * n + i; }
#else                    // [n][3]: one 24-byte record per particle
__device__ __forceinline__ size_t st_at(size_t, size_t i, int k) { return 3 * i + (size_t)k; }
#endif
__device__ __forceinline__ f3 st_x(StateC s, size_t i) { const v2f a = s.p[st_at(s.n, i, 0)], b = s.p[st_at(s.n, i, 1)]; return mk(a.x, a.y, b.x); }
__device__ __forceinline__ float4 st_x4(StateC s, size_t i) { const f3 x = st_x(s, i); return make_float4(x.x, x.y, x.z, 0.f); }
__device__ __forceinline__ void st_store(StateM s, size_t i, f3 x, f3 v)
{
    s.p[st_at(s.n, i, 0)] = (v2f){x.x, x.y}; s.p[st_at(s.n, i, 1)] = (v2f){x.z, v.z}; s.p[st_at(s.n, i, 2)] = (v2f){v.x, v.y};
}

// -DR2S_PHASE_PROBE: wall-clock stamps of one finishing wavefront per particle (k_contact_finish), in program order
#ifdef R2S_PHASE_PROBE
__device__ long long g_query_probe[1024 * 32];
struct QProbe { int wave, n; };
#define R2S_QP_PARAM , QProbe& qp
#define R2S_QP_ARG , qp
#define R2S_QP_DECL(w) QProbe qp = {(w), 0}
#define R2S_QSTAMP() do { if ((threadIdx.x & 63) == 0 && qp.wave >= 0 && qp.wave < 1024 && qp.n < 32) g_query_probe[qp.wave * 32 + qp.n] = (long long)wall_clock64(); ++qp.n; } while (0)
extern "C" int r2s_phys_debug_query_probe(long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_query_probe), sizeof(long long) * (size_t)n * 32);
}
#else
#define R2S_QP_PARAM
#define R2S_QP_ARG
#define R2S_QP_DECL(w) do { } while (0)
#define R2S_QSTAMP() do { } while (0)
#endif

struct MeshHit {
    bool result;
    float sign;
    int face; // ORIGINAL (caller) face id
    f3 pt;    // closest point, world frame
    int mm, fm; // mesh_map / face_map of `face` (filled by mesh_query_regs and mesh_query_block; mesh_query_lane leaves the lookup to the caller)
    int hint;   // mesh_query_block: the cluster of the closest face (where a re-query a few micrometres away should look first)
};

__device__ __forceinline__ float box_dist2(f3 q, const float* bb)
{
    const float dx = fmaxf(fmaxf(bb[0] - q.x, q.x - bb[3]), 0.f);
    const float dy = fmaxf(fmaxf(bb[1] - q.y, q.y - bb[4]), 0.f);
    const float dz = fmaxf(fmaxf(bb[2] - q.z, q.z - bb[5]), 0.f);
    return dx * dx + dy * dy + dz * dz;
}

// wavefront-wide reductions (all 64 lanes must be active).  The callers are lone wavefronts whose instruction stream is the
// critical path of a substep, so the lane exchanges are DPP modifiers (a few cycles each) and four readlanes, not twelve
// dependent ds_bpermute round trips through the LDS crossbar (~0.3 us per 64-bit reduction, measured in k_contact_finish).
template <int CTRL> __device__ __forceinline__ unsigned dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    v = min(v, dpp_u32<0xB1>(v));  // quad_perm [1,0,3,2]
    v = min(v, dpp_u32<0x4E>(v));  // quad_perm [2,3,0,1]
    v = min(v, dpp_u32<0x141>(v)); // row_half_mirror: the other quad pair of each 8 lanes
    v = min(v, dpp_u32<0x140>(v)); // row_mirror: the other half of each row of 16
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                   c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
    const unsigned hi = (unsigned)(v >> 32);
    const unsigned mh = wave_min_u32(hi);
    const unsigned ml = wave_min_u32(hi == mh ? (unsigned)v : 0xffffffffu);
    return ((unsigned long long)mh << 32) | ml;
}
__device__ __forceinline__ float wave_sum(float v)
{
    v += __uint_as_float(dpp_u32<0xB1>(__float_as_uint(v)));
    v += __uint_as_float(dpp_u32<0x4E>(__float_as_uint(v)));
    v += __uint_as_float(dpp_u32<0x141>(__float_as_uint(v)));
    v += __uint_as_float(dpp_u32<0x140>(__float_as_uint(v)));
    return (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)))
         + (__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)) + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48)));
}
__device__ __forceinline__ float bcast(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ int bcasti(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// rigid transform of a large dynamic mesh at (env, substep): world = R * rest + t, stored row-major R[9] then t[3]
struct Xf {
    float r[9], t[3];
};
__device__ __forceinline__ f3 xf_apply(const Xf& X, f3 a)
{
    return mk(X.r[0] * a.x + X.r[1] * a.y + X.r[2] * a.z + X.t[0], X.r[3] * a.x + X.r[4] * a.y + X.r[5] * a.z + X.t[1],
              X.r[6] * a.x + X.r[7] * a.y + X.r[8] * a.z + X.t[2]);
}
__device__ __forceinline__ f3 xf_inverse(const Xf& X, f3 w)
{
    const f3 d = mk(w.x - X.t[0], w.y - X.t[1], w.z - X.t[2]);
    return mk(X.r[0] * d.x + X.r[3] * d.y + X.r[6] * d.z, X.r[1] * d.x + X.r[4] * d.y + X.r[7] * d.z, X.r[2] * d.x + X.r[5] * d.y + X.r[8] * d.z);
}
__device__ __forceinline__ f3 xf_rotate(const Xf& X, f3 a)
{
    return mk(X.r[0] * a.x + X.r[1] * a.y + X.r[2] * a.z, X.r[3] * a.x + X.r[4] * a.y + X.r[5] * a.z, X.r[6] * a.x + X.r[7] * a.y + X.r[8] * a.z);
}
__device__ __forceinline__ Xf xf_load(const PhysDev& p, int e, int step, int m)
{
    Xf X;
    const int k = p.mesh_xf[m];
    if (k < 0) { // static mesh or small mesh: identity
#pragma unroll
        for (int j = 0; j < 9; ++j) X.r[j] = (j % 4 == 0) ? 1.f : 0.f;
        X.t[0] = X.t[1] = X.t[2] = 0.f;
        return X;
    }
    const float* src = p.xf + (((size_t)e * p.n_sub + step) * p.n_xf + k) * 12;
#pragma unroll
    for (int j = 0; j < 9; ++j) X.r[j] = src[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) X.t[j] = src[9 + j];
    return X;
}

// vertex of a SMALL mesh in world frame at (env, substep): dynamic vertices follow the interpolated motion
__device__ __forceinline__ f3 mesh_vertex(const PhysDev& p, int e, int step, int vid)
{
    if (vid < p.n_dyn_pts) return ld3(p.interp_pts, ((size_t)e * p.n_sub + step) * p.n_dyn_pts + vid);
    return ld3(p.mesh_pts, (size_t)e * p.nV + vid);
}

// wp.mesh_query_point_sign_winding_number(mesh, q, max_dist=0.02, accuracy=3.0, threshold=0.6) restated for scenes with a
// large mesh, answered by a whole WORKGROUP for one point:
//   closest point  = lexicographic minimum of (squared distance, original face id) over every face with distance^2 <
//                    max_dist^2 — the first strict minimum of a sequential scan.  The faces of a large mesh are Morton-sorted
//                    runs of 64 (clusters) in groups of eight (super-clusters), with rest-frame boxes on both levels, queried
//                    in the rest frame through the substep's rigid transform; small meshes of the same scene are visited
//                    through the face table with their per-substep boxes.
//   sign           = for a large closed manifold (checked at construction) the angle-weighted pseudonormal of the closest
//                    feature (Baerentzen & Aanaes), which equals the winding-number sign for closed meshes; otherwise the
//                    exact winding number over the faces of every mesh that is not a large closed manifold.
// What bounds this code is not memory latency but the INSTRUCTION STREAM of a lone wavefront (about 2 ns per instruction with
// nothing else to issue: a first version that scanned 512 cluster boxes per wavefront, eight per lane, spent 3.4 of its 20 us
// there — in-kernel wall-clock stamps, tools/probes/query_probe.py): so the work per wavefront is kept short —
//   1. every wavefront: one super-cluster box per lane, the eight clusters of the nearest one, the 64 faces of the nearest of
//      those (a tight `best` before anything else is looked at);
//   2. every wavefront: the super-clusters still closer than `best` (one lane each), their clusters (eight super-clusters per
//      round, one lane per cluster): the clusters still closer than `best` are the candidates;
//   3. candidate r is visited by wavefront r % WPB (one lane per face); the wavefronts' results meet in LDS.
// Steps 1-2 are computed redundantly (identically) by every wavefront, so there is one barrier per query, executed whether or
// not the query is wanted (`want` must be uniform over the workgroup).
__device__ __forceinline__ const float* tri_ptr(const PhysDev& p, int f) { return p.tri_rest + ((size_t)(f >> 6) * 9) * 64 + (f & 63); } // + k * 64
__device__ __forceinline__ Xf xf_load_slot(const PhysDev& p, int e, int step, int k)
{
    Xf X;
    const float* src = p.xf + (((size_t)e * p.n_sub + step) * p.n_xf + k) * 12;
#pragma unroll
    for (int j = 0; j < 9; ++j) X.r[j] = src[j];
#pragma unroll
    for (int j = 0; j < 3; ++j) X.t[j] = src[9 + j];
    return X;
}

constexpr int QWPB = 4; // wavefronts per query (k_contact_finish's workgroup)
struct QShare {
    unsigned long long key[2][QWPB]; // double-buffered by query parity: a fast wavefront's next result must not overwrite
    float pt[2][QWPB][6];            // what a slow one is still reading (closest point, q - p; mesh frame)
    int meta[2][QWPB][5];            // stored face, feature region, mesh kind, transform slot, cluster
    volatile int sup[QWPB][8];       // per wavefront: lanes that own the super-clusters of the current round
    int arrived[4];                  // server units (pairs / quads of wavefronts): the last barrier generation each wavefront has arrived at
    unsigned spin;                   // server units: passes of the unit barrier's wait before it gives up (4 x PhysDev::spin_limit: an LDS read per pass)
    float fs[2][4];                  // pair mode, owning servers: the two wavefronts' sums of the particle's spring forces
};
// mesh_query_regs is run by TWO wavefronts: a 128-thread workgroup of k_contact_finish<3> (barrier = __syncthreads), or one of the four
// wavefront PAIRS of a server workgroup of k_steps_resident, each on its own particle at its own pace (barrier = a counter in the pair's
// QShare).  `parity` carries the mode: bit 0 the buffer parity, bit 8 pair mode, bits 16.. the pair barrier's generation.
constexpr int QPAIR = 1 << 8, QFAIL = 1 << 9, QQUAD = 1 << 10; // QFAIL: a partner never arrived (bounded wait; the caller reports a fault and leaves)
// QQUAD (with QPAIR): FOUR wavefronts per query — a lone wavefront issues one instruction per four cycles whatever its parallelism, so
// the per-triangle arithmetic of a query (closest point: ~250 instructions; solid angle: ~200) is split by KIND over the four SIMDs of
// the CU: wavefronts 0, 1 the closest points of triangles 0..63 / 64..127, wavefronts 2, 3 their solid angles.
// A hardware barrier (s_barrier) counts the wavefronts of the WORKGROUP; a unit is two or four of a server workgroup's eight.  Each
// wavefront of the unit has a word in the unit's QShare with the last generation it has arrived at: arriving is one LDS store of that
// number (every active lane stores the same value to the same address — nothing to elect, correct for any lane mask the compiler may
// have split the call into), waiting is reading the partners' words until they say the same.  A wavefront's LDS operations execute in
// order, so a partner that sees the number also sees what was written before it.  The wait is bounded (QFAIL -> the server reports
// fault 4 and leaves).  (First form: an arrival counter, fetch-add by an elected lane + spin with s_sleep; this one is the same speed
// and half the code.  A non-inlined function gives the kernel a stack: scratch memory, and with it fewer resident workgroups than the
// launch needs — measured: hand-offs timing out all over the rope.)
__device__ __forceinline__ void pair_barrier(QShare& sm, int& parity)
{
    const int gen = (parity >> 16) + 1;
    parity = (parity & 0xffff) | (gen << 16);
    const bool quad = (parity & QQUAD) != 0;
    const int w = (int)(threadIdx.x >> 6) & (quad ? 3 : 1);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // this wavefront's LDS writes before its arrival
    __hip_atomic_store(&sm.arrived[w], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    for (unsigned spins = 0;; ++spins) {
        int m = __hip_atomic_load(&sm.arrived[w ^ 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (quad) m = min(m, min(__hip_atomic_load(&sm.arrived[w ^ 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), __hip_atomic_load(&sm.arrived[w ^ 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)));
        if (m >= gen) break;
        if (spins >= sm.spin) { parity |= QFAIL; break; }            // a fraction of a second by default (R2S_RES_SPIN_LIMIT shortens it with the other limits): a partner is gone (never in a sound launch)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// What the two queries of a particle's substep (mesh_collision's query and the gripper / pusher branch's re-query, :322-324, :397) share,
// and what does not depend on the particle at all — so that neither is a dependent round trip inside a query (round 5: the pusher's
// finishing launch spent ~2.5 of a query's ~6.5 us waiting for cluster records and triangle corners it had just had in registers):
__device__ __forceinline__ int4 cl_info_of(const PhysDev& p, int c) // p.cl_info[c], computed where the layout allows it
{
    const int k = c - p.lm_c0;
    if (k >= 0 && k < p.lm_nc) {
        const int f0 = p.lm_f0 + 64 * k;
        return make_int4(p.lm_mesh, p.lm_y | (min(64, p.lm_f1 - f0) << 8), p.lm_slot, f0);
    }
    return p.cl_info[c];
}
struct FaceRegs { f3 a, b, c; int forig; };   // this lane's face of a cluster: rest-frame corners, original (caller) face id
struct BlkAux {
    Xf X;              // rigid transform of the first large dynamic mesh at (env, substep): loaded with the particle's record
    int hint;          // the cluster to look at first in the substep's FIRST query: the particle's closest cluster one substep ago (-1: search)
    int c0;            // the cluster whose faces are in `fr` (-1 none): a re-query a few micrometres away starts there without a load
    int4 c0_info;      // its cl_info record
    FaceRegs fr;
    bool sup_ok;       // n_sup <= 64: this lane's super-cluster record and rest-frame box, loaded at kernel entry (they depend on the lane only)
    int4 si;
    float sbox[6];
};
__device__ __forceinline__ void blk_aux_init(const PhysDev& p, BlkAux& A, int lane)
{
#pragma unroll
    for (int j = 0; j < 9; ++j) A.X.r[j] = (j % 4 == 0) ? 1.f : 0.f;
    A.X.t[0] = A.X.t[1] = A.X.t[2] = 0.f;
    A.hint = -1; A.c0 = -1; A.c0_info = make_int4(0, 0, -1, 0);
    A.fr.a = A.fr.b = A.fr.c = mk(0.f, 0.f, 0.f); A.fr.forig = 0;
    A.sup_ok = p.n_sup > 0 && p.n_sup <= 64;
    A.si = make_int4(0, 0, -1, 0);
#pragma unroll
    for (int k = 0; k < 6; ++k) A.sbox[k] = 0.f;
    if (A.sup_ok) {
        const int s = min(lane, p.n_sup - 1);
        A.si = p.sup_info[s];
#pragma unroll
        for (int k = 0; k < 6; ++k) A.sbox[k] = p.sup_box[k * p.n_sup + s];
    }
}
__device__ __forceinline__ MeshHit mesh_query_block(const PhysDev& p, int step, f3 q, int e, bool want, int hint, QShare& sm, int& parity, BlkAux& A R2S_QP_PARAM)
{
    MeshHit out = {false, 0.f, 0, mk(0.f, 0.f, 0.f), 0, 0, -1};
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const float MAXD2 = MESH_MAX_DIST * MESH_MAX_DIST;
    float best = MAXD2;
    unsigned long long bestkey = ~0ull;
    f3 bcp = mk(0.f, 0.f, 0.f), bdl = mk(0.f, 0.f, 0.f); // closest point and q - p, both in the mesh's frame
    int bstored = 0, bregion = 0, bkind = 0, bxf = -1, bcl = -1;
    const Xf& X0 = A.X;
    // X0: the first large dynamic mesh's transform of this (env, substep), loaded by the caller together with the particle's
    // state (one load for both queries of a particle); further ones (rare) are fetched where needed
    if (want) {
        const f3 q_rest0 = p.n_xf > 0 ? xf_inverse(X0, q) : q;
        auto rest_point = [&](int k) -> f3 { // the query point in the frame the triangles of transform slot k are stored in
            return k < 0 ? q : (k == 0 ? q_rest0 : xf_inverse(xf_load_slot(p, e, step, k), q));
        };
        auto box6 = [&](const float* base, int stride, int idx, f3 qq) -> float { // component-major boxes
            const float dx = fmaxf(fmaxf(base[idx] - qq.x, qq.x - base[3 * stride + idx]), 0.f);
            const float dy = fmaxf(fmaxf(base[stride + idx] - qq.y, qq.y - base[4 * stride + idx]), 0.f);
            const float dz = fmaxf(fmaxf(base[2 * stride + idx] - qq.z, qq.z - base[5 * stride + idx]), 0.f);
            return dx * dx + dy * dy + dz * dz;
        };
        // one lane per face of a run of stored faces (a = b = c3 come in the frame `qq` is in)
        auto reduce = [&](unsigned long long key, f3 cp, f3 qq, int region, int stored, int kind, int slot, int cluster) {
            const unsigned long long mn = wave_min_u64(key);
            if (mn < bestkey) {
                bestkey = mn;
                best = __uint_as_float((unsigned)(mn >> 32));
                const int w = __builtin_ctzll(__builtin_amdgcn_ballot_w64(key == mn));
                bcp = mk(bcast(cp.x, w), bcast(cp.y, w), bcast(cp.z, w));
                bdl = qq - bcp;
                bstored = bcasti(stored, w); bregion = bcasti(region, w);
                bkind = kind; bxf = slot; bcl = cluster;
            }
        };
        auto load_faces = [&](int4 ci) -> FaceRegs { // this lane's face of the cluster with record `ci` (lanes beyond its face count: its first face)
            const int f = ci.w + (lane < (ci.y >> 8) ? lane : 0);
            const float* t9 = tri_ptr(p, f);
            FaceRegs r;
            r.a = mk(t9[0], t9[64], t9[128]); r.b = mk(t9[192], t9[256], t9[320]); r.c = mk(t9[384], t9[448], t9[512]);
            r.forig = p.face_orig[f];
            return r;
        };
        auto visit = [&](int cluster, int4 ci, const FaceRegs& fr) { // a cluster of a large mesh: rest-frame triangle records
            const int nf = ci.y >> 8, kind = ci.y & 3, slot = ci.z;
            const f3 qq = rest_point(slot);
            const bool act = lane < nf;
            float u, v;
            int region;
            closest_bary(fr.a, fr.b, fr.c, qq, u, v, region);
            const f3 cp = fr.a * u + fr.b * v + fr.c * (1.f - u - v);
            const f3 d = cp - qq;
            const float d2 = dot(d, d);
            const unsigned long long key = (act && d2 < MAXD2) ? (((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)fr.forig) : ~0ull;
            reduce(key, cp, qq, region, ci.w + (act ? lane : 0), kind, slot, cluster);
        };
        // ---- small meshes of the scene (gripper fingers next to a large obstacle): world frame, through the face table
        for (int k = 0; k < p.n_small; ++k) {
            const int m = p.small_mesh[k];
            const float* bb = m < p.n_dyn_mesh ? p.aabb_dyn + (((size_t)e * p.n_sub + step) * p.n_dyn_mesh + m) * 6
                                               : p.aabb_static + ((size_t)e * (p.n_mesh - p.n_dyn_mesh) + (m - p.n_dyn_mesh)) * 6;
            if (!(box_dist2(q, bb) < best * 1.0001f + 1e-12f)) continue;
            const int kind = p.mesh_kind[m];
            for (int fb = p.mesh_face_off[m]; fb < p.mesh_face_off[m + 1]; fb += 64) {
                const int f = fb + lane;
                unsigned long long key = ~0ull;
                f3 cp = mk(0.f, 0.f, 0.f);
                int region = 0;
                if (f < p.mesh_face_off[m + 1]) {
                    const f3 a = mesh_vertex(p, e, step, p.faces[3 * f]), b = mesh_vertex(p, e, step, p.faces[3 * f + 1]),
                             c3 = mesh_vertex(p, e, step, p.faces[3 * f + 2]);
                    float u, v;
                    closest_bary(a, b, c3, q, u, v, region);
                    cp = a * u + b * v + c3 * (1.f - u - v);
                    const f3 d = cp - q;
                    const float d2 = dot(d, d);
                    if (d2 < MAXD2) key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)p.face_orig[f];
                }
                reduce(key, cp, q, region, f, kind, -1, -1);
            }
        }
        // ---- large meshes
        int C0 = -1; // the cluster every wavefront has visited
        // `best0`: the bound after the part every wavefront computed identically (small meshes, the hinted / nearest cluster).
        // The ballots that NUMBER the candidates (smask, cm, r) use it, so that candidate r is the same cluster in every
        // wavefront; the per-wavefront `best` — which diverges as soon as the wavefronts visit different clusters — only prunes
        // a wavefront's own visits.  (Round 2 numbered with the diverging bound: a cluster could get a different rank in
        // different wavefronts and be visited by none — more than eight super-clusters in reach, or more than 64 in total.)
        float best0 = best;
        if (hint >= 0) { // a query next to a previous answer: its cluster first, no search for the nearest box — and no load when its faces are still here
            C0 = hint;
            if (A.c0 != C0) { A.c0_info = cl_info_of(p, C0); A.fr = load_faces(A.c0_info); A.c0 = C0; }
            visit(C0, A.c0_info, A.fr);
            R2S_QSTAMP(); // nearest cluster done
        }
        for (int sb = 0; sb < p.n_sup; sb += 64) {
            const int s = min(sb + lane, p.n_sup - 1);
            const bool pre = sb == 0 && A.sup_ok;
            const int4 si = pre ? A.si : p.sup_info[s]; // {first cluster, clusters, transform slot, mesh kind}
            float d2s;
            if (pre) {
                const f3 qq = rest_point(si.z);
                const float dx = fmaxf(fmaxf(A.sbox[0] - qq.x, qq.x - A.sbox[3]), 0.f), dy = fmaxf(fmaxf(A.sbox[1] - qq.y, qq.y - A.sbox[4]), 0.f),
                            dz = fmaxf(fmaxf(A.sbox[2] - qq.z, qq.z - A.sbox[5]), 0.f);
                d2s = dx * dx + dy * dy + dz * dz;
            } else
                d2s = box6(p.sup_box, p.n_sup, s, rest_point(si.z));
            if (sb + lane >= p.n_sup) d2s = 3.0e38f;
            if (sb == 0 && bestkey == ~0ull && hint < 0) { // step 1: nearest first
                const unsigned long long near = wave_min_u64(((unsigned long long)__float_as_uint(d2s) << 32) | (unsigned)lane);
                if (__uint_as_float((unsigned)(near >> 32)) < best * 1.0001f + 1e-12f) {
                    const int L = (int)(near & 63);
                    const int c0 = bcasti(si.x, L), ncl = bcasti(si.y, L), slot = bcasti(si.z, L);
                    const int4 cil = p.cl_info[c0 + min(lane, max(ncl - 1, 0))];   // every lane its cluster's record, with the boxes: no round trip behind the choice
                    const float d2c = lane < ncl ? box6(p.cl_box, p.n_cl, c0 + lane, rest_point(slot)) : 3.0e38f;
                    const unsigned long long nc = wave_min_u64(((unsigned long long)__float_as_uint(d2c) << 32) | (unsigned)lane);
                    if (__uint_as_float((unsigned)(nc >> 32)) < best * 1.0001f + 1e-12f) {
                        const int Lc = (int)(nc & 63);
                        C0 = c0 + Lc;
                        A.c0_info = make_int4(bcasti(cil.x, Lc), bcasti(cil.y, Lc), bcasti(cil.z, Lc), bcasti(cil.w, Lc));
                        A.fr = load_faces(A.c0_info); A.c0 = C0;
                        visit(C0, A.c0_info, A.fr);
                    }
                }
                R2S_QSTAMP(); // nearest cluster done
            }
            if (sb == 0) best0 = best; // identical in every wavefront up to here
            // step 2: the super-clusters still in reach, eight per round
            unsigned long long smask = __builtin_amdgcn_ballot_w64(d2s < best0 * 1.0001f + 1e-12f);
            int r = 0; // running candidate number (the same in every wavefront)
            while (smask) {
                const int rank = __builtin_popcountll(smask & ((1ull << lane) - 1ull));
                if (((smask >> lane) & 1ull) && rank < 8) sm.sup[wave][rank] = lane;
                const int cnt = min(__builtin_popcountll(smask), 8);
                for (int k = 0; k < cnt; ++k) smask &= smask - 1;
                const int idx = lane >> 3, j = lane & 7;
                const int owner = sm.sup[wave][idx];
                const int c0 = __shfl(si.x, owner), ncl = __shfl(si.y, owner), slot = __shfl(si.z, owner);
                const bool valid = idx < cnt && j < ncl;
                const int c = valid ? c0 + j : 0;
                const int4 cil = p.cl_info[c];                              // with the box (same round trip): a candidate's record is a readlane away
                const float d2c = valid ? box6(p.cl_box, p.n_cl, c, rest_point(slot)) : 3.0e38f;
                unsigned long long cm = __builtin_amdgcn_ballot_w64(d2c < best0 * 1.0001f + 1e-12f && c != C0);
                while (cm) { // step 3: this wavefront's share of the candidates
                    const int L = __builtin_ctzll(cm);
                    cm &= cm - 1;
                    if ((r++ % QWPB) != wave) continue;
                    if (!(bcast(d2c, L) < best * 1.0001f + 1e-12f)) continue; // cannot beat this wavefront's best any more
                    const int cc = bcasti(c, L);
                    const int4 ci = make_int4(bcasti(cil.x, L), bcasti(cil.y, L), bcasti(cil.z, L), bcasti(cil.w, L));
                    visit(cc, ci, load_faces(ci));
                }
            }
        }
        R2S_QSTAMP(); // this wavefront's candidates done
    }
    // ---- the wavefronts' results meet (always: the number of barriers must not depend on the data)
    const int par = parity;
    parity ^= 1;
    if (lane == 0) {
        sm.key[par][wave] = bestkey;
        sm.pt[par][wave][0] = bcp.x; sm.pt[par][wave][1] = bcp.y; sm.pt[par][wave][2] = bcp.z;
        sm.pt[par][wave][3] = bdl.x; sm.pt[par][wave][4] = bdl.y; sm.pt[par][wave][5] = bdl.z;
        sm.meta[par][wave][0] = bstored; sm.meta[par][wave][1] = bregion; sm.meta[par][wave][2] = bkind; sm.meta[par][wave][3] = bxf; sm.meta[par][wave][4] = bcl;
    }
    __syncthreads();
    if (!want) return out;
    int fw = 0;
    bestkey = sm.key[par][0];
#pragma unroll
    for (int w = 1; w < QWPB; ++w) {
        const unsigned long long k = sm.key[par][w];
        if (k < bestkey) { bestkey = k; fw = w; }
    }
    bcp = mk(sm.pt[par][fw][0], sm.pt[par][fw][1], sm.pt[par][fw][2]);
    bdl = mk(sm.pt[par][fw][3], sm.pt[par][fw][4], sm.pt[par][fw][5]);
    bstored = sm.meta[par][fw][0]; bregion = sm.meta[par][fw][1]; bkind = sm.meta[par][fw][2]; bxf = sm.meta[par][fw][3]; bcl = sm.meta[par][fw][4];
    const bool found = bestkey != ~0ull;
    const int bface = found ? (int)(unsigned)(bestkey & 0xffffffffull) : 0; // a miss reports face 0, like warp's zero-initialised query
    float sign = 1.f;
    const int mm = p.mesh_map[bface], fm = p.face_map[bface]; // in flight with the pseudonormal
    f3 bpt = bcp;
    if (found) {
        if (bxf >= 0) bpt = bxf == 0 ? xf_apply(X0, bcp) : xf_apply(xf_load_slot(p, e, step, bxf), bcp);
        if (bkind == 1) {
            const f3 n = ld3(p.pnorm, (size_t)bstored * 7 + bregion); // rest frame, like bdl
            sign = dot(bdl, n) < 0.f ? -1.f : 1.f;
        } else {
            // exact winding number (the reference's sign rule, :322-324) over the faces of every mesh that is not a
            // large closed manifold (those contribute 0 outside themselves).  Faces of a large open mesh are visited in
            // its rest frame (solid angles are rotation invariant).
            float ws = 0.f;
            for (int m = 0; m < p.n_mesh; ++m) {
                const int kind = p.mesh_kind[m];
                if (kind == 1) continue;
                const bool rest = (kind & 1) != 0;
                const f3 qm = rest ? xf_inverse(xf_load(p, e, step, m), q) : q;
                for (int f = p.mesh_face_off[m] + lane; f < p.mesh_face_off[m + 1]; f += 64) {
                    f3 a, b, c3;
                    if (rest) {
                        const float* t9 = tri_ptr(p, f);
                        a = mk(t9[0], t9[64], t9[128]) - qm; b = mk(t9[192], t9[256], t9[320]) - qm; c3 = mk(t9[384], t9[448], t9[512]) - qm;
                    } else {
                        a = mesh_vertex(p, e, step, p.faces[3 * f]) - qm; b = mesh_vertex(p, e, step, p.faces[3 * f + 1]) - qm;
                        c3 = mesh_vertex(p, e, step, p.faces[3 * f + 2]) - qm;
                    }
                    const float la = len(a), lb = len(b), lc = len(c3);
                    const float det = dot(a, cross(b, c3));
                    const float den = la * lb * lc + dot(a, b) * lc + dot(b, c3) * la + dot(c3, a) * lb;
                    ws += 2.f * atan2f(det, den);
                }
            }
            const float wn = wave_sum(ws) / (float)(4.0 * 3.14159265358979323846);
            sign = wn > WIND_THRESHOLD ? -1.f : 1.f;
        }
    }
    R2S_QSTAMP(); // sign done
    out.result = found && lane == 0; // the answer belongs to the particle of lane 0 (the other lanes only helped)
    out.sign = sign;
    out.face = bface;
    out.pt = bpt;
    out.mm = mm;
    out.fm = fm;
    out.hint = found ? bcl : -1;
    return out;
}

// Per-lane version of the same query for scenes whose meshes are all small (gripper fingers, box obstacles): plain
// loops over the faces of the meshes whose AABB is within max_dist, exact winding number over all faces.  It keeps the
// fused substep kernel at 60 VGPRs (the cooperative version needs 94, one occupancy step lower), so kernels are
// instantiated for both and the handle picks by scene.
__device__ MeshHit mesh_query_lane(const PhysDev& p, int e, int step, f3 q, bool want)
{
    MeshHit r = {false, 0.f, 0, mk(0.f, 0.f, 0.f), 0, 0};
    if (!want) return r;
    float best = MESH_MAX_DIST * MESH_MAX_DIST;
    const float cull = best * 1.0001f + 1e-12f;
    for (int m = 0; m < p.n_mesh; ++m) {
        const float* bb = m < p.n_dyn_mesh ? p.aabb_dyn + (((size_t)e * p.n_sub + step) * p.n_dyn_mesh + m) * 6
                                           : p.aabb_static + ((size_t)e * (p.n_mesh - p.n_dyn_mesh) + (m - p.n_dyn_mesh)) * 6;
        if (box_dist2(q, bb) > cull) continue;
        for (int f = p.mesh_face_off[m]; f < p.mesh_face_off[m + 1]; ++f) { // stored order == original order for small meshes
            const f3 a = mesh_vertex(p, e, step, p.faces[3 * f]), b = mesh_vertex(p, e, step, p.faces[3 * f + 1]),
                     c = mesh_vertex(p, e, step, p.faces[3 * f + 2]);
            float u, v;
            int region;
            closest_bary(a, b, c, q, u, v, region);
            const f3 cp = a * u + b * v + c * (1.f - u - v);
            const f3 d = cp - q;
            const float d2 = dot(d, d);
            if (d2 < best) { best = d2; r.result = true; r.face = f; r.pt = cp; }
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

// Small scenes (every mesh small, <= 128 faces in total: two 44-face fingers + a box obstacle): k_contact_finish<3> keeps the
// substep's triangles in registers, loaded ONCE per particle (index -> vertex: two dependent round trips) and used by the
// closest-point search, the winding number AND the re-query of a finger contact.  TWO wavefronts per particle, one triangle
// per lane: the instruction stream of a lone wavefront is what a listed particle costs (about 3 ns per instruction with
// nothing else to issue; closest point + solid angle of a triangle are ~300 instructions), so two triangles per lane in
// one wavefront cost 2.3 us per query and one triangle per lane in two wavefronts about half (in-kernel stamps,
// tools/probes/query_probe.py).  Both wavefronts run the whole finishing code on the same particle; the first one stores.
struct TriRegs {
    f3 a, b, c;
    int mm, fm, face; // mesh_map / face_map of the lane's face
    bool ok;
    f3 ctr, om, dv0, dv1; // the substep's eef centre, angular velocity and the two finger velocities (same for every lane)
};
// the lane's face: corner ids and caller-side maps do not depend on the particle — loaded at kernel entry, in flight with the list entry
struct TriIds { int ia, ib, ic, mm, fm, face; bool ok; };
__device__ __forceinline__ TriIds load_tri_ids(const PhysDev& p, int lane, int wave)
{
    TriIds d;
    const int f = lane + 64 * wave;
    d.ok = f < p.nF;
    d.face = min(f, p.nF - 1);
    d.ia = p.faces[3 * d.face]; d.ib = p.faces[3 * d.face + 1]; d.ic = p.faces[3 * d.face + 2]; // stored order == caller order for small meshes
    d.mm = p.mesh_map[d.face]; d.fm = p.face_map[d.face];
    return d;
}
__device__ __forceinline__ TriRegs load_tris(const PhysDev& p, int e, int step, const TriIds& d)
{
    TriRegs t;
    t.ok = d.ok; t.face = d.face; t.mm = d.mm; t.fm = d.fm;
    t.a = mesh_vertex(p, e, step, d.ia); t.b = mesh_vertex(p, e, step, d.ib); t.c = mesh_vertex(p, e, step, d.ic);
    t.ctr = ld3(p.interp_center, (size_t)e * p.n_sub + step); t.om = ld3(p.dyn_omega, e);
    t.dv0 = ld3(p.dyn_vel, (size_t)e * 2); t.dv1 = ld3(p.dyn_vel, (size_t)e * 2 + 1);
    return t;
}
// Same answer as mesh_query_lane on such a scene: lexicographic minimum of (distance^2, face id) over the faces closer than
// max_dist, sign from the exact winding number over all faces.  `q` and `want` are uniform over the workgroup (the particle of
// lane 0); one barrier per call whether or not the query is wanted.
template <bool QUAD = false>
__device__ __forceinline__ MeshHit mesh_query_regs(const TriRegs& t, f3 q, bool want, QShare& sm, int& parity)
{
    MeshHit out = {false, 0.f, 0, mk(0.f, 0.f, 0.f), 0, 0};
    constexpr bool quad = QUAD; // (compile time: as a run-time mode the two-wavefront form lost 0.2 us per query to the split)
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6) & (quad ? 3 : 1);
    const bool do_cp = !quad || wave < 2, do_wn = !quad || wave >= 2; // (quad: the two kinds of per-triangle arithmetic on different SIMDs)
    const float MAXD2 = MESH_MAX_DIST * MESH_MAX_DIST;
    const int par = parity & 1;
    parity ^= 1;
    if (want) {
        // (per-triangle arithmetic first, reductions behind it: the wait states of the cross-lane instructions then have the other
        // reduction's instructions to hide behind — the order the two-wavefront form always had)
        f3 cp = mk(0.f, 0.f, 0.f);
        unsigned long long key = ~0ull;
        float sa = 0.f;
        if (do_cp) {
            float u, v;
            int region;
            closest_bary(t.a, t.b, t.c, q, u, v, region);
            cp = t.a * u + t.b * v + t.c * (1.f - u - v);
            const f3 d = cp - q;
            const float d2 = dot(d, d);
            key = (t.ok && d2 < MAXD2) ? (((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)t.face) : ~0ull;
        }
        if (do_wn) {
            const f3 a = t.a - q, b = t.b - q, c3 = t.c - q;
            const float la = len(a), lb = len(b), lc = len(c3);
            const float det = dot(a, cross(b, c3));
            const float den = la * lb * lc + dot(a, b) * lc + dot(b, c3) * la + dot(c3, a) * lb;
            sa = t.ok ? 2.f * atan2f(det, den) : 0.f;
        }
        if (do_wn) {
            const float ws = wave_sum(sa);
            if (lane == 0) sm.pt[par][wave][3] = ws;
        }
        if (do_cp) {
            const unsigned long long mn = wave_min_u64(key);
            const int w = mn != ~0ull ? __builtin_ctzll(__builtin_amdgcn_ballot_w64(key == mn)) : 0;
            const f3 pt = mk(bcast(cp.x, w), bcast(cp.y, w), bcast(cp.z, w));
            const int wmm = bcasti(t.mm, w), wfm = bcasti(t.fm, w), mm0 = bcasti(t.mm, 0), fm0 = bcasti(t.fm, 0);
            if (lane == 0) {
                sm.key[par][wave] = mn;
                sm.pt[par][wave][0] = pt.x; sm.pt[par][wave][1] = pt.y; sm.pt[par][wave][2] = pt.z;
                sm.meta[par][wave][0] = wmm; sm.meta[par][wave][1] = wfm; sm.meta[par][wave][2] = mm0; sm.meta[par][wave][3] = fm0;
            }
        }
    }
    if (parity & QPAIR) pair_barrier(sm, parity);
    else __syncthreads();
    if (!want) return out;
    const unsigned long long k0 = sm.key[par][0], k1 = sm.key[par][1];
    const int fw = k1 < k0 ? 1 : 0;
    const unsigned long long mn = fw ? k1 : k0;
    const bool found = mn != ~0ull;
    const float wn = (quad ? sm.pt[par][2][3] + sm.pt[par][3][3] : sm.pt[par][0][3] + sm.pt[par][1][3]) / (float)(4.0 * 3.14159265358979323846);
    out.result = found && lane == 0; // the answer belongs to the particle of lane 0
    out.sign = wn > WIND_THRESHOLD ? -1.f : 1.f;
    out.face = found ? (int)(unsigned)(mn & 0xffffffffull) : 0; // a miss reports face 0, like warp's zero-initialised query
    out.pt = mk(sm.pt[par][fw][0], sm.pt[par][fw][1], sm.pt[par][fw][2]);
    out.mm = found ? sm.meta[par][fw][0] : sm.meta[par][0][2];
    out.fm = found ? sm.meta[par][fw][1] : sm.meta[par][0][3];
    return out;
}

// ---- finishing at the HEAD of the next launch (large batches in contact; round 5) --------------------------------------------------
// The contact flavours of a large batch ran two dependent launches per substep: the fused kernel, then k_contact_finish for what it
// could not finish in its own thread (deferred mesh queries, particles with self-collision candidates) — 18.6 us of kernels in a
// 23.3 us period per chain of the headline scene, and every block of the next substep waited for the few hundred particles of the
// finishing launch.  With p.pf the finishing code of substep k is the first pf_nfin workgroups of substep k + 1's launch
// (k_substep_pf): it starts at once (workgroups are dispatched in order: it is resident before any fused block), the fused blocks of
// k + 1 start next to it, and only a block that HOLDS an unfinished particle in its window waits for it:
//   * the fused kernel of substep k stores PF_SENT in all six words of the state record of every particle it leaves to the finishers;
//   * a finisher stores the particle's finished state in the particle's own 128-byte line of p.pf_res as three 16-byte granules
//     {value, tag, value, tag}, tag = k + 1, write-through (sc1) — the data is the flag (cdna_hip_programming.md, Guideline 16 R2;
//     the resident stepper's hand-off), one writer per line (a write-through store into a line of which the writer's L2 holds an
//     older copy does not leave the line's other bytes alone: see SRV_LINE);
//   * a block of substep k + 1 that stages a PF_SENT record polls that line with L1-bypassing loads until the three tags read k + 1
//     (bounded: fault code 6, never a hang) and stages the finished record instead;
//   * what the finishers READ — the list, positions (xbc), post-force velocities (vbc / vdef), marks — is kept per substep parity,
//     because the fused blocks of k + 1 publish theirs at the same time; the state arrays are not read by finishers at all;
//   * the last substep of an env step is finished by the stand-alone k_contact_finish, which writes the state array as before: no
//     PF_SENT record survives a r2s_phys_step.
// One launch boundary per substep instead of two, and the finishing latency overlaps the blocks that do not depend on it.  The
// price: the launch carries the registers of the larger role (4 instead of 6 workgroups per CU for the fused blocks of the headline).
// Same arithmetic on the same inputs in the same order as the two-launch flavour: bit-identical states (tests/test_pf_gpu.py).
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr unsigned PF_SENT = 0x7fc5e7a1u;          // a quiet NaN with a payload
constexpr int PF_LINE = 128;
constexpr int PF_AUX_STORE = 16;                   // buffer-instruction cache policy: sc1 = agent scope, write-through
constexpr int PF_AUX_LOAD = 16 | (int)0x80000000;  // sc1 + the compiler-side volatile bit (= sc0 sc1 in the instruction): L1-bypassing
// the first fault of a launch wins and records where it happened (p.fault + 3 .. + 14 = the handle's words [4..15]): code, work item,
// substep, and six words of context — what the host's error message prints
__device__ __forceinline__ void resident_fault(const PhysDev& p, int code, int item, int k, unsigned a, unsigned b, unsigned c, unsigned d, unsigned e2, unsigned f)
{
    if (!p.fault) return;
    if (atomicCAS(p.fault, 0, code) == 0) {
        int* w = p.fault + 3;
        w[0] = code; w[1] = item; w[2] = k; w[3] = (int)a; w[4] = (int)b; w[5] = (int)c; w[6] = (int)d; w[7] = (int)e2; w[8] = (int)f;
    }
}
__device__ __forceinline__ bool pf_pending(v2f a) { return __float_as_uint(a.x) == PF_SENT && __float_as_uint(a.y) == PF_SENT; }
__device__ __forceinline__ void pf_mark(StateM s, size_t i) // "not finished in this launch": every word, so that a reader of any plane sees it
{
    const v2f w = {__uint_as_float(PF_SENT), __uint_as_float(PF_SENT)};
    s.p[st_at(s.n, i, 0)] = w; s.p[st_at(s.n, i, 1)] = w; s.p[st_at(s.n, i, 2)] = w;
}
__device__ __forceinline__ void pf_store(const PhysDev& p, size_t ei, f3 x, f3 v, unsigned tag)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p.pf_res, 0, 0x7fffffff, 0x00020000);
    const unsigned off = (unsigned)ei * (unsigned)PF_LINE;
    const v4u w0 = {__float_as_uint(x.x), tag, __float_as_uint(x.y), tag}, w1 = {__float_as_uint(x.z), tag, __float_as_uint(v.z), tag},
              w2 = {__float_as_uint(v.x), tag, __float_as_uint(v.y), tag};
    __builtin_amdgcn_raw_buffer_store_b128(w0, r, off, 0, PF_AUX_STORE);
    __builtin_amdgcn_raw_buffer_store_b128(w1, r, off + 16u, 0, PF_AUX_STORE);
    __builtin_amdgcn_raw_buffer_store_b128(w2, r, off + 32u, 0, PF_AUX_STORE);
}
// the finished record (state words xy | z vz | vxy) of particle ei = env * N + particle from the substep before `step`; waits for it
__device__ __forceinline__ void pf_wait(const PhysDev& p, size_t ei, int step, int item, v2f& a, v2f& b, v2f& c)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p.pf_res, 0, 0x7fffffff, 0x00020000);
    const unsigned off = (unsigned)ei * (unsigned)PF_LINE, tag = (unsigned)step;
    for (unsigned spins = 0;; ++spins) {
        const v4u d0 = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, PF_AUX_LOAD), d1 = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16u, 0, PF_AUX_LOAD),
                  d2 = __builtin_amdgcn_raw_buffer_load_b128(r, off + 32u, 0, PF_AUX_LOAD);
        if (d0.y == tag && d0.w == tag && d1.y == tag && d1.w == tag && d2.y == tag && d2.w == tag) {
            a = (v2f){__uint_as_float(d0.x), __uint_as_float(d0.z)}; b = (v2f){__uint_as_float(d1.x), __uint_as_float(d1.z)};
            c = (v2f){__uint_as_float(d2.x), __uint_as_float(d2.z)};
            return;
        }
        if (spins >= p.spin_limit) { // the finisher never delivered: the state is invalid from here on, and the next r2s_phys_step says so
            resident_fault(p, 6, item, step, (unsigned)ei, d0.y, d0.w, d1.y, d2.y, spins);
            a = (v2f){0.f, 0.f}; b = a; c = a;
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// ---- spring forces: gather form of eval_springs (:61-104) ----------------------------------------
// Force on particle i from neighbour j:  [k (L/rest - 1) + c ((vj - vi) . d)] d,  d = (xj - xi) / max(L, 1e-6).
// This is exactly the reference's +F on springs[s][0] and -F on springs[s][1] (the sign flips cancel), summed
// in adjacency order instead of atomic order.  The hot loop: FMA contraction allowed, 1-ulp rsq instead of
// sqrt + three divides (the reference's own float atomics reorder sums far more than this perturbs them).
#pragma clang fp contract(fast)

// One neighbour, 18 VALU instructions: with d = xj - xi (NOT normalised), r = 1 / |d|, L = |d|, t = (vj - vi) . d
//     F = [k (L / rest - 1) + c (dv . d r)] d r  =  [(a L - k) + (c r) t] r d  =  [a - k r + c r^2 t] d,      a = k / rest (per slot),
// so the unit vector is never formed (3 multiplies), L / rest - 1 and the stiffness product fold into one FMA, and the
// 1e-6 floor of the reference's normalisation (d / max(L, 1e-6), :84) becomes a 1e-30 seed of the squared length: padding
// slots (d = 0, k = a = 0, dv = 0) contribute exactly 0 without a v_max, real springs (rest > 1e-4) never get near it.
// (x, y) pairs ride in packed registers (v_pk_add / v_pk_fma: one instruction, same issue time as two scalar ones).
__device__ __forceinline__ void spring_term(v2f xy, float zj, v2f vxy, float vzj, f3 xi, f3 vi, float k, float a, float dashpot,
                                            v2f& fxy, float& fz)
{
    const v2f dxy = xy - (v2f){xi.x, xi.y};
    const float dz = zj - xi.z;
    const float d2 = fmaf(dxy.x, dxy.x, fmaf(dxy.y, dxy.y, fmaf(dz, dz, 1e-30f)));
    const float rinv = __builtin_amdgcn_rsqf(d2);
    const v2f dvxy = vxy - (v2f){vi.x, vi.y};
    const float dvz = vzj - vi.z;
    const float t = fmaf(dvxy.x, dxy.x, fmaf(dvxy.y, dxy.y, dvz * dz));
    // sc = [(a L - k) + (c r) t] r  with  L r = |d|^2 r^2 = 1:  a - k r + (c r^2) t — one VALU instruction fewer per slot than
    // forming the magnitude first (round 3: 19.5 -> 19.2 us per batched substep; same rounding class: both cancel a against k r)
    const float sc = fmaf(dashpot * (rinv * rinv), t, fmaf(-k, rinv, a));
    fxy += dxy * sc;
    fz = fmaf(dz, sc, fz);
}

// Hot path.  The block's LDS window is three 8-byte planes  xy[RCAP] | (z, vz)[RCAP] | vxy[RCAP]  with a compile-time
// capacity, and the adjacency stores the neighbour's BYTE offset (record * 8): a slot is three ds_read_b64 off ONE
// address register with immediate plane offsets — no address arithmetic beyond unpacking the u16.  The adjacency is
// read in groups of 4 slots (one 8-byte + two 16-byte coalesced loads per lane).  ALL THREE words of group g+1 are in
// flight while group g is evaluated (ping-pong registers, no copies), and those of group 0 are issued BEFORE the staging
// barrier (see substep_body): the adjacency is an L2 stream shared by the environments, ~0.6 us away under load, and a
// wavefront walks 9 groups — with the stiffness words loaded at the start of their own group (round 1) every group
// exposed that latency and the gather was bound by it, not by VALU issue (cutting 15 % of its instructions changed nothing).
// Byte offsets of the window planes.  One record of padding between planes on purpose: with plane strides that are multiples
// of 512 B the compiler fuses two of a slot's three reads into one ds_read2st64_b64 — which the LDS serves at HALF the rate of
// two ds_read_b64 (MI355X_MICROARCH.md, LDS table: 8 vs 2 + 2 cycles per wavefront instruction).  The gather is LDS-bound
// (3 reads per slot, ~35 slots per particle), so the fused form costs 10 LDS cycles per slot instead of 6.
template <int RCAP> __device__ __forceinline__ constexpr int PLANE1() { return RCAP * 8 + 8; }
template <int RCAP> __device__ __forceinline__ constexpr int PLANE2() { return 2 * (RCAP * 8 + 8); }

struct AdjGroup {
    uint2 idx;     // 4 x u16 window byte offsets
    float4 k, a;   // stiffness, stiffness / rest length
};
// `row` is the group's first element and WAVE-UNIFORM (slice offset / 4 + group * 64, both scalar), `lane` the only per-lane part:
// the three loads then use a scalar base with one loop-invariant 32-bit lane offset each, instead of a 64-bit per-lane address
// computed with vector instructions for every group (round 3: 20 of the 148 VALU instructions of a two-group loop trip were that).
__device__ __forceinline__ AdjGroup adj_load(const PhysDev& p, int row, int lane)
{
    // raw buffer loads: scalar resource + scalar element offset (`row`) + one 32-bit lane offset — buffer_load_dwordx2 / x4 ... offen.
    // (Plain pointer arithmetic with a uniform base still compiled to a 64-bit vector add per load.)  Word 3 = 0x00020000: raw 32-bit
    // data format of gfx9; the range check (num_records) is off the table: offsets are built from the handle's own tables.
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void*)p.adj_idx, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)p.adj_k, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)p.adj_ir, 0, 0x7fffffff, 0x00020000);
    const v2u i2 = __builtin_amdgcn_raw_buffer_load_b64(ri, (unsigned)lane * 8u, row * 8, 0);
    const v4f k4 = __builtin_amdgcn_raw_buffer_load_b128(rk, (unsigned)lane * 16u, row * 16, 0);
    const v4f a4 = __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)lane * 16u, row * 16, 0);
    AdjGroup g;
    g.idx = make_uint2(i2.x, i2.y); g.k = make_float4(k4.x, k4.y, k4.z, k4.w); g.a = make_float4(a4.x, a4.y, a4.z, a4.w);
    return g;
}

template <int RCAP>
__device__ __forceinline__ void spring_group(const PhysDev& p, const AdjGroup& g, const __attribute__((address_space(3))) char* win, f3 xi,
                                             f3 vi, v2f& fxy, float& fz)
{
    typedef __attribute__((address_space(3))) const v2f lds_f2;
    const unsigned off[GROUP] = {g.idx.x & 0xffffu, g.idx.x >> 16, g.idx.y & 0xffffu, g.idx.y >> 16};
    const float k[GROUP] = {g.k.x, g.k.y, g.k.z, g.k.w};
    const float a[GROUP] = {g.a.x, g.a.y, g.a.z, g.a.w};
#pragma unroll
    for (int u = 0; u < GROUP; ++u) {
        const v2f xy = *(lds_f2*)(win + off[u]);
        const v2f zz = *(lds_f2*)(win + off[u] + PLANE1<RCAP>());
        const v2f vxy = *(lds_f2*)(win + off[u] + PLANE2<RCAP>());
        spring_term(xy, zz.x, vxy, zz.y, xi, vi, k[u], a[u], p.dashpot, fxy, fz);
    }
}

template <int RCAP>
__device__ __forceinline__ f3 spring_force_lds(const PhysDev& p, const StateC xv,
                                               const __attribute__((address_space(3))) char* win, size_t env_base, int sl, int ln, f3 xi,
                                               f3 vi, int srow, int ngroups, AdjGroup g0, int pf_step = -1)
{
    v2f fxy = {0.f, 0.f};
    float fz = 0.f;
    AdjGroup a = g0, b = g0;
    int g = 0;
    for (; g + 2 <= ngroups; g += 2) { // ngroups is wave-uniform (one slice per wavefront): scalar branches
        b = adj_load(p, srow + (g + 1) * SLICE, ln);
        spring_group<RCAP>(p, a, win, xi, vi, fxy, fz);
        a = adj_load(p, srow + min(g + 2, ngroups - 1) * SLICE, ln); // unconditional (the last trip re-reads a group it will not use): no
                                                                  // branch inside the loop body (guarded: 24.0 vs 22.8 us with one chain)
        spring_group<RCAP>(p, b, win, xi, vi, fxy, fz);
    }
    if (g < ngroups) spring_group<RCAP>(p, a, win, xi, vi, fxy, fz);
    // neighbours outside the LDS window: slot-major coalesced adjacency, records gathered from global memory
    // (only when a block's halo exceeds the window capacity; never for the benchmark objects)
    const int4* __restrict__ ra = p.radj + p.rslice_off[sl] + ln;
    const int rdeg = p.rslice_deg[sl];
    for (int n = 0; n < rdeg; ++n) {
        const int4 en = ra[n * SLICE];
        const size_t gi = env_base + (size_t)en.x;
        v2f jxy = xv.p[st_at(xv.n, gi, 0)], jz = xv.p[st_at(xv.n, gi, 1)], jv = xv.p[st_at(xv.n, gi, 2)];
        if (pf_step >= 0 && pf_pending(jxy)) pf_wait(p, gi, pf_step, -1, jxy, jz, jv); // (k_substep_pf: a neighbour the previous substep left to the finishers)
        spring_term(jxy, jz.x, jv, jz.y, xi, vi, __int_as_float(en.y), __int_as_float(en.z), p.dashpot, fxy, fz);
    }
    return {fxy.x, fxy.y, fz};
}

#pragma clang fp contract(off)

// Exact early-out of the mesh query.  The response only fires when signed distance < margin (5 mm for gripper meshes, 1 mm
// otherwise).  A point outside the AABB of a CLOSED mesh is outside the mesh (winding number 0 < 0.6, sign +1) and its
// distance to the mesh is at least its distance to the AABB: if that is >= the margin for every mesh, nothing can happen and
// the query is skipped.  Meshes that are not closed manifolds (checked at construction) only get the query's own 2 cm range
// as the bound.  `pad` widens the test (particles whose velocity is not final yet); `near` = within NEAR_PAD of a margin.
__device__ __forceinline__ float mesh_margin(const PhysDev& p, int m)
{
    return (p.mesh_kind[m] & 2) ? MESH_MAX_DIST : ((m < p.n_dyn_mesh && !p.use_pusher) ? 0.005f : 0.001f);
}
// `staged` (the resident stepper): the substep's boxes [n_mesh][6], already in LDS — the two dependent loads below are then off the
// critical path of a substep in which something is in reach
__device__ __forceinline__ bool mesh_need(const PhysDev& p, int e, int step, f3 next_x, float pad, bool& near, const float* staged = nullptr)
{
    bool need = false;
    near = false;
    for (int m = 0; m < p.n_mesh; ++m) {
        const float* bb = staged ? staged + 6 * m
                        : m < p.n_dyn_mesh ? p.aabb_dyn + (((size_t)e * p.n_sub + step) * p.n_dyn_mesh + m) * 6
                                           : p.aabb_static + ((size_t)e * (p.n_mesh - p.n_dyn_mesh) + (m - p.n_dyn_mesh)) * 6;
        const float mg = mesh_margin(p, m) + pad;
        const float d2 = box_dist2(next_x, bb);
        need = need || d2 < mg * mg * 1.0001f;
        near = near || d2 < (mg + NEAR_PAD) * (mg + NEAR_PAD);
    }
    return need;
}

// The resident stepper's side channel into finish_wave: the finished state coming back, and boxes for the mesh early-out — per mesh
// (meshes beyond RES_MAX_MESH share the last box) the union of its world boxes over all substeps of the launch, and the union of
// those.  A particle farther from a union than margin + RES_RANGE_PAD is not within reach of that mesh at any substep
// (every box lies inside its union, so its distance is at least the union's): when no lane of the wavefront is inside that range of
// the total union, and then of any mesh's, the per-substep tests — and the loads of the substep's boxes, two dependent round trips
// in the critical path of every substep — are skipped; otherwise the exact tests run as in k_substep.
constexpr int RES_MAX_MESH = 4;
// Reach of the early-out beyond a mesh's margin.  Small batches pick their flavour from "a query was NEEDED" (a particle inside a margin),
// not from "something is NEAR" (margin + 3 cm), so a resident launch only has to find the particles inside a margin exactly; with the
// 3 cm of the large-batch rule every block under a hovering gripper ran the exact per-substep tests (two dependent loads in the
// finishing code) and paced the whole chain: 2.7 instead of 2.3 us per substep.  The "near" flag of a resident launch is therefore only
// raised from within this reach.
constexpr float RES_RANGE_PAD = 0.002f;
struct ResidentIO {
    bool srv_on, srv_need; // in: needy particles go to a query server instead of being queried in place; out: this lane's particle does
    const float* step_boxes; // in (LDS) or null: this substep's mesh boxes [n_mesh][6], staged by the launch at the top of the substep
    f3 x, v;               // out: the particle's new state
    const float* boxes;    // in (LDS, wave-uniform values — 35 registers per lane if they lived there): [0..5] union of everything, [6] (largest
                           // margin + RES_RANGE_PAD)^2 widened by 1e-4 relative; then per mesh slot m at 8 + 8 m: [0..5] its union over the substeps,
                           // [6] (its margin + RES_RANGE_PAD)^2, widened (0 for unused slots: never in range)
};
__device__ __forceinline__ bool resident_in_range(const ResidentIO& io, f3 next_x, bool fin)
{
    if (__builtin_amdgcn_ballot_w64(fin && box_dist2(next_x, io.boxes) < io.boxes[6]) == 0ull) return false;
    bool any = false;
#pragma unroll
    for (int m = 0; m < RES_MAX_MESH; ++m) any = any || box_dist2(next_x, io.boxes + 8 + 8 * m) < io.boxes[8 + 8 * m + 6];
    return __builtin_amdgcn_ballot_w64(fin && any) != 0ull;
}

// ---- everything after the velocity update: mesh collision, ground, store ------------------------------------
// Called by EVERY lane of a workgroup at the same point (the mesh queries of MESH 2 / 3 are workgroup-cooperative, with a
// barrier inside); `fin` says whether this lane has a particle to finish, `store` whether it is the one that writes it back.
// Shared by the fused substep and the finishing kernels.
// Large-mesh scenes: hand a particle to the substep's finishing launch through its environment's list (a particle is listed at most
// once per substep: N slots cannot overflow).  ncand > 0 = tagged: the particle also has self-collision candidates, `v` is its published
// pre-impulse velocity and k_contact_finish applies the impulses first.
__device__ __forceinline__ bool mesh_rec_push(const PhysDev& p, int e, int step, int i, int ncand, f3 x0, f3 v)
{
    const int slot = atomicAdd(p.rec_cnt + (size_t)e * p.n_sub + step, 1);
    if (slot >= p.N) return false;
    int4* r = p.mesh_rec + 2 * (par_off(p, step) + (size_t)e * p.N + slot);
    const int hint = p.mq_hint ? p.mq_hint[(size_t)e * p.N + i] : -1; // the cluster of its closest face one substep ago rides in the record (bits 19..30)
    r[0] = make_int4(ncand | ((hint + 1) << 19), ncand > 0 ? (i | (int)0x80000000) : i, __float_as_int(x0.x), __float_as_int(x0.y));
    r[1] = make_int4(__float_as_int(x0.z), __float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z));
    return true;
}

// MESH: 0 no meshes, 1 small meshes only (per-lane queries), 2 a large mesh is present (never queried in the fused kernel)
// MAIN + p.mesh_defer (the fused kernel and k_self_finish): a particle that needs a mesh query is not queried here.  A query
// is thousands of instructions (closest point over the near meshes' faces + the exact winding number over all faces, twice
// for finger contacts) or, for a large mesh, a walk through its box hierarchy — and the particles that need one sit next to
// each other, so one wavefront would run dozens back to back while the rest of the chip waits (measured: 36 touching
// particles stretched a 9 us substep to 195 us).  Instead it stores its velocity, appends itself to the substep's list and
// is finished by k_contact_finish, one WORKGROUP per particle, all of them in flight at once.  Without p.mesh_defer (the
// flavour captured while nothing is near a mesh) the rare needy particle is queried in place.
// MESH: 3 = small scene with the triangles in registers (k_contact_finish<3>; two wavefronts per particle)
// NEED: 0 = decide by the exact early-out; 1 = query without testing (the fused kernel already found the particle in reach of a
// mesh: saves the finishing kernel one dependent round trip for the boxes); 2 = never query (the fused kernel's WIDENED test
// found nothing in reach: mesh_collision then only advances the position, :321 / :420)
// KEEP (the resident stepper): every lane with `fin` also returns its new state in keep->x / keep->v and only stores it when
// xv_out.p is set (the launch's last substep); the mesh boxes of the early-out come from *keep.
// PFOUT (the finishers at the head of the next launch, p.pf): the finished state goes to the particle's line of p.pf_res, tagged step + 1,
// instead of the state array.  Returns whether THIS call finished (and stored / kept) the lane's particle.
template <int MESH, bool MAIN = false, int NEED = 0, bool KEEP = false, bool QUAD = false, bool PFOUT = false>
__device__ __forceinline__ bool finish_wave(const PhysDev& p, int e, int i, size_t eb, int step, int write_forces, f3 x0, f3 v, bool fin,
                                            const StateM xv_out, const TriRegs* tr, QShare* qs, int* qpar, BlkAux* xf0, bool store, ResidentIO* keep R2S_QP_PARAM)
{
    f3 x = x0;
    // mesh_collision, :295-421 — advances x by v*dt for EVERY particle (:321, :420)
    if (MESH) {
        f3 vin = v;
        f3 next_x = x0 + vin * p.dt;
        f3 next_v = vin;
        bool in_range = true;
        if (KEEP && keep->boxes) in_range = resident_in_range(*keep, next_x, fin); // wave-uniform (a single-substep launch carries no unions)
      if (in_range) {
        bool need = false, near = false;
        if (NEED == 1) need = fin;
        else if (NEED == 0 && fin) need = mesh_need(p, e, step, next_x, 0.f, near, KEEP ? keep->step_boxes : nullptr);
        if (MAIN) { // count the particles near a mesh (the host picks the next step's graph flavour from the total) and, in
                    // deferring mode, hand the ones that need a query to k_contact_finish
            // only "anything near?" is consumed (the host picks the next step's flavour from it): one plain store per wavefront
            // instead of a per-lane atomicAdd on a single word (thousands per substep while an object sits next to a mesh:
            // 1.5 - 2 us per substep in the pusher and grasp scenes)
            const unsigned long long nm = __builtin_amdgcn_ballot_w64(near);
            if (nm && (int)(threadIdx.x & 63) == __builtin_ctzll(nm)) p.mesh_cnt[p.n_sub] = 1;
            // "a query was needed" (a particle inside a margin's reach), sticky until the host has read it: what small batches pick the
            // next step's flavour from (r2s_phys_step) — their free flavour is the resident launch, worth keeping while the gripper merely hovers
            const unsigned long long qm = __builtin_amdgcn_ballot_w64(need);
            if (qm && (int)(threadIdx.x & 63) == __builtin_ctzll(qm)) p.fault[1] = 1;
            if (MESH == 2) { // large scenes always defer (the fused kernel carries no query code), through the per-environment records
                if (need && mesh_rec_push(p, e, step, i, 0, x0, v)) { fin = false; need = false; }
            } else if (need && p.mesh_defer) {
                const int slot = atomicAdd(p.mesh_cnt + step, 1);
                if (slot < p.mesh_cap) {
                    const size_t po = par_off(p, step);
                    p.vdef[po + eb + i] = make_float4(v.x, v.y, v.z, 0.f);
                    p.xbc[po + eb + i] = make_float4(x0.x, x0.y, x0.z, 0.f);
                    p.mesh_list[po + slot] = make_int2(e, i);
                    fin = false; // finished by k_contact_finish
                    need = false;
                }
                // list full (never with the sizing below): fall through to the in-place query
            } else if (KEEP && need && keep->srv_on) { // resident launch with query servers: not finished here (see k_steps_resident)
                keep->srv_need = true;
                fin = false;
                need = false;
            }
        }
        // large scenes (MESH 2): never queried in the fused kernel; in k_contact_finish by the whole workgroup for the particle of
        // lane 0 (every wavefront of the workgroup runs this function on the same particle; `store` marks the one that writes)
        constexpr bool IN_PLACE = !(MAIN && MESH == 2) && NEED != 2;
        MeshHit q = {false, 0.f, 0, mk(0.f, 0.f, 0.f), 0, 0};
        if (IN_PLACE)
            q = MESH == 3 ? mesh_query_regs<QUAD>(*tr, mk(bcast(next_x.x, 0), bcast(next_x.y, 0), bcast(next_x.z, 0)), bcasti((int)need, 0) != 0, *qs, *qpar)
              : MESH == 2 ? mesh_query_block(p, step, mk(bcast(next_x.x, 0), bcast(next_x.y, 0), bcast(next_x.z, 0)), e,
                                             bcasti((int)need, 0) != 0, xf0->hint, *qs, *qpar, *xf0 R2S_QP_ARG) // workgroup-cooperative, call site 1
                          : mesh_query_lane(p, e, step, next_x, need);
        R2S_QSTAMP(); // first query back
        // per-lane response; lanes that must re-query (gripper branch, :394-408) park their state and meet again below
        bool requery = false;
        f3 normal = mk(0.f, 0.f, 0.f), v_normal = mk(0.f, 0.f, 0.f), v_normal_new = mk(0.f, 0.f, 0.f);
        float margin = 0.f;
        bool hit = false;
        if (q.result) {
            int is_gripper;
            const int mm = MESH >= 2 ? q.mm : p.mesh_map[q.face];
            if (!p.use_pusher) is_gripper = mm == 0 ? 1 : (mm == 1 ? 2 : 0);
            else is_gripper = mm >= 0 ? 1 : 0;
            f3 delta = next_x - q.pt;
            float dist = len(delta) * q.sign;
            margin = (is_gripper >= 1 && !p.use_pusher) ? 0.005f : 0.001f;
            float err = dist - margin;
            if (err < 0.f) {
                hit = true;
                normal = normalize0(delta) * q.sign;
                f3 rdv = mk(0.f, 0.f, 0.f);
                float ce, cf;
                if (is_gripper >= 1) {
                    const f3 ctr = MESH == 3 ? tr->ctr : ld3(p.interp_center, (size_t)e * p.n_sub + step);
                    const f3 om = MESH == 3 ? tr->om : ld3(p.dyn_omega, e);
                    const f3 dv = MESH == 3 ? (is_gripper == 1 ? tr->dv0 : tr->dv1) : ld3(p.dyn_vel, (size_t)e * 2 + (is_gripper == 1 ? 0 : 1));
                    rdv = dv + cross(om, x0 - ctr);
                    vin = vin - rdv;
                    ce = p.cee; cf = p.cef;
                } else {
                    ce = p.ce; cf = p.cf;
                }
                v_normal = normal * dot(vin, normal);
                const f3 v_tao = vin - v_normal;
                const float vnl = len(v_normal);
                const float vtl = fmaxf(len(v_tao), 1e-6f);
                v_normal_new = v_normal * (-ce);
                const float a = fmaxf(0.f, 1.f - cf * (1.f + ce) * vnl / vtl);
                next_v = v_normal_new + v_tao * a;
                if (is_gripper >= 1) {
                    next_v = next_v + rdv;
                    next_x = x0 + next_v * p.dt;
                    requery = true; // the reference rebinds `query` (:397)
                } else {
                    next_x = next_x - normal * err;
                }
            }
        }
        MeshHit q2 = {false, 0.f, 0, mk(0.f, 0.f, 0.f), 0, 0};
        if (IN_PLACE)
            q2 = MESH == 3 ? mesh_query_regs<QUAD>(*tr, mk(bcast(next_x.x, 0), bcast(next_x.y, 0), bcast(next_x.z, 0)), bcasti((int)requery, 0) != 0, *qs, *qpar)
               : MESH == 2 ? mesh_query_block(p, step, mk(bcast(next_x.x, 0), bcast(next_x.y, 0), bcast(next_x.z, 0)), e,
                                              bcasti((int)requery, 0) != 0, bcasti(q.hint, 0), *qs, *qpar, *xf0 R2S_QP_ARG) // call site 2
                           : mesh_query_lane(p, e, step, next_x, requery);
        R2S_QSTAMP(); // response + second query back
        if (requery) {
            if (q2.result) {
                const f3 delta = next_x - q2.pt;
                const float dist = len(delta) * q2.sign;
                const float err = dist - margin;
                if (err < 0.f) {
                    normal = normalize0(delta) * q2.sign;
                    next_x = next_x - normal * err;
                }
            }
            q = q2; // face of the LAST query (0 if the re-query missed)
        }
        // large-mesh scenes: where this particle's next substep should look first (the finishing workgroup's storing lane)
        if (MESH == 2 && IN_PLACE && !MAIN && fin && store && p.mq_hint) p.mq_hint[eb + i] = q.hint;
        if (hit && write_forces && store) {
            const f3 fo = (v_normal_new - v_normal) / p.dt;
            float* cf3 = p.coll_forces + ((size_t)e * p.nF + (MESH >= 2 ? q.fm : p.face_map[q.face])) * 3;
            atomicAdd(cf3, fo.x);
            atomicAdd(cf3 + 1, fo.y);
            atomicAdd(cf3 + 2, fo.z);
            atomicAdd(p.hit_cnt + e, 1);
        }
      }
        x = next_x;
        v = next_v;
    }

    // integrate_ground_collision, :424-474
    if (fin && (store || KEEP)) {
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
        if (PFOUT) { if (store) pf_store(p, eb + (size_t)i, xn, v1, (unsigned)step + 1u); }
        else if (store && (!KEEP || xv_out.p != nullptr)) st_store(xv_out, eb + i, xn, v1);
        if (KEEP) { keep->x = xn; keep->v = v1; }
        return true;
    }
    return false;
}

// ---- the fused substep ------------------------------------------------------------------------------
// One workgroup (B threads) = B consecutive (Morton-ordered) particles of one environment, one particle per lane, one
// 64-particle ELL slice per wavefront.  Linear workgroup id L: XCD = L % 8 (observed dispatch order; a speed assumption
// only); XCD c owns the contiguous range [c*cb, (c+1)*cb) of (block, env) work items, env fastest, so its slice of the
// adjacency and its particles stay in its 4 MB L2.
// Layouts <B, RCAP> (threads, LDS window records): <256,1024> 24 KB (6 workgroups per CU) for large batches, <128,768>
// where more, smaller workgroups fill the chip better.  (A <512,1536> layout held to 64 VGPRs / 80 SGPRs keeps all 960
// workgroups of the 32-env benchmark resident at once; measured 26.0 vs 25.5 us — the kernel is bound by per-CU VALU
// throughput in the gather, not by residency; lowering residency with LDS padding is slower: 26.3 / 26.5 / 27.3 / 28.5 us
// for 6 / 5 / 4 / 3 workgroups per CU.)
#ifdef R2S_PHASE_PROBE
__device__ long long g_phase_probe[8192 * 4]; // wall-clock (100 MHz) stamps per workgroup: entry, staged, springs done, end
#define R2S_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_phase_probe[blockIdx.x * 4 + (k)] = (long long)wall_clock64(); } while (0)
extern "C" int r2s_phys_debug_phase_probe(long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_probe), sizeof(long long) * (size_t)n * 4);
}
#else
#define R2S_STAMP(k) do { } while (0)
#endif

// PF: the launch is a k_substep_pf — `bid` = the workgroup's number among the fused blocks (behind the finishers), records of particles the
// previous substep left unfinished are PF_SENT and come from the finishers' result lines, particles this substep leaves unfinished get PF_SENT
template <int B, int RCAP, bool SELF, int MESH, bool PF = false>
__device__ __forceinline__ void substep_body(const PhysDev& p, const StateC xv_in, const StateM xv_out, int step,
                                             int write_forces, int bid)
{
    static_assert(B % SLICE == 0 && RCAP >= B && RCAP * 8 <= 65536, "window offsets are u16 bytes");
    __shared__ __attribute__((aligned(16))) v2f win_s[3 * (RCAP + 1)]; // planes xy | (z, vz) | vxy, 24 B per record (+ 1 pad each)
    const int xcd = bid & 7, q = bid >> 3;
    const int item = xcd * p.cb + q;
    if (q >= p.cb || item >= p.nb * p.ne) return; // whole workgroup
    // (round 5, measured and removed: XCD x owning the blocks b = x mod 8 instead of a contiguous range — so that the blocks of a contact
    // region, neighbours in Morton order, and with them the blocks that wait for a finisher in k_substep_pf, spread over all eight XCDs:
    // 22.8 vs 22.2 us per contact substep of the headline, 17.8 vs 17.0 free: the halo locality of contiguous ranges is worth more)
    const int b = item / p.ne, e = p.e0 + (item - b * p.ne);
    R2S_STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int i = b * B + tid;
    const bool valid = i < p.N;
    const size_t eb = (size_t)e * p.N;
    const int ic = min(i, p.N - 1);
    // first adjacency group of this wavefront's slice: in flight while the LDS window is staged
    const int sl = __builtin_amdgcn_readfirstlane(ic / SLICE);
    const int srow = __builtin_amdgcn_readfirstlane(p.slice_off[sl] / GROUP); // wave-uniform
    const int ngroups = __builtin_amdgcn_readfirstlane(p.slice_deg[sl] / GROUP);
    AdjGroup g0;
    g0.idx = make_uint2(0u, 0u); g0.k = make_float4(0.f, 0.f, 0.f, 0.f); g0.a = g0.k;
    if (ngroups > 0) g0 = adj_load(p, srow, lane);
    // stage the block's own records (record r < B is particle b*B + r) and its halo (record B + k is halo particle k).
    // All loads of a thread are issued before its first LDS write: two dependent round trips (halo id, then state)
    // per workgroup instead of two per staging round.
    constexpr int K = (RCAP + B - 1) / B;
    // R2S_STAGE_BATCH: staging rounds whose loads are in flight together.  All K of them (round 2) keep 28 staging registers live
    // next to the prefetched adjacency group; two at a time leave the kernel at ~50 VGPRs outside the mesh code at the price of a
    // second pair of dependent round trips per workgroup, which the other five resident workgroups hide.
#ifndef R2S_STAGE_BATCH
#define R2S_STAGE_BATCH 2
#endif
#ifndef R2S_STAGE64
#define R2S_STAGE64 2
#endif
    // (one-wavefront workgroups of the small-batch layout: all K rounds in flight at once was measured and changes nothing, 9.8 vs 9.5 us —
    // that kernel is bound by the lone wavefront walking all of a particle's slots)
    constexpr int KB = B == 64 ? (R2S_STAGE64 < K ? R2S_STAGE64 : K) : ((R2S_STAGE_BATCH > 0 && R2S_STAGE_BATCH < K) ? R2S_STAGE_BATCH : K);
    const int h0 = p.halo_off[b], per_env = B + (p.halo_off[b + 1] - h0);
    v2f own_a = {0.f, 0.f}, own_b = own_a, own_c = own_a; // this lane's own record (round 0)
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += KB) {
        int part[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int r = tid + (k0 + k) * B;
            part[k] = (k0 + k >= K) ? p.N : (r < B ? i : (r < per_env ? p.halo_ids[h0 + r - B] : p.N));
        }
        v2f qa[KB], qb[KB], qc[KB]; // xy | (z, vz) | vxy: the state planes are the window's planes
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const size_t g = eb + (size_t)min(part[k], p.N - 1);
            if (part[k] < p.N) { qa[k] = xv_in.p[st_at(xv_in.n, g, 0)]; qb[k] = xv_in.p[st_at(xv_in.n, g, 1)]; qc[k] = xv_in.p[st_at(xv_in.n, g, 2)]; }
            else { qa[k] = (v2f){0.f, 0.f}; qb[k] = qa[k]; qc[k] = qa[k]; }
        }
        if (PF) { // records the previous substep left to the finishers at the head of THIS launch: wait for theirs (a few blocks per environment)
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (part[k] < p.N && pf_pending(qa[k])) pf_wait(p, eb + (size_t)part[k], step, item, qa[k], qb[k], qc[k]);
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const int r = tid + (k0 + k) * B;
            if (r < RCAP && part[k] < p.N) {
                win_s[r] = qa[k];
                win_s[RCAP + 1 + r] = qb[k];
                win_s[2 * (RCAP + 1) + r] = qc[k];
            }
        }
        if (k0 == 0) { own_a = qa[0]; own_b = qb[0]; own_c = qc[0]; }
    }
    __syncthreads();
    R2S_STAMP(1);
    // no early exit: lanes without a particle stay in the wavefront (the mesh queries at the end are wave-cooperative)
    // and simply compute on clamped indices without storing anything
    const f3 x0 = mk(own_a.x, own_a.y, own_b.x), v0 = mk(own_c.x, own_c.y, own_b.y); // round 0 staged this lane's own record
    const float m1 = p.masses[ic];

    // eval_springs + update_vel_from_force
    const __attribute__((address_space(3))) char* win = (const __attribute__((address_space(3))) char*)win_s;
    f3 v = vel_update(p, v0, spring_force_lds<RCAP>(p, xv_in, win, eb, sl, lane, x0, v0, srow, ngroups, g0, PF ? step : -1), m1);
#ifdef R2S_PHASE_PROBE
    if (v.x == 1.2345e33f) return; // keep the stamp after the gather
#endif
    R2S_STAMP(2);

    // Self collision (object_collision, :230-268) needs the partners' post-force velocities.  Particles that have
    // contact candidates (rare; the list is rebuilt once per env step) only publish their own v_before_collision
    // here and are finished by k_self_finish, which reads the partners' published values; everyone else is done.
    bool fin = valid;
    if (SELF) {
        const int ncand = valid ? p.coll_num[eb + i] : 0;
        if (ncand > 0) {
            const size_t po = par_off(p, step);
            p.vbc[po + eb + i] = make_float4(v.x, v.y, v.z, 0.f);
            p.xbc[po + eb + i] = make_float4(x0.x, x0.y, x0.z, 0.f);
            fin = false; // finished by k_self_finish / k_contact_finish
            if (MESH != 0 && (p.mesh_defer || MESH == 2)) {
                // Will it also need a mesh query?  Its velocity is not final (the impulses come later), so the test is widened
                // by 2 mm (= 40 m/s of velocity change in one substep); over-inclusion is harmless, the query itself is exact.
                // Such a particle goes to the mesh list TAGGED: k_contact_finish applies its impulses and queries in one go.
                bool near;
                if (mesh_need(p, e, step, x0 + v * p.dt, 0.002f, near)) {
                    p.fault[1] = 1;
                    if (MESH == 2) {
                        if (mesh_rec_push(p, e, step, i, ncand, x0, v)) p.cand_mark[po + eb + i] = step + 1;
                    } else {
                        const int slot = atomicAdd(p.mesh_cnt + step, 1);
                        if (slot < p.mesh_cap) {
                            p.mesh_list[po + slot] = make_int2(e | (ncand << 12), i | (int)0x80000000);
                            p.cand_mark[po + eb + i] = step + 1;
                        }
                    }
                }
                const unsigned long long nm = __builtin_amdgcn_ballot_w64(near);
                if (nm && lane == __builtin_ctzll(nm)) p.mesh_cnt[p.n_sub] = 1;
            }
        }
    }
    R2S_QP_DECL(-1);
    const bool done = finish_wave<MESH, MESH != 0>(p, e, i, eb, step, write_forces, x0, v, fin, xv_out, nullptr, nullptr, nullptr, nullptr, true, nullptr R2S_QP_ARG);
    if (PF && valid && !done) pf_mark(xv_out, eb + i); // left to the finishers at the head of the next launch
    R2S_STAMP(3);
}

// (A <256,896> layout — 21.5 KB of LDS, 7 workgroups per CU, the fused kernel held to 72 VGPRs, so that only 96 instead of 352
// of the benchmark's 1888 work items are left for a second round — was measured in round 2: 23.3 vs 22.1 us per substep with
// two chains, 24.4 vs 24.8 with one.  More residency does not pay; the layouts stay <256,1024> and <128,768>.)
// <256,1024>: 24.6 KB of LDS allow six workgroups per CU; the register allocator is told so (84 -> 79 VGPRs, no spills: five ->
// six wavefronts per SIMD).  On its own that is worth nothing measurable (19.8 vs 19.9 us), with the staging batch of two and the
// shorter spring term 18.8 vs 19.5.
template <int B, int RCAP, bool SELF, int MESH>
__global__ void __launch_bounds__(B, (B == 256 ? 6 : 1)) k_substep(const PhysDev p, const StateC xv_in, const StateM xv_out, int step,
                                               int write_forces)
{
    substep_body<B, RCAP, SELF, MESH>(p, xv_in, xv_out, step, write_forces, (int)blockIdx.x);
}
// ---- the resident stepper: every substep of an env step in ONE launch (small batches) ---------------------------
// A batch whose (block, env) work items are all on the chip at once — one environment of the reference's own evaluation loop
// (eval_policy.py drives ONE simulator), up to a few — is bound by latency, not by throughput: k_substep for the 8 k-particle
// rope is 7.0 us per launch in a 7.6 us launch period for 0.15 us of arithmetic (profiles/r3_bench_kernel_stats_rope_1env.md):
// two dependent staging round trips, one wavefront walking a particle's ~35 slots alone, the finishing code, the kernel
// boundary.  This kernel keeps the env step on the chip instead:
//   * one workgroup = ONE 64-particle ELL slice x EIGHT wavefronts (two per SIMD: one's LDS / dependent-issue latency is the other's issue
//     slot); wavefront w evaluates groups w, w+8, ... of every particle of the slice (its <= RES_NG interior and RES_NG halo adjacency
//     groups live in registers for the whole launch: no adjacency stream at all), the eight partial forces meet in LDS and are added in
//     a fixed tree; wavefronts 0..2 then finish the particle redundantly (same inputs, same instructions, same result), wavefront 0
//     owns the side effects, wavefront q publishes plane q;
//   * own particles stay in the block's LDS window from substep to substep; only the HALO crosses workgroups: after a substep the three
//     planes of the block's 64 records go out as 16-byte {value, tag, value, tag} write-through stores (sc1) into a double-buffered
//     exchange array, and the neighbours poll exactly the records of their halo list until both tags read the substep's number — the
//     data is the flag (cdna_hip_programming.md, Guideline 16 R2: no fence, no flag, no grid barrier: a workgroup only ever waits for
//     the blocks it shares springs with).  Two buffers are enough: a block publishes version v+1 (overwriting v-1) only after it has
//     read version v of every neighbour, and a neighbour publishes v only after it has read v-1 of this block (halo lists are
//     symmetric: they follow the springs);
//   * the 64-particle layout lists a particle's neighbours inside the block first: those groups are evaluated while the neighbours'
//     records are still on their way, the poll follows, then the halo groups;
//   * tags are substep numbers within the launch (1 ..), the exchange array is zeroed by a kernel node ahead of every launch;
//     polls are bounded (RES_SPIN_LIMIT passes, then the sticky fault word and out: never a hang); launches of one device are
//     serialised across handles (resident_enter): a launch needs all its workgroups on the chip at once, one per CU.
// Used for the flavour "no particle has self-collision candidates, no mesh query was needed in the last step" (in-place queries for
// the first particle that enters a margin, like k_substep without p.mesh_defer); every other flavour runs this kernel with ONE substep
// per launch (below).  Results differ from k_substep's in the last bits (eight partial sums, reciprocal mass).
constexpr int RES_THREADS = 512;                // eight wavefronts: two per SIMD, so that one's LDS and dependent-issue latency is the other's issue slot
constexpr int RES_NG = 2;                       // interior and halo adjacency groups a wavefront keeps in registers (each: every 8th group of the slice)
constexpr unsigned RES_SPIN_LIMIT = 1u << 21;   // poll passes before a workgroup gives up (each >= one L2 round trip: seconds)
constexpr int RES_AUX_SC1 = 16;                 // buffer-instruction cache policy: sc1 = agent scope (write-through store, L1-bypassing load)
#ifndef R2S_RES_AUXLD
#define R2S_RES_AUXLD (16 | (int)0x80000000)
#endif
#ifndef R2S_RES_PRE
#define R2S_RES_PRE 2
#endif
#ifndef R2S_RES_DELAY
#define R2S_RES_DELAY 0
#endif
constexpr int RES_AUX_LOAD = R2S_RES_AUXLD;     // poll loads: sc1 + the compiler-side volatile bit (= sc0 sc1 in the instruction)
constexpr int RES_PRE = R2S_RES_PRE;            // interior groups evaluated BEFORE the first poll pass is issued

// NG groups back to back, no branch in between: the LDS reads of a later group are scheduled under the arithmetic of an earlier one
// (two wavefronts share a SIMD here, six in the fused substep: most of the latency the instruction stream exposes is paid)
struct GroupRecs { v2f xy[GROUP], zz[GROUP], vv[GROUP]; };
template <int RCAP>
__device__ __forceinline__ void group_read(const AdjGroup& g, const __attribute__((address_space(3))) char* win, GroupRecs& r)
{
    typedef __attribute__((address_space(3))) const v2f lds_f2;
    const unsigned off[GROUP] = {g.idx.x & 0xffffu, g.idx.x >> 16, g.idx.y & 0xffffu, g.idx.y >> 16};
#pragma unroll
    for (int u = 0; u < GROUP; ++u) {
        r.xy[u] = *(lds_f2*)(win + off[u]);
        r.zz[u] = *(lds_f2*)(win + off[u] + PLANE1<RCAP>());
        r.vv[u] = *(lds_f2*)(win + off[u] + PLANE2<RCAP>());
    }
}
// (Evaluating the four slots of a group stage by stage behind scheduling barriers — four independent instructions between an
// instruction and its consumer instead of one slot's dependent chain after the other — was measured and changes nothing: 2.37 vs 2.36 us
// per substep.  With two wavefronts per SIMD the chain latency is covered; what a substep waits for is the hand-off.)
__device__ __forceinline__ void group_eval(const PhysDev& p, const AdjGroup& g, const GroupRecs& r, f3 xi, f3 vi, v2f& fxy, float& fz)
{
    const float k[GROUP] = {g.k.x, g.k.y, g.k.z, g.k.w};
    const float a[GROUP] = {g.a.x, g.a.y, g.a.z, g.a.w};
#pragma unroll
    for (int u = 0; u < GROUP; ++u) spring_term(r.xy[u], r.zz[u].x, r.vv[u], r.zz[u].y, xi, vi, k[u], a[u], p.dashpot, fxy, fz);
}
// The compiler's own schedule of spring_group waits for each slot's three reads and then runs that slot's dependent chain (fine with
// six wavefronts per SIMD to switch to, 180 cycles per slot for a lone one); here the records of group j + 1 are read before group j is
// evaluated, and scheduling barriers keep the two from being sunk back together.
template <int RCAP, int NG>
__device__ __forceinline__ void spring_groups(const PhysDev& p, const AdjGroup* g, const __attribute__((address_space(3))) char* win, f3 xi, f3 vi,
                                              v2f& fxy, float& fz)
{
    GroupRecs r[2];
    group_read<RCAP>(g[0], win, r[0]);
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        if (j + 1 < NG) group_read<RCAP>(g[j + 1], win, r[(j + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        group_eval(p, g[j], r[j & 1], xi, vi, fxy, fz);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int RCAP>
__device__ __forceinline__ void spring_groups_n(const PhysDev& p, int n, const AdjGroup* g, const __attribute__((address_space(3))) char* win, f3 xi,
                                                f3 vi, v2f& fxy, float& fz)
{
    static_assert(RES_NG == 2, "one case per count");
    if (n == 2) spring_groups<RCAP, 2>(p, g, win, xi, vi, fxy, fz);
    else if (n == 1) spring_groups<RCAP, 1>(p, g, win, xi, vi, fxy, fz);
}

// ---- mesh-query servers of the resident launch (round 4) ---------------------------------------------------------------------------
// A resident launch must not answer mesh queries inside the blocks that own the particles: a query is thousands of instructions, and
// every block of the environment waits, hand-off by hand-off, for the slowest (measured in round 3 on the rope in a grasp: 54.8 us per
// substep with per-lane queries in the finishing wavefronts against 2.5 us in free motion; the per-substep kernels + finishing launch the
// step then fell back to: 11.8 us, 22.5 with the gripper closed on the rope).  A one-environment launch leaves about half of the chip idle (130 blocks of the 8 k-particle rope on
// 256 CUs), so the launch carries extra workgroups — SERVERS, four wavefront pairs (or, with workgroups to spare, two quads: QQUAD) each —
// and a particle that needs a query is handed to a unit of its own ("pair" below) through the same tagged write-through granules the blocks exchange their halos with:
//   claim    the first time a particle needs a query its block (wavefront 0 of the finishers) takes the next free pair (one atomic) and
//            writes {env * N + particle, first substep}; the pair serves that particle until the launch ends;
//   request  three 16-byte granules {x0, post-force v} tagged 2 (k + 1) + 1, with the claim;
//   result   the pair runs finish_wave<3> (k_contact_finish's small-scene code: the substep's triangles one per lane in two wavefronts,
//            loaded BEFORE the inputs arrive; mesh response, re-query, per-face forces on the last substep, ground) and returns the
//            particle's new state in three granules tagged k + 1 (two halves of a line, by version parity); wavefront 0 of the block
//            polls them, publishes all three planes of the particle to the exchange array, goes on;
//   owning   (default, p.srv_own) from the claim on the particle is the PAIR's: every later substep it gathers the particle's
//            neighbours of version k itself — lane n of the pair holds slot n of the adjacency row and polls that neighbour's exchange
//            granules (or, one hand-off earlier for a served neighbour, the result line of that neighbour's own pair) — sums the
//            springs, updates the velocity, and continues as above.  The block no longer stands between two substeps of a particle in
//            contact (result -> block -> forces -> request -> pair was three hand-offs per substep, 9.2 - 10.5 us for the rope in a
//            grasp; pair -> pair is one: 5.7 us, of which 3.6 are the unit's two queries and the response).  The pair may run one version ahead of its
//            block, never two: before it writes version k + 1 over version k - 1 it has seen the block's republished copy of k - 1;
//   per-substep requests (R2S_RES_SRV_OWN=0, the first protocol): the block sums the forces and sends a request per substep — or one
//            granule tagged 2 (k + 1) when the particle is out of every mesh's reach in substep k (the pair skips ahead);
//   end      a block that leaves the launch ends its pairs (tag SRV_END) and counts itself out; pairs nobody claimed leave when every
//            block has; an owning pair leaves after the launch's last substep.
// No more pairs than particles that ever need one are busy; a claim beyond the last pair is answered in place by the block's wavefront 0
// (with owning pairs: in every later substep too; the launch reports it — p.fault[2] — and the host leaves the resident launch until the
// contact is over).  Slots are handed out so that the first claims each get a server workgroup, i.e. a CU, of their own.  Every poll is bounded like the halo polls (sticky fault word, never a hang); the
// launch is resident as a whole (per XCD: the grid's round-robin share of every XCD <= its CUs), which the host guarantees when it sizes
// the grid.  The sums of an owning pair
// are fixed trees over its lanes — the same in every run, not the order of the block's eight wavefronts (results differ from the
// request protocol's in the last bits; both hold the oracle's 1e-5 and the per-substep kernels' 2e-6).
// the first fault of a launch wins and records where it happened (p.fault + 3 .. + 14 = the handle's words [4..15]): code, work item,
// substep, and six words of context — what the host's error message prints
// (resident_fault: defined with the head-of-launch finishing helpers above finish_wave)
constexpr unsigned SRV_END = 0x7ffffffeu;
// Every granule array below is laid out so that no 128-byte line has writers in two workgroups (= possibly two XCDs, whose L2s are not
// coherent): a claim per line, a line of requests (written by the particle's block) and a line of results (by its server pair) per
// particle.  With 96-byte records back to back — a neighbour's results and this particle's request in one line — a request or a claim was
// lost now and then (the rope in a grasp: one env step in ~1 000 timed out with the request's first granule visible and its second or
// third still carrying the previous tag, for as long as anybody looked): a write-through store of 16 bytes into a line of which the
// writer's L2 holds an older copy is not guaranteed to leave the other bytes of the line in memory alone.
constexpr int SRV_LINE = 128, SRV_REC = 2 * SRV_LINE, SRV_RES = SRV_LINE;
constexpr int SRV_MAX_SLOTS = 512, SRV_MIN_WG = 8;
constexpr int SRV_CTL_OFF = SRV_LINE * SRV_MAX_SLOTS, SRV_DBG_OFF = SRV_CTL_OFF + SRV_LINE, SRV_CLAIM_BYTES = SRV_DBG_OFF + SRV_LINE * SRV_MAX_SLOTS; // four pairs per server workgroup: at most 128 server workgroups; fewer than 8 are not worth the claims
__device__ __forceinline__ v4u srv_load(const __amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, RES_AUX_LOAD); }
// The server's polls: EVERY lane loads the same granule and the wavefront branches on it.  A wave64 memory instruction is served in
// several passes, and nothing promises that an L1-bypassing load of a granule that is being rewritten hands all lanes the same version
// (some lanes leaving the request poll with the tag of a "skip", the rest a moment later with the next "need", would run the finishing
// code and its two-wavefront barriers with partial lane masks).  The first lane's copy is the wavefront's.
__device__ __forceinline__ v4u srv_load_uniform(const __amdgpu_buffer_rsrc_t r, unsigned off)
{
    const v4u d = srv_load(r, off);
    const v4u u = {(unsigned)__builtin_amdgcn_readfirstlane((int)d.x), (unsigned)__builtin_amdgcn_readfirstlane((int)d.y),
                   (unsigned)__builtin_amdgcn_readfirstlane((int)d.z), (unsigned)__builtin_amdgcn_readfirstlane((int)d.w)};
    return u;
}
__device__ __forceinline__ void srv_store(const __amdgpu_buffer_rsrc_t r, unsigned off, unsigned a, unsigned b, unsigned tag)
{
    const v4u w = {a, tag, b, tag};
    __builtin_amdgcn_raw_buffer_store_b128(w, r, off, 0, RES_AUX_SC1);
}
__device__ __forceinline__ bool srv_claimed(v4u c) { return c.y == 1u && c.w == 1u; }

__device__ void resident_server(const PhysDev& p, int first, int n_steps, int write_forces_last)
{
    __shared__ QShare qsrv[4];
    // a UNIT serves one particle: a pair of wavefronts (four units per workgroup) or, when the launch has server workgroups to spare, a quad
    // (two units: see QQUAD) — `r` is the wavefront's place in its unit
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, wpp = p.srv_quad ? 4 : 2, pair = wave / wpp, r = wave % wpp;
    if (tid < 4) { qsrv[tid].arrived[0] = 0; qsrv[tid].arrived[1] = 0; qsrv[tid].arrived[2] = 0; qsrv[tid].arrived[3] = 0; qsrv[tid].spin = p.spin_limit < (1u << 30) ? 4u * p.spin_limit : 0xffffffffu; }
    __syncthreads();
    // slots are claimed in increasing order: slot = pair * (server workgroups) + workgroup, so that the first claims each get a CU of their
    // own (a pair that shares its two SIMDs with another busy pair of the same workgroup ran its queries slower)
    const int g = pair * (p.srv_slots / (8 / wpp)) + ((int)blockIdx.x - 8 * p.cb);
    if (g >= p.srv_slots) return; // (whole pairs)
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.srv_claim, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(p.srv_rr, 0, 0x7fffffff, 0x00020000);
    // every lane loads the same words: decisions are wave-uniform, and the two wavefronts of a pair reach the same ones (a claim is
    // written — write-through, drained — before its block counts itself out, so "everybody left and no claim" is final)
    unsigned ei = 0, k = 0;
    for (unsigned spins = 0;; ++spins) {
        v4u c = srv_load_uniform(rc, (unsigned)g * (unsigned)SRV_LINE);
        if (srv_claimed(c)) { ei = c.x; k = c.z; break; }
        if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(p.srv_ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >= p.nb * p.ne) {
            c = srv_load_uniform(rc, (unsigned)g * (unsigned)SRV_LINE);
            if (srv_claimed(c)) { ei = c.x; k = c.z; break; }
            return;
        }
        if (spins >= p.spin_limit) return; // (a stuck launch is reported by the blocks' own limits)
        // an idle pair polls rarely (a claim is waited for once per particle and launch; ~500 idle wavefronts polling at the rate of the
        // hand-offs slowed every halo exchange of the launch: 2.87 vs 2.50 us per free substep of the rope)
        __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127);
    }
    ei = (unsigned)__builtin_amdgcn_readfirstlane((int)ei); k = (unsigned)__builtin_amdgcn_readfirstlane((int)k);
    const int e = (int)(ei / (unsigned)p.N), i = (int)(ei % (unsigned)p.N);
    const size_t eb = (size_t)e * p.N;
    const unsigned base = ei * (unsigned)SRV_REC;
    const TriIds tids = load_tri_ids(p, lane, r & 1);
    int qpar = QPAIR | (p.srv_quad ? QQUAD : 0);
    ResidentIO io;
    io.srv_on = false; io.srv_need = false; io.boxes = nullptr; io.step_boxes = nullptr;
    io.x = mk(0.f, 0.f, 0.f); io.v = io.x;
    const StateM none = {nullptr, 0};
    // where this wavefront of the pair is (fault reports only): {phase, substep, last request tag, barrier generation} behind the control words
    const unsigned dbg = (unsigned)SRV_DBG_OFF + (unsigned)g * (unsigned)SRV_LINE + (unsigned)r * 16u;
#define R2S_SRV_STATE(ph, tg) do { if (lane == 0) { const v4u w_ = {(unsigned)(ph), k, (unsigned)(tg), (unsigned)qpar}; __builtin_amdgcn_raw_buffer_store_b128(w_, rc, dbg, 0, RES_AUX_SC1); } } while (0)
    // An OWNING pair (p.srv_own) takes one request — the claim's substep, forces already summed by the block — and from then on advances the
    // particle by itself: lane n of the pair (128 lanes >= the slice's slots) holds slot n of the particle's adjacency row, polls that
    // neighbour's three exchange granules of version k (the same records the blocks hand their halos over with; a served neighbour's are
    // republished by its block), evaluates the one spring, the pair sums, and the substep continues as for a request.  The block is no
    // longer between two substeps of a particle in contact: it takes the result, republishes it, and that is all.
    const bool own = p.srv_own != 0;
    const int pl = r * 64 + lane; // (a quad's last two wavefronts hold no slots: a slice has at most 128)
    const unsigned xn = ((unsigned)p.N + 7u) & ~7u, xe = (unsigned)e * 6u * xn, xb = 3u * xn * 16u; // (k_steps_resident's exchange array)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(p.xch, 0, 0x7fffffff, 0x00020000);
    unsigned noff = 0, roff = 0;
    float sk = 0.f, sa = 0.f, m1 = 1.f, inv_m1 = 1.f;
    bool live = false;
    if (own) {
        const int b = i / SLICE, l = i - b * SLICE;
        const int srow = p.slice_off[b] / GROUP, nslot = p.slice_deg[b];
        if (pl < nslot) {
            const size_t el = (size_t)(srow + (pl / GROUP) * SLICE + l) * GROUP + (size_t)(pl % GROUP);
            const unsigned off = ((const unsigned short*)p.adj_idx)[el];
            sk = ((const float*)p.adj_k)[el]; sa = ((const float*)p.adj_ir)[el];
            const int w = (int)(off >> 3);
            const int gid = w < SLICE ? b * SLICE + w : p.halo_ids[p.halo_off[b] + (w - SLICE)];
            live = gid != i && (sk != 0.f || sa != 0.f); // (padding and inactive slots point at the owner: zero force)
            noff = (xe + (unsigned)gid) * 16u;
            roff = ((unsigned)e * (unsigned)p.N + (unsigned)gid) * (unsigned)SRV_REC + (unsigned)SRV_RES;
        }
        m1 = p.masses[i]; inv_m1 = 1.0f / m1;
    }
    f3 sx = mk(0.f, 0.f, 0.f), sv = sx; // the particle's state of version k, once the pair has produced one
    bool have = false;
    const unsigned k_first = k; // the claim's substep
#ifdef R2S_PHASE_PROBE // wall clock (100 MHz) of the pair's first wavefront by phase, summed over the substeps it served: wait | force + sum | finish | store; [4] substeps, [5] poll passes
    long long sp_acc[6] = {0, 0, 0, 0, 0, 0}, sp_t = (long long)wall_clock64();
#define R2S_SSTAMP(kk) do { const long long now_ = (long long)wall_clock64(); sp_acc[kk] += now_ - sp_t; sp_t = now_; } while (0)
#else
#define R2S_SSTAMP(kk) do { } while (0)
#endif
    while ((int)k < n_steps) {
        TriRegs tr = load_tris(p, e, first + (int)k, tids); // in flight while the request / the neighbours' records are awaited
        R2S_SRV_STATE(1, 0);
        f3 x0, v;
        unsigned t0 = 0;
        if (!(own && have)) {
            v4u r0 = {0u, 0u, 0u, 0u};
            for (unsigned spins = 0;; ++spins) {
                r0 = srv_load_uniform(rr, base);
                t0 = r0.y;
                if (r0.w == t0 && t0 >= 2u * (k + 1u)) break;
                if (spins >= p.spin_limit) return;
                __builtin_amdgcn_s_sleep(1);
            }
            if (t0 == SRV_END) { R2S_SRV_STATE(9, t0); return; }
            const unsigned ks = (t0 >> 1) - 1u; // a later substep's tag: the ones in between were skipped (a request always waits for its result)
            if (!(t0 & 1u)) { k = ks + 1u; continue; }
            if (ks != k) { k = ks; tr = load_tris(p, e, first + (int)k, tids); }
            R2S_SRV_STATE(2, t0);
            v4u r1 = {0u, 0u, 0u, 0u}, r2 = r1;
            for (unsigned spins = 0;; ++spins) {
                r1 = srv_load_uniform(rr, base + 16u); r2 = srv_load_uniform(rr, base + 32u);
                if (r1.y == t0 && r1.w == t0 && r2.y == t0 && r2.w == t0) break;
                if (spins >= p.spin_limit) return;
            }
            x0 = mk(__uint_as_float(r0.x), __uint_as_float(r0.z), __uint_as_float(r1.x));
            v = mk(__uint_as_float(r1.z), __uint_as_float(r2.x), __uint_as_float(r2.z));
        } else {
            const unsigned bo = noff + (k & 1u) * xb;
            v4u d0 = {0u, 0u, 0u, 0u}, d1 = d0, d2 = d0;
            bool pend = live;
            // the result of this substep (version k + 1) overwrites version k - 1 in its half of the result line: not before the block has
            // taken that one — seen from here when the block's republished copy of it is in the exchange array (first lane of the pair)
            bool pend_ack = pl == 0 && k >= k_first + 2u;
            const unsigned ao = (xe + (unsigned)i) * 16u + ((k - 1u) & 1u) * xb;
            for (unsigned spins = 0;; ++spins) {
                if (pend) {
                    // the neighbour's records of version k: in the exchange array (published by its block) or, one hand-off earlier for a
                    // served neighbour, where its own pair left them (same three granules, same version tag)
                    d0 = srv_load(rx, bo); d1 = srv_load(rx, bo + xn * 16u); d2 = srv_load(rx, bo + 2u * xn * 16u);
                    const unsigned ro = roff + (k & 1u) * 64u;
                    const v4u e0 = srv_load(rr, ro), e1 = srv_load(rr, ro + 16u), e2 = srv_load(rr, ro + 32u);
                    pend = !(d0.y == k && d0.w == k && d1.y == k && d1.w == k && d2.y == k && d2.w == k);
                    if (pend && e0.y == k && e0.w == k && e1.y == k && e1.w == k && e2.y == k && e2.w == k) { d0 = e0; d1 = e1; d2 = e2; pend = false; }
                }
                if (pend_ack) {
                    const v4u a0 = srv_load(rx, ao);
                    pend_ack = !(a0.y == k - 1u && a0.w == k - 1u);
                }
                const unsigned long long pm = __builtin_amdgcn_ballot_w64(pend || pend_ack);
#ifdef R2S_PHASE_PROBE
                ++sp_acc[5];
#endif
                if (pm == 0ull) break;
                if (spins >= p.spin_limit) { // a neighbour's record of version k never came
                    if (lane == __builtin_ctzll(pm)) resident_fault(p, 5, g, (int)k, (unsigned)wave, ei, bo / 16u, d0.y, d1.y, d2.y);
                    return;
                }
            }
            R2S_SSTAMP(0);
            v2f fxy = {0.f, 0.f};
            float fz = 0.f;
            if (live) spring_term((v2f){__uint_as_float(d0.x), __uint_as_float(d0.z)}, __uint_as_float(d1.x), (v2f){__uint_as_float(d2.x), __uint_as_float(d2.z)},
                                  __uint_as_float(d1.z), sx, sv, sk, sa, p.dashpot, fxy, fz);
            const float fx = wave_sum(fxy.x), fy = wave_sum(fxy.y), fw = wave_sum(fz); // fixed trees: the same sums in every run
            QShare& qs = qsrv[pair];
            if (lane == 0 && r < 2) { qs.fs[r][0] = fx; qs.fs[r][1] = fy; qs.fs[r][2] = fw; }
            pair_barrier(qs, qpar); // (the two barriers of the queries below separate these reads from the next substep's writes)
            const f3 f = mk(qs.fs[0][0] + qs.fs[1][0], qs.fs[0][1] + qs.fs[1][1], qs.fs[0][2] + qs.fs[1][2]);
            x0 = sx;
            v = vel_update_rcp(p, sv, f, m1, inv_m1);
            R2S_SSTAMP(1);
        }
        const bool last = (int)k == n_steps - 1;
        R2S_SRV_STATE(3, t0);
        R2S_QP_DECL(r == 0 ? g : -1); // (probe builds: the stamps of the pair's last substep — before, first query back, second back, after)
        R2S_QSTAMP();
        if (p.srv_quad)
            finish_wave<3, false, 1, true, true>(p, e, i, eb, first + (int)k, last ? write_forces_last : 0, x0, v, lane == 0, none, &tr, &qsrv[pair], &qpar, nullptr,
                                                 r == 0, &io R2S_QP_ARG);
        else
            finish_wave<3, false, 1, true>(p, e, i, eb, first + (int)k, last ? write_forces_last : 0, x0, v, lane == 0, none, &tr, &qsrv[pair], &qpar, nullptr,
                                           r == 0, &io R2S_QP_ARG);
        R2S_QSTAMP();
        if (qpar & QFAIL) { // the pair's other wavefront did not reach a barrier of this substep
            if (lane == 0) resident_fault(p, 4, g, (int)k, (unsigned)wave, ei, (unsigned)qpar, (unsigned)qsrv[pair].arrived[0], (unsigned)qsrv[pair].arrived[1], t0);
            return;
        }
        if (have) R2S_SSTAMP(2);
        // (lane 0 carries the particle; the other lanes' io is their own scratch)
        sx = mk(bcast(io.x.x, 0), bcast(io.x.y, 0), bcast(io.x.z, 0));
        sv = mk(bcast(io.v.x, 0), bcast(io.v.y, 0), bcast(io.v.z, 0));
        if (r == 0 && lane == 0) {
            // version k + 1 into half (k + 1) & 1 of the result line: an owning pair may be a substep ahead of its block (it waits for its
            // neighbours' records, not for a request), never two — version k + 2 needs a neighbour's version k + 1, which nobody has before
            // the block has taken version k (from the block itself, or through its republished copy)
            const unsigned tag = k + 1u, ro = base + (unsigned)SRV_RES + (tag & 1u) * 64u;
            srv_store(rr, ro, __float_as_uint(io.x.x), __float_as_uint(io.x.y), tag);
            srv_store(rr, ro + 16u, __float_as_uint(io.x.z), __float_as_uint(io.v.z), tag);
            srv_store(rr, ro + 32u, __float_as_uint(io.v.x), __float_as_uint(io.v.y), tag);
        }
#ifdef R2S_PHASE_PROBE
        if (have) { R2S_SSTAMP(3); ++sp_acc[4]; } else sp_t = (long long)wall_clock64();
#endif
        have = true;
        k = k + 1u;
    }
#ifdef R2S_PHASE_PROBE
    if (r == 0 && lane == 0 && g < 1024) for (int kk = 0; kk < 6; ++kk) g_phase_probe[16384 + g * 8 + kk] = sp_acc[kk];
#endif
}

// The same kernel is the small-batch layout's PER-SUBSTEP kernel (n_steps = 1: no hand-off at all, the window comes from the state
// arrays, wavefront 0 alone finishes and owns every side effect): the contact flavours — deferred mesh queries, self-collision
// candidates (SELF; only ever with n_steps = 1) — keep their finishing kernels and a launch per substep, but a block's springs are
// still shared by eight wavefronts instead of walked by one (k_substep<64,512,..>: 8.0 us per substep of the rope, this: see DESIGN §4).
template <int RCAP, bool SELF, int MESH>
__global__ void __launch_bounds__(RES_THREADS, 2) k_steps_resident(const PhysDev p, const StateC xv_in, const StateM xv_out, int first, int n_steps,
                                                                    int write_forces_last)
{
    constexpr int B = SLICE, NW = RES_THREADS / 64;
    constexpr int KT = ((RCAP - B) * 3 + RES_THREADS - 1) / RES_THREADS; // hand-off tasks (halo record, plane) per lane
    typedef __attribute__((address_space(3))) v2f lds_v2f;
    __shared__ __attribute__((aligned(16))) v2f win_s[3 * (RCAP + 1)]; // planes xy | (z, vz) | vxy like the fused substep's window
    __shared__ float4 part_s[NW][B]; // partial forces of the eight wavefronts: one 16-byte write per lane, eight 16-byte reads per finishing lane
    __shared__ volatile int fail_s;
    if ((int)blockIdx.x >= 8 * p.cb) { // workgroups beyond the blocks' own: mesh-query servers (small scenes only)
        if (MESH == 1 && !SELF) resident_server(p, first, n_steps, write_forces_last);
        return;
    }
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int item = xcd * p.cb + q;      // XCD c owns a contiguous run of blocks: most hand-offs stay inside one L2
    if (q >= p.cb || item >= p.nb * p.ne) return;
    const int b = item / p.ne, e = p.e0 + (item - b * p.ne);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = b * B + lane;
    const bool valid = i < p.N;
    const int ic = min(i, p.N - 1);
    const size_t eb = (size_t)e * p.N;
    const unsigned xn = ((unsigned)p.N + 7u) & ~7u; // plane stride: whole 128-byte lines, so that no line has two writer blocks (see SRV_LINE)
    const unsigned xe = (unsigned)e * 6u * xn, xb = 3u * xn * 16u; // exchange array: [env][buffer][plane][particle, padded to 8] x 16 B
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(p.xch, 0, 0x7fffffff, 0x00020000);
    __attribute__((address_space(3))) char* win_w = (__attribute__((address_space(3))) char*)win_s;
    const __attribute__((address_space(3))) char* win = win_w;
    // wavefronts 0..2 (alone on their SIMDs while the others wait) finish the particle, wavefront q publishes plane q; a single substep
    // publishes nothing: wavefront 0 alone
    const bool finisher = wave < (n_steps == 1 ? 1 : 3);

    // ---- once per launch: hand-off tasks, window of substep 0 from the state arrays, adjacency into registers ----
    const int h0 = p.halo_off[b], nh = p.halo_off[b + 1] - h0, nt = 3 * nh;
    unsigned t_off[KT], t_lds[KT], pend0 = 0;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int t = tid + RES_THREADS * k;
        t_off[k] = 0; t_lds[k] = 0;
        if (t < nt) {
            const int pl = t / nh, r = t - pl * nh;
            const int hid = p.halo_ids[h0 + r];
            t_off[k] = (xe + (unsigned)pl * xn + (unsigned)hid) * 16u;
            t_lds[k] = (unsigned)(pl * (RCAP + 1) + B + r) * 8u;
            pend0 |= 1u << k;
            win_s[pl * (RCAP + 1) + B + r] = xv_in.p[st_at(xv_in.n, eb + (size_t)hid, pl)];
        }
    }
    if (wave < 3) win_s[wave * (RCAP + 1) + lane] = xv_in.p[st_at(xv_in.n, eb + (size_t)ic, wave)];
    // group g of the slice belongs to wavefront g % NW; groups [0, gi) only touch the block's own records, [gi, ng) its halo
    const int srow = __builtin_amdgcn_readfirstlane(p.slice_off[b] / GROUP);
    const int ng = __builtin_amdgcn_readfirstlane(p.slice_deg[b] / GROUP), gi = __builtin_amdgcn_readfirstlane(p.slice_int[b] / GROUP);
    const int n_own = gi > wave ? (gi - wave + NW - 1) / NW : 0;            // this wavefront's interior groups: wave, wave + NW, ... < gi
    const int n_all = ng > wave ? (ng - wave + NW - 1) / NW : 0, n_halo = n_all - n_own;
    AdjGroup ag_own[RES_NG], ag_halo[RES_NG];
#pragma unroll
    for (int j = 0; j < RES_NG; ++j) {
        ag_own[j].idx = make_uint2(0u, 0u); ag_own[j].k = make_float4(0.f, 0.f, 0.f, 0.f); ag_own[j].a = ag_own[j].k;
        ag_halo[j] = ag_own[j];
        if (j < n_own) ag_own[j] = adj_load(p, srow + (wave + NW * j) * SLICE, lane);
        if (j < n_halo) ag_halo[j] = adj_load(p, srow + (wave + NW * (n_own + j)) * SLICE, lane);
    }
    const float m1 = p.masses[ic];
    if (tid == 0) fail_s = 0;
    const float inv_m1 = 1.0f / m1;
    ResidentIO io;
    io.x = mk(0.f, 0.f, 0.f); io.v = io.x;
    const bool srv_on = MESH == 1 && !SELF && p.srv_slots > 0 && n_steps > 1;
    io.srv_on = srv_on; io.srv_need = false;
    constexpr int RES_STAGE_MESH = 8;        // meshes whose per-substep boxes are staged in LDS at the top of every substep (more: loaded where they are used)
    __shared__ float sbox_s[6 * RES_STAGE_MESH];
    const bool stage_boxes = MESH != 0 && n_steps > 1 && p.n_mesh <= RES_STAGE_MESH;
    io.step_boxes = stage_boxes ? sbox_s : nullptr;
    bool srv_mine = false; // wavefront 0: this lane's particle has a server pair
    bool srv_ever = false; // finishing wavefronts, owning servers: this lane's particle has needed a query in this launch (it is its pair's from then on)
    const bool srv_own = srv_on && p.srv_own != 0;
    int srv_slot = -1;
    const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(p.srv_rr, 0, 0x7fffffff, 0x00020000);
    const unsigned sbase = ((unsigned)e * (unsigned)p.N + (unsigned)ic) * (unsigned)SRV_REC;
    __shared__ float box_s[8 * (1 + RES_MAX_MESH)];
    io.boxes = (MESH && n_steps > 1) ? box_s : nullptr;
    if (MESH && n_steps > 1) { // unions of the mesh boxes over the launch's substeps (once per launch: a few loads per lane, a reduction through LDS)
        float mb[RES_MAX_MESH][6];
#pragma unroll
        for (int m = 0; m < RES_MAX_MESH; ++m)
#pragma unroll
            for (int c = 0; c < 6; ++c) mb[m][c] = c < 3 ? 3e38f : -3e38f;
        const int n_static = p.n_mesh - p.n_dyn_mesh, n_box = n_steps * p.n_dyn_mesh + n_static;
        for (int t = tid; t < n_box; t += RES_THREADS) {
            const int m = t < n_static ? p.n_dyn_mesh + t : (t - n_static) % p.n_dyn_mesh, slot = min(m, RES_MAX_MESH - 1);
            const float* bb = t < n_static ? p.aabb_static + ((size_t)e * n_static + t) * 6
                                           : p.aabb_dyn + (((size_t)e * p.n_sub + first) * p.n_dyn_mesh + (t - n_static)) * 6;
#pragma unroll
            for (int mm = 0; mm < RES_MAX_MESH; ++mm)
                if (mm == slot)
#pragma unroll
                    for (int c = 0; c < 3; ++c) { mb[mm][c] = fminf(mb[mm][c], bb[c]); mb[mm][3 + c] = fmaxf(mb[mm][3 + c], bb[3 + c]); }
        }
        __shared__ float ub_s[NW][RES_MAX_MESH][6];
#pragma unroll
        for (int m = 0; m < RES_MAX_MESH; ++m)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                float u = mb[m][c];
                for (int o = 32; o > 0; o >>= 1) {
                    const float other = __shfl_xor(u, o, 64);
                    u = c < 3 ? fminf(u, other) : fmaxf(u, other);
                }
                if (lane == 0) ub_s[wave][m][c] = u;
            }
        __syncthreads();
        if (tid == 0) {
            float ub[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f}, mgmax = 0.f;
            for (int m = 0; m < RES_MAX_MESH; ++m) {
                for (int c = 0; c < 6; ++c) {
                    float u = ub_s[0][m][c];
                    for (int w = 1; w < NW; ++w) u = c < 3 ? fminf(u, ub_s[w][m][c]) : fmaxf(u, ub_s[w][m][c]);
                    box_s[8 + 8 * m + c] = u;
                    ub[c] = c < 3 ? fminf(ub[c], u) : fmaxf(ub[c], u);
                }
                float mg = 0.f;
                for (int mm = m; mm < p.n_mesh; mm += (m == RES_MAX_MESH - 1 ? 1 : p.n_mesh)) mg = fmaxf(mg, mesh_margin(p, mm)); // slot m: mesh m (the last slot: every mesh from it on)
                const float r = mg + RES_RANGE_PAD;
                box_s[8 + 8 * m + 6] = m < p.n_mesh ? r * r * 1.0001f : 0.f;
                if (m < p.n_mesh) mgmax = fmaxf(mgmax, mg);
            }
            for (int c = 0; c < 6; ++c) box_s[c] = ub[c];
            const float r = mgmax + RES_RANGE_PAD;
            box_s[6] = r * r * 1.0001f;
        }
        // visible to the finishing wavefronts after barrier A of the first substep
    }
#ifdef R2S_PHASE_PROBE // wall clock (100 MHz) spent per phase by wavefront 0, summed over the launch: own gather + poll | halo gather + reduce | finish | publish; [4] poll passes
    long long pr_acc[5] = {0, 0, 0, 0, 0}, pr_t = (long long)wall_clock64();
    const long long pr_w0 = pr_t, pr_c0 = (long long)__builtin_readcyclecounter(); // shader clock = cycles / wall ticks x 100 MHz
#define R2S_RSTAMP(kk) do { const long long now_ = (long long)wall_clock64(); pr_acc[kk] += now_ - pr_t; pr_t = now_; } while (0)
#else
#define R2S_RSTAMP(kk) do { } while (0)
#endif

    for (int k = 0; k < n_steps; ++k) {
        const int step = first + k;
        const bool last = k == n_steps - 1;
        __syncthreads(); // A: the block's own records of version k are in the window (k = 0: its halo too)
        if (stage_boxes && tid >= RES_THREADS - 64 && lane < 6 * p.n_mesh) { // the last wavefront: this substep's mesh boxes -> LDS (read after barrier C)
            const int m = lane / 6, c = lane - 6 * m;
            const float* bb = m < p.n_dyn_mesh ? p.aabb_dyn + (((size_t)e * p.n_sub + step) * p.n_dyn_mesh + m) * 6
                                               : p.aabb_static + ((size_t)e * (p.n_mesh - p.n_dyn_mesh) + (m - p.n_dyn_mesh)) * 6;
            sbox_s[lane] = bb[c];
        }
        const v2f oa = win_s[lane], ob = win_s[RCAP + 1 + lane], oc = win_s[2 * (RCAP + 1) + lane];
        const f3 x0 = mk(oa.x, oa.y, ob.x), v0 = mk(oc.x, oc.y, ob.y);

        // the halo of version k (the state after k substeps of this launch) comes from buffer k & 1.  The interior springs go first — the
        // neighbours' records are still on their way anyway — then the first poll pass (RES_PRE = RES_NG; issuing it before or between the
        // interior groups only adds passes that find nothing: 2.41 / 2.35 / 2.29 us per substep for RES_PRE 0 / 1 / 2)
        const unsigned bofs = (unsigned)(k & 1) * xb;
        unsigned pend = k > 0 ? pend0 : 0u;
        v4u d[KT];
        v2f fxy = {0.f, 0.f};
        float fz = 0.f;
        const int n_pre = min(n_own, RES_PRE);
        if (RES_PRE > 0) spring_groups_n<RCAP>(p, n_pre, ag_own, win, x0, v0, fxy, fz);
        asm volatile("" ::: "memory");
        if (R2S_RES_DELAY > 0 && k > 0) __builtin_amdgcn_s_sleep(R2S_RES_DELAY);
#pragma unroll
        for (int kk = 0; kk < KT; ++kk)
            if (pend & (1u << kk)) d[kk] = __builtin_amdgcn_raw_buffer_load_b128(rx, t_off[kk] + bofs, 0, RES_AUX_LOAD);
        asm volatile("" ::: "memory");
        if (RES_PRE < RES_NG) spring_groups_n<RCAP>(p, n_own - n_pre, ag_own + (RES_PRE < RES_NG ? RES_PRE : 0), win, x0, v0, fxy, fz);

        for (unsigned spins = 0;; ++spins) {
#pragma unroll
            for (int kk = 0; kk < KT; ++kk)
                if ((pend & (1u << kk)) && d[kk].y == (unsigned)k && d[kk].w == (unsigned)k) {
                    *(lds_v2f*)(win_w + t_lds[kk]) = (v2f){__uint_as_float(d[kk].x), __uint_as_float(d[kk].z)};
                    pend &= ~(1u << kk);
                }
#ifdef R2S_PHASE_PROBE
            ++pr_acc[4];
#endif
            if (__builtin_amdgcn_ballot_w64(pend != 0) == 0ull) break;
            if (spins >= p.spin_limit) {
                {   // the first still-pending task of the first lane that has one: which neighbour record, what its tags read
                    const unsigned long long pm = __builtin_amdgcn_ballot_w64(pend != 0);
                    if (pm && lane == __builtin_ctzll(pm)) {
                        int kk0 = 0;
#pragma unroll
                        for (int kk = KT - 1; kk >= 0; --kk) if (pend & (1u << kk)) kk0 = kk;
                        unsigned off0 = 0, ty = 0, tw = 0;
#pragma unroll
                        for (int kk = 0; kk < KT; ++kk) if (kk == kk0) { off0 = t_off[kk]; ty = d[kk].y; tw = d[kk].w; }
                        resident_fault(p, 2, item, k, (unsigned)wave, (unsigned)lane, off0 / 16u, ty, tw, pend);
                        fail_s = 1;
                    }
                }
                break;
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < KT; ++kk)
                if (pend & (1u << kk)) d[kk] = __builtin_amdgcn_raw_buffer_load_b128(rx, t_off[kk] + bofs, 0, RES_AUX_LOAD);
        }
        __syncthreads(); // B: the halo records are in the window
        if (fail_s) break;
        R2S_RSTAMP(0);

        spring_groups_n<RCAP>(p, n_halo, ag_halo, win, x0, v0, fxy, fz);
        part_s[wave][lane] = make_float4(fxy.x, fxy.y, fz, 0.f);
        __syncthreads(); // C
        if (finisher) {
            f3 f;
            {
                static_assert(NW == 8, "the fixed summation tree below");
                float4 q[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) q[w] = part_s[w][lane];
                f.x = ((q[0].x + q[1].x) + (q[2].x + q[3].x)) + ((q[4].x + q[5].x) + (q[6].x + q[7].x));
                f.y = ((q[0].y + q[1].y) + (q[2].y + q[3].y)) + ((q[4].y + q[5].y) + (q[6].y + q[7].y));
                f.z = ((q[0].z + q[1].z) + (q[2].z + q[3].z)) + ((q[4].z + q[5].z) + (q[6].z + q[7].z));
            }
#ifdef R2S_PHASE_PROBE
            if (f.x == 1.2345e33f) return;
#endif
            R2S_RSTAMP(1);

            // update_vel_from_force, mesh_collision, integrate_ground_collision — the same in the three finishing wavefronts; wavefront 0
            // stores / accumulates
            const f3 v = vel_update_rcp(p, v0, f, m1, inv_m1);
            StateM out = xv_out;
            if (!last) out.p = nullptr;
            io.x = x0; io.v = v0;
            bool fin = valid && !(srv_own && srv_ever); // (a particle a server pair owns is not finished here — wavefront 0 takes its state from the pair)
            if (SELF) { // as in substep_body: particles with candidates publish v_before_collision and are finished by k_self_finish / k_contact_finish
                const int ncand = valid ? p.coll_num[eb + i] : 0;
                if (ncand > 0) {
                    const size_t po = par_off(p, step);
                    p.vbc[po + eb + i] = make_float4(v.x, v.y, v.z, 0.f);
                    p.xbc[po + eb + i] = make_float4(x0.x, x0.y, x0.z, 0.f);
                    fin = false;
                    if (MESH != 0 && (p.mesh_defer || MESH == 2)) {
                        bool near;
                        if (mesh_need(p, e, step, x0 + v * p.dt, 0.002f, near)) {
                            p.fault[1] = 1;
                            if (MESH == 2) {
                                if (mesh_rec_push(p, e, step, i, ncand, x0, v)) p.cand_mark[po + eb + i] = step + 1;
                            } else {
                                const int slot = atomicAdd(p.mesh_cnt + step, 1);
                                if (slot < p.mesh_cap) {
                                    p.mesh_list[po + slot] = make_int2(e | (ncand << 12), i | (int)0x80000000);
                                    p.cand_mark[po + eb + i] = step + 1;
                                }
                            }
                        }
                        const unsigned long long nm = __builtin_amdgcn_ballot_w64(near);
                        if (nm && lane == __builtin_ctzll(nm)) p.mesh_cnt[p.n_sub] = 1;
                    }
                }
            }
            R2S_QP_DECL(-1);
            io.srv_need = false;
            finish_wave<MESH, MESH != 0, 0, true>(p, e, i, eb, step, last ? write_forces_last : 0, x0, v, fin, out, nullptr, nullptr, nullptr, nullptr, wave == 0, &io R2S_QP_ARG);
            // particles that need a mesh query were handed to a server pair (resident_server), not finished above.  Wavefront 0 alone waits
            // for their results — the other two finishing wavefronts leave those lanes to it (three wavefronts polling the same granules
            // tripled the poll traffic on the hand-offs of a block with twenty particles in a finger's reach) — and publishes all three planes
            bool sneed = false, early_pub = false; // early_pub: wavefront 0 has published its finished lanes already (wave-uniform)
            if (MESH == 1 && !SELF && srv_on) {
                const bool need_now = io.srv_need; // the same in the three finishing wavefronts (same inputs, same instructions)
                srv_ever = srv_ever || need_now;
                sneed = srv_own ? srv_ever : need_now;
                const unsigned uk = (unsigned)k;
                if (wave == 0) {
                    bool inplace = srv_own && srv_ever && !srv_mine && !need_now; // owning servers, no pair was left at its first need: in place from then on
                    bool claimed_now = false;
                    if (need_now && !srv_mine) {
                        const int slot = atomicAdd(p.srv_ctl, 1);
                        if (slot < p.srv_slots) {
                            srv_mine = true; srv_slot = slot; claimed_now = true;
                            const __amdgpu_buffer_rsrc_t rcl = __builtin_amdgcn_make_buffer_rsrc(p.srv_claim, 0, 0x7fffffff, 0x00020000);
                            srv_store(rcl, (unsigned)slot * (unsigned)SRV_LINE, (unsigned)e * (unsigned)p.N + (unsigned)i, uk, 1u);
                        } else {
                            inplace = true; // no pair left: answered in place, below —
                            p.fault[2] = 1; // — thousands of instructions inside the hand-off chain: the host takes the next steps off the resident launch
                        }
                    }
                    if (srv_mine && (!srv_own || claimed_now)) { // owning servers: ONE request, with the claim
                        if (need_now) {
                            const unsigned tag = 2u * (uk + 1u) + 1u;
                            srv_store(rsv, sbase + 16u, __float_as_uint(x0.z), __float_as_uint(v.x), tag);
                            srv_store(rsv, sbase + 32u, __float_as_uint(v.y), __float_as_uint(v.z), tag);
                            srv_store(rsv, sbase, __float_as_uint(x0.x), __float_as_uint(x0.y), tag);
                        } else
                            srv_store(rsv, sbase, 0u, 0u, 2u * (uk + 1u)); // nothing in reach in this substep: the pair skips it
                    }
                    if (__builtin_amdgcn_ballot_w64(inplace) != 0ull) // (wave-uniform branch: finish_wave's queries are per lane here)
                        finish_wave<MESH, false, 1, true>(p, e, i, eb, step, last ? write_forces_last : 0, x0, v, sneed && inplace, out, nullptr, nullptr, nullptr, nullptr,
                                                          true, &io R2S_QP_ARG);
                    if (__builtin_amdgcn_ballot_w64(sneed && !inplace) != 0ull) {
                        // the lanes that are finished publish BEFORE the wait: their records are what the pairs (and the neighbour blocks) need
                        // for the next substep — behind the wait, every substep of a particle in contact paid a second hand-off for them
                        if (!last) {
                            early_pub = true;
                            if (!sneed) {
                                if (valid) {
                                    const v4u w = {__float_as_uint(io.x.x), (unsigned)(k + 1), __float_as_uint(io.x.y), (unsigned)(k + 1)};
                                    __builtin_amdgcn_raw_buffer_store_b128(w, rx, (xe + (unsigned)i) * 16u + (unsigned)((k + 1) & 1) * xb, 0, RES_AUX_SC1);
                                }
                                win_s[lane] = (v2f){io.x.x, io.x.y};
                            }
                        }
                        for (unsigned spins = 0;; ++spins) {
                            bool ok = true;
                            if (sneed && !inplace) {
                                const unsigned ro = sbase + (unsigned)SRV_RES + ((uk + 1u) & 1u) * 64u;
                                const v4u d0 = srv_load(rsv, ro), d1 = srv_load(rsv, ro + 16u), d2 = srv_load(rsv, ro + 32u);
                                if (d0.y == uk + 1u && d0.w == uk + 1u && d1.y == uk + 1u && d1.w == uk + 1u && d2.y == uk + 1u && d2.w == uk + 1u) {
                                    io.x = mk(__uint_as_float(d0.x), __uint_as_float(d0.z), __uint_as_float(d1.x));
                                    io.v = mk(__uint_as_float(d2.x), __uint_as_float(d2.z), __uint_as_float(d1.z));
                                } else
                                    ok = false;
                            }
                            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                            if (spins >= p.spin_limit) {
                                const unsigned long long pm = __builtin_amdgcn_ballot_w64(!ok);
                                if (pm && lane == __builtin_ctzll(pm)) {
                                    const __amdgpu_buffer_rsrc_t rcl = __builtin_amdgcn_make_buffer_rsrc(p.srv_claim, 0, 0x7fffffff, 0x00020000);
                                    const unsigned dbo = (unsigned)SRV_DBG_OFF + (unsigned)max(srv_slot, 0) * (unsigned)SRV_LINE;
                                    const v4u d0 = srv_load(rsv, sbase + (unsigned)SRV_RES + ((uk + 1u) & 1u) * 64u), sa = srv_load(rcl, dbo), sb = srv_load(rcl, dbo + 16u);
                                    // context: particle | slot, result tag seen, then the pair's two wavefronts: phase << 28 | substep << 14 | barrier generation, request tag
                                    resident_fault(p, 3, item, k, (unsigned)i | ((unsigned)srv_slot << 20), d0.y, (sa.x << 28) | (sa.y << 14) | (sa.w >> 16), sa.z,
                                                   (sb.x << 28) | (sb.y << 14) | (sb.w >> 16), sb.z);
                                    fail_s = 1;
                                }
                                break;
                            }
                        }
                        if (last && sneed && !inplace && xv_out.p != nullptr) st_store(xv_out, eb + i, io.x, io.v);
                    }
                }
            }
#ifdef R2S_PHASE_PROBE
            if (io.x.x == 1.2345e33f) return;
#endif
            R2S_RSTAMP(2);

            if (!last) { // publish version k + 1 (plane `wave`; wavefront 0: all three planes of its served lanes) and refresh the block's own records in the window
                const unsigned tag = (unsigned)(k + 1);
                const unsigned pub = (xe + (unsigned)i) * 16u + (unsigned)((k + 1) & 1) * xb;
                if (!(sneed && wave != 0) && !(early_pub && !sneed)) {
                    const float va = wave == 0 ? io.x.x : (wave == 1 ? io.x.z : io.v.x), vb = wave == 0 ? io.x.y : (wave == 1 ? io.v.z : io.v.y);
                    if (valid) {
                        const v4u w = {__float_as_uint(va), tag, __float_as_uint(vb), tag};
                        __builtin_amdgcn_raw_buffer_store_b128(w, rx, pub + (unsigned)wave * xn * 16u, 0, RES_AUX_SC1);
                    }
                    win_s[wave * (RCAP + 1) + lane] = (v2f){va, vb};
                }
                if (sneed && wave == 0) { // (sneed implies valid)
                    const v4u w1 = {__float_as_uint(io.x.z), tag, __float_as_uint(io.v.z), tag}, w2 = {__float_as_uint(io.v.x), tag, __float_as_uint(io.v.y), tag};
                    __builtin_amdgcn_raw_buffer_store_b128(w1, rx, pub + xn * 16u, 0, RES_AUX_SC1);
                    __builtin_amdgcn_raw_buffer_store_b128(w2, rx, pub + 2u * xn * 16u, 0, RES_AUX_SC1);
                    win_s[(RCAP + 1) + lane] = (v2f){io.x.z, io.v.z};
                    win_s[2 * (RCAP + 1) + lane] = (v2f){io.v.x, io.v.y};
                }
            }
            R2S_RSTAMP(3);
        }
    }
    if (MESH == 1 && !SELF && srv_on) { // end this block's server pairs, then count the block out (pairs nobody claimed leave when every block has)
        if (wave == 0 && srv_mine) srv_store(rsv, sbase, 0u, 0u, SRV_END);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.srv_ctl + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef R2S_PHASE_PROBE
    if (tid == 0 && item < 8192 / 2) {
        for (int kk = 0; kk < 4; ++kk) g_phase_probe[item * 8 + kk] = pr_acc[kk];
        g_phase_probe[item * 8 + 4] = pr_acc[4];
        g_phase_probe[item * 8 + 5] = (long long)__builtin_readcyclecounter() - pr_c0;
        g_phase_probe[item * 8 + 6] = (long long)wall_clock64() - pr_w0;
    }
#endif
}

// object_collision for ONE particle by a whole wavefront / a group of lanes: the lanes stride over its candidates (up to 500,
// each a dependent gather of the partner's position and published velocity), `G` = lanes per particle (a power of two).
template <int G>
__device__ __forceinline__ f3 self_impulse(const PhysDev& p, size_t po, size_t eb, int i, bool act, f3 x0, f3 v, int sub,
                                           int cnt)
{
    float valid = 0.f, m1 = 1.f;
    f3 Jsum = mk(0.f, 0.f, 0.f);
    if (act) {
        m1 = p.masses[i];
        const int mask1 = p.masks[i];
        for (int k = sub; k < cnt; k += G) { // cnt rides in the list entry: the candidate indices load in the same round trip as x0 / v
            const int j = p.coll_idx[(eb + i) * (size_t)p.coll_cap + k];
            const f3 x2 = xyz(p.xbc[po + eb + j]);
            const f3 v2 = xyz(p.vbc[po + eb + j]); // j lists i too (the candidate relation is symmetric; a capped row still has
                                                    // coll_num > 0), so j published its position and velocity in the fused kernel (po: this substep's parity)
            const float m2 = p.masses[j];
            const f3 dis = x2 - x0;
            const float dis_len = len(dis);
            const f3 rv = v2 - v;
            if (mask1 != p.masks[j] && dis_len < p.cd && dot(dis, rv) < -1e-4f) {
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
    }
#pragma unroll
    for (int o = G / 2; o >= 1; o >>= 1) { // xor shuffles stay inside the aligned group of G lanes
        valid += __shfl_xor(valid, o); Jsum.x += __shfl_xor(Jsum.x, o); Jsum.y += __shfl_xor(Jsum.y, o); Jsum.z += __shfl_xor(Jsum.z, o);
    }
    return (act && valid > 0.f) ? v - (Jsum / valid) / m1 : v;
}

// object_collision + loop (:132-193, :230-268) for the particles on the candidate list, then the rest of the substep.
// 16 lanes per particle: the lanes stride over its candidates (up to 500, each a dependent gather of the partner's position
// and published velocity — serial in one lane that was 25+ us for a squeezed limb), the group sums J and the hit count, and
// the group's first lane carries the particle through finish_wave (which defers it to k_contact_finish if it also touches a
// mesh).  The per-pair arithmetic is the reference's, the sum order over candidates is lane-strided instead of sequential.
template <int MESH>
__global__ void __launch_bounds__(256) k_self_finish(const PhysDev p, const StateC xv_in, const StateM xv_out, int step,
                                                     int write_forces)
{
    constexpr int G = 16;
    const int sub = (int)(threadIdx.x & (G - 1)), grp = (int)(threadIdx.x / G), gpb = (int)(blockDim.x / G);
    // the candidate lists are per environment: group g of the launch walks slots g / ne, g / ne + stride, ... of environment e0 + g % ne
    // (this chain's environments only; a wave-uniform trip count: the group shuffles inside run with their lanes together)
    const int g = (int)blockIdx.x * gpb + grp, stride = (int)gridDim.x * gpb / p.ne;
    const int e = p.e0 + g % p.ne;
    const int n = g / p.ne < stride ? p.cand_cnt_env[e] : 0;
    for (int t = g / p.ne; __builtin_amdgcn_ballot_w64(t < n) != 0ull; t += stride) {
        const bool act = t < n;
        const int2 ei = p.cand_list[(size_t)e * p.N + (act ? t : 0)];
        const int i = ei.y, cnt = ei.x >> 12;
        const size_t eb = (size_t)e * p.N, po = par_off(p, step);
        const f3 x0 = xyz(p.xbc[po + eb + i]);
        const f3 v = self_impulse<G>(p, po, eb, i, act, x0, xyz(p.vbc[po + eb + i]), sub, cnt);
        R2S_QP_DECL(-1);
        finish_wave<MESH, MESH != 0>(p, e, i, eb, step, write_forces, x0, v, act && sub == 0, xv_out, nullptr, nullptr, nullptr, nullptr, true, nullptr R2S_QP_ARG);
    }
}

// ONE finishing kernel per substep for everything the fused kernel could not finish in its own thread (captured into the
// graph flavours used while something is near a mesh):
//   part 1  the mesh list, one WORKGROUP per particle: particles whose query was deferred, and — tagged — particles that
//           also have self-collision candidates (their impulses are applied first, 64 lanes over the candidates);
//           MESHQ = 3: every mesh small, the substep's triangles live in registers (two wavefronts, 128 threads);
//           MESHQ = 2: a large mesh, box hierarchy (four wavefronts);
//   part 2  (WITH_SELF) the remaining particles of the candidate list, 16 lanes each, finished in place.
// Both parts only read what the fused kernel published, so they need no order between them: one launch boundary per
// substep instead of two (k_self_finish + a mesh kernel), and the two kinds of work overlap.
// The body is shared by the stand-alone kernel (k_contact_finish: its own launch behind the fused kernel; results into the state array)
// and by the head of k_substep_pf (PFOUT: the finishers of the PREVIOUS substep at the head of a launch; results into p.pf_res).
// `L` / `n_wg`: this finishing workgroup's number and their count; `nthr`: its live threads (128 for MESHQ 3, else 256).
// Nothing here reads the state arrays: positions come from the records / p.xbc, velocities from p.vbc / p.vdef, all of the substep's parity.
template <int MESHQ, bool WITH_SELF, bool PFOUT>
__device__ __forceinline__ void contact_finish_body(const PhysDev& p, const StateM xv_out, int step, int write_forces, int L, int n_wg, int nthr, QShare& qshare)
{
    // The few wavefronts of this code are a chain of dependent round trips that the whole env step waits for, and they share
    // the chip with the fused kernels: let them win the instruction-issue arbitration on their SIMDs.
#ifndef R2S_NO_FINISH_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    // Latency is everything here (a wavefront per particle, a handful of dependent round trips, the env step waits): the list
    // entry is loaded together with the count (speculatively: entries past the count are stale, never used), it carries the
    // candidate count so that the candidate indices load with x0 / v, and the box test is skipped (NEED = 1 / 2).
    // one WORKGROUP per listed particle — four wavefronts (MESHQ 2) or two (MESHQ 3, 128 threads) that run the same code on the
    // same particle (identical results) and share the triangles of the queries; only the first wavefront stores
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    const size_t po = par_off(p, step);
    int qpar = 0;
#ifdef R2S_PHASE_PROBE
    const long long probe_entry = (long long)wall_clock64();
#endif
    // Large-mesh scenes (MESHQ 2): the workgroups are (environment, slot) pairs — environment fastest, so the workgroups dispatched first are
    // slot 0 of every environment, the ones that have work — and the list is the ENVIRONMENT's, of self-contained records: record count,
    // record (x0, v, candidate count) and the mesh's rigid transform are ONE round trip (measured on the 25k-face pusher scene: 23.8 ->
    // 22.3 us per contact substep).  Small scenes keep the chain-wide list of (env, particle) entries: their triangles
    // hang on the triangle ids, a second round trip either way, and the per-environment form cost them 0.3 - 0.8 us.
    // (MESHQ 2 also serves scenes of SMALL meshes with more than 128 faces in total: their fused kernel is the MESH 1 one and lists
    // chain-wide — `per_env`, uniform, tells the two apart at run time: the records exist only when a large mesh does)
    const bool per_env = MESHQ == 2 && p.mesh_rec != nullptr;
    const int t_stride = per_env ? n_wg / p.ne : n_wg;           // (a head of k_substep_pf is padded to a multiple of 8 workgroups: the surplus idles)
    const int t0 = per_env ? L / p.ne : L;
    const bool in_grid = !per_env || t0 < t_stride;
    const int e_wg = p.e0 + (per_env ? L % p.ne : 0);
    const int4* rec = per_env ? p.mesh_rec + 2 * (po + (size_t)e_wg * p.N) : nullptr;
    int2 ei = make_int2(0, 0);
    int4 ra = make_int4(0, 0, 0, 0), rc = ra;
    if (per_env) { ra = rec[2 * min(t0, p.N - 1)]; rc = rec[2 * min(t0, p.N - 1) + 1]; }
    else ei = p.mesh_list[po + min(t0, p.mesh_cap - 1)];
    TriIds tid = {0, 0, 0, 0, 0, 0, false};
    if (MESHQ == 3) tid = load_tri_ids(p, lane, wave);
    Xf Xw; // the substep's rigid transform of the first large dynamic mesh of this workgroup's environment (identity without one)
#pragma unroll
    for (int j = 0; j < 9; ++j) Xw.r[j] = (j % 4 == 0) ? 1.f : 0.f;
    Xw.t[0] = Xw.t[1] = Xw.t[2] = 0.f;
    if (per_env && p.n_xf > 0) Xw = xf_load_slot(p, __builtin_amdgcn_readfirstlane(e_wg), step, 0);
    const int n_mesh = !in_grid ? 0 : per_env ? min(p.rec_cnt[(size_t)e_wg * p.n_sub + step], p.N) : min(p.mesh_cnt[step], p.mesh_cap);
    // MESHQ 2: what a particle's queries share and what does not depend on the particle (super-cluster records and boxes) — issued BEHIND the
    // record and the count (memory operations return in order: the first query's point must not wait for these; they are needed after its first cluster)
    BlkAux aux;
    if (MESHQ == 2) blk_aux_init(p, aux, lane);
    for (int t = t0; t < n_mesh; t += t_stride) { // a workgroup-uniform trip count (barriers inside)
        bool tagged;
        int e, i, cnt;
        if (per_env) {
            if (t != t0) { ra = rec[2 * t]; rc = rec[2 * t + 1]; }
            tagged = ra.y < 0; e = e_wg; i = ra.y & 0x7fffffff; cnt = ra.x & 0x7ffff; aux.hint = (ra.x >> 19) - 1;
        } else {
            if (t != t0) ei = p.mesh_list[po + t];
            tagged = ei.y < 0; e = ei.x & 0xfff; i = ei.y & 0x7fffffff; cnt = ei.x >> 12;
        }
        const size_t eb = (size_t)e * p.N;
        TriRegs tr;
        if (MESHQ == 3) tr = load_tris(p, e, step, tid); // in flight while the impulses are summed
        aux.X = Xw;
        if (MESHQ == 2 && !per_env && p.n_xf > 0) aux.X = xf_load_slot(p, __builtin_amdgcn_readfirstlane(e), step, 0);
        if (!per_env) aux.hint = -1;
        f3 x0, v;
        if (per_env) {
            x0 = mk(__int_as_float(ra.z), __int_as_float(ra.w), __int_as_float(rc.x));
            v = mk(__int_as_float(rc.y), __int_as_float(rc.z), __int_as_float(rc.w));
        } else {
            x0 = xyz(p.xbc[po + eb + i]);
            v = xyz(tagged ? p.vbc[po + eb + i] : p.vdef[po + eb + i]);
        }
        if (WITH_SELF && tagged) v = self_impulse<64>(p, po, eb, i, true, x0, v, lane, cnt);
        R2S_QP_DECL(step == p.n_sub - 2 ? t * (MESHQ == 2 ? 4 : 2) + wave : -1); // stamps of the last-but-one substep (no force accumulation)
#ifdef R2S_PHASE_PROBE
        if (lane == 0 && qp.wave >= 0 && qp.wave < 1024) g_query_probe[qp.wave * 32 + 31] = probe_entry;
#endif
        R2S_QSTAMP(); // entry loaded, x0 / v (and the impulses) done
        finish_wave<MESHQ, false, 1, false, false, PFOUT>(p, e, i, eb, step, write_forces, x0, v, lane == 0, xv_out, &tr, &qshare, &qpar, &aux, wave == 0, nullptr R2S_QP_ARG);
        R2S_QSTAMP(); // stored
    }
    if (WITH_SELF) {
#ifdef R2S_PHASE_PROBE
        // stamps 28 / 29 / 30: part 2 entered / left, kernel entry of this wavefront; 27: the largest candidate count a group of this wavefront walked.
        // Indexed from the END of the grid (part 2 fills it from there: the busy wavefronts are the ones recorded), rows 512.. of the probe table
        const int gw = 512 + (n_wg - 1 - L) * (nthr >> 6) + wave;
        int probe_cnt = 0;
        if (lane == 0 && gw < 1024 && step == p.n_sub - 2) { g_query_probe[gw * 32 + 28] = (long long)wall_clock64(); g_query_probe[gw * 32 + 30] = probe_entry; }
#endif
        constexpr int G = 16;
        const int sub = (int)(threadIdx.x & (G - 1)), grp = (int)(threadIdx.x / G), gpb = nthr / G;
        // part 1 fills the grid from its first workgroup, part 2 from its LAST: a wavefront that spent 7 us on a mesh particle
        // should not also be the one that starts a candidate particle afterwards (in-kernel stamps: the kernel ended at 10.8 us,
        // 3.3 us after the last mesh particle, with most of the grid idle)
        // the candidate lists are per ENVIRONMENT (round 5; one list for the batch had every chain walk all of it — with the 256
        // finishing workgroups at the head of a k_substep_pf launch that was a second round, the tail of the launch): group g, counted
        // from the back of the grid, walks slots g / ne, g / ne + stride, ... of environment e0 + g % ne
        const int rb = n_wg - 1 - L;
        const int g = rb * gpb + grp, gstride = n_wg * gpb / p.ne;
        const int e = p.e0 + g % p.ne;
        const size_t eb = (size_t)e * p.N;
        const int t0g = g / p.ne;
        int2 ci = p.cand_list[eb + (size_t)min(t0g, p.N - 1)];                 // speculative, with the count (one round trip)
        const int n = t0g < gstride ? p.cand_cnt_env[e] : 0;
        for (int t = t0g; __builtin_amdgcn_ballot_w64(t < n) != 0ull; t += gstride) { // wave-uniform trip count (the group shuffles run with their lanes together)
            if (t != t0g || t >= n) ci = p.cand_list[eb + (size_t)(t < n ? t : 0)]; // (the speculative entry of a slot past the count is stale or was never written: never index with it)
            const int i = ci.y, cnt = ci.x >> 12;
            const bool act = t < n && p.cand_mark[po + eb + i] != step + 1; // not already done in part 1
            const f3 x0 = xyz(p.xbc[po + eb + i]);
            const f3 vpre = xyz(p.vbc[po + eb + i]);
            const f3 v = self_impulse<G>(p, po, eb, i, act, x0, vpre, sub, cnt);
#ifdef R2S_PHASE_PROBE
            probe_cnt = max(probe_cnt, act ? cnt : 0);
#endif
            // the fused kernel's test — widened by 2 mm = 40 m/s of velocity change in one substep — found no mesh in reach of this
            // particle: no query, mesh_collision only advances it.  The bound is CHECKED: an impulse beyond it raises a sticky
            // fault word that the next r2s_phys_step reports (the reference would have applied a mesh response here).
            if (act && sub == 0) {
                const f3 dvi = v - vpre;
                if (dot(dvi, dvi) * p.dt * p.dt > 0.002f * 0.002f) *p.fault = 1;
            }
            R2S_QP_DECL(-1);
            finish_wave<MESHQ == 3 ? 1 : 2, false, 2, false, false, PFOUT>(p, e, i, eb, step, write_forces, x0, v, act && sub == 0, xv_out, nullptr, nullptr, nullptr, nullptr, true, nullptr R2S_QP_ARG);
        }
#ifdef R2S_PHASE_PROBE
        for (int o = 32; o > 0; o >>= 1) probe_cnt = max(probe_cnt, __shfl_xor(probe_cnt, o));
        if (lane == 0 && gw < 1024 && step == p.n_sub - 2) { g_query_probe[gw * 32 + 29] = (long long)wall_clock64(); g_query_probe[gw * 32 + 27] = probe_cnt; }
#endif
    }
}

template <int MESHQ, bool WITH_SELF>
__global__ void __launch_bounds__(256) k_contact_finish(const PhysDev p, const StateC xv_in, const StateM xv_out, int step,
                                                        int write_forces)
{
    __shared__ QShare qshare;
    contact_finish_body<MESHQ, WITH_SELF, false>(p, xv_out, step, write_forces, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y),
                                                 (int)blockDim.x, qshare);
}

// ---- the fused substep with the finishers of the PREVIOUS substep at its head (p.pf; see "finishing at the HEAD of the next launch") ----
// Workgroups [0, p.pf_nfin): contact_finish_body for substep `step - 1` (nothing when `fin_skip`: the first launch of a sequence); the
// rest: substep_body<PF> for substep `step`.  MESHQ 3 finishers live in the workgroup's first two wavefronts; the other two leave at once
// (a hardware barrier counts the wavefronts that have not ended).  One register budget for both roles: the larger one's.
// Register budget of the small-scene form (MESHQ 3: the headline): the fused role needs 72 VGPRs (six wavefronts per SIMD), the finishers 105
// (four).  Measured per batched substep of the headline in the grasp (tools/profiling/variant_bench.py, one box): the launch held to 4 / 5 / 6
// wavefronts per SIMD 23.3 / 22.05 / 22.8 us (two launches: 24.3) — five: 95 VGPRs, three dwords of the finishers spilled.
#ifndef R2S_PF_WAVES3
#define R2S_PF_WAVES3 5
#endif
#ifndef R2S_PF_WAVES3_NOSELF
#define R2S_PF_WAVES3_NOSELF R2S_PF_WAVES3
#endif
template <int B, int RCAP, bool SELF, int MESH, int MESHQ>
__global__ void __launch_bounds__(B, (MESHQ == 3 ? (SELF ? R2S_PF_WAVES3 : R2S_PF_WAVES3_NOSELF) : 1)) k_substep_pf(const PhysDev p, const StateC xv_in, const StateM xv_out, int step, int write_forces, int fin_skip)
{
    if ((int)blockIdx.x < p.pf_nfin) {
        constexpr int NTHR = MESHQ == 3 ? 128 : 256;
        static_assert(B >= NTHR, "the finishers need their wavefronts");
        if (fin_skip || (int)threadIdx.x >= NTHR) return;
        __shared__ QShare qshare_pf;
        contact_finish_body<MESHQ, SELF, true>(p, xv_out, step - 1, 0, (int)blockIdx.x, p.pf_nfin, NTHR, qshare_pf);
        return;
    }
    substep_body<B, RCAP, SELF, MESH, true>(p, xv_in, xv_out, step, write_forces, (int)blockIdx.x - p.pf_nfin);
}

// {particles with candidates, mesh hits of the last substep, grasped environments} -> out[3] (bench.py's phase log: no host sync)
__global__ void k_log_contacts(int E, const int* __restrict__ cand_count, const int* __restrict__ hit_cnt, const int* __restrict__ grasped,
                               int* __restrict__ out)
{
    int hits = 0, g = 0;
    for (int e = threadIdx.x; e < E; e += 64) { hits += hit_cnt ? hit_cnt[e] : 0; g += grasped ? (grasped[e] != 0) : 0; }
    for (int o = 32; o > 0; o >>= 1) { hits += __shfl_down(hits, o, 64); g += __shfl_down(g, o, 64); }
    if (threadIdx.x == 0) { out[0] = cand_count ? *cand_count : 0; out[1] = hits; out[2] = g; }
}

__global__ void k_sum_i32(const int* __restrict__ a, int n, int stride, int* __restrict__ out)
{
    int s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += a[(size_t)i * stride];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// ---- state pack / unpack: caller order [env][user index][3]  <->  internal [env][Morton index]{x,v} -------
__global__ void k_pack(int N, int E, const int* __restrict__ inv, const float* __restrict__ x, const float* __restrict__ v, const StateM xv,
                       const int* __restrict__ env_mask)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (u >= N || (env_mask && env_mask[e] == 0)) return;
    const size_t src = ((size_t)e * N + u) * 3, dst = (size_t)e * N + inv[u];
    float* f = (float*)xv.p; // x and v may be set separately: plane 1 holds one component of each
    if (x) { xv.p[st_at(xv.n, dst, 0)] = (v2f){x[src], x[src + 1]}; f[2 * st_at(xv.n, dst, 1)] = x[src + 2]; }
    if (v) { xv.p[st_at(xv.n, dst, 2)] = (v2f){v[src], v[src + 1]}; f[2 * st_at(xv.n, dst, 1) + 1] = v[src + 2]; }
}
__global__ void k_unpack(int N, int E, const int* __restrict__ inv, const StateC xv, float* __restrict__ x, float* __restrict__ v)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (u >= N) return;
    const size_t dst = ((size_t)e * N + u) * 3, src = (size_t)e * N + inv[u];
    const v2f b = xv.p[st_at(xv.n, src, 1)];
    if (x) { const v2f a = xv.p[st_at(xv.n, src, 0)]; x[dst] = a.x; x[dst + 1] = a.y; x[dst + 2] = b.x; }
    if (v) { const v2f c = xv.p[st_at(xv.n, src, 2)]; v[dst] = c.x; v[dst + 1] = c.y; v[dst + 2] = b.y; }
}
// candidate lists back to the caller's indexing (debug / parity taps)
__global__ void k_lists_to_user(int N, int E, int cap, const int* __restrict__ perm, const int* __restrict__ num, const int* __restrict__ idx,
                                int* __restrict__ num_u, int* __restrict__ idx_u)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const size_t src = (size_t)e * N + i, dst = (size_t)e * N + perm[i];
    const int c = num[src];
    num_u[dst] = c;
    for (int k = 0; k < c; ++k) idx_u[dst * cap + k] = perm[idx[src * cap + k]];
}

// ---- mesh AABBs per (env, substep, dynamic mesh) and per (env, static mesh) ----------------------------
__global__ void k_mesh_aabb_dyn(int E, int n_sub, int n_dyn_mesh, int n_dyn_pts, const int* __restrict__ mesh_vert_off,
                                const int* __restrict__ mesh_kind, const float* __restrict__ interp, float* __restrict__ aabb)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * n_sub * n_dyn_mesh) return;
    const int m = t % n_dyn_mesh;
    if (mesh_kind[m] & 1) return; // large rigid meshes: box from the transformed rest box (k_mesh_xf)
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

// Rigid transform of every large dynamic mesh at every (env, substep), recovered from three reference vertices of the
// interpolated motion (rest frame = vertex positions at construction): orthonormal frames on both sides, R = Fcur Frest^T,
// t = c0 - R r0.  Also the mesh's world AABB (the 8 transformed corners of its rest box: a superset, which keeps the
// early-out conservative) and a rigidity check on a sample of vertices (max deviation -> rigid_err, float bits).
__global__ void k_mesh_xf(int E, int n_sub, int n_dyn_mesh, int n_dyn_pts, int n_xf, const int* __restrict__ xf_mesh,
                          const int* __restrict__ xf_ref, const int* __restrict__ mesh_vert_off, const float* __restrict__ rest,
                          const float* __restrict__ rest_box, const float* __restrict__ interp, float* __restrict__ xf,
                          float* __restrict__ aabb_dyn, unsigned* __restrict__ rigid_err)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * n_sub * n_xf) return;
    const int k = t % n_xf;
    const size_t es = t / n_xf;
    const int m = xf_mesh[k];
    const float* cur = interp + es * n_dyn_pts * 3;
    auto frame = [](f3 p0, f3 p1, f3 p2, f3& e1, f3& e2, f3& e3) {
        e1 = normalize0(p1 - p0);
        const f3 w = p2 - p0;
        e2 = normalize0(w - e1 * dot(w, e1));
        e3 = cross(e1, e2);
    };
    const int i0 = xf_ref[3 * k], i1 = xf_ref[3 * k + 1], i2 = xf_ref[3 * k + 2];
    f3 a1, a2, a3, b1, b2, b3;
    frame(ld3(rest, i0), ld3(rest, i1), ld3(rest, i2), a1, a2, a3);
    frame(ld3(cur, i0), ld3(cur, i1), ld3(cur, i2), b1, b2, b3);
    Xf X;
    // R = b1 a1^T + b2 a2^T + b3 a3^T
    X.r[0] = b1.x * a1.x + b2.x * a2.x + b3.x * a3.x; X.r[1] = b1.x * a1.y + b2.x * a2.y + b3.x * a3.y; X.r[2] = b1.x * a1.z + b2.x * a2.z + b3.x * a3.z;
    X.r[3] = b1.y * a1.x + b2.y * a2.x + b3.y * a3.x; X.r[4] = b1.y * a1.y + b2.y * a2.y + b3.y * a3.y; X.r[5] = b1.y * a1.z + b2.y * a2.z + b3.y * a3.z;
    X.r[6] = b1.z * a1.x + b2.z * a2.x + b3.z * a3.x; X.r[7] = b1.z * a1.y + b2.z * a2.y + b3.z * a3.y; X.r[8] = b1.z * a1.z + b2.z * a2.z + b3.z * a3.z;
    const f3 r0 = ld3(rest, i0), c0 = ld3(cur, i0);
    const f3 rr = xf_rotate(X, r0);
    X.t[0] = c0.x - rr.x; X.t[1] = c0.y - rr.y; X.t[2] = c0.z - rr.z;
    float* o = xf + (size_t)t * 12;
    for (int j = 0; j < 9; ++j) o[j] = X.r[j];
    for (int j = 0; j < 3; ++j) o[9 + j] = X.t[j];
    const float* rb = rest_box + (size_t)k * 6;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int c = 0; c < 8; ++c) {
        const f3 w = xf_apply(X, mk(rb[(c & 1) ? 3 : 0], rb[(c & 2) ? 4 : 1], rb[(c & 4) ? 5 : 2]));
        lo[0] = fminf(lo[0], w.x); hi[0] = fmaxf(hi[0], w.x); lo[1] = fminf(lo[1], w.y); hi[1] = fmaxf(hi[1], w.y);
        lo[2] = fminf(lo[2], w.z); hi[2] = fmaxf(hi[2], w.z);
    }
    float* bb = aabb_dyn + (es * n_dyn_mesh + m) * 6;
    for (int j = 0; j < 3; ++j) { bb[j] = lo[j] - 1e-6f; bb[3 + j] = hi[j] + 1e-6f; }
    const int v0 = mesh_vert_off[m], v1 = mesh_vert_off[m + 1];
    const int stride = max(1, (v1 - v0) / 48);
    float worst = 0.f;
    for (int v = v0; v < v1; v += stride) worst = fmaxf(worst, len(xf_apply(X, ld3(rest, v)) - ld3(cur, v)));
    atomicMax(rigid_err, __float_as_uint(worst));
}

// ---- on-device gripper / pusher kinematics + grasp state machine ----------------------------------------------
// What SpringMassDynamicsModule.step computes on the host before it calls set_mesh_interactive (phystwin.py:362-513),
// for every environment at once and without the D2H read of collision_forces.  float32 where the reference uses float32
// torch ops (same operation order, no FMA contraction), float64 for the host-side python / scipy part (openness state
// machine, interp1d of the finger vertices).
struct EefIn {
    const float* xyz;      // [E,3]   eef_xyz (first gripper)
    const float* vel;      // [E,3]   eef_vel
    const float* rot;      // [E,3,3] eef_rot
    const float* rot_vel;  // [E,3]   eef_rot_vel (axis-angle rate)
    const float* open;     // [E]     gripper_openness
};

// scipy.interpolate.interp1d(kind='linear') over x = arange(K) / (K-1.0), evaluated like scipy's _call_linear:
// hi = clip(searchsorted(x, x_new, 'left'), 1, K-1), slope = (y_hi - y_lo) / (x_hi - x_lo), y = slope * (x_new - x_lo) + y_lo.
__device__ __forceinline__ void eef_knot(double x_new, int K, int& lo, double& x_lo, double& inv_dx_num, double& x_hi)
{
    const double den = (double)(K - 1);
    int a = 0, b = K; // first index with x[i] >= x_new
    while (a < b) { const int m = (a + b) >> 1; if ((double)m / den < x_new) a = m + 1; else b = m; }
    const int hi = min(max(a, 1), K - 1);
    lo = hi - 1;
    x_lo = (double)lo / den; x_hi = (double)hi / den;
    inv_dx_num = x_hi - x_lo;
}

// One workgroup per environment: state machine (thread 0), then the per-vertex quantities that do not depend on the
// substep — relative_eef_pts at the substep-0 end (rel0), eef_pts_delta (delta) — and the finger closing velocities.
__global__ void __launch_bounds__(256) k_eef_prepare(int E, int M, int K, int use_pusher, const double* __restrict__ table, float ix, float iy,
                                                     float iz, float thr, int f_left, int f_right, int nF,
                                                     const float* __restrict__ coll_forces, EefIn in, double* __restrict__ cur_open,
                                                     int* __restrict__ grasped, int* __restrict__ has_state, float* __restrict__ rel0,
                                                     float* __restrict__ delta, float* __restrict__ dyn_vel, float* __restrict__ dyn_omega,
                                                     float two_dt_n)
{
    const int e = blockIdx.x, tid = threadIdx.x;
    __shared__ double s_open[2];
    __shared__ float s_red[2][3][256 / 64];
    if (tid == 0) {
        double now, before;
        if (use_pusher) {
            now = before = 1.0; cur_open[e] = 1.0; // phystwin.py:464, :474-477
        } else {
            double openness = (double)in.open[e]; // gripper_openness.item()
            double cur = has_state[e] ? cur_open[e] : openness; // :371-372
            int g = grasped[e];
            const float* F = coll_forces + (size_t)e * nF * 3;
            float n2[2];
            for (int side = 0; side < 2; ++side) { // :380-389: faces 18, 19, 1 of each finger, float32 sums and norm
                const float* f0 = F + (size_t)((side ? f_right : f_left) + 18) * 3;
                const float* f1 = F + (size_t)((side ? f_right : f_left) + 19) * 3;
                const float* f2 = F + (size_t)((side ? f_right : f_left) + 1) * 3;
                const float x = (f0[0] + f1[0]) + f2[0], y = (f0[1] + f1[1]) + f2[1], z = (f0[2] + f1[2]) + f2[2];
                n2[side] = sqrtf((x * x + y * y) + z * z);
            }
            before = cur;
            if (n2[0] < 100.f && n2[1] < 100.f) g = 0; // :393-394
            if (openness < cur) {                       // :395-405
                if (n2[0] > thr && n2[1] > thr) { openness = cur; g = 1; }
                else if (g) { cur = fmax(openness, cur - 0.05); openness = cur; }
                else cur = openness;
            } else cur = openness;
            cur_open[e] = cur; grasped[e] = g; has_state[e] = 1;
            now = fmin(fmax(openness, 0.0), 1.0); before = fmin(fmax(before, 0.0), 1.0); // np.clip, :411, :419
        }
        s_open[0] = now; s_open[1] = before;
    }
    __syncthreads();
    int lo_n, lo_b; double xl_n, dx_n, xh_n, xl_b, dx_b, xh_b;
    eef_knot(s_open[0], K, lo_n, xl_n, dx_n, xh_n);
    eef_knot(s_open[1], K, lo_b, xl_b, dx_b, xh_b);
    const float* R = in.rot + (size_t)e * 9;
    float accL[3] = {0.f, 0.f, 0.f}, accR[3] = {0.f, 0.f, 0.f};
    const int half = M / 2;
    for (int v = tid; v < M; v += 256) {
        float pn[3], pb[3];
        for (int c = 0; c < 3; ++c) {
            const double yl = table[((size_t)lo_n * M + v) * 3 + c], yh = table[((size_t)(lo_n + 1) * M + v) * 3 + c];
            pn[c] = (float)(((yh - yl) / dx_n) * (s_open[0] - xl_n) + yl);
            const double zl = table[((size_t)lo_b * M + v) * 3 + c], zh = table[((size_t)(lo_b + 1) * M + v) * 3 + c];
            pb[c] = (float)(((zh - zl) / dx_b) * (s_open[1] - xl_b) + zl);
        }
        float d[3] = {pn[0] - pb[0], -(pn[1] - pb[1]), -(pn[2] - pb[2])};              // :422-424 (flip y, z)
        float r[3] = {pb[0] - ix, -(pb[1] - iy), -(pb[2] - iz)};                          // :425-427
        float* o = rel0 + ((size_t)e * M + v) * 3; o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
        float* q = delta + ((size_t)e * M + v) * 3; q[0] = d[0]; q[1] = d[1]; q[2] = d[2];
        if (!use_pusher) { // closing velocity: (delta @ eef_rot[0]^T) / (2 dt n), :446-447
            const float c0 = ((d[0] * R[0] + d[1] * R[1]) + d[2] * R[2]) / two_dt_n;
            const float c1 = ((d[0] * R[3] + d[1] * R[4]) + d[2] * R[5]) / two_dt_n;
            const float c2 = ((d[0] * R[6] + d[1] * R[7]) + d[2] * R[8]) / two_dt_n;
            float* a = v < half ? accL : accR;
            a[0] += c0; a[1] += c1; a[2] += c2;
        }
    }
    // block sums of the two halves (the reference takes torch means; summation order differs in the last bits)
    for (int side = 0; side < 2; ++side)
        for (int c = 0; c < 3; ++c) {
            float x = side ? accR[c] : accL[c];
            for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
            if ((tid & 63) == 0) s_red[side][c][tid >> 6] = x;
        }
    __syncthreads();
    if (tid < 3) {
        const float ev = in.vel[(size_t)e * 3 + tid] * 0.5f; // :443
        if (use_pusher) {
            dyn_vel[(size_t)e * 6 + tid] = ev; dyn_vel[(size_t)e * 6 + 3 + tid] = 0.f;
        } else {
            const float sl = (s_red[0][tid][0] + s_red[0][tid][1]) + (s_red[0][tid][2] + s_red[0][tid][3]);
            const float sr = (s_red[1][tid][0] + s_red[1][tid][1]) + (s_red[1][tid][2] + s_red[1][tid][3]);
            dyn_vel[(size_t)e * 6 + tid] = ev + sl / (float)max(half, 1);          // :448-454
            dyn_vel[(size_t)e * 6 + 3 + tid] = ev + sr / (float)max(M - half, 1);
        }
        dyn_omega[(size_t)e * 3 + tid] = -in.rot_vel[(size_t)e * 3 + tid] * 0.5f; // :457
    }
}

// kornia.geometry.conversions.axis_angle_to_rotation_matrix (third-party, not under the reference tree; restated from
// its published source): Rodrigues with w = aa / (theta + 1e-6) where theta^2 > 1e-6, first-order matrix otherwise.
__device__ __forceinline__ void eef_aa_to_matrix(float ax, float ay, float az, float* r)
{
    const float theta2 = (ax * ax + ay * ay) + az * az;
    if (theta2 > 1e-6f) {
        const float theta = sqrtf(theta2);
        const float wx = ax / (theta + 1e-6f), wy = ay / (theta + 1e-6f), wz = az / (theta + 1e-6f);
        const float c = cosf(theta), sn = sinf(theta), k = 1.0f - c;
        r[0] = c + wx * wx * k;        r[1] = wx * wy * k - wz * sn; r[2] = wy * sn + wx * wz * k;
        r[3] = wz * sn + wx * wy * k;  r[4] = c + wy * wy * k;       r[5] = -wx * sn + wy * wz * k;
        r[6] = -wy * sn + wx * wz * k; r[7] = wx * sn + wy * wz * k; r[8] = c + wz * wz * k;
    } else {
        r[0] = 1.f; r[1] = -az; r[2] = ay; r[3] = az; r[4] = 1.f; r[5] = -ax; r[6] = -ay; r[7] = ax; r[8] = 1.f;
    }
}

// interpolated_dynamic_points / interpolated_center for every (env, substep) and every vertex the stepper reads
// (all vertices of small meshes; for large rigid meshes only the three reference vertices and the rigidity sample).
__global__ void __launch_bounds__(256) k_eef_points(int E, int n_sub, int M, int n_need, const int* __restrict__ need, EefIn in,
                                                    const float* __restrict__ rel0, const float* __restrict__ delta, float dt, float dt_n,
                                                    float* __restrict__ interp, float* __restrict__ center)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y, e = blockIdx.z;
    if (k >= n_need) return;
    const int v = need[k];
    const float dts = (float)(s + 1) * dt;                                   // linspace(1, n, n) * dt, :374
    const float* X = in.xyz + (size_t)e * 3; const float* V = in.vel + (size_t)e * 3; const float* W = in.rot_vel + (size_t)e * 3;
    const float* R = in.rot + (size_t)e * 9;
    const float nx = X[0] + V[0] * dts, ny = X[1] + V[1] * dts, nz = X[2] + V[2] * dts;   // eef_xyz_next, :376
    float D[9];
    eef_aa_to_matrix(W[0] * dts, W[1] * dts, W[2] * dts, D);               // :377-378
    float Rn[9];                                                             // eef_rot_next = D^T @ eef_rot, :379
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = (D[0 * 3 + i] * R[0 * 3 + j] + D[1 * 3 + i] * R[1 * 3 + j]) + D[2 * 3 + i] * R[2 * 3 + j];
    const float* r0 = rel0 + ((size_t)e * M + v) * 3; const float* d = delta + ((size_t)e * M + v) * 3;
    const float rx = r0[0] + (d[0] / dt_n) * dts, ry = r0[1] + (d[1] / dt_n) * dts, rz = r0[2] + (d[2] / dt_n) * dts; // :429
    float* o = interp + (((size_t)e * n_sub + s) * M + v) * 3;             // xyz_next + rel @ Rn^T, :432
    o[0] = nx + ((rx * Rn[0] + ry * Rn[1]) + rz * Rn[2]);
    o[1] = ny + ((rx * Rn[3] + ry * Rn[4]) + rz * Rn[5]);
    o[2] = nz + ((rx * Rn[6] + ry * Rn[7]) + rz * Rn[8]);
    if (k == 0) { float* c = center + ((size_t)e * n_sub + s) * 3; c[0] = nx; c[1] = ny; c[2] = nz; } // :436
}

// ---- warp-style hash grid -----------------------------------------------------------------------------
__device__ __forceinline__ int grid_cell(int x, int y, int z)
{
    const int origin = 1 << 20;
    x = max(0, x + origin); y = max(0, y + origin); z = max(0, z + origin);
    return (z % GRID_DIM) * (GRID_DIM * GRID_DIM) + (y % GRID_DIM) * GRID_DIM + (x % GRID_DIM);
}

// One (cell key, USER index) pair per particle, emitted in user order so that the stable sort leaves every cell's
// points in ascending user index — the traversal order of warp's grid (its ids are the caller's indices).
__global__ void k_grid_keys(int N, int E, const int* __restrict__ inv, const StateC xv, float cell_inv,
                            uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (u >= N) return;
    const float4 q = st_x4(xv, (size_t)e * N + inv[u]);
    const int c = grid_cell((int)(q.x * cell_inv), (int)(q.y * cell_inv), (int)(q.z * cell_inv));
    keys[(size_t)e * N + u] = ((uint32_t)e << GRID_CELL_BITS) | (uint32_t)c;
    vals[(size_t)e * N + u] = (uint32_t)u;
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

// build_resting_collision_pairs, :272-291 (bitset instead of N x N bytes; rows/bits are INTERNAL indices, the
// `index < i` test is on USER indices like the reference)
__global__ void k_build_resting(int N, int E, int words, const int* __restrict__ perm, const int* __restrict__ inv,
                                const StateC xv, float radius, float cell_inv, const uint32_t* __restrict__ keys,
                                const uint32_t* __restrict__ ids, uint32_t* __restrict__ bits, const int* __restrict__ env_mask)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N || (env_mask && env_mask[e] == 0)) return;
    const int ui = perm[i];
    const float4 q = st_x4(xv, (size_t)e * N + i);
    const QBox b = query_box(q, radius, cell_inv);
    uint32_t* my = bits + ((size_t)e * N) * words;
    for (int z = b.zs; z <= b.ze; ++z)
        for (int y = b.ys; y <= b.ye; ++y)
            for (int x = b.xs; x <= b.xe; ++x) {
                int s, t;
                cell_range(keys, e * N, (e + 1) * N, ((uint32_t)e << GRID_CELL_BITS) | (uint32_t)grid_cell(x, y, z), s, t);
                for (int k = s; k < t; ++k) {
                    const int uj = (int)ids[k];
                    if (uj < ui) {
                        const int j = inv[uj];
                        atomicOr(&my[(size_t)i * words + (j >> 5)], 1u << (j & 31));
                        atomicOr(&my[(size_t)j * words + (i >> 5)], 1u << (i & 31));
                    }
                }
            }
}

// update_potential_collision, :196-227 (same candidate order: cells x-fastest, user ids ascending inside a cell)
__global__ void k_candidates(int N, int E, int words, int cap, const int* __restrict__ inv, const StateC xv,
                             const int* __restrict__ masks, float cd, float radius, float cell_inv, const uint32_t* __restrict__ keys,
                             const uint32_t* __restrict__ ids, const uint32_t* __restrict__ bits, int* __restrict__ coll_idx,
                             int* __restrict__ coll_num, int* __restrict__ max_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const size_t eb = (size_t)e * N;
    const float4 q = st_x4(xv, eb + i);
    const f3 x1 = xyz(q);
    const int mask1 = masks[i];
    // The reference visits every cell overlapping [x - 5cd, x + 5cd] and keeps j only if |xj - xi| < cd.  Such a j
    // lies in a cell overlapping [x - cd, x + cd] (int() truncation is monotonic), and dropping the other cells
    // keeps the relative order of the survivors: visiting the smaller box yields the identical list, ~5x cheaper.
    (void)radius;
    const QBox b = query_box(q, cd, cell_inv);
    const uint32_t* row = bits + (eb + i) * words;
    int cnt = 0;
    for (int z = b.zs; z <= b.ze; ++z)
        for (int y = b.ys; y <= b.ye; ++y)
            for (int x = b.xs; x <= b.xe; ++x) {
                int s, t;
                cell_range(keys, e * N, (e + 1) * N, ((uint32_t)e << GRID_CELL_BITS) | (uint32_t)grid_cell(x, y, z), s, t);
                for (int k = s; k < t; ++k) {
                    const int j = inv[ids[k]];
                    if (j == i) continue;
                    const f3 dis = st_x(xv, eb + j) - x1;
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

// Direct cell table for the per-env-step candidate rebuild: tab[(env << 21) | cell] = [first, last+1) in the sorted key
// array; all-zero between calls (the mark kernel fills the occupied cells, k_cell_clear wipes exactly those again), so a
// lookup is one load instead of two 14-step binary searches.  xs[k] = position and INTERNAL index of the k-th sorted
// particle, so a cell's points stream as consecutive 16-byte records instead of three dependent gathers each.
__global__ void k_cell_clear(int N, int E, const uint32_t* __restrict__ keys, int2* __restrict__ tab)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    tab[keys[(size_t)blockIdx.y * N + k]] = make_int2(0, 0);
}
// ---- candidate rebuild on a FINE grid (cell = collision_dist) ---------------------------------------------------------
// The reference's grid has cells of 5 cd and keeps only neighbours closer than cd, so a query walks ~250 points to keep
// a handful.  Here the points are binned at cell = cd, a query looks at its 3x3x3 fine cells (~16 points), and the
// survivors are put into the reference's order afterwards: that order is (coarse cell in z,y,x-lexicographic traversal,
// user index inside a cell), and a survivor's coarse cell differs from the query's by at most one per axis, so the sort
// key is (rank of the coarse-cell offset in 0..26, user index).  Identical lists, ~15x fewer distance tests.
__device__ __forceinline__ uint32_t fine_cell(int x, int y, int z) { return ((uint32_t)(z & 127) << 14) | ((uint32_t)(y & 127) << 7) | (uint32_t)(x & 127); }

__global__ void k_fine_keys(int N, int E, const StateC xv, float cd_inv, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const float4 q = st_x4(xv, (size_t)e * N + i);
    keys[(size_t)e * N + i] = ((uint32_t)e << GRID_CELL_BITS) | fine_cell((int)(q.x * cd_inv), (int)(q.y * cd_inv), (int)(q.z * cd_inv));
    vals[(size_t)e * N + i] = (uint32_t)i;
}
__global__ void k_fine_mark(int N, int E, const StateC xv, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ ids,
                            int2* __restrict__ tab, float4* __restrict__ xs)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (k >= N) return;
    const size_t g = (size_t)e * N + k;
    const uint32_t key = keys[g];
    if (k == 0 || keys[g - 1] != key) tab[key].x = (int)g;
    if (k == N - 1 || keys[g + 1] != key) tab[key].y = (int)g + 1;
    const int j = (int)ids[g];
    const float4 q = st_x4(xv, (size_t)e * N + j);
    xs[g] = make_float4(q.x, q.y, q.z, __int_as_float(j));
}
__device__ __forceinline__ uint64_t cand_key(float4 qi, float4 qj, float cell_inv, int user_j)
{
    const int dx = (int)(qj.x * cell_inv) - (int)(qi.x * cell_inv), dy = (int)(qj.y * cell_inv) - (int)(qi.y * cell_inv),
              dz = (int)(qj.z * cell_inv) - (int)(qi.z * cell_inv);
    return ((uint64_t)(uint32_t)(((dz + 1) * 3 + (dy + 1)) * 3 + (dx + 1)) << 32) | (uint32_t)user_j;
}
__global__ void k_candidates_fine(int N, int E, int words, int cap, const StateC xv, const int* __restrict__ masks,
                                  const int* __restrict__ perm, float cd, float cd_inv, float cell_inv, const int2* __restrict__ tab,
                                  const float4* __restrict__ xs, const uint32_t* __restrict__ bits, int* __restrict__ coll_idx,
                                  int* __restrict__ coll_num, int* __restrict__ max_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const size_t eb = (size_t)e * N;
    const float4 q = st_x4(xv, eb + i);
    const f3 x1 = xyz(q);
    const int mask1 = masks[i];
    const int fx = (int)(q.x * cd_inv), fy = (int)(q.y * cd_inv), fz = (int)(q.z * cd_inv);
    const uint32_t* row = bits + (eb + i) * words;
    int* out = coll_idx + (eb + i) * (size_t)cap;
    int cnt = 0;
    uint64_t worst = 0; // largest key kept so far (only needed once the row is full)
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int2 st = tab[((uint32_t)e << GRID_CELL_BITS) | fine_cell(fx + dx, fy + dy, fz + dz)];
                for (int k = st.x; k < st.y; ++k) {
                    const float4 c = xs[k];
                    const int j = __float_as_int(c.w);
                    if (j == i) continue;
                    if (!(len(xyz(c) - x1) < cd)) continue;
                    if (row[j >> 5] & (1u << (j & 31))) continue; // resting pair (stored symmetrically)
                    if (mask1 == masks[j]) continue;
                    const uint64_t key = cand_key(q, c, cell_inv, perm[j]);
                    cnt++;
                    int n = min(cnt - 1, cap); // entries currently in the row
                    if (n == cap) { // full: keep the cap smallest keys = the first cap of the reference's traversal
                        if (key > worst) continue;
                        n = cap - 1; // the current worst (last entry) drops out
                    }
                    int pos = n; // insertion sort by key
                    while (pos > 0) {
                        const int jp = out[pos - 1];
                        const float4 cp = st_x4(xv, eb + jp);
                        if (cand_key(q, cp, cell_inv, perm[jp]) < key) break;
                        out[pos] = jp;
                        --pos;
                    }
                    out[pos] = j;
                    if (n + 1 == cap) { const int jl = out[cap - 1]; worst = cand_key(q, st_x4(xv, eb + jl), cell_inv, perm[jl]); }
                }
            }
    coll_num[eb + i] = min(cnt, cap);
    if (cnt > 0) atomicMax(max_count, cnt);
}

// compact (env, particle) list of the particles that have candidates (order irrelevant: each is independent)
__global__ void k_cand_list(int N, int E, const int* __restrict__ coll_num, int2* __restrict__ list, int* __restrict__ count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (i >= N) return;
    const int c = coll_num[(size_t)e * N + i];
    if (c > 0) { // per-environment lists (count[4 + e] entries at list + e * N), count[0] = all of them
        list[(size_t)e * N + atomicAdd(count + 4 + e, 1)] = make_int2(e | (c << 12), i); // env (< 2048) | candidate count << 12
        atomicAdd(count, 1);
    }
}

} // namespace

// =========================================================================================================
struct R2SPhys {
    R2SPhysParams prm{};
    int E = 0, N = 0, S = 0, n_slices = 0, ell_len = 0;
    int nb = 0, cb = 0, halo_max = 0; // particle blocks, work items per XCD, largest halo (LDS sizing)
    int pb = 256, rcap = 1024;          // layout: particles (= threads) per workgroup, LDS window capacity in records
    int coll_cap = 500;
    int words = 0;
    int n_mesh = 0, n_dyn_mesh = 0, nF = 0, nV = 0, n_dyn_pts = 0;
    // host copies needed to rebuild stiffness
    std::vector<int> h_springs;
    std::vector<float> h_rest;
    std::vector<int> h_adj_spring; // ELL slot -> spring id (or -1 for padding)
    std::vector<int> h_adj_nbr;    // ELL slot -> neighbour particle (internal id)
    std::vector<int> h_adj_self;   // ELL slot -> owning particle (internal id; padding target)
    std::vector<int> h_adj_loc;    // ELL slot -> LDS record of the neighbour in the owner's block, or ~global id
    std::vector<int> h_perm, h_inv, h_slice_off, h_slice_deg, h_rslice_off, h_rslice_deg, h_slice_int;
    std::vector<int> h_radj_spring, h_radj_nbr, h_radj_self; // remote ELL slot -> spring / neighbour / owner
    std::vector<int> h_mesh_map, h_face_map;
    // device
    v2f* xv[2] = {nullptr, nullptr}; // ping-pong state, three 8-byte planes each (StateC / StateM)
    StateM state(int b) const { return {xv[b], (size_t)E * N}; }
    int cur = 0;
    int *d_slice_off = nullptr, *d_slice_deg = nullptr, *d_rslice_off = nullptr, *d_rslice_deg = nullptr, *d_slice_int = nullptr;
    unsigned short* d_adj_idx = nullptr;
    float *d_adj_k = nullptr, *d_adj_ir = nullptr;
    int4* d_radj = nullptr;
    int rell_len = 0;
    int *d_halo_off = nullptr, *d_halo_ids = nullptr, *d_perm = nullptr, *d_inv = nullptr;
    int *d_num_user = nullptr, *d_idx_user = nullptr;
    float* d_masses = nullptr;
    int* d_masks = nullptr;
    int *d_coll_num = nullptr, *d_coll_idx = nullptr, *d_max_count = nullptr;
    float4* d_vbc = nullptr; float4* d_xbc = nullptr; // [2][E,N] each (substep parity)
    void* d_pf_res = nullptr;   // k_substep_pf: a 128-byte result line per particle
    bool pf_ok = false;         // large-batch layout with meshes: the contact flavours can run with the finishers at the head of the next launch
    int pf_pref = 1;            // R2S_PF=0 / r2s_phys_set_pf(h, 0): keep the two-launch contact flavours (A/B measurements, the bit-identity test)
    int2* d_mesh_list = nullptr; int* d_mesh_cnt = nullptr; int mesh_cap = 0; // deferred mesh queries: [E * N] (a chain's slice starts at its first env), [chains][n_sub + 1]
    float4* d_vdef = nullptr;
    int* d_mq_hint = nullptr;
    int lm[7] = {0, 0, 0, 0, 0, 0, -1}; // PhysDev::lm_c0, lm_nc, lm_f0, lm_f1, lm_y, lm_mesh, lm_slot
    int4* d_mesh_rec = nullptr; int* d_rec_cnt = nullptr; // large-mesh scenes: per-environment records [E][N][2] and their counters [E][n_sub]
    int* d_cand_mark = nullptr;
    // Counters of an env step ([0] particles near a mesh, [1] sticky fault word, [2] a query was needed, [3] server pairs ran out, [4..15] fault
    // context) travel to pinned memory behind the step and pick the FLAVOUR of a later step.  Which later step is fixed (round 5): step t
    // runs the flavour that follows from the counters of step t - LAG, waited for if they have not landed (they have: two env steps ago) —
    // never "whatever copy happens to have arrived", which made the bits of a run in contact depend on host timing (the flavours sum in
    // different orders).  A full set_state starts a new history: its first LAG steps run the default flavour (queries in place).
    static constexpr int LAG = 2, RING = 4;
    int* d_mesh_total = nullptr; int* h_ring = nullptr; // pinned [RING][16]
    hipEvent_t ring_ev[RING] = {}; bool ring_pending[RING] = {}, ring_stale_fault[RING] = {};
    uint64_t step_no = 0; // env steps enqueued since the last full set_state
    void* d_xch = nullptr;    // resident stepper: exchange array (96 B per particle)
    void* d_srv_claim = nullptr; void* d_srv_rr = nullptr; // resident stepper's mesh-query servers: a 128-byte line per claim, one of control words, one per pair of fault-report state; a line of request and a line of result granules per particle
    bool srv_exhausted = false; // a launch ran out of server pairs: per-substep kernels + finishing launch until the contact is over
    bool srv_ok = false;      // small scene (every mesh small, <= 128 faces in total): a resident launch may carry query servers
    int srv_wg_cap = SRV_MAX_SLOTS / 4; // R2S_RES_SRV_WG: at most this many server workgroups per launch
    int n_cu = 256;
    unsigned spin_limit = RES_SPIN_LIMIT;
    int srv_quad = -1;        // R2S_RES_SRV_QUAD=0 / 1: pairs / quads whatever the launch has room for (-1: quads when it has >= 64 server workgroups)
    int srv_quad_for(int n_srv) const { return srv_quad >= 0 ? srv_quad : (n_srv >= 64 ? 1 : 0); }
    int srv_own = 1;          // R2S_RES_SRV_OWN=0: one request per substep instead of pairs that own their particle
    bool split_ok = false;    // 64-particle layout whose slices fit k_steps_resident's registers (no remote neighbours, <= 64 interior / halo slots)
    bool resident_ok = false; // the handle can run the env step as ONE resident launch (k_steps_resident) in its free flavour
    int resident_pref = 1;    // R2S_RESIDENT=0 / r2s_phys_set_tuning: never pick the 64-particle layout / the resident launch
    int chains_override = 0;  // > 0: tuning override of chains() (R2S_CHAINS at create, r2s_phys_set_tuning later)
    int force_defer = -1;     // >= 0: force the deferred-query flavour on / off (tests, tuning)
    int last_flavour[4] = {0, 0, 0, 1}; // of the last step: self-collision variant, mesh template, deferred queries, chains
    std::vector<float> h_logY; // last log stiffness (re-clamped when spring_Y_min / max change)
    int* d_hit_cnt = nullptr;
    hipEvent_t rigid_event = nullptr;
    int mesh_defer = 0; // this env step's graph flavour: 1 = needy particles are finished by k_contact_finish (contact likely), 0 = in place
    int2* d_cand_list = nullptr;
    int* d_cand_count = nullptr;
    int* h_cand_count = nullptr; // pinned; filled asynchronously by update_collision_graph
    hipEvent_t cand_event = nullptr;
    bool cand_pending = false;
    int n_cand = 0;              // particles with candidates after the last update (host view)
    int chains() const // parallel kernel chains of the captured env step
    {
        // chains are separate graphs on separate streams (hardware queues; more than four lose: 26 / 42 us per substep with six).
        // Measured per batched substep, free / contact (tools/profiling/variant_bench.py, round 3): 32 sloth envs (1888 work items)
        // 1 chain 23.7 / 25.7, 2 chains 20.8 / 26.8, 4 chains 19.1 / 26.0 us; 8 sloth envs x 4 views (472 items) 10.6 / 17.3,
        // 9.5 / 17.2, 10.6 / 19.3; 32 T-block envs with the 25k-face rod (288 items) 10.4 / 25.9, 10.5 / 24.4, 12.7 / 26.3.
        const int64_t items = (int64_t)nb * E;
        int c = items >= 1536 ? 4 : (items >= 256 ? 2 : 1);
        if (pb == 64) c = 1; // small batches (the resident layout): one chain
        c = std::min(c, E);
        if (chains_override > 0) c = std::max(1, std::min(chains_override, std::min(E, 8)));
        return c;
    }
    uint32_t *d_bits = nullptr, *d_keys[2] = {nullptr, nullptr}, *d_ids[2] = {nullptr, nullptr};
    int2* d_cell_tab = nullptr;  // [E << 21] direct cell table (null when it would exceed 4 GiB: binary search instead)
    float4* d_cell_xs = nullptr; // [E,N] sorted positions + internal index
    char* d_sort_tmp = nullptr;
    size_t sort_bytes = 0;
    int *d_faces = nullptr, *d_mesh_map = nullptr, *d_face_map = nullptr, *d_mesh_face_off = nullptr, *d_mesh_vert_off = nullptr;
    int *d_face_orig = nullptr, *d_mesh_kind = nullptr,
        *d_mesh_xf = nullptr, *d_xf_mesh = nullptr, *d_xf_ref = nullptr;
    float *d_cl_box = nullptr, *d_xf = nullptr, *d_rest_pts = nullptr, *d_pnorm = nullptr, *d_xf_rest_box = nullptr, *d_tri_rest = nullptr;
    int4* d_cl_info = nullptr; int4* d_sup_info = nullptr; float* d_sup_box = nullptr; int* d_small_mesh = nullptr; int n_sup = 0, n_small = 0;
    unsigned* d_rigid_err = nullptr;
    unsigned* h_rigid_err = nullptr; // pinned
    bool rigid_pending = false;
    int n_cl = 0, n_xf = 0;
    bool any_large = false; // some mesh has more than 256 faces -> cluster hierarchy + wave-cooperative queries
    // on-device eef kinematics (r2s_phys_set_eef_table / r2s_phys_set_eef_motion)
    std::vector<int> h_mesh_kind, h_voff, h_foff, h_xf_mesh, h_xf_ref;
    double* d_eef_table = nullptr; int eef_knots = 0; float eef_init[3] = {0.f, 0.f, 0.f}; float eef_thr = 0.f;
    double* d_eef_open = nullptr; int *d_eef_grasped = nullptr, *d_eef_has = nullptr, *d_eef_need = nullptr; int eef_n_need = 0;
    float *d_eef_rel0 = nullptr, *d_eef_delta = nullptr;
    float *d_mesh_pts = nullptr, *d_interp = nullptr, *d_center = nullptr, *d_dyn_vel = nullptr, *d_dyn_omega = nullptr;
    float *d_aabb_dyn = nullptr, *d_aabb_static = nullptr, *d_coll_forces = nullptr;
    // graph
    // two captured variants of the num_substeps step: [0] no particle has candidates (one kernel per substep),
    // [1] some do (fused kernel + self-collision finishing kernel per substep)
    // slot = defer * 4 + variant * 2 + start buffer: with an odd substep count (667) the state buffer flips every env step, so both
    // parities are kept instead of re-capturing 667 nodes per step
    // One graph per CHAIN and flavour, launched on the chain's own stream by r2s_phys_step (round 3; round 2 captured the chains as
    // branches of ONE graph: the same kernels then ran 20.8 / 24.2 / 28.4 us per batched substep free / with an idle finishing
    // launch / in contact, against 19.1 / 19.8 / 26.0 as separate graphs on four streams — a branch of a hipGraph is not a
    // hardware queue of its own, a stream is).
    static constexpr int MAX_CHAINS = 8;
    hipGraph_t graph[MAX_CHAINS][8] = {};
    hipGraphExec_t graph_exec[MAX_CHAINS][8] = {};
    hipEvent_t chain_fork = nullptr, chain_join[MAX_CHAINS] = {};
    // timing
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int last_kernels = 0;
    bool ev_pending = false;

    PhysDev dev() const
    {
        PhysDev p{};
        p.N = N; p.E = E; p.n_sub = prm.num_substeps;
        p.nb = nb; p.cb = cb; p.e0 = 0; p.ne = E;
        p.slice_off = d_slice_off; p.slice_deg = d_slice_deg; p.slice_int = d_slice_int; p.adj_idx = (const uint2*)d_adj_idx; p.adj_k = (const float4*)d_adj_k; p.adj_ir = (const float4*)d_adj_ir; p.rslice_off = d_rslice_off; p.rslice_deg = d_rslice_deg; p.radj = d_radj;
        p.halo_off = d_halo_off; p.halo_ids = d_halo_ids; p.perm = d_perm; p.inv = d_inv;
        p.masses = d_masses; p.masks = d_masks;
        p.dt = prm.dt; p.dashpot = prm.dashpot_damping; p.drag_factor = expf(-prm.dt * prm.drag_damping);
        p.rf = prm.reverse_z ? -1.f : 1.f; p.cd = prm.collision_dist;
        auto cl = [](float x, float lo, float hi) { return std::min(std::max(x, lo), hi); };
        p.ce = cl(prm.collide_elas, 0.f, 1.f); p.cf = cl(prm.collide_fric, 0.f, 2.f);
        p.cee = cl(prm.collide_eef_elas, 0.f, 1.f); p.cef = cl(prm.collide_eef_fric, 0.f, 2.f);
        p.cse = cl(prm.collide_self_elas, 0.f, 1.f); p.csf = cl(prm.collide_self_fric, 0.f, 2.f);
        p.self_collision = prm.self_collision; p.use_pusher = prm.use_pusher;
        p.coll_num = d_coll_num; p.coll_idx = d_coll_idx; p.coll_cap = coll_cap;
        p.vbc = d_vbc; p.xbc = d_xbc; p.par_stride = (size_t)E * N; p.cand_list = d_cand_list; p.cand_count = d_cand_count; p.cand_cnt_env = d_cand_count ? d_cand_count + 4 : nullptr;
        p.pf = 0; p.pf_nfin = 0; p.pf_res = d_pf_res;
        p.mesh_list = d_mesh_list; p.mesh_cnt = d_mesh_cnt; p.mesh_cap = mesh_cap; p.mesh_defer = mesh_defer; p.vdef = d_vdef; p.cand_mark = d_cand_mark; p.mesh_rec = d_mesh_rec; p.rec_cnt = d_rec_cnt; p.mq_hint = d_mq_hint;
        p.lm_c0 = lm[0]; p.lm_nc = lm[1]; p.lm_f0 = lm[2]; p.lm_f1 = lm[3]; p.lm_y = lm[4]; p.lm_mesh = lm[5]; p.lm_slot = lm[6];
        p.n_mesh = n_mesh; p.n_dyn_mesh = n_dyn_mesh; p.nF = nF; p.nV = nV; p.n_dyn_pts = n_dyn_pts;
        p.faces = d_faces; p.mesh_map = d_mesh_map; p.face_map = d_face_map; p.mesh_face_off = d_mesh_face_off;
        p.face_orig = d_face_orig; p.n_cl = n_cl; p.n_xf = n_xf;
        p.cl_box = d_cl_box; p.mesh_kind = d_mesh_kind; p.mesh_xf = d_mesh_xf; p.xf_mesh = d_xf_mesh; p.xf = d_xf; p.rest_pts = d_rest_pts;
        p.pnorm = d_pnorm; p.tri_rest = d_tri_rest; p.cl_info = d_cl_info;
        p.n_sup = n_sup; p.n_small = n_small; p.sup_box = d_sup_box; p.sup_info = d_sup_info; p.small_mesh = d_small_mesh;
        p.mesh_pts = d_mesh_pts; p.interp_pts = d_interp; p.interp_center = d_center; p.dyn_vel = d_dyn_vel; p.dyn_omega = d_dyn_omega;
        p.aabb_dyn = d_aabb_dyn; p.aabb_static = d_aabb_static; p.coll_forces = d_coll_forces; p.hit_cnt = d_hit_cnt;
        p.fault = d_mesh_total ? d_mesh_total + 1 : nullptr;
        p.xch = d_xch;
        p.spin_limit = spin_limit;
        p.srv_own = srv_own; p.srv_quad = 0; p.srv_slots = 0; p.srv_claim = d_srv_claim; p.srv_rr = d_srv_rr; p.srv_ctl = d_srv_claim ? (int*)((char*)d_srv_claim + SRV_CTL_OFF) : nullptr;
        return p;
    }
};

namespace {

template <typename T>
int dev_alloc(T** p, size_t count)
{
    R2S_HIP_TRY(r2s::dev_malloc((void**)p, sizeof(T) * (count ? count : 1)));
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

void drop_graph_fwd(R2SPhys* h);

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
    if (log_Y != h->h_logY.data()) h->h_logY.assign(log_Y, log_Y + h->S);
    stiffness_from_log(h, log_Y, k, act);
    // Slots of inactive springs (gate exp(logY) > Ymin fails, :75) and padding slots point at the particle
    // itself with k = 0: then d = 0 and dv = 0, so neither the spring nor the dashpot term contributes.
    auto fbits = [](float f) { int i; memcpy(&i, &f, 4); return i; };
    // Padding slots and slots of inactive springs (gate exp(logY) > Ymin fails, :75) point at the owner itself with
    // k = 0: d = 0 and dv = 0, so neither the spring nor the dashpot term contributes.
    std::vector<unsigned short> ell_idx(h->ell_len);
    std::vector<float> ell_k(h->ell_len, 0.f), ell_ir(h->ell_len, 0.f);
    std::vector<int4> rell(h->rell_len);
    // host lists are slot-major (slot n of lane l of slice s at off[s] + n*64 + l); the device wants them GROUP-major
    // (off[s] + (n/4)*256 + l*4 + n%4) and the window record as a byte offset (record * 8)
    for (int sl = 0; sl < h->n_slices; ++sl)
        for (int n = 0; n < h->h_slice_deg[sl]; ++n)
            for (int ln = 0; ln < SLICE; ++ln) {
                const int t = h->h_slice_off[sl] + n * SLICE + ln;
                const int o = h->h_slice_off[sl] + (n / GROUP) * (SLICE * GROUP) + ln * GROUP + n % GROUP;
                const int sp = h->h_adj_spring[t], self = h->h_adj_self[t];
                if (sp >= 0 && act[sp]) { ell_idx[o] = (unsigned short)(h->h_adj_loc[t] * 8); ell_k[o] = k[sp]; ell_ir[o] = k[sp] / h->h_rest[sp]; }
                else ell_idx[o] = (unsigned short)((self % h->pb) * 8);
            }
    for (int t = 0; t < h->rell_len; ++t) {
        const int sp = h->h_radj_spring[t], self = h->h_radj_self[t];
        rell[t] = (sp >= 0 && act[sp]) ? make_int4(h->h_radj_nbr[t], fbits(k[sp]), fbits(k[sp] / h->h_rest[sp]), 0) : make_int4(self, 0, 0, 0);
    }
    int rc = upload(h->d_adj_idx, ell_idx.data(), ell_idx.size(), s);
    if (rc) return rc;
    rc = upload(h->d_adj_k, ell_k.data(), ell_k.size(), s);
    if (rc) return rc;
    rc = upload(h->d_adj_ir, ell_ir.data(), ell_ir.size(), s);
    if (rc) return rc;
    return upload(h->d_radj, rell.data(), rell.size(), s);
}

template <int B, int RCAP>
void launch_substep_layout(const PhysDev& p, dim3 grid, const StateC in, const StateM out, int step, int write_forces, bool with_self, int mesh,
                           hipStream_t s)
{
#define R2S_LAUNCH(SELF, MESH) hipLaunchKernelGGL((k_substep<B, RCAP, SELF, MESH>), grid, dim3(B), 0, s, p, in, out, step, write_forces)
    if (with_self) { if (mesh == 2) R2S_LAUNCH(true, 2); else if (mesh == 1) R2S_LAUNCH(true, 1); else R2S_LAUNCH(true, 0); }
    else { if (mesh == 2) R2S_LAUNCH(false, 2); else if (mesh == 1) R2S_LAUNCH(false, 1); else R2S_LAUNCH(false, 0); }
#undef R2S_LAUNCH
}

// true when the env step's flavour carries k_contact_finish (deferred mesh queries + self-collision impulses in one launch)
bool has_contact_finish(const R2SPhys* h, const PhysDev& p)
{
    const int mesh = h->nF > 0 ? (h->any_large ? 2 : 1) : 0;
    return mesh != 0 && (p.mesh_defer || mesh == 2);
}

void launch_fused(R2SPhys* h, const PhysDev& p, int in_buf, int step, int write_forces, bool with_self, hipStream_t s)
{
    dim3 grid(8u * (unsigned)p.cb);
    const int mesh = h->nF > 0 ? (h->any_large ? 2 : 1) : 0;
    const StateC in = h->state(in_buf);
    const StateM out = h->state(in_buf ^ 1);
    if (h->pb == 256) launch_substep_layout<256, 1024>(p, grid, in, out, step, write_forces, with_self, mesh, s);
    else if (h->pb == 128) launch_substep_layout<128, 768>(p, grid, in, out, step, write_forces, with_self, mesh, s);
    else if (!h->split_ok) launch_substep_layout<64, 512>(p, grid, in, out, step, write_forces, with_self, mesh, s);
    else { // small-batch layout: eight wavefronts per 64-particle block (k_steps_resident with one substep)
#define R2S_LAUNCH64(SELF, MESH) hipLaunchKernelGGL((k_steps_resident<512, SELF, MESH>), grid, dim3(RES_THREADS), 0, s, p, in, out, step, 1, write_forces)
        if (with_self) { if (mesh == 2) R2S_LAUNCH64(true, 2); else if (mesh == 1) R2S_LAUNCH64(true, 1); else R2S_LAUNCH64(true, 0); }
        else { if (mesh == 2) R2S_LAUNCH64(false, 2); else if (mesh == 1) R2S_LAUNCH64(false, 1); else R2S_LAUNCH64(false, 0); }
#undef R2S_LAUNCH64
    }
}

// What the fused kernel left unfinished: with something near a mesh (mesh_defer) ONE combined finishing kernel per substep —
// deferred mesh queries, one workgroup per particle, plus the self-collision impulses; otherwise only k_self_finish
// while candidates exist (mesh queries of the rare needy particle in place).
void launch_finish(R2SPhys* h, const PhysDev& p, int in_buf, int step, int write_forces, bool with_self, hipStream_t s)
{
    const int mesh = h->nF > 0 ? (h->any_large ? 2 : 1) : 0;
    const StateC in = h->state(in_buf);
    const StateM out = h->state(in_buf ^ 1);
    if (has_contact_finish(h, p)) {
        const bool small = mesh == 1 && h->nF <= 128; // every mesh small: the substep's triangles fit two per lane
        // 2048 wavefronts (an idle launch costs the same ~2.5 us with 16 workgroups: it is the launch boundary), grid-stride: workgroups of
        // two (small) or four wavefronts; large-mesh scenes: (environments of this chain, slots), a workgroup strides over ITS environment's records
        const dim3 g = mesh == 2 ? dim3((unsigned)p.ne, (unsigned)std::max(16, 512 / std::max(1, p.ne))) : dim3(small ? 1024 : 512);
#define R2S_FIN(Q, S) hipLaunchKernelGGL((k_contact_finish<Q, S>), g, dim3(small ? 128 : 256), 0, s, p, in, out, step, write_forces)
        if (small) { if (with_self) R2S_FIN(3, true); else R2S_FIN(3, false); }
        else { if (with_self) R2S_FIN(2, true); else R2S_FIN(2, false); }
#undef R2S_FIN
    } else if (with_self) {
        // grid-stride over the device-side candidate list; sized for the host's view of the count
        const unsigned blocks = 512; // grid-stride over the device-side candidate list, 16 lanes per listed particle
        if (mesh == 1) hipLaunchKernelGGL((k_self_finish<1>), dim3(blocks), dim3(256), 0, s, p, in, out, step, write_forces);
        else hipLaunchKernelGGL((k_self_finish<0>), dim3(blocks), dim3(256), 0, s, p, in, out, step, write_forces);
    }
}

int launch_substep(R2SPhys* h, const PhysDev& p, int in_buf, int step, int write_forces, bool with_self, hipStream_t s)
{
    launch_fused(h, p, in_buf, step, write_forces, with_self, s);
    launch_finish(h, p, in_buf, step, write_forces, with_self, s);
    return R2S_OK;
}

// The contact flavours of a large batch with the finishers at the head of the next launch (k_substep_pf; PhysDev::pf)
bool pf_flavour(const R2SPhys* h, const PhysDev& p) { return h->pf_ok && h->pf_pref != 0 && h->pb == 256 && has_contact_finish(h, p); }
// finishing workgroups at the head of a launch: enough for the lists of a batch in contact without a second round (a workgroup strides
// over its list if there is more), few enough not to stand between the launch and its fused blocks — every workgroup of the launch
// holds the fused role's LDS window, so an idle finisher costs a block's slot for the microsecond it takes to read an empty list
int pf_head_size(const R2SPhys* h, int ne, bool with_self)
{
    const int mesh = h->nF > 0 ? (h->any_large ? 2 : 1) : 0;
    // small scenes: 32 workgroups per environment (deferred queries from the front, candidate particles — eight to a workgroup — from the back;
    // 16 and 24 ran a second round in the headline's grasp: 25.8 / 24.4 vs 22.8 us), 16 while no particle has candidates
    int n = mesh == 2 ? ne * std::max(16, 256 / std::max(1, ne)) : std::min(1024, (with_self ? 32 : 16) * ne);
    if (const char* ev = getenv("R2S_PF_HEAD")) n = std::max(8, atoi(ev) * ne); // tuning: finishing workgroups per environment
    return (n + 7) & ~7;
}
void launch_fused_pf(R2SPhys* h, const PhysDev& p, int in_buf, int step, int write_forces, bool with_self, int fin_skip, hipStream_t s)
{
    const dim3 grid((unsigned)p.pf_nfin + 8u * (unsigned)p.cb);
    const int mesh = h->nF > 0 ? (h->any_large ? 2 : 1) : 0;
    const bool small = mesh == 1 && h->nF <= 128;
    const StateC in = h->state(in_buf);
    const StateM out = h->state(in_buf ^ 1);
#define R2S_PF(SELF, MESH, Q) hipLaunchKernelGGL((k_substep_pf<256, 1024, SELF, MESH, Q>), grid, dim3(256), 0, s, p, in, out, step, write_forces, fin_skip)
    if (with_self) { if (mesh == 2) R2S_PF(true, 2, 2); else if (small) R2S_PF(true, 1, 3); else R2S_PF(true, 1, 2); }
    else { if (mesh == 2) R2S_PF(false, 2, 2); else if (small) R2S_PF(false, 1, 3); else R2S_PF(false, 1, 2); }
#undef R2S_PF
}

// Enqueue substeps [first, first+n) starting from buffer `start_buf`; the final state is left in buffer
// start_buf ^ (n & 1).
// A captured hipMemsetAsync node zeroes on the first replay only with this runtime (later replays fill the buffer with
// stale host words — caught by a multi-step force test), so the accumulator is cleared by a kernel node instead.
__global__ void k_zero_f32(float* __restrict__ p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.f;
}

// The env step's flavour that runs as one resident launch: no particle with self-collision candidates, nothing near a mesh.
bool resident_flavour(const R2SPhys* h, bool with_self, int mesh_defer)
{
    return h->resident_ok && h->resident_pref != 0 && !with_self && !(h->nF > 0 && mesh_defer);
}
constexpr int RES_MAX_ITEMS = 256; // (block, env) work items of a resident launch: one 512-thread workgroup per CU (two wavefronts per SIMD, each
                                   // with its 2 + 2 adjacency groups in registers), all on the chip at once

int enqueue_steps(R2SPhys* h, int first, int n, int start_buf, bool with_self, hipStream_t s, int e0 = 0, int ne = -1, bool zero_forces = true, int chain_id = 0)
{
    PhysDev p = h->dev();
    if (ne < 0) ne = h->E;
    p.e0 = e0; p.ne = ne; p.cb = (h->nb * ne + 7) / 8;
    const int chain = chain_id;
    if (h->nF > 0) { // this chain's slice of the deferred-query list and counters ([n_sub] = the near-a-mesh count of the env step)
        p.mesh_list = h->d_mesh_list + (size_t)e0 * h->N; // a chain lists only its own environments' particles: at most ne * N per substep
        p.mesh_cap = ne * h->N;
        p.mesh_cnt = h->d_mesh_cnt + (size_t)chain * (h->prm.num_substeps + 1);
        hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((h->prm.num_substeps + 256) / 256)), dim3(256), 0, s, (float*)p.mesh_cnt, (size_t)h->prm.num_substeps + 1);
        if (h->d_rec_cnt) { // large-mesh scenes: the record counters of this chain's environments
            const size_t cw = (size_t)ne * h->prm.num_substeps;
            hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((cw + 255) / 256)), dim3(256), 0, s, (float*)(h->d_rec_cnt + (size_t)e0 * h->prm.num_substeps), cw);
        }
        if (with_self && p.mesh_defer) { // marks of the previous env step must not match this step's substep numbers (both parities)
            const size_t cnt = (size_t)ne * h->N;
            for (int par = 0; par < 2; ++par)
                hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, (float*)(h->d_cand_mark + (size_t)par * h->E * h->N + (size_t)e0 * h->N), cnt);
        }
    }
    if (resident_flavour(h, with_self, p.mesh_defer)) {
        // one launch for all n substeps: the final state goes to the OTHER buffer whatever n is (a late workgroup may still be reading
        // its substep-0 window from the input buffer while an early one stores its last substep) — r2s_phys_step flips accordingly
        if (h->nF > 0 && zero_forces) {
            const size_t cnt = 3 * (size_t)ne * h->nF;
            hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, h->d_coll_forces + 3 * (size_t)e0 * h->nF, cnt);
            hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, (float*)(h->d_hit_cnt + e0), (size_t)ne);
        }
        const size_t xn = ((size_t)h->N + 7) & ~(size_t)7; // (plane stride of the kernel: whole 128-byte lines)
        const size_t words = (size_t)24 * ne * xn; // this chain's environments: 2 buffers x 3 planes x 16 B per particle
        hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, (float*)h->d_xch + (size_t)24 * e0 * xn, words);
        // mesh-query servers: workgroups beyond the blocks' own, as many as the chip has CUs left (the whole launch is resident at once)
        int n_srv = 0;
        if (h->srv_ok && n > 1) {
            // workgroups go to the XCDs round-robin and every XCD must hold its share at once: the grid (8 * cb block workgroups — up to 7
            // of them idle, but their CUs may be on other XCDs than the servers that would need them — plus the servers) <= CUs
            n_srv = std::min(h->n_cu - 8 * p.cb, h->srv_wg_cap);
            if (n_srv < SRV_MIN_WG) n_srv = 0;
        }
        if (n_srv > 0) {
            p.srv_quad = h->srv_quad_for(n_srv); p.srv_slots = (p.srv_quad ? 2 : 4) * n_srv;
            hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((SRV_DBG_OFF / 4 + 255) / 256)), dim3(256), 0, s, (float*)h->d_srv_claim, (size_t)SRV_DBG_OFF / 4); // claims and control words
            const size_t rw = (size_t)(SRV_REC / 4) * ne * h->N;
            hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((rw + 255) / 256)), dim3(256), 0, s, (float*)h->d_srv_rr + (size_t)(SRV_REC / 4) * e0 * h->N, rw);
        }
        const dim3 grid(8u * (unsigned)p.cb + (unsigned)n_srv);
        const StateC in = h->state(start_buf);
        const StateM out = h->state(start_buf ^ 1);
        const bool with_mesh = h->nF > 0;
        if (with_mesh) hipLaunchKernelGGL((k_steps_resident<512, false, 1>), grid, dim3(RES_THREADS), 0, s, p, in, out, first, n, 1);
        else hipLaunchKernelGGL((k_steps_resident<512, false, 0>), grid, dim3(RES_THREADS), 0, s, p, in, out, first, n, 1);
        return R2S_OK;
    }
    int buf = start_buf;
    const bool pf = pf_flavour(h, p);
    if (pf) { // the finishers of substep k ride at the head of substep k + 1's launch; the last substep's are the stand-alone launch
        p.pf = 1; p.pf_nfin = pf_head_size(h, ne, with_self);
        const size_t words = (size_t)(PF_LINE / 4) * ne * h->N; // this chain's result lines: tags of the previous env step must not match
        hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, (float*)h->d_pf_res + (size_t)(PF_LINE / 4) * e0 * h->N, words);
    }
    for (int k = 0; k < n; ++k) {
        const int last = (k == n - 1);
        if (last && h->nF > 0 && zero_forces) { // this chain's slice of the accumulator
            const size_t cnt = 3 * (size_t)ne * h->nF;
            hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, h->d_coll_forces + 3 * (size_t)e0 * h->nF, cnt);
            hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, (float*)(h->d_hit_cnt + e0), (size_t)ne);
        }
        if (pf) {
            launch_fused_pf(h, p, buf, first + k, last, with_self, k == 0, s);
            if (last) { PhysDev q = p; q.pf = 0; launch_finish(h, q, buf, first + k, last, with_self, s); }
        } else {
            int rc = launch_substep(h, p, buf, first + k, last, with_self, s);
            if (rc) return rc;
        }
        buf ^= 1;
    }
    return R2S_OK;
}

void drop_graph(R2SPhys* h)
{
    for (int c = 0; c < R2SPhys::MAX_CHAINS; ++c)
        for (int v = 0; v < 8; ++v) {
            if (h->graph_exec[c][v]) (void)hipGraphExecDestroy(h->graph_exec[c][v]);
            if (h->graph[c][v]) (void)hipGraphDestroy(h->graph[c][v]);
            h->graph_exec[c][v] = nullptr; h->graph[c][v] = nullptr;
        }
}

void drop_graph_fwd(R2SPhys* h) { drop_graph(h); }

// Environments are independent, so the env step runs as `chains` parallel kernel chains over disjoint environment ranges:
// while one chain's workgroups stage their windows (memory phase, VALU idle) or sit in the launch gap between two substeps,
// another chain's are in the gather (VALU phase), and one chain's finishing kernel runs next to the others' fused kernels.
// Each chain is captured into its own graph.
int capture_graph(R2SPhys* h, int variant, int start_buf)
{
    const int slot = h->mesh_defer * 4 + variant * 2 + (start_buf & 1);
    const int chains = h->chains();
    for (int c = 0; c < chains; ++c) {
        if (h->graph_exec[c][slot]) (void)hipGraphExecDestroy(h->graph_exec[c][slot]);
        if (h->graph[c][slot]) (void)hipGraphDestroy(h->graph[c][slot]);
        h->graph_exec[c][slot] = nullptr; h->graph[c][slot] = nullptr;
        const int e0 = (int)((int64_t)h->E * c / chains), e1 = (int)((int64_t)h->E * (c + 1) / chains);
        hipStream_t cs;
        R2S_HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        if (const hipError_t eb = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal); eb != hipSuccess) { (void)hipStreamDestroy(cs); R2S_HIP_TRY(eb); }
        const int rc = enqueue_steps(h, 0, h->prm.num_substeps, start_buf, variant == 1, cs, e0, e1 - e0, true, c);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(cs, &g);
        (void)hipStreamDestroy(cs);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        R2S_HIP_TRY(e);
        h->graph[c][slot] = g;
        R2S_HIP_TRY(hipGraphInstantiate(&h->graph_exec[c][slot], g, nullptr, nullptr, 0));
    }
    return R2S_OK;
}

// The side streams of the chains are shared by every handle of a device: the runtime multiplexes streams onto a handful of
// hardware queues (four by default), and a second handle with streams of its own — the parity-gate rollout next to the bench's,
// two steppers in one test — put its chains on queues that were already taken: 25 -> 33-40 us per batched substep, the same
// collapse as six or eight chains in one handle.  Created on first use, never destroyed (process lifetime).
hipStream_t chain_side_stream(int c)
{
    static hipStream_t pool[16][R2SPhys::MAX_CHAINS] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!pool[dev][c] && hipStreamCreateWithFlags(&pool[dev][c], hipStreamNonBlocking) != hipSuccess) pool[dev][c] = nullptr;
    return pool[dev][c];
}

// Resident launches of one device run one after the other, whatever handle and stream they come from: a launch needs ALL its
// workgroups on the chip at once (one per CU: the kernel's registers leave room for one 512-thread workgroup per CU), and two of them
// dispatched side by side from two hardware queues can each hold part of the chip while waiting for workgroups that no longer fit —
// both would spin into their poll limit.  Each launch waits for the event of the previous one and records its own (same process only:
// two PROCESSES stepping small batches on one GPU can still collide; the poll limit then turns the collision into an error return).
struct ResidentGate {
    std::mutex mu;
    hipEvent_t last[16] = {};
    bool recorded[16] = {};
};
ResidentGate& resident_gate() { static ResidentGate g; return g; }
int resident_enter(hipStream_t s, int* dev_out)
{
    int dev = 0;
    R2S_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return R2S_ERR_INVALID;
    ResidentGate& g = resident_gate();
    g.mu.lock(); // held until resident_leave: wait + launch + record must not interleave with another thread's
    if (g.recorded[dev]) {
        const hipError_t e = hipStreamWaitEvent(s, g.last[dev], 0);
        if (e != hipSuccess) { g.mu.unlock(); R2S_HIP_TRY(e); }
    }
    *dev_out = dev;
    return R2S_OK;
}
int resident_leave(hipStream_t s, int dev)
{
    ResidentGate& g = resident_gate();
    hipError_t e = hipSuccess;
    if (!g.last[dev]) e = hipEventCreateWithFlags(&g.last[dev], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(g.last[dev], s);
    if (e == hipSuccess) g.recorded[dev] = true;
    g.mu.unlock();
    R2S_HIP_TRY(e);
    return R2S_OK;
}

// Launch the captured env step: chain 0 on the caller's stream, the others on the device's side streams between a fork and a join.
int launch_graphs(R2SPhys* h, int slot, hipStream_t s)
{
    const int chains = h->chains();
    if (chains > 1) {
        if (!h->chain_fork) R2S_HIP_TRY(hipEventCreateWithFlags(&h->chain_fork, hipEventDisableTiming));
        R2S_HIP_TRY(hipEventRecord(h->chain_fork, s));
    }
    for (int c = 1; c < chains; ++c) {
        hipStream_t sc = chain_side_stream(c);
        if (!sc) return R2S_ERR_HIP;
        if (!h->chain_join[c]) R2S_HIP_TRY(hipEventCreateWithFlags(&h->chain_join[c], hipEventDisableTiming));
        R2S_HIP_TRY(hipStreamWaitEvent(sc, h->chain_fork, 0));
        R2S_HIP_TRY(hipGraphLaunch(h->graph_exec[c][slot], sc));
        R2S_HIP_TRY(hipEventRecord(h->chain_join[c], sc));
    }
    R2S_HIP_TRY(hipGraphLaunch(h->graph_exec[0][slot], s));
    for (int c = 1; c < chains; ++c) R2S_HIP_TRY(hipStreamWaitEvent(s, h->chain_join[c], 0));
    return R2S_OK;
}

// Host view of "does any particle have candidates": the count was copied to pinned memory by the last
// update_collision_graph; waiting on its event here costs nothing once that copy has landed.
int resolve_cand_count(R2SPhys* h)
{
    if (h->cand_pending) {
        R2S_HIP_TRY(hipEventSynchronize(h->cand_event));
        h->n_cand = *h->h_cand_count;
        h->cand_pending = false;
    }
    return R2S_OK;
}

// Per-substep rigid transforms (+ world boxes, + rigidity check) of the large dynamic meshes from the current interpolation.
int update_mesh_transforms(R2SPhys* h, hipStream_t s)
{
    if (h->n_xf == 0) return R2S_OK;
    const int tot = h->E * h->prm.num_substeps * h->n_xf;
    R2S_HIP_TRY(hipMemsetAsync(h->d_rigid_err, 0, sizeof(unsigned), s));
    hipLaunchKernelGGL(k_mesh_xf, dim3((tot + 255) / 256), dim3(256), 0, s, h->E, h->prm.num_substeps, h->n_dyn_mesh, h->n_dyn_pts, h->n_xf, h->d_xf_mesh,
                       h->d_xf_ref, h->d_mesh_vert_off, h->d_rest_pts, h->d_xf_rest_box, h->d_interp, h->d_xf, h->d_aabb_dyn, h->d_rigid_err);
    R2S_HIP_TRY(hipMemcpyAsync(h->h_rigid_err, h->d_rigid_err, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    if (!h->rigid_event) R2S_HIP_TRY(hipEventCreateWithFlags(&h->rigid_event, hipEventDisableTiming));
    R2S_HIP_TRY(hipEventRecord(h->rigid_event, s));
    h->rigid_pending = true;
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int grid_sort(R2SPhys* h, hipStream_t s, const uint32_t** keys, const uint32_t** ids, bool fine = false)
{
    const float cell = h->prm.collision_dist * 5.0f;
    const float cell_inv = 1.0f / cell;
    dim3 grid((h->N + TPB - 1) / TPB, h->E);
    if (fine) hipLaunchKernelGGL(k_fine_keys, grid, dim3(TPB), 0, s, h->N, h->E, h->state(h->cur), 1.0f / h->prm.collision_dist, h->d_keys[0], h->d_ids[0]);
    else hipLaunchKernelGGL(k_grid_keys, grid, dim3(TPB), 0, s, h->N, h->E, h->d_inv, h->state(h->cur), cell_inv, h->d_keys[0], h->d_ids[0]);
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
    if (h->coll_cap >= (1 << 19)) { delete h; return R2S_ERR_INVALID; } // list entries pack env (12 bits) | candidate count (19 bits + sign)
    const int N = h->N, E = h->E, S = h->S;
    for (int sp = 0; sp < S; ++sp)
        if (d->init_springs[2 * sp] < 0 || d->init_springs[2 * sp] >= N || d->init_springs[2 * sp + 1] < 0 || d->init_springs[2 * sp + 1] >= N) {
            delete h; return R2S_ERR_INVALID;
        }
    h->h_springs.assign(d->init_springs, d->init_springs + 2 * (size_t)S);
    h->h_rest.assign(d->init_rest_lengths, d->init_rest_lengths + S);
    int rc = R2S_OK;
#define TRY(x) do { rc = (x); if (rc != R2S_OK) { r2s_phys_destroy(h); return rc; } } while (0)

    // ---- Morton order of the particles (env 0's initial positions; the topology is shared by all envs) ----
    // layout: <256,1024> by default; measured against <128,768> on the 1-env rope, the 32-env T block (equal: those are
    // launch-latency bound) and the 32-env pusher scene (256 is 25 % faster).  R2S_LAYOUT=128|256 overrides.
    {
        // (<320,1120> and <384,1280> — layouts whose 1536 / 1280 work items of the 32-env benchmark are ALL resident at once in a
        // 64-VGPR build: no second, half-empty round of workgroups — were measured in round 3: 23.5 / 23.6 us per batched substep
        // against 20.8 for <256,1024> in the same run.  A single round is not faster; the layouts were removed again.)
        // <64,512>: batches small enough to be resident at once (RES_MAX_ITEMS work items of 64 particles) — the layout of the resident
        // stepper (k_steps_resident); the per-substep kernels of the contact flavours run on it as one-wavefront workgroups.
        const int sizes[3] = {256, 128, 64}, caps[3] = {1024, 768, 512};
        int pick = 0;
        if (const char* ev = getenv("R2S_RESIDENT")) h->resident_pref = atoi(ev) != 0;
        if (h->resident_pref && (int64_t)((h->N + 63) / 64) * h->E <= RES_MAX_ITEMS) pick = 2;
        if (const char* ev = getenv("R2S_LAYOUT")) { const int v = atoi(ev); for (int k = 0; k < 3; ++k) if (v == sizes[k]) pick = k; } // tuning knob
        h->pb = sizes[pick]; h->rcap = caps[pick];
        // the remaining tuning knobs are also read here, ONCE per handle (r2s_phys_set_tuning changes them afterwards)
        if (const char* ev = getenv("R2S_CHAINS")) h->chains_override = atoi(ev);
        if (const char* ev = getenv("R2S_MESH_DEFER")) h->force_defer = atoi(ev) != 0;
    }
    const int PB = h->pb, SL = SLICE;
    h->h_perm.resize(N); h->h_inv.resize(N);
    {
        float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
        for (int i = 0; i < N; ++i)
            for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], d->init_vertices[3 * i + k]); hi[k] = std::max(hi[k], d->init_vertices[3 * i + k]); }
        const float ext = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), std::max(hi[2] - lo[2], 1e-9f));
        auto spread = [](uint64_t v) { // 21 bits -> every third bit
            v &= 0x1fffff; v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull;
            v = (v | v << 8) & 0x100f00f00f00f00full; v = (v | v << 4) & 0x10c30c30c30c30c3ull; v = (v | v << 2) & 0x1249249249249249ull;
            return v;
        };
        std::vector<std::pair<uint64_t, int>> code(N);
        for (int i = 0; i < N; ++i) {
            uint64_t c = 0;
            for (int k = 0; k < 3; ++k) {
                const double t = (d->init_vertices[3 * i + k] - lo[k]) / ext;
                c |= spread((uint64_t)std::min(1048575.0, std::max(0.0, t * 1048575.0))) << k;
            }
            code[i] = {c, i};
        }
        std::sort(code.begin(), code.end());
        // Inside a block the order is free (the whole block shares one LDS window): sort by descending
        // spring count so that the 64 particles of a slice have nearly equal degree and the sliced-ELL padding
        // (extra adjacency bytes, LDS reads and flops for nothing) shrinks from ~15 % to a few %.
        std::vector<int> degree(N, 0);
        for (int sp = 0; sp < S; ++sp) { degree[h->h_springs[2 * sp]]++; degree[h->h_springs[2 * sp + 1]]++; }
        for (int b0 = 0; b0 < N; b0 += PB)
            std::stable_sort(code.begin() + b0, code.begin() + std::min(N, b0 + PB),
                             [&](const std::pair<uint64_t, int>& a, const std::pair<uint64_t, int>& b) { return degree[a.second] > degree[b.second]; });
        for (int i = 0; i < N; ++i) { h->h_perm[i] = code[i].second; h->h_inv[code[i].second] = i; }
    }
    TRY(dev_alloc(&h->d_perm, N)); TRY(dev_alloc(&h->d_inv, N));
    TRY(upload(h->d_perm, h->h_perm.data(), N, s)); TRY(upload(h->d_inv, h->h_inv.data(), N, s));

    // ---- sliced-ELL adjacency (gather form), internal indices, neighbours sorted by index ----
    std::vector<std::vector<std::pair<int, int>>> adj(N); // (neighbour, spring id)
    for (int sp = 0; sp < S; ++sp) {
        const int a = h->h_inv[h->h_springs[2 * sp]], b = h->h_inv[h->h_springs[2 * sp + 1]];
        adj[a].push_back({b, sp});
        adj[b].push_back({a, sp});
    }
    for (auto& l : adj) std::sort(l.begin(), l.end());
    h->n_slices = (N + SL - 1) / SL;
    h->nb = (N + PB - 1) / PB;
    h->cb = (h->nb * E + 7) / 8; // (block, env) work items per XCD
    // LDS window of a block = its own records + the most-referenced outside neighbours (halo) up to the layout's
    // capacity; everything else is a "remote" neighbour gathered from global memory (measured 4x slower per slot; not
    // binding for the benchmark objects: largest halo 651 of 768 records in the <256,1024> layout).
    int HALO_CAP = h->rcap - PB;
    if (const char* ev = getenv("R2S_HALO_CAP")) HALO_CAP = std::min(HALO_CAP, std::max(0, atoi(ev))); // tuning knob
    std::vector<int> halo_off(h->nb + 1, 0), halo_ids;
    std::vector<int> slot_of(N, -1);
    std::vector<std::vector<int>> halo_of_block(h->nb);
    h->halo_max = 0;
    for (int b = 0; b < h->nb; ++b) {
        std::vector<std::pair<int, int>> cnt; // (-refs, id)
        {
            std::vector<int> refs;
            for (int i = b * PB; i < std::min(N, (b + 1) * PB); ++i)
                for (auto& nb : adj[i]) if (nb.first / PB != b) refs.push_back(nb.first);
            std::sort(refs.begin(), refs.end());
            for (size_t a = 0; a < refs.size();) {
                size_t e2 = a;
                while (e2 < refs.size() && refs[e2] == refs[a]) ++e2;
                cnt.push_back({-(int)(e2 - a), refs[a]});
                a = e2;
            }
        }
        std::sort(cnt.begin(), cnt.end());
        if ((int)cnt.size() > HALO_CAP) cnt.resize(HALO_CAP);
        std::vector<int>& hl = halo_of_block[b];
        for (auto& c : cnt) hl.push_back(c.second);
        std::sort(hl.begin(), hl.end()); // ascending ids: the staging gather walks memory forwards
        halo_off[b + 1] = halo_off[b] + (int)hl.size();
        halo_ids.insert(halo_ids.end(), hl.begin(), hl.end());
        h->halo_max = std::max(h->halo_max, (int)hl.size());
    }
    // split every particle's neighbours into local (LDS window) and remote lists
    std::vector<std::vector<std::array<int, 3>>> loc(N), rem(N); // {neighbour, spring, lds record}
    for (int b = 0; b < h->nb; ++b) {
        const std::vector<int>& hl = halo_of_block[b];
        for (size_t k = 0; k < hl.size(); ++k) slot_of[hl[k]] = PB + (int)k;
        for (int i = b * PB; i < std::min(N, (b + 1) * PB); ++i)
            for (auto& nb : adj[i]) {
                const int j = nb.first;
                if (j / PB == b) loc[i].push_back({j, nb.second, j % PB});
                else if (slot_of[j] >= 0) loc[i].push_back({j, nb.second, slot_of[j]});
                else rem[i].push_back({j, nb.second, -1});
            }
        for (size_t k = 0; k < hl.size(); ++k) slot_of[hl[k]] = -1;
    }
    // 64-particle layout (a block IS a slice): every particle's list starts with its neighbours inside the block, padded to the slice's
    // largest interior count (whole groups), then the halo neighbours — the resident stepper evaluates the interior groups while the
    // neighbouring blocks' halo records are still on their way (k_steps_resident).  Padding entries carry spring -1 like the ELL's own.
    h->h_slice_int.assign(h->n_slices, 0);
    if (PB == SL)
        for (int b = 0; b < h->nb; ++b) {
            int imax = 0;
            for (int i = b * PB; i < std::min(N, (b + 1) * PB); ++i) {
                std::stable_partition(loc[i].begin(), loc[i].end(), [&](const std::array<int, 3>& a) { return a[2] < PB; });
                int ni = 0;
                for (auto& a : loc[i]) ni += a[2] < PB;
                imax = std::max(imax, ni);
            }
            imax = (imax + GROUP - 1) / GROUP * GROUP;
            h->h_slice_int[b] = imax;
            for (int i = b * PB; i < std::min(N, (b + 1) * PB); ++i) {
                int ni = 0;
                for (auto& a : loc[i]) ni += a[2] < PB;
                loc[i].insert(loc[i].begin() + ni, (size_t)(imax - ni), std::array<int, 3>{i, -1, i % PB});
            }
        }
    // (A bank-aware slot order — every lane takes, per slot, the neighbour whose record falls on the least-used LDS bank pair of
    // its half-wave — was measured in round 1 with fused reads and again in round 2 with plain ds_read_b64: 22.2 vs 22.1 us per
    // substep.  LDS bank conflicts are not what bounds the gather; the adjacency stays in index order.)
    auto build_ell = [&](const std::vector<std::vector<std::array<int, 3>>>& lists, std::vector<int>& off, std::vector<int>& deg,
                         std::vector<int>& a_spring, std::vector<int>& a_nbr, std::vector<int>& a_self, std::vector<int>* a_loc) {
        off.assign(h->n_slices, 0); deg.assign(h->n_slices, 0);
        int total = 0;
        for (int sl = 0; sl < h->n_slices; ++sl) {
            int dmax = 0;
            for (int i = sl * SL; i < std::min(N, (sl + 1) * SL); ++i) dmax = std::max(dmax, (int)lists[i].size());
            dmax = (dmax + GROUP - 1) / GROUP * GROUP; // whole groups, no remainder loop
            off[sl] = total; deg[sl] = dmax;
            total += dmax * SL;
        }
        a_spring.assign(total, -1); a_nbr.assign(total, 0); a_self.assign(total, 0);
        if (a_loc) a_loc->assign(total, 0);
        for (int sl = 0; sl < h->n_slices; ++sl)
            for (int ln = 0; ln < SL; ++ln) {
                const int i = sl * SL + ln;
                const int self = i < N ? i : (i / PB) * PB;
                for (int n = 0; n < deg[sl]; ++n) {
                    const int t = off[sl] + n * SL + ln;
                    a_self[t] = self;
                    if (a_loc) (*a_loc)[t] = self % PB;
                    if (i < N && n < (int)lists[i].size()) {
                        a_nbr[t] = lists[i][n][0]; a_spring[t] = lists[i][n][1];
                        if (a_loc) (*a_loc)[t] = lists[i][n][2];
                    }
                }
            }
        return total;
    };
    h->ell_len = build_ell(loc, h->h_slice_off, h->h_slice_deg, h->h_adj_spring, h->h_adj_nbr, h->h_adj_self, &h->h_adj_loc);
    h->rell_len = build_ell(rem, h->h_rslice_off, h->h_rslice_deg, h->h_radj_spring, h->h_radj_nbr, h->h_radj_self, nullptr);
    TRY(dev_alloc(&h->d_slice_off, h->n_slices)); TRY(dev_alloc(&h->d_slice_deg, h->n_slices));
    TRY(dev_alloc(&h->d_rslice_off, h->n_slices)); TRY(dev_alloc(&h->d_rslice_deg, h->n_slices));
    TRY(dev_alloc(&h->d_adj_idx, h->ell_len)); TRY(dev_alloc(&h->d_adj_k, h->ell_len)); TRY(dev_alloc(&h->d_adj_ir, h->ell_len));
    TRY(dev_alloc(&h->d_radj, h->rell_len));
    TRY(dev_alloc(&h->d_halo_off, halo_off.size())); TRY(dev_alloc(&h->d_halo_ids, halo_ids.size()));
    TRY(upload(h->d_halo_off, halo_off.data(), halo_off.size(), s)); TRY(upload(h->d_halo_ids, halo_ids.data(), halo_ids.size(), s));
    TRY(upload(h->d_slice_off, h->h_slice_off.data(), h->h_slice_off.size(), s)); TRY(upload(h->d_slice_deg, h->h_slice_deg.data(), h->h_slice_deg.size(), s));
    TRY(dev_alloc(&h->d_slice_int, h->n_slices)); TRY(upload(h->d_slice_int, h->h_slice_int.data(), h->h_slice_int.size(), s));
    TRY(upload(h->d_rslice_off, h->h_rslice_off.data(), h->h_rslice_off.size(), s)); TRY(upload(h->d_rslice_deg, h->h_rslice_deg.data(), h->h_rslice_deg.size(), s));
    TRY(dev_alloc(&h->d_masses, N)); TRY(dev_alloc(&h->d_masks, N));
    {
        std::vector<float> zero_logy(std::max(S, 1), 0.f);
        TRY(upload_stiffness(h, S > 0 ? d->init_spring_Y : zero_logy.data(), s));
    }
    {
        std::vector<float> m(N);
        std::vector<int> masks(N);
        for (int i = 0; i < N; ++i) {
            const int u = h->h_perm[i];
            m[i] = d->init_masses[u];
            masks[i] = d->init_collision_mask ? d->init_collision_mask[u] : u;
        }
        TRY(upload(h->d_masses, m.data(), N, s));
        TRY(upload(h->d_masks, masks.data(), N, s));
    }
    // ---- state ----
    for (int b = 0; b < 2; ++b) {
        TRY(dev_alloc(&h->xv[b], (size_t)E * N * 3));
        R2S_HIP_TRY(hipMemsetAsync(h->xv[b], 0, sizeof(v2f) * (size_t)E * N * 3, s));
    }
    {
        const size_t n = (size_t)E * N;
        std::vector<float> pk(6 * n, 0.f); // three planes of (float, float)
        for (int e = 0; e < E; ++e)
            for (int u = 0; u < N; ++u) {
                const size_t src = ((size_t)e * N + u) * 3, dst = (size_t)e * N + h->h_inv[u];
                const float* x = d->init_vertices + src;
                const float zero[3] = {0.f, 0.f, 0.f};
                const float* v = d->init_velocities ? d->init_velocities + src : zero;
                auto at = [&](int k) { // the device's st_at
#ifdef R2S_STATE_PLANES
                    return 2 * ((size_t)k * n + dst);
#else
                    return 2 * (3 * dst + (size_t)k);
#endif
                };
                pk[at(0)] = x[0]; pk[at(0) + 1] = x[1];
                pk[at(1)] = x[2]; pk[at(1) + 1] = v[2];
                pk[at(2)] = v[0]; pk[at(2) + 1] = v[1];
            }
        TRY(upload((float*)h->xv[0], pk.data(), pk.size(), s));
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
        // ---- mesh query acceleration (the role of wp.Mesh's BVH, :673/:899) ----
        // small mesh (<= LARGE_FACES faces): one cluster, brute force; large mesh: faces Morton-sorted by centroid and cut
        // into clusters of 64 with rest-frame boxes; must be a closed manifold (sign by pseudonormal) and, if dynamic,
        // move rigidly (the per-substep transform is recovered from three reference vertices).
        constexpr int LARGE_FACES = 256, CL = 64;
        std::vector<int> mesh_kind(h->n_mesh, 0), mesh_xf(h->n_mesh, -1), xf_mesh, xf_ref;
        std::vector<float> xf_rest_box;
        std::vector<int> stored(3 * (size_t)h->nF), face_orig(h->nF), face_mesh(h->nF), cl_f0, cl_f1, cl_mesh;
        std::vector<float> cl_box, pnorm(21 * (size_t)h->nF, 0.f);
        const float* V = d->mesh_vertices;
        auto vtx = [&](int gv, int k) { return (double)V[3 * (size_t)gv + k]; };
        for (int m = 0; m < h->n_mesh; ++m) {
            const int f0 = foff[m], nf = foff[m + 1] - foff[m];
            std::vector<int> order(nf);
            for (int k = 0; k < nf; ++k) order[k] = f0 + k;
            // Weld coincident vertices (STL exports repeat every corner), then test for a closed, consistently oriented
            // manifold: every undirected edge is used by exactly two faces, once in each direction.  mesh_kind bits:
            // 1 = large (> 256 faces: clusters + wave-cooperative queries), 2 = open / not a manifold (sign from the exact
            // winding number like the reference, :322-324, and no "outside the box is outside the mesh" early-out).
            std::vector<int> canon(voff[m + 1] - voff[m]);
            {
                std::vector<std::array<float, 3>> pos(canon.size());
                std::vector<int> ord(canon.size());
                for (size_t v = 0; v < canon.size(); ++v) { ord[v] = (int)v; for (int k = 0; k < 3; ++k) pos[v][k] = V[3 * (size_t)(voff[m] + v) + k]; }
                std::sort(ord.begin(), ord.end(), [&](int a, int b) { return pos[a] != pos[b] ? pos[a] < pos[b] : a < b; });
                for (size_t q = 0; q < ord.size(); ++q) canon[ord[q]] = (q > 0 && pos[ord[q]] == pos[ord[q - 1]]) ? canon[ord[q - 1]] : ord[q];
            }
            auto cv = [&](int gv) { return voff[m] + canon[gv - voff[m]]; }; // welded (canonical) global vertex id
            std::vector<std::array<long long, 3>> edges; // (min v, max v, +-(face+1)) on welded ids
            bool manifold = nf > 0;
            for (int k = 0; k < nf; ++k) {
                const int tri[3] = {cv(faces[3 * (f0 + k)]), cv(faces[3 * (f0 + k) + 1]), cv(faces[3 * (f0 + k) + 2])};
                if (tri[0] == tri[1] || tri[1] == tri[2] || tri[2] == tri[0]) manifold = false; // degenerate after welding
                for (int c = 0; c < 3; ++c) {
                    const int p0 = tri[c], p1 = tri[(c + 1) % 3];
                    edges.push_back({std::min(p0, p1), std::max(p0, p1), p0 < p1 ? (long long)(k + 1) : -(long long)(k + 1)});
                }
            }
            std::sort(edges.begin(), edges.end());
            manifold = manifold && edges.size() % 2 == 0;
            for (size_t q = 0; manifold && q + 1 < edges.size(); q += 2)
                manifold = edges[q][0] == edges[q + 1][0] && edges[q][1] == edges[q + 1][1] && (edges[q][2] < 0) != (edges[q + 1][2] < 0) &&
                           (q + 2 >= edges.size() || edges[q + 2][0] != edges[q][0] || edges[q + 2][1] != edges[q][1]);
            if (!manifold) mesh_kind[m] |= 2;
            if (nf > LARGE_FACES) {
                mesh_kind[m] |= 1;
                double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
                for (int v = voff[m]; v < voff[m + 1]; ++v) for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], vtx(v, k)); hi[k] = std::max(hi[k], vtx(v, k)); }
                const double ext = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-12});
                auto spread = [](uint64_t v) { v &= 0x1fffff; v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull;
                                               v = (v | v << 8) & 0x100f00f00f00f00full; v = (v | v << 4) & 0x10c30c30c30c30c3ull; v = (v | v << 2) & 0x1249249249249249ull; return v; };
                std::vector<std::pair<uint64_t, int>> code(nf);
                for (int k = 0; k < nf; ++k) {
                    uint64_t c = 0;
                    for (int ax = 0; ax < 3; ++ax) {
                        const double cen = (vtx(faces[3 * (f0 + k)], ax) + vtx(faces[3 * (f0 + k) + 1], ax) + vtx(faces[3 * (f0 + k) + 2], ax)) / 3.0;
                        c |= spread((uint64_t)std::min(1048575.0, std::max(0.0, (cen - lo[ax]) / ext * 1048575.0))) << ax;
                    }
                    code[k] = {c, f0 + k};
                }
                std::sort(code.begin(), code.end());
                for (int k = 0; k < nf; ++k) order[k] = code[k].second;
                // pseudonormals (Baerentzen & Aanaes 2005) of a closed manifold, in the rest frame, on welded vertices
                std::vector<std::array<double, 3>> fn(nf);
                std::vector<std::array<double, 3>> vn(voff[m + 1] - voff[m], {0, 0, 0});
                auto sub3 = [&](int a, int b, double* o) { for (int k = 0; k < 3; ++k) o[k] = vtx(a, k) - vtx(b, k); };
                for (int k = 0; manifold && k < nf; ++k) {
                    const int ia = cv(faces[3 * (f0 + k)]), ib = cv(faces[3 * (f0 + k) + 1]), ic = cv(faces[3 * (f0 + k) + 2]);
                    double ab[3], ac[3]; sub3(ib, ia, ab); sub3(ic, ia, ac);
                    double n[3] = {ab[1] * ac[2] - ab[2] * ac[1], ab[2] * ac[0] - ab[0] * ac[2], ab[0] * ac[1] - ab[1] * ac[0]};
                    const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                    for (int q = 0; q < 3; ++q) fn[k][q] = l > 0 ? n[q] / l : 0.0;
                    const int tri[3] = {ia, ib, ic};
                    for (int c = 0; c < 3; ++c) {
                        const int p0 = tri[c], p1 = tri[(c + 1) % 3], p2 = tri[(c + 2) % 3];
                        double e1[3], e2[3]; sub3(p1, p0, e1); sub3(p2, p0, e2);
                        const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
                        const double cs = (l1 > 0 && l2 > 0) ? (e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2]) / (l1 * l2) : 1.0;
                        const double ang = std::acos(std::min(1.0, std::max(-1.0, cs)));
                        for (int q = 0; q < 3; ++q) vn[p0 - voff[m]][q] += ang * fn[k][q];
                    }
                }
                auto edge_normal = [&](int p0, int p1, double* o) {
                    std::array<long long, 3> key = {std::min(p0, p1), std::max(p0, p1), LLONG_MIN};
                    auto it = std::lower_bound(edges.begin(), edges.end(), key);
                    for (int q = 0; q < 3; ++q) o[q] = 0;
                    for (int r = 0; r < 2; ++r, ++it) { const int ff = (int)std::llabs((*it)[2]) - 1; for (int q = 0; q < 3; ++q) o[q] += fn[ff][q]; }
                };
                for (int k = 0; manifold && k < nf; ++k) {
                    const int of = order[k], kk = of - f0;            // original face, index inside the mesh
                    const size_t st = (size_t)(f0 + k) * 21;          // stored slot
                    const int tri[3] = {cv(faces[3 * of]), cv(faces[3 * of + 1]), cv(faces[3 * of + 2])};
                    for (int q = 0; q < 3; ++q) pnorm[st + q] = (float)fn[kk][q];
                    for (int c = 0; c < 3; ++c) for (int q = 0; q < 3; ++q) pnorm[st + 3 * (1 + c) + q] = (float)vn[tri[c] - voff[m]][q];
                    for (int c = 0; c < 3; ++c) { double en[3]; edge_normal(tri[c], tri[(c + 1) % 3], en); for (int q = 0; q < 3; ++q) pnorm[st + 3 * (4 + c) + q] = (float)en[q]; }
                }
                if (m < h->n_dyn_mesh) { // reference vertices of the rigid transform: first, farthest from it, max triangle area
                    const int i0 = voff[m];
                    int i1 = i0, i2 = i0; double bd = -1, ba = -1;
                    for (int v = voff[m]; v < voff[m + 1]; ++v) { double dd[3]; sub3(v, i0, dd); const double l = dd[0] * dd[0] + dd[1] * dd[1] + dd[2] * dd[2]; if (l > bd) { bd = l; i1 = v; } }
                    for (int v = voff[m]; v < voff[m + 1]; ++v) {
                        double a1[3], a2[3]; sub3(i1, i0, a1); sub3(v, i0, a2);
                        const double cx = a1[1] * a2[2] - a1[2] * a2[1], cy = a1[2] * a2[0] - a1[0] * a2[2], cz = a1[0] * a2[1] - a1[1] * a2[0];
                        const double ar = cx * cx + cy * cy + cz * cz; if (ar > ba) { ba = ar; i2 = v; }
                    }
                    mesh_xf[m] = (int)xf_mesh.size();
                    xf_mesh.push_back(m); xf_ref.push_back(i0); xf_ref.push_back(i1); xf_ref.push_back(i2);
                    for (int k = 0; k < 3; ++k) xf_rest_box.push_back((float)lo[k]);
                    for (int k = 0; k < 3; ++k) xf_rest_box.push_back((float)hi[k]);
                }
            }
            for (int k = 0; k < nf; ++k) {
                const int of = order[k], st = f0 + k;
                for (int q = 0; q < 3; ++q) stored[3 * st + q] = faces[3 * of + q];
                face_orig[st] = of; face_mesh[st] = m;
            }
            const int step = (mesh_kind[m] & 1) ? CL : std::max(nf, 1);
            for (int k = 0; k < nf; k += step) {
                cl_f0.push_back(f0 + k); cl_f1.push_back(f0 + std::min(nf, k + step)); cl_mesh.push_back(m);
                float bb[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f};
                for (int st = f0 + k; st < f0 + std::min(nf, k + step); ++st)
                    for (int c = 0; c < 3; ++c) for (int q = 0; q < 3; ++q) { const float x = V[3 * (size_t)stored[3 * st + c] + q]; bb[q] = std::min(bb[q], x); bb[3 + q] = std::max(bb[3 + q], x); }
                for (int q = 0; q < 6; ++q) cl_box.push_back(bb[q]);
            }
        }
        h->n_cl = (int)cl_f0.size(); h->n_xf = (int)xf_mesh.size();
        if (h->n_xf > 62) { // the cluster record holds the transform slot in 6 bits
            r2s::set_last_error_msg("more than 62 large dynamic collision meshes (unsupported)");
            r2s_phys_destroy(h);
            return R2S_ERR_INVALID;
        }
        h->h_mesh_kind = mesh_kind; h->h_voff = voff; h->h_foff = foff; h->h_xf_mesh = xf_mesh; h->h_xf_ref = xf_ref;
        for (int m = 0; m < h->n_mesh; ++m) h->any_large = h->any_large || (mesh_kind[m] & 1);
        TRY(dev_alloc(&h->d_face_orig, h->nF));
        TRY(dev_alloc(&h->d_cl_box, cl_box.size())); TRY(dev_alloc(&h->d_mesh_kind, h->n_mesh)); TRY(dev_alloc(&h->d_mesh_xf, h->n_mesh));
        TRY(dev_alloc(&h->d_rest_pts, 3 * (size_t)h->nV)); TRY(dev_alloc(&h->d_pnorm, pnorm.size()));
        TRY(dev_alloc(&h->d_xf_mesh, xf_mesh.size())); TRY(dev_alloc(&h->d_xf_ref, xf_ref.size())); TRY(dev_alloc(&h->d_xf_rest_box, xf_rest_box.size()));
        TRY(dev_alloc(&h->d_xf, (size_t)E * n_sub * std::max(1, h->n_xf) * 12)); TRY(dev_alloc(&h->d_rigid_err, 4));
        R2S_HIP_TRY(hipMemsetAsync(h->d_rigid_err, 0, 16, s));
        R2S_HIP_TRY(hipHostMalloc((void**)&h->h_rigid_err, 64, hipHostMallocDefault));
        *h->h_rigid_err = 0;
        TRY(upload(h->d_face_orig, face_orig.data(), face_orig.size(), s));
        {
            std::vector<float> box_t(cl_box.size());
            for (int c = 0; c < h->n_cl; ++c)
                for (int k = 0; k < 6; ++k) box_t[(size_t)k * h->n_cl + c] = cl_box[(size_t)c * 6 + k];
            TRY(upload(h->d_cl_box, box_t.data(), box_t.size(), s));
        }
        TRY(upload(h->d_mesh_kind, mesh_kind.data(), mesh_kind.size(), s));
        TRY(upload(h->d_mesh_xf, mesh_xf.data(), mesh_xf.size(), s)); TRY(upload(h->d_rest_pts, d->mesh_vertices, 3 * (size_t)h->nV, s));
        TRY(upload(h->d_pnorm, pnorm.data(), pnorm.size(), s)); TRY(upload(h->d_xf_mesh, xf_mesh.data(), xf_mesh.size(), s));
        {
            std::vector<float> tri(9 * 64 * (size_t)((h->nF + 63) / 64), 0.f);
            for (int f = 0; f < h->nF; ++f)
                for (int c = 0; c < 3; ++c)
                    for (int k = 0; k < 3; ++k) tri[((size_t)(f >> 6) * 9 + c * 3 + k) * 64 + (f & 63)] = d->mesh_vertices[3 * (size_t)stored[3 * f + c] + k];
            std::vector<int4> info(h->n_cl);
            for (int c = 0; c < h->n_cl; ++c) info[c] = make_int4(cl_mesh[c], mesh_kind[cl_mesh[c]] | ((mesh_xf[cl_mesh[c]] + 1) << 2) | ((cl_f1[c] - cl_f0[c]) << 8), mesh_xf[cl_mesh[c]], cl_f0[c]);
            for (int c = 0; c < h->n_cl && h->lm[1] == 0; ++c) // the first large mesh: its cluster records as arithmetic (PhysDev::lm_*), checked against the table
                if (mesh_kind[cl_mesh[c]] & 1) {
                    const int m = cl_mesh[c];
                    int nc = 0;
                    while (c + nc < h->n_cl && cl_mesh[c + nc] == m) ++nc;
                    int lm[7] = {c, nc, cl_f0[c], cl_f1[c + nc - 1], mesh_kind[m] | ((mesh_xf[m] + 1) << 2), m, mesh_xf[m]};
                    bool ok = true;
                    for (int k = 0; k < nc && ok; ++k) {
                        const int f0 = lm[2] + 64 * k;
                        const int4 want = info[c + k];
                        ok = want.x == m && want.y == (lm[4] | (std::min(64, lm[3] - f0) << 8)) && want.z == mesh_xf[m] && want.w == f0;
                    }
                    if (ok) std::copy(lm, lm + 7, h->lm);
                    break;
                }
            TRY(dev_alloc(&h->d_tri_rest, tri.size())); TRY(upload(h->d_tri_rest, tri.data(), tri.size(), s));
            TRY(dev_alloc(&h->d_cl_info, info.size())); TRY(upload(h->d_cl_info, info.data(), info.size(), s));
            // super-clusters: runs of eight consecutive clusters of one large mesh (Morton order keeps them compact)
            std::vector<int4> sup;
            std::vector<float> sbox;
            std::vector<int> small;
            for (int c = 0; c < h->n_cl;) {
                const int m = cl_mesh[c];
                if (!(mesh_kind[m] & 1)) { small.push_back(m); ++c; continue; }
                int n = 1;
                while (n < 8 && c + n < h->n_cl && cl_mesh[c + n] == m) ++n;
                float bb[6] = {3e38f, 3e38f, 3e38f, -3e38f, -3e38f, -3e38f};
                for (int k = c; k < c + n; ++k)
                    for (int q = 0; q < 3; ++q) { bb[q] = std::min(bb[q], cl_box[(size_t)k * 6 + q]); bb[3 + q] = std::max(bb[3 + q], cl_box[(size_t)k * 6 + 3 + q]); }
                sup.push_back(make_int4(c, n, mesh_xf[m], mesh_kind[m]));
                for (int q = 0; q < 6; ++q) sbox.push_back(bb[q]);
                c += n;
            }
            h->n_sup = (int)sup.size(); h->n_small = (int)small.size();
            std::vector<float> sbox_t(sbox.size());
            for (int k = 0; k < h->n_sup; ++k)
                for (int q = 0; q < 6; ++q) sbox_t[(size_t)q * h->n_sup + k] = sbox[(size_t)k * 6 + q];
            TRY(dev_alloc(&h->d_sup_info, std::max<size_t>(sup.size(), 1))); TRY(upload(h->d_sup_info, sup.data(), sup.size(), s));
            TRY(dev_alloc(&h->d_sup_box, std::max<size_t>(sbox_t.size(), 1))); TRY(upload(h->d_sup_box, sbox_t.data(), sbox_t.size(), s));
            TRY(dev_alloc(&h->d_small_mesh, std::max<size_t>(small.size(), 1))); TRY(upload(h->d_small_mesh, small.data(), small.size(), s));
        }
        TRY(upload(h->d_xf_ref, xf_ref.data(), xf_ref.size(), s)); TRY(upload(h->d_xf_rest_box, xf_rest_box.data(), xf_rest_box.size(), s));
        faces = stored; // the device face table is in stored (cluster) order
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
        TRY(dev_alloc(&h->d_hit_cnt, (size_t)E));
        R2S_HIP_TRY(hipMemsetAsync(h->d_hit_cnt, 0, sizeof(int) * (size_t)E, s));
        if (h->n_dyn_mesh > 0) {
            const int tot = E * n_sub * h->n_dyn_mesh;
            hipLaunchKernelGGL(k_mesh_aabb_dyn, dim3((tot + 255) / 256), dim3(256), 0, s, E, n_sub, h->n_dyn_mesh, h->n_dyn_pts, h->d_mesh_vert_off, h->d_mesh_kind,
                               h->d_interp, h->d_aabb_dyn);
            TRY(update_mesh_transforms(h, s));
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
    if (h->nF > 0) { // deferred mesh queries
        h->mesh_cap = E * N; // a particle is listed at most once per substep: the list cannot overflow
        // (everything a substep hands to its finishing code exists twice, by substep parity: PhysDev::par_stride)
        TRY(dev_alloc(&h->d_mesh_list, (size_t)2 * h->mesh_cap));
        R2S_HIP_TRY(hipMemsetAsync(h->d_mesh_list, 0, sizeof(int2) * (size_t)2 * h->mesh_cap, s));
        TRY(dev_alloc(&h->d_mesh_cnt, (size_t)8 * (h->prm.num_substeps + 1)));
        TRY(dev_alloc(&h->d_vdef, (size_t)2 * E * N));
        if (h->any_large) {
            TRY(dev_alloc(&h->d_mesh_rec, (size_t)4 * E * N));
            R2S_HIP_TRY(hipMemsetAsync(h->d_mesh_rec, 0, sizeof(int4) * (size_t)4 * E * N, s));
            if (h->n_cl < 4094 && getenv("R2S_NO_MQ_HINT") == nullptr) { // the hint rides in 12 bits of a record word
                TRY(dev_alloc(&h->d_mq_hint, (size_t)E * N));
                R2S_HIP_TRY(hipMemsetAsync(h->d_mq_hint, 0xFF, sizeof(int) * (size_t)E * N, s)); // -1: no hint
            }
            TRY(dev_alloc(&h->d_rec_cnt, (size_t)E * h->prm.num_substeps));
            R2S_HIP_TRY(hipMemsetAsync(h->d_rec_cnt, 0, sizeof(int) * (size_t)E * h->prm.num_substeps, s));
        }
        if (h->prm.self_collision) { TRY(dev_alloc(&h->d_cand_mark, (size_t)2 * E * N)); R2S_HIP_TRY(hipMemsetAsync(h->d_cand_mark, 0, sizeof(int) * (size_t)2 * E * N, s)); }
        // the finishers at the head of the next launch (k_substep_pf): large-batch layout only; a result line per particle
        if (const char* ev = getenv("R2S_PF")) h->pf_pref = atoi(ev) != 0;
        h->pf_ok = h->pb == 256 && (uint64_t)E * N * PF_LINE < 0x7fffffffull;
        if (h->pf_ok) {
            TRY(dev_alloc((char**)&h->d_pf_res, (size_t)PF_LINE * E * N));
            R2S_HIP_TRY(hipMemsetAsync(h->d_pf_res, 0, (size_t)PF_LINE * E * N, s));
        }
        R2S_HIP_TRY(hipMemsetAsync(h->d_mesh_cnt, 0, sizeof(int) * 8 * (size_t)(h->prm.num_substeps + 1), s));
    }
    if (h->nF > 0 || h->prm.self_collision) TRY(dev_alloc(&h->d_xbc, (size_t)2 * E * N));
    TRY(dev_alloc(&h->d_mesh_total, 16)); // [4..15]: where the first fault of the resident stepper happened (diagnostics); [0] particles near a mesh in the last step, [1] sticky fault word (PhysDev::fault), [2] a mesh query was needed, [3] a resident launch ran out of server pairs
    R2S_HIP_TRY(hipMemsetAsync(h->d_mesh_total, 0, sizeof(int) * 16, s));
    R2S_HIP_TRY(hipHostMalloc((void**)&h->h_ring, sizeof(int) * 16 * R2SPhys::RING, hipHostMallocDefault));
    memset(h->h_ring, 0, sizeof(int) * 16 * R2SPhys::RING);
    for (int k = 0; k < R2SPhys::RING; ++k) R2S_HIP_TRY(hipEventCreateWithFlags(&h->ring_ev[k], hipEventDisableTiming));
    {
        // the resident launch needs: the 64-particle layout, every neighbour inside the block's window (a remote neighbour would be read
        // from the state arrays, which a resident launch only touches at its two ends), no large mesh (its queries are workgroup-cooperative
        // and always deferred), and all work items on the chip at once
        bool remote = false;
        for (int t = 0; t < h->rell_len && !remote; ++t) remote = h->h_radj_spring[t] >= 0;
        bool fits = true; // a wavefront keeps at most RES_NG interior and RES_NG halo groups of a slice in registers (every eighth group each)
        for (int sl = 0; sl < h->n_slices && h->pb == 64; ++sl)
            fits = fits && h->h_slice_int[sl] <= (RES_THREADS / 64) * RES_NG * GROUP && h->h_slice_deg[sl] - h->h_slice_int[sl] <= (RES_THREADS / 64) * RES_NG * GROUP;
        h->split_ok = h->pb == 64 && !remote && fits;
        int dev = 0, n_cu = 0; // one workgroup per CU, all resident at once: the device decides how many that is
        R2S_HIP_TRY(hipGetDevice(&dev));
        R2S_HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        // Residency is ASKED for, not assumed (round 5): the launch needs one 512-thread workgroup of k_steps_resident per CU at once — blocks
        // and servers alike, they are the same kernel.  The occupancy query answers for this build's registers and LDS on this device; a
        // compiler that pushes the kernel over its budget, or a CU budget smaller than the chip (R2S_RES_CU_BUDGET: a CU-masked partition, a
        // shared device; also what the test uses), turns into the per-substep flavour at create time instead of poll-limit faults later.
        // (The query over-reports by at most one block per CU for SGPR-heavy 256-thread kernels on ROCm 7.2; this launch needs ONE
        // 512-thread block per CU and the kernel is register-bound at two wavefronts per SIMD: the answer is exact where it matters, 0 vs >= 1.)
        if (const char* ev = getenv("R2S_RES_CU_BUDGET")) n_cu = std::max(0, std::min(n_cu, atoi(ev)));
        int occ_min = 1 << 30;
        {
            int occ = 0;
            const void* kernels[2] = {(const void*)k_steps_resident<512, false, 0>, (const void*)k_steps_resident<512, false, 1>};
            for (int k = 0; k < (h->nF > 0 ? 2 : 1); ++k) {
                R2S_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernels[k], RES_THREADS, 0));
                occ_min = std::min(occ_min, occ);
            }
        }
        h->resident_ok = h->split_ok && !h->any_large && (int64_t)h->nb * E <= std::min(RES_MAX_ITEMS, n_cu) && occ_min >= 1;
        if (h->split_ok && !h->any_large && !h->resident_ok && h->resident_pref) {
            char buf[256];
            snprintf(buf, sizeof buf, "resident stepper not used: %lld work items, %d CUs available to it, %d workgroup(s) of the kernel fit a CU at once — "
                     "the env step runs as one launch per substep", (long long)h->nb * E, n_cu, occ_min == (1 << 30) ? -1 : occ_min);
            r2s::set_last_error_msg(buf); // informational: r2s_phys_create still returns R2S_OK
        }
        h->n_cu = n_cu;
        if (h->resident_ok) {
            const size_t xn = ((size_t)N + 7) & ~(size_t)7;
            TRY(dev_alloc((char**)&h->d_xch, (size_t)96 * E * xn));
            R2S_HIP_TRY(hipMemsetAsync(h->d_xch, 0, (size_t)96 * E * xn, s));
            // query servers ride in the launch when every mesh is small enough for k_contact_finish<3>'s code (triangles in registers, one per
            // lane in two wavefronts) and the blocks leave CUs free; R2S_RES_SERVERS=0: queries in place / per-substep flavour as in round 3
            bool pref = true;
            if (const char* ev = getenv("R2S_RES_SERVERS")) pref = atoi(ev) != 0;
            if (const char* ev = getenv("R2S_RES_SPIN_LIMIT")) h->spin_limit = (unsigned)std::max(1024, atoi(ev));
            if (const char* ev = getenv("R2S_RES_SRV_OWN")) h->srv_own = atoi(ev) != 0;
            if (const char* ev = getenv("R2S_RES_SRV_QUAD")) h->srv_quad = atoi(ev) != 0;
            if (const char* ev = getenv("R2S_RES_SRV_WG")) h->srv_wg_cap = std::max(1, std::min(atoi(ev), SRV_MAX_SLOTS / 4));
            h->srv_ok = pref && h->nF > 0 && h->nF <= 128 && n_cu - 8 * (int)(((int64_t)h->nb * E + 7) / 8) >= SRV_MIN_WG;
            if (h->srv_ok) {
                TRY(dev_alloc((char**)&h->d_srv_claim, (size_t)SRV_CLAIM_BYTES)); // claims | 2 control words | per pair and wavefront: 16 B of state for fault reports
                TRY(dev_alloc((char**)&h->d_srv_rr, (size_t)SRV_REC * E * N));
                R2S_HIP_TRY(hipMemsetAsync(h->d_srv_claim, 0, (size_t)SRV_CLAIM_BYTES, s));
                R2S_HIP_TRY(hipMemsetAsync(h->d_srv_rr, 0, (size_t)SRV_REC * E * N, s));
            }
        }
    }
    if (h->prm.self_collision) {
        TRY(dev_alloc(&h->d_vbc, (size_t)2 * E * N));
        TRY(dev_alloc(&h->d_cand_list, (size_t)E * N));
        R2S_HIP_TRY(hipMemsetAsync(h->d_cand_list, 0, sizeof(int2) * (size_t)E * N, s));
        TRY(dev_alloc(&h->d_cand_count, (size_t)4 + E)); // [0] all particles with candidates, [4 + e] those of environment e
        R2S_HIP_TRY(hipMemsetAsync(h->d_cand_count, 0, sizeof(int) * ((size_t)4 + E), s));
        R2S_HIP_TRY(hipHostMalloc((void**)&h->h_cand_count, 64, hipHostMallocDefault));
        *h->h_cand_count = 0;
        R2S_HIP_TRY(hipEventCreateWithFlags(&h->cand_event, hipEventDisableTiming));
        h->words = (N + 31) / 32;
        TRY(dev_alloc(&h->d_coll_idx, (size_t)E * N * h->coll_cap));
        TRY(dev_alloc(&h->d_bits, (size_t)E * N * h->words));
        for (int b = 0; b < 2; ++b) { TRY(dev_alloc(&h->d_keys[b], (size_t)E * N)); TRY(dev_alloc(&h->d_ids[b], (size_t)E * N)); }
        rocprim::double_buffer<uint32_t> dk((uint32_t*)nullptr, (uint32_t*)nullptr), dv((uint32_t*)nullptr, (uint32_t*)nullptr);
        R2S_HIP_TRY(rocprim::radix_sort_pairs(nullptr, h->sort_bytes, dk, dv, (size_t)E * N, 0u, 32u, s));
        TRY(dev_alloc(&h->d_sort_tmp, h->sort_bytes));
        if (E <= 256) { // 16 MiB per environment
            TRY(dev_alloc(&h->d_cell_tab, (size_t)E << GRID_CELL_BITS));
            R2S_HIP_TRY(hipMemsetAsync(h->d_cell_tab, 0, sizeof(int2) * ((size_t)E << GRID_CELL_BITS), s));
            TRY(dev_alloc(&h->d_cell_xs, (size_t)E * N));
        }
        TRY(r2s_phys_create_resting_case(h, stream_));
    }
    R2S_HIP_TRY(hipStreamSynchronize(s));
    // Capture every flavour the env step can take — {no candidates, candidates} x {nothing near a mesh, deferred mesh queries}
    // x both parities of the state buffer (667 substeps is odd) — now: a flavour switch in the middle of a rollout (first
    // contact, first candidates) must not pay ~5 ms of capture.  All finishing kernels have fixed grid-stride grids.
    for (int defer = h->any_large ? 1 : 0; defer <= (h->nF > 0 ? 1 : 0); ++defer) // a scene with a large mesh always defers
        for (int variant = 0; variant <= (h->prm.self_collision ? 1 : 0); ++variant)
            for (int par = 0; par < ((h->prm.num_substeps & 1) ? 2 : 1); ++par) {
                h->mesh_defer = defer;
                TRY(capture_graph(h, variant, par));
            }
    h->mesh_defer = h->any_large ? 1 : 0;
#undef TRY
    *out = h;
    return R2S_OK;
}

void r2s_phys_destroy(R2SPhys* h)
{
    if (!h) return;
    (void)hipDeviceSynchronize();
    drop_graph(h);
    void* ptrs[] = {h->xv[0], h->xv[1], h->d_slice_off, h->d_slice_deg, h->d_slice_int, h->d_rslice_off, h->d_rslice_deg, h->d_adj_idx, h->d_adj_k, h->d_adj_ir, h->d_radj, h->d_halo_off, h->d_halo_ids, h->d_perm, h->d_inv, h->d_num_user, h->d_idx_user, h->d_masses, h->d_masks,
                    h->d_coll_num, h->d_coll_idx, h->d_max_count, h->d_vbc, h->d_xbc, h->d_pf_res, h->d_mesh_list, h->d_mesh_cnt, h->d_vdef, h->d_mesh_rec, h->d_mq_hint, h->d_rec_cnt, h->d_cand_mark, h->d_mesh_total, h->d_cand_list, h->d_cand_count, h->d_bits, h->d_keys[0], h->d_keys[1], h->d_ids[0], h->d_ids[1], h->d_sort_tmp, h->d_cell_tab, h->d_cell_xs,
                    h->d_faces, h->d_face_orig, h->d_cl_box, h->d_mesh_kind, h->d_mesh_xf, h->d_xf_mesh, h->d_xf_ref,
                    h->d_xf, h->d_rest_pts, h->d_pnorm, h->d_tri_rest, h->d_cl_info, h->d_sup_info, h->d_sup_box, h->d_small_mesh, h->d_xf_rest_box, h->d_rigid_err, h->d_mesh_map, h->d_face_map, h->d_mesh_face_off, h->d_mesh_vert_off, h->d_mesh_pts, h->d_interp, h->d_center,
                    h->d_dyn_vel, h->d_dyn_omega, h->d_aabb_dyn, h->d_aabb_static, h->d_coll_forces,
                    h->d_eef_table, h->d_eef_open, h->d_eef_grasped, h->d_eef_has, h->d_eef_need, h->d_eef_rel0, h->d_eef_delta, h->d_hit_cnt, h->d_xch,
                    h->d_srv_claim, h->d_srv_rr};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (h->h_cand_count) (void)hipHostFree(h->h_cand_count);
    if (h->h_ring) (void)hipHostFree(h->h_ring);
    for (int k = 0; k < R2SPhys::RING; ++k) if (h->ring_ev[k]) (void)hipEventDestroy(h->ring_ev[k]);
    if (h->rigid_event) (void)hipEventDestroy(h->rigid_event);
    if (h->h_rigid_err) (void)hipHostFree(h->h_rigid_err);
    if (h->cand_event) (void)hipEventDestroy(h->cand_event);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->chain_fork) (void)hipEventDestroy(h->chain_fork);
    for (int c = 0; c < R2SPhys::MAX_CHAINS; ++c)
        if (h->chain_join[c]) (void)hipEventDestroy(h->chain_join[c]);
    delete h;
}

int r2s_phys_set_state(R2SPhys* h, const float* x, const float* v, r2s_stream_t stream_)
{
    if (!h) return R2S_ERR_INVALID;
    hipLaunchKernelGGL(k_pack, dim3((h->N + 255) / 256, h->E), dim3(256), 0, (hipStream_t)stream_, h->N, h->E, h->d_inv, x, v, h->state(h->cur), (const int*)nullptr);
    R2S_HIP_TRY(hipGetLastError());
    // A fault word (r2s_phys_step) says "the state is invalid": a state set by the caller makes the handle usable again.
    if (h->d_mesh_total) {
        R2S_HIP_TRY(hipMemsetAsync(h->d_mesh_total + 1, 0, sizeof(int), (hipStream_t)stream_));
        R2S_HIP_TRY(hipMemsetAsync(h->d_mesh_total + 4, 0, 12 * sizeof(int), (hipStream_t)stream_)); // ... and where the fault happened
        for (int k = 0; k < R2SPhys::RING; ++k) { h->ring_stale_fault[k] = true; } // copies in flight or unread may carry the old word: dropped
        h->step_no = 0;            // a new history: the next LAG steps run the default flavour
        h->srv_exhausted = false;
    }
    return R2S_OK;
}

int r2s_phys_set_state_envs(R2SPhys* h, const float* x, const float* v, const int32_t* env_mask, r2s_stream_t stream_)
{
    if (!h || !env_mask) return R2S_ERR_INVALID;
    // an episode reset of SOME environments: the fault word is per handle and may belong to an environment that keeps running — untouched
    hipLaunchKernelGGL(k_pack, dim3((h->N + 255) / 256, h->E), dim3(256), 0, (hipStream_t)stream_, h->N, h->E, h->d_inv, x, v, h->state(h->cur), env_mask);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

__global__ void k_set_static_pts(int E, int n, int nV, int n_dyn_pts, const int* __restrict__ mask, const float* __restrict__ src, float* __restrict__ mesh_pts)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, e = blockIdx.y;
    if (t >= 3 * n || (mask && mask[e] == 0)) return;
    mesh_pts[((size_t)e * nV + n_dyn_pts) * 3 + t] = src[(size_t)e * 3 * n + t];
}

int r2s_phys_set_static_mesh_points(R2SPhys* h, const float* pts, const int32_t* env_mask, r2s_stream_t stream_)
{
    if (!h || !pts || h->nF == 0) return R2S_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream_;
    const int n = h->nV - h->n_dyn_pts;
    if (n <= 0) return R2S_ERR_INVALID;
    for (int m = h->n_dyn_mesh; m < h->n_mesh; ++m)
        if (h->h_mesh_kind[m] & 1) { // its triangles live in a rest-frame table with a box hierarchy built at create: it cannot move per environment
            r2s::set_last_error_msg("r2s_phys_set_static_mesh_points: a static collision mesh with more than 256 faces cannot be re-posed (unsupported)");
            return R2S_ERR_INVALID;
        }
    hipLaunchKernelGGL(k_set_static_pts, dim3((unsigned)((3 * n + 255) / 256), (unsigned)h->E), dim3(256), 0, s, h->E, n, h->nV, h->n_dyn_pts, env_mask, pts, h->d_mesh_pts);
    const int ns = h->n_mesh - h->n_dyn_mesh, tot = h->E * ns;
    hipLaunchKernelGGL(k_mesh_aabb_static, dim3((tot + 255) / 256), dim3(256), 0, s, h->E, ns, h->n_dyn_mesh, h->nV, h->d_mesh_vert_off, h->d_mesh_pts, h->d_aabb_static);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_get_state(R2SPhys* h, float* x, float* v, r2s_stream_t stream_)
{
    if (!h) return R2S_ERR_INVALID;
    hipLaunchKernelGGL(k_unpack, dim3((h->N + 255) / 256, h->E), dim3(256), 0, (hipStream_t)stream_, h->N, h->E, h->d_inv, h->state(h->cur), x, v);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

__global__ void k_zero_env_rows(size_t words_per_env, const int* __restrict__ env_mask, uint32_t* __restrict__ bits)
{
    const int e = blockIdx.y;
    if (env_mask[e] == 0) return;
    uint32_t* row = bits + (size_t)e * words_per_env;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < words_per_env; k += (size_t)gridDim.x * blockDim.x) row[k] = 0u;
}

static int create_resting_case(R2SPhys* h, const int32_t* env_mask, hipStream_t s)
{
    const uint32_t *keys, *ids;
    int rc = grid_sort(h, s, &keys, &ids);
    if (rc) return rc;
    const size_t per_env = (size_t)h->N * h->words;
    if (env_mask) hipLaunchKernelGGL(k_zero_env_rows, dim3(256, (unsigned)h->E), dim3(256), 0, s, per_env, env_mask, h->d_bits);
    else R2S_HIP_TRY(hipMemsetAsync(h->d_bits, 0, sizeof(uint32_t) * (size_t)h->E * per_env, s));
    const float r = h->prm.collision_dist * 5.0f;
    dim3 grid((h->N + TPB - 1) / TPB, h->E);
    hipLaunchKernelGGL(k_build_resting, grid, dim3(TPB), 0, s, h->N, h->E, h->words, h->d_perm, h->d_inv, h->state(h->cur), r, 1.0f / r, keys, ids, h->d_bits,
                       (const int*)env_mask);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_create_resting_case(R2SPhys* h, r2s_stream_t stream_)
{
    if (!h || !h->prm.self_collision) return R2S_ERR_INVALID;
    return create_resting_case(h, nullptr, (hipStream_t)stream_);
}

int r2s_phys_create_resting_case_envs(R2SPhys* h, const int32_t* env_mask, r2s_stream_t stream_)
{
    if (!h || !h->prm.self_collision || !env_mask) return R2S_ERR_INVALID;
    return create_resting_case(h, env_mask, (hipStream_t)stream_);
}

int r2s_phys_update_collision_graph(R2SPhys* h, r2s_stream_t stream_)
{
    if (!h || !h->prm.self_collision) return R2S_ERR_INVALID; // `assert self.self_collision`, :807
    hipStream_t s = (hipStream_t)stream_;
    const uint32_t *keys, *ids;
    int rc = h->d_cell_tab ? grid_sort(h, s, &keys, &ids, true) : grid_sort(h, s, &keys, &ids);
    if (rc) return rc;
    R2S_HIP_TRY(hipMemsetAsync(h->d_max_count, 0, sizeof(int), s));
    const float r = h->prm.collision_dist * 5.0f;
    dim3 grid((h->N + TPB - 1) / TPB, h->E);
    if (h->d_cell_tab) {
        const float cd = h->prm.collision_dist;
        hipLaunchKernelGGL(k_fine_mark, grid, dim3(TPB), 0, s, h->N, h->E, h->state(h->cur), keys, ids, h->d_cell_tab, h->d_cell_xs);
        hipLaunchKernelGGL(k_candidates_fine, grid, dim3(TPB), 0, s, h->N, h->E, h->words, h->coll_cap, h->state(h->cur), h->d_masks, h->d_perm, cd, 1.0f / cd,
                           1.0f / r, h->d_cell_tab, h->d_cell_xs, h->d_bits, h->d_coll_idx, h->d_coll_num, h->d_max_count);
        hipLaunchKernelGGL(k_cell_clear, grid, dim3(TPB), 0, s, h->N, h->E, keys, h->d_cell_tab);
    } else {
        hipLaunchKernelGGL(k_candidates, grid, dim3(TPB), 0, s, h->N, h->E, h->words, h->coll_cap, h->d_inv, h->state(h->cur), h->d_masks, h->prm.collision_dist, r,
                           1.0f / r, keys, ids, h->d_bits, h->d_coll_idx, h->d_coll_num, h->d_max_count);
    }
    R2S_HIP_TRY(hipMemsetAsync(h->d_cand_count, 0, sizeof(int) * ((size_t)4 + h->E), s));
    hipLaunchKernelGGL(k_cand_list, grid, dim3(TPB), 0, s, h->N, h->E, h->d_coll_num, h->d_cand_list, h->d_cand_count);
    R2S_HIP_TRY(hipMemcpyAsync(h->h_cand_count, h->d_cand_count, sizeof(int), hipMemcpyDeviceToHost, s));
    R2S_HIP_TRY(hipEventRecord(h->cand_event, s));
    h->cand_pending = true;
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
        hipLaunchKernelGGL(k_mesh_aabb_dyn, dim3((tot + 255) / 256), dim3(256), 0, s, E, n_sub, h->n_dyn_mesh, h->n_dyn_pts, h->d_mesh_vert_off, h->d_mesh_kind,
                           h->d_interp, h->d_aabb_dyn);
        int rc = update_mesh_transforms(h, s);
        if (rc) return rc;
    }
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_set_eef_table(R2SPhys* h, int32_t n_knots, const double* eef_pts, const float* init_eef_xyz, float grasp_force_threshold,
                           r2s_stream_t stream_)
{
    if (!h || h->n_dyn_mesh == 0 || n_knots < 2 || !eef_pts || !init_eef_xyz) return R2S_ERR_INVALID;
    if (!h->prm.use_pusher && (h->n_dyn_mesh < 2 || h->h_foff[1] - h->h_foff[0] < 20 || h->h_foff[2] - h->h_foff[1] < 20))
        return R2S_ERR_INVALID; // the grasp test reads faces 1, 18, 19 of the two finger meshes (phystwin.py:386-387)
    hipStream_t s = (hipStream_t)stream_;
    const int E = h->E, M = h->n_dyn_pts;
    if (h->d_eef_table) { (void)hipFree(h->d_eef_table); h->d_eef_table = nullptr; }
    int rc = dev_alloc(&h->d_eef_table, (size_t)n_knots * M * 3);
    if (rc) return rc;
    R2S_HIP_TRY(hipMemcpyAsync(h->d_eef_table, eef_pts, sizeof(double) * (size_t)n_knots * M * 3, hipMemcpyHostToDevice, s));
    h->eef_knots = n_knots; h->eef_thr = grasp_force_threshold;
    for (int k = 0; k < 3; ++k) h->eef_init[k] = init_eef_xyz[k];
    if (!h->d_eef_open) {
        if ((rc = dev_alloc(&h->d_eef_open, E)) || (rc = dev_alloc(&h->d_eef_grasped, E)) || (rc = dev_alloc(&h->d_eef_has, E)) ||
            (rc = dev_alloc(&h->d_eef_rel0, (size_t)E * M * 3)) || (rc = dev_alloc(&h->d_eef_delta, (size_t)E * M * 3)))
            return rc;
        // vertices the stepper reads: everything of a small mesh; reference + rigidity-sample vertices of a large one (k_mesh_xf)
        std::vector<int> need;
        for (int m = 0; m < h->n_dyn_mesh; ++m) {
            const int v0 = h->h_voff[m], v1 = h->h_voff[m + 1];
            if (!(h->h_mesh_kind[m] & 1)) { for (int v = v0; v < v1; ++v) need.push_back(v); continue; }
            const int stride = std::max(1, (v1 - v0) / 48);
            for (int v = v0; v < v1; v += stride) need.push_back(v);
            for (size_t k = 0; k < h->h_xf_mesh.size(); ++k)
                if (h->h_xf_mesh[k] == m) for (int j = 0; j < 3; ++j) need.push_back(h->h_xf_ref[3 * k + j]);
        }
        std::sort(need.begin(), need.end()); need.erase(std::unique(need.begin(), need.end()), need.end());
        h->eef_n_need = (int)need.size();
        if ((rc = dev_alloc(&h->d_eef_need, need.size())) || (rc = upload(h->d_eef_need, need.data(), need.size(), s))) return rc;
    }
    R2S_HIP_TRY(hipMemsetAsync(h->d_eef_open, 0, sizeof(double) * E, s));
    R2S_HIP_TRY(hipMemsetAsync(h->d_eef_grasped, 0, sizeof(int) * E, s));
    R2S_HIP_TRY(hipMemsetAsync(h->d_eef_has, 0, sizeof(int) * E, s));
    R2S_HIP_TRY(hipStreamSynchronize(s)); // eef_pts is a host buffer of the caller
    return R2S_OK;
}

int r2s_phys_set_eef_motion(R2SPhys* h, const float* eef_xyz, const float* eef_vel, const float* eef_rot, const float* eef_rot_vel,
                            const float* gripper_openness, r2s_stream_t stream_)
{
    if (!h || !h->d_eef_table || !eef_xyz || !eef_vel || !eef_rot || !eef_rot_vel || (!h->prm.use_pusher && !gripper_openness)) return R2S_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream_;
    const int E = h->E, n_sub = h->prm.num_substeps, M = h->n_dyn_pts;
    EefIn in{eef_xyz, eef_vel, eef_rot, eef_rot_vel, gripper_openness};
    const float dt = h->prm.dt;
    const float dt_n = (float)((double)h->prm.dt * n_sub), two_dt_n = (float)(2.0 * (double)h->prm.dt * n_sub);
    const int f_left = h->h_foff[0], f_right = h->n_dyn_mesh > 1 ? h->h_foff[1] : h->h_foff[0];
    hipLaunchKernelGGL(k_eef_prepare, dim3(E), dim3(256), 0, s, E, M, h->eef_knots, (int)h->prm.use_pusher, h->d_eef_table, h->eef_init[0], h->eef_init[1],
                       h->eef_init[2], h->eef_thr, f_left, f_right, h->nF, h->d_coll_forces, in, h->d_eef_open, h->d_eef_grasped, h->d_eef_has, h->d_eef_rel0,
                       h->d_eef_delta, h->d_dyn_vel, h->d_dyn_omega, two_dt_n);
    hipLaunchKernelGGL(k_eef_points, dim3((h->eef_n_need + 255) / 256, n_sub, E), dim3(256), 0, s, E, n_sub, M, h->eef_n_need, h->d_eef_need, in, h->d_eef_rel0,
                       h->d_eef_delta, dt, dt_n, h->d_interp, h->d_center);
    const int tot = E * n_sub * h->n_dyn_mesh;
    hipLaunchKernelGGL(k_mesh_aabb_dyn, dim3((tot + 255) / 256), dim3(256), 0, s, E, n_sub, h->n_dyn_mesh, h->n_dyn_pts, h->d_mesh_vert_off, h->d_mesh_kind,
                       h->d_interp, h->d_aabb_dyn);
    int rc = update_mesh_transforms(h, s);
    if (rc) return rc;
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_eef_state(R2SPhys* h, double** current_openness, int32_t** grasped)
{
    if (!h || !h->d_eef_open) return R2S_ERR_INVALID;
    if (current_openness) *current_openness = h->d_eef_open;
    if (grasped) *grasped = h->d_eef_grasped;
    return R2S_OK;
}

// Episode reset of the environments whose mask entry is non-zero (no mask: all): what a NEW SpringMassDynamicsModule starts from
// (phystwin.py:39-102 builds one per reset) — current_openness = None, grasped = False (phystwin.py:358-360), collision_forces zero.
__global__ void k_reset_envs(int nF, const int* __restrict__ mask, double* __restrict__ open, int* __restrict__ grasped, int* __restrict__ has,
                             float* __restrict__ coll_forces, int* __restrict__ hit_cnt)
{
    const int e = (int)blockIdx.y;
    if (mask && mask[e] == 0) return;
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t == 0) {
        if (open) { open[e] = 0.0; grasped[e] = 0; has[e] = 0; }
        if (hit_cnt) hit_cnt[e] = 0;
    }
    if (coll_forces && t < 3 * nF) coll_forces[(size_t)e * 3 * nF + t] = 0.f;
}

int r2s_phys_reset_envs(R2SPhys* h, const int32_t* env_mask, r2s_stream_t stream_)
{
    if (!h) return R2S_ERR_INVALID;
    const unsigned gx = (unsigned)std::max(1, (3 * h->nF + 255) / 256);
    hipLaunchKernelGGL(k_reset_envs, dim3(gx, (unsigned)h->E), dim3(256), 0, (hipStream_t)stream_, h->nF, env_mask, h->d_eef_open, h->d_eef_grasped,
                       h->d_eef_has, h->nF > 0 ? h->d_coll_forces : nullptr, h->d_hit_cnt);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_mesh_motion(R2SPhys* h, float** interp_points, float** interp_center, float** dynamic_velocity, float** dynamic_omega)
{
    if (!h || h->n_dyn_mesh == 0) return R2S_ERR_INVALID;
    if (interp_points) *interp_points = h->d_interp;
    if (interp_center) *interp_center = h->d_center;
    if (dynamic_velocity) *dynamic_velocity = h->d_dyn_vel;
    if (dynamic_omega) *dynamic_omega = h->d_dyn_omega;
    return R2S_OK;
}

int r2s_phys_step(R2SPhys* h, int n_substeps, int first_substep, r2s_stream_t stream_)
{
    if (!h) return R2S_ERR_INVALID;
    r2s::set_last_error_msg(""); // an R2S_ERR_INVALID of this call with a message carries THIS call's reason
    hipStream_t s = (hipStream_t)stream_;
    const int full = h->prm.num_substeps;
    const bool use_graph = (n_substeps <= 0 || n_substeps == full) && first_substep == 0;
    const int n = use_graph ? full : n_substeps;
    if (first_substep < 0 || first_substep + n > full) return R2S_ERR_INVALID;
    if (h->timing) {
        if (!h->ev0) { R2S_HIP_TRY(hipEventCreate(&h->ev0)); R2S_HIP_TRY(hipEventCreate(&h->ev1)); }
        R2S_HIP_TRY(hipEventRecord(h->ev0, s));
    }
    int rc0 = h->prm.self_collision ? resolve_cand_count(h) : R2S_OK;
    if (rc0) return rc0;
    // large dynamic meshes must move rigidly: the check ran with the last set_mesh_interactive / set_eef_motion; its result
    // is read without waiting (pinned word + event), so a violation is reported by the first step() after it landed
    if (h->rigid_pending && hipEventQuery(h->rigid_event) == hipSuccess) {
        h->rigid_pending = false;
        float worst; memcpy(&worst, h->h_rigid_err, 4);
        if (!(worst < 1e-4f)) {
            r2s::set_last_error_msg("a dynamic collision mesh with more than 256 faces does not move rigidly (unsupported)");
            return R2S_ERR_INVALID;
        }
    }
    const int variant = (h->prm.self_collision && h->n_cand > 0) ? 1 : 0;
    // the counters this step's flavour follows from: those of step (step_no - LAG), waited for (long landed in any real loop)
    int cnt[16] = {0};
    bool have_cnt = false;
    if (h->step_no >= (uint64_t)R2SPhys::LAG) {
        const int k = (int)((h->step_no - R2SPhys::LAG) % R2SPhys::RING);
        if (h->ring_pending[k]) { R2S_HIP_TRY(hipEventSynchronize(h->ring_ev[k])); h->ring_pending[k] = false; }
        memcpy(cnt, h->h_ring + 16 * k, sizeof cnt);
        if (h->ring_stale_fault[k]) cnt[1] = 0;
        have_cnt = true;
    }
    // the sticky fault word is looked for in the NEWER copies too when they have landed: it only ends the run, it picks no flavour
    for (uint64_t back = 1; back < (uint64_t)R2SPhys::LAG && back <= h->step_no && cnt[1] == 0; ++back) {
        const int k = (int)((h->step_no - back) % R2SPhys::RING);
        if (h->ring_pending[k] && hipEventQuery(h->ring_ev[k]) == hipSuccess) h->ring_pending[k] = false;
        if (!h->ring_pending[k] && !h->ring_stale_fault[k] && h->h_ring[16 * k + 1] != 0) { cnt[1] = h->h_ring[16 * k + 1]; memcpy(cnt + 4, h->h_ring + 16 * k + 4, 12 * sizeof(int)); }
    }
    if (cnt[1] != 0) { // the sticky fault word of an earlier step
        if (cnt[1] >= 2 && cnt[1] <= 6) {
            char buf[640];
            const int* w = cnt + 4;
            snprintf(buf, sizeof buf, "resident stepper: a workgroup waited for %s beyond the poll limit (the launch was not resident at once, or the device "
                     "is shared with a kernel that never ends); the state is invalid [first fault: code %d, work item %d, substep %d of the launch, context %d %d %d %d %d %d]",
                     cnt[1] == 2 ? "a neighbour block's substep" : cnt[1] == 5 ? "a neighbour's record (server pair)" : cnt[1] == 6 ? "a particle finished by the head of its launch" : "a mesh-query server's result", w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8]);
            r2s::set_last_error_msg(buf);
        }
        else
            r2s::set_last_error_msg("a self-collision impulse changed a particle's velocity by more than 40 m/s within one substep: its mesh-contact "
                                    "test (widened by 2 mm) may have been skipped where the reference applies it (unsupported)");
        return R2S_ERR_INVALID;
    }
    if (h->nF > 0) { // defer the mesh queries to k_contact_finish when the last finished step saw particles near a mesh
        // an unfinished count keeps the previous flavour.  Large batches defer as soon as anything is NEAR a mesh (an idle finishing launch
        // costs them ~0.7 us per substep, an in-place query in the fused kernel up to 190); small batches only once a query was NEEDED:
        // their free flavour is the resident launch (2.3 vs 7.2 us per substep for the rope), and a gripper hovering within 3 cm is free
        // motion — the step in which the first particle enters a margin pays for its in-place queries once
        h->mesh_defer = (h->resident_ok && h->resident_pref ? cnt[2] : cnt[0]) > 0 ? 1 : 0;
        // ... and with query servers in the launch (round 4) a small batch stays resident THROUGH contact: a particle that needs a query is
        // answered by a server pair of the same launch (resident_server); only the self-collision flavour still takes the per-substep path
        if (have_cnt) { // (more particles in contact than the launch has pairs: answered in place — correct, and 20 x slower than the finishing launch)
            if (cnt[3] > 0) h->srv_exhausted = true;
            else if (cnt[2] == 0) h->srv_exhausted = false;
        }
        if (h->resident_ok && h->resident_pref && h->srv_ok && variant == 0 && !h->srv_exhausted) h->mesh_defer = 0;
        if (h->force_defer >= 0) h->mesh_defer = h->force_defer; // test / tuning: force a flavour
        if (h->any_large) h->mesh_defer = 1;
    }
    h->last_flavour[0] = variant | (h->split_ok ? 2 : 0); h->last_flavour[1] = h->nF > 0 ? (h->any_large ? 2 : 1) : 0; h->last_flavour[2] = h->mesh_defer;
    h->last_flavour[3] = use_graph ? h->chains() : 1;
    if (h->nF > 0 && h->pf_ok && h->pf_pref != 0 && h->pb == 256 && (h->mesh_defer || h->any_large)) h->last_flavour[2] = 3; // 3 = deferred, finishers at the head of the next launch
    const bool resident = resident_flavour(h, variant == 1, h->mesh_defer);
    if (resident) {
        h->last_flavour[2] = 2; // 2 = the resident launch (never with deferred queries)
        int n_srv = h->srv_ok && n > 1 ? std::min(h->n_cu - 8 * ((h->nb * h->E + 7) / 8), h->srv_wg_cap) : 0; // as enqueue_steps sizes the grid
        if (n_srv < SRV_MIN_WG) n_srv = 0;
        h->last_flavour[3] = 1 | (n_srv << 8) | ((h->srv_own ? 1 : 0) << 20) | ((n_srv > 0 && h->srv_quad_for(n_srv) ? 1 : 0) << 21);
    }
    int gate_dev = -1;
    if (resident) { int rcg = resident_enter(s, &gate_dev); if (rcg) return rcg; }
    struct GateLeave { hipStream_t s; int dev; ~GateLeave() { if (dev >= 0) (void)resident_leave(s, dev); } } gate_leave{s, gate_dev};
    if (use_graph) {
        // every flavour was captured at construction (capture_all); only set_params / set_tuning drop them
        const int slot = h->mesh_defer * 4 + variant * 2 + (h->cur & 1);
        if (!h->graph_exec[0][slot]) {
            const int keep = h->mesh_defer;
            int rc = capture_graph(h, variant, h->cur);
            h->mesh_defer = keep;
            if (rc) return rc;
        }
        int rcl = launch_graphs(h, slot, s);
        if (rcl) return rcl;
    } else {
        int rc = enqueue_steps(h, first_substep, n, h->cur, variant == 1, s);
        if (rc) return rc;
    }
    h->cur ^= resident ? 1 : (n & 1);
    { // this step's counters (+ the fault word) -> the step's slot of the pinned ring, read LAG steps from now
        const int k = (int)(h->step_no % R2SPhys::RING);
        R2S_HIP_TRY(hipMemsetAsync(h->d_mesh_total, 0, sizeof(int), s));
        if (h->nF > 0) hipLaunchKernelGGL(k_sum_i32, dim3(1), dim3(64), 0, s, h->d_mesh_cnt + h->prm.num_substeps, 8, h->prm.num_substeps + 1, h->d_mesh_total);
        R2S_HIP_TRY(hipMemcpyAsync(h->h_ring + 16 * k, h->d_mesh_total, 16 * sizeof(int), hipMemcpyDeviceToHost, s));
        R2S_HIP_TRY(hipMemsetAsync(h->d_mesh_total + 2, 0, 2 * sizeof(int), s)); // "a query was needed", "no server pair was left": counted from here on
        R2S_HIP_TRY(hipEventRecord(h->ring_ev[k], s));
        h->ring_pending[k] = true; h->ring_stale_fault[k] = false;
        ++h->step_no;
    }
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
    if (capacity) *capacity = h->coll_cap;
    if (!h->prm.self_collision) { if (number) *number = h->d_coll_num; if (indices) *indices = nullptr; return R2S_OK; }
    // internal (Morton) indexing -> the caller's indexing, into side buffers (parity taps, not a hot path)
    if (!h->d_num_user) {
        R2S_HIP_TRY(r2s::dev_malloc((void**)&h->d_num_user, sizeof(int) * (size_t)h->E * h->N));
        R2S_HIP_TRY(r2s::dev_malloc((void**)&h->d_idx_user, sizeof(int) * (size_t)h->E * h->N * h->coll_cap));
    }
    R2S_HIP_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_lists_to_user, dim3((h->N + 255) / 256, h->E), dim3(256), 0, 0, h->N, h->E, h->coll_cap, h->d_perm, h->d_coll_num,
                       h->d_coll_idx, h->d_num_user, h->d_idx_user);
    R2S_HIP_TRY(hipDeviceSynchronize());
    if (number) *number = h->d_num_user;
    if (indices) *indices = h->d_idx_user;
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

int r2s_phys_set_params(R2SPhys* h, const R2SPhysParams* p, r2s_stream_t stream_)
{
    if (!h || !p) return R2S_ERR_INVALID;
    if (p->num_substeps != h->prm.num_substeps || p->self_collision != h->prm.self_collision || p->use_pusher != h->prm.use_pusher)
        return R2S_ERR_INVALID; // structural fields are fixed at construction
    if (!(p->dt > 0.f) || !(p->collision_dist > 0.f) || !(p->spring_Y_max >= p->spring_Y_min)) return R2S_ERR_INVALID;
    const bool restiffen = p->spring_Y_min != h->prm.spring_Y_min || p->spring_Y_max != h->prm.spring_Y_max;
    h->prm = *p;
    drop_graph(h); // kernel arguments are baked into the graph; re-captured lazily by the next step
    // the stiffness gate and clamp (:75, :93) are baked into the adjacency tables: rebuild them from the last log stiffness
    if (restiffen && h->S > 0 && !h->h_logY.empty()) return upload_stiffness(h, h->h_logY.data(), (hipStream_t)stream_);
    return R2S_OK;
}

int r2s_phys_set_tuning(R2SPhys* h, int chains, int mesh_defer)
{
    if (!h) return R2S_ERR_INVALID;
    h->chains_override = chains > 0 ? chains : 0;
    h->force_defer = mesh_defer < 0 ? -1 : (mesh_defer != 0);
    drop_graph(h);
    return R2S_OK;
}

int r2s_phys_side_stream(int32_t k, r2s_stream_t* out)
{
    if (!out || k < 1 || k >= R2SPhys::MAX_CHAINS) return R2S_ERR_INVALID;
    hipStream_t s = chain_side_stream(k);
    if (!s) return R2S_ERR_HIP;
    *out = (r2s_stream_t)s;
    return R2S_OK;
}

int r2s_phys_set_pf(R2SPhys* h, int on)
{
    if (!h) return R2S_ERR_INVALID;
    h->pf_pref = on != 0;
    drop_graph(h);
    return R2S_OK;
}

int r2s_phys_set_resident(R2SPhys* h, int on)
{
    if (!h) return R2S_ERR_INVALID;
    h->resident_pref = on != 0;
    drop_graph(h);
    return R2S_OK;
}

int r2s_phys_last_flavour(R2SPhys* h, int32_t* out /* [4] */)
{
    if (!h || !out) return R2S_ERR_INVALID;
    for (int k = 0; k < 4; ++k) out[k] = h->last_flavour[k];
    return R2S_OK;
}

int r2s_phys_log_contacts(R2SPhys* h, int32_t* out3_dev, r2s_stream_t stream_)
{
    if (!h || !out3_dev) return R2S_ERR_INVALID;
    hipLaunchKernelGGL(k_log_contacts, dim3(1), dim3(64), 0, (hipStream_t)stream_, h->E, h->d_cand_count, h->d_hit_cnt, h->d_eef_grasped, out3_dev);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_contact_stats(R2SPhys* h, int32_t* particles_with_candidates, int32_t** mesh_hits_dev)
{
    if (!h) return R2S_ERR_INVALID;
    if (particles_with_candidates) {
        if (h->prm.self_collision) { int rc = resolve_cand_count(h); if (rc) return rc; }
        *particles_with_candidates = h->n_cand;
    }
    if (mesh_hits_dev) *mesh_hits_dev = h->d_hit_cnt;
    return R2S_OK;
}

// Diagnostics: how many particles the fused kernel handed to k_contact_finish in each substep of the last env step
// (summed over the kernel chains), out[n_sub] = 1 if anything was near a mesh.  Synchronises the stream.
int r2s_phys_deferred_counts(R2SPhys* h, int32_t* out, r2s_stream_t stream_)
{
    if (!h || !out) return R2S_ERR_INVALID;
    const int n = h->prm.num_substeps + 1;
    for (int k = 0; k < n; ++k) out[k] = 0;
    if (h->nF == 0) return R2S_OK;
    std::vector<int> tmp((size_t)8 * n);
    R2S_HIP_TRY(hipMemcpyAsync(tmp.data(), h->d_mesh_cnt, sizeof(int) * tmp.size(), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    R2S_HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    for (int c = 0; c < 8; ++c)
        for (int k = 0; k < n; ++k) out[k] += tmp[(size_t)c * n + k];
    if (h->d_rec_cnt) { // large-mesh scenes count their records per environment
        std::vector<int> rc((size_t)h->E * (n - 1));
        R2S_HIP_TRY(hipMemcpyAsync(rc.data(), h->d_rec_cnt, sizeof(int) * rc.size(), hipMemcpyDeviceToHost, (hipStream_t)stream_));
        R2S_HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
        for (int e = 0; e < h->E; ++e)
            for (int k = 0; k < n - 1; ++k) out[k] += rc[(size_t)e * (n - 1) + k];
    }
    return R2S_OK;
}

// Diagnostics: particles with self-collision candidates that the fused kernel ALSO handed to the mesh list (tagged entries:
// impulses + query in one go by k_contact_finish part 1) at least once during the last r2s_phys_step.  Synchronises the stream.
int r2s_phys_tagged_count(R2SPhys* h, int32_t* out, r2s_stream_t stream_)
{
    if (!h || !out) return R2S_ERR_INVALID;
    *out = 0;
    if (!h->d_cand_mark) return R2S_OK;
    const size_t en = (size_t)h->E * h->N;
    std::vector<int> tmp(2 * en); // marks are kept per substep parity
    R2S_HIP_TRY(hipMemcpyAsync(tmp.data(), h->d_cand_mark, sizeof(int) * tmp.size(), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    R2S_HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    int n = 0;
    for (size_t k = 0; k < en; ++k) n += (tmp[k] != 0 || tmp[en + k] != 0);
    *out = n;
    return R2S_OK;
}

// Candidate lists written by the caller (collision_number / collision_indices are plain arrays in the reference, :544-552):
// HOST arrays in the caller's indexing, converted to the internal order.  Parity tests use it to replay the reference's
// own lists; the next update_collision_graph overwrites them.
int r2s_phys_set_collision_lists(R2SPhys* h, const int32_t* number, const int32_t* indices, r2s_stream_t stream_)
{
    if (!h || !h->prm.self_collision || !number || !indices) return R2S_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream_;
    const int N = h->N, E = h->E, cap = h->coll_cap;
    std::vector<int> num((size_t)E * N), idx((size_t)E * N * cap, 0);
    for (int e = 0; e < E; ++e)
        for (int i = 0; i < N; ++i) {
            const size_t src = (size_t)e * N + h->h_perm[i], dst = (size_t)e * N + i;
            const int c = number[src];
            if (c < 0 || c > cap) return R2S_ERR_INVALID;
            num[dst] = c;
            for (int k = 0; k < c; ++k) {
                const int uj = indices[src * cap + k];
                if (uj < 0 || uj >= N) return R2S_ERR_INVALID;
                idx[dst * cap + k] = h->h_inv[uj];
            }
        }
    // object_collision reads v_before_collision[j] of every listed partner j, which only particles that have a list of their own
    // publish: the candidate relation the reference builds is symmetric (:196-227; a row capped at the capacity may drop
    // partners, its particle still publishes) — a partner without any list would be read stale, so such lists are refused.
    for (int e = 0; e < E; ++e)
        for (int i = 0; i < N; ++i) {
            const size_t row = (size_t)e * N + i;
            for (int k = 0; k < num[row]; ++k)
                if (num[(size_t)e * N + idx[row * cap + k]] == 0) {
                    r2s::set_last_error_msg("r2s_phys_set_collision_lists: a listed partner has no candidate list of its own (the relation must be symmetric)");
                    return R2S_ERR_INVALID;
                }
        }
    int rc = upload(h->d_coll_num, num.data(), num.size(), s);
    if (rc) return rc;
    rc = upload(h->d_coll_idx, idx.data(), idx.size(), s);
    if (rc) return rc;
    R2S_HIP_TRY(hipMemsetAsync(h->d_cand_count, 0, sizeof(int) * ((size_t)4 + h->E), s));
    hipLaunchKernelGGL(k_cand_list, dim3((N + TPB - 1) / TPB, E), dim3(TPB), 0, s, N, E, h->d_coll_num, h->d_cand_list, h->d_cand_count);
    R2S_HIP_TRY(hipMemcpyAsync(h->h_cand_count, h->d_cand_count, sizeof(int), hipMemcpyDeviceToHost, s));
    R2S_HIP_TRY(hipEventRecord(h->cand_event, s));
    h->cand_pending = true;
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_phys_layout_stats(R2SPhys* h, int64_t* out /* [8] */)
{
    if (!h || !out) return R2S_ERR_INVALID;
    int64_t real = 0, fallback = 0;
    for (int t = 0; t < h->ell_len; ++t) if (h->h_adj_spring[t] >= 0) ++real;
    for (int t = 0; t < h->rell_len; ++t) if (h->h_radj_spring[t] >= 0) { ++real; ++fallback; }
    out[0] = h->nb; out[1] = h->halo_max; out[2] = h->ell_len + h->rell_len; out[3] = real; out[4] = fallback;
    out[5] = (int64_t)h->rcap * 24; out[6] = h->chains(); out[7] = h->cb;
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
