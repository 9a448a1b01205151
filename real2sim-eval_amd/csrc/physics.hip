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
//   * Round 5: in the contact flavours of a large batch that finishing code no longer is a launch of its own — it rides at the HEAD
//     of the next substep's launch (k_substep_pf, physics_finish.h; hand-off through PF_SENT records and tagged result lines,
//     physics_substep.h), and which flavour an env step runs follows from the counters of the step two before it (r2s_phys_step).
//
// One translation unit in six files (round 5; the token stream is unchanged): this file — PhysDev, small math, state layout, the host
// driver and the C ABI — and, included below inside the anonymous namespace, physics_mesh_query.h, physics_substep.h,
// physics_resident.h, physics_finish.h, physics_aux.h.

#include "r2s_common.h"
#include <rocprim/rocprim.hpp>
#include "../../include/r2s_physics.h"
#include "physics_flavour.h"
#include <algorithm>
#include <array>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>
#include <functional>

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
    const float4* tri_pre;     // small scenes with batched finishing: [E][n_sub][5][nF] the triangles of every (environment, substep) ready to stage —
                               // {a.xyz b.x} {b.yz c.xy} {c.z, mesh_map, face_map, mesh} {box lo.xyz hi.x} {hi.yz - -} — written once per env step (k_tri_pre)
    const int* mesh_inward;    // [n_mesh] 1: a closed manifold whose faces are oriented INWARD (negative signed volume at construction): its winding number is
                               // -1 inside, never above the 0.6 threshold — the batched small-scene finisher's plane-side shortcut must not be used for it
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
    void* vx;                  // resident stepper, self-collision flavour: {x0, post-force v} of the particles with candidates, laid out like xch
                               // ([E][2 substep parities][3 planes][N padded to 8] x 16 B), written by their block's wavefront 0, polled by their candidates' blocks
    // resident stepper, mesh-query SERVERS (small scenes; see k_steps_resident): workgroups of the same launch beyond the blocks' own,
    // two wavefronts per served particle
    int srv_slots;             // server wavefront pairs of this launch (0: none — queries in place)
    int srv_low;               // a claim of slot >= srv_low reports "units running low" (30 % of srv_slots; R2S_RES_SRV_LOW = per cent, 100: only when none is left)
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

// ---- the device code, by section (one translation unit: see the note at the top of each file) ----
#include "physics_mesh_query.h"
#include "physics_substep.h"
#include "physics_resident.h"
#include "physics_finish.h"
#include "physics_aux.h"

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
    int4* d_mesh_rec = nullptr; int* d_rec_cnt = nullptr; // large-mesh scenes and small scenes with batched finishing: per-environment records [E][N][2] and their counters [E][n_sub]
    bool fin_batch = false;     // small scene (every mesh small, <= 128 faces, <= FB_MAX_MESH meshes): the finishing code takes 16 records of an environment at a time (contact_finish_batch; R2S_FIN_BATCH=0: one workgroup per particle, rounds 2-5)
    int* d_cand_mark = nullptr;
    // Counters of an env step ([0] particles near a mesh, [1] sticky fault word, [2] a query was needed, [3] server pairs ran out, [4..15] fault
    // context) travel to pinned memory behind the step and pick the FLAVOUR of a later step.  Which later step is fixed (round 5): step t
    // runs the flavour that follows from the counters of step t - LAG, waited for if they have not landed (they have: two env steps ago) —
    // never "whatever copy happens to have arrived", which made the bits of a run in contact depend on host timing (the flavours sum in
    // different orders).  A full set_state starts a new history: its first LAG steps run the default flavour (queries in place).
    static constexpr int LAG = 2, RING = 4;   // LAG: the default; `lag` below is what the handle uses
    // Round 6: SMALL batches (the resident layout) use lag 1.  Their contact flavours differ by a factor of ten in cost when the wrong one runs
    // (a resident launch whose server units run out answers in place: 55 - 140 ms for an env step of the one-environment toy against 8 - 15), and
    // a closing grasp quadruples its contacts from one env step to the next (24 -> 95 particles in the step the grasp latches) — two steps of
    // lag cannot see that coming, one can.  The price: r2s_phys_step waits for the PREVIOUS step's counters, i.e. the host no longer runs a step
    // ahead of the device — nothing in a closed loop (the caller has waited for that step's frames), a few per cent of an enqueue-only loop of
    // 2 - 4 ms steps.  Large batches (0.8 ms of host time per step for four graph launches) keep 2.  Any FIXED lag keeps runs bit-reproducible.
    int lag = LAG;
    int* d_mesh_total = nullptr; int* h_ring = nullptr; // pinned [RING][16]
    hipEvent_t ring_ev[RING] = {}; bool ring_pending[RING] = {}, ring_stale_fault[RING] = {};
    uint64_t step_no = 0; // env steps enqueued since the last full set_state
    void* d_xch = nullptr;    // resident stepper: exchange array (96 B per particle)
    void* d_vx = nullptr;     // resident stepper, self-collision flavour: {x0, post-force v} of the particles with candidates, laid out like d_xch
    int res_self = 1;         // R2S_RES_SELF=0: a small batch with live candidates takes the per-substep kernels (rounds 3-4)
    int self_srv = 0;         // this env step: query servers in the resident launch of the self-collision flavour (a query was needed two steps ago)
    int res_self_srv = 1;     // R2S_RES_SELF_SRV=0: no query servers next to the self-collision flavour (a needed query then sends the next steps to the per-substep kernels); 2: always
    void* d_srv_claim = nullptr; void* d_srv_rr = nullptr; // resident stepper's mesh-query servers: a 128-byte line per claim, one of control words, one per pair of fault-report state; a line of request and a line of result granules per particle
    bool srv_exhausted = false; // a launch ran out of server pairs: per-substep kernels + finishing launch until the contact is over
    bool srv_ok = false;      // small scene (every mesh small, <= 128 faces in total): a resident launch may carry query servers
    int srv_wg_cap = SRV_MAX_SLOTS / 4; // R2S_RES_SRV_WG: at most this many server workgroups per launch
    int n_cu = 256;
    unsigned spin_limit = RES_SPIN_LIMIT;
    int srv_quad = -1;        // R2S_RES_SRV_QUAD=0 / 1: pairs / quads whatever the launch has room for (-1: quads when it has >= 64 server workgroups)
    int srv_quad_for(int n_srv) const { return r2s_flavour::fl_srv_quad(flavour_caps(), n_srv); }
    int srv_own = 1;          // R2S_RES_SRV_OWN=0: one request per substep instead of pairs that own their particle
    int srv_low_pct = 30;     // R2S_RES_SRV_LOW: share of a launch's server units whose claim reports "running low" (the host leaves the resident launch before a claim goes unanswered)
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
    // capabilities + switches of this handle as the flavour functions read them (physics_flavour.h); the counters are the caller's to fill
    R2SFlavourIn flavour_caps() const
    {
        R2SFlavourIn c{};
        c.n_faces = nF; c.any_large = any_large; c.block = pb; c.split_ok = split_ok; c.resident_ok = resident_ok; c.srv_ok = srv_ok;
        c.pf_ok = pf_ok; c.has_vx = d_vx != nullptr; c.self_collision = prm.self_collision; c.n_blocks = nb; c.n_env = E; c.n_cu = n_cu;
        c.srv_wg_cap = srv_wg_cap; c.resident_pref = resident_pref; c.res_self = res_self; c.res_self_srv = res_self_srv; c.pf_pref = pf_pref;
        c.force_defer = force_defer; c.chains_override = chains_override; c.srv_own = srv_own; c.srv_quad = srv_quad;
        c.srv_exhausted = srv_exhausted; c.n_candidates = n_cand; c.n_substeps = prm.num_substeps; c.full_step = 1;
        return c;
    }
    int chains() const { return r2s_flavour::fl_chains(flavour_caps()); } // parallel kernel chains of the captured env step
    R2SFlavourOut last_pick{};   // what the last r2s_phys_step decided
    uint32_t *d_bits = nullptr, *d_keys[2] = {nullptr, nullptr}, *d_ids[2] = {nullptr, nullptr};
    int2* d_cell_tab = nullptr;  // [E << 21] direct cell table (null when it would exceed 4 GiB: binary search instead)
    float4* d_cell_xs = nullptr; // [E,N] sorted positions + internal index
    char* d_sort_tmp = nullptr;
    size_t sort_bytes = 0;
    int *d_faces = nullptr, *d_mesh_map = nullptr, *d_face_map = nullptr, *d_mesh_face_off = nullptr, *d_mesh_vert_off = nullptr;
    float4* d_tri_pre = nullptr;
    int *d_face_orig = nullptr, *d_mesh_kind = nullptr, *d_mesh_inward = nullptr,
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
    hipGraph_t graph[MAX_CHAINS][12] = {};      // [mesh_defer * 4 + variant * 2 + parity]; [8 + 2 + parity]: the resident self-collision flavour WITH servers (self_srv)
    hipGraphExec_t graph_exec[MAX_CHAINS][12] = {};
    hipGraph_t graph_tail[MAX_CHAINS][12] = {};      // chains captured as head + tail (capture_graph): the tail; null where the capture was not split
    hipGraphExec_t graph_exec_tail[MAX_CHAINS][12] = {};
    int graph_head = 64;                             // substeps in the head graph (R2S_GRAPH_HEAD; 0: one graph per chain)
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
        p.cl_box = d_cl_box; p.mesh_kind = d_mesh_kind; p.mesh_inward = d_mesh_inward; p.tri_pre = d_tri_pre; p.mesh_xf = d_mesh_xf; p.xf_mesh = d_xf_mesh; p.xf = d_xf; p.rest_pts = d_rest_pts;
        p.pnorm = d_pnorm; p.tri_rest = d_tri_rest; p.cl_info = d_cl_info;
        p.n_sup = n_sup; p.n_small = n_small; p.sup_box = d_sup_box; p.sup_info = d_sup_info; p.small_mesh = d_small_mesh;
        p.mesh_pts = d_mesh_pts; p.interp_pts = d_interp; p.interp_center = d_center; p.dyn_vel = d_dyn_vel; p.dyn_omega = d_dyn_omega;
        p.aabb_dyn = d_aabb_dyn; p.aabb_static = d_aabb_static; p.coll_forces = d_coll_forces; p.hit_cnt = d_hit_cnt;
        p.fault = d_mesh_total ? d_mesh_total + 1 : nullptr;
        p.xch = d_xch; p.vx = d_vx;
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
bool has_contact_finish(const R2SPhys* h, const PhysDev& p) { return r2s_flavour::fl_contact_finish(h->flavour_caps(), p.mesh_defer); }

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

// batched small-scene finishing: workgroups per environment.  Part 1 takes the records 16 at a time from the first slot on (8 slots = 128 listed
// particles per environment without a second round), part 2 the candidate particles 16 to a workgroup from the last slot back.  With candidates
// 32: at 16 the two parts met in the middle slots of the environments with the most contacts (~100 listed + ~140 candidate particles), those
// workgroups ran part 2 behind part 1 and the blocks waiting for them ended the launch 3 us later — 30.4 / 28.2 / 28.4 / 27.6 us per batched
// substep of the held grasp with 16 / 20 / 24 / 32 slots (hovering: 19.3 / 19.4 / 19.5 / 19.7).
int fin_batch_slots(bool with_self)
{
    int n = with_self ? 16 : 8;
    if (const char* ev = getenv("R2S_FIN_SLOTS")) n = std::max(1, atoi(ev));
    return n;
}

// What the fused kernel left unfinished: with something near a mesh (mesh_defer) ONE combined finishing kernel per substep —
// deferred mesh queries, one workgroup per particle, plus the self-collision impulses; otherwise only k_self_finish
// while candidates exist (mesh queries of the rare needy particle in place).
void launch_finish(R2SPhys* h, const PhysDev& p, int in_buf, int step, int write_forces, bool with_self, hipStream_t s)
{
    const int mesh = h->nF > 0 ? (h->any_large ? 2 : 1) : 0;
    const StateC in = h->state(in_buf);
    const StateM out = h->state(in_buf ^ 1);
    if (has_contact_finish(h, p) && h->fin_batch) {
        // batched small-scene finishing: (environments of this chain, slots) workgroups of 256 threads; a slot takes 16 records of ITS environment at a time
        const dim3 g((unsigned)p.ne, (unsigned)fin_batch_slots(with_self));
        if (with_self) hipLaunchKernelGGL((k_contact_finish_batch<true>), g, dim3(256), 0, s, p, in, out, step, write_forces);
        else hipLaunchKernelGGL((k_contact_finish_batch<false>), g, dim3(256), 0, s, p, in, out, step, write_forces);
    } else if (has_contact_finish(h, p)) {
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
bool pf_flavour(const R2SPhys* h, const PhysDev& p) { return r2s_flavour::fl_pf(h->flavour_caps(), p.mesh_defer); }
// finishing workgroups at the head of a launch: enough for the lists of a batch in contact without a second round (a workgroup strides
// over its list if there is more), few enough not to stand between the launch and its fused blocks — every workgroup of the launch
// holds the fused role's LDS window, so an idle finisher costs a block's slot for the microsecond it takes to read an empty list
int pf_head_size(const R2SPhys* h, int ne, bool with_self)
{
    const int mesh = h->nF > 0 ? (h->any_large ? 2 : 1) : 0;
    // small scenes: 32 workgroups per environment (deferred queries from the front, candidate particles — eight to a workgroup — from the back;
    // 16 and 24 ran a second round in the headline's grasp: 25.8 / 24.4 vs 22.8 us), 16 while no particle has candidates
    int n = mesh == 2 ? ne * std::max(16, 256 / std::max(1, ne)) : h->fin_batch ? fin_batch_slots(with_self) * ne : std::min(1024, (with_self ? 32 : 16) * ne);
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
    if (with_self) { if (mesh == 2) R2S_PF(true, 2, 2); else if (h->fin_batch) R2S_PF(true, 1, 4); else if (small) R2S_PF(true, 1, 3); else R2S_PF(true, 1, 2); }
    else { if (mesh == 2) R2S_PF(false, 2, 2); else if (h->fin_batch) R2S_PF(false, 1, 4); else if (small) R2S_PF(false, 1, 3); else R2S_PF(false, 1, 2); }
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

// The env step's flavour that runs as one resident launch: a small batch, no deferred mesh queries (candidates: the SELF templates, round 5).
bool resident_flavour(const R2SPhys* h, bool with_self, int mesh_defer, int n) { return r2s_flavour::fl_resident(h->flavour_caps(), with_self, mesh_defer, n); }
constexpr int RES_MAX_ITEMS = 256; // (block, env) work items of a resident launch: one 512-thread workgroup per CU (two wavefronts per SIMD, each
                                   // with its 2 + 2 adjacency groups in registers), all on the chip at once

int enqueue_steps(R2SPhys* h, int first, int n, int start_buf, bool with_self, hipStream_t s, int e0 = 0, int ne = -1, bool zero_forces = true, int chain_id = 0,
                  int split_at = 0, const std::function<int()>* on_split = nullptr)
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
    if (resident_flavour(h, with_self, p.mesh_defer, n)) {
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
        // (the claim lines are the handle's: one resident launch at a time — a forced second chain runs without servers; workgroups go to
        // the XCDs round-robin and every XCD must hold its share at once: 8 * cb block workgroups + the servers <= CUs: fl_n_srv)
        const int n_srv = r2s_flavour::fl_n_srv(h->flavour_caps(), with_self, h->self_srv, n);
        if (n_srv > 0) {
            if (with_self) p.srv_own = 0; // next to the self-collision flavour the pairs only ANSWER queries (a request per particle and substep, carrying
                                          // the velocity after the impulses): an owned particle would have to take part in the candidates' hand-off itself
            p.srv_quad = h->srv_quad_for(n_srv); p.srv_slots = (p.srv_quad ? 2 : 4) * n_srv;
            p.srv_low = (int)((int64_t)p.srv_slots * h->srv_low_pct / 100);
            hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((SRV_DBG_OFF / 4 + 255) / 256)), dim3(256), 0, s, (float*)h->d_srv_claim, (size_t)SRV_DBG_OFF / 4); // claims and control words
            const size_t rw = (size_t)(SRV_REC / 4) * ne * h->N;
            hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((rw + 255) / 256)), dim3(256), 0, s, (float*)h->d_srv_rr + (size_t)(SRV_REC / 4) * e0 * h->N, rw);
        }
        const dim3 grid(8u * (unsigned)p.cb + (unsigned)n_srv);
        const StateC in = h->state(start_buf);
        const StateM out = h->state(start_buf ^ 1);
        const bool with_mesh = h->nF > 0;
        if (with_self) { // candidates' {x0, v} records: tags of the previous launch must not match
            hipLaunchKernelGGL(k_zero_f32, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, (float*)h->d_vx + (size_t)24 * e0 * xn, words);
            if (with_mesh && n_srv > 0) hipLaunchKernelGGL((k_steps_resident<512, true, 1, true>), grid, dim3(RES_THREADS), 0, s, p, in, out, first, n, 1);
            else if (with_mesh) hipLaunchKernelGGL((k_steps_resident<512, true, 1, false>), grid, dim3(RES_THREADS), 0, s, p, in, out, first, n, 1);
            else hipLaunchKernelGGL((k_steps_resident<512, true, 0>), grid, dim3(RES_THREADS), 0, s, p, in, out, first, n, 1);
        } else if (with_mesh) hipLaunchKernelGGL((k_steps_resident<512, false, 1>), grid, dim3(RES_THREADS), 0, s, p, in, out, first, n, 1);
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
        if (on_split && k == split_at && k > 0) { if (const int rc = (*on_split)()) return rc; } // graph capture: the launches from here on go to the tail graph
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
        for (int v = 0; v < 12; ++v) {
            if (h->graph_exec[c][v]) (void)hipGraphExecDestroy(h->graph_exec[c][v]);
            if (h->graph[c][v]) (void)hipGraphDestroy(h->graph[c][v]);
            if (h->graph_exec_tail[c][v]) (void)hipGraphExecDestroy(h->graph_exec_tail[c][v]);
            if (h->graph_tail[c][v]) (void)hipGraphDestroy(h->graph_tail[c][v]);
            h->graph_exec[c][v] = nullptr; h->graph[c][v] = nullptr; h->graph_exec_tail[c][v] = nullptr; h->graph_tail[c][v] = nullptr;
        }
}

void drop_graph_fwd(R2SPhys* h) { drop_graph(h); }

// Environments are independent, so the env step runs as `chains` parallel kernel chains over disjoint environment ranges:
// while one chain's workgroups stage their windows (memory phase, VALU idle) or sit in the launch gap between two substeps,
// another chain's are in the gather (VALU phase), and one chain's finishing kernel runs next to the others' fused kernels.
// Each chain is captured into its own graph.
// The captured flavours: {no candidates, candidates} x {queries in place / resident, deferred} x parity of the state buffer, and the
// resident self-collision flavour with query servers in the launch (self_srv: only while queries are being needed — idle servers and
// their hand-off code cost that flavour 0.5 us per substep)
int graph_slot(const R2SPhys* h, int variant, int start_buf)
{
    return r2s_flavour::fl_graph_slot(variant, h->mesh_defer, h->self_srv) + (start_buf & 1);
}

int capture_graph(R2SPhys* h, int variant, int start_buf)
{
    const int slot = graph_slot(h, variant, start_buf);
    const int chains = h->chains();
    for (int c = 0; c < chains; ++c) {
        if (h->graph_exec[c][slot]) (void)hipGraphExecDestroy(h->graph_exec[c][slot]);
        if (h->graph[c][slot]) (void)hipGraphDestroy(h->graph[c][slot]);
        if (h->graph_exec_tail[c][slot]) (void)hipGraphExecDestroy(h->graph_exec_tail[c][slot]);
        if (h->graph_tail[c][slot]) (void)hipGraphDestroy(h->graph_tail[c][slot]);
        h->graph_exec[c][slot] = nullptr; h->graph[c][slot] = nullptr; h->graph_exec_tail[c][slot] = nullptr; h->graph_tail[c][slot] = nullptr;
        const int e0 = (int)((int64_t)h->E * c / chains), e1 = (int)((int64_t)h->E * (c + 1) / chains);
        hipStream_t cs;
        R2S_HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        if (const hipError_t eb = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal); eb != hipSuccess) { (void)hipStreamDestroy(cs); R2S_HIP_TRY(eb); }
        // Head and tail (round 6): hipGraphLaunch of a 667-node chain takes 0.2 ms of host time and the chain starts when the call is through —
        // in the closed loop (the policy waits for the observation before the next step is launched) the four chains started 0.2 ms apart and the
        // step paid 0.6 ms for it.  Each chain is therefore two graphs: a head of HEAD substeps that is launched in microseconds — all chains' heads
        // first — and the tail, enqueued behind it on the same stream while the heads run.  The same launches in the same order on the same stream.
        hipGraph_t g_head = nullptr;
        const std::function<int()> split = [&]() -> int {
            R2S_HIP_TRY(hipStreamEndCapture(cs, &g_head));
            R2S_HIP_TRY(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            return R2S_OK;
        };
        const int head = h->graph_head < h->prm.num_substeps ? h->graph_head : 0;
        const int rc = enqueue_steps(h, 0, h->prm.num_substeps, start_buf, variant == 1, cs, e0, e1 - e0, true, c, head, head > 0 ? &split : nullptr);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(cs, &g);
        (void)hipStreamDestroy(cs);
        if (rc) { if (g) (void)hipGraphDestroy(g); if (g_head) (void)hipGraphDestroy(g_head); return rc; }
        R2S_HIP_TRY(e);
        if (g_head) { // the capture was split: `g` is the tail
            h->graph[c][slot] = g_head; h->graph_tail[c][slot] = g;
            R2S_HIP_TRY(hipGraphInstantiate(&h->graph_exec[c][slot], g_head, nullptr, nullptr, 0));
            R2S_HIP_TRY(hipGraphInstantiate(&h->graph_exec_tail[c][slot], g, nullptr, nullptr, 0));
        } else {
            h->graph[c][slot] = g;
            R2S_HIP_TRY(hipGraphInstantiate(&h->graph_exec[c][slot], g, nullptr, nullptr, 0));
        }
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
    for (int c = 1; c < chains; ++c) { // every chain's head first (or its only graph) ...
        hipStream_t sc = chain_side_stream(c);
        if (!sc) return R2S_ERR_HIP;
        if (!h->chain_join[c]) R2S_HIP_TRY(hipEventCreateWithFlags(&h->chain_join[c], hipEventDisableTiming));
        R2S_HIP_TRY(hipStreamWaitEvent(sc, h->chain_fork, 0));
        R2S_HIP_TRY(hipGraphLaunch(h->graph_exec[c][slot], sc));
    }
    R2S_HIP_TRY(hipGraphLaunch(h->graph_exec[0][slot], s));
    for (int c = 1; c < chains; ++c) { // ... then the tails, behind their heads on the same streams
        hipStream_t sc = chain_side_stream(c);
        if (h->graph_exec_tail[c][slot]) R2S_HIP_TRY(hipGraphLaunch(h->graph_exec_tail[c][slot], sc));
        R2S_HIP_TRY(hipEventRecord(h->chain_join[c], sc));
    }
    if (h->graph_exec_tail[0][slot]) R2S_HIP_TRY(hipGraphLaunch(h->graph_exec_tail[0][slot], s));
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

// Small scenes with batched finishing: the triangles of every (environment, substep) as the finishing workgroups stage them, written once per env
// step behind the kinematics — a finishing workgroup then stages with ONE coalesced round trip instead of two dependent ones (face -> vertex ids ->
// positions), 0.8 us less on the path every waiting block of the next launch sits behind.  171 MB per env step of the headline (32 x 667 x 100 x 80 B).
__global__ void __launch_bounds__(256) k_tri_pre(const PhysDev p, float4* __restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t tot = (size_t)p.E * p.n_sub * p.nF;
    if (t >= tot) return;
    const int f = (int)(t % p.nF);
    const size_t es = t / p.nF;
    const int e = (int)(es / p.n_sub), step = (int)(es % p.n_sub);
    const f3 a = mesh_vertex(p, e, step, p.faces[3 * f]), b = mesh_vertex(p, e, step, p.faces[3 * f + 1]), c = mesh_vertex(p, e, step, p.faces[3 * f + 2]);
    int m = 0;
    for (int k = 1; k < p.n_mesh; ++k) m += f >= p.mesh_face_off[k] ? 1 : 0;
    float4* o = out + es * 5 * p.nF + f;
    o[0] = make_float4(a.x, a.y, a.z, b.x);
    o[(size_t)p.nF] = make_float4(b.y, b.z, c.x, c.y);
    o[(size_t)2 * p.nF] = make_float4(c.z, __int_as_float(p.mesh_map[f]), __int_as_float(p.face_map[f]), __int_as_float(m));
    o[(size_t)3 * p.nF] = make_float4(fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z)), fmaxf(a.x, fmaxf(b.x, c.x)));
    o[(size_t)4 * p.nF] = make_float4(fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z)), 0.f, 0.f);
}
void update_tri_pre(R2SPhys* h, hipStream_t s)
{
    if (!h->d_tri_pre) return;
    const size_t tot = (size_t)h->E * h->prm.num_substeps * h->nF;
    hipLaunchKernelGGL(k_tri_pre, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, h->dev(), h->d_tri_pre);
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

// The sticky fault word of an earlier env step as r2s_last_error() text (cnt: the 16 words of the step's ring slot).
void report_fault(const int* cnt)
{
    if (cnt[1] >= 2 && cnt[1] <= 7) {
        char buf[640];
        const int* w = cnt + 4;
        snprintf(buf, sizeof buf, "resident stepper: a workgroup waited for %s beyond the poll limit (the launch was not resident at once, or the device "
                 "is shared with a kernel that never ends); the state is invalid [first fault: code %d, work item %d, substep %d of the launch, context %d %d %d %d %d %d]",
                 cnt[1] == 2 ? "a neighbour block's substep" : cnt[1] == 5 ? "a neighbour's record (server pair)" : cnt[1] == 6 ? "a particle finished by the head of its launch" : cnt[1] == 7 ? "a self-collision partner's record" : "a mesh-query server's result", w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7], w[8]);
        r2s::set_last_error_msg(buf);
    } else if (cnt[1] == 8) {
        r2s::set_last_error_msg("a per-environment list of particles left to the finishing code overflowed (more than N entries in one substep: cannot happen in a sound launch); the state is invalid");
    } else
        r2s::set_last_error_msg("a self-collision impulse changed a particle's velocity by more than 40 m/s within one substep: its mesh-contact "
                                "test (widened by 2 mm) may have been skipped where the reference applies it (unsupported)");
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
        if (const char* ev = getenv("R2S_GRAPH_HEAD")) h->graph_head = std::max(0, atoi(ev)); // substeps in a chain's head graph (0: one graph per chain)
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
        {   // orientation of each mesh: signed volume of its faces about the origin at construction (rigid motion keeps it)
            std::vector<int> inward(h->n_mesh, 0);
            for (int m = 0; m < h->n_mesh; ++m) {
                double vol = 0.0;
                for (int f = foff[m]; f < foff[m + 1]; ++f) {
                    const int ia = faces[3 * f], ib = faces[3 * f + 1], ic = faces[3 * f + 2];
                    const double a[3] = {vtx(ia, 0), vtx(ia, 1), vtx(ia, 2)}, b[3] = {vtx(ib, 0), vtx(ib, 1), vtx(ib, 2)}, c[3] = {vtx(ic, 0), vtx(ic, 1), vtx(ic, 2)};
                    vol += a[0] * (b[1] * c[2] - b[2] * c[1]) + a[1] * (b[2] * c[0] - b[0] * c[2]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
                }
                inward[m] = vol < 0.0 ? 1 : 0;
            }
            TRY(dev_alloc(&h->d_mesh_inward, h->n_mesh));
            TRY(upload(h->d_mesh_inward, inward.data(), inward.size(), s));
        }
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
        h->fin_batch = !h->any_large && h->nF <= FB_MAX_F && h->n_mesh <= FB_MAX_MESH && !(getenv("R2S_FIN_BATCH") && atoi(getenv("R2S_FIN_BATCH")) == 0);
        if (h->fin_batch && !(getenv("R2S_TRI_PRE") && atoi(getenv("R2S_TRI_PRE")) == 0) && (size_t)E * h->prm.num_substeps * h->nF * 80 < ((size_t)4 << 30)) {
            TRY(dev_alloc(&h->d_tri_pre, (size_t)5 * E * h->prm.num_substeps * h->nF));
            update_tri_pre(h, s);
        }
        if (h->any_large || h->fin_batch) {
            TRY(dev_alloc(&h->d_mesh_rec, (size_t)4 * E * N));
            R2S_HIP_TRY(hipMemsetAsync(h->d_mesh_rec, 0, sizeof(int4) * (size_t)4 * E * N, s));
            if (h->any_large && h->n_cl < 4094 && getenv("R2S_NO_MQ_HINT") == nullptr) { // the hint rides in 12 bits of a record word
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
            const void* kernels[5] = {(const void*)k_steps_resident<512, false, 0>, (const void*)k_steps_resident<512, false, 1>,
                                      (const void*)k_steps_resident<512, true, 0>, (const void*)k_steps_resident<512, true, 1, false>,
                                      (const void*)k_steps_resident<512, true, 1, true>};
            for (int k = 0; k < 5; ++k) {
                if ((k == 1 || k >= 3) && h->nF == 0) continue;                 // (the mesh templates of a scene without meshes are never launched)
                if (k >= 2 && !h->prm.self_collision) continue;     // (nor the self-collision flavour of a handle without it)
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
        h->lag = h->resident_ok ? 1 : R2SPhys::LAG;
        if (const char* ev = getenv("R2S_FLAVOUR_LAG")) h->lag = std::max(1, std::min(R2SPhys::RING - 1, atoi(ev)));
        if (h->resident_ok) {
            const size_t xn = ((size_t)N + 7) & ~(size_t)7;
            TRY(dev_alloc((char**)&h->d_xch, (size_t)96 * E * xn));
            R2S_HIP_TRY(hipMemsetAsync(h->d_xch, 0, (size_t)96 * E * xn, s));
            if (const char* ev = getenv("R2S_RES_SELF")) h->res_self = atoi(ev) != 0;
            if (const char* ev = getenv("R2S_RES_SELF_SRV")) h->res_self_srv = std::max(0, std::min(2, atoi(ev))); // 2: servers in every launch of the flavour (tests)
            if (h->prm.self_collision) {
                TRY(dev_alloc((char**)&h->d_vx, (size_t)96 * E * xn));
                R2S_HIP_TRY(hipMemsetAsync(h->d_vx, 0, (size_t)96 * E * xn, s));
            }
            // query servers ride in the launch when every mesh is small enough for k_contact_finish<3>'s code (triangles in registers, one per
            // lane in two wavefronts) and the blocks leave CUs free; R2S_RES_SERVERS=0: queries in place / per-substep flavour as in round 3
            bool pref = true;
            if (const char* ev = getenv("R2S_RES_SERVERS")) pref = atoi(ev) != 0;
            if (const char* ev = getenv("R2S_RES_SPIN_LIMIT")) h->spin_limit = (unsigned)std::max(1024, atoi(ev));
            if (const char* ev = getenv("R2S_RES_SRV_LOW")) h->srv_low_pct = std::max(1, std::min(100, atoi(ev)));
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
                if (variant == 1 && defer == 0 && h->resident_ok && h->srv_ok && h->res_self && h->res_self_srv && h->d_vx && h->chains() == 1) {
                    h->self_srv = 1; // ... and the resident self-collision flavour with query servers in the launch
                    TRY(capture_graph(h, variant, par));
                    h->self_srv = 0;
                }
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
                    h->d_faces, h->d_face_orig, h->d_cl_box, h->d_mesh_kind, h->d_mesh_inward, h->d_tri_pre, h->d_mesh_xf, h->d_xf_mesh, h->d_xf_ref,
                    h->d_xf, h->d_rest_pts, h->d_pnorm, h->d_tri_rest, h->d_cl_info, h->d_sup_info, h->d_sup_box, h->d_small_mesh, h->d_xf_rest_box, h->d_rigid_err, h->d_mesh_map, h->d_face_map, h->d_mesh_face_off, h->d_mesh_vert_off, h->d_mesh_pts, h->d_interp, h->d_center,
                    h->d_dyn_vel, h->d_dyn_omega, h->d_aabb_dyn, h->d_aabb_static, h->d_coll_forces,
                    h->d_eef_table, h->d_eef_open, h->d_eef_grasped, h->d_eef_has, h->d_eef_need, h->d_eef_rel0, h->d_eef_delta, h->d_hit_cnt, h->d_xch, h->d_vx,
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

int r2s_phys_set_static_mesh_points(R2SPhys* h, const float* pts, int32_t n_static_vertices, const int32_t* env_mask, r2s_stream_t stream_)
{
    r2s::set_last_error_msg("");
    if (!h || !pts || h->nF == 0) return R2S_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream_;
    const int n = h->nV - h->n_dyn_pts;
    if (n <= 0) return R2S_ERR_INVALID;
    if (n_static_vertices != n) { // the array covers EVERY static mesh, in the order of the constructor: anything else would be read past its end
        char buf[200];
        snprintf(buf, sizeof buf, "r2s_phys_set_static_mesh_points: %d vertices per environment handed in, the static meshes have %d", (int)n_static_vertices, n);
        r2s::set_last_error_msg(buf);
        return R2S_ERR_INVALID;
    }
    for (int m = h->n_dyn_mesh; m < h->n_mesh; ++m)
        if (h->h_mesh_kind[m] & 1) { // its triangles live in a rest-frame table with a box hierarchy built at create: it cannot move per environment
            r2s::set_last_error_msg("r2s_phys_set_static_mesh_points: a static collision mesh with more than 256 faces cannot be re-posed (unsupported)");
            return R2S_ERR_INVALID;
        }
    hipLaunchKernelGGL(k_set_static_pts, dim3((unsigned)((3 * n + 255) / 256), (unsigned)h->E), dim3(256), 0, s, h->E, n, h->nV, h->n_dyn_pts, env_mask, pts, h->d_mesh_pts);
    const int ns = h->n_mesh - h->n_dyn_mesh, tot = h->E * ns;
    hipLaunchKernelGGL(k_mesh_aabb_static, dim3((tot + 255) / 256), dim3(256), 0, s, h->E, ns, h->n_dyn_mesh, h->nV, h->d_mesh_vert_off, h->d_mesh_pts, h->d_aabb_static);
    update_tri_pre(h, s);
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
    update_tri_pre(h, s);
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
    update_tri_pre(h, s);
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
    // the counters this step's flavour follows from: those of step (step_no - LAG), waited for (long landed in any real loop)
    int cnt[16] = {0};
    bool have_cnt = false;
    if (h->step_no >= (uint64_t)h->lag) {
        const int k = (int)((h->step_no - h->lag) % R2SPhys::RING);
        if (h->ring_pending[k]) { R2S_HIP_TRY(hipEventSynchronize(h->ring_ev[k])); h->ring_pending[k] = false; }
        memcpy(cnt, h->h_ring + 16 * k, sizeof cnt);
        if (h->ring_stale_fault[k]) cnt[1] = 0;
        have_cnt = true;
    }
    // the sticky fault word is looked for in the NEWER copies too when they have landed: it only ends the run, it picks no flavour
    for (uint64_t back = 1; back < (uint64_t)h->lag && back <= h->step_no && cnt[1] == 0; ++back) {
        const int k = (int)((h->step_no - back) % R2SPhys::RING);
        if (h->ring_pending[k] && hipEventQuery(h->ring_ev[k]) == hipSuccess) h->ring_pending[k] = false;
        if (!h->ring_pending[k] && !h->ring_stale_fault[k] && h->h_ring[16 * k + 1] != 0) { cnt[1] = h->h_ring[16 * k + 1]; memcpy(cnt + 4, h->h_ring + 16 * k + 4, 12 * sizeof(int)); }
    }
    if (cnt[1] != 0) { // the sticky fault word of an earlier step
        report_fault(cnt);
        return R2S_ERR_INVALID;
    }
    // Which flavour: a pure function of those counters, the handle's capabilities and the switches (physics_flavour.h: the rules are
    // written out there; tests/test_flavour_matrix.py enumerates them).  Nothing below decides anything.
    R2SFlavourIn fin = h->flavour_caps();
    fin.have_counters = have_cnt; fin.near_mesh = cnt[0]; fin.query_needed = cnt[2]; fin.servers_ran_out = cnt[3];
    fin.n_substeps = n; fin.full_step = use_graph;
    R2SFlavourOut& fl = h->last_pick;
    r2s_flavour::pick_flavour(fin, fl);
    const int variant = fl.variant;
    h->mesh_defer = fl.mesh_defer; h->self_srv = fl.self_srv; h->srv_exhausted = fl.srv_exhausted != 0;
    h->last_flavour[0] = variant | (h->split_ok ? 2 : 0); h->last_flavour[1] = fl.mesh;
    h->last_flavour[2] = fl.resident ? 2 : (fl.pf ? 3 : fl.mesh_defer); // 2 = the resident launch (never with deferred queries), 3 = deferred, finishers at the head of the next launch
    h->last_flavour[3] = fl.resident ? (1 | (fl.n_srv << 8) | (fl.srv_own << 20) | (fl.srv_quad << 21)) : fl.chains;
    const bool resident = fl.resident != 0;
    int gate_dev = -1;
    if (resident) { int rcg = resident_enter(s, &gate_dev); if (rcg) return rcg; }
    struct GateLeave { hipStream_t s; int dev; ~GateLeave() { if (dev >= 0) (void)resident_leave(s, dev); } } gate_leave{s, gate_dev};
    if (use_graph) {
        // every flavour was captured at construction (capture_all); only set_params / set_tuning drop them
        const int slot = graph_slot(h, variant, h->cur);
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

int r2s_phys_debug_pick_flavour(const R2SFlavourIn* in, R2SFlavourOut* out)
{
    if (!in || !out) return R2S_ERR_INVALID;
    r2s_flavour::pick_flavour(*in, *out);
    return R2S_OK;
}

int r2s_phys_debug_flavour_input(R2SPhys* h, R2SFlavourIn* out)
{
    if (!h || !out) return R2S_ERR_INVALID;
    *out = h->flavour_caps();
    return R2S_OK;
}

int r2s_phys_last_flavour_ex(R2SPhys* h, R2SFlavourOut* out)
{
    if (!h || !out) return R2S_ERR_INVALID;
    *out = h->last_pick;
    return R2S_OK;
}

int r2s_phys_check_fault(R2SPhys* h, r2s_stream_t stream_)
{
    r2s::set_last_error_msg("");
    if (!h) return R2S_ERR_INVALID;
    if (!h->d_mesh_total) return R2S_OK;
    int cnt[16] = {0};
    R2S_HIP_TRY(hipMemcpyAsync(cnt, h->d_mesh_total, sizeof cnt, hipMemcpyDeviceToHost, (hipStream_t)stream_));
    R2S_HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));
    if (cnt[1] == 0) return R2S_OK;
    report_fault(cnt);
    return R2S_ERR_INVALID;
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
