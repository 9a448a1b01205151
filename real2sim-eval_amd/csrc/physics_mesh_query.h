// physics_mesh_query.h — part of physics.hip's ONE translation unit (included there, inside its anonymous namespace, in this order: physics_mesh_query.h,
// physics_substep.h, physics_resident.h, physics_finish.h, physics_aux.h); not a stand-alone header.  Round 5 split the 4 800-line file by section;
// the token stream the compiler sees is unchanged.
// Here: mesh queries: closest point on a triangle, wavefront reductions, the workgroup-cooperative query over the box hierarchy of a large mesh (mesh_query_block), the per-lane query (mesh_query_lane), the small-scene query with the triangles in registers (mesh_query_regs) and its unit barrier.

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

