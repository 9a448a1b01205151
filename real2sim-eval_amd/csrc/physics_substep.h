// physics_substep.h — part of physics.hip's ONE translation unit (included there, inside its anonymous namespace, in this order: physics_mesh_query.h,
// physics_substep.h, physics_resident.h, physics_finish.h, physics_aux.h); not a stand-alone header.  Round 5 split the 4 800-line file by section;
// the token stream the compiler sees is unchanged.
// Here: the fused substep: hand-off helpers of the finishers at the head of the next launch (PF_SENT, result lines), the spring gather from the LDS window, everything after the velocity update (finish_wave: mesh collision, ground, store), substep_body and k_substep.

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
#ifdef R2S_WIN16 // experiment: a 16-byte {x, y, z, vz} plane + an 8-byte {vx, vy} plane: ds_read_b128 + ds_read_b64 per slot instead of three ds_read_b64
        typedef float v4f_ __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) const v4f_ lds_f4;
        const v4f_ xz = *(lds_f4*)(win + 2u * off[u]);
        const v2f vxy = *(lds_f2*)(win + off[u] + 2 * (RCAP * 8 + 8));
        spring_term((v2f){xz.x, xz.y}, xz.z, vxy, xz.w, xi, vi, k[u], a[u], p.dashpot, fxy, fz);
#else
        const v2f xy = *(lds_f2*)(win + off[u]);
        const v2f zz = *(lds_f2*)(win + off[u] + PLANE1<RCAP>());
        const v2f vxy = *(lds_f2*)(win + off[u] + PLANE2<RCAP>());
        spring_term(xy, zz.x, vxy, zz.y, xi, vi, k[u], a[u], p.dashpot, fxy, fz);
#endif
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
__device__ __forceinline__ int mesh_rec_reserve(const PhysDev& p, int e, int step)
{
    // One atomic per WAVEFRONT, not per listed lane (round 6): the lanes of a fused block share their environment, but `e` lives in a vector register,
    // so the compiler cannot see the address is uniform and issued a returning atomic per lane — 20 serialised round trips to one L2 word in a block
    // under a pad.  Lanes of different environments in one wavefront (k_self_finish's groups) keep the per-lane form.
    int slot;
    {
        const unsigned long long act = __builtin_amdgcn_ballot_w64(true); // the lanes that push (this code runs under their branch)
        const int e0 = __builtin_amdgcn_readfirstlane(e);
        if (__builtin_amdgcn_ballot_w64(e != e0) == 0ull) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
            int base = 0;
            if (rank == 0) base = atomicAdd(p.rec_cnt + (size_t)e0 * p.n_sub + step, (int)__builtin_popcountll(act));
            slot = __builtin_amdgcn_readfirstlane(base) + rank;
        } else
            slot = atomicAdd(p.rec_cnt + (size_t)e * p.n_sub + step, 1);
    }
    return slot;
}
__device__ __forceinline__ void mesh_rec_write(const PhysDev& p, int e, int step, int slot, int i, int ncand, f3 x0, f3 v)
{
    int4* r = p.mesh_rec + 2 * (par_off(p, step) + (size_t)e * p.N + slot);
    const int hint = p.mq_hint ? p.mq_hint[(size_t)e * p.N + i] : -1; // the cluster of its closest face one substep ago rides in the record (bits 19..30)
    r[0] = make_int4(ncand | ((hint + 1) << 19), ncand > 0 ? (i | (int)0x80000000) : i, __float_as_int(x0.x), __float_as_int(x0.y));
    r[1] = make_int4(__float_as_int(x0.z), __float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z));
}
__device__ __forceinline__ bool mesh_rec_push(const PhysDev& p, int e, int step, int i, int ncand, f3 x0, f3 v)
{
    const int slot = mesh_rec_reserve(p, e, step);
    if (slot >= p.N) return false;
    mesh_rec_write(p, e, step, slot, i, ncand, x0, v);
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
            } else if (need && p.mesh_defer && p.mesh_rec) { // small scene, batched finishing (round 6): through the per-environment records as well
                if (mesh_rec_push(p, e, step, i, 0, x0, v)) { fin = false; need = false; }
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
// EXTWIN: the LDS window is the caller's (k_substep_pf with the batched finishers: the two roles of that launch share one allocation)
template <int B, int RCAP, bool SELF, int MESH, bool PF = false, bool EXTWIN = false>
__device__ __forceinline__ void substep_body(const PhysDev& p, const StateC xv_in, const StateM xv_out, int step,
                                             int write_forces, int bid, __attribute__((address_space(3))) v2f* win_ext = nullptr)
{
    static_assert(B % SLICE == 0 && RCAP >= B && RCAP * 8 <= 65536, "window offsets are u16 bytes");
    __shared__ __attribute__((aligned(16))) v2f win_own[EXTWIN ? 1 : 3 * (RCAP + 1)]; // planes xy | (z, vz) | vxy, 24 B per record (+ 1 pad each)
    __attribute__((address_space(3))) v2f* const win_s = EXTWIN ? win_ext : (__attribute__((address_space(3))) v2f*)win_own;
    const int xcd = bid & 7, q = bid >> 3;
    const int item = xcd * p.cb + q;
    if (q >= p.cb || item >= p.nb * p.ne) return; // whole workgroup
    // (round 5, measured and removed: XCD x owning the blocks b = x mod 8 instead of a contiguous range — so that the blocks of a contact
    // region, neighbours in Morton order, and with them the blocks that wait for a finisher in k_substep_pf, spread over all eight XCDs:
    // 22.8 vs 22.2 us per contact substep of the headline, 17.8 vs 17.0 free: the halo locality of contiguous ranges is worth more)
#ifndef R2S_NO_SCALAR_ENV
    // block and environment are the same in every lane, but the integer division leaves them in vector registers and everything indexed by them (mesh boxes,
    // per-environment lists, the state base) was addressed per lane: named scalars, so those become scalar loads / scalar address arithmetic (round 6)
    const int b = __builtin_amdgcn_readfirstlane(item / p.ne), e = __builtin_amdgcn_readfirstlane(p.e0 + (item - b * p.ne));
#else
    const int b = item / p.ne, e = p.e0 + (item - b * p.ne);
#endif
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
#define R2S_STAGE_BATCH 4   // (round 6: all four rounds of the <256,1024> layout in flight — 71 VGPRs, still six wavefronts per SIMD; a block that waits for a finisher then
                            // has nothing left to load behind the wait: 31.7 -> 31.2 us per batched substep in the held grasp, 23.5 -> 22.9 while closing, free unchanged)
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
#ifdef R2S_WIN16
                win_s[2 * r] = qa[k];
                win_s[2 * r + 1] = qb[k];
                win_s[2 * (RCAP + 1) + r] = qc[k];
#else
                win_s[r] = qa[k];
                win_s[RCAP + 1 + r] = qb[k];
                win_s[2 * (RCAP + 1) + r] = qc[k];
#endif
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
    // Per-environment record lists (large-mesh scenes; small scenes with batched finishing, round 6) in a flavour that defers: the two kinds of
    // listed particles — with candidates (tagged, test widened by 2 mm) and without — are tested here and pushed TOGETHER, one atomic per
    // wavefront for both (they were two dependent atomic round trips, the second inside finish_wave: the blocks under a pad ended 3 us after the
    // others and with them the launch); what is left for finish_wave is "no mesh in reach" (NEED 2: advance, ground, store).
    const bool rec_mode = PF && MESH != 0 && p.mesh_rec != nullptr && (p.mesh_defer || MESH == 2); // uniform; in the launches with the finishers at their head only (compiled
                                                                                                   // into the plain k_substep too it cost the free flavour 8 VGPRs and a spill)
    int tag_ncand = 0;
    if (SELF) {
        const int ncand = valid ? p.coll_num[eb + i] : 0;
        if (ncand > 0) {
            const size_t po = par_off(p, step);
            p.vbc[po + eb + i] = make_float4(v.x, v.y, v.z, 0.f);
            p.xbc[po + eb + i] = make_float4(x0.x, x0.y, x0.z, 0.f);
            fin = false; // finished by k_self_finish / k_contact_finish
            if (rec_mode) tag_ncand = ncand;
            else if (MESH != 0 && (p.mesh_defer || MESH == 2)) {
                // Will it also need a mesh query?  Its velocity is not final (the impulses come later), so the test is widened
                // by 2 mm (= 40 m/s of velocity change in one substep); over-inclusion is harmless, the query itself is exact.
                // Such a particle goes to the mesh list TAGGED: k_contact_finish applies its impulses and queries in one go.
                bool near;
                if (mesh_need(p, e, step, x0 + v * p.dt, 0.002f, near)) {
                    p.fault[1] = 1;
                    if (MESH == 2 || p.mesh_rec) {
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
    bool done;
    if (rec_mode) {
        bool near = false, need = false;
        if (fin || tag_ncand > 0) need = mesh_need(p, e, step, x0 + v * p.dt, tag_ncand > 0 ? 0.002f : 0.f, near);
        const unsigned long long nm = __builtin_amdgcn_ballot_w64(near);
        if (nm && lane == __builtin_ctzll(nm)) p.mesh_cnt[p.n_sub] = 1;   // "anything near?" (the host picks a later step's flavour from it)
        const unsigned long long qm = __builtin_amdgcn_ballot_w64(need);
        if (qm && lane == __builtin_ctzll(qm)) p.fault[1] = 1;          // "a query was needed"
        // the slot is reserved (one atomic per wavefront), the rest of the block's particles are finished WHILE it is on its way, the record is written
        // behind that: the round trip to the counter was 1 - 1.5 us in series in the blocks under a pad, the ones that end the launch.  A particle is
        // listed at most once per substep and the list has N slots: a slot beyond them cannot happen — if it does, the run ends with a fault
        int slot = 0;
        if (need) { slot = mesh_rec_reserve(p, e, step); fin = false; }
        done = finish_wave<MESH, false, 2>(p, e, i, eb, step, write_forces, x0, v, fin, xv_out, nullptr, nullptr, nullptr, nullptr, true, nullptr R2S_QP_ARG);
        if (need) {
            if (slot < p.N) {
                mesh_rec_write(p, e, step, slot, i, tag_ncand, x0, v);
                if (tag_ncand > 0) p.cand_mark[par_off(p, step) + eb + i] = step + 1;
            } else
                resident_fault(p, 8, item, step, (unsigned)(eb + i), (unsigned)slot, 0u, 0u, 0u, 0u);
        }
    } else
        done = finish_wave<MESH, MESH != 0>(p, e, i, eb, step, write_forces, x0, v, fin, xv_out, nullptr, nullptr, nullptr, nullptr, true, nullptr R2S_QP_ARG);
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
