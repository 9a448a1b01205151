// physics_resident.h — part of physics.hip's ONE translation unit (included there, inside its anonymous namespace, in this order: physics_mesh_query.h,
// physics_substep.h, physics_resident.h, physics_finish.h, physics_aux.h); not a stand-alone header.  Round 5 split the 4 800-line file by section;
// the token stream the compiler sees is unchanged.
// Here: the resident stepper of small batches: every substep of an env step in ONE launch (k_steps_resident) and its mesh-query server workgroups (resident_server).

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
// candidates (SELF with n_steps = 1; with n_steps > 1 see `self_res` below) — keep their finishing kernels and a launch per substep, but a block's springs are
// still shared by eight wavefronts instead of walked by one (k_substep<64,512,..>: 8.0 us per substep of the rope, this: see DESIGN §4).
// SRV: the launch may carry mesh-query servers (small scenes).  The self-collision flavour has both forms: the servers' hand-off code costs
// it a spilled register and 0.3 us per substep (rope folded onto itself: 5.55 vs 5.85), so it only pays for them while queries are needed.
template <int RCAP, bool SELF, int MESH, bool SRV = (MESH == 1 && !SELF)>
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
        if (SRV) resident_server(p, first, n_steps, write_forces_last);
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
    const bool srv_on = SRV && p.srv_slots > 0 && n_steps > 1; // (SELF: only when the host put servers into the launch — answering, not owning: p.srv_own = 0)
    io.srv_on = srv_on; io.srv_need = false;
    // SELF with more than one substep per launch (round 5: the resident stepper's self-collision flavour): a particle with candidates
    // publishes {x0, post-force v} of every substep in tagged write-through records (p.vx, laid out like the exchange array, tag = k + 1).
    // The (particle, candidate) pairs of the block are TASKS spread over all eight wavefronts — the five that do not finish particles
    // first: they ask for the partners' records right behind barrier C, while the finishing wavefronts still sum the forces —; a task's
    // impulse goes to LDS and wavefront 0 sums its lane's impulses in list order, the reference's own summation order (:150-191), applies
    // the average and finishes the particle.  (Wavefront 0 alone, candidate after candidate: 4.2 us of finishing per substep on the rope
    // folded onto itself — up to six dependent hand-offs and the arithmetic of six impulses in one wavefront.)  The candidate lists are
    // constant over the env step: tasks, partners' masses and mask tests are set up once per launch.
    const bool self_res = SELF && n_steps > 1;
    const int ncand_l = (SELF && valid) ? p.coll_num[eb + i] : 0;
    const __amdgpu_buffer_rsrc_t rvx = __builtin_amdgcn_make_buffer_rsrc(p.vx, 0, 0x7fffffff, 0x00020000);
    constexpr int RES_CB = 1;        // the tail (tasks beyond the LDS table: a block with more than RES_TCAP pairs): candidates wavefront 0 asks for together
    constexpr int RES_KS = 2;        // tasks per thread
    constexpr int RES_TCAP = SELF ? RES_THREADS * RES_KS : 1; // tasks with a slot in the LDS table
    __shared__ float4 contrib_s[RES_TCAP];       // per task: impulse (xyz), 1 if the pair collides (w)
    __shared__ float4 selfx_s[SELF ? B : 1], selfv_s[SELF ? B : 1]; // wavefront 0's particles: (x0, mass), post-force velocity
    __shared__ int coff_s[SELF ? B + 1 : 1];     // first task of each lane of wavefront 0; [B]: the block's task count
    const int mask1 = SELF ? p.masks[ic] : 0;
    int t_blk = 0;
    int tk_j[RES_KS], tk_li[RES_KS];
    float tk_m2[RES_KS];             // partner's mass; < 0: same mask (:155: never collides)
#pragma unroll
    for (int m = 0; m < RES_KS; ++m) { tk_j[m] = -1; tk_li[m] = 0; tk_m2[m] = 1.f; }
    if (SELF && self_res) {
        if (wave == 0) {
            int incl = ncand_l;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
            coff_s[lane] = incl - ncand_l;
            if (lane == 63) coff_s[B] = incl;
        }
        __syncthreads();
        t_blk = __builtin_amdgcn_readfirstlane(coff_s[B]);
#pragma unroll
        for (int m = 0; m < RES_KS; ++m) {
            // the five wavefronts that finish nothing take the first tasks
            const int t = (tid >= 192 ? tid - 192 : tid + (RES_THREADS - 192)) + RES_THREADS * m;
            if (t < min(t_blk, RES_TCAP)) {
                int lo = 0, hi = B - 1; // the last lane whose first task is <= t and that has tasks
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (coff_s[mid] <= t) lo = mid; else hi = mid - 1; }
                const int il = b * B + lo, j = p.coll_idx[(eb + il) * (size_t)p.coll_cap + (t - coff_s[lo])];
                tk_j[m] = j; tk_li[m] = lo;
                tk_m2[m] = p.masks[il] != p.masks[j] ? p.masses[j] : -1.f;
            }
        }
    }
    const bool self_blk = SELF && self_res && t_blk > 0; // (the same in every wavefront of the block)
    const int my_coff = (SELF && self_res && wave == 0) ? coff_s[lane] : 0; // wavefront 0: this lane's first task
    const unsigned vxe = (unsigned)e * 6u * xn; // laid out like the exchange array: [env][substep parity][plane][particle, padded to 8] x 16 B
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
            box_s[7] = (mgmax + NEAR_PAD) * (mgmax + NEAR_PAD); // "near": within margin + 3 cm of where any mesh is during this launch (below)
        }
        // visible to the finishing wavefronts after barrier A of the first substep
    }
#ifdef R2S_PHASE_PROBE // wall clock (100 MHz) spent per phase by wavefront 0, summed over the launch: own gather + poll | halo gather + reduce | finish | publish; [4] poll passes
    long long pr_acc[5] = {0, 0, 0, 0, 0}, pr_t = (long long)wall_clock64();
    const long long pr_w0 = pr_t, pr_c0 = (long long)__builtin_readcyclecounter(); // shader clock = cycles / wall ticks x 100 MHz
#define R2S_RSTAMP_(kk) do { const long long now_ = (long long)wall_clock64(); pr_acc[kk] += now_ - pr_t; pr_t = now_; } while (0)
#ifdef R2S_SELF_STAMP2 // the tail of the self-collision flavour: everything up to barrier E | impulse sum | second finish_wave | publish
#define R2S_RSTAMP(kk) do { } while (0)
#define R2S_RSTAMP2(kk) R2S_RSTAMP_(kk)
#else
#define R2S_RSTAMP(kk) R2S_RSTAMP_(kk)
#define R2S_RSTAMP2(kk) do { } while (0)
#endif
#else
#define R2S_RSTAMP(kk) do { } while (0)
#define R2S_RSTAMP2(kk) do { } while (0)
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
        // Once per launch (round 6): is any particle within margin + NEAR_PAD of the union of the mesh boxes over the launch's substeps?  The
        // per-substep tests of a resident launch only look RES_RANGE_PAD (2 mm) beyond a margin — by design: a hovering gripper must not cost
        // them anything —, so until round 5 the host's "near" counter of a small batch rose at most one env step before the first query: too
        // late for a flavour that is picked from the counters of the step TWO before it (the answering servers of the self-collision
        // flavour, physics_flavour.h).  One box test per particle and LAUNCH costs nothing and gives the host its 3 cm of warning.
        if (MESH != 0 && n_steps > 1 && k == 0 && wave == 0) {
            const unsigned long long nm = __builtin_amdgcn_ballot_w64(valid && box_dist2(x0, box_s) < box_s[7]);
            if (nm && lane == __builtin_ctzll(nm)) p.mesh_cnt[p.n_sub] = 1;
        }

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
        f3 v = v0;
        StateM out = xv_out;
        if (!last) out.p = nullptr;
        bool sneed = false, early_pub = false; // early_pub: this wavefront has published its finished lanes already (wave-uniform)
        bool fin = false;
        v4u td[RES_KS][3];   // self-collision tasks: the partners' records
        unsigned pendc = 0;
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
            v = vel_update_rcp(p, v0, f, m1, inv_m1);
            io.x = x0; io.v = v0;
            fin = valid && !(srv_own && srv_ever); // (a particle a server pair owns is not finished here — wavefront 0 takes its state from the pair)
            if (SELF && self_res) {
                if (ncand_l > 0) {
                    fin = false; // finished by wavefront 0 below, once the impulses of its candidates are in
                    if (wave == 0) {
                        const unsigned tag = (unsigned)(k + 1), o = (vxe + (unsigned)(k & 1) * 3u * xn + (unsigned)i) * 16u;
                        srv_store(rvx, o, __float_as_uint(x0.x), __float_as_uint(x0.y), tag);
                        srv_store(rvx, o + xn * 16u, __float_as_uint(x0.z), __float_as_uint(v.x), tag);
                        srv_store(rvx, o + 2u * xn * 16u, __float_as_uint(v.y), __float_as_uint(v.z), tag);
                        selfx_s[lane] = make_float4(x0.x, x0.y, x0.z, m1);
                        selfv_s[lane] = make_float4(v.x, v.y, v.z, 0.f);
                    }
                }
            } else if (SELF) { // as in substep_body: particles with candidates publish v_before_collision and are finished by k_self_finish / k_contact_finish
                const int ncand = ncand_l;
                if (ncand > 0) {
                    const size_t po = par_off(p, step);
                    p.vbc[po + eb + i] = make_float4(v.x, v.y, v.z, 0.f);
                    p.xbc[po + eb + i] = make_float4(x0.x, x0.y, x0.z, 0.f);
                    fin = false;
                    if (MESH != 0 && (p.mesh_defer || MESH == 2)) {
                        bool near;
                        if (mesh_need(p, e, step, x0 + v * p.dt, 0.002f, near)) {
                            p.fault[1] = 1;
                            if (MESH == 2 || p.mesh_rec) { // (small scenes with batched finishing list per environment too: round 6)
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
        }
        if (SELF && self_res) sneed = ncand_l > 0; // wavefront 0 finishes these lanes behind barrier E and publishes all three planes of them
        if (self_blk) {
            // ---- the block's (particle, candidate) tasks: partner's record of THIS substep (polled: its block may still be in its gather) ----
            const unsigned tag = (unsigned)(k + 1), pbase = (vxe + (unsigned)(k & 1) * 3u * xn) * 16u;
#pragma unroll
            for (int m = 0; m < RES_KS; ++m)
                if (tk_j[m] >= 0 && tk_m2[m] >= 0.f) pendc |= 1u << m;
            for (unsigned spins = 0; pendc != 0u; ++spins) {
#pragma unroll
                for (int m = 0; m < RES_KS; ++m)
                    if (pendc & (1u << m)) {
                        const unsigned o = pbase + (unsigned)tk_j[m] * 16u;
                        td[m][0] = srv_load(rvx, o); td[m][1] = srv_load(rvx, o + xn * 16u); td[m][2] = srv_load(rvx, o + 2u * xn * 16u);
                    }
                asm volatile("" ::: "memory");
#pragma unroll
                for (int m = 0; m < RES_KS; ++m)
                    if ((pendc & (1u << m)) && td[m][0].y == tag && td[m][0].w == tag && td[m][1].y == tag && td[m][1].w == tag && td[m][2].y == tag && td[m][2].w == tag)
                        pendc &= ~(1u << m);
                if (pendc != 0u && spins >= p.spin_limit) {
                    resident_fault(p, 7, item, k, (unsigned)tid, (unsigned)tk_j[0], (unsigned)tk_j[1], td[0][0].y, pendc, (unsigned)t_blk);
                    fail_s = 1;
                    break;
                }
            }
            __syncthreads(); // D: wavefront 0's {x0, post-force v} of this substep are in LDS
        }
        // everything after the velocity update for the lanes WITHOUT candidates, behind barrier D: a block that is late — the one the others
        // wait for — finds its partners' records in already, and its impulses must not queue behind work that can run beside them.
        // In a block with tasks wavefronts 1 and 2 do this (1 stores and publishes planes 0 and 1, 2 plane 2); wavefront 0 only finishes
        // the lanes with candidates, behind barrier E
        if (finisher && !(self_blk && wave == 0 && !srv_on)) {
            R2S_QP_DECL(-1);
            io.srv_need = false;
            const bool done1 = finish_wave<MESH, MESH != 0, 0, true>(p, e, i, eb, step, last ? write_forces_last : 0, x0, v, fin, out, nullptr, nullptr, nullptr, nullptr,
                                                                     wave == ((self_blk && !srv_on) ? 1 : 0), &io R2S_QP_ARG);
            if (self_blk && !last) {
                // the lanes that are finished publish BEFORE the impulses are in: their records are what the neighbour blocks need for the
                // next substep
                early_pub = true;
                const unsigned tag = (unsigned)(k + 1), pub = (xe + (unsigned)i) * 16u + (unsigned)((k + 1) & 1) * xb;
                if (srv_on) { // servers in the launch: the three finishing wavefronts as in a launch without candidates, each its plane
                    if (done1) {
                        const float va = wave == 0 ? io.x.x : (wave == 1 ? io.x.z : io.v.x), vb = wave == 0 ? io.x.y : (wave == 1 ? io.v.z : io.v.y);
                        if (valid) {
                            const v4u w = {__float_as_uint(va), tag, __float_as_uint(vb), tag};
                            __builtin_amdgcn_raw_buffer_store_b128(w, rx, pub + (unsigned)wave * xn * 16u, 0, RES_AUX_SC1);
                        }
                        win_s[wave * (RCAP + 1) + lane] = (v2f){va, vb};
                    }
                } else if (!sneed) {
                    if (wave == 1) {
                        if (valid) {
                            const v4u w0 = {__float_as_uint(io.x.x), tag, __float_as_uint(io.x.y), tag}, w1 = {__float_as_uint(io.x.z), tag, __float_as_uint(io.v.z), tag};
                            __builtin_amdgcn_raw_buffer_store_b128(w0, rx, pub, 0, RES_AUX_SC1);
                            __builtin_amdgcn_raw_buffer_store_b128(w1, rx, pub + xn * 16u, 0, RES_AUX_SC1);
                        }
                        win_s[lane] = (v2f){io.x.x, io.x.y};
                        win_s[(RCAP + 1) + lane] = (v2f){io.x.z, io.v.z};
                    } else {
                        if (valid) {
                            const v4u w2 = {__float_as_uint(io.v.x), tag, __float_as_uint(io.v.y), tag};
                            __builtin_amdgcn_raw_buffer_store_b128(w2, rx, pub + 2u * xn * 16u, 0, RES_AUX_SC1);
                        }
                        win_s[2 * (RCAP + 1) + lane] = (v2f){io.v.x, io.v.y};
                    }
                }
            }
        }
        if (self_blk && wave == 0 && !srv_on) early_pub = !last; // (its lanes without candidates were published by wavefronts 1 and 2)
        if (self_blk) {
#pragma unroll
            for (int m = 0; m < RES_KS; ++m) {
                if (tk_j[m] < 0) continue;
                const int t = (tid >= 192 ? tid - 192 : tid + (RES_THREADS - 192)) + RES_THREADS * m;
                float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tk_m2[m] >= 0.f && !(pendc & (1u << m))) {
                    const float4 sx = selfx_s[tk_li[m]], sv = selfv_s[tk_li[m]];
                    const f3 xa = mk(sx.x, sx.y, sx.z), va = mk(sv.x, sv.y, sv.z);
                    const float ma = sx.w, m2 = tk_m2[m];
                    const f3 x2 = mk(__uint_as_float(td[m][0].x), __uint_as_float(td[m][0].z), __uint_as_float(td[m][1].x));
                    const f3 v2 = mk(__uint_as_float(td[m][1].z), __uint_as_float(td[m][2].x), __uint_as_float(td[m][2].z));
                    const f3 dis = x2 - xa;
                    const float dis_len = len(dis);
                    const f3 rv = v2 - va;
                    if (dis_len < p.cd && dot(dis, rv) < -1e-4f) {
                        const f3 nrm = dis / fmaxf(dis_len, 1e-6f);
                        const f3 v_rel_n = nrm * dot(rv, nrm);
                        const float inv = 1.f / ma + 1.f / m2;
                        const f3 impulse_n = (v_rel_n * (-(1.f + p.cse))) / inv;
                        const float vnl = len(v_rel_n);
                        const f3 v_rel_t = rv - v_rel_n;
                        const float vtl = fmaxf(len(v_rel_t), 1e-6f);
                        const float a = fmaxf(0.f, 1.f - p.csf * (1.f + p.cse) * vnl / vtl);
                        const f3 impulse_t = (v_rel_t * (a - 1.f)) / inv;
                        const f3 J = impulse_n + impulse_t;
                        c4 = make_float4(J.x, J.y, J.z, 1.f);
                    }
                }
                contrib_s[t] = c4;
            }
            __syncthreads(); // E: the impulses are in LDS
#ifdef R2S_SELF_STAMP
            R2S_RSTAMP(2); // (probe: phase 2 = force sum .. E, phase 3 = the rest of the substep)
#endif
            R2S_RSTAMP2(0);
        }
        if (finisher) {
            R2S_QP_DECL(-1);
            // particles that need a mesh query were handed to a server pair (resident_server), not finished above.  Wavefront 0 alone waits
            // for their results — the other two finishing wavefronts leave those lanes to it (three wavefronts polling the same granules
            // tripled the poll traffic on the hand-offs of a block with twenty particles in a finger's reach) — and publishes all three planes
            if (SELF && self_res) {
                if (wave == 0 && __builtin_amdgcn_ballot_w64(sneed) != 0ull) {
                    // object_collision (:132-193, :230-268) for this lane's particle: the impulses of its candidates in list order
                    float validc = 0.f;
                    f3 Jsum = mk(0.f, 0.f, 0.f);
                    // (the contributions are read four at a time — independent LDS reads — and added one after the other: a read per
                    // candidate, each behind the previous add, was 0.6 us of the 1.4 us between barrier E and the block's publish)
                    const int c_lds = sneed ? min(ncand_l, max(RES_TCAP - my_coff, 0)) : 0;
                    for (int c = 0; c < c_lds; c += 4) {
                        float4 c4[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) c4[u] = contrib_s[min(my_coff + c + u, RES_TCAP - 1)];
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (c + u < c_lds && c4[u].w != 0.f) { validc += 1.f; Jsum = Jsum + mk(c4[u].x, c4[u].y, c4[u].z); }
                    }
                    R2S_RSTAMP2(1);
                    // the tail: candidates without a slot in the task table (a block with more than RES_TCAP pairs), by this lane itself,
                    // RES_CB records asked for together
                    const unsigned tag = (unsigned)(k + 1), pbase = (vxe + (unsigned)(k & 1) * 3u * xn) * 16u;
                    bool stuck = false;
                    for (int c0 = c_lds; c0 < ncand_l && !stuck; c0 += RES_CB) {
                        int jj[RES_CB];
                        v4u d[RES_CB][3];
                        unsigned pendc = 0;
#pragma unroll
                        for (int c = 0; c < RES_CB; ++c) {
                            jj[c] = c0 + c < ncand_l ? p.coll_idx[(eb + i) * (size_t)p.coll_cap + c0 + c] : 0;
                            if (c0 + c < ncand_l) pendc |= 1u << c;
                        }
                        for (unsigned spins = 0;; ++spins) {
#pragma unroll
                            for (int c = 0; c < RES_CB; ++c)
                                if (pendc & (1u << c)) {
                                    const unsigned o = pbase + (unsigned)jj[c] * 16u;
                                    d[c][0] = srv_load(rvx, o); d[c][1] = srv_load(rvx, o + xn * 16u); d[c][2] = srv_load(rvx, o + 2u * xn * 16u);
                                }
                            asm volatile("" ::: "memory");
#pragma unroll
                            for (int c = 0; c < RES_CB; ++c)
                                if ((pendc & (1u << c)) && d[c][0].y == tag && d[c][0].w == tag && d[c][1].y == tag && d[c][1].w == tag && d[c][2].y == tag && d[c][2].w == tag)
                                    pendc &= ~(1u << c);
                            if (pendc == 0u) break;
                            if (spins >= p.spin_limit) { stuck = true; break; }
                        }
                        if (stuck) { resident_fault(p, 7, item, k, (unsigned)i, (unsigned)jj[0], d[0][0].y, (unsigned)c0, pendc, (unsigned)ncand_l); fail_s = 1; break; }
#pragma unroll
                        for (int c = 0; c < RES_CB; ++c) {
                            if (c0 + c >= ncand_l) continue;
                            const int j = jj[c];
                            const f3 x2 = mk(__uint_as_float(d[c][0].x), __uint_as_float(d[c][0].z), __uint_as_float(d[c][1].x));
                            const f3 v2 = mk(__uint_as_float(d[c][1].z), __uint_as_float(d[c][2].x), __uint_as_float(d[c][2].z));
                            const float m2 = p.masses[j];
                            const f3 dis = x2 - x0;
                            const float dis_len = len(dis);
                            const f3 rv = v2 - v;
                            if (mask1 != p.masks[j] && dis_len < p.cd && dot(dis, rv) < -1e-4f) {
                                validc += 1.f;
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
                    const f3 vi = (sneed && validc > 0.f) ? v - (Jsum / validc) / m1 : v;
                    finish_wave<MESH, MESH != 0, 0, true>(p, e, i, eb, step, last ? write_forces_last : 0, x0, vi, sneed, out, nullptr, nullptr, nullptr, nullptr, true, &io R2S_QP_ARG);
                    if (srv_on && sneed) v = vi; // (a lane that needs a mesh query on top: the request below carries the velocity AFTER the impulses)
                }
            }
            if (SRV && srv_on) {
                const bool need_now = io.srv_need; // the same in the three finishing wavefronts (same inputs, same instructions)
                srv_ever = srv_ever || need_now;
                const bool sself = sneed;                          // (SELF: lanes with candidates — wavefront 0 has finished them above unless they need a query)
                const bool ssrv = srv_own ? srv_ever : need_now;   // lanes whose state comes from a server pair
                const unsigned uk = (unsigned)k;
                if (wave == 0) {
                    bool inplace = srv_own && srv_ever && !srv_mine && !need_now; // owning servers, no pair was left at its first need: in place from then on
                    bool claimed_now = false;
                    if (need_now && !srv_mine) {
                        const int slot = atomicAdd(p.srv_ctl, 1);
                        if (slot >= p.srv_low) p.fault[2] = 1; // units running LOW (30 % claimed; round 6) or out: the host takes the following steps off the resident
                                                                         // launch BEFORE a claim goes unanswered (the flavour follows the counters of an earlier step, and a closing
                                                                         // grasp doubles its contacts in that time: 12 -> 14 -> 37 -> 61 -> 79 listed particles per env step of the toy)
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
                        finish_wave<MESH, false, 1, true>(p, e, i, eb, step, last ? write_forces_last : 0, x0, v, ssrv && inplace, out, nullptr, nullptr, nullptr, nullptr,
                                                          true, &io R2S_QP_ARG);
                    if (__builtin_amdgcn_ballot_w64(ssrv && !inplace) != 0ull) {
                        // the lanes that are finished publish BEFORE the wait: their records are what the pairs (and the neighbour blocks) need
                        // for the next substep — behind the wait, every substep of a particle in contact paid a second hand-off for them
                        if (!last) {
                            early_pub = true;
                            if (!ssrv && !sself) {
                                if (valid) {
                                    const v4u w = {__float_as_uint(io.x.x), (unsigned)(k + 1), __float_as_uint(io.x.y), (unsigned)(k + 1)};
                                    __builtin_amdgcn_raw_buffer_store_b128(w, rx, (xe + (unsigned)i) * 16u + (unsigned)((k + 1) & 1) * xb, 0, RES_AUX_SC1);
                                }
                                win_s[lane] = (v2f){io.x.x, io.x.y};
                            }
                        }
                        for (unsigned spins = 0;; ++spins) {
                            bool ok = true;
                            if (ssrv && !inplace) {
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
                        if (last && ssrv && !inplace && xv_out.p != nullptr) st_store(xv_out, eb + i, io.x, io.v);
                    }
                }
                sneed = ssrv || sself;
            }
#ifdef R2S_PHASE_PROBE
            if (io.x.x == 1.2345e33f) return;
#endif
#ifndef R2S_SELF_STAMP
            R2S_RSTAMP(2);
#endif
            R2S_RSTAMP2(2);

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
            R2S_RSTAMP2(3);
        }
    }
    if (SRV && srv_on) { // end this block's server pairs, then count the block out (pairs nobody claimed leave when every block has)
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

