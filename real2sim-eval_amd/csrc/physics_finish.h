// physics_finish.h — part of physics.hip's ONE translation unit (included there, inside its anonymous namespace, in this order: physics_mesh_query.h,
// physics_substep.h, physics_resident.h, physics_finish.h, physics_aux.h); not a stand-alone header.  Round 5 split the 4 800-line file by section;
// the token stream the compiler sees is unchanged.
// Here: the finishing code: self-collision impulses (self_impulse), k_self_finish, contact_finish_body + k_contact_finish, and the fused substep with the finishers of the previous substep at its head (k_substep_pf).

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

// Part 2 of the finishing code (WITH_SELF): the particles of the candidate lists that part 1 did not take (no mesh in reach), 16 lanes each,
// finished in place.  `FW`: the finish_wave template without query code that matches the scene (1 small meshes, 2 a large one).
// Shared by contact_finish_body and the batched small-scene finisher (contact_finish_batch).
template <int FW, bool PFOUT>
__device__ __forceinline__ void finish_candidates(const PhysDev& p, const StateM xv_out, int step, int write_forces, int L, int n_wg, int nthr, size_t po, int lane, int wave, long long probe_entry)
{
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
            finish_wave<FW, false, 2, false, false, PFOUT>(p, e, i, eb, step, write_forces, x0, v, act && sub == 0, xv_out, nullptr, nullptr, nullptr, nullptr, true, nullptr R2S_QP_ARG);
        }
#ifdef R2S_PHASE_PROBE
        for (int o = 32; o > 0; o >>= 1) probe_cnt = max(probe_cnt, __shfl_xor(probe_cnt, o));
        if (lane == 0 && gw < 1024 && step == p.n_sub - 2) { g_query_probe[gw * 32 + 29] = (long long)wall_clock64(); g_query_probe[gw * 32 + 27] = probe_cnt; }
#endif
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
        finish_candidates<(MESHQ == 3 ? 1 : 2), PFOUT>(p, xv_out, step, write_forces, L, n_wg, nthr, po, lane, wave, probe_entry);
#else
        finish_candidates<(MESHQ == 3 ? 1 : 2), PFOUT>(p, xv_out, step, write_forces, L, n_wg, nthr, po, lane, wave, 0);
#endif
    }
}

// Part 2 for the batched finisher, per ENVIRONMENT (round 6): the workgroups of an environment behind its part-1 workgroups (`slot_rel` of
// `nslot_rel`) walk THAT environment's candidate list, 16 lanes per particle.  With the interleaved mapping above part 1 (from the first slot)
// and part 2 (from the last) need twice the workgroups to stay apart — and every finishing workgroup of a launch holds one of the chip's
// 1 024 slots for the 2 - 3 us it takes to find out it has nothing to do: the contact flavours are bound by exactly those slots.
template <int FW, bool PFOUT>
__device__ __forceinline__ void finish_candidates_env(const PhysDev& p, const StateM xv_out, int step, int write_forces, int e, int slot_rel, int nslot_rel, size_t po)
{
    constexpr int G = 16;
    const int sub = (int)(threadIdx.x & (G - 1)), grp = (int)(threadIdx.x / G), gpb = 256 / G;
    const size_t eb = (size_t)e * p.N;
    const int t0g = slot_rel * gpb + grp, gstride = nslot_rel * gpb;
    int2 ci = p.cand_list[eb + (size_t)min(max(t0g, 0), p.N - 1)];                 // speculative, with the count (one round trip)
    const int n = slot_rel >= 0 ? p.cand_cnt_env[e] : 0;
    for (int t = t0g; __builtin_amdgcn_ballot_w64(t < n) != 0ull; t += gstride) { // wave-uniform trip count (the group shuffles run with their lanes together)
        if (t != t0g || t >= n) ci = p.cand_list[eb + (size_t)(t < n ? t : 0)];
        const int i = ci.y, cnt = ci.x >> 12;
        const bool act = t < n && p.cand_mark[po + eb + i] != step + 1; // not already done in part 1
        const f3 x0 = xyz(p.xbc[po + eb + i]);
        const f3 vpre = xyz(p.vbc[po + eb + i]);
        const f3 v = self_impulse<G>(p, po, eb, i, act, x0, vpre, sub, cnt);
        if (act && sub == 0) { // (the bound the skipped mesh test relies on: see finish_candidates)
            const f3 dvi = v - vpre;
            if (dot(dvi, dvi) * p.dt * p.dt > 0.002f * 0.002f) *p.fault = 1;
        }
        R2S_QP_DECL(-1);
        finish_wave<FW, false, 2, false, false, PFOUT>(p, e, i, eb, step, write_forces, x0, v, act && sub == 0, xv_out, nullptr, nullptr, nullptr, nullptr, true, nullptr R2S_QP_ARG);
    }
}

// ---- small scenes, BATCHED finishing (round 6; MESHQ 4) ---------------------------------------------------------------------------------
// Until round 5 a listed particle of a small scene (every mesh small, <= 128 faces: two 44-face fingers + a box) was finished by a WORKGROUP
// of its own — two wavefronts, one triangle per lane (MESHQ 3).  That form is latency-optimal for a handful of particles (the resident
// stepper's servers still use it) and wasteful for many: every query evaluates the closest point AND the solid angle of all ~100 faces
// (~500 instructions of a lone wavefront) for ONE particle.  Once the grasp really closes (round 6: the pads squeeze the toy's limbs,
// 85 - 100 listed particles per environment and substep instead of ~15) the 32 finishing workgroups per environment walked three particles
// each back to back, ~9 us of the whole chip's VALU issue per batched substep, and the blocks that wait for them paced the launch: 42 us
// per batched substep of the headline against 18 free.
// Here a WAVEFRONT takes FB_PW = 4 records of one environment's list at once (the per-environment records of the large-mesh scenes,
// mesh_rec_push), lane = (particle, slice), and prunes before it computes:
//   1. every lane: the distance^2 from its particle's point to the BOX of each triangle of its slice (triangles s, s + 16, ... from an LDS
//      copy of the substep's triangles and their boxes, staged once per workgroup) — a lower bound of the distance to the triangle;
//   2. every lane: the exact closest point (closest_bary) of the triangle with its smallest bound; the minimum over the particle's sixteen
//      lanes is an upper bound `ub` of the answer;
//   3. the (particle, triangle) pairs whose lower bound does not exceed ub (widened by 1e-4 relative: the bounds round) are the only ones
//      that can still win; the wavefront packs them into a queue and evaluates them 64 at a time (typically one trip), minima per
//      particle through LDS atomics on the 64-bit (distance^2 bits, face id) key.
// Same answer: the lexicographic minimum of (distance^2, face id) over the faces closer than max_dist, each distance from the same
// closest-point arithmetic (FMA contraction off); a pruned triangle is strictly farther than the winner.
// The sign: the reference's rule is "exact winding number over all faces > 0.6" (:322-324).  A closed, consistently oriented manifold
// (checked at construction; outward orientation by its signed volume) contributes exactly 1 inside and 0 outside, and nothing for a point
// outside its box.  So: no mesh box holds the point and no mesh is open -> +1 without a solid angle; the point is only inside the box of
// the mesh of its closest point, that mesh is closed + outward and the closest point lies in a face's interior -> the side of that face's
// plane decides; anything else -> the winding number over the faces of the meshes that can contribute, all pairs, as before.
// Same decisions, same response arithmetic in the same order as finish_wave: states bit-identical to the MESHQ 3 flavour
// (tests/test_fin_batch_gpu.py).  No workgroup barrier inside a query: the wavefronts of a workgroup only share the staged triangles.
#ifndef R2S_FB_SL
#define R2S_FB_SL 16
#endif
constexpr int FB_MAX_F = 128, FB_MAX_MESH = 8, FB_SL = R2S_FB_SL, FB_PW = 64 / FB_SL, FB_NJ = FB_MAX_F / FB_SL; // slices per particle, particles per wavefront, triangles per lane
struct __attribute__((aligned(16))) BatchWave {
    float q[FB_PW][4];                    // the particles' query points
    unsigned long long key[FB_PW];        // best (distance^2 bits, face) so far
    float cp[FB_PW][4];                   // its closest point; [3] bits: region of the closest feature | side of the face's plane << 8 (1: behind it)
    int meta[FB_PW][4];                   // mesh_map, face_map, mesh of that face
    float st[FB_PW][20];                  // a particle's state between the phases of its substep (the sixteen lanes of a particle hold the same values: parked here,
                                          // not in sixteen copies of registers — the finishing role's registers are what the merged launch spills)
    float lb[FB_NJ][64];                  // this lane's lower bounds (distance^2 to the boxes of triangles sl, sl + 16, ...) of the first query
    unsigned short queue[FB_PW * FB_MAX_F]; // surviving (particle << 8 | face) pairs
};
struct __attribute__((aligned(16))) BatchShare {
    float4 tri[FB_MAX_F][3];              // {a.x a.y a.z b.x} {b.y b.z c.x c.y} {c.z, bits(mesh_map), bits(face_map), bits(mesh)} of the substep's triangles
    float tbox[FB_MAX_F][6];              // their boxes lo.xyz hi.xyz
    float4 box[FB_MAX_MESH][2];           // the substep's world box of mesh m {lo.x lo.y lo.z hi.x} {hi.y hi.z, flags, -}; flags bits: 1 = not a closed manifold, 2 = closed but oriented inward
    float eef[12];                        // interp_center, dyn_omega, dyn_vel[0], dyn_vel[1] of the environment
    BatchWave w[4];
};
static_assert(sizeof(BatchShare) <= sizeof(v2f) * 3 * (1024 + 1), "the batched finishers share the fused role's LDS window in k_substep_pf");

template <int O> __device__ __forceinline__ unsigned fb_xor_u32(unsigned v) // lane ^ O, O < 16
{
    if (O == 1) return dpp_u32<0xB1>(v);   // quad_perm [1,0,3,2]
    if (O == 2) return dpp_u32<0x4E>(v);   // quad_perm [2,3,0,1]
    return (unsigned)__shfl_xor((int)v, O);
}
template <int O> __device__ __forceinline__ float fb_xor_f32(float v) { return __uint_as_float(fb_xor_u32<O>(__float_as_uint(v))); }
template <int O> __device__ __forceinline__ unsigned long long fb_xor_u64(unsigned long long v)
{
    return ((unsigned long long)fb_xor_u32<O>((unsigned)(v >> 32)) << 32) | fb_xor_u32<O>((unsigned)v);
}
// LDS traffic between the lanes of ONE wavefront: its DS operations execute in program order, the compiler only has to keep that order
__device__ __forceinline__ void fb_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// object_collision for the tagged records of a batch: the 16 lanes of a particle ARE the 16-lane group of part 2 / k_self_finish
// (self_impulse<16>: lane l takes candidates l, l + 16, ...; the xor tree 8, 4, 2, 1) — the same sums, the same bits.  The loads are split
// from the arithmetic so that the two dependent round trips (candidate index, then the partner's published state) overlap the staging of
// the workgroup's triangles: in-kernel stamps of the first batched form had 3.5 us between "staged" and the first query's bounds, nearly
// all of it these gathers.
static_assert(FB_SL == 16, "the particle's lanes are the 16-lane group of self_impulse<16>");
struct ImpPre {
    bool on;            // this lane has a first candidate (tagged record, slice < candidate count)
    int j, mask1, mask2;
    float m1, m2;
    float4 x2, v2;
};
__device__ __forceinline__ void imp_pre1(const PhysDev& p, size_t eb, int i, bool tagged, int sl, int cnt, ImpPre& q)
{
    q.on = tagged && sl < cnt;
    q.j = 0; q.mask1 = 0; q.m1 = 1.f;
    if (tagged) { q.m1 = p.masses[i]; q.mask1 = p.masks[i]; }
    if (q.on) q.j = p.coll_idx[(eb + i) * (size_t)p.coll_cap + sl];
}
__device__ __forceinline__ void imp_pre2(const PhysDev& p, size_t po, size_t eb, ImpPre& q)
{
    q.x2 = make_float4(0.f, 0.f, 0.f, 0.f); q.v2 = q.x2; q.m2 = 1.f; q.mask2 = 0;
    if (q.on) { q.x2 = p.xbc[po + eb + q.j]; q.v2 = p.vbc[po + eb + q.j]; q.m2 = p.masses[q.j]; q.mask2 = p.masks[q.j]; }
}
__device__ __forceinline__ void imp_term(const PhysDev& p, f3 x0, f3 v, float m1, int mask1, f3 x2, f3 v2, float m2, int mask2, float& valid, f3& Jsum)
{
    const f3 dis = x2 - x0;
    const float dis_len = len(dis);
    const f3 rv = v2 - v;
    if (mask1 != mask2 && dis_len < p.cd && dot(dis, rv) < -1e-4f) {
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
__device__ __forceinline__ f3 self_impulse_pre(const PhysDev& p, size_t po, size_t eb, int i, bool tagged, f3 x0, f3 v, int sl, int cnt, const ImpPre& q)
{
    float valid = 0.f;
    f3 Jsum = mk(0.f, 0.f, 0.f);
    bool have = q.on;
    f3 x2 = xyz(q.x2), v2 = xyz(q.v2);
    float m2 = q.m2;
    int mask2 = q.mask2;
    for (int k = sl; __builtin_amdgcn_ballot_w64(have) != 0ull;) { // one copy of the impulse arithmetic: the first trip on the prefetched partner, further trips (more than 16 candidates: rare) load theirs
        if (have) imp_term(p, x0, v, q.m1, q.mask1, x2, v2, m2, mask2, valid, Jsum);
        k += 16;
        have = tagged && k < cnt;
        if (have) {
            const int j = p.coll_idx[(eb + i) * (size_t)p.coll_cap + k];
            x2 = xyz(p.xbc[po + eb + j]); v2 = xyz(p.vbc[po + eb + j]); m2 = p.masses[j]; mask2 = p.masks[j];
        }
    }
    valid += fb_xor_f32<8>(valid); Jsum.x += fb_xor_f32<8>(Jsum.x); Jsum.y += fb_xor_f32<8>(Jsum.y); Jsum.z += fb_xor_f32<8>(Jsum.z);
    valid += fb_xor_f32<4>(valid); Jsum.x += fb_xor_f32<4>(Jsum.x); Jsum.y += fb_xor_f32<4>(Jsum.y); Jsum.z += fb_xor_f32<4>(Jsum.z);
    valid += fb_xor_f32<2>(valid); Jsum.x += fb_xor_f32<2>(Jsum.x); Jsum.y += fb_xor_f32<2>(Jsum.y); Jsum.z += fb_xor_f32<2>(Jsum.z);
    valid += fb_xor_f32<1>(valid); Jsum.x += fb_xor_f32<1>(Jsum.x); Jsum.y += fb_xor_f32<1>(Jsum.y); Jsum.z += fb_xor_f32<1>(Jsum.z);
    return (tagged && valid > 0.f) ? v - (Jsum / valid) / q.m1 : v;
}

// meshes whose faces can contribute to the winding number of point q: the open ones, and the closed ones whose box holds q
__device__ __forceinline__ unsigned fb_wn_mask(const PhysDev& p, const BatchShare& sh, f3 q)
{
    unsigned m = 0u;
#pragma unroll 1
    for (int k = 0; k < p.n_mesh; ++k) { // (a loop on purpose: unrolled, the compiler reads every box ahead — 16 more registers per mesh)
        const float4 b0 = sh.box[k][0], b1 = sh.box[k][1];
        const bool in = q.x >= b0.x && q.x <= b0.w && q.y >= b0.y && q.y <= b1.x && q.z >= b0.z && q.z <= b1.y;
        if (in || (__float_as_int(b1.z) & 1) != 0) m |= 1u << k;
    }
    return m;
}
__device__ __forceinline__ void fb_tri(const BatchShare& sh, int f, f3& a, f3& b, f3& c)
{
    const float4 t0 = sh.tri[f][0], t1 = sh.tri[f][1];
    const float cz = sh.tri[f][2].x;
    a = mk(t0.x, t0.y, t0.z); b = mk(t0.w, t1.x, t1.y); c = mk(t1.z, t1.w, cz);
}
// (distance^2 bits << 32 | face) of triangle f for point q, ~0 beyond max_dist; closest point, feature region | side of the face's plane << 8
// (1: q lies behind the face), and the face's caller-side maps {mesh_map, face_map, mesh}
// (-DR2S_FB_CALL_EVAL: ONE copy of the closest-point code for the three evaluation sites, as a call — measured in round 6: the kernels need
// MORE registers that way, 164 vs 153 VGPRs; not the build)
struct FbEval { float cx, cy, cz, d2; int region; };
#ifdef R2S_FB_CALL_EVAL
__device__ __noinline__
#else
__device__ __forceinline__
#endif
FbEval fb_closest(float ax, float ay, float az, float bx, float by, float bz, float cx, float cy, float cz, float qx, float qy, float qz)
{
    const f3 a = mk(ax, ay, az), b = mk(bx, by, bz), c = mk(cx, cy, cz), q = mk(qx, qy, qz);
    float u, v;
    int region;
    closest_bary(a, b, c, q, u, v, region);
    const f3 cp = a * u + b * v + c * (1.f - u - v);
    const f3 d = cp - q;
    FbEval r;
    r.cx = cp.x; r.cy = cp.y; r.cz = cp.z;
    r.d2 = dot(d, d);
    r.region = region | (dot(q - cp, cross(b - a, c - a)) < 0.f ? 256 : 0);
    return r;
}
__device__ __forceinline__ unsigned long long fb_eval(const BatchShare& sh, int f, f3 q, f3& cp, int& region, int4& meta)
{
    const float4 t0 = sh.tri[f][0], t1 = sh.tri[f][1], t2 = sh.tri[f][2];
    const FbEval r = fb_closest(t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, q.x, q.y, q.z);
    cp = mk(r.cx, r.cy, r.cz);
    region = r.region;
    meta = make_int4(__float_as_int(t2.y), __float_as_int(t2.z), __float_as_int(t2.w), 0);
    return r.d2 < MESH_MAX_DIST * MESH_MAX_DIST ? (((unsigned long long)__float_as_uint(r.d2) << 32) | (unsigned)f) : ~0ull;
}
// minimum over the 16 lanes of a particle (a row of the wavefront), every lane gets it: rotations inside the row (DPP modifiers)
template <int CTRL> __device__ __forceinline__ unsigned long long fb_dpp_u64(unsigned long long v)
{
    return ((unsigned long long)dpp_u32<CTRL>((unsigned)(v >> 32)) << 32) | dpp_u32<CTRL>((unsigned)v);
}
__device__ __forceinline__ unsigned long long fb_row_min_u64(unsigned long long v)
{
    { const unsigned long long o = fb_dpp_u64<0x128>(v); v = o < v ? o : v; } // row_ror:8
    { const unsigned long long o = fb_dpp_u64<0x124>(v); v = o < v ? o : v; } // row_ror:4
    { const unsigned long long o = fb_dpp_u64<0x122>(v); v = o < v ? o : v; } // row_ror:2
    { const unsigned long long o = fb_dpp_u64<0x121>(v); v = o < v ? o : v; } // row_ror:1
    return v;
}

#ifdef R2S_PHASE_PROBE
struct BProbe { int row, n; bool on; };
#define R2S_BP_PARAM , BProbe& bp
#define R2S_BP_ARG , bp
#define R2S_BP_STAMP() do { if (bp.on && (threadIdx.x & 63) == 0 && bp.n < 27) g_query_probe[bp.row * 32 + bp.n] = (long long)wall_clock64(); ++bp.n; } while (0)
#define R2S_BP_VALUE(v) do { if (bp.on && (threadIdx.x & 63) == 0 && bp.n < 27) g_query_probe[bp.row * 32 + bp.n] = (long long)(v); ++bp.n; } while (0)
#else
#define R2S_BP_PARAM
#define R2S_BP_ARG
#define R2S_BP_STAMP() do { } while (0)
#define R2S_BP_VALUE(v) do { } while (0)
#endif
// One query for the FB_PW particles of this wavefront (lane = particle * FB_SL + slice); no workgroup barrier.  The answer comes back in
// every lane of the particle; `d2` = its distance^2.
// REQ (the gripper branch's re-query, :397, a few micrometres from the first query's point): steps 1 and 2 are skipped.  With d1 the first
// answer's distance and delta the distance between the two query points, the re-query's answer is at most d1 + delta away, and a triangle
// that close to the new point was within d1 + 2 delta of the old one: the pairs whose FIRST bound is <= `thr` = (d1 + 2 delta)^2 (widened)
// are the only ones that can win; `lb` still holds those bounds.
template <bool REQ>
__device__ __forceinline__ MeshHit batch_query(const PhysDev& p, BatchShare& sh, BatchWave& w, f3 q, bool want, float thr_req, float& d2_out, float (&lm)[3] R2S_BP_PARAM)
{
    const int lane = (int)(threadIdx.x & 63);
    const int sl = lane & (FB_SL - 1), pi = lane / FB_SL;
    const float MAXD2 = MESH_MAX_DIST * MESH_MAX_DIST;
    MeshHit out = {false, 0.f, 0, mk(0.f, 0.f, 0.f), 0, 0};
    d2_out = 0.f;
    if (__builtin_amdgcn_ballot_w64(want) == 0ull) return out; // wave-uniform
    const int nF = p.nF;
    if (sl == 0) *(float4*)w.q[pi] = make_float4(q.x, q.y, q.z, 0.f);
    float thr = thr_req;
    int minf = -1;
    bool rest = true; // some lane may hold a second pair that can still win: the packed path below
    f3 cp = mk(0.f, 0.f, 0.f);
    int region = 0;
    int4 meta = make_int4(0, 0, 0, 0);
    unsigned long long key = ~0ull;
    if (!REQ) {
        // 1. lower bounds: this lane's triangles sl, sl + 16, ...; the smallest (its triangle: minf) and the second smallest are kept
        float minlb = 3.0e38f, min2 = 3.0e38f;
#pragma unroll 2
        for (int j = 0; j * FB_SL < nF; ++j) { // wave-uniform
            const int f = sl + j * FB_SL;
            const float* tb = sh.tbox[f < nF ? f : nF - 1];
            const float dx = fmaxf(fmaxf(tb[0] - q.x, q.x - tb[3]), 0.f), dy = fmaxf(fmaxf(tb[1] - q.y, q.y - tb[4]), 0.f), dz = fmaxf(fmaxf(tb[2] - q.z, q.z - tb[5]), 0.f);
            const float d2 = (want && f < nF) ? dx * dx + dy * dy + dz * dz : 3.0e38f;
            w.lb[j][lane] = d2;
            if (d2 < minlb) { min2 = minlb; minlb = d2; minf = f; }
            else min2 = fminf(min2, d2);
        }
        lm[0] = minlb; lm[1] = __int_as_float(minf); lm[2] = min2;
        if (!(minlb < MAXD2 * 1.0001f + 1e-12f)) minf = -1; // (a triangle whose box is beyond max_dist cannot answer)
        R2S_BP_STAMP(); // bounds done
        // 2. the exact distance of each lane's most promising triangle; the best of the particle's lanes is an upper bound of the answer
        if (__builtin_amdgcn_ballot_w64(minf >= 0) != 0ull) {
            const unsigned long long k = fb_eval(sh, minf >= 0 ? minf : 0, q, cp, region, meta);
            if (minf >= 0) key = k;
        }
        const unsigned long long mn = fb_row_min_u64(key);
        if (sl == 0) w.key[pi] = mn;
        if (key == mn && mn != ~0ull) { *(float4*)w.cp[pi] = make_float4(cp.x, cp.y, cp.z, __int_as_float(region)); *(int4*)w.meta[pi] = meta; }
        const float ub = mn != ~0ull ? __uint_as_float((unsigned)(mn >> 32)) : MAXD2;
        thr = ub * 1.0001f + 1e-12f;
        rest = __builtin_amdgcn_ballot_w64(want && lm[2] <= thr) != 0ull; // no lane's SECOND bound is in reach: nothing behind step 2 (the common case)
        R2S_BP_STAMP(); // upper bound known
    } else {
        R2S_BP_STAMP();
        rest = __builtin_amdgcn_ballot_w64(want && lm[2] <= thr) != 0ull;
        if (!rest) {
            // the common case of the re-query: no lane holds more than ONE pair in reach, its first query's nearest box — one evaluation per
            // lane and the row minimum, no packing
            minf = (want && lm[0] <= thr) ? __float_as_int(lm[1]) : -1;
            if (__builtin_amdgcn_ballot_w64(minf >= 0) != 0ull) {
                const unsigned long long k = fb_eval(sh, minf >= 0 ? minf : 0, q, cp, region, meta);
                if (minf >= 0) key = k;
            }
            const unsigned long long mn = fb_row_min_u64(key);
            if (sl == 0) w.key[pi] = mn;
            if (key == mn && mn != ~0ull) { *(float4*)w.cp[pi] = make_float4(cp.x, cp.y, cp.z, __int_as_float(region)); *(int4*)w.meta[pi] = meta; }
        } else if (sl == 0) w.key[pi] = ~0ull;
        R2S_BP_STAMP();
    }
    // 3. the pairs that can still win (usually none behind step 2), packed; 64 at a time
    int base = 0;
    if (rest)
    for (int j = 0; j * FB_SL < nF; ++j) { // (a loop on purpose: unrolled, with the eight bounds read ahead, the kernel needs 167 instead of 122 VGPRs)
        const int f = sl + j * FB_SL;
        const bool surv = want && w.lb[j][lane] <= thr && f != minf;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(surv);
        if (m != 0ull) {
            const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (surv) w.queue[base + rank] = (unsigned short)((pi << 8) | f);
            base += __builtin_popcountll(m);
        }
    }
    fb_wave_sync();
    R2S_BP_STAMP(); // survivors packed
    R2S_BP_VALUE(base);
    for (int it = 0; it < base; it += 64) { // wave-uniform
        const int idx = it + lane;
        const bool on = idx < base;
        const int pf = on ? (int)w.queue[idx] : 0;
        const int pj = pf >> 8, fj = pf & 255;
        const float4 qq = *(const float4*)w.q[pj];
        f3 cpj;
        int rj;
        int4 mj;
        const unsigned long long k = fb_eval(sh, fj, mk(qq.x, qq.y, qq.z), cpj, rj, mj);
        if (on && k != ~0ull) __hip_atomic_fetch_min(&w.key[pj], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        fb_wave_sync();
        if (on && k != ~0ull && k == w.key[pj]) { *(float4*)w.cp[pj] = make_float4(cpj.x, cpj.y, cpj.z, __int_as_float(rj)); *(int4*)w.meta[pj] = mj; } // (a later trip's better key overwrites)
        fb_wave_sync();
    }
    fb_wave_sync();
    R2S_BP_STAMP(); // survivors evaluated
    if (!want) return out;
    const unsigned long long mn = w.key[pi];
    const float4 bc = *(const float4*)w.cp[pi];
    const int4 mt = *(const int4*)w.meta[pi];
    const bool found = mn != ~0ull;
    out.result = found;
    out.face = found ? (int)(unsigned)(mn & 0xffffffffull) : 0; // a miss reports face 0, like warp's zero-initialised query
    out.pt = mk(bc.x, bc.y, bc.z);
    d2_out = found ? __uint_as_float((unsigned)(mn >> 32)) : 0.f;
    if (found) { out.mm = mt.x; out.fm = mt.y; }
    else { const float4 tf = sh.tri[0][2]; out.mm = __float_as_int(tf.y); out.fm = __float_as_int(tf.z); }
    // sign (see the header): which meshes can contribute at all?
    const unsigned wmask = found ? fb_wn_mask(p, sh, q) : 0u;
    const int mstar = mt.z, rg = __float_as_int(bc.w);
    const bool simple = found && (rg & 255) == 0 && __float_as_int(sh.box[found ? mstar : 0][1].z) == 0;
    const unsigned gmask = !found ? 0u : simple ? (wmask & ~(1u << mstar)) : wmask;
    float sign = (simple && gmask == 0u && ((wmask >> mstar) & 1u) != 0u && (rg & 256) != 0) ? -1.f : 1.f;
    const unsigned long long gen = __builtin_amdgcn_ballot_w64(gmask != 0u);
    R2S_BP_VALUE(__builtin_popcountll(gen));
    if (gen != 0ull) { // rare: the exact winding number over the faces of the meshes in wmask, this lane's slice
        float sa = 0.f;
        for (int f0 = 0; f0 < nF; f0 += FB_SL) {
            const int f = f0 + sl;
            const int fc = f < nF ? f : nF - 1;
            const bool cnt = gmask != 0u && f < nF && ((wmask >> __float_as_int(sh.tri[fc][2].w)) & 1u) != 0u;
            if (__builtin_amdgcn_ballot_w64(cnt) != 0ull) {
                f3 a, b, c;
                fb_tri(sh, fc, a, b, c);
                const f3 a2 = a - q, b2 = b - q, c2 = c - q;
                const float la = len(a2), lb2 = len(b2), lc = len(c2);
                const float det = dot(a2, cross(b2, c2));
                const float den = la * lb2 * lc + dot(a2, b2) * lc + dot(b2, c2) * la + dot(c2, a2) * lb2;
                if (cnt) sa += 2.f * atan2f(det, den);
            }
        }
        sa += fb_xor_f32<8>(sa); sa += fb_xor_f32<4>(sa); sa += fb_xor_f32<2>(sa); sa += fb_xor_f32<1>(sa);
        if (gmask != 0u) sign = sa / (float)(4.0 * 3.14159265358979323846) > WIND_THRESHOLD ? -1.f : 1.f;
    }
    out.sign = sign;
    return out;
}

// The finishing code of a small scene's substep, batched: part 1 the environment's records FB_PW per wavefront, part 2 the candidate lists as
// before.  Workgroup L of n_wg = (environment L % ne, slot L / ne); 256 threads.
template <bool WITH_SELF, bool PFOUT>
__device__ __forceinline__ void contact_finish_batch(const PhysDev& p, const StateM xv_out, int step, int write_forces, int L, int n_wg, BatchShare& sh)
{
#ifndef R2S_NO_FINISH_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sl = lane & (FB_SL - 1), pi = lane / FB_SL;
    const size_t po = par_off(p, step);
#ifdef R2S_PHASE_PROBE
    const long long probe_entry = (long long)wall_clock64();
    const int probe_row = L * 4 + wave;
    const bool probe_on = step == p.n_sub - 2 && probe_row < 512;
    BProbe bp = {probe_row, 0, probe_on};
#define probe_n bp.n
#define R2S_BSTAMP() do { if (probe_on && lane == 0 && probe_n < 27) g_query_probe[probe_row * 32 + probe_n] = (long long)wall_clock64(); ++probe_n; } while (0)
#else
#define R2S_BSTAMP() do { } while (0)
#endif
    const int nslot = n_wg / p.ne, slot = L / p.ne;   // (a head padded to a multiple of 8 workgroups: the surplus has no slot)
    const int e = p.e0 + L % p.ne;
    const size_t eb = (size_t)e * p.N;
    const int4* rec = p.mesh_rec + 2 * (po + eb);
    const int unit = slot * 4 + wave, n_units = nslot * 4; // a wavefront is the unit of part 1
    const int tb0 = unit * FB_PW;
    int4 ra = rec[2 * min(tb0 + pi, p.N - 1)], rc = rec[2 * min(tb0 + pi, p.N - 1) + 1]; // speculative, with the count (entries past the count are stale, never used)
    const int n_rec = slot < nslot ? min(p.rec_cnt[(size_t)e * p.n_sub + step], p.N) : 0;
    if (slot * 4 * FB_PW < n_rec) { // workgroup-uniform: some wavefront of this workgroup has records
        // stage the substep's triangles, their boxes, the meshes' boxes and the gripper motion (index -> vertex: two dependent round trips, once per
        // workgroup) — with the two round trips of the first batch's candidate gathers issued next to them
        const int nF = p.nF;
        const bool pre = p.tri_pre != nullptr; // uniform: the triangles were laid out for this by k_tri_pre — one coalesced round trip
        int ia = 0, ib = 0, ic = 0, mm = 0, fm = 0;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0, r4 = r0;
        if (tid < nF) {
            if (pre) {
                const float4* tp = p.tri_pre + ((size_t)e * p.n_sub + step) * 5 * nF + tid;
                r0 = tp[0]; r1 = tp[(size_t)nF]; r2 = tp[(size_t)2 * nF]; r3 = tp[(size_t)3 * nF]; r4 = tp[(size_t)4 * nF];
            } else {
                ia = p.faces[3 * tid]; ib = p.faces[3 * tid + 1]; ic = p.faces[3 * tid + 2]; // stored order == caller order for small meshes
                mm = p.mesh_map[tid]; fm = p.face_map[tid];
            }
        }
        ImpPre ip;
        ip.on = false;
#ifndef R2S_FB_NO_IMP_PREFETCH
        if (WITH_SELF) imp_pre1(p, eb, ra.y & 0x7fffffff, tb0 + pi < n_rec && ra.y < 0, sl, ra.x & 0x7ffff, ip);
#endif
        f3 a = mk(0.f, 0.f, 0.f), b = a, c = a;
        if (tid < nF && !pre) { a = mesh_vertex(p, e, step, ia); b = mesh_vertex(p, e, step, ib); c = mesh_vertex(p, e, step, ic); }
#ifndef R2S_FB_NO_IMP_PREFETCH
        if (WITH_SELF) imp_pre2(p, po, eb, ip);
#endif
        if (tid < nF && pre) {
            sh.tri[tid][0] = r0; sh.tri[tid][1] = r1; sh.tri[tid][2] = r2;
            sh.tbox[tid][0] = r3.x; sh.tbox[tid][1] = r3.y; sh.tbox[tid][2] = r3.z; sh.tbox[tid][3] = r3.w; sh.tbox[tid][4] = r4.x; sh.tbox[tid][5] = r4.y;
        } else if (tid < nF) {
            int m = 0;
            for (int k = 1; k < p.n_mesh; ++k) m += tid >= p.mesh_face_off[k] ? 1 : 0;
            sh.tri[tid][0] = make_float4(a.x, a.y, a.z, b.x);
            sh.tri[tid][1] = make_float4(b.y, b.z, c.x, c.y);
            sh.tri[tid][2] = make_float4(c.z, __int_as_float(mm), __int_as_float(fm), __int_as_float(m));
            sh.tbox[tid][0] = fminf(a.x, fminf(b.x, c.x)); sh.tbox[tid][1] = fminf(a.y, fminf(b.y, c.y)); sh.tbox[tid][2] = fminf(a.z, fminf(b.z, c.z));
            sh.tbox[tid][3] = fmaxf(a.x, fmaxf(b.x, c.x)); sh.tbox[tid][4] = fmaxf(a.y, fmaxf(b.y, c.y)); sh.tbox[tid][5] = fmaxf(a.z, fmaxf(b.z, c.z));
        } else if (tid >= 128 && tid < 128 + p.n_mesh) {
            const int m = tid - 128;
            const float* bb = m < p.n_dyn_mesh ? p.aabb_dyn + (((size_t)e * p.n_sub + step) * p.n_dyn_mesh + m) * 6
                                               : p.aabb_static + ((size_t)e * (p.n_mesh - p.n_dyn_mesh) + (m - p.n_dyn_mesh)) * 6;
            sh.box[m][0] = make_float4(bb[0], bb[1], bb[2], bb[3]);
            sh.box[m][1] = make_float4(bb[4], bb[5], __int_as_float(((p.mesh_kind[m] & 2) ? 1 : 0) | (p.mesh_inward[m] ? 2 : 0)), 0.f);
        } else if (tid >= 192 && tid < 204) {
            const int k = tid - 192;
            sh.eef[k] = k < 3 ? p.interp_center[((size_t)e * p.n_sub + step) * 3 + k] : k < 6 ? p.dyn_omega[(size_t)e * 3 + (k - 3)] : p.dyn_vel[(size_t)e * 6 + (k - 6)];
        }
        __syncthreads();
        R2S_BSTAMP(); // staged
        BatchWave& w = sh.w[wave];
        for (int tb = tb0; tb < n_rec; tb += n_units * FB_PW) { // wave-uniform trip count
            const int t = tb + pi;
            const bool act = t < n_rec;
            if (tb != tb0) {
                ra = rec[2 * min(t, p.N - 1)]; rc = rec[2 * min(t, p.N - 1) + 1];
#ifndef R2S_FB_NO_IMP_PREFETCH
                if (WITH_SELF) { imp_pre1(p, eb, ra.y & 0x7fffffff, act && ra.y < 0, sl, ra.x & 0x7ffff, ip); imp_pre2(p, po, eb, ip); }
#endif
            }
            const bool tagged = act && ra.y < 0;
            float* st = w.st[pi]; // the particle's parked state: [0..2] x0, [3..5] v, [6] i | [7..9] next_x, [10..12] next_v, [13..15] per-face force, [16] margin, [17] flags, [18] face_map
            {
                const int i = act ? (ra.y & 0x7fffffff) : 0, cnt = ra.x & 0x7ffff;
                const f3 x0 = mk(__int_as_float(ra.z), __int_as_float(ra.w), __int_as_float(rc.x));
                f3 v = mk(__int_as_float(rc.y), __int_as_float(rc.z), __int_as_float(rc.w));
    #ifdef R2S_FB_NO_IMP_PREFETCH
                if (WITH_SELF && __builtin_amdgcn_ballot_w64(tagged) != 0ull) v = self_impulse<16>(p, po, eb, i, tagged, x0, v, sl, cnt);
#else
                if (WITH_SELF && __builtin_amdgcn_ballot_w64(tagged) != 0ull) v = self_impulse_pre(p, po, eb, i, tagged, x0, v, sl, cnt, ip);
#endif
                if (sl == 0) { st[0] = x0.x; st[1] = x0.y; st[2] = x0.z; st[3] = v.x; st[4] = v.y; st[5] = v.z; st[6] = __int_as_float(i); }
            }
            fb_wave_sync();
            R2S_BSTAMP(); // record + impulses done
            // mesh_collision (:295-421) — the arithmetic of finish_wave, expression for expression
            float d2_1, d2_2, lm[3] = {3.0e38f, 0.f, 3.0e38f}; // lm: this lane's smallest bound, its triangle, its second smallest bound (first query)
            MeshHit q;
            {
                const f3 x0 = mk(st[0], st[1], st[2]), vin = mk(st[3], st[4], st[5]);
                q = batch_query<false>(p, sh, w, x0 + vin * p.dt, act, 0.f, d2_1, lm R2S_BP_ARG);
            }
            R2S_BSTAMP(); // first query back
            bool requery = false;
            float thr2 = 0.f;
            f3 next_x;
            {
                const f3 x0 = mk(st[0], st[1], st[2]);
                f3 vin = mk(st[3], st[4], st[5]);
                next_x = x0 + vin * p.dt;
                const f3 q1x = next_x;
                f3 next_v = vin;
                bool hit = false;
                f3 normal = mk(0.f, 0.f, 0.f), v_normal = mk(0.f, 0.f, 0.f), v_normal_new = mk(0.f, 0.f, 0.f);
                float margin = 0.f;
                if (q.result) {
                    int is_gripper;
                    const int mm = q.mm;
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
                            const f3 ctr = mk(sh.eef[0], sh.eef[1], sh.eef[2]);
                            const f3 om = mk(sh.eef[3], sh.eef[4], sh.eef[5]);
                            const f3 dv = is_gripper == 1 ? mk(sh.eef[6], sh.eef[7], sh.eef[8]) : mk(sh.eef[9], sh.eef[10], sh.eef[11]);
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
                // the re-query's point is `delta` from the first one's: see batch_query<true> (bounds widened for the rounding of delta and of the square root)
                const float dq = len(next_x - q1x) * 1.0001f + 1e-9f;
                const float reach = sqrtf(d2_1) * 1.0001f + 2.f * dq;
                thr2 = reach * reach * 1.0001f + 1e-12f;
                if (sl == 0) {
                    const f3 fo = (v_normal_new - v_normal) / p.dt;
                    st[7] = next_x.x; st[8] = next_x.y; st[9] = next_x.z; st[10] = next_v.x; st[11] = next_v.y; st[12] = next_v.z;
                    st[13] = fo.x; st[14] = fo.y; st[15] = fo.z; st[16] = margin; st[17] = __int_as_float((hit ? 1 : 0) | (requery ? 2 : 0)); st[18] = __int_as_float(q.fm);
                }
            }
            fb_wave_sync();
            const MeshHit q2 = batch_query<true>(p, sh, w, next_x, requery, thr2, d2_2, lm R2S_BP_ARG);
            R2S_BSTAMP(); // response + second query back
            if (act && sl == 0) { // the particle's storing lane
                f3 x = mk(st[7], st[8], st[9]);
                const f3 vv = mk(st[10], st[11], st[12]);
                const int flags = __float_as_int(st[17]);
                int qface_fm = __float_as_int(st[18]);
                if (flags & 2) {
                    if (q2.result) {
                        const f3 delta = x - q2.pt;
                        const float dist = len(delta) * q2.sign;
                        const float err = dist - st[16];
                        if (err < 0.f) {
                            const f3 normal = normalize0(delta) * q2.sign;
                            x = x - normal * err;
                        }
                    }
                    qface_fm = q2.fm; // face of the LAST query (0 if the re-query missed)
                }
                if ((flags & 1) && write_forces) {
                    float* cf3 = p.coll_forces + ((size_t)e * p.nF + qface_fm) * 3;
                    atomicAdd(cf3, st[13]);
                    atomicAdd(cf3 + 1, st[14]);
                    atomicAdd(cf3 + 2, st[15]);
                    atomicAdd(p.hit_cnt + e, 1);
                }
                // integrate_ground_collision, :424-474
                const f3 gn = mk(0.f, 0.f, 1.f) * p.rf;
                const float x_z = x.z, v_z = vv.z;
                const float next_x_z = (x_z + v_z * p.dt) * p.rf;
                f3 v1;
                float toi;
                if (next_x_z < 0.f && v_z * p.rf < -1e-4f) {
                    const f3 vn = gn * dot(vv, gn);
                    const f3 v_tao = vv - vn;
                    const float vnl = len(vn);
                    const float vtl = fmaxf(len(v_tao), 1e-6f);
                    const f3 vn_new = vn * (-p.ce);
                    const float a = fmaxf(0.f, 1.f - p.cf * (1.f + p.ce) * vnl / vtl);
                    v1 = vn_new + v_tao * a;
                    toi = -(x_z - 0.f) / v_z;
                } else {
                    v1 = vv;
                    toi = 0.f;
                }
                const f3 xn = x + vv * toi + v1 * (p.dt - toi);
                const int i = __float_as_int(st[6]);
                if (PFOUT) pf_store(p, eb + (size_t)i, xn, v1, (unsigned)step + 1u);
                else st_store(xv_out, eb + i, xn, v1);
            }
            fb_wave_sync();
            R2S_BSTAMP(); // stored
        }
    }
#ifdef R2S_PHASE_PROBE
    if (probe_on && lane == 0 && probe_n > 1) g_query_probe[probe_row * 32 + 31] = probe_entry;
    else if (probe_on && lane == 0) g_query_probe[probe_row * 32] = 0; // (staged, but no records for this wavefront: not a row of the table)
#endif
#undef R2S_BSTAMP
#undef probe_n
#ifndef R2S_FB_NO_PART2
    if (WITH_SELF) {
        // part 2 behind part 1's workgroups of the same environment: slots [n1, nslot); if part 1 fills every slot (never in the scenes measured), all
        // of them take part 2 behind their part-1 work
        const int n1 = min((n_rec + 4 * FB_PW - 1) / (4 * FB_PW), nslot);
        const bool all = n1 >= nslot;
        finish_candidates_env<1, PFOUT>(p, xv_out, step, write_forces, e, slot >= nslot ? -1 : all ? slot : slot - n1, all ? nslot : nslot - n1, po);
    }
#endif
}

template <bool WITH_SELF>
__global__ void __launch_bounds__(256) k_contact_finish_batch(const PhysDev p, const StateC xv_in, const StateM xv_out, int step, int write_forces)
{
    __shared__ BatchShare sh;
    contact_finish_batch<WITH_SELF, false>(p, xv_out, step, write_forces, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y), sh);
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
// MESHQ 4 (round 6): the batched small-scene finishers (contact_finish_batch) — all four wavefronts of a finishing workgroup work, and the
// two roles share ONE LDS allocation (the window of the fused role is the larger one).
#ifndef R2S_PF_WAVES4
#define R2S_PF_WAVES4 4          // with candidates: the finishing role needs ~125 VGPRs; held to 96 (five wavefronts per SIMD) it spills 17 dwords and the held grasp runs 35.5 instead of 31.0 us per batched substep
#endif
#ifndef R2S_PF_WAVES4_NOSELF
#define R2S_PF_WAVES4_NOSELF 5   // without candidates it fits 96 without a spill: five (a hovering gripper: 19.5 instead of 20.6 us per batched substep)
#endif
template <int B, int RCAP, bool SELF, int MESH, int MESHQ>
__global__ void __launch_bounds__(B, (MESHQ == 3 ? (SELF ? R2S_PF_WAVES3 : R2S_PF_WAVES3_NOSELF) : MESHQ == 4 ? (SELF ? R2S_PF_WAVES4 : R2S_PF_WAVES4_NOSELF) : 1)) k_substep_pf(const PhysDev p, const StateC xv_in, const StateM xv_out, int step, int write_forces, int fin_skip)
{
    if constexpr (MESHQ == 4) {
        static_assert(B == 256, "the batched finishers are 256-thread workgroups");
        constexpr size_t WIN = sizeof(v2f) * 3 * (RCAP + 1), SH = sizeof(BatchShare);
        __shared__ __attribute__((aligned(16))) char smem[WIN > SH ? WIN : SH];
        if ((int)blockIdx.x < p.pf_nfin) {
            if (fin_skip) return;
            contact_finish_batch<SELF, true>(p, xv_out, step - 1, 0, (int)blockIdx.x, p.pf_nfin, *(BatchShare*)smem);
            return;
        }
        substep_body<B, RCAP, SELF, MESH, true, true>(p, xv_in, xv_out, step, write_forces, (int)blockIdx.x - p.pf_nfin, (__attribute__((address_space(3))) v2f*)smem);
    } else {
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

