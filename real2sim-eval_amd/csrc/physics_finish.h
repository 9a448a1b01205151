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

