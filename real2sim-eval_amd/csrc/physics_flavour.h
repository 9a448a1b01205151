// physics_flavour.h — which flavour an env step runs, as a PURE host function (round 6; included by physics.hip ahead of everything else).
//
// Until round 5 this was a dozen booleans inside r2s_phys_step, re-derived in part by enqueue_steps and by the capture loop of
// r2s_phys_create.  Everything the choice depends on is now one POD (R2SFlavourIn, include/r2s_physics.h): counters of the env step two
// before this one, the handle's capabilities, the switches — and everything it yields another (R2SFlavourOut).  r2s_phys_step fills the
// input from its handle (flavour_input) and calls pick_flavour; enqueue_steps and the capture loop ask the same helpers (fl_*) with the
// same input, so the graph that is captured for a slot is the graph pick_flavour names for it.  No device, no handle, no state in here:
// tests/test_flavour_matrix.py enumerates the function through r2s_phys_debug_pick_flavour without a GPU.
//
// The rules, in the order they apply (the reference has ONE flavour: spring_mass_warp.py:823-943 launches the same nine kernels every
// substep; all flavours below run that arithmetic with different work splits):
//   variant      1 while any particle has self-collision candidates (the count of update_collision_graph, synchronous): SELF templates.
//   mesh         0 no meshes / 1 every mesh small (<= 256 faces) / 2 a large mesh is present.
//   mesh_defer   needy particles are LISTED by the fused kernel and finished by the finishing code.  Large batches: as soon as anything
//                was NEAR a mesh (margin + 3 cm) two steps ago — an idle finishing head costs ~0.7 us per substep, an in-place query in the
//                fused kernel up to 190.  Small batches (resident layout): once a query was NEEDED — their free flavour is the resident
//                launch, worth keeping while the gripper hovers.  A scene with a large mesh always defers.  `force_defer` overrides.
//   resident     small batch (resident_ok), preference on, no deferred queries, and — with candidates — the self-collision flavour of the
//                resident launch available (res_self, the record array, more than one substep in the launch).
//   servers      a resident launch of a small scene carries query-server workgroups (srv_ok, one chain, more than one substep): always in
//                the flavour without candidates (they OWN their particles); in the self-collision flavour (they ANSWER) only while they may
//                be needed — round 6: from the NEAR counter (anything within margin + 3 cm of the launch's mesh boxes two steps ago), so
//                the servers are in the launch BEFORE the first particle enters a margin.  Until round 5 this followed "a query was
//                NEEDED": the first two env steps of a contact on top of live candidates answered in place, 118 ms each on the
//                one-environment toy against 8.6 (DESIGN §8 item 4 of round 5).  With servers available a small batch never defers;
//                a launch that ran out of server units — or, round 6, claimed more than 30 % of them: a closing grasp doubles its contacts in the two
//                steps the counters lag — (servers_ran_out) sends the following steps to the per-substep kernels + finishing
//                launch until no query is needed any more (srv_exhausted, sticky).
//   pf           large-batch layout with meshes, preference on, and the flavour carries finishing code: the finishers of substep k ride
//                at the head of substep k + 1's launch (k_substep_pf), bit-identical to the two-launch form.
//   chains       4 from 1536 work items, 2 from 256, else 1; the 64-particle layout always 1; never more than environments.
#pragma once
#include "../../include/r2s_physics.h"
#include <algorithm>
#include <cstdio>
#include <cstring>

namespace r2s_flavour {

constexpr int FL_SRV_MIN_WG = 8;  // (= SRV_MIN_WG, physics_resident.h: fewer server workgroups than this are not worth the claims)

inline int fl_mesh(const R2SFlavourIn& c) { return c.n_faces > 0 ? (c.any_large ? 2 : 1) : 0; }

inline int fl_chains(const R2SFlavourIn& c)
{
    // chains are separate graphs on separate streams (hardware queues; more than four lose: 26 / 42 us per substep with six).
    // Measured per batched substep, free / contact (tools/profiling/variant_bench.py, round 3): 32 sloth envs (1888 work items)
    // 1 chain 23.7 / 25.7, 2 chains 20.8 / 26.8, 4 chains 19.1 / 26.0 us; 8 sloth envs x 4 views (472 items) 10.6 / 17.3,
    // 9.5 / 17.2, 10.6 / 19.3; 32 T-block envs with the 25k-face rod (288 items) 10.4 / 25.9, 10.5 / 24.4, 12.7 / 26.3.
    const int64_t items = (int64_t)c.n_blocks * c.n_env;
    int n = items >= 1536 ? 4 : (items >= 256 ? 2 : 1);
    if (c.block == 64) n = 1; // small batches (the resident layout): one chain
    n = std::min(n, c.n_env);
    if (c.chains_override > 0) n = std::max(1, std::min(c.chains_override, std::min(c.n_env, 8)));
    return n;
}

// the flavour carries k_contact_finish (deferred mesh queries + self-collision impulses in one finishing kernel)
inline bool fl_contact_finish(const R2SFlavourIn& c, int mesh_defer) { return fl_mesh(c) != 0 && (mesh_defer || fl_mesh(c) == 2); }

// ... with its finishers at the head of the next launch
inline bool fl_pf(const R2SFlavourIn& c, int mesh_defer) { return c.pf_ok && c.pf_pref != 0 && c.block == 256 && fl_contact_finish(c, mesh_defer); }

// the env step as ONE resident launch (a single substep with candidates is the per-substep form: k_steps_resident x 1 + k_self_finish)
inline bool fl_resident(const R2SFlavourIn& c, bool with_self, int mesh_defer, int n)
{
    return c.resident_ok && c.resident_pref != 0 && !(with_self && !(c.res_self && c.has_vx && n > 1)) && !(c.n_faces > 0 && mesh_defer);
}

// server workgroups of a resident launch: as many as the chip has CUs left behind the blocks' own (the whole launch is resident at once;
// workgroups go to the XCDs round-robin, so the grid — 8 * cb block workgroups + servers — must not exceed the CUs)
inline int fl_n_srv(const R2SFlavourIn& c, bool with_self, int self_srv, int n)
{
    if (!(c.srv_ok && n > 1 && fl_chains(c) == 1 && (!with_self || (c.res_self_srv && self_srv)))) return 0;
    const int cb = (int)(((int64_t)c.n_blocks * c.n_env + 7) / 8);
    const int n_srv = std::min(c.n_cu - 8 * cb, c.srv_wg_cap);
    return n_srv < FL_SRV_MIN_WG ? 0 : n_srv;
}
inline int fl_srv_quad(const R2SFlavourIn& c, int n_srv) { return c.srv_quad >= 0 ? c.srv_quad : (n_srv >= 64 ? 1 : 0); }

// slot of the captured graph (the caller adds the state buffer's parity): {queries in place / resident, deferred} x {no candidates,
// candidates}, and the resident self-collision flavour WITH servers on a slot of its own
inline int fl_graph_slot(int variant, int mesh_defer, int self_srv) { return (variant == 1 && !mesh_defer && self_srv ? 8 : mesh_defer * 4) + variant * 2; }

inline void pick_flavour(const R2SFlavourIn& c, R2SFlavourOut& o)
{
    std::memset(&o, 0, sizeof o);
    const int n = c.n_substeps;
    const int variant = (c.self_collision && c.n_candidates > 0) ? 1 : 0;
    const int mesh = fl_mesh(c);
    const bool small_batch = c.resident_ok && c.resident_pref;
    const int chains = fl_chains(c);
    int mesh_defer = 0, self_srv = 0, exhausted = c.srv_exhausted;
    if (mesh != 0) {
        mesh_defer = (small_batch ? c.query_needed : c.near_mesh) > 0 ? 1 : 0;
        if (c.have_counters) { // (more particles in contact than the launch had units: answered in place — correct, and 20 x slower than the finishing launch)
            if (c.servers_ran_out > 0) exhausted = 1;
            else if (c.query_needed == 0) exhausted = 0;
        }
        const bool self_srv_ok = c.res_self && c.res_self_srv && c.has_vx && n > 1;
        const bool srv_avail = small_batch && c.srv_ok && chains == 1 && !exhausted;
        self_srv = variant == 1 && self_srv_ok && srv_avail && (c.near_mesh > 0 || c.query_needed > 0 || c.res_self_srv == 2) ? 1 : 0;
        if (srv_avail && (variant == 0 || self_srv_ok)) mesh_defer = 0; // stays resident THROUGH contact: a needy particle goes to a server unit of the launch
        if (c.force_defer >= 0) mesh_defer = c.force_defer;              // test / tuning: force a flavour
        if (c.any_large) mesh_defer = 1;
    }
    const bool resident = fl_resident(c, variant == 1, mesh_defer, n);
    if (!resident) self_srv = 0; // (a forced deferred flavour: the servers belong to the resident launch)
    o.variant = variant; o.mesh = mesh; o.mesh_defer = mesh_defer; o.self_srv = self_srv; o.resident = resident ? 1 : 0;
    o.srv_exhausted = exhausted;
    o.contact_finish = !resident && fl_contact_finish(c, mesh_defer) ? 1 : 0;
    o.pf = !resident && fl_pf(c, mesh_defer) ? 1 : 0;
    o.chains = c.full_step ? chains : 1;
    o.graph_slot = fl_graph_slot(variant, mesh_defer, self_srv);
    if (resident) {
        o.n_srv = fl_n_srv(c, variant == 1, self_srv, n);
        o.srv_quad = o.n_srv > 0 ? fl_srv_quad(c, o.n_srv) : 0;
        o.srv_own = o.n_srv > 0 && c.srv_own && variant == 0 ? 1 : 0; // next to the self-collision flavour the units only ANSWER
        o.chains = 1;
    }
    // which flavours of one scene end in the same bits: chains and pf never change them (same kernels' arithmetic on the same inputs in the
    // same order); everything else sums in another order (in-place vs listed queries, 16 vs 64 lanes over a candidate list, the resident
    // launch's eight partial force sums, a server unit's fixed trees)
    o.sum_class = resident ? 100 + 10 * variant + (o.n_srv > 0 ? (o.srv_own ? 2 : 1) : 0) : (c.split_ok ? 50 : (c.block == 256 ? 0 : (c.block == 128 ? 20 : 30))) + 2 * variant + mesh_defer; // (per layout: a block's window decides which neighbours are gathered from LDS and which from memory — the same sums, but nothing holds them to the same bits)
    const int rcap = c.block == 256 ? 1024 : (c.block == 128 ? 768 : 512);
    const char* sc = variant ? "true" : "false";
    char* k = o.kernel;
    const size_t cap = sizeof o.kernel;
    if (resident) {
        int w = snprintf(k, cap, "k_steps_resident<%d,%s,%d>", rcap, sc, mesh);
        if (o.n_srv > 0)
            snprintf(k + w, cap - (size_t)w, " + %d query-server workgroups in the launch (%s)", o.n_srv,
                     o.srv_own ? (o.srv_quad ? "a quad of wavefronts owns its particle from the claim on" : "a pair of wavefronts owns its particle from the claim on")
                               : "a request per substep");
    } else {
        int w;
        if (c.split_ok) w = snprintf(k, cap, "k_steps_resident<%d,%s,%d> x 1 substep", rcap, sc, mesh);
        else w = snprintf(k, cap, "%s<%d,%d,%s,%d>", o.pf ? "k_substep_pf" : "k_substep", c.block, rcap, sc, mesh);
        snprintf(k + w, cap - (size_t)w, "%s", o.pf ? " (finishers of substep k at the head of substep k+1's launch)" : o.contact_finish ? " + k_contact_finish" : (variant ? " + k_self_finish" : ""));
    }
}

} // namespace r2s_flavour
