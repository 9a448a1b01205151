"""Which kernels an env step runs, enumerated WITHOUT a GPU: flavour selection is a pure function of the counters
of the env step two before, the handle's capabilities and the switches (csrc/physics_flavour.h, exported as
r2s_phys_debug_pick_flavour; r2s_phys_step calls the same function).  Every row below is one input -> the kernel string, the graph
slot, the number of chains and the servers it must yield.  The reference has a single flavour (spring_mass_warp.py:823-943 launches the
same nine kernels every substep): all of these run its arithmetic with different work splits; `sum_class` says which of them end in
the same bits (tests/test_flavour_pairs_gpu.py holds that on the device)."""
import itertools
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "real2sim-eval_amd"))


def pick(**kw):
    from r2s_hip.physics import pick_flavour

    return pick_flavour(**kw)


# ---- handles as r2s_phys_create would describe them ----------------------------------------------------------------------------------
def large_batch(mesh="small", **kw):
    """32 sloth environments (59 blocks of 256 particles each): the headline's handle.  mesh: none / small (two fingers + box) / large (rod)."""
    d = dict(block=256, n_blocks=59, n_env=32, self_collision=1, n_cu=256, srv_wg_cap=128, resident_pref=1, res_self=1, res_self_srv=1, pf_pref=1,
             srv_own=1, n_substeps=667, full_step=1, n_faces={"none": 0, "small": 100, "large": 24448 + 12}[mesh], any_large=int(mesh == "large"),
             pf_ok=int(mesh != "none"))
    d.update(kw)
    return d


def small_batch(mesh="small", **kw):
    """One rope environment (130 blocks of 64 particles): resident layout, servers possible for a small scene."""
    d = dict(block=64, n_blocks=130, n_env=1, split_ok=1, resident_ok=1, srv_ok=int(mesh == "small"), has_vx=1, self_collision=1, n_cu=256, srv_wg_cap=128,
             resident_pref=1, res_self=1, res_self_srv=1, pf_pref=1, srv_own=1, n_substeps=667, full_step=1,
             n_faces={"none": 0, "small": 100, "large": 24448}[mesh], any_large=int(mesh == "large"))
    if mesh == "large":
        d["resident_ok"] = 0       # (r2s_phys_create: no resident launch next to a large mesh)
    d.update(kw)
    return d


CASES = [
    # ---- large batch ------------------------------------------------------------------------------------------------------------------
    ("large, no meshes, free", large_batch("none"), dict(kernel="k_substep<256,1024,false,0>", chains=4, graph_slot=0, resident=0, pf=0)),
    ("large, no meshes, candidates", large_batch("none", n_candidates=7), dict(kernel="k_substep<256,1024,true,0> + k_self_finish", chains=4, graph_slot=2)),
    ("large, small meshes, new history", large_batch(), dict(kernel="k_substep<256,1024,false,1>", mesh_defer=0, graph_slot=0)),
    ("large, small meshes, nothing near", large_batch(have_counters=1), dict(kernel="k_substep<256,1024,false,1>", mesh_defer=0, pf=0)),
    ("large, small meshes, near", large_batch(have_counters=1, near_mesh=3),
     dict(kernel="k_substep_pf<256,1024,false,1> (finishers of substep k at the head of substep k+1's launch)", mesh_defer=1, pf=1, contact_finish=1, graph_slot=4, chains=4)),
    ("large, small meshes, near + candidates", large_batch(have_counters=1, near_mesh=3, n_candidates=900),
     dict(kernel="k_substep_pf<256,1024,true,1> (finishers of substep k at the head of substep k+1's launch)", graph_slot=6, variant=1)),
    ("large, near, pf switched off", large_batch(have_counters=1, near_mesh=1, n_candidates=5, pf_pref=0),
     dict(kernel="k_substep<256,1024,true,1> + k_contact_finish", pf=0, contact_finish=1, graph_slot=6)),
    ("large, needed but not near (cannot happen; near wins)", large_batch(have_counters=1, query_needed=1), dict(mesh_defer=0, kernel="k_substep<256,1024,false,1>")),
    ("large, candidates only", large_batch(have_counters=1, n_candidates=5), dict(kernel="k_substep<256,1024,true,1> + k_self_finish", graph_slot=2)),
    ("large, forced in place", large_batch(have_counters=1, near_mesh=9, force_defer=0), dict(mesh_defer=0, kernel="k_substep<256,1024,false,1>")),
    ("large, forced deferred", large_batch(force_defer=1), dict(mesh_defer=1, pf=1)),
    ("large mesh always defers", large_batch("large", n_blocks=9), dict(kernel="k_substep_pf<256,1024,false,2> (finishers of substep k at the head of substep k+1's launch)", mesh_defer=1, mesh=2, chains=2, graph_slot=4)),
    ("large mesh, forced in place is refused", large_batch("large", n_blocks=9, force_defer=0), dict(mesh_defer=1)),
    ("large mesh, two launches", large_batch("large", n_blocks=9, pf_pref=0, n_candidates=3), dict(kernel="k_substep<256,1024,true,2> + k_contact_finish", graph_slot=6)),
    ("chains: 9 envs x 59 blocks", large_batch(n_env=9), dict(chains=2)),
    ("chains: 2 envs x 59 blocks", large_batch(n_env=2), dict(chains=1)),
    ("chains override", large_batch(chains_override=3), dict(chains=3)),
    ("chains override capped by environments", large_batch(n_env=2, chains_override=8), dict(chains=2)),
    ("eager partial step: one chain", large_batch(full_step=0, n_substeps=20), dict(chains=1)),
    ("128-particle layout has no pf", large_batch(block=128, pf_ok=0, have_counters=1, near_mesh=1), dict(kernel="k_substep<128,768,false,1> + k_contact_finish", pf=0)),
    # ---- small batch (resident layout) -----------------------------------------------------------------------------------------------
    ("small, no meshes, free", small_batch("none"), dict(kernel="k_steps_resident<512,false,0>", resident=1, n_srv=0, chains=1)),
    ("small, no meshes, candidates", small_batch("none", n_candidates=40), dict(kernel="k_steps_resident<512,true,0>", resident=1, variant=1)),
    ("small, candidates, resident self flavour off", small_batch("none", n_candidates=40, res_self=0),
     dict(kernel="k_steps_resident<512,true,0> x 1 substep + k_self_finish", resident=0)),
    ("small, one substep with candidates is per-substep", small_batch("none", n_candidates=40, n_substeps=1, full_step=0), dict(resident=0)),
    ("small scene, free: owning servers ride along", small_batch(),
     dict(kernel="k_steps_resident<512,false,1> + 120 query-server workgroups in the launch (a quad of wavefronts owns its particle from the claim on)", n_srv=120, srv_own=1, srv_quad=1, mesh_defer=0)),
    ("small scene, query needed: stays resident", small_batch(have_counters=1, near_mesh=1, query_needed=1), dict(resident=1, mesh_defer=0, n_srv=120)),
    ("small scene, pairs forced", small_batch(srv_quad=0), dict(srv_quad=0, kernel="k_steps_resident<512,false,1> + 120 query-server workgroups in the launch (a pair of wavefronts owns its particle from the claim on)")),
    ("small scene, request protocol", small_batch(srv_own=0), dict(srv_own=0, kernel="k_steps_resident<512,false,1> + 120 query-server workgroups in the launch (a request per substep)")),
    ("small scene, servers ran out", small_batch(have_counters=1, near_mesh=1, query_needed=1, servers_ran_out=1),
     dict(resident=0, mesh_defer=1, srv_exhausted=1, kernel="k_steps_resident<512,false,1> x 1 substep + k_contact_finish")),
    ("small scene, exhausted stays while queries are needed", small_batch(have_counters=1, near_mesh=1, query_needed=1, srv_exhausted=1), dict(resident=0, srv_exhausted=1)),
    ("small scene, exhausted clears when nothing is needed", small_batch(have_counters=1, srv_exhausted=1), dict(resident=1, srv_exhausted=0)),
    ("small scene, candidates, nothing near: no servers", small_batch(have_counters=1, n_candidates=30),
     dict(kernel="k_steps_resident<512,true,1>", self_srv=0, n_srv=0, graph_slot=2)),
    # round 6: the answering servers come with NEAR, two steps before the first particle can be inside a margin
    ("small scene, candidates, NEAR: answering servers", small_batch(have_counters=1, n_candidates=30, near_mesh=1),
     dict(kernel="k_steps_resident<512,true,1> + 120 query-server workgroups in the launch (a request per substep)", self_srv=1, srv_own=0, n_srv=120, graph_slot=10, resident=1)),
    ("small scene, candidates, needed", small_batch(have_counters=1, n_candidates=30, near_mesh=1, query_needed=1), dict(self_srv=1, resident=1, mesh_defer=0)),
    ("small scene, candidates, servers always (switch 2)", small_batch(n_candidates=30, res_self_srv=2), dict(self_srv=1, n_srv=120)),
    ("small scene, candidates, self servers off + needed: per-substep kernels", small_batch(have_counters=1, n_candidates=30, near_mesh=1, query_needed=1, res_self_srv=0),
     dict(self_srv=0, resident=0, mesh_defer=1, kernel="k_steps_resident<512,true,1> x 1 substep + k_contact_finish")),
    ("small scene, candidates, self servers off, only near: resident without servers", small_batch(have_counters=1, n_candidates=30, near_mesh=1, res_self_srv=0),
     dict(self_srv=0, resident=1, n_srv=0)),
    ("small, resident switched off", small_batch(resident_pref=0, have_counters=1, near_mesh=1), dict(resident=0, mesh_defer=1, kernel="k_steps_resident<512,false,1> x 1 substep + k_contact_finish")),
    ("small, resident switched off, free", small_batch(resident_pref=0), dict(resident=0, kernel="k_steps_resident<512,false,1> x 1 substep")),
    ("small, two chains forced: no servers", small_batch(n_env=2, n_blocks=60, chains_override=2), dict(n_srv=0, resident=1)),
    ("small, too few CUs left for servers", small_batch(n_blocks=250), dict(n_srv=0)),
    ("small, server cap", small_batch(srv_wg_cap=16), dict(n_srv=16, srv_quad=0)),
    ("small batch next to a large mesh: per-substep kernels, always deferred", small_batch("large"), dict(resident=0, mesh_defer=1, mesh=2, kernel="k_steps_resident<512,false,2> x 1 substep + k_contact_finish")),
    ("small batch whose slices do not fit the resident registers", small_batch("small", split_ok=0, resident_ok=0, srv_ok=0, have_counters=1, near_mesh=1),
     dict(kernel="k_substep<64,512,false,1> + k_contact_finish", resident=0)),
]


@pytest.mark.parametrize("name,fin,want", CASES, ids=[c[0] for c in CASES])
def test_flavour_of(name, fin, want):
    got = pick(**fin)
    for k, v in want.items():
        assert got[k] == v, (name, k, got)


def test_the_matrix_is_wide_enough():
    assert len(CASES) >= 24


def test_invariants_over_the_whole_input_space():
    """Every combination of {batch} x {mesh} x {candidates} x {near, needed, ran out} x {switches}: the output is consistent with itself."""
    n = 0
    for batch, mesh in itertools.product((large_batch, small_batch), ("none", "small", "large")):
        for cand, have, near, need, out_, pf, res, rss, fd in itertools.product((0, 5), (0, 1), (0, 1), (0, 1), (0, 1), (0, 1), (0, 1), (0, 1, 2), (-1, 0, 1)):
            fin = batch(mesh, n_candidates=cand, have_counters=have, near_mesh=near, query_needed=need, servers_ran_out=out_, pf_pref=pf, resident_pref=res,
                        res_self_srv=rss, force_defer=fd)
            o = pick(**fin)
            n += 1
            assert o["variant"] == int(cand > 0) and o["mesh"] == {"none": 0, "small": 1, "large": 2}[mesh]
            if mesh == "none":
                assert o["mesh_defer"] == 0 and o["pf"] == 0 and o["n_srv"] == 0 and o["contact_finish"] == 0
            if mesh == "large":
                assert o["mesh_defer"] == 1 and not o["resident"]
            if o["resident"]:
                assert not o["mesh_defer"] and not o["pf"] and o["chains"] == 1 and fin["block"] == 64
                assert o["kernel"].startswith("k_steps_resident<512,") and "x 1 substep" not in o["kernel"]
            else:
                assert o["n_srv"] == 0
            if o["pf"]:
                assert o["contact_finish"] and fin["block"] == 256 and "k_substep_pf" in o["kernel"]
            if o["self_srv"]:
                assert o["variant"] == 1 and o["resident"] and o["n_srv"] > 0 and not o["srv_own"]
            if o["n_srv"]:
                assert 8 <= o["n_srv"] <= fin["srv_wg_cap"] and 8 * ((fin["n_blocks"] * fin["n_env"] + 7) // 8) + o["n_srv"] <= fin["n_cu"]
            assert 0 <= o["graph_slot"] <= 10 and o["graph_slot"] % 2 == 0
            # flavours that share a graph slot are the same flavour
            assert o["graph_slot"] == (8 if (o["variant"] and not o["mesh_defer"] and o["self_srv"]) else 4 * o["mesh_defer"]) + 2 * o["variant"]
    assert n == 6 * 2 ** 7 * 9


def test_switching_pf_or_chains_never_changes_the_sum_class():
    """pf and the number of chains are work SPLITS of the same arithmetic in the same order: bit-identical states (held on the device by
    tests/test_pf_gpu.py and tests/test_flavour_pairs_gpu.py); everything else may sum in another order."""
    for near, cand in itertools.product((0, 1), (0, 9)):
        base = large_batch(have_counters=1, near_mesh=near, n_candidates=cand)
        a = pick(**base)
        assert pick(**dict(base, pf_pref=0))["sum_class"] == a["sum_class"]
        assert pick(**dict(base, chains_override=1))["sum_class"] == a["sum_class"]
        assert pick(**dict(base, force_defer=1 - a["mesh_defer"]))["sum_class"] != a["sum_class"]
    s = small_batch(have_counters=1)
    assert pick(**s)["sum_class"] != pick(**dict(s, resident_pref=0))["sum_class"]
    assert pick(**s)["sum_class"] != pick(**dict(s, srv_own=0))["sum_class"]
