"""Round 6: the BATCHED finishing code of small scenes (csrc/physics_finish.h contact_finish_batch: a workgroup takes 16 records of one
environment's list at once, lane = (particle, triangle slice), the substep's triangles staged in LDS) against the form of rounds 2-5 (one
workgroup per listed particle, one triangle per lane, R2S_FIN_BATCH=0).  Same closest-point arithmetic and (distance^2, face id) order, the
winding number restricted to the meshes that can contribute (a closed manifold adds nothing outside its box), the same response arithmetic
in the same order, candidates summed in the order of the 16-lane groups: the states must agree BIT FOR BIT, every env step, through the
closing ramp, the latched grasp (spring_mass_warp.py:295-421 with the gripper branch's re-query) and the lift — at the head of the next
launch and as the stand-alone finishing launch.  What holds the result to the reference is the rest of the GPU suite: batched is the default."""
import os

import numpy as np
import pytest

from util_parity import record

pytestmark = pytest.mark.gpu


def _rollout(cfg, n_env, batch, pf, steps, close_at=2, close_rate=0.1, **kw):
    import torch
    from r2s_hip.rollout import BatchedRollout

    old = os.environ.get("R2S_FIN_BATCH")
    os.environ["R2S_FIN_BATCH"] = "1" if batch else "0"      # read by r2s_phys_create
    try:
        ro = BatchedRollout(cfg, n_env=n_env, close_at=close_at, close_rate=close_rate, **kw)
    finally:
        if old is None:
            del os.environ["R2S_FIN_BATCH"]
        else:
            os.environ["R2S_FIN_BATCH"] = old
    ro.phys.set_pf(pf)
    xs, vs, fl, fo, gr = [], [], [], [], []
    for _ in range(steps):
        ro.physics_step()
        ro.t += 1
        xs.append(ro.phys.x.clone()); vs.append(ro.phys.v.clone()); fl.append(ro.phys.last_flavour())
        fo.append(ro.phys.collision_forces().clone())
        gr.append(ro.phys.eef_state()[1].clone() if not ro.use_pusher else None)
    torch.cuda.synchronize()
    st = ro.contact_stats()
    ro.phys.step(0, 0)            # a sticky fault would raise here
    torch.cuda.synchronize()
    return dict(x=[x.cpu().numpy() for x in xs], v=[v.cpu().numpy() for v in vs], fl=fl, f=[f.cpu().numpy() for f in fo],
                grasped=[None if g is None else g.cpu().numpy() for g in gr], stats=st, deferred=ro.phys.deferred_counts(), tagged=ro.phys.tagged_count())


@pytest.mark.parametrize("pf", [True, False], ids=["finishers at the head of the next launch", "stand-alone finishing launch"])
def test_grasp_of_the_toy_batched_finishing_equals_one_workgroup_per_particle_bit_for_bit(pf):
    """9 environments of the headline scene (two chains): closing ramp from env step 2, the pads load up, the grasp latches from the
    stepper's own forces around step 12, lift.  Deferred queries, tagged records (candidates + mesh), re-queries of the gripper branch."""
    steps = 16
    a = _rollout("sloth_32env", 9, True, pf, steps)
    b = _rollout("sloth_32env", 9, False, pf, steps)
    assert [f["kernel"] for f in a["fl"]] == [f["kernel"] for f in b["fl"]]
    assert a["stats"]["mesh_contacts"] > 9 * 20 and a["stats"]["self_collision_candidates"] > 0 and a["tagged"] > 0, (a["stats"], a["tagged"])
    assert a["stats"]["grasped_envs"] == 9, a["stats"]
    assert np.array_equal(a["deferred"], b["deferred"]) and a["tagged"] == b["tagged"]
    worst_f = 0.0
    for k in range(steps):
        assert np.array_equal(a["x"][k], b["x"][k]) and np.array_equal(a["v"][k], b["v"][k]), (k, float(np.abs(a["x"][k] - b["x"][k]).max()))
        assert np.array_equal(a["grasped"][k], b["grasped"][k]), k
        # per-face forces are float atomics in both forms (their order is the hardware's): equal up to that
        sc = max(1.0, float(np.abs(b["f"][k]).max()))
        worst_f = max(worst_f, float(np.abs(a["f"][k] - b["f"][k]).max()) / sc)
    assert worst_f < 1e-5, worst_f
    record(f"batched finishing vs one workgroup per particle, toy grasp, pf={pf}", env_steps=steps, envs=9, mesh_contacts=a["stats"]["mesh_contacts"],
           deferred_per_substep_max=int(a["deferred"][:-1].max()), tagged=a["tagged"], x_max_abs=0.0, force_rel=worst_f, tol=0)


def test_rope_grasp_small_layout_stand_alone_launch():
    """The rope of configs[0] forced onto the per-substep kernels + finishing launch (the 64-particle layout's deferred flavour: what a small
    batch runs when its resident launch ran out of server units): batched vs one workgroup per particle."""
    import torch
    from r2s_hip.rollout import BatchedRollout

    res = []
    for batch in (True, False):
        os.environ["R2S_FIN_BATCH"] = "1" if batch else "0"
        try:
            ro = BatchedRollout("rope_1env", close_at=2, close_rate=0.1)
        finally:
            del os.environ["R2S_FIN_BATCH"]
        ro.phys.set_resident(False)          # per-substep kernels of the 64-particle layout; a needed query then defers (k_contact_finish)
        xs = []
        for _ in range(14):
            ro.physics_step(); ro.t += 1
            xs.append(ro.phys.x.clone())
        torch.cuda.synchronize()
        res.append(([x.cpu().numpy() for x in xs], ro.contact_stats(), ro.phys.last_flavour(), ro.phys.deferred_counts()))
    (xa, sa, fa, da), (xb, sb, fb, db) = res
    assert fa["kernel"] == fb["kernel"] and "k_contact_finish" in fa["kernel"], fa
    assert sa["mesh_contacts"] > 0 and da[:-1].max() > 0 and np.array_equal(da, db)
    for k in range(len(xa)):
        assert np.array_equal(xa[k], xb[k]), (k, float(np.abs(xa[k] - xb[k]).max()))
