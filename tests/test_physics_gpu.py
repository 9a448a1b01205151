"""GPU parity of the fused HIP substep kernel against the CPU oracle, through the C ABI
(r2s_hip.physics.PhysBatch / the drop-in sim.physics.SpringMassSystemWarp).

Tolerance (BASELINE.json north_star): particle positions within 1e-5 abs.  The reference sums spring forces
with float atomics in a non-deterministic order; the HIP path sums in adjacency order and the oracle in
spring order, so agreement is to rounding, not bit-exact."""
import os

import numpy as np
import pytest

from util_physics import cfg, gripper_motion, hip_env, make_object, oracle_env, two_blobs
from util_parity import close

pytestmark = pytest.mark.gpu
ATOL = 1e-5


def _run_pair(ob, n_steps, n_sub, update_graph=False, **kw):
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, **kw)
    o.total_candidates = 0
    for _ in range(n_steps):
        if update_graph:
            o.update_collision_graph()
            h.update_collision_graph()
            o.total_candidates += int(o.coll_num.sum())
        o.step()
        h.step()
    return o, h


def test_free_fall_and_springs_contact_free_full_env_step():
    """Q9: 667 substeps, no ground contact, no meshes, no self collision."""
    ob = make_object("rope", 800, seed=3, lift=0.3)
    ob["points"][:, 2] += 0.01 * np.sin(40 * ob["points"][:, 0])  # pre-strain the springs (peak speed ~2 m/s)
    o, h = _run_pair(ob, 1, 667, self_collision=False)
    x = h.x[0].cpu().numpy(); v = h.v[0].cpu().numpy()
    assert close(x, o.x, ATOL)
    # error budget: the float64 shadow of the oracle bounds what float32 rounding alone does on this input
    o64 = oracle_env(ob, f64=True, num_substeps=667, self_collision=False); o64.step()
    assert close(x, o64.x, ATOL)
    assert close(v, o.v, 2e-3)  # velocities are O(1) m/s; positions are the gated quantity
    assert np.abs(o.x - ob["points"]).max() > 1e-3  # something actually moved


def test_ground_contact_rope_drop():
    ob = make_object("rope", 700, seed=1, lift=0.0005)
    ob["v0"] = np.zeros_like(ob["points"]); ob["v0"][:, 2] = -0.5; ob["v0"][:, 0] = 0.2
    o, h = _run_pair(ob, 1, 200, self_collision=False)
    x = h.x[0].cpu().numpy()
    assert close(x, o.x, ATOL)
    assert (o.x[:, 2] >= -1e-6).all()


def test_reverse_z_ground():
    ob = make_object("rope", 300, seed=2)
    ob["points"][:, 2] *= -1.0
    ob["v0"] = np.zeros_like(ob["points"]); ob["v0"][:, 2] = +0.5
    o, h = _run_pair(ob, 1, 100, self_collision=False, reverse_z=True)
    assert close(h.x[0].cpu().numpy(), o.x, ATOL)


def test_candidate_lists_match_oracle_exactly():
    ob = two_blobs(seed=0, gap=0.003, speed=0.0)
    # build resting set while the blobs are far apart, then bring them into contact range
    far = dict(ob); far["points"] = ob["points"].copy(); nA = len(ob["points"]) // 2
    far["points"][nA:, 0] += 0.2
    o = oracle_env(far, num_substeps=10)
    h = hip_env(far, num_substeps=10)
    import torch
    o.x[:] = ob["points"]
    h.set_state(torch.from_numpy(ob["points"])[None])
    mo = o.update_collision_graph()
    h.update_collision_graph()
    num, idx = h.collision_lists()
    num = num[0].cpu().numpy(); idx = idx[0].cpu().numpy()
    assert mo > 0 and h.collision_max_count() == mo
    assert np.array_equal(num, o.coll_num)
    for i in np.nonzero(num)[0]:
        assert np.array_equal(idx[i, : num[i]], o.coll_idx[i, : num[i]]), i


def test_resting_pairs_exclude_initial_neighbours():
    ob = make_object("sloth", 400, seed=5)
    o = oracle_env(ob, num_substeps=10)
    h = hip_env(ob, num_substeps=10)
    # squash the object so that many particles come within collision_dist of their lattice neighbours
    import torch
    sq = ob["points"].copy(); c = sq.mean(0); sq = c + (sq - c) * 0.5
    o.x[:] = sq
    h.set_state(torch.from_numpy(sq)[None])
    o.update_collision_graph(); h.update_collision_graph()
    num, idx = h.collision_lists()
    assert np.array_equal(num[0].cpu().numpy(), o.coll_num)
    # every close pair of a compact blob was inside the 5*cd query box at rest -> all are resting pairs
    assert o.coll_num.sum() == 0


def test_self_collision_two_blobs():
    ob = two_blobs(seed=1, gap=0.06, speed=3.0)
    n_sub = 200  # 10 ms per "env step": blobs close 3 cm per step
    o, h = _run_pair(ob, 4, n_sub, update_graph=True, collide_self_fric=0.3)
    assert o.total_candidates > 0, "scenario must produce contacts"
    x = h.x[0].cpu().numpy()
    assert close(x, o.x, 5e-5)  # contacts amplify rounding differences (SURVEY.md §7 hard parts)
    # the impulse really acted: blob B lost approach speed
    assert o.v[len(o.v) // 2:, 0].mean() > -2.9


def test_self_collision_with_concurrent_chains_and_odd_substeps(monkeypatch):
    """9 environments (the env step then runs as 2 concurrent kernel chains over env ranges 0-3 / 4-8), an odd substep
    count (the state buffer flips parity every step, exercising both cached graphs) and live self-collision contacts
    (fused kernel + k_self_finish per chain): every environment must reproduce the single-environment oracle."""
    monkeypatch.setenv("R2S_CHAINS", "2")  # small batches default to one chain
    ob = two_blobs(seed=1, gap=0.06, speed=3.0)
    n_sub = 201
    kw = dict(num_substeps=n_sub, collide_self_fric=0.3)
    o = oracle_env(ob, **kw)
    h = hip_env(ob, n_env=9, **kw)
    assert h.layout_stats()["chains"] == 2
    tot = 0
    for _ in range(4):
        o.update_collision_graph(); h.update_collision_graph()
        tot += int(o.coll_num.sum())
        o.step(); h.step()
    assert tot > 0, "scenario must produce contacts"
    x = h.x.cpu().numpy()
    for e in range(9):
        assert close(x[e], o.x, 5e-5), e
    assert close(x, x[0:1], 1e-6)


@pytest.mark.parametrize("n_env,chains", [(1, None), (9, "2")])
def test_ten_substeps_with_live_self_collision_contacts_hold_1e_5(monkeypatch, n_env, chains):
    """"particle positions within 1e-5 abs ... after 10 substeps with contacts" — the horizon the two long runs above
    (800 substeps, held to 5e-5: contacts amplify rounding differences) do not state.  Same scenario; at the first env step whose
    rebuild finds candidates the oracle takes the device state, both rebuild and both run TEN substeps in which impulses act."""
    import torch

    if chains:
        monkeypatch.setenv("R2S_CHAINS", chains)
    ob = two_blobs(seed=1, gap=0.06, speed=3.0)
    kw = dict(num_substeps=200, collide_self_fric=0.3)
    o = oracle_env(ob, **kw)
    h = hip_env(ob, n_env=n_env, **kw)
    done = False
    for _ in range(5):
        h.update_collision_graph()
        x, v = h.sync_state()
        o.x[:] = x[0].cpu().numpy(); o.v[:] = v[0].cpu().numpy()
        o.update_collision_graph()
        if int(o.coll_num.sum()) > 0:
            free = oracle_env(ob, self_collision=False, **kw)          # the same ten substeps without impulses
            free.x[:] = o.x; free.v[:] = o.v
            o.step(10, 0); free.step(10, 0); h.step(10, 0)
            assert float(np.abs(o.v - free.v).max()) > 1e-2, "impulses must act inside the ten substeps"
            xs = h.x.cpu().numpy()
            for e in range(n_env):
                assert close(xs[e], o.x, 1e-5, what=f"10 substeps with self-collision contacts, env {e} of {n_env}"), e
            done = True
            break
        h.step()
    assert done, "scenario must produce contacts"


def test_static_box_mesh_collision():
    from r2s_hip import synth

    ob = make_object("rope", 400, seed=4, lift=0.06)
    ob["v0"] = np.zeros_like(ob["points"]); ob["v0"][:, 2] = -1.0
    c = ob["points"].mean(0)
    box = synth.box_mesh((c[0], c[1], 0.02), (0.05, 0.05, 0.04))
    kw = dict(static_meshes=[box], self_collision=False)
    o, h = _run_pair(ob, 1, 300, **kw)
    x = h.x[0].cpu().numpy()
    assert close(x, o.x, ATOL)
    f = h.collision_forces()[0].cpu().numpy()
    assert np.allclose(f, o.collision_forces, rtol=1e-3, atol=1e-1)
    # particles above the box were stopped ~1 mm above its top face (z = 0.04)
    over = (np.abs(o.x[:, 0] - c[0]) < 0.02) & (np.abs(o.x[:, 1] - c[1]) < 0.02)
    assert over.any() and (o.x[over, 2] > 0.04).all()


def test_gripper_fingers_dynamic_mesh():
    from r2s_hip import synth

    n_sub = 120
    ob = make_object("sloth", 500, seed=6)
    c = ob["points"].mean(0)
    top = ob["points"][:, 2].max()
    fl = synth.finger_mesh((c[0], c[1] - 0.02, top + 0.03))
    fr = synth.finger_mesh((c[0], c[1] + 0.02, top + 0.03))
    interp, centers, dv, om = gripper_motion([fl, fr], n_sub, 5e-5, vel=(0.0, 0.0, -6.0), closing=1.0)
    kw = dict(dynamic_meshes=[fl, fr], self_collision=False)
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, **kw)
    import torch
    for _ in range(1):
        o.set_mesh_interactive(interp, centers, dv, om)
        h.set_mesh_interactive(torch.from_numpy(interp)[None].cuda(), torch.from_numpy(centers)[None].cuda(),
                               torch.from_numpy(dv)[None].cuda(), torch.from_numpy(om)[None].cuda())
        o.step(); h.step()
    x = h.x[0].cpu().numpy()
    assert close(x, o.x, ATOL)
    f = h.collision_forces()[0].cpu().numpy()
    assert np.abs(o.collision_forces).max() > 0, "fingers must touch the object in this scenario"
    # Per-face attribution has genuine ties (a contact point on an edge shared by two triangles is equidistant
    # from both; the first strict minimum wins and 1e-7 differences in x flip it), so gate the per-finger totals
    # tightly and the per-face split loosely.
    mm = h.mesh_map
    for m in (0, 1):
        tot_o, tot_h = o.collision_forces[mm == m].sum(0), f[mm == m].sum(0)
        assert np.allclose(tot_h, tot_o, rtol=1e-3, atol=np.abs(tot_o).max() * 1e-3), (m, tot_o, tot_h)
    assert np.abs(f - o.collision_forces).sum() < 0.3 * np.abs(o.collision_forces).sum()


@pytest.mark.parametrize("defer", ["0", "1"], ids=["queries in place", "finishing kernel"])
def test_small_meshes_with_more_than_128_faces_in_total(defer, monkeypatch):
    """Every mesh small (<= 256 faces) but 176 faces in total: too many for the finishing kernel's triangles-in-registers
    variant (128), so k_contact_finish answers through the face table (the small-mesh branch of the workgroup query) — two
    closing fingers above the toy plus two static finger-sized posts it is pushed against; both flavours against the oracle."""
    import torch
    from r2s_hip import synth

    monkeypatch.setenv("R2S_MESH_DEFER", defer)
    n_sub = 100
    ob = make_object("sloth", 500, seed=8)
    c = ob["points"].mean(0)
    top, x_hi = ob["points"][:, 2].max(), ob["points"][:, 0].max()
    fl = synth.finger_mesh((c[0], c[1] - 0.02, top + 0.03))
    fr = synth.finger_mesh((c[0], c[1] + 0.02, top + 0.03))
    posts = [synth.finger_mesh((x_hi + 0.0105, c[1], top * z), size=(0.02, 0.06, 0.4 * top)) for z in (0.3, 0.7)]   # 0.5 mm behind the +x extremity
    assert sum(len(m[1]) for m in (fl, fr, *posts)) > 128
    interp, centers, dv, om = gripper_motion([fl, fr], n_sub, 5e-5, vel=(0.0, 0.0, -6.0), closing=1.0)
    ob["v0"] = np.zeros_like(ob["points"]); ob["v0"][:, 0] = 1.0                       # drifting into the posts
    kw = dict(dynamic_meshes=[fl, fr], static_meshes=posts, self_collision=False)
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, **kw)
    o.set_mesh_interactive(interp, centers, dv, om)
    t = lambda a: torch.from_numpy(a)[None].cuda()  # noqa: E731
    h.set_mesh_interactive(t(interp), t(centers), t(dv), t(om))
    o.step(); h.step()
    fl_ = h.last_flavour()
    assert fl_["mesh_template"] == 1 and fl_["deferred_mesh_queries"] == (defer == "1")
    nd = len(fl[1]) + len(fr[1])
    assert np.abs(o.collision_forces[:nd]).max() > 0 and np.abs(o.collision_forces[nd:]).max() > 0, "fingers and posts must both be touched"
    assert close(h.x[0], o.x, ATOL, what=f"176 small faces, {'finishing kernel' if defer == '1' else 'in place'}")
    f = h.collision_forces()[0].cpu().numpy()
    for sl in (slice(0, nd), slice(nd, None)):
        tot_o, tot_h = o.collision_forces[sl].sum(0), f[sl].sum(0)
        assert np.allclose(tot_h, tot_o, rtol=2e-3, atol=np.abs(tot_o).max() * 2e-3), (tot_o, tot_h)


def test_two_large_dynamic_meshes_finely_tessellated_fingers():
    """Two dynamic meshes with more than 256 faces each (fingers re-tessellated to 320 triangles): both are large rigid meshes
    with their own per-substep transform — the second one's is not the register-resident one of the query — and they are
    gripper meshes (relative-velocity frame, 5 mm margin, re-query starting at the previous answer's cluster)."""
    import torch
    from r2s_hip import synth

    n_sub = 100
    ob = make_object("sloth", 500, seed=6)
    c = ob["points"].mean(0)
    top = ob["points"][:, 2].max()
    fl = synth.finger_mesh((c[0], c[1] - 0.02, top + 0.03), n_faces=320)
    fr = synth.finger_mesh((c[0], c[1] + 0.02, top + 0.03), n_faces=320)
    assert len(fl[1]) > 256 and len(fr[1]) > 256
    interp, centers, dv, om = gripper_motion([fl, fr], n_sub, 5e-5, vel=(0.0, 0.0, -6.0), closing=1.0)
    kw = dict(dynamic_meshes=[fl, fr], self_collision=False)
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, **kw)
    o.set_mesh_interactive(interp, centers, dv, om)
    t = lambda a: torch.from_numpy(a)[None].cuda()  # noqa: E731
    h.set_mesh_interactive(t(interp), t(centers), t(dv), t(om))
    o.step(); h.step()
    assert h.last_flavour()["mesh_template"] == 2
    mm = h.mesh_map
    assert np.abs(o.collision_forces[mm == 0]).max() > 0 and np.abs(o.collision_forces[mm == 1]).max() > 0, "both fingers must touch"
    assert close(h.x[0], o.x, ATOL, what="two large dynamic (gripper) meshes vs oracle")
    f = h.collision_forces()[0].cpu().numpy()
    for m in (0, 1):
        tot_o, tot_h = o.collision_forces[mm == m].sum(0), f[mm == m].sum(0)
        assert np.allclose(tot_h, tot_o, rtol=2e-3, atol=np.abs(tot_o).max() * 2e-3), (m, tot_o, tot_h)


def test_pusher_mesh_with_more_than_64_super_clusters():
    """A 41k-face rod: more than 64 super-clusters (512 faces each), so the first level of the box hierarchy takes two passes
    of one box per lane; against the oracle's brute force."""
    import torch
    from r2s_hip import synth
    from util_physics import rigid_motion

    n_sub = 10
    ob = make_object("T", 1200, seed=15)
    top = ob["points"][:, 2].max(); x_lo = ob["points"][:, 0].min()
    y_face = float(np.median(ob["points"][ob["points"][:, 0] < x_lo + 0.005, 1]))
    rod = synth.cylinder_mesh((x_lo - 0.0052, y_face, top * 0.5 + 0.02), radius=0.005, length=0.2, n_seg=128, n_rings=160)
    assert len(rod[1]) > 64 * 512
    interp, centers, dv, om = rigid_motion(rod, n_sub, 5e-5, vel=(2.0, 0.0, 0.0), omega=(0.0, 0.0, 2.0))
    kw = dict(dynamic_meshes=[rod], self_collision=False, use_pusher=True, collide_eef_fric=0.2)
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, **kw)
    o.set_mesh_interactive(interp, centers, dv, om)
    t = lambda a: torch.from_numpy(a)[None].cuda()  # noqa: E731
    h.set_mesh_interactive(t(interp), t(centers), t(dv), t(om))
    o.step(); h.step()
    assert np.abs(o.collision_forces).max() > 0, "the rod must touch the block in this scenario"
    assert close(h.x[0], o.x, 1e-5, what="41k-face rod (two passes over the super-cluster boxes) vs oracle")
    tot_o, tot_h = o.collision_forces.sum(0), h.collision_forces()[0].cpu().numpy().sum(0)
    assert np.allclose(tot_h, tot_o, rtol=2e-3, atol=np.abs(tot_o).max() * 2e-3), (tot_o, tot_h)


def test_collision_forces_are_cleared_on_every_replay_of_the_step():
    """collision_forces holds the LAST substep's forces of the LAST step (the reference zeroes the accumulator in every
    substep): a second step without contact must read all zeros.  Regression for a captured memset that only cleared on
    the first replay."""
    import torch
    from r2s_hip import synth

    n_sub = 60
    ob = make_object("sloth", 500, seed=6)
    c = ob["points"].mean(0)
    top = ob["points"][:, 2].max()
    fl = synth.finger_mesh((c[0], c[1] - 0.02, top + 0.03))
    fr = synth.finger_mesh((c[0], c[1] + 0.02, top + 0.03))
    kw = dict(dynamic_meshes=[fl, fr], self_collision=False)
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, **kw)
    t = lambda a: torch.from_numpy(a)[None].cuda()  # noqa: E731
    seen = []
    for vel in ((0.0, 0.0, -12.0), (0.0, 0.0, 0.0), (0.0, 0.0, 40.0), (0.0, 0.0, 0.0)):
        interp, centers, dv, om = gripper_motion([fl, fr], n_sub, 5e-5, vel=vel, closing=0.0)
        fl = (interp[-1][: len(fl[0])], fl[1]); fr = (interp[-1][len(fl[0]):], fr[1])
        o.set_mesh_interactive(interp, centers, dv, om)
        h.set_mesh_interactive(t(interp), t(centers), t(dv), t(om))
        o.step(); h.step()
        f = h.collision_forces()[0].cpu().numpy()
        seen.append(float(np.abs(o.collision_forces).max()))
        if seen[-1] == 0.0:
            assert np.abs(f).max() == 0.0
        else:
            assert np.allclose(f.sum(0), o.collision_forces.sum(0), rtol=1e-3, atol=np.abs(o.collision_forces).max() * 1e-3)
    assert max(seen[:2]) > 0 and seen[-1] == 0.0, seen


def test_batched_envs_are_independent_and_match_single():
    ob = make_object("rope", 500, seed=7, lift=0.1)
    import torch
    h1 = hip_env(ob, num_substeps=50, self_collision=False)
    h4 = hip_env(ob, num_substeps=50, self_collision=False, n_env=4)
    x4 = np.repeat(ob["points"][None], 4, 0).copy()
    for e in range(4):
        x4[e, :, 0] += 0.01 * e
    h4.set_state(torch.from_numpy(x4))
    h1.step(); h4.step()
    a = h1.x[0].cpu().numpy(); b = h4.x.cpu().numpy()
    for e in range(4):
        shifted = b[e].copy(); shifted[:, 0] -= 0.01 * e
        assert close(shifted, a, 2e-6)


def test_drop_in_surface_smoke():
    import torch
    from sim.physics import SpringMassSystemWarp

    ob = make_object("rope", 300, seed=8, lift=0.05)
    c = cfg(num_substeps=30)
    dev = "cuda:0"
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    sim = SpringMassSystemWarp(c, dev, t(ob["points"]), t(ob["springs"]), t(ob["rest"]), torch.ones(len(ob["points"]), device=dev),
                               num_object_points=len(ob["points"]), init_spring_Y=t(ob["log_Y"]),
                               collide_elas=torch.tensor([0.5]), collide_fric=torch.tensor([0.3]),
                               collide_eef_elas=torch.tensor([0.0]), collide_eef_fric=torch.tensor([1.0]),
                               collide_self_elas=torch.tensor([0.5]), collide_self_fric=torch.tensor([0.3]))
    x_before = sim.wp_state.wp_x.clone()
    sim.update_collision_graph()
    sim.graph.launch()
    o = oracle_env(ob, num_substeps=30)
    o.update_collision_graph(); o.step()
    assert sim.wp_state.wp_x.shape == (len(ob["points"]), 3)
    assert close(sim.wp_state.wp_x.cpu().numpy(), o.x, ATOL)
    assert (sim.wp_state.wp_x - x_before).abs().max() > 0


@pytest.mark.parametrize("defer", ["0", "1"], ids=["flavour request: in place (a large scene defers anyway)", "queries deferred to k_contact_finish"])
def test_pusher_25k_face_mesh_cluster_query_vs_oracle(defer, monkeypatch):
    """configs[3] ingredient: a ~25k-face closed pusher mesh (cluster hierarchy + rigid transform + pseudonormal sign on
    the GPU) against the oracle's brute-force closest point + exact winding number.  A scene with a large mesh always
    hands its queries to k_contact_finish (one workgroup per touching particle), whatever flavour is asked for."""
    import torch

    monkeypatch.setenv("R2S_MESH_DEFER", defer)
    from r2s_hip import synth
    from util_physics import rigid_motion

    n_sub = 16
    ob = make_object("T", 2229, seed=9)
    c = ob["points"].mean(0); top = ob["points"][:, 2].max()
    x_lo = ob["points"][:, 0].min()
    # vertical pusher rod just outside the block's -x face, sweeping into it at 1.5 m/s while yawing slowly
    rod = synth.cylinder_mesh((x_lo - 0.0052, c[1] + 0.03, top * 0.5 + 0.02), radius=0.005, length=0.2)
    assert len(rod[1]) > 20000
    interp, centers, dv, om = rigid_motion(rod, n_sub, 5e-5, vel=(2.0, 0.0, 0.0), omega=(0.0, 0.0, 3.0))
    kw = dict(dynamic_meshes=[rod], self_collision=False, use_pusher=True, collide_eef_fric=0.2)
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, **kw)
    o.set_mesh_interactive(interp, centers, dv, om)
    h.set_mesh_interactive(torch.from_numpy(interp)[None].cuda(), torch.from_numpy(centers)[None].cuda(),
                           torch.from_numpy(dv)[None].cuda(), torch.from_numpy(om)[None].cuda())
    o.step(); h.step()
    x = h.x[0].cpu().numpy()
    moved = np.abs(o.x - ob["points"]).max()
    assert np.abs(o.collision_forces).max() > 0, "the rod must touch the block in this scenario"
    assert close(x, o.x, 1e-5), (np.abs(x - o.x).max(), moved)
    f = h.collision_forces()[0].cpu().numpy()
    tot_o, tot_h = o.collision_forces.sum(0), f.sum(0)
    assert np.allclose(tot_h, tot_o, rtol=2e-3, atol=np.abs(tot_o).max() * 2e-3), (tot_o, tot_h)


@pytest.mark.parametrize("kind", ["unwelded", "open"])
def test_large_meshes_that_are_not_welded_closed_manifolds_are_accepted_like_the_reference(kind):
    """wp.mesh_query_point_sign_winding_number (spring_mass_warp.py:322-324) takes any triangle soup.  A > 256-face
    mesh whose vertices are repeated per triangle (an STL export) must be welded and treated like the closed manifold it is;
    an OPEN one (caps missing) cannot use pseudonormals and falls back to the exact winding number instead of being
    rejected.  Both against the oracle's brute-force query."""
    import torch
    from r2s_hip import synth
    from util_physics import rigid_motion

    n_sub = 12
    ob = make_object("T", 700, seed=12)
    top = ob["points"][:, 2].max(); x_lo = ob["points"][:, 0].min()
    y_face = float(np.median(ob["points"][ob["points"][:, 0] < x_lo + 0.005, 1]))       # the T's bar reaches furthest in -x
    v, f = synth.cylinder_mesh((x_lo - 0.0052, y_face, top * 0.5 + 0.01), radius=0.005, length=0.08, n_seg=16, n_rings=12)
    assert len(f) > 256
    if kind == "unwelded":
        v, f = v[f.reshape(-1)].copy(), np.arange(3 * len(f), dtype=np.int32).reshape(-1, 3)
    else:
        f = f[: 2 * 16 * 12].copy()              # side wall only
        assert len(f) > 256
    rod = (v, f)
    interp, centers, dv, om = rigid_motion(rod, n_sub, 5e-5, vel=(2.0, 0.0, 0.0), omega=(0.0, 0.0, 2.0))
    kw = dict(dynamic_meshes=[rod], self_collision=False, use_pusher=True, collide_eef_fric=0.2)
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, **kw)
    o.set_mesh_interactive(interp, centers, dv, om)
    t = lambda a: torch.from_numpy(a)[None].cuda()  # noqa: E731
    h.set_mesh_interactive(t(interp), t(centers), t(dv), t(om))
    o.step(); h.step()
    assert h.last_flavour()["mesh_template"] == 2
    assert np.abs(o.collision_forces).max() > 0, "the rod must touch the block in this scenario"
    assert close(h.x[0], o.x, 1e-5, what=f"{kind} large mesh vs oracle")


def test_large_pusher_mesh_next_to_a_small_static_obstacle_vs_oracle():
    """A scene that mixes both kinds of collision mesh: the ~12k-face pusher rod (box hierarchy in its rest frame, pseudonormal
    sign) pressing the block against a small static box (12 faces: visited through the face table in the world frame, its
    winding number summed exactly).  k_contact_finish answers both in one query per particle; the oracle brute-forces
    every face.  Also checks the per-substep deferred-query counts the finishing kernel reports."""
    import torch
    from r2s_hip import synth
    from util_physics import rigid_motion

    n_sub = 16
    ob = make_object("T", 1500, seed=21)
    top = ob["points"][:, 2].max(); x_lo, x_hi = ob["points"][:, 0].min(), ob["points"][:, 0].max()
    y_face = float(np.median(ob["points"][ob["points"][:, 0] < x_lo + 0.005, 1]))
    rod = synth.cylinder_mesh((x_lo - 0.0052, y_face, top * 0.5 + 0.02), radius=0.005, length=0.2, n_seg=96, n_rings=60)
    assert len(rod[1]) > 10000
    y_back = float(np.median(ob["points"][ob["points"][:, 0] > x_hi - 0.005, 1]))
    wall = synth.box_mesh((x_hi + 0.0105, y_back, top * 0.5), (0.02, 0.08, top))      # 0.5 mm behind the block's +x face
    interp, centers, dv, om = rigid_motion(rod, n_sub, 5e-5, vel=(2.0, 0.0, 0.0), omega=(0.0, 0.0, 1.0))
    kw = dict(dynamic_meshes=[rod], static_meshes=[wall], self_collision=False, use_pusher=True, collide_eef_fric=0.2)
    ob["v0"] = np.zeros_like(ob["points"]); ob["v0"][:, 0] = 1.0                      # the block drifts into the wall
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, **kw)
    o.set_mesh_interactive(interp, centers, dv, om)
    t = lambda a: torch.from_numpy(a)[None].cuda()  # noqa: E731
    h.set_mesh_interactive(t(interp), t(centers), t(dv), t(om))
    o.step(); h.step()
    fl = h.last_flavour()
    assert fl["mesh_template"] == 2 and fl["deferred_mesh_queries"]
    fo = o.collision_forces
    n_rod = len(rod[1])
    assert np.abs(fo[:n_rod]).max() > 0 and np.abs(fo[n_rod:]).max() > 0, "both the rod and the wall must be touched"
    assert close(h.x[0], o.x, 1e-5, what="large rod + small wall vs oracle")
    f = h.collision_forces()[0].cpu().numpy()
    for name, sl in (("rod", slice(0, n_rod)), ("wall", slice(n_rod, None))):
        tot_o, tot_h = fo[sl].sum(0), f[sl].sum(0)
        assert np.allclose(tot_h, tot_o, rtol=2e-3, atol=np.abs(tot_o).max() * 2e-3), (name, tot_o, tot_h)
    dc = h.deferred_counts()
    assert dc.shape == (n_sub + 1,) and dc[-1] != 0 and dc[:-1].max() > 0 and dc[:-1].max() <= h.N


def test_more_listed_particles_than_finishing_workgroups_small_scene(monkeypatch):
    """k_contact_finish walks its list with a grid-stride loop of whole workgroups (barriers inside): a substep that lists more
    particles than the launch has workgroups (1024 for small scenes) must still finish every one of them, exactly once.  A
    rope lying in an OPEN box (bottom removed: not a closed manifold, so the query's own 2 cm range is the early-out bound
    and most of the rope is listed every substep), 8 identical environments, deferral forced; every environment against the
    oracle."""
    from r2s_hip import synth

    monkeypatch.setenv("R2S_MESH_DEFER", "1")
    n_sub, n_env = 10, 8
    ob = make_object("rope", 800, seed=5, lift=0.0395)                                # lowest particles 0.5 mm above the box top
    ob["v0"] = np.zeros_like(ob["points"]); ob["v0"][:, 2] = -0.5
    c = ob["points"].mean(0)
    v, f = synth.box_mesh((c[0], c[1], 0.02), (0.5, 0.08, 0.04))
    f = f[[k for k in range(len(f)) if not (v[f[k], 2] < 0.001).all()]]          # drop the two bottom triangles
    assert len(f) == 10
    kw = dict(static_meshes=[(v, f)], self_collision=False)
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, n_env=n_env, **kw)
    o.step(); h.step()
    dc = h.deferred_counts()
    assert dc[:-1].max() > 1024, dc[:-1].max()
    assert np.abs(o.collision_forces).max() > 0
    x = h.x.cpu().numpy()
    for e in range(n_env):
        assert close(x[e], o.x, ATOL, what=f"env {e}: {int(dc[:-1].max())} listed particles per substep")
    fh = h.collision_forces().cpu().numpy()
    assert np.allclose(fh[0], o.collision_forces, rtol=1e-3, atol=1e-1) and np.allclose(fh[-1], fh[0], rtol=1e-4, atol=1e-2)


def test_more_listed_particles_than_finishing_workgroups_large_mesh():
    """The same for the large-mesh finishing kernel (512 workgroups of four wavefronts): two T-blocks resting on a finely
    tessellated static slab (2 688 faces, closed: box hierarchy + pseudonormals), every bottom particle inside the 1 mm margin."""
    from r2s_hip import synth

    n_sub, n_env = 8, 6
    ob = make_object("T", 2229, seed=4)
    ob["v0"] = np.zeros_like(ob["points"]); ob["v0"][:, 2] = -0.3
    c = ob["points"].mean(0); z0 = ob["points"][:, 2].min()
    slab = synth.cylinder_mesh((c[0], c[1], z0 - 0.0105), radius=0.3, length=0.02, n_seg=64, n_rings=20)   # top at z0 - 0.5 mm
    assert len(slab[1]) > 2000
    kw = dict(static_meshes=[slab], self_collision=False)
    o = oracle_env(ob, num_substeps=n_sub, **kw)
    h = hip_env(ob, num_substeps=n_sub, n_env=n_env, **kw)
    o.step(); h.step()
    assert h.last_flavour()["mesh_template"] == 2
    dc = h.deferred_counts()
    assert dc[:-1].max() > 512, dc[:-1].max()
    assert np.abs(o.collision_forces).max() > 0
    x = h.x.cpu().numpy()
    for e in range(n_env):
        assert close(x[e], o.x, ATOL, what=f"large slab, env {e}: {int(dc[:-1].max())} listed particles per substep")


def test_a_large_dynamic_mesh_that_deforms_is_reported_by_a_later_step():
    """The rigidity check of large dynamic meshes is read without blocking the stream (pinned word + event): the violation
    surfaces at the first step() after the check has landed, as R2S_ERR_INVALID."""
    import torch
    from r2s_hip import synth
    from r2s_hip._lib import R2SError
    from util_physics import rigid_motion

    n_sub = 4
    ob = make_object("T", 300, seed=13)
    rod = synth.cylinder_mesh((0.0, 0.0, 0.5), radius=0.005, length=0.08, n_seg=16, n_rings=12)
    h = hip_env(ob, num_substeps=n_sub, dynamic_meshes=[rod], self_collision=False, use_pusher=True)
    interp, centers, dv, om = rigid_motion(rod, n_sub, 5e-5)
    interp = interp.copy(); interp[:, ::2, 0] *= 1.05            # stretch every other vertex: not a rigid motion
    t = lambda a: torch.from_numpy(a)[None].cuda()  # noqa: E731
    h.set_mesh_interactive(t(interp), t(centers), t(dv), t(om))
    torch.cuda.synchronize()
    with pytest.raises(R2SError, match="does not move rigidly"):
        for _ in range(3):
            h.step()
            torch.cuda.synchronize()


def test_hip_stepper_matches_fixtures_from_the_reference_kernel_bodies():
    """tests/golden/physics_kernels.npz was produced by executing the reference's own kernel source (make_physics_golden.py):
    the HIP stepper reproduces the springs / gate / ground trajectory (A) and the finger + static-box contact trajectory with
    per-face forces (C)."""
    import torch

    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "physics_kernels.npz"))
    obA = dict(points=G["A_x0"], springs=G["A_springs"], rest=G["A_rest"], log_Y=G["A_logY"], v0=G["A_v0"])
    n = len(G["A_x_traj"])
    h = hip_env(obA, num_substeps=n, self_collision=False, spring_Y_min=float(G["A_Ymin"]))
    for k in range(n):
        h.step(1, k)
        assert close(h.x[0].cpu().numpy(), G["A_x_traj"][k], 2e-6), k
        assert close(h.v[0].cpu().numpy(), G["A_v_traj"][k], 2e-4), k
    n_dyn = int(G["C_n_dyn"])
    verts, faces, mm = G["C_verts"], G["C_faces"], G["C_mesh_map"]
    nl, nr = int((mm == 0).sum()), int((mm == 1).sum())
    vl = int(faces[:nl].max()) + 1
    dyn = [(verts[:vl], faces[:nl]), (verts[vl:n_dyn], faces[nl:nl + nr] - vl)]
    sta = [(verts[n_dyn:], faces[nl + nr:] - n_dyn)]
    obC = dict(points=G["C_x0"], springs=G["C_springs"], rest=G["C_rest"], log_Y=G["C_logY"], v0=G["C_v0"])
    n = len(G["C_x_traj"])
    h = hip_env(obC, num_substeps=n, self_collision=False, dynamic_meshes=dyn, static_meshes=sta, collide_eef_elas=0.5, collide_eef_fric=1.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].cuda()  # noqa: E731
    h.set_mesh_interactive(t(G["C_interp"]), t(G["C_centers"]), t(G["C_dyn_vel"]), t(G["C_dyn_omega"]))
    for k in range(n):
        h.step(1, k)
        assert close(h.x[0].cpu().numpy(), G["C_x_traj"][k], 2e-6), k
        assert close(h.v[0].cpu().numpy(), G["C_v_traj"][k], 2e-3), k
    # forces of the last substep: per-finger / per-mesh totals (the per-face split has genuine ties, see above)
    f, ref = h.collision_forces()[0].cpu().numpy(), G["C_forces_traj"][-1]
    for m in (0, 1, -1):
        assert np.allclose(f[mm == m].sum(0), ref[mm == m].sum(0), rtol=2e-3, atol=1e-3 * max(1.0, np.abs(ref).max())), m
