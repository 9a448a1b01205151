"""Round-3 parity: the graph flavours the headline's CONTACT phase runs — self-collision candidates AND mesh contact in one
substep (spring_mass_warp.py:230-268 object_collision, then :295-421 mesh_collision on the impulse-corrected velocity,
step order :823-943) — against the oracle:

  * small scene, queries deferred:  k_substep<..,true,1> + k_contact_finish<3,true>  (tagged list entries: impulses with 64
    lanes, then the query; the rest of the candidate list in 16-lane groups with NEED = 2)
  * small scene, queries in place:  k_substep<..,true,1> + k_self_finish<1>
  * a large (> 256 faces) pusher:   k_substep<..,true,2> + k_contact_finish<2,true>
  * the bench's own scene (sloth_arms, grasp trace) for 20 substeps in the grasp, driven through EefOracle
  * configs[3] at its per-GPU size (32 environments, 25k-face rod): determinism and batch independence

Round 2's mesh tests all ran with self_collision=False, and its self-collision tests without meshes."""
import numpy as np
import pytest

from util_parity import close, record
from util_physics import far_apart, gripper_motion, hip_env, make_object, oracle_env, rigid_motion, two_sheets

pytestmark = pytest.mark.gpu
ATOL = 1e-5  # BASELINE.json: particle positions within 1e-5 abs


def _t(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _sheets_between_fingers(n_sub, closing=1.0):
    from r2s_hip import synth

    ob, nA = two_sheets()
    c = ob["points"].mean(0)
    # pads 0.2 mm outside the 5 mm gripper margin of the sheets (2 mm either side of the mid-plane), closing at 1 m/s each
    fl = synth.finger_mesh((c[0], c[1] - 0.002 - 0.0052 - 0.005, c[2]))
    fr = synth.finger_mesh((c[0], c[1] + 0.002 + 0.0052 + 0.005, c[2]))
    motion = gripper_motion([fl, fr], n_sub, 5e-5, vel=(0.0, 0.0, 0.0), closing=closing)
    return ob, nA, [fl, fr], motion


@pytest.mark.parametrize("n_env", [1, 5, 9], ids=["1 env", "5 envs, one chain", "9 envs, two chains"])
@pytest.mark.parametrize("defer", ["0", "1"], ids=["queries in place (k_self_finish)", "finishing kernel (k_contact_finish<3,true>)"])
def test_two_sheets_squeezed_between_closing_fingers(defer, n_env, monkeypatch):
    """Every particle under the pads has a live self-collision candidate (the sheet opposite, 4 mm away) and is inside the 5 mm
    finger margin: the fused kernel publishes its velocity, tags it onto the mesh list, and the finishing launch applies the
    averaged impulses (:264-266) before the finger response (:343-410, relative-velocity frame, re-query)."""
    import torch

    monkeypatch.setenv("R2S_MESH_DEFER", defer)
    monkeypatch.setenv("R2S_RES_SELF_SRV", "2")       # servers from the first step on (default: once a query was needed, two steps later)
    if n_env == 9:
        monkeypatch.setenv("R2S_CHAINS", "2")
    n_sub = 36   # the float64 shadow of the oracle agrees with its float32 run to 3e-8 over 40 substeps of this scene; past ~42 a
    #              contact decision flips between the two and they part by 0.1 mm (scratch run recorded in DESIGN.md §2)
    ob, nA, fingers, (interp, centers, dv, om) = _sheets_between_fingers(n_sub)
    far = far_apart(ob, nA)
    kw = dict(num_substeps=n_sub, dynamic_meshes=fingers, self_collision=True)
    o = oracle_env(far, **kw)
    h = hip_env(far, n_env=n_env, **kw)
    if n_env == 9:
        assert h.layout_stats()["chains"] == 2
    o.x[:] = ob["points"]
    h.set_state(torch.from_numpy(ob["points"])[None].repeat(n_env, 1, 1))
    assert o.update_collision_graph() > 0
    h.update_collision_graph()
    with_cand = int((o.coll_num > 0).sum())
    o.set_mesh_interactive(interp, centers, dv, om)
    rep = lambda a: _t(a)[None].repeat(n_env, *([1] * a.ndim))  # noqa: E731
    h.set_mesh_interactive(rep(interp), rep(centers), rep(dv), rep(om))
    o.step(); h.step()
    fl = h.last_flavour()
    assert fl["self_collision_kernel"] and fl["mesh_template"] == 1 and fl["deferred_mesh_queries"] == (defer == "1"), fl
    if defer == "0" and n_env < 9:
        # round 5: a small batch stays ONE resident launch through this — the candidates' hand-off AND query servers that answer (a request
        # per particle and substep, carrying the velocity after the impulses); a forced second chain (the 9-environment case) runs without
        # servers: their claim lines are the handle's
        assert fl["resident"] and fl.get("query_server_workgroups", 0) > 0 and not fl.get("servers_own_their_particle", True), fl
    if defer == "0" and n_env == 9:
        assert fl["resident"] and fl.get("query_server_workgroups", 0) == 0, fl
    tagged = h.tagged_count()
    if defer == "1":
        assert tagged >= 20 * n_env, tagged      # ~50 particles per environment sit under the pads with a candidate
        assert h.deferred_counts()[:-1].max() > 0
    assert with_cand > 300 and np.abs(o.collision_forces).max() > 0, "candidates and finger contact must both be live"
    # the impulses really acted: the same run without candidate lists ends somewhere else
    o2 = oracle_env(far, **kw)
    o2.x[:] = ob["points"]
    o2.set_mesh_interactive(interp, centers, dv, om)
    o2.step()
    assert np.abs(o2.x - o.x).max() > 5e-5
    x = h.x.cpu().numpy()
    for e in range(n_env):
        assert close(x[e], o.x, ATOL, what=f"two sheets between fingers, defer={defer}, env {e} of {n_env}"), (e, np.abs(x[e] - o.x).max())
    assert close(h.v[0], o.v, 5e-3, what=f"v, two sheets between fingers, defer={defer}")
    f = h.collision_forces().cpu().numpy()
    mm = h.mesh_map
    for m in (0, 1):
        tot_o, tot_h = o.collision_forces[mm == m].sum(0), f[0][mm == m].sum(0)
        assert np.allclose(tot_h, tot_o, rtol=2e-3, atol=np.abs(tot_o).max() * 2e-3), (m, tot_o, tot_h)
    record("two sheets between fingers", tagged_entries=tagged, particles_with_candidates=with_cand, n_env=n_env, tol=0)


@pytest.mark.parametrize("n_env", [1, 5])
def test_sheet_pushed_into_sheet_by_a_large_pusher_mesh_with_self_collision(n_env):
    """k_substep<..,true,2> + k_contact_finish<2,true>: a 416-face rod (> 256: box hierarchy, rigid transform, pseudonormal
    sign, 1 mm pusher margin) pushes one sheet into the other at 2 m/s while yawing; the pushed particles carry candidates."""
    import torch
    from r2s_hip import synth

    n_sub = 40
    ob, nA = two_sheets()
    far = far_apart(ob, nA)
    c = ob["points"].mean(0)
    rod = synth.cylinder_mesh((c[0], c[1] - 0.002 - 0.005 - 0.0011, c[2]), radius=0.005, length=0.08, n_seg=16, n_rings=12)
    assert len(rod[1]) > 256
    interp, centers, dv, om = rigid_motion(rod, n_sub, 5e-5, vel=(0.0, 2.0, 0.0), omega=(0.0, 0.0, 2.0))
    kw = dict(num_substeps=n_sub, dynamic_meshes=[rod], use_pusher=True, collide_eef_fric=0.2, self_collision=True)
    o = oracle_env(far, **kw)
    h = hip_env(far, n_env=n_env, **kw)
    o.x[:] = ob["points"]
    h.set_state(torch.from_numpy(ob["points"])[None].repeat(n_env, 1, 1))
    assert o.update_collision_graph() > 0
    h.update_collision_graph()
    o.set_mesh_interactive(interp, centers, dv, om)
    rep = lambda a: _t(a)[None].repeat(n_env, *([1] * a.ndim))  # noqa: E731
    h.set_mesh_interactive(rep(interp), rep(centers), rep(dv), rep(om))
    o.step(); h.step()
    fl = h.last_flavour()
    assert fl["self_collision_kernel"] and fl["mesh_template"] == 2 and fl["deferred_mesh_queries"], fl
    tagged = h.tagged_count()
    assert tagged > 0 and np.abs(o.collision_forces).max() > 0
    o2 = oracle_env(far, **kw)
    o2.x[:] = ob["points"]
    o2.set_mesh_interactive(interp, centers, dv, om)
    o2.step()
    assert np.abs(o2.x - o.x).max() > 1e-3, "the impulses must matter in this scenario"
    x = h.x.cpu().numpy()
    for e in range(n_env):
        assert close(x[e], o.x, ATOL, what=f"sheet pushed into sheet by a 416-face rod, env {e} of {n_env}"), (e, np.abs(x[e] - o.x).max())
    tot_o, tot_h = o.collision_forces.sum(0), h.collision_forces()[0].cpu().numpy().sum(0)
    assert np.allclose(tot_h, tot_o, rtol=2e-3, atol=np.abs(tot_o).max() * 2e-3), (tot_o, tot_h)
    record("sheet pushed into sheet by a large rod", tagged_entries=tagged, n_env=n_env, tol=0)


def test_particles_at_the_centre_of_a_35k_face_sphere_many_super_clusters_in_reach():
    """with more than eight super-clusters in reach — or more than 64 in total — the candidate clusters of
    mesh_query_block were numbered with a bound that differs between the wavefronts, so a cluster could be visited by none.
    Worst case for the hierarchy: particles ~1 mm from the centre of a 34 880-face sphere of radius 15 mm (68 super-clusters,
    all within a fraction of a millimetre of the same distance); the reference's query (brute force in the oracle) finds the
    closest face among them, the response projects the particle out through it."""
    from r2s_hip import synth

    rng = np.random.default_rng(3)
    ctr = np.array([0.0, 0.0, 0.05])
    d = rng.normal(size=(48, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = (ctr + d * rng.uniform(0.0006, 0.0025, (48, 1))).astype(np.float32)
    springs, rest = synth.build_springs(pts)
    ob = dict(points=pts, springs=springs, rest=rest, log_Y=np.log(np.full(len(springs), 1e3, np.float32)), v0=np.zeros_like(pts))
    sph = synth.sphere_mesh(ctr)
    assert len(sph[1]) > 64 * 512
    kw = dict(num_substeps=2, static_meshes=[sph], self_collision=False)
    o = oracle_env(ob, **kw)
    h = hip_env(ob, **kw)
    o.step(); h.step()
    assert h.last_flavour()["mesh_template"] == 2
    r = np.linalg.norm(o.x - ctr, axis=1)
    assert r.min() > 0.0155, "every particle must have been projected out of the sphere"
    # the projection runs from ~1 mm off the centre through the closest point 14 mm away: a last-bit difference in the closest
    # point's direction is amplified ~16x (the oracle's own float32 / float64 runs differ by 3.6e-6 here), hence 2e-5
    assert close(h.x[0], o.x, 2e-5, what="particles near the centre of a 34 880-face sphere vs oracle")
    # the closest FACE itself, exactly, for the static query the first substep makes: per-face forces name it
    fo, fh = o.collision_forces, h.collision_forces()[0].cpu().numpy()
    assert (np.abs(fo).sum(1) > 0).sum() > 0
    hit_o, hit_h = set(np.flatnonzero(np.abs(fo).sum(1) > 0)), set(np.flatnonzero(np.abs(fh).sum(1) > 0))
    record("sphere: faces that received a force", oracle=len(hit_o), hip=len(hit_h), differing=len(hit_o ^ hit_h), tol=0)
    assert len(hit_o ^ hit_h) <= max(2, len(hit_o) // 10), (sorted(hit_o ^ hit_h))


def test_asymmetric_collision_lists_are_refused():
    """object_collision reads the published velocity of every listed partner; a partner without a list of its
    own never publishes, so r2s_phys_set_collision_lists refuses such lists instead of reading stale memory."""
    from r2s_hip._lib import R2SError

    ob = make_object("rope", 300, seed=3, lift=0.05)
    h = hip_env(ob, num_substeps=4, self_collision=True)
    num = np.zeros((1, h.N), np.int32)
    idx = np.zeros((1, h.N, 4), np.int32)
    num[0, 5] = 1; idx[0, 5, 0] = 9          # 5 lists 9, 9 lists nobody
    with pytest.raises(R2SError):
        h.set_collision_lists(num, idx)
    num[0, 9] = 1; idx[0, 9, 0] = 5
    h.set_collision_lists(num, idx)
    h.step()
    assert bool(np.isfinite(h.x.cpu().numpy()).all())


def test_an_impulse_beyond_the_reach_bound_is_reported_by_a_later_step_and_a_new_state_clears_it(monkeypatch):
    """ADVICE r2 (low), the checked bound: a particle with candidates whose pre-impulse test (widened by 2 mm = 40 m/s of velocity
    change in one substep) found no mesh in reach is finished without a query (k_contact_finish part 2).  One sheet thrown at the
    other at 200 m/s gets impulses beyond that bound: the kernel raises the sticky fault word, the first step() after that step
    has finished returns R2S_ERR_INVALID with the reason, every later one too — until set_state hands in a new state."""
    import torch
    from r2s_hip import synth
    from r2s_hip._lib import R2SError

    monkeypatch.setenv("R2S_MESH_DEFER", "1")       # the finishing kernel in the graph although the fingers are 5 cm away
    n_sub = 4
    ob, nA = two_sheets()
    far = far_apart(ob, nA)
    c = ob["points"].mean(0)
    fingers = [synth.finger_mesh((c[0], c[1] - 0.05, c[2])), synth.finger_mesh((c[0], c[1] + 0.05, c[2]))]
    interp, centers, dv, om = gripper_motion(fingers, n_sub, 5e-5, vel=(0.0, 0.0, 0.0), closing=0.0)
    h = hip_env(far, n_env=1, num_substeps=n_sub, dynamic_meshes=fingers, self_collision=True)
    x0 = torch.from_numpy(ob["points"])[None]
    v0 = torch.zeros_like(x0)
    toward = np.sign(ob["points"][nA:, 1].mean() - ob["points"][:nA, 1].mean())
    v0[0, :nA, 1] = 200.0 * float(toward)
    t = lambda a: _t(a)[None]  # noqa: E731

    def arm(v):
        h.set_state(x0, v)
        h.update_collision_graph()
        h.set_mesh_interactive(t(interp), t(centers), t(dv), t(om))

    arm(v0)
    h.step()                                          # raises the word on the device; this call still returns OK
    fl = h.last_flavour()
    assert fl["self_collision_kernel"] and fl["deferred_mesh_queries"], fl
    torch.cuda.synchronize()
    for _ in range(2):                                # sticky
        with pytest.raises(R2SError, match="40 m/s"):
            h.step()
    arm(torch.zeros_like(x0))                         # a new state: usable again
    for _ in range(3):
        h.step()
        torch.cuda.synchronize()
    assert bool(torch.isfinite(h.x).all())


@pytest.mark.parametrize("servers", ["1", "0"], ids=["resident launch: candidates' hand-off + query servers", "per-substep kernels + k_contact_finish"])
def test_bench_scene_one_env_20_substeps_in_the_grasp_vs_oracle_driven_through_eef_oracle(servers, monkeypatch):
    """ONE environment of bench.py's own workload (sloth_arms, 15 066 particles, grasp trace) through the
    product path; in the first env step after the fingers closed — arms pressed together, finger contact, the flavour the
    headline's contact phase times — 20 substeps against PhysOracle driven by EefOracle, and the side-camera frame against the
    raster oracle.  The same routine is bench.py's --parity-gate.  Round 5: one environment stays in the resident launch through this —
    the self-collision flavour with query servers that answer (a request per particle and substep, carrying the velocity after the
    impulses); `R2S_RES_SELF_SRV=0` is the form of rounds 3-4, two launches per substep."""
    from oracle import parity_gate

    monkeypatch.setenv("R2S_RES_SELF_SRV", servers)
    monkeypatch.setenv("R2S_RES_SRV_LOW", "100")     # (round 6: a launch that claimed 30 % of its units sends the next step to the per-substep kernels; this test pins the resident path)
    r = parity_gate.run("sloth_32env", num_substeps=667, n_compare=20, close_at=2)
    record(f"bench scene (sloth_arms, grasp), 20 substeps in contact, R2S_RES_SELF_SRV={servers}", **{k: v for k, v in r.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}, tol=1e-5)
    assert r["mesh_contact"] and r["particles_with_candidates"] > 0, r
    if servers == "1":
        assert r["flavour"].startswith("k_steps_resident<512,true,1> + ") and "a request per substep" in r["flavour"], r["flavour"]
    else:
        assert "true,1>" in r["flavour"] and "k_contact_finish" in r["flavour"], r["flavour"]   # (the small-batch layout, two launches per substep)
        assert r["tagged_entries"] >= 0 and r["deferred_per_substep_max"] > 0, r
    assert r["eef_pts_max_abs"] < 2e-6 and r["eef_center_max_abs"] < 5e-7, r
    assert r["x_max_abs"] < 1e-5, r
    assert r["hard_rgb_mismatch_pixels"] == 0 and r["hard_depth_mismatch_pixels"] == 0, r
    assert r["threshold_flip_pixels"] <= 1e-4 * r["pixels"] and r["median_depth_crossing_pixels"] <= 1e-4 * r["pixels"], r
    assert r["passed"], r


def test_bench_gate_on_the_batched_flavour_two_chains_first_and_last_environment():
    """the gate bench.py runs before a multi-environment window — a 9-environment batch of the bench's own scene
    (large-batch layout, two concurrent kernel chains), environments 0 and 8 (one per chain) each against an oracle of its own, in
    the contact flavour (candidates + finger contact), + the side and wrist frames of environment 8."""
    from oracle import parity_gate

    r = parity_gate.run("sloth_32env", num_substeps=667, n_compare=20, close_at=2, n_env=9)
    record("bench scene, 9-env batch (2 chains), envs 0 and 8, 20 substeps in contact", **{k: v for k, v in r.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}, tol=1e-5)
    # (round 5: the large-batch contact flavour runs the finishers at the head of the next substep's launch: k_substep_pf)
    assert r["chains"] == 2 and r["envs_checked"] == [0, 8] and "k_substep_pf<256,1024,true,1>" in r["flavour"], r
    assert r["mesh_contact"] and r["particles_with_candidates"] > 0 and r["x_max_abs"] < 1e-5, r
    assert r["hard_rgb_mismatch_pixels"] == 0 and r["hard_depth_mismatch_pixels"] == 0, r
    assert r["passed"], r


def test_T_pusher_32_envs_batch_independence_determinism_and_gate(monkeypatch):
    """configs[3] at its per-GPU size: 32 environments, the ~25k-face rod against the T block, self-collision rebuild on (its
    real settings).  (1) the parity gate of that scene (one environment vs PhysOracle + EefOracle's pusher branch, in contact);
    (2) two runs of the 32-environment batch are bit-identical; (3) environment 0 of the batch equals a 1-environment run of the
    same scene (batch independence), every environment moved its block, none of them left the table."""
    import torch
    from oracle import parity_gate
    from r2s_hip.rollout import BatchedRollout

    r = parity_gate.run("T_pusher_32env", num_substeps=200, n_compare=16, close_at=1, render=False)
    record("T_pusher scene, 16 substeps with the rod against the block", **{k: v for k, v in r.items() if isinstance(v, (int, float)) and not isinstance(v, bool)}, tol=1e-5)
    assert r["mesh_contact"] and r["x_max_abs"] < 1e-5 and r["passed"], r

    def run(n_env):
        ro = BatchedRollout("T_pusher_32env", n_env=n_env, num_substeps=200, seed=0, close_at=1)
        for _ in range(4):
            ro.step()
        torch.cuda.synchronize()
        st = ro.contact_stats()
        return ro.phys.x.cpu().numpy().copy(), ro.out_color.cpu().numpy().copy(), st, ro.ob["points"] + ro.env_shift[:, None]

    xa, ca, st, x_init = run(32)
    xb, cb, _, _ = run(32)
    assert st["mesh_contacts"] > 0 and st["flavour"]["mesh_template"] == 2
    assert np.array_equal(xa, xb) and np.array_equal(ca, cb), "two runs of the same batch must be bit-identical"
    # a 1-environment handle is laid out in 64-particle blocks (the resident stepper's layout: neighbours in another order, the same
    # sums to the last bits); with the large-batch layout forced it must reproduce environment 0 of the batch bit for bit
    x1, c1, _, _ = run(1)
    assert np.abs(xa[0] - x1[0]).max() < 2e-6, np.abs(xa[0] - x1[0]).max()
    monkeypatch.setenv("R2S_RESIDENT", "0")
    x1, c1, _, _ = run(1)
    assert np.array_equal(xa[0], x1[0]), np.abs(xa[0] - x1[0]).max()
    moved = np.abs(xa - x_init).max(axis=(1, 2))
    assert (moved > 1e-4).all() and (moved < 0.05).all() and np.isfinite(xa).all()
