"""The resident stepper (k_steps_resident, physics.hip): small batches — one environment of the reference's own evaluation
loop (eval_policy.py:180-240 -> phystwin.py:104-147 -> spring_mass_warp.py:723-726, `for i in range(num_substeps): step()`),
up to a few — run all substeps of an env step in ONE launch whose workgroups hand their halo particles to each other through
tagged write-through records.  Checked against the oracle, against the per-substep kernels of the same handle, for determinism,
for ragged sizes / several environments / partial steps, with moving fingers hovering and touching, and next to a second stream
that keeps the chip busy."""
import numpy as np
import pytest

from util_parity import close, record
from util_physics import gripper_motion, hip_env, make_object, oracle_env, two_blobs

pytestmark = pytest.mark.gpu


def _falling(shape, n, seed, lift=0.0005):
    ob = make_object(shape, n, seed=seed)
    ob["points"][:, 2] += lift - ob["points"][:, 2].min()      # lowest particle `lift` above the ground
    ob["v0"] = np.zeros_like(ob["points"])
    ob["v0"][:, 2] = -0.3
    return ob


@pytest.mark.parametrize("shape,n,n_env,n_sub", [("rope", 600, 1, 201), ("sloth", 3000, 3, 120), ("T", 2229, 2, 100), ("rope", 70, 1, 64)],
                         ids=["rope 1 env odd substeps", "sloth 3 envs", "T 2 envs even substeps", "two blocks, one ragged"])
def test_resident_launch_vs_oracle_and_vs_the_per_substep_kernels(shape, n, n_env, n_sub):
    """Free flight onto the ground (spring forces, drag, ground contact): the env step as one resident launch against the oracle
    (1e-5, BASELINE.json) and against the same handle running the per-substep kernels of the same layout (set_resident(False));
    the two differ by summation order only."""
    ob = _falling(shape, n, seed=3)
    kw = dict(num_substeps=n_sub, self_collision=False)
    o = oracle_env(ob, **kw)
    h = hip_env(ob, n_env=n_env, **kw)
    g = hip_env(ob, n_env=n_env, **kw)
    g.set_resident(False)
    st = h.layout_stats()
    assert st["lds_bytes"] == 512 * 24 and st["fallback_slots"] == 0, st
    for _ in range(3):
        o.step(); h.step(); g.step()
    fl, fg = h.last_flavour(), g.last_flavour()
    assert fl["resident"] and fl["kernel"].startswith("k_steps_resident<512,"), fl
    assert not fg["resident"] and fg["kernel"].startswith("k_steps_resident<512,false,") and "x 1 substep" in fg["kernel"], fg
    x, xg = h.x.cpu().numpy(), g.x.cpu().numpy()
    err_o = float(np.abs(x - o.x[None]).max())
    err_g = float(np.abs(x - xg).max())
    record(f"resident stepper, {shape} {n} x {n_env} envs, {3 * n_sub} substeps", x_max_abs_vs_oracle=err_o, x_max_abs_vs_per_substep_kernels=err_g, tol=1e-5)
    assert err_o < 1e-5 and err_g < 2e-6, (err_o, err_g)
    assert np.abs(h.v.cpu().numpy() - g.v.cpu().numpy()).max() < 1e-3
    assert np.array_equal(x, np.repeat(x[0:1], n_env, 0)), "environments with identical inputs must agree bit for bit"
    assert o.v[:, 2].max() > -0.2, "the scenario must reach the ground (free fall alone leaves every particle faster than -0.3 m/s)"


def test_environments_with_different_states_and_a_stiffness_update_between_steps():
    """Environments of one resident launch are independent: three environments of the same object with different initial velocities
    and heights, each against its OWN oracle run (identical environments would hide an indexing slip between the per-environment
    slices of the exchange array).  Between two env steps the stiffness is replaced (set_spring_Y: the adjacency the launch keeps in
    registers is reloaded by the next launch)."""
    from r2s_hip.physics import PhysBatch
    from util_physics import DEFAULTS

    ob = _falling("sloth", 3000, seed=13)
    E, n_sub = 3, 100
    x0 = np.repeat(ob["points"][None], E, 0).copy()
    v0 = np.zeros_like(x0)
    for e in range(E):
        x0[e, :, 2] += 0.0004 * e
        v0[e, :, 2] = -0.2 - 0.15 * e
        v0[e, :, 0] = 0.05 * (e - 1)
    kw = dict(DEFAULTS, num_substeps=n_sub, self_collision=False)
    h = PhysBatch(init_vertices=x0, init_springs=ob["springs"], init_rest_lengths=ob["rest"], init_masses=np.ones(len(ob["points"]), np.float32),
                  init_spring_Y=ob["log_Y"], init_velocities=v0, **kw)
    oracles = []
    for e in range(E):
        oe = dict(ob, points=x0[e], v0=v0[e])
        oracles.append(oracle_env(oe, num_substeps=n_sub, self_collision=False))
    logy2 = (ob["log_Y"] + np.log(0.6)).astype(np.float32)
    for k in range(3):
        if k == 2:
            h.set_spring_Y(logy2)
            for o in oracles:
                o.log_Y[:] = logy2            # the oracle reads its stiffness array every substep
        h.step()
        for o in oracles:
            o.step()
    assert h.last_flavour()["resident"]
    x = h.x.cpu().numpy()
    errs = [float(np.abs(x[e] - oracles[e].x).max()) for e in range(E)]
    record("resident stepper, 3 different environments + stiffness update", x_max_abs_per_env=max(errs), tol=1e-5)
    assert max(errs) < 1e-5, errs
    assert np.abs(x[0] - x[2]).max() > 1e-3, "the environments must actually differ"


def test_resident_launch_is_deterministic_and_partial_steps_compose():
    """Two handles, same inputs: bit-identical states.  And step(n, first) pieces — each its own resident launch, final state
    always in the OTHER buffer whatever the parity of n — compose to the full step bit for bit."""
    ob = _falling("sloth", 3000, seed=5)
    kw = dict(num_substeps=90, self_collision=False)
    a, b, c = (hip_env(ob, n_env=2, **kw) for _ in range(3))
    for _ in range(2):
        a.step(); b.step()
        c.step(20, 0); c.step(31, 20); c.step(39, 51)
    assert a.last_flavour()["resident"] and c.last_flavour()["resident"]
    xa, xb, xc = a.x.cpu().numpy(), b.x.cpu().numpy(), c.x.cpu().numpy()
    assert np.array_equal(xa, xb) and np.array_equal(a.v.cpu().numpy(), b.v.cpu().numpy())
    assert np.array_equal(xa, xc) and np.array_equal(a.v.cpu().numpy(), c.v.cpu().numpy())


@pytest.mark.parametrize("servers", [True, False], ids=["query servers in the launch", "R2S_RES_SERVERS=0 (round-3 behaviour)"])
def test_resident_launch_with_fingers_hovering_then_touching(monkeypatch, servers):
    """Moving finger meshes: while they hover the union-box early-out skips every per-mesh test and the batch stays on the resident
    launch (small batches defer once a query was NEEDED, not when something is merely near); when the fingers come down the
    wavefronts in reach run the exact tests and hand the particles that need a query to the launch's server pairs (round 4) — the
    batch stays resident — or, without servers, answer them in place inside the launch (the first env step in reach: the host then
    switches to the deferred flavour one step later).  All against the oracle, forces included."""
    import torch
    from r2s_hip import synth

    if not servers:
        monkeypatch.setenv("R2S_RES_SERVERS", "0")

    n_sub = 120
    ob = make_object("sloth", 500, seed=6)
    c = ob["points"].mean(0)
    top = ob["points"][:, 2].max()
    kw = dict(self_collision=False, num_substeps=n_sub)

    def run(height, vel, expect_contact):
        fl = synth.finger_mesh((c[0], c[1] - 0.02, top + height))
        fr = synth.finger_mesh((c[0], c[1] + 0.02, top + height))
        interp, centers, dv, om = gripper_motion([fl, fr], n_sub, 5e-5, vel=vel, closing=1.0 if expect_contact else 0.0)
        o = oracle_env(ob, dynamic_meshes=[fl, fr], **kw)
        h = hip_env(ob, dynamic_meshes=[fl, fr], **kw)
        o.set_mesh_interactive(interp, centers, dv, om)
        h.set_mesh_interactive(torch.from_numpy(interp)[None].cuda(), torch.from_numpy(centers)[None].cuda(),
                               torch.from_numpy(dv)[None].cuda(), torch.from_numpy(om)[None].cuda())
        o.step(); h.step()
        assert h.last_flavour()["resident"], h.last_flavour()
        x = h.x[0].cpu().numpy()
        f = h.collision_forces()[0].cpu().numpy()
        hit = float(np.abs(o.collision_forces).max()) > 0
        assert hit == expect_contact
        assert np.abs(x - o.x).max() < 1e-5, np.abs(x - o.x).max()
        for m in (0, 1):
            tot_o, tot_h = o.collision_forces[h.mesh_map == m].sum(0), f[h.mesh_map == m].sum(0)
            assert np.allclose(tot_h, tot_o, rtol=1e-3, atol=max(np.abs(tot_o).max() * 1e-3, 1e-6)), (m, tot_o, tot_h)
        near = int(h.deferred_counts()[n_sub])
        h.step(); h.step()                                     # the flavour of step t follows from what step t - 2 saw (fixed lag, round 5)
        return float(np.abs(x - o.x).max()), near, h.last_flavour()

    e_far, near_far, fl = run(0.09, (0.0, 0.0, -0.5), False)  # 9 cm up, 3 mm down per step
    assert near_far == 0 and fl["resident"]
    e_hov, _, fl = run(0.02, (0.0, 0.0, -0.05), False)        # hovering 2 cm above the toy: "near" for a large batch, free motion for a small one
    assert fl["resident"], "a small batch stays on the resident launch while no particle needs a mesh query"
    e_hit, near_hit, fl = run(0.03, (0.0, 0.0, -6.0), True)   # the scenario of test_gripper_fingers_dynamic_mesh
    assert near_hit != 0, "particles inside a margin must be reported"
    if servers:
        assert fl["resident"], "with query servers in the launch a small batch stays resident through contact"
    else:
        assert not fl["resident"] and fl["deferred_mesh_queries"], "two steps after a step that needed queries the deferred flavour runs"
    record(f"resident stepper with finger meshes (hovering / {'server' if servers else 'in-place'} queries)", x_max_abs_hover=e_far, x_max_abs_contact=e_hit, tol=1e-5)


@pytest.mark.parametrize("n_env,own", [(1, True), (2, True), (1, False)],
                         ids=["1 env, pairs own their particle", "2 envs, pairs own their particle", "1 env, R2S_RES_SRV_OWN=0 (a request per substep)"])
def test_resident_launch_stays_resident_through_a_held_grasp(monkeypatch, n_env, own):
    """one environment IN CONTACT used to leave the resident launch (per-substep kernels + finishing launch: 11.8 us
    per substep for the rope against 2.5 free).  With query servers in the launch the fingers close on the rope, squeeze it, hold it and
    lift it over five env steps and EVERY step is one resident launch: positions against the oracle (1e-5, BASELINE.json), against the
    per-substep kernels + finishing launch of a second handle (2e-6: the same queries, another summation order of the springs), the
    per-finger force totals of the last substep, and the server protocol's corner cases on the way — particles that enter a margin in
    different substeps (claims mid-launch), particles that leave it again, a step in which nothing touches.  Both protocols: a pair
    that OWNS its particle from the claim on (its springs from the neighbours' exchange records, velocity, mesh response, ground — the
    default) and the first one (the block sends a request per substep, the pair answers the query)."""
    import torch
    from r2s_hip import synth

    if not own:
        monkeypatch.setenv("R2S_RES_SRV_OWN", "0")
    n_sub = 300
    ob = make_object("rope", 900, seed=4)
    c = ob["points"].mean(0)
    top = ob["points"][:, 2].max()
    kw = dict(self_collision=False, num_substeps=n_sub)
    z = 0.002 + 0.025 + 0.004                                            # finger centres: tips 6 mm above the table after the descent below
    fingers = [synth.finger_mesh((c[0], c[1] - 0.030, z + 0.012), pad_normal=(0.0, 1.0, 0.0)), synth.finger_mesh((c[0], c[1] + 0.030, z + 0.012), pad_normal=(0.0, -1.0, 0.0))]
    o = oracle_env(ob, dynamic_meshes=fingers, **kw)
    h = hip_env(ob, n_env=n_env, dynamic_meshes=fingers, **kw)
    g = hip_env(ob, n_env=n_env, dynamic_meshes=fingers, **kw)
    g.set_resident(False)
    T = n_sub * 5e-5
    #        eef velocity           closing speed (each finger towards the other), m/s
    script = [((0.0, 0.0, -0.012 / T), 0.0),          # down: nothing touches yet (rope radius 12 mm, pads 30 mm from the axis)
              ((0.0, 0.0, 0.0), 0.010 / T),           # close: pads from 13 mm to 3 mm off the rope's surface -> particles enter the 5 mm margin mid-step
              ((0.0, 0.0, 0.0), 0.002 / T),           # squeeze by 2 mm
              ((0.0, 0.0, 0.0), 0.0),                 # hold
              ((0.0, 0.0, 0.01 / T), 0.0)]            # lift 1 cm with the rope in the fingers
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].repeat(n_env, *([1] * a.ndim)).cuda()  # noqa: E731
    worst_o, worst_g, touched = 0.0, 0.0, []
    for k, (vel, closing) in enumerate(script):
        interp, centers, dv, om = gripper_motion(fingers, n_sub, 5e-5, vel=vel, closing=closing)
        o.set_mesh_interactive(interp, centers, dv, om)
        for hh in (h, g):
            hh.set_mesh_interactive(tt(interp), tt(centers), tt(dv), tt(om))
        o.step(); h.step(); g.step()
        fl, fg = h.last_flavour(), g.last_flavour()
        assert fl["resident"], (k, fl)
        assert not fg["resident"], (k, fg)
        x, xg = h.x.cpu().numpy(), g.x.cpu().numpy()
        eo, eg = float(np.abs(x - o.x[None]).max()), float(np.abs(x - xg).max())
        worst_o, worst_g = max(worst_o, eo), max(worst_g, eg)
        assert eo < 1e-5 and eg < 2e-6, (k, eo, eg)
        f = h.collision_forces().cpu().numpy()
        hit = float(np.abs(o.collision_forces).max()) > 0
        touched.append(hit)
        for e in range(n_env):
            for m in (0, 1):
                tot_o, tot_h = o.collision_forces[h.mesh_map == m].sum(0), f[e][h.mesh_map == m].sum(0)
                assert np.allclose(tot_h, tot_o, rtol=1e-3, atol=max(np.abs(tot_o).max() * 1e-3, 1e-6)), (k, e, m, tot_o, tot_h)
        nl = len(fingers[0][0])
        fingers = [(interp[-1][:nl], fingers[0][1]), (interp[-1][nl:], fingers[1][1])]
    assert touched[0] is False and all(touched[2:]), touched
    assert float(o.x[:, 2].max()) > top + 0.002, "the rope must have been lifted"
    h.step()            # a timed-out hand-off (halo or server) would have raised the sticky fault: this call reports it
    record(f"resident launch through a held grasp, {n_env} env(s), 5 env steps, {'owning pairs' if own else 'a request per substep'}", x_max_abs_vs_oracle=worst_o, x_max_abs_vs_per_substep_kernels=worst_g, tol=1e-5)


def test_resident_launch_next_to_a_busy_second_stream():
    """Hand-offs under load: a second stream keeps every CU streaming through HBM while the resident launches run (uneven
    arrival of the workgroups, loaded memory queues).  The states must equal the quiet run bit for bit."""
    import torch

    ob = _falling("sloth", 3000, seed=7)
    kw = dict(num_substeps=150, self_collision=False)
    quiet, busy = hip_env(ob, n_env=3, **kw), hip_env(ob, n_env=3, **kw)
    for _ in range(3):
        quiet.step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")      # 1 GiB
    for _ in range(3):
        with torch.cuda.stream(side):
            for _ in range(6):
                big.mul_(1.0001)
        busy.step()
    torch.cuda.synchronize()
    assert busy.last_flavour()["resident"]
    assert np.array_equal(quiet.x.cpu().numpy(), busy.x.cpu().numpy()) and np.array_equal(quiet.v.cpu().numpy(), busy.v.cpu().numpy())


def test_two_handles_on_two_streams_do_not_starve_each_other():
    """A resident launch needs all its workgroups on the chip at once, one per CU; two handles stepping on two streams — 141 + 141
    workgroups on 256 CUs — would each hold part of the chip and spin for workgroups that no longer fit.  The library runs the
    resident launches of a device one after the other (event chain across handles): both rollouts finish and equal their solo runs."""
    import torch

    ob = _falling("sloth", 3000, seed=11)
    kw = dict(num_substeps=100, self_collision=False)
    solo = hip_env(ob, n_env=3, **kw)
    for _ in range(4):
        solo.step()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        a = hip_env(ob, n_env=3, **kw)
    with torch.cuda.stream(s2):
        b = hip_env(ob, n_env=3, **kw)
    torch.cuda.synchronize()
    for _ in range(4):
        with torch.cuda.stream(s1):
            a.step(sync_state=False)
        with torch.cuda.stream(s2):
            b.step(sync_state=False)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        xa = a.sync_state()[0].cpu().numpy()
    with torch.cuda.stream(s2):
        xb = b.sync_state()[0].cpu().numpy()
    assert a.last_flavour()["resident"] and b.last_flavour()["resident"]
    xs = solo.x.cpu().numpy()
    assert np.array_equal(xa, xs) and np.array_equal(xb, xs)
    a.step(); b.step()          # the sticky fault word of a timed-out hand-off would make these raise


def test_large_batches_and_forced_layouts_keep_the_per_substep_path(monkeypatch):
    """More work items than the chip holds at once (or R2S_RESIDENT=0): the large-batch layout and one kernel per substep."""
    ob = _falling("sloth", 3000, seed=9)
    kw = dict(num_substeps=40, self_collision=False)
    h = hip_env(ob, n_env=9, **kw)           # 47 blocks of 64 x 9 envs > 256 work items
    h.step()
    assert not h.last_flavour()["resident"] and h.layout_stats()["lds_bytes"] == 1024 * 24
    monkeypatch.setenv("R2S_RESIDENT", "0")
    g = hip_env(ob, n_env=1, **kw)
    g.step()
    assert not g.last_flavour()["resident"] and g.layout_stats()["lds_bytes"] == 1024 * 24


def test_resident_launch_with_query_servers_under_load_is_bit_identical_to_the_quiet_run():
    """The server protocol under UNEVEN load (cdna_hip_programming.md Guideline 16: test every hand-off with loaded memory queues and
    late workgroups): the bench's own `rope_1env` rollout — descend, close on the rope at step 3 (grasp detection holds the opening),
    lift, open at step 12 (release: pairs keep serving "skip" substeps), 18 env steps of 667 substeps — once on a quiet chip and once
    next to a second stream that keeps every CU streaming through HBM.  Claims, requests and results arrive in another order; the
    particle states must not differ in a single bit, every step must have stayed ONE resident launch, and nothing may have timed out."""
    import torch
    from r2s_hip.rollout import BatchedRollout

    def run(busy):
        ro = BatchedRollout("rope_1env", close_at=3, open_at=12, seed=2)
        side = torch.cuda.Stream()
        big = torch.empty(1 << 27, dtype=torch.float32, device="cuda") if busy else None      # 512 MiB
        flav, hits, ztop = [], [], []
        for _ in range(18):
            if busy:
                with torch.cuda.stream(side):
                    for _ in range(4):
                        big.mul_(1.0001)
            ro.step()
            flav.append(ro.phys.last_flavour())
            hits.append(ro.contact_stats()["mesh_contacts"])
            ztop.append(float(ro.phys.x[0, :, 2].max()))
        torch.cuda.synchronize()
        ro.phys.step(0, 0)                    # a sticky fault (a poll that hit its limit) would raise here
        torch.cuda.synchronize()
        return ro.phys.x.cpu().numpy().copy(), ro.phys.v.cpu().numpy().copy(), flav, hits, ro.phys.eef_state(), ztop

    xq, vq, fq, hq, sq, zq = run(False)
    xb, vb, fb, hb, sb, zb = run(True)
    assert all(f["resident"] and f["query_server_workgroups"] > 0 for f in fq + fb), [f["kernel"] for f in fq if not f["resident"]]
    assert max(hq[4:11]) > 0, ("the fingers must be on the rope between closing and opening", hq)
    assert np.array_equal(xq, xb) and np.array_equal(vq, vb), float(np.abs(xq - xb).max())
    assert hq == hb and torch.equal(sq[0], sb[0]) and torch.equal(sq[1], sb[1])
    assert np.isfinite(xq).all() and zq == zb
    assert max(zq[4:12]) > zq[0] + 0.004, ("the rope must have been lifted while the fingers held it", zq)


@pytest.mark.gpu
def test_more_particles_in_contact_than_server_pairs_takes_the_step_off_the_resident_launch(monkeypatch):
    """A launch whose server pairs run out (here: capped at 8 workgroups = 32 pairs, 45 particles of the rope inside the pads' margins)
    answers the surplus in place — correct, and slow enough to stall every block of the launch — and says so; the host then runs the
    following env steps as per-substep kernels + finishing launch until the contact is over, and returns to the resident launch after
    the release.  The run must agree with the unconstrained one (same physics, other summation orders) and raise nothing."""
    import torch
    from r2s_hip.rollout import BatchedRollout

    def run(cap):
        if cap:
            monkeypatch.setenv("R2S_RES_SRV_WG", str(cap))
        else:
            monkeypatch.delenv("R2S_RES_SRV_WG", raising=False)
        ro = BatchedRollout("rope_1env", close_at=3, open_at=9, seed=2)
        flav, ztop = [], []
        for _ in range(14):
            ro.step()                           # (no synchronisation: the flavour of step t follows from the counters of step t - 2, waited for — round 5)
            flav.append(ro.phys.last_flavour())
            ztop.append(float(ro.phys.x[0, :, 2].max()))
        ro.phys.step(0, 0)
        torch.cuda.synchronize()
        return ro.phys.x.cpu().numpy().copy(), flav, ztop

    xc, fc, zc = run(8)
    xf, ff, zf = run(0)
    assert all(f["resident"] for f in ff)
    off = [k for k, f in enumerate(fc) if not f["resident"]]
    assert off, "with 32 pairs for 45 particles some env steps must have left the resident launch"
    assert all(fc[k]["deferred_mesh_queries"] for k in off)
    assert fc[-1]["resident"] and fc[0]["resident"], ("free motion before the grasp and after the release is resident again", [f["resident"] for f in fc])
    assert np.isfinite(xc).all() and max(zc[4:9]) > zc[0] + 0.004, zc
    assert np.abs(xc - xf).max() < 2e-3, float(np.abs(xc - xf).max())   # 14 env steps of a grasp: a chaotic system summed in three different orders
    record("rope grasp with the server pairs capped below the contact count", x_max_abs_vs_uncapped=float(np.abs(xc - xf).max()), steps_off_the_resident_launch=len(off), tol=2e-3)


@pytest.mark.gpu
def test_a_cu_budget_too_small_for_the_resident_launch_is_refused_at_create_time(monkeypatch):
    """the resident launch needs all its workgroups on the chip at once, and r2s_phys_create ASKS
    (hipOccupancyMaxActiveBlocksPerMultiprocessor for k_steps_resident, the device's CU count, R2S_RES_CU_BUDGET for a partition / a shared
    device) instead of assuming.  With a budget of 64 CUs the 130 work items of the one-environment rope do not fit: the handle must pick the
    per-substep kernels of the same layout at create time, say why, and step through a grasp without a fault word."""
    import torch
    from r2s_hip import _lib
    from r2s_hip.rollout import BatchedRollout

    monkeypatch.setenv("R2S_RES_CU_BUDGET", "64")
    ro = BatchedRollout("rope_1env", close_at=2, seed=2, settle_steps=2)
    flav = []
    for _ in range(5):
        ro.step()
        flav.append(ro.phys.last_flavour())
    torch.cuda.synchronize()
    ro.phys.step(0, 0)                    # a sticky fault would raise here
    torch.cuda.synchronize()
    assert not any(f["resident"] for f in flav), [f["kernel"] for f in flav]
    assert all("x 1 substep" in f["kernel"] for f in flav), [f["kernel"] for f in flav]   # the small-batch layout, one launch per substep
    assert bool(torch.isfinite(ro.phys.x).all())
    monkeypatch.delenv("R2S_RES_CU_BUDGET")
    ro2 = BatchedRollout("rope_1env", close_at=2, seed=2, settle_steps=2)
    ro2.step()
    assert ro2.phys.last_flavour()["resident"], "the same scene on the whole chip runs the resident launch"


@pytest.mark.gpu
@pytest.mark.parametrize("with_meshes", [False, True], ids=["no meshes (k_steps_resident<512,true,0>)", "finger meshes out of reach (<512,true,1>)"])
def test_resident_launch_with_live_self_collision_candidates(monkeypatch, with_meshes):
    """VERDICT r4 item 7 (spring_mass_warp.py:132-268, 857-866): a small batch with live self-collision candidates stays ONE resident
    launch per env step — a particle with candidates publishes {x0, post-force v} of every substep in a tagged record, wavefront 0 of its
    block polls its candidates' records, sums the impulses in list order and finishes it.  Two blobs thrown at each other: the launch must
    be resident with the self-collision flavour, agree with the oracle like the per-substep kernels do (5e-5 over 800 substeps with
    contacts, 1e-5 over the first env step in which impulses act), with the per-substep flavour of the same handle layout (R2S_RES_SELF=0),
    and two runs must end in the same bits."""
    import torch
    from r2s_hip import synth

    ob = two_blobs(seed=1, gap=0.06, speed=3.0)
    n_sub = 200
    kw = dict(num_substeps=n_sub, collide_self_fric=0.3)
    if with_meshes:
        c = ob["points"].mean(0)
        kw["dynamic_meshes"] = [synth.finger_mesh((c[0], c[1] - 0.3, c[2] + 0.3)), synth.finger_mesh((c[0], c[1] + 0.3, c[2] + 0.3))]

    def run(res_self):
        if res_self:
            monkeypatch.delenv("R2S_RES_SELF", raising=False)
        else:
            monkeypatch.setenv("R2S_RES_SELF", "0")
        o = oracle_env(ob, **kw)
        h = hip_env(ob, **kw)
        flav, first_err, cands = [], None, 0
        for _ in range(4):
            o.update_collision_graph(); h.update_collision_graph()
            had = int(o.coll_num.sum()) > 0
            cands += int(o.coll_num.sum())
            o.step(); h.step()
            flav.append(h.last_flavour())
            if had and first_err is None:
                first_err = float(np.abs(h.x[0].cpu().numpy() - o.x).max())
        torch.cuda.synchronize()
        x_end = h.x[0].cpu().numpy().copy()
        h.step()                          # a sticky fault (a poll that hit its limit) would raise here
        torch.cuda.synchronize()
        return x_end, o, flav, first_err, cands

    xa, o, fa, e1, cands = run(True)
    assert cands > 0, "scenario must produce contacts"
    with_cand = [f for f in fa if f["self_collision_kernel"]]
    assert with_cand and all(f["resident"] for f in fa), [f["kernel"] for f in fa]
    assert all("k_steps_resident<512,true" in f["kernel"] for f in with_cand), [f["kernel"] for f in with_cand]
    assert close(xa, o.x, 5e-5, what=f"resident launch with live candidates, 4 x {n_sub} substeps, meshes={with_meshes}")
    xb, _, fb, *_ = run(False)
    assert not all(f["resident"] for f in fb), "R2S_RES_SELF=0: the steps with candidates take the per-substep kernels"
    assert float(np.abs(xa - xb).max()) < 2e-5, float(np.abs(xa - xb).max())     # two summation orders through 800 substeps with contacts
    xc, *_ = run(True)
    assert np.array_equal(xa, xc), "two runs, the same bits"
    record(f"resident launch with live self-collision candidates (meshes={with_meshes})", x_max_abs=float(np.abs(xa - o.x).max()),
           x_max_abs_first_contact_step=e1, x_vs_per_substep_flavour=float(np.abs(xa - xb).max()), tol=5e-5)


@pytest.mark.gpu
def test_rope_folded_onto_itself_stays_in_the_resident_launch():
    """VERDICT r4 item 7, its "done" list: a rope folded onto itself (rest shape a hairpin, gravity lays the upper leg onto the lower one:
    3 000 of the 8 000 particles carry candidates from env step 3 on, up to six each — the bench's `rope_fold_1env`) stays in
    `k_steps_resident`, the self-collision flavour of the one-launch stepper with the finger meshes in the scene.  Thousands of pairs in
    sustained contact are a chaotic system at float32 round-off — the ORACLE itself, started one ulp away, is 1e-3 off after one env
    step (tests/test_physics_oracle_kat.py pins that) — so the whole-step comparison at 1e-5 is made where it can be (the next test) and this
    one holds the flavour to the oracle over the first 5 substeps of an env step in full contact (1e-6: before the first decision can
    flip), to its own per-substep form over the same substeps, and to a bounded, finite state over the whole step."""
    from oracle import parity_gate

    r = parity_gate.run("rope_fold_1env", n_env=1, n_compare=5, close_at=2, render=False)
    assert r["particles_with_candidates"] > 1000, r
    assert r["flavour"].startswith("k_steps_resident<512,true,1>") and "substep +" not in r["flavour"], r["flavour"]
    assert r["x_max_abs"] < 1e-6 and r["v_max_abs"] < 2e-3, r
    record("rope folded onto itself (3 000 particles with candidates), resident launch, first 5 substeps of an env step vs oracle",
           x_max_abs=r["x_max_abs"], v_max_abs=r["v_max_abs"], particles_with_candidates=r["particles_with_candidates"], tol=1e-6)
    import torch
    from r2s_hip.rollout import BatchedRollout
    ro = BatchedRollout("rope_fold_1env")
    for _ in range(8):
        ro.physics_step(); ro.t += 1
    torch.cuda.synchronize()
    fl, st = ro.phys.last_flavour(), ro.contact_stats()
    assert fl["resident"] and fl["self_collision_kernel"] and st["self_collision_candidates"] > 1000, (fl, st)
    x = ro.phys.x.cpu().numpy()
    assert np.isfinite(x).all() and x[..., 2].min() > -1e-3 and x[..., 2].max() < 0.2
    ro.phys.step()                    # a sticky fault (a poll that hit its limit) would raise here
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_tip_of_the_rope_folded_back_resident_launch_against_the_oracle_and_bit_for_bit_against_the_per_substep_kernels(monkeypatch):
    """The other half of item 7's "done" list, as far as float32 lets it be had.  8 cm of the rope's end folded back over it touches down in
    env step 3 with a handful of candidates (at most three per particle).  One whole env step of 667 substeps in the resident launch:
    against the oracle 5e-7 over the first 100 substeps and 4e-5 over all 667 — between substep 100 and 300 ONE contact decision
    (`dis < collision_dist`, `dot(dis, rv) < -1e-4`, :155) falls the other way on the device, whose spring sums run in another order than
    the oracle's; 1e-5 over a whole step needs no such decision to sit within an ulp, which the two-blob scene above has (1.5e-6 over
    800 substeps) and this one has not.  What can be held exactly is held exactly: the resident launch against the per-substep kernels
    (k_steps_resident x 1 substep + k_self_finish, the flavour rounds 2-4 held to the oracle) — the same decisions, the same sums in
    the same order with at most three candidates a particle: every env step of the rollout BIT FOR BIT."""
    import torch
    from oracle import parity_gate
    from r2s_hip.rollout import BatchedRollout

    monkeypatch.delenv("R2S_RES_SELF", raising=False)
    r = parity_gate.run("rope_tip_fold_1env", n_env=1, n_compare=100, close_at=1, render=False)
    assert r["particles_with_candidates"] > 0 and r["compared_at_env_step"] == 3, r
    assert r["flavour"].startswith("k_steps_resident<512,true,1>") and "substep +" not in r["flavour"], r["flavour"]
    assert r["x_max_abs"] < 2e-6, r
    r2 = parity_gate.run("rope_tip_fold_1env", n_env=1, n_compare=667, close_at=1, render=False)
    assert r2["flavour"] == r["flavour"] and r2["x_max_abs"] < 1e-4, r2
    record("tip of the rope folded back, resident launch with live candidates vs oracle", x_max_abs_100_substeps=r["x_max_abs"],
           x_max_abs_667_substeps=r2["x_max_abs"], particles_with_candidates=r["particles_with_candidates"], tol=1e-4)

    def rollout(res_self):
        monkeypatch.setenv("R2S_RES_SELF", "1" if res_self else "0")
        ro = BatchedRollout("rope_tip_fold_1env")
        xs, fl, nc = [], [], []
        for _ in range(7):
            ro.physics_step(); ro.t += 1
            xs.append(ro.phys.x.cpu().numpy().copy()); fl.append(ro.phys.last_flavour()); nc.append(ro.contact_stats()["self_collision_candidates"])
        ro.phys.step()
        torch.cuda.synchronize()
        return xs, fl, nc

    xa, fa, na = rollout(True)
    xb, fb, nb = rollout(False)
    assert max(na) > 0 and na == nb, (na, nb)
    with_c = [k for k in range(7) if fa[k]["self_collision_kernel"]]
    assert with_c and all(fa[k]["resident"] for k in with_c) and not any(fb[k]["resident"] for k in with_c), ([f["kernel"] for f in fa], [f["kernel"] for f in fb])
    for k in range(7):
        assert np.array_equal(xa[k], xb[k]), (k, float(np.abs(xa[k] - xb[k]).max()))
    monkeypatch.delenv("R2S_RES_SELF", raising=False)


@pytest.mark.gpu
def test_resident_self_collision_with_more_pairs_in_a_block_than_the_task_table_holds():
    """The block's (particle, candidate) pairs beyond the 1 024 slots of the task table are wavefront 0's own, candidate after candidate,
    behind the tabled ones in list order.  Injected lists (r2s_phys_set_collision_lists, the reference's collision_indices /
    collision_number): every particle of the flying blob lists its 24 nearest of the other blob and is listed by them — 1 536 pairs and
    more in every block of 64 — the same lists in the oracle; most pairs never come within collision_dist, the ones that do collide."""
    import torch

    ob = two_blobs(seed=1, gap=0.03, speed=1.0)
    kw = dict(num_substeps=200, collide_self_fric=0.3)
    o, h = oracle_env(ob, **kw), hip_env(ob, **kw)
    N = len(ob["points"]); nA = int((ob["v0"][:, 0] == 0).sum())
    hits = 0
    for _ in range(4):
        x = o.x.astype(np.float64)
        d = np.linalg.norm(x[nA:, None, :] - x[None, :nA, :], axis=2)          # [B, A]
        near = np.argsort(d, axis=1, kind="stable")[:, :24]
        lists = [[] for _ in range(N)]
        for bi in range(N - nA):
            for a in near[bi]:
                lists[nA + bi].append(int(a)); lists[int(a)].append(nA + bi)
        num = np.array([len(l) for l in lists], np.int32)
        idx = np.zeros((N, int(num.max())), np.int32)
        for i, l in enumerate(lists):
            idx[i, : len(l)] = sorted(l)
        assert num[nA:].min() == 24 and num.max() <= o.coll_idx.shape[1]
        o.coll_num[:] = num; o.coll_idx[:, : idx.shape[1]] = idx
        h.set_collision_lists(num[None], idx[None])
        v_before = o.v.copy()
        o.step(); h.step()
        fl = h.last_flavour()
        assert fl["resident"] and "k_steps_resident<512,true" in fl["kernel"], fl
        hits += int(np.abs(o.v - v_before).max() > 0.5)
    torch.cuda.synchronize()
    x_end = h.x[0].cpu().numpy().copy()
    h.step()
    torch.cuda.synchronize()
    assert hits > 0, "the blobs must collide inside the window"
    assert close(x_end, o.x, 5e-5, what="resident launch, 1 536+ candidate pairs per block (task table + wavefront 0's tail)")
    record("resident launch with more candidate pairs in a block than the task table holds", x_max_abs=float(np.abs(x_end - o.x).max()), tol=5e-5)
