"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.  The parity gate tied to timing (SURVEY.md §8d "every timed configuration first
passes ..."): one environment of the bench's OWN scene and action trace is driven through the product path (BatchedRollout ->
C ABI -> HIP kernels) up to the phase the timed window measures, then a few substeps of the contact flavour and one rendered
frame are compared with the oracle.  Used by tests/test_contact_flavours_gpu.py and by ``bench.py --parity-gate`` (before the
timed window; the oracle is the checker here, never the thing measured).

What runs, for a gripper scene (sloth_32env, rope_1env, sloth_multicam_8env):
  1. a 1-environment BatchedRollout of the config with the grasp at env step ``close_at`` (settled like the bench's);
  2. every env step, oracle/eef_oracle.EefOracle (phystwin.py:362-513 restated) is stepped next to the device kinematics on the
     same inputs — end-effector pose / rates / commanded opening of the synthetic trace, the previous step's per-face forces —
     and its interpolated vertices / centres / finger velocities are compared with the device's (the grasp state machine
     exactly);
  3. in the first env step AFTER the fingers closed (arms pressed together: live self-collision candidates + finger contact,
     the flavour k_substep<..,true,1> + k_contact_finish<3,true>), the particle state is copied to oracle.PhysOracle
     (spring_mass_warp.py:823-943 restated), both rebuild their candidate lists, both run ``n_compare`` substeps driven by the
     EefOracle's arrays / the device kinematics, positions are compared (gate 1e-5 abs, BASELINE.json);
  4. the side-camera frame of the environment is rendered by the product path and by the raster oracle (forward.cu:262-394
     restated) and compared (|d| <= 1e-5 + 1e-4 |ref|, at most 1e-4 of the pixels outside).
For the pusher scene (T_pusher_32env) step 2 uses the pusher branch (phystwin.py:462-510) and step 3 runs in the first env step
that starts with the rod against the block.
"""
from __future__ import annotations

import time

import numpy as np


def _eef_pts_func(table):
    from .eef_oracle import make_eef_pts_func

    return make_eef_pts_func(table)


def run(config="sloth_32env", device="cuda:0", seed=0, num_substeps=667, n_compare=20, close_at=2, max_steps=None, render=True,
        settle_steps=None, n_env=1, res=None, close_rate=None):
    """Returns a dict: x_max_abs, v_max_abs, rgb / depth mismatch classes, ... and ``passed``.

    ``n_env`` > 1 runs the gate on a BATCH of that many environments — the large-batch layout and, from 256 work items on, several
    concurrent kernel chains: the flavour a multi-environment bench window times (bench.py passes 9: two chains) — and checks the
    first and the last environment (they sit in different chains) against an oracle each; ``flavour`` / ``chains`` / ``layout`` in
    the result name what ran.  ``res``: (W, H) override of the config's frame size.  ``close_rate`` (round 6; bench.py passes its own): the
    gripper closes like a policy closes it — the commanded opening ramps down, the grasp state machine latches from the stepper's own
    forces — and the substeps compared are those of the first env step that STARTS in the grasp (opening frozen, both pads loaded,
    the arms pressed together): the state a real episode spends its time in.

    ``x_ulp_spread``: the same ``n_compare`` substeps by the ORACLE from a start state one ulp away (every coordinate moved to the next
    float) against the oracle itself — how far apart two correct float32 runs of this scene are after the compared substeps.  A scene in
    sustained contact amplifies round-off through its contact decisions (tests/test_physics_oracle_kat.py); ``x_max_abs`` is to be read
    against this figure, not against zero."""
    import torch

    from . import PhysOracle, raster_forward
    from .eef_oracle import EefOracle
    from r2s_hip import synth
    from r2s_hip.rollout import BatchedRollout

    t_start = time.perf_counter()
    kw = {} if res is None else dict(res=res)
    if close_rate:
        kw["close_rate"] = float(close_rate)
    ro = BatchedRollout(config, device=device, seed=seed, n_env=n_env, num_substeps=num_substeps, close_at=close_at, settle_steps=settle_steps, **kw)
    ramp = bool(close_rate) and ro.schedule == "grasp"
    ph = ro.phys
    E = ro.n_env
    envs = sorted({0, E - 1})
    fn = _eef_pts_func(ro.eef_table)
    eos = {e: EefOracle(ro.dt, num_substeps, 3e4, use_pusher=ro.use_pusher) for e in envs}
    sta = None
    if (ph.mesh_map < 0).any():
        c = ro.ob["points"].mean(0)
        sta = [synth.box_mesh((c[0] + 0.25, c[1] + 0.2, 0.135), (0.2, 0.13, 0.27))]
    os_ = {e: PhysOracle(ro.ob["points"] + ro.env_shift[e], ro.ob["springs"], ro.ob["rest"], ro.ob["log_Y"], num_substeps=num_substeps,
                         self_collision=ph.self_collision, dynamic_meshes=ro.fingers, static_meshes=sta, use_pusher=ro.use_pusher,
                         collide_eef_fric=0.2 if ro.use_pusher else 1.0) for e in envs}
    assert all(np.array_equal(o.mesh_map, ph.mesh_map) for o in os_.values())
    out = dict(config=config, particles=int(ro.N), envs_in_batch=int(E), envs_checked=envs, substeps_compared=int(n_compare),
               eef_pts_max_abs=0.0, eef_center_max_abs=0.0, eef_vel_max_abs=0.0)
    max_steps = max_steps if max_steps is not None else close_at + (16 if ramp else 6)
    # "lissajous" traces (the gripper hovers above the object) never touch inside a short window: the gate then compares the
    # flavour such a window times — free motion next to the gripper meshes — and does not ask for contact
    expects_contact = ro.schedule in ("grasp", "push")
    out["expects_contact"] = bool(expects_contact)
    compared = False
    for t in range(max_steps):
        # ---- the caller side, oracle next to device (same inputs) ----
        if ph.self_collision:
            ph.update_collision_graph()
        act = ro.synthetic_action(ro.t)
        F_prev = ph.collision_forces().cpu().numpy()
        refs = {}
        for e in envs:
            g = lambda k: act[k][e:e + 1].cpu().numpy()  # noqa: E731
            op = None if ro.use_pusher else float(act["gripper_openness"][e].item())
            refs[e] = eos[e].step(g("eef_xyz"), g("eef_vel"), g("eef_rot"), g("eef_rot_vel"), op, fn, ro.eef_init, F_prev[e], ph.mesh_map)
        ro.apply_action(act)
        pts, ctr, dv, om = [a.cpu().numpy() if a is not None else None for a in ph.mesh_motion(points=not ro.use_pusher)]
        for e in envs:
            ref = refs[e]
            if pts is not None:
                out["eef_pts_max_abs"] = max(out["eef_pts_max_abs"], float(np.abs(pts[e] - ref["interp_points"]).max()))
            out["eef_center_max_abs"] = max(out["eef_center_max_abs"], float(np.abs(ctr[e] - ref["interp_center"]).max()))
            nv = ref["dynamic_velocity"].shape[0]
            out["eef_vel_max_abs"] = max(out["eef_vel_max_abs"], float(np.abs(dv[e, :nv] - ref["dynamic_velocity"]).max()))
        if not ro.use_pusher:
            cur, grasped = ph.eef_state()
            for e in envs:
                if cur[e].item() != eos[e].current_openness or bool(grasped[e]) != eos[e].grasped:
                    out["state_machine_mismatch_at_step"] = t
        # ---- the stepper: compare a window of substeps once the scene is in the flavour the bench times in contact ----
        # gripper scenes: the first env step after the closing step whose candidate rebuild finds live pairs (the arms pressed
        # together; the rope in the fingers has none and is taken as it is); pusher scene: the first env step that STARTS with the rod
        # against the block
        # (the flavour of an env step follows from the counters of the step TWO before it — r2s_phys_step's fixed lag.  A large batch defers
        # its queries as soon as anything is NEAR a mesh — the descent raises that long before the closing step — so the step after the
        # closing step already runs the steady contact flavour, the one a timed window is spent in; a small batch defers once a query was
        # NEEDED — the closing step — and runs it from the step after next)
        if t >= close_at + (1 if ph.layout_stats()["lds_bytes"] == 1024 * 24 else 2) and not compared:
            x, v = ph.sync_state()
            n_cand = 0
            for e in envs:
                o = os_[e]
                o.x[:] = x[e].cpu().numpy(); o.v[:] = v[e].cpu().numpy()
                if ph.self_collision:
                    o.update_collision_graph()
                    n_cand = max(n_cand, int((o.coll_num > 0).sum()))
            last_chance = t == max_steps - 1
            wants_cand = ph.self_collision and not ro.use_pusher and ro.ob_shape == "sloth"
            in_grasp = (not ramp) or all(eos[e].grasped for e in envs)       # (ramp: wait for the step that starts in the grasp)
            if (ro.use_pusher or n_cand > 0 or not expects_contact or not wants_cand) and in_grasp or last_chance:
                saved = {e: (os_[e].x.copy(), os_[e].v.copy()) for e in envs}
                hit = False
                spread = 0.0
                for e in envs:
                    ref = refs[e]
                    o = os_[e]
                    o.set_mesh_interactive(ref["interp_points"], ref["interp_center"], ref["dynamic_velocity"], ref["dynamic_omega"])
                    # the oracle one ulp away from itself (same lists, same mesh motion): what round-off alone does to this scene in n_compare substeps
                    o.x[:] = np.nextafter(saved[e][0], np.float32(np.inf)).astype(o.x.dtype)
                    o.step(n_compare, 0)
                    x_nb = o.x.copy()
                    o.x[:], o.v[:] = saved[e]
                    o.step(n_compare, 0)
                    spread = max(spread, float(np.abs(x_nb - o.x).max()))
                    hit = hit or float(np.abs(o.collision_forces).max()) > 0
                if expects_contact and not hit and not last_chance:
                    for e in envs:                                  # nothing touches yet (the rod has not reached the block, the fingers
                        os_[e].x[:], os_[e].v[:] = saved[e]         # are still closing): try the next env step
                else:
                    ph.step(n_compare, 0)
                    fl = ph.last_flavour()
                    xs, vs = ph.x.cpu().numpy(), ph.v.cpu().numpy()
                    out.update(x_max_abs=max(float(np.abs(xs[e] - os_[e].x).max()) for e in envs),
                               v_max_abs=max(float(np.abs(vs[e] - os_[e].v).max()) for e in envs),
                               x_ulp_spread=spread, grasped_at_compare=[bool(eos[e].grasped) for e in envs] if not ro.use_pusher else None,
                               particles_with_candidates=n_cand, tagged_entries=int(ph.tagged_count()), mesh_contact=bool(hit),
                               deferred_per_substep_max=int(ph.deferred_counts()[:n_compare].max()), flavour=fl["kernel"],
                               chains=int(ph.layout_stats()["chains"]),
                               layout="{blocks} blocks per env, {lds_bytes} B LDS window".format(**ph.layout_stats()), compared_at_env_step=t)
                    compared = True
                    ph.step(num_substeps - n_compare, n_compare)       # the rest of this env step, on the device only
        if compared:
            break
        ph.step(0, 0)
        ro.t += 1
    if not compared:
        out.update(x_max_abs=float("inf"), v_max_abs=float("inf"), particles_with_candidates=0, tagged_entries=0, mesh_contact=False)
    # ---- frames of the last checked environment: product path vs raster oracle, every camera ----
    if render:
        ro.render()
        col, dep = ro.observations()
        torch.cuda.synchronize()
        e = envs[-1]
        sc = ro.scene_numpy(e)
        tot = dict(pixels=0, threshold_flip_pixels=0, median_depth_crossing_pixels=0, hard_rgb_mismatch_pixels=0, hard_depth_mismatch_pixels=0)
        worst_calm, worst_abs = 0.0, 0.0
        views = list(range(min(ro.views, 2)))
        for vi in views:
            cam = ro.camera_numpy(e, vi)
            _, col_ref, _, dep_ref, frag = raster_forward(sc["means3D"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                                                          cam["tanfovx"], cam["tanfovy"], ro.H, ro.W, cam["bg"], shs=sc["shs"], scales=sc["scales"],
                                                          rotations=sc["rotations"], z_threshold=cam["z_threshold"], fragile=True)
            c = col[e, vi].cpu().numpy().astype(np.float64); d = dep[e, vi].cpu().numpy().astype(np.float64)
            err = np.abs(c - col_ref)
            bad_rgb = (err > 1e-5 + 1e-4 * np.abs(col_ref)).any(0)
            bad_dep = (np.abs(d - dep_ref) > 1e-4 * np.abs(dep_ref))[0]
            f0, f1 = (frag & 1) != 0, (frag & 3) != 0
            tot["pixels"] += int(bad_rgb.size)
            tot["threshold_flip_pixels"] += int((bad_rgb & f0).sum())
            tot["hard_rgb_mismatch_pixels"] += int((bad_rgb & ~f0).sum())
            tot["median_depth_crossing_pixels"] += int((bad_dep & f1).sum())
            tot["hard_depth_mismatch_pixels"] += int((bad_dep & ~f1).sum())
            calm = (~f1)[None] & (np.abs(col_ref) > 1e-2)
            if calm.any():
                worst_calm = max(worst_calm, float((err[calm] / np.abs(col_ref[calm])).max()))
            worst_abs = max(worst_abs, float(err.max()))
        out.update(tot, rgb_max_rel=worst_calm, rgb_max_abs_incl_flips=worst_abs,
                   frame=f"{ro.W}x{ro.H}, env {e}, cameras {views} (0 = side, 1 = wrist on the gripper)",
                   rgb_rule="every pixel WITHOUT a near-threshold decision (oracle's fragile mask: alpha < 1/255, power > 0, test_T < 1e-4 within a "
                            "relative 5e-5) holds |d| <= 1e-5 + 1e-4 |ref| — rgb_max_rel is the worst relative error over their lit channels; "
                            "threshold_flip_pixels are reported on their own and may touch at most 1e-4 of the pixels; hard mismatches must be 0")
    else:
        out.update(rgb_max_rel=None, pixels=None)
    ok_phys = out["x_max_abs"] < 1e-5 and "state_machine_mismatch_at_step" not in out
    if expects_contact:
        ok_phys = ok_phys and out["mesh_contact"]
        if ph.self_collision and not ro.use_pusher and ro.ob_shape == "sloth":
            ok_phys = ok_phys and out["particles_with_candidates"] > 0
    ok_img = (not render) or (out["hard_rgb_mismatch_pixels"] == 0 and out["hard_depth_mismatch_pixels"] == 0
                              and out["threshold_flip_pixels"] <= 1e-4 * out["pixels"] and out["median_depth_crossing_pixels"] <= 1e-4 * out["pixels"])
    out["gates"] = {"x_max_abs": 1e-5, "threshold_flip_fraction": 1e-4, "median_depth_crossing_fraction": 1e-4, "hard_mismatch_pixels": 0,
                    "rgb": "|d| <= 1e-5 + 1e-4 |ref|", "depth": "|d| <= 1e-4 |ref|"}
    out["passed"] = bool(ok_phys and ok_img)
    out["seconds"] = time.perf_counter() - t_start
    del ro
    return out
