"""Every legal pair of flavours of ONE scene, on the device: flavours the pure selection function puts into the same
`sum_class` (csrc/physics_flavour.h: chains and the finishers' place — a launch of their own or the head of the next substep's — are
work SPLITS of the same arithmetic in the same order) end in the SAME BITS; flavours of different classes (queries in place vs listed,
the resident launch's eight partial force sums, a server unit's fixed trees, 16 vs 64 lanes over a candidate list) agree within a stated
tolerance.  The scene: two sheets squeezed between closing fingers (tests/test_contact_flavours_gpu.py: every particle under the pads has a
live self-collision candidate AND is inside a finger's margin — both kinds of finishing work in every substep), 36 substeps, where the
float64 shadow of the oracle agrees with its float32 run to 3e-8: any flavour's error against the oracle is round-off, not a flipped
contact decision.  The reference has one flavour (spring_mass_warp.py:823-943)."""
import itertools

import numpy as np
import pytest

from test_contact_flavours_gpu import _sheets_between_fingers, _t
from util_physics import far_apart, hip_env, oracle_env

pytestmark = pytest.mark.gpu
N_ENV, N_SUB = 9, 36
TOL_ACROSS_CLASSES = 1e-6    # m, positions after 36 substeps in full contact (each flavour is within ~3e-8 of the oracle here)

VARIANTS = {
    # name: (environment at create, pf, resident)
    "large layout, deferred, finishers at the head (pf)": (dict(R2S_LAYOUT="256", R2S_RESIDENT="0", R2S_MESH_DEFER="1"), True, None),
    "large layout, deferred, two launches": (dict(R2S_LAYOUT="256", R2S_RESIDENT="0", R2S_MESH_DEFER="1"), False, None),
    "large layout, deferred, pf, two chains": (dict(R2S_LAYOUT="256", R2S_RESIDENT="0", R2S_MESH_DEFER="1", R2S_CHAINS="2"), True, None),
    "large layout, queries in place + k_self_finish": (dict(R2S_LAYOUT="256", R2S_RESIDENT="0", R2S_MESH_DEFER="0"), True, None),
    "small layout, resident launch, answering servers": (dict(R2S_RES_SELF_SRV="2"), True, True),
    "small layout, resident launch, queries in place": (dict(R2S_RES_SELF_SRV="0", R2S_MESH_DEFER="0"), True, True),
    "small layout, per-substep kernels, deferred": (dict(R2S_MESH_DEFER="1"), True, False),
    "small layout, per-substep kernels, deferred, two chains": (dict(R2S_MESH_DEFER="1", R2S_CHAINS="2"), True, False),
}


def _run(name, monkeypatch):
    import torch

    env, pf, resident = VARIANTS[name]
    for k in ("R2S_LAYOUT", "R2S_RESIDENT", "R2S_MESH_DEFER", "R2S_CHAINS", "R2S_RES_SELF_SRV"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ob, nA, fingers, (interp, centers, dv, om) = _sheets_between_fingers(N_SUB)
    far = far_apart(ob, nA)
    h = hip_env(far, n_env=N_ENV, num_substeps=N_SUB, dynamic_meshes=fingers, self_collision=True)
    h.set_pf(pf)
    if resident is not None:
        h.set_resident(resident)
    h.set_state(torch.from_numpy(ob["points"])[None].repeat(N_ENV, 1, 1))
    h.update_collision_graph()
    rep = lambda a: _t(a)[None].repeat(N_ENV, *([1] * a.ndim))  # noqa: E731
    h.set_mesh_interactive(rep(interp), rep(centers), rep(dv), rep(om))
    h.step()
    fl = h.last_flavour()
    out = (h.x.cpu().numpy().copy(), h.v.cpu().numpy().copy(), fl)
    h.close()
    return out


def test_flavours_of_one_class_are_bit_identical_and_all_agree_within_round_off(monkeypatch):
    from util_parity import record

    runs = {name: _run(name, monkeypatch) for name in VARIANTS}
    ob, nA, fingers, (interp, centers, dv, om) = _sheets_between_fingers(N_SUB)
    o = oracle_env(far_apart(ob, nA), num_substeps=N_SUB, dynamic_meshes=fingers, self_collision=True)
    o.x[:] = ob["points"]
    assert o.update_collision_graph() > 0
    o.set_mesh_interactive(interp, centers, dv, om)
    o.step()
    kernels = {name: r[2]["kernel"] for name, r in runs.items()}
    classes = {name: r[2]["sum_class"] for name, r in runs.items()}
    # the variants really are different flavours ...
    assert "k_substep_pf<256,1024,true,1>" in kernels["large layout, deferred, finishers at the head (pf)"], kernels
    assert kernels["large layout, deferred, two launches"] == "k_substep<256,1024,true,1> + k_contact_finish", kernels
    assert runs["large layout, deferred, pf, two chains"][2]["chains"] == 2
    assert kernels["large layout, queries in place + k_self_finish"] == "k_substep<256,1024,true,1> + k_self_finish", kernels
    assert "query-server workgroups" in kernels["small layout, resident launch, answering servers"] and "a request per substep" in kernels["small layout, resident launch, answering servers"], kernels
    assert kernels["small layout, resident launch, queries in place"] == "k_steps_resident<512,true,1>", kernels
    assert kernels["small layout, per-substep kernels, deferred"] in ("k_steps_resident<512,true,1> x 1 substep + k_contact_finish", "k_substep<64,512,true,1> + k_contact_finish"), kernels
    assert runs["small layout, per-substep kernels, deferred, two chains"][2]["chains"] == 2
    # ... of which these are work splits of one another:
    same = [("large layout, deferred, finishers at the head (pf)", "large layout, deferred, two launches"),
            ("large layout, deferred, finishers at the head (pf)", "large layout, deferred, pf, two chains"),
            ("small layout, per-substep kernels, deferred", "small layout, per-substep kernels, deferred, two chains")]
    for a, b in same:
        assert classes[a] == classes[b], (a, b, classes)
    worst_across, table = 0.0, []
    for a, b in itertools.combinations(VARIANTS, 2):
        dx = float(np.abs(runs[a][0] - runs[b][0]).max())
        table.append((a, b, classes[a] == classes[b], dx))
        if classes[a] == classes[b]:
            assert np.array_equal(runs[a][0], runs[b][0]) and np.array_equal(runs[a][1], runs[b][1]), (a, b, dx)
        else:
            worst_across = max(worst_across, dx)
            assert dx < TOL_ACROSS_CLASSES, (a, b, dx)
    assert len({c for c in classes.values()}) == 5, classes
    worst_oracle = max(float(np.abs(r[0][e] - o.x).max()) for r in runs.values() for e in range(N_ENV))
    assert worst_oracle < 1e-5
    # environments of a batch are the same scene: one flavour, one result per batch
    for name, r in runs.items():
        assert all(np.array_equal(r[0][0], r[0][e]) for e in range(1, N_ENV)), name
    record("flavour_pairs_two_sheets", flavours=len(VARIANTS), classes=len(set(classes.values())), pairs=len(table), bit_identical_pairs=sum(t[2] for t in table),
           worst_across_classes=worst_across, worst_vs_oracle=worst_oracle, tol_across_classes=TOL_ACROSS_CLASSES)
