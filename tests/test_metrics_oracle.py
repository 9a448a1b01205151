"""The numpy restatement of the success predicates reproduces the fixtures produced by the reference's own scripts."""
import os

import numpy as np

from oracle import metrics_oracle as mo

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases(path):
    d = np.load(path)
    names = sorted({k.split("__")[0] for k in d.files if "__" in k})
    return d, names


def test_rope_plane_crossings_match_the_reference_script():
    d, names = _cases(os.path.join(G, "success_rope.npz"))
    assert {"through", "beside", "degenerate"} <= set(names)
    for n in names:
        got = mo.plane_crossings(d[f"{n}__x"], d[f"{n}__springs"], d[f"{n}__bbox_min"], d[f"{n}__bbox_max"])
        assert got == (int(d[f"{n}__y_min_count"]), int(d[f"{n}__y_max_count"])), n
        if n != "degenerate":
            assert mo.rope_routed(d[f"{n}__x"], d[f"{n}__springs"]) == bool(d[f"{n}__routed"]), n
    assert int(d["through__y_min_count"]) >= 100 and int(d["degenerate__y_min_count"]) == 4


def test_pusht_mse_matches_the_reference_script():
    d, names = _cases(os.path.join(G, "success_T.npz"))
    for n in names:
        mse = mo.pusht_mse(d[f"{n}__x"], d["target"])
        assert abs(float(mse) - float(d[f"{n}__mse"])) <= 1e-6 * float(d[f"{n}__mse"]) + 1e-12, n
        assert bool(mse < 0.002) == bool(d[f"{n}__success"]), n
    assert bool(d["edge_in__success"]) and not bool(d["edge_out__success"])


def test_points_in_obb_known_answer():
    R = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])          # box x-axis along world +y
    c, h = np.array([1.0, 2.0, 0.5]), np.array([0.5, 0.1, 0.25])
    pts = np.array([[1.0, 2.0, 0.5], [1.0, 2.49, 0.5], [1.0, 2.51, 0.5], [1.09, 2.0, 0.74], [1.11, 2.0, 0.5], [1.0, 2.0, 0.76], [1.0, 1.5, 0.25]])
    assert mo.points_in_obb(pts, c, R, h) == 4
