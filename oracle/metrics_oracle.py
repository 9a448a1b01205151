"""CPU restatement (numpy) of the evaluation scripts' success predicates — TEST INFRASTRUCTURE ONLY (tests/ import it as
the checker of csrc/metrics.hip).  Pinned: tests/test_metrics_oracle.py checks it against tests/golden/success_*.npz,
produced by the reference's own calculate_success_rope.py / calculate_success_T.py (tests/golden/make_success_golden.py).
The sloth predicate (calculate_success_sloth.py:152-168) needs open3d, which is absent: its point-in-OBB test is restated
from open3d's documented behaviour and is UNPINNED."""
import numpy as np


def plane_crossings(x, springs, bbox_min, bbox_max, eps=1e-12):
    """calculate_success_rope.py:40-129 -> (y_min_count, y_max_count)."""
    V = np.asarray(x, dtype=float)
    E = np.asarray(springs, dtype=int)
    p0, p1 = V[E[:, 0]], V[E[:, 1]]
    x_min, y_min, z_min = np.asarray(bbox_min, float)
    x_max, y_max, z_max = np.asarray(bbox_max, float)

    def hits(y_plane):
        y0, y1 = p0[:, 1], p1[:, 1]
        dy = y1 - y0
        parallel = np.abs(dy) <= eps
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(parallel, 0.0, (y_plane - y0) / np.where(parallel, 1.0, dy))
        on = (~parallel) & (t >= -eps) & (t <= 1.0 + eps)
        xi = p0[:, 0] + t * (p1[:, 0] - p0[:, 0])
        zi = p0[:, 2] + t * (p1[:, 2] - p0[:, 2])
        inside = (xi >= x_min - eps) & (xi <= x_max + eps) & (zi >= z_min - eps) & (zi <= z_max + eps)
        cop = parallel & (np.abs(y0 - y_plane) <= eps)
        e0 = (p0[:, 0] >= x_min - eps) & (p0[:, 0] <= x_max + eps) & (p0[:, 2] >= z_min - eps) & (p0[:, 2] <= z_max + eps)
        e1 = (p1[:, 0] >= x_min - eps) & (p1[:, 0] <= x_max + eps) & (p1[:, 2] >= z_min - eps) & (p1[:, 2] <= z_max + eps)
        return (on & inside) | (cop & (e0 | e1))

    return int(np.count_nonzero(hits(y_min))), int(np.count_nonzero(hits(y_max)))


ROPE_CLIP = (np.array([0.62 - 0.035 / 2, 0.05 - 0.035 / 2, 0.0]), np.array([0.62 + 0.035 / 2, 0.05 + 0.035 / 2, 0.03]))  # :150-160


def rope_routed(x, springs):
    lo, hi = plane_crossings(x, springs, *ROPE_CLIP)
    return lo >= 100 and hi >= 100                                                                    # :166


def pusht_mse(x, x_target):
    """calculate_success_T.py:26-27 (success: < 0.002)."""
    return ((np.asarray(x, np.float32) - np.asarray(x_target, np.float32)) ** 2).sum(1).mean()


def points_in_obb(x, center, R, half_extent):
    """Points with |R^T (p - c)| <= half_extent per axis (the sloth script scales the box by 1.05 first, :160)."""
    local = (np.asarray(x, float) - np.asarray(center, float)) @ np.asarray(R, float)
    return int(np.count_nonzero(np.all(np.abs(local) <= np.asarray(half_extent, float), axis=1)))
