/* r2s_metrics.h — task-success predicates of the evaluation scripts, evaluated on the device for a batch of
 * environments, so that the per-environment success flag of the final metric all-gather never
 * needs the particle state on the host.  C ABI; every pointer marked "dev" is device memory of the current device;
 * results are written to device memory, nothing synchronises.
 *
 * Replaces (reference, offline on pickled states):
 *   experiments/utils/calculate_success_rope.py:40-129   count_xz_plane_intersections  -> r2s_metric_plane_crossings
 *   experiments/utils/calculate_success_T.py:17-29       is_pusht_success (mse)        -> r2s_metric_mse
 *   experiments/utils/calculate_success_sloth.py:152-168 points inside the scaled OBB  -> r2s_metric_points_in_obb
 */
#ifndef R2S_METRICS_H
#define R2S_METRICS_H
#include <stdint.h>
#include "r2s_raster.h" /* r2s_stream_t, error codes */
#ifdef __cplusplus
extern "C" {
#endif

/* Number of spring segments crossing the planes y = bbox_min[1] and y = bbox_max[1] inside the x-z rectangle of the
 * box, with the script's tolerance rules (float64, eps; coplanar segments count when an endpoint is inside).
 * x: dev float32 [n_env, n_points, 3]; springs: dev int32 [n_springs, 2]; counts: dev int32 [n_env, 2] = (y_min, y_max). */
int r2s_metric_plane_crossings(int32_t n_env, int32_t n_points, const float* x, int32_t n_springs, const int32_t* springs,
                               const double* bbox_min, const double* bbox_max, double eps, int32_t* counts, r2s_stream_t stream);

/* mean over points of the squared distance to a target configuration: ((x - x_target)**2).sum(1).mean().
 * x: dev float32 [n_env, n_points, 3]; x_target: dev float32 [n_points, 3]; mse: dev float64 [n_env]. */
int r2s_metric_mse(int32_t n_env, int32_t n_points, const float* x, const float* x_target, double* mse, r2s_stream_t stream);

/* Number of points inside an oriented box: |R^T (p - center)| <= half_extent per axis (open3d
 * OrientedBoundingBox::GetPointIndicesWithinBoundingBox).  center [3], R [9] row-major (columns = box axes),
 * half_extent [3]: HOST float64.  count: dev int32 [n_env]. */
int r2s_metric_points_in_obb(int32_t n_env, int32_t n_points, const float* x, const double* center, const double* R,
                             const double* half_extent, int32_t* count, r2s_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* R2S_METRICS_H */
