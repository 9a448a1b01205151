/*
 * r2s_skinning.h — C ABI of the linear-blend-skinning step that sits between the two hot paths
 * (SURVEY.md §8f row f1): Gaussians follow the PhysTwin particles ("bones").
 *
 * Replaces the reference function
 *   interpolate_motions(bones, motions, relations, xyz, rot=None, quat=None, weights, weights_indices, device)
 *   sim/utils/gs/transform_utils.py:58-212
 * as called once per env step by GSRenderer.update_rendervar (sim/renderer/gs_renderer.py:738-747: quat=None,
 * precomputed relations / weights).  The reference builds it from ~40 torch ops incl. a batched 3x3 SVD, matrix_rank
 * and det; here it is two HIP kernels (per-bone Kabsch rotation in float64, per-point blend).
 *
 * Conventions as in r2s_raster.h: device pointers for per-step data, host pointers for the one-time topology,
 * work enqueued on `stream`, int status (R2S_OK or negative R2S_ERR_*).  One handle serves a batch of environments
 * that share relations / weights (same PhysTwin, same Gaussian template).
 */
#ifndef R2S_SKINNING_H
#define R2S_SKINNING_H

#include <stddef.h>
#include <stdint.h>
#include "r2s_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct R2SSkin R2SSkin; /* opaque */

/* relations: host [n_bones, k_rel] neighbour bones (knn_relations, gs_renderer.py:195-200);
 * weights / weights_indices: host [n_points, k_wgt] (knn_weights, gs_renderer.py:202-211). */
int r2s_skin_create(int32_t n_bones, int32_t k_rel, const int32_t* relations, int32_t n_points, int32_t k_wgt,
                    const float* weights, const int32_t* weights_indices, R2SSkin** out, r2s_stream_t stream);
void r2s_skin_destroy(R2SSkin* h);

/* xyz_out[e, p] = sum_j w_pj ( R_bj (xyz[e, p] - bones[e, bj]) + motions[e, bj] + bones[e, bj] ),  transform_utils.py:178-189,
 * R_b = proper rotation closest to F_b = sum_k (a'_k)(a_k)^T over the bone's k_rel neighbours (:79-114).  If any bone of an
 * environment has a rank < 2 fit, every rotation of that environment is the identity, as in the reference (:157-162).
 * bones, motions: device [n_env, n_bones, 3]; xyz, xyz_out: device [n_env, n_points, 3] (may alias). */
int r2s_skin_interpolate_motions(R2SSkin* h, int32_t n_env, const float* bones, const float* motions, const float* xyz,
                                 float* xyz_out, r2s_stream_t stream);

/* The same with explicit environment strides (in floats) for xyz and xyz_out: the points of environment e start at
 * xyz + e * xyz_env_stride.  Lets the skinned Gaussians live inside the rasteriser's per-environment set (object splats
 * followed by the table / robot scan) and be updated in place, instead of being copied there (gs_renderer.py:886-903 builds
 * the scene with torch.cat every step). */
int r2s_skin_interpolate_motions_strided(R2SSkin* h, int32_t n_env, const float* bones, const float* motions, const float* xyz,
                                         int64_t xyz_env_stride, float* xyz_out, int64_t out_env_stride, r2s_stream_t stream);

/* interpolate_motions(quat=...) (transform_utils.py:197-210; the simulator passes quat=None, offline tools rotate the splats):
 * with the bone rotations of the LAST r2s_skin_interpolate_motions* call, quat_out = normalise(sum_j w_j q(R_bj)) (x) quat, the
 * Hamilton product with the blended bone rotation first; q(R) is kornia's rotation_matrix_to_quaternion, normalised.  quat /
 * quat_out: device [n_env, n_points, 4] (w, x, y, z) with the given environment strides in floats; may alias. */
int r2s_skin_rotate_quats(R2SSkin* h, int32_t n_env, const float* quat, int64_t quat_env_stride, float* quat_out, int64_t out_env_stride,
                          r2s_stream_t stream);

/* Device pointer to the per-bone rotations of the last call, [n_env, n_bones, 9] row-major (parity taps), and the
 * per-environment "rank-deficient fit -> identity" flags [n_env]. */
int r2s_skin_debug(R2SSkin* h, const float** rotations, const int32_t** identity_flags);

#ifdef __cplusplus
}
#endif
#endif /* R2S_SKINNING_H */
