/*
 * r2s_robot.h — C ABI of the articulated-robot Gaussian placement (second half: scene assembly).
 *
 * Replaces, for a batch of environments,
 *   transform_gs_xarm_gripper / transform_gs_xarm_pusher    sim/utils/robot/robot_pc_transformations.py:12-55, :94-133
 *   RobotPcSampler.transform_gs_torch + quat_mult_torch      sim/utils/robot/robot_pc_sampler.py:17-24, :118-161
 * as called once per env step by GSRenderer.update_rendervar (sim/renderer/gs_renderer.py:886-921): every Gaussian of the
 * "table" scan carries a link id (total_mask); the Gaussians of the listed links follow their link rigidly,
 *   mat_l   = (link_pose_l(qpos) @ offset_l) @ inv(link_pose_l(base_qpos) @ offset_l)
 *   mean'   = mean @ mat_l[:3,:3]^T + mat_l[:3,3]
 *   quat'   = quat_mult(rotation_matrix_to_quaternion(mat_l[:3,:3]), normalize(quat))
 * and the renderer normalises every rotation afterwards (:906).  Forward kinematics (SAPIEN / pinocchio) stays with the
 * caller: the per-link poses are an INPUT.  The reference runs ~15 boolean-mask gathers, 4x4 inverses and scatters plus three
 * torch.cat of the whole scene per env step and environment; here two kernels write the moved means / rotations straight
 * into the rasteriser's per-environment Gaussian set.
 *
 * Conventions as in r2s_raster.h: host pointers for the one-time description, device pointers for per-step data, work
 * enqueued on `stream`, int status (R2S_OK or negative R2S_ERR_*).
 */
#ifndef R2S_ROBOT_H
#define R2S_ROBOT_H

#include <stddef.h>
#include <stdint.h>
#include "r2s_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct R2SRobotGS R2SRobotGS; /* opaque */

/* n_links: links of the robot (len(sapien_robot.get_links()): 18 gripper arm, 11 pusher arm); link_listed[n_links] != 0 for
 * the links whose Gaussians move (link_id_list, robot_pc_transformations.py:33 / :113); offsets: host float64 [n_links,4,4]
 * (RobotPcSampler.offsets, the URDF collision origins); link_pose_base: host float32 [n_links,4,4] (FK of base_qpos).
 * n_gauss Gaussians of the scan: rest means [n,3], rotations [n,4] as stored (unnormalised), total_mask [n] (link id; ids
 * that are negative, >= n_links or not listed stay where they are). */
int r2s_robot_gs_create(int32_t n_links, const int32_t* link_listed, const double* offsets, const float* link_pose_base,
                        int32_t n_gauss, const float* means, const float* rotations, const int32_t* total_mask,
                        R2SRobotGS** out, r2s_stream_t stream);
void r2s_robot_gs_destroy(R2SRobotGS* h);

/* One env step.  link_pose: device float32 [n_env, n_links, 4, 4] (FK of the current qpos per environment).
 * means_out / rotations_out: device, environment e's scan starts at means_out + e * means_env_stride (floats) and
 * rotations_out + e * rotations_env_stride.  write_static != 0 writes all n_gauss entries (the static ones with their rest
 * values: first call / after the caller overwrote the set); 0 writes only the Gaussians of the listed links.
 * normalize != 0 applies the renderer's final F.normalize (gs_renderer.py:906) to every rotation written. */
int r2s_robot_gs_transform(R2SRobotGS* h, int32_t n_env, const float* link_pose, float* means_out, int64_t means_env_stride,
                           float* rotations_out, int64_t rotations_env_stride, int32_t normalize, int32_t write_static,
                           r2s_stream_t stream);

/* Parity tap: device pointer to the per-(env, link) records of the last call, [n_env, n_links, 16] floats =
 * mat[:3,:4] row-major (12) + quaternion (w, x, y, z). */
int r2s_robot_gs_debug(R2SRobotGS* h, const float** link_records);

#ifdef __cplusplus
}
#endif
#endif /* R2S_ROBOT_H */
