/*
 * r2s_camera.h — C ABI of the per-environment wrist camera (the caller of the rasteriser).
 *
 * Replaces, for a batch of environments and without leaving the device,
 *   GSRenderer.render_wrist         sim/renderer/gs_renderer.py:953-1000   (eef2c -> w2c from the gripper pose, :966-985)
 *   setup_camera                    sim/utils/gs/transform_utils.py:7-31   (viewmatrix, projmatrix, campos of the settings)
 * which the reference evaluates every frame on the host: e2b = [eef_rot | eef_xyz] (float32), b2eef = inv(e2b) (numpy,
 * float32), w2c = eef2c @ b2eef (float64, eef2c is the float64 calibration), cast to float32, then
 *   viewmatrix = w2c^T (memory = column-major w2c), campos = inv(w2c)[:3,3],
 *   projmatrix = w2c^T . opengl_proj^T,   opengl_proj = [[2fx/w,0,-(w-2cx)/w,0],[0,2fy/h,-(h-2cy)/h,0],[0,0,far/(far-near),-far.near/(far-near)],[0,0,1,0]]
 * with a 4x4 H2D upload, a torch.inverse and a bmm per frame and environment.  Here one lane per environment writes the three
 * arrays straight into the buffers the rasteriser's prepared frames point at (r2s_raster.h: R2SRasterFrame.viewmatrix /
 * projmatrix / cam_pos), so the wrist view follows each environment's gripper with no host work per step.
 *
 * Arithmetic: the two 4x4 inverses are evaluated in float64 by cofactors and rounded to float32 where the reference holds
 * float32 values (its LAPACK / rocSOLVER inverses differ from the exact inverse in the last bits; parity is by tolerance, see
 * tests/test_wrist_camera_gpu.py); the projection product is float32 in the reference's k = 0..3 order.
 */
#ifndef R2S_CAMERA_H
#define R2S_CAMERA_H

#include <stddef.h>
#include <stdint.h>
#include "r2s_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

/* eef_xyz [n_env,3], eef_rot [n_env,3,3] (row-major rotation of gripper 0 of each environment): DEVICE float32.
 * eef2c: HOST float64 [16], row-major 4x4 (metadata_wrist['eef2c'], gs_renderer.py:181-193); K: HOST float64 [9] intrinsics.
 * Outputs, DEVICE float32: viewmatrix [n_env,16], projmatrix [n_env,16], campos [n_env,3] — the fields of
 * GaussianRasterizationSettings (diff_gaussian_rasterization/__init__.py:135-147) per environment. */
int r2s_wrist_camera(int32_t n_env, const float* eef_xyz, const float* eef_rot, const double* eef2c, const double* K, int32_t width,
                     int32_t height, double near_plane, double far_plane, float* viewmatrix, float* projmatrix, float* campos,
                     r2s_stream_t stream);

/* obs['robot']['eef_quat'] of BaseEnv.get_obs (sim/envs/env.py:62-66; phystwin.py:117: kornia rotation_matrix_to_quaternion of the current
 * end-effector rotation) for a batch, on the device in ONE launch: rot [n,3,3] row-major -> quat [n,4] (w, x, y, z), DEVICE float32.
 * (The torch restatement of the same branch scheme is ~40 small launches per observation: 0.2 ms of host time in a closed loop.) */
int r2s_rot_to_quat(int32_t n, const float* rot, float* quat, r2s_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* R2S_CAMERA_H */
