// Per-environment wrist camera on the device: GSRenderer.render_wrist's eef2c -> w2c (gs_renderer.py:966-985) followed by
// setup_camera (transform_utils.py:7-31), one lane per environment.  See include/r2s_camera.h.
#include "r2s_common.h"
#include "../../include/r2s_camera.h"

namespace {

struct CamConst {
    double eef2c[16];
    float proj[16]; // opengl_proj, row-major, rounded to float32 like torch.tensor(...).float()
};

// inverse of a 4x4 whose last row is (0, 0, 0, 1): [A t; 0 1]^-1 = [A^-1, -A^-1 t; 0 1], A^-1 by cofactors (float64)
__device__ __forceinline__ void inv_affine(const double* m, double* o)
{
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const double det = a * A + b * B + c * C;
    const double r = 1.0 / det;
    o[0] = A * r; o[1] = -(b * i - c * h) * r; o[2] = (b * f - c * e) * r;
    o[4] = B * r; o[5] = (a * i - c * g) * r; o[6] = -(a * f - c * d) * r;
    o[8] = C * r; o[9] = -(a * h - b * g) * r; o[10] = (a * e - b * d) * r;
    const double tx = m[3], ty = m[7], tz = m[11];
    o[3] = -(o[0] * tx + o[1] * ty + o[2] * tz);
    o[7] = -(o[4] * tx + o[5] * ty + o[6] * tz);
    o[11] = -(o[8] * tx + o[9] * ty + o[10] * tz);
    o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = 1.0;
}

#pragma clang fp contract(off)
__global__ void k_wrist_camera(int E, const float* __restrict__ eef_xyz, const float* __restrict__ eef_rot, const CamConst cc,
                               float* __restrict__ view, float* __restrict__ proj, float* __restrict__ campos)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    // e2b = [eef_rot | eef_xyz], float32 values (gs_renderer.py:975-978)
    double e2b[16];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) e2b[4 * r + c] = (double)eef_rot[9 * e + 3 * r + c];
        e2b[4 * r + 3] = (double)eef_xyz[3 * e + r];
    }
    e2b[12] = e2b[13] = e2b[14] = 0.0; e2b[15] = 1.0;
    double b2eef[16];
    inv_affine(e2b, b2eef);                                   // np.linalg.inv of a float32 matrix: the result is float32
    for (int k = 0; k < 16; ++k) b2eef[k] = (double)(float)b2eef[k];
    // w2c = eef2c @ b2eef @ eye (float64), then torch.tensor(w2c).float() (transform_utils.py:9)
    float w2c[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += cc.eef2c[4 * r + k] * b2eef[4 * k + c];
            w2c[4 * r + c] = (float)s;
        }
    // cam_center = torch.inverse(w2c)[:3, 3] (:10)
    double w2cd[16], c2w[16];
    for (int k = 0; k < 16; ++k) w2cd[k] = (double)w2c[k];
    inv_affine(w2cd, c2w);
    campos[3 * e] = (float)c2w[3]; campos[3 * e + 1] = (float)c2w[7]; campos[3 * e + 2] = (float)c2w[11];
    // viewmatrix = w2c^T (:11); projmatrix = w2c^T . opengl_proj^T (:12-16), float32, k = 0..3
    float* v = view + 16 * (size_t)e;
    float* p = proj + 16 * (size_t)e;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            v[4 * i + j] = w2c[4 * j + i];
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += w2c[4 * k + i] * cc.proj[4 * j + k];
            p[4 * i + j] = s;
        }
}

// kornia.geometry.conversions.rotation_matrix_to_quaternion (>= 0.7: w, x, y, z) as r2s_hip.rollout.rotation_matrix_to_quaternion states it,
// float32, one lane per matrix: the published branch scheme — trace > 0: sq = 2 sqrt(trace + 1 + eps), else the largest diagonal entry
// picks the component the square root computes — the sign the scheme yields, not canonicalised; a division by max(sq, FLT_MIN).
__global__ void k_rot_to_quat(int n, const float* __restrict__ R, float eps, float* __restrict__ q)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const float* m = R + 9 * (size_t)e;
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    const float tr = m00 + m11 + m22;
    const float tiny = 1.17549435e-38f;
    float w, x, y, z;
    if (tr > 0.f) {
        const float s = sqrtf(tr + 1.0f + eps) * 2.0f, d = fmaxf(s, tiny);
        w = 0.25f * s; x = (m21 - m12) / d; y = (m02 - m20) / d; z = (m10 - m01) / d;
    } else if (m00 > m11 && m00 > m22) {
        const float s = sqrtf(1.0f + m00 - m11 - m22 + eps) * 2.0f, d = fmaxf(s, tiny);
        w = (m21 - m12) / d; x = 0.25f * s; y = (m01 + m10) / d; z = (m02 + m20) / d;
    } else if (m11 > m22) {
        const float s = sqrtf(1.0f + m11 - m00 - m22 + eps) * 2.0f, d = fmaxf(s, tiny);
        w = (m02 - m20) / d; x = (m01 + m10) / d; y = 0.25f * s; z = (m12 + m21) / d;
    } else {
        const float s = sqrtf(1.0f + m22 - m00 - m11 + eps) * 2.0f, d = fmaxf(s, tiny);
        w = (m10 - m01) / d; x = (m02 + m20) / d; y = (m12 + m21) / d; z = 0.25f * s;
    }
    float* o = q + 4 * (size_t)e;
    o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}

} // namespace

extern "C" int r2s_rot_to_quat(int32_t n, const float* rot, float* quat, r2s_stream_t stream)
{
    if (n <= 0 || !rot || !quat) return R2S_ERR_INVALID;
    hipLaunchKernelGGL(k_rot_to_quat, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, n, rot, 1e-8f, quat);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

extern "C" int r2s_wrist_camera(int32_t n_env, const float* eef_xyz, const float* eef_rot, const double* eef2c, const double* K, int32_t width,
                                int32_t height, double near_plane, double far_plane, float* viewmatrix, float* projmatrix, float* campos,
                                r2s_stream_t stream)
{
    if (n_env <= 0 || !eef_xyz || !eef_rot || !eef2c || !K || width <= 0 || height <= 0 || !viewmatrix || !projmatrix || !campos ||
        !(far_plane > near_plane))
        return R2S_ERR_INVALID;
    CamConst cc;
    for (int k = 0; k < 16; ++k) { cc.eef2c[k] = eef2c[k]; cc.proj[k] = 0.f; }
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5], w = width, h = height, n = near_plane, f = far_plane;
    cc.proj[0] = (float)(2.0 * fx / w); cc.proj[2] = (float)(-(w - 2.0 * cx) / w);
    cc.proj[5] = (float)(2.0 * fy / h); cc.proj[6] = (float)(-(h - 2.0 * cy) / h);
    cc.proj[10] = (float)(f / (f - n)); cc.proj[11] = (float)(-(f * n) / (f - n));
    cc.proj[14] = 1.f;
    hipLaunchKernelGGL(k_wrist_camera, dim3((n_env + 63) / 64), dim3(64), 0, (hipStream_t)stream, n_env, eef_xyz, eef_rot, cc, viewmatrix, projmatrix,
                       campos);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}
