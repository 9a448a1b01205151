// Articulated-robot Gaussians follow their links (row f1 of SURVEY.md §8f, scene assembly) — see include/r2s_robot.h.
//
//   k_link_records  one lane per (env, link): mat = (pose @ offset) @ inv(base @ offset) in float32 like the reference's torch
//                   ops (robot_pc_sampler.py:139-149; the inverse of the constant base pose is taken once, on the host), and
//                   kornia's rotation_matrix_to_quaternion of its rotation block (restated from kornia's published source:
//                   trace / largest-diagonal branches, eps = 1e-8, (w, x, y, z)).  64-byte record per (env, link).
//   k_robot_gs      one lane per (env, Gaussian) of the scan, Gaussians sorted by link so that a wavefront reads one or two
//                   records (scalar-cache hits): mean' = mean R^T + t (:150), quat' = quat_mult(q_link, normalize(quat))
//                   (:17-24, :153), optional final normalisation (gs_renderer.py:906); written straight into the raster's
//                   per-environment set (env stride given by the caller).  28 B read + 28 B written per moved Gaussian.

#include "r2s_common.h"
#include "../../include/r2s_robot.h"
#include <algorithm>
#include <cmath>
#include <vector>

namespace {
#pragma clang fp contract(off)

struct LinkRec {
    float m[12]; // mat[:3,:4] row-major
    float q[4];  // w, x, y, z
};

__global__ void __launch_bounds__(64) k_link_records(int E, int L, const float* __restrict__ pose, const float* __restrict__ offset,
                                                     const float* __restrict__ base_inv, const int* __restrict__ listed, LinkRec* __restrict__ rec)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * L) return;
    const int l = t % L;
    LinkRec r;
    if (!listed[l]) { // never read
        for (int k = 0; k < 12; ++k) r.m[k] = (k % 5 == 0) ? 1.f : 0.f;
        r.q[0] = 1.f; r.q[1] = r.q[2] = r.q[3] = 0.f;
        rec[t] = r;
        return;
    }
    const float* P = pose + (size_t)t * 16;
    const float* O = offset + (size_t)l * 16;
    const float* B = base_inv + (size_t)l * 16;
    float A[16], M[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) A[4 * i + j] = ((P[4 * i] * O[j] + P[4 * i + 1] * O[4 + j]) + P[4 * i + 2] * O[8 + j]) + P[4 * i + 3] * O[12 + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) M[4 * i + j] = ((A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j]) + A[4 * i + 2] * B[8 + j]) + A[4 * i + 3] * B[12 + j];
    for (int k = 0; k < 12; ++k) r.m[k] = M[k];
    const float m00 = M[0], m01 = M[1], m02 = M[2], m10 = M[4], m11 = M[5], m12 = M[6], m20 = M[8], m21 = M[9], m22 = M[10];
    const float eps = 1e-8f, tiny = 1.17549435e-38f;
    const float trace = (m00 + m11) + m22;
    float sq, qw, qx, qy, qz;
    if (trace > 0.f) {
        sq = sqrtf(trace + 1.0f + eps) * 2.0f;
        const float d = fmaxf(sq, tiny);
        qw = 0.25f * sq; qx = (m21 - m12) / d; qy = (m02 - m20) / d; qz = (m10 - m01) / d;
    } else if (m00 > m11 && m00 > m22) {
        sq = sqrtf(1.0f + m00 - m11 - m22 + eps) * 2.0f;
        const float d = fmaxf(sq, tiny);
        qw = (m21 - m12) / d; qx = 0.25f * sq; qy = (m01 + m10) / d; qz = (m02 + m20) / d;
    } else if (m11 > m22) {
        sq = sqrtf(1.0f + m11 - m00 - m22 + eps) * 2.0f;
        const float d = fmaxf(sq, tiny);
        qw = (m02 - m20) / d; qx = (m01 + m10) / d; qy = 0.25f * sq; qz = (m12 + m21) / d;
    } else {
        sq = sqrtf(1.0f + m22 - m00 - m11 + eps) * 2.0f;
        const float d = fmaxf(sq, tiny);
        qw = (m10 - m01) / d; qx = (m02 + m20) / d; qy = (m12 + m21) / d; qz = 0.25f * sq;
    }
    r.q[0] = qw; r.q[1] = qx; r.q[2] = qy; r.q[3] = qz;
    rec[t] = r;
}

__device__ __forceinline__ float4 f_normalize(float4 q) // torch.nn.functional.normalize(dim=-1): q / max(||q||, 1e-12)
{
    const float n = fmaxf(sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w), 1e-12f);
    return make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
}

__global__ void __launch_bounds__(256) k_robot_gs(int n, int L, const int* __restrict__ order, const int* __restrict__ link, const float* __restrict__ means,
                                                  const float4* __restrict__ quats_n, const LinkRec* __restrict__ rec, float* __restrict__ means_out,
                                                  long long means_stride, float* __restrict__ rot_out, long long rot_stride, int normalize, int t0)
{
    const int t = t0 + blockIdx.x * blockDim.x + threadIdx.x; // t0 > 0 skips the static Gaussians (sorted first)
    const int e = blockIdx.y;
    if (t >= n) return;
    const int g = order[t], l = link[t]; // sorted by link: a wavefront sees one or two records
    float x = means[3 * (size_t)g], y = means[3 * (size_t)g + 1], z = means[3 * (size_t)g + 2];
    float4 q = quats_n[g]; // normalize(params[rotation_name]) of the whole scan, robot_pc_transformations.py:29
    if (l >= 0) {
        const LinkRec& r = rec[(size_t)e * L + l];
        const float nx = ((x * r.m[0] + y * r.m[1]) + z * r.m[2]) + r.m[3];
        const float ny = ((x * r.m[4] + y * r.m[5]) + z * r.m[6]) + r.m[7];
        const float nz = ((x * r.m[8] + y * r.m[9]) + z * r.m[10]) + r.m[11];
        x = nx; y = ny; z = nz;
        const float w1 = r.q[0], x1 = r.q[1], y1 = r.q[2], z1 = r.q[3], w2 = q.x, x2 = q.y, y2 = q.z, z2 = q.w;
        q = make_float4(w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2);
    }
    if (normalize) q = f_normalize(q);
    float* mo = means_out + (size_t)e * means_stride + 3 * (size_t)g;
    mo[0] = x; mo[1] = y; mo[2] = z;
    float* ro = rot_out + (size_t)e * rot_stride + 4 * (size_t)g;
    ro[0] = q.x; ro[1] = q.y; ro[2] = q.z; ro[3] = q.w;
}

} // namespace

struct R2SRobotGS {
    int L = 0, n = 0, n_static = 0, cap_env = 0;
    int *d_listed = nullptr, *d_order = nullptr, *d_link = nullptr;
    float *d_offset = nullptr, *d_base_inv = nullptr, *d_means = nullptr;
    float4* d_quats = nullptr;
    LinkRec* d_rec = nullptr;
};

namespace {
// inverse of a 4x4 in float64 (Gauss-Jordan with partial pivoting); the reference takes torch.linalg.inv of the float32 base
// matrix on every call (:148) — the base pose never changes, so it is inverted once here
bool invert4(const double* a, double* inv)
{
    double m[4][8];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { m[i][j] = a[4 * i + j]; m[i][4 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(m[r][c]) > std::fabs(m[p][c])) p = r;
        if (std::fabs(m[p][c]) < 1e-300) return false;
        if (p != c) for (int j = 0; j < 8; ++j) std::swap(m[p][j], m[c][j]);
        const double d = m[c][c];
        for (int j = 0; j < 8; ++j) m[c][j] /= d;
        for (int r = 0; r < 4; ++r) if (r != c) { const double f = m[r][c]; for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j]; }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) inv[4 * i + j] = m[i][4 + j];
    return true;
}
} // namespace

extern "C" {

int r2s_robot_gs_create(int32_t n_links, const int32_t* link_listed, const double* offsets, const float* link_pose_base, int32_t n_gauss,
                        const float* means, const float* rotations, const int32_t* total_mask, R2SRobotGS** out, r2s_stream_t stream_)
{
    hipStream_t s = (hipStream_t)stream_;
    if (!out || n_links <= 0 || !link_listed || !offsets || !link_pose_base || n_gauss < 0 || (n_gauss > 0 && (!means || !rotations || !total_mask)))
        return R2S_ERR_INVALID;
    R2SRobotGS* h = new (std::nothrow) R2SRobotGS();
    if (!h) return R2S_ERR_ALLOC;
    h->L = n_links; h->n = n_gauss;
    std::vector<float> off32(16 * (size_t)n_links), binv(16 * (size_t)n_links);
    std::vector<int> listed(n_links);
    for (int l = 0; l < n_links; ++l) {
        listed[l] = link_listed[l] != 0;
        float a32[16];
        for (int k = 0; k < 16; ++k) off32[16 * l + k] = (float)offsets[16 * l + k]; // .to(dtype), :143
        const float* P = link_pose_base + 16 * (size_t)l;
        const float* O = off32.data() + 16 * (size_t)l;
        for (int i = 0; i < 4; ++i) // mat_base = link_base_pose @ tf_obj_to_link in float32, :147
            for (int j = 0; j < 4; ++j) a32[4 * i + j] = ((P[4 * i] * O[j] + P[4 * i + 1] * O[4 + j]) + P[4 * i + 2] * O[8 + j]) + P[4 * i + 3] * O[12 + j];
        double a64[16], i64[16];
        for (int k = 0; k < 16; ++k) a64[k] = a32[k];
        if (!invert4(a64, i64)) { if (listed[l]) { delete h; return R2S_ERR_INVALID; } for (int k = 0; k < 16; ++k) i64[k] = (k % 5 == 0); }
        for (int k = 0; k < 16; ++k) binv[16 * l + k] = (float)i64[k];
    }
    // Gaussians sorted by link (static ones first): neighbouring lanes read the same record
    std::vector<int> order(n_gauss), link(n_gauss);
    std::vector<int> key(n_gauss);
    for (int g = 0; g < n_gauss; ++g) { const int m = total_mask[g]; key[g] = (m >= 0 && m < n_links && listed[m]) ? m : -1; order[g] = g; }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    for (int t = 0; t < n_gauss; ++t) { link[t] = key[order[t]]; h->n_static += link[t] < 0; }
    std::vector<float4> qn(n_gauss);
    for (int g = 0; g < n_gauss; ++g) { // F.normalize of the stored rotations, in float32
        const float a = rotations[4 * (size_t)g], b = rotations[4 * (size_t)g + 1], c = rotations[4 * (size_t)g + 2], d = rotations[4 * (size_t)g + 3];
        const float nrm = std::max(std::sqrt(((a * a + b * b) + c * c) + d * d), 1e-12f);
        qn[g] = make_float4(a / nrm, b / nrm, c / nrm, d / nrm);
    }
    auto fail = [&](int rc) { r2s_robot_gs_destroy(h); return rc; };
    const size_t n1 = std::max(n_gauss, 1);
    if (r2s::dev_malloc((void**)&h->d_listed, sizeof(int) * n_links) != hipSuccess || r2s::dev_malloc((void**)&h->d_offset, sizeof(float) * 16 * n_links) != hipSuccess ||
        r2s::dev_malloc((void**)&h->d_base_inv, sizeof(float) * 16 * n_links) != hipSuccess || r2s::dev_malloc((void**)&h->d_order, sizeof(int) * n1) != hipSuccess ||
        r2s::dev_malloc((void**)&h->d_link, sizeof(int) * n1) != hipSuccess || r2s::dev_malloc((void**)&h->d_means, sizeof(float) * 3 * n1) != hipSuccess ||
        r2s::dev_malloc((void**)&h->d_quats, sizeof(float4) * n1) != hipSuccess)
        return fail(R2S_ERR_ALLOC);
    R2S_HIP_TRY(hipMemcpyAsync(h->d_listed, listed.data(), sizeof(int) * n_links, hipMemcpyHostToDevice, s));
    R2S_HIP_TRY(hipMemcpyAsync(h->d_offset, off32.data(), sizeof(float) * 16 * n_links, hipMemcpyHostToDevice, s));
    R2S_HIP_TRY(hipMemcpyAsync(h->d_base_inv, binv.data(), sizeof(float) * 16 * n_links, hipMemcpyHostToDevice, s));
    if (n_gauss > 0) {
        R2S_HIP_TRY(hipMemcpyAsync(h->d_order, order.data(), sizeof(int) * n_gauss, hipMemcpyHostToDevice, s));
        R2S_HIP_TRY(hipMemcpyAsync(h->d_link, link.data(), sizeof(int) * n_gauss, hipMemcpyHostToDevice, s));
        R2S_HIP_TRY(hipMemcpyAsync(h->d_means, means, sizeof(float) * 3 * (size_t)n_gauss, hipMemcpyHostToDevice, s));
        R2S_HIP_TRY(hipMemcpyAsync(h->d_quats, qn.data(), sizeof(float4) * (size_t)n_gauss, hipMemcpyHostToDevice, s));
    }
    R2S_HIP_TRY(hipStreamSynchronize(s));
    *out = h;
    return R2S_OK;
}

void r2s_robot_gs_destroy(R2SRobotGS* h)
{
    if (!h) return;
    (void)hipDeviceSynchronize();
    void* ptrs[] = {h->d_listed, h->d_order, h->d_link, h->d_offset, h->d_base_inv, h->d_means, h->d_quats, h->d_rec};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete h;
}

int r2s_robot_gs_transform(R2SRobotGS* h, int32_t n_env, const float* link_pose, float* means_out, int64_t means_env_stride, float* rotations_out,
                           int64_t rotations_env_stride, int32_t normalize, int32_t write_static, r2s_stream_t stream_)
{
    hipStream_t s = (hipStream_t)stream_;
    if (!h || n_env <= 0 || !link_pose || (h->n > 0 && (!means_out || !rotations_out))) return R2S_ERR_INVALID;
    if (n_env > h->cap_env) {
        if (h->d_rec) (void)hipFree(h->d_rec);
        h->d_rec = nullptr; h->cap_env = 0;
        R2S_HIP_TRY(r2s::dev_malloc((void**)&h->d_rec, sizeof(LinkRec) * (size_t)n_env * h->L));
        h->cap_env = n_env;
    }
    const int tot = n_env * h->L;
    hipLaunchKernelGGL(k_link_records, dim3((tot + 63) / 64), dim3(64), 0, s, n_env, h->L, link_pose, h->d_offset, h->d_base_inv, h->d_listed, h->d_rec);
    const int t0 = write_static ? 0 : h->n_static;
    if (h->n - t0 > 0)
        hipLaunchKernelGGL(k_robot_gs, dim3((h->n - t0 + 255) / 256, n_env), dim3(256), 0, s, h->n, h->L, h->d_order, h->d_link, h->d_means, h->d_quats, h->d_rec, means_out,
                           (long long)means_env_stride, rotations_out, (long long)rotations_env_stride, normalize, t0);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_robot_gs_debug(R2SRobotGS* h, const float** link_records)
{
    if (!h || !h->d_rec || !link_records) return R2S_ERR_INVALID;
    *link_records = reinterpret_cast<const float*>(h->d_rec);
    return R2S_OK;
}

} // extern "C"
