// Linear-blend skinning of Gaussians to PhysTwin particles for MI355X (gfx950).
//
// Built from scratch against the behaviour of the reference's `interpolate_motions`
//   sim/utils/gs/transform_utils.py:58-212   (called with quat=None and precomputed weights, gs_renderer.py:738-747).
// Two kernels per env step for a whole batch of environments:
//   k_bone_fit   one thread per (env, bone): F = sum_k a'_k a_k^T over the k_rel neighbour offsets before / after the motion,
//                its proper closest rotation (Kabsch) from a float64 Jacobi eigen-decomposition of F^T F, and the rank
//                test of torch.linalg.matrix_rank; writes one 64-byte record {R, bone, motion} per bone.
//   k_skin       one thread per (env, point): blends k_wgt bone records.
// The reference gets the same numbers from a batched torch.svd + matrix_rank + det and ~40 elementwise ops.

#include "r2s_common.h"
#include "../../include/r2s_skinning.h"
#include <algorithm>
#include <vector>

namespace {

struct __attribute__((aligned(16))) BoneRec {
    float r[9];
    float b[3];
    float m[3];
    float pad;
};

// cyclic Jacobi on a symmetric 3x3 (double): A = V diag(l) V^T
__device__ void jacobi3(double a[3][3], double v[3][3])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.0 : 0.0;
    // Stop once the off-diagonal part is below 1e-18 of the trace: a further rotation has tan < 1e-17, cos rounds to 1 and nothing it adds
    // reaches the 16th digit of A or V (round 6; until then the loop ran on to 1e-300, i.e. eight or nine sweeps until the off-diagonals
    // underflowed, for the same doubles: k_bone_fit 112 -> 60 us per 32-environment call).
    const double tr = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        if (off <= 1e-36 * tr * tr || off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { // A <- A J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) { // A <- J^T A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

__global__ void __launch_bounds__(256) k_bone_fit(int N, int k_rel, const int* __restrict__ relations, const float* __restrict__ bones,
                                                  const float* __restrict__ motions, BoneRec* __restrict__ rec, int* __restrict__ ident_flag)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (b >= N) return;
    const size_t eb = (size_t)e * N;
    const float bx = bones[(eb + b) * 3], by = bones[(eb + b) * 3 + 1], bz = bones[(eb + b) * 3 + 2];
    const float mx = motions[(eb + b) * 3], my = motions[(eb + b) * 3 + 1], mz = motions[(eb + b) * 3 + 2];
    // F = sum_k a'_k a_k^T in float32 like the reference's batched matmul (:79-83), then promoted
    float F[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int k = 0; k < k_rel; ++k) {
        const int j = relations[(size_t)b * k_rel + k];
        const float jx = bones[(eb + j) * 3], jy = bones[(eb + j) * 3 + 1], jz = bones[(eb + j) * 3 + 2];
        const float a[3] = {jx - bx, jy - by, jz - bz};
        const float n[3] = {(jx + motions[(eb + j) * 3]) - (bx + mx), (jy + motions[(eb + j) * 3 + 1]) - (by + my),
                            (jz + motions[(eb + j) * 3 + 2]) - (bz + mz)};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) F[r][c] += n[r] * a[c];
    }
    double A[3][3], V[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = (double)F[0][r] * F[0][c] + (double)F[1][r] * F[1][c] + (double)F[2][r] * F[2][c]; // F^T F
    jacobi3(A, V);
    // order the singular values descending
    double l[3] = {fmax(A[0][0], 0.0), fmax(A[1][1], 0.0), fmax(A[2][2], 0.0)};
    int o[3] = {0, 1, 2};
    if (l[o[0]] < l[o[1]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
    if (l[o[1]] < l[o[2]]) { int t = o[1]; o[1] = o[2]; o[2] = t; }
    if (l[o[0]] < l[o[1]]) { int t = o[0]; o[0] = o[1]; o[1] = t; }
    const double s0 = sqrt(l[o[0]]), s1 = sqrt(l[o[1]]), s2 = sqrt(l[o[2]]);
    const double tol = 3.0 * 1.1920928955078125e-07 * s0; // torch.linalg.matrix_rank default: max(m,n) * eps(float32) * sigma_max
    const int rank = (s0 > tol) + (s1 > tol) + (s2 > tol);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (rank >= 2) {
        double v1[3] = {V[0][o[0]], V[1][o[0]], V[2][o[0]]}, v2[3] = {V[0][o[1]], V[1][o[1]], V[2][o[1]]};
        double u1[3], u2[3];
        for (int r = 0; r < 3; ++r) {
            u1[r] = (F[r][0] * v1[0] + F[r][1] * v1[1] + F[r][2] * v1[2]) / s0;
            u2[r] = (F[r][0] * v2[0] + F[r][1] * v2[1] + F[r][2] * v2[2]) / s1;
        }
        // re-orthonormalise (guards the squared condition number of F^T F)
        double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
        for (int r = 0; r < 3; ++r) u1[r] /= n1;
        const double d12 = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
        for (int r = 0; r < 3; ++r) u2[r] -= d12 * u1[r];
        double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
        for (int r = 0; r < 3; ++r) u2[r] /= n2;
        const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
        const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
        // R = U S V^T with S chosen so that det R = +1 (:96-114)  ==  u1 v1^T + u2 v2^T + (u1 x u2)(v1 x v2)^T
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[3 * r + c] = u1[r] * v1[c] + u2[r] * v2[c] + u3[r] * v3[c];
    } else {
        atomicOr(ident_flag + e, 1); // the reference then falls back to identity for EVERY bone (:157-162)
    }
    BoneRec out;
    for (int k = 0; k < 9; ++k) out.r[k] = (float)R[k];
    out.b[0] = bx; out.b[1] = by; out.b[2] = bz;
    out.m[0] = mx; out.m[1] = my; out.m[2] = mz;
    out.pad = 0.f;
    rec[eb + b] = out;
}

// xyz' = sum_j w_j (R_bj (x - b_j) + m_j + b_j), transform_utils.py:178-189, in the reference's order of operations
// Thread t works on point order[t]: the points are walked sorted by their first bone, so that neighbouring lanes (and
// neighbouring workgroups) gather the same few 64-byte bone records — caller order has no locality (2.2 GB fetched per
// call on the benchmark scene before, for 0.25 GB of algorithmic traffic).  weights / widx are stored [k][t] (coalesced).
__global__ void __launch_bounds__(256) k_skin(int E, int N, int P, int k_wgt, const int* __restrict__ order, const float* __restrict__ weights,
                                              const int* __restrict__ widx, const BoneRec* __restrict__ rec, const int* __restrict__ ident_flag,
                                              const float* xyz, long long xyz_stride, float* out, long long out_stride) // xyz may alias out (in-place skinning): no __restrict__
{
#pragma clang fp contract(off)
    // Workgroup -> (environment, chunk of points) so that an environment is skinned by ONE XCD (linear workgroup id % 8 = XCD, a speed
    // assumption only): its bone records (64 B x n_bones, ~1 MB) are gathered 16 times per Gaussian and then stay in that XCD's 4 MB L2.
    // With the environment in blockIdx.y every XCD pulled every environment's records: 611 MB per call for 253 MB of algorithmic traffic.
    const int chunks = (P + (int)blockDim.x - 1) / (int)blockDim.x;
    const int xcd = (int)(blockIdx.x & 7), q = (int)(blockIdx.x >> 3);
    const int e = xcd + 8 * (q / chunks);
    const int t = (q % chunks) * (int)blockDim.x + (int)threadIdx.x;
    if (e >= E || t >= P) return;
    const int pt = order[t];
    const size_t ep = (size_t)e * xyz_stride + (size_t)pt * 3, eo = (size_t)e * out_stride + (size_t)pt * 3; // env strides in floats
    const float x = xyz[ep], y = xyz[ep + 1], z = xyz[ep + 2];
    const bool ident = ident_flag[e] != 0;
    float ax = 0.f, ay = 0.f, az = 0.f;
    auto blend = [&](const float4 q0, const float4 q1, const float4 q2, const float4 q3, const float w) { // r0..r3 | r4..r7 | r8 b0 b1 b2 | m0 m1 m2 pad
        const float dx = x - q2.y, dy = y - q2.z, dz = z - q2.w;
        float tx, ty, tz;
        if (ident) { tx = dx; ty = dy; tz = dz; }
        else {
            tx = q0.x * dx + q0.y * dy + q0.z * dz;
            ty = q0.w * dx + q1.x * dy + q1.y * dz;
            tz = q1.z * dx + q1.w * dy + q2.x * dz;
        }
        tx = tx + q3.x + q2.y; ty = ty + q3.y + q2.z; tz = tz + q3.z + q2.w;
        ax += tx * w; ay += ty * w; az += tz * w;
    };
    // The kernel waits for its gathers (VALU busy 0.12): four bones at a time — their indices, then their sixteen 16-byte gathers, are in
    // flight together; the sums run in the reference's order (bone 0 first).
#ifndef R2S_SKIN_BATCH
#define R2S_SKIN_BATCH 8 // bones whose gathers are in flight together: 1 (the loop until round 6) 208 us per 32-environment call, 2: 167, 4: 157, 8: 148 (two wavefronts per SIMD)
#endif
    constexpr int SB = R2S_SKIN_BATCH;
    int k = 0;
    for (; k + SB <= k_wgt; k += SB) {
        int j[SB]; float w[SB]; float4 r[SB][4];
#pragma unroll
        for (int u = 0; u < SB; ++u) { j[u] = widx[(size_t)(k + u) * P + t]; w[u] = weights[(size_t)(k + u) * P + t]; }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const float4* q = reinterpret_cast<const float4*>(rec + (size_t)e * N + j[u]);
            r[u][0] = q[0]; r[u][1] = q[1]; r[u][2] = q[2]; r[u][3] = q[3];
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) blend(r[u][0], r[u][1], r[u][2], r[u][3], w[u]);
    }
    for (; k < k_wgt; ++k) {
        const int j = widx[(size_t)k * P + t];
        const float w = weights[(size_t)k * P + t];
        const float4* q = reinterpret_cast<const float4*>(rec + (size_t)e * N + j);
        blend(q[0], q[1], q[2], q[3], w);
    }
    out[eo] = ax; out[eo + 1] = ay; out[eo + 2] = az;
}

// interpolate_motions(quat=...), transform_utils.py:197-210: every bone's rotation as a unit quaternion
// (kornia.geometry.conversions.rotation_matrix_to_quaternion — third party, restated like in robot_gs.hip: trace / largest-diagonal
// branches, eps 1e-8, (w, x, y, z) — then normalised), blended per Gaussian with the skinning weights, normalised, and composed
// with the Gaussian's own quaternion (Hamilton product, blended rotation first).
__device__ __forceinline__ void rotmat_to_quat(const float* m, float q[4])
{
    const float m00 = m[0], m01 = m[1], m02 = m[2], m10 = m[3], m11 = m[4], m12 = m[5], m20 = m[6], m21 = m[7], m22 = m[8];
    const float trace = m00 + m11 + m22;
    constexpr float eps = 1e-8f, tiny = 1.17549435e-38f;
    auto sdiv = [&](float a, float b) { return a / fmaxf(b, tiny); }; // safe_zero_division
    if (trace > 0.f) {
        const float sq = sqrtf(trace + 1.0f + eps) * 2.0f;
        q[0] = 0.25f * sq; q[1] = sdiv(m21 - m12, sq); q[2] = sdiv(m02 - m20, sq); q[3] = sdiv(m10 - m01, sq);
    } else if (m00 > m11 && m00 > m22) {
        const float sq = sqrtf(1.0f + m00 - m11 - m22 + eps) * 2.0f;
        q[0] = sdiv(m21 - m12, sq); q[1] = 0.25f * sq; q[2] = sdiv(m01 + m10, sq); q[3] = sdiv(m02 + m20, sq);
    } else if (m11 > m22) {
        const float sq = sqrtf(1.0f + m11 - m00 - m22 + eps) * 2.0f;
        q[0] = sdiv(m02 - m20, sq); q[1] = sdiv(m01 + m10, sq); q[2] = 0.25f * sq; q[3] = sdiv(m12 + m21, sq);
    } else {
        const float sq = sqrtf(1.0f + m22 - m00 - m11 + eps) * 2.0f;
        q[0] = sdiv(m10 - m01, sq); q[1] = sdiv(m02 + m20, sq); q[2] = sdiv(m12 + m21, sq); q[3] = 0.25f * sq;
    }
    const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f); // F.normalize
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

__global__ void __launch_bounds__(256) k_skin_quat(int N, int P, int k_wgt, const int* __restrict__ order, const float* __restrict__ weights,
                                                   const int* __restrict__ widx, const BoneRec* __restrict__ rec, const int* __restrict__ ident_flag,
                                                   const float* quat, long long quat_stride, float* out, long long out_stride)
{
#pragma clang fp contract(off)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    if (t >= P) return;
    const int pt = order[t];
    const bool ident = ident_flag[e] != 0;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < k_wgt; ++k) {
        const int j = widx[(size_t)k * P + t];
        const float w = weights[(size_t)k * P + t];
        float q[4] = {1.f, 0.f, 0.f, 0.f};
        if (!ident) rotmat_to_quat(rec[(size_t)e * N + j].r, q);
        for (int c = 0; c < 4; ++c) a[c] += q[c] * w;
    }
    const float n = fmaxf(sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + a[3] * a[3]), 1e-12f);
    for (int c = 0; c < 4; ++c) a[c] /= n;
    const float* q2 = quat + (size_t)e * quat_stride + (size_t)pt * 4;
    const float b0 = q2[0], b1 = q2[1], b2 = q2[2], b3 = q2[3];
    float* o = out + (size_t)e * out_stride + (size_t)pt * 4;
    o[0] = a[0] * b0 - a[1] * b1 - a[2] * b2 - a[3] * b3;
    o[1] = a[0] * b1 + a[1] * b0 + a[2] * b3 - a[3] * b2;
    o[2] = a[0] * b2 - a[1] * b3 + a[2] * b0 + a[3] * b1;
    o[3] = a[0] * b3 + a[1] * b2 - a[2] * b1 + a[3] * b0;
}

} // namespace

struct R2SSkin {
    int N = 0, k_rel = 0, P = 0, k_wgt = 0;
    int *d_rel = nullptr, *d_widx = nullptr, *d_flag = nullptr, *d_order = nullptr;
    float* d_w = nullptr;
    BoneRec* d_rec = nullptr;
    float* d_rot = nullptr; // debug copy
    int cap_env = 0;
};

extern "C" {

int r2s_skin_create(int32_t n_bones, int32_t k_rel, const int32_t* relations, int32_t n_points, int32_t k_wgt, const float* weights,
                    const int32_t* weights_indices, R2SSkin** out, r2s_stream_t stream_)
{
    hipStream_t s = (hipStream_t)stream_;
    if (!out || n_bones <= 0 || k_rel <= 0 || n_points < 0 || k_wgt <= 0 || !relations || (n_points > 0 && (!weights || !weights_indices)))
        return R2S_ERR_INVALID;
    for (size_t t = 0; t < (size_t)n_bones * k_rel; ++t) if (relations[t] < 0 || relations[t] >= n_bones) return R2S_ERR_INVALID;
    for (size_t t = 0; t < (size_t)n_points * k_wgt; ++t) if (weights_indices[t] < 0 || weights_indices[t] >= n_bones) return R2S_ERR_INVALID;
    R2SSkin* h = new (std::nothrow) R2SSkin();
    if (!h) return R2S_ERR_ALLOC;
    h->N = n_bones; h->k_rel = k_rel; h->P = n_points; h->k_wgt = k_wgt;
    auto fail = [&](int rc) { r2s_skin_destroy(h); return rc; };
    if (r2s::dev_malloc((void**)&h->d_rel, sizeof(int) * (size_t)n_bones * k_rel) != hipSuccess) return fail(R2S_ERR_ALLOC);
    if (r2s::dev_malloc((void**)&h->d_widx, sizeof(int) * std::max<size_t>((size_t)n_points * k_wgt, 1)) != hipSuccess) return fail(R2S_ERR_ALLOC);
    if (r2s::dev_malloc((void**)&h->d_w, sizeof(float) * std::max<size_t>((size_t)n_points * k_wgt, 1)) != hipSuccess) return fail(R2S_ERR_ALLOC);
    R2S_HIP_TRY(hipMemcpyAsync(h->d_rel, relations, sizeof(int) * (size_t)n_bones * k_rel, hipMemcpyHostToDevice, s));
    std::vector<int> order(n_points), widx_t((size_t)n_points * k_wgt);
    std::vector<float> w_t((size_t)n_points * k_wgt);
    if (n_points > 0) {
        if (r2s::dev_malloc((void**)&h->d_order, sizeof(int) * (size_t)n_points) != hipSuccess) return fail(R2S_ERR_ALLOC);
        for (int i = 0; i < n_points; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weights_indices[(size_t)a * k_wgt] < weights_indices[(size_t)b * k_wgt]; });
        for (int t = 0; t < n_points; ++t)
            for (int k = 0; k < k_wgt; ++k) {
                widx_t[(size_t)k * n_points + t] = weights_indices[(size_t)order[t] * k_wgt + k];
                w_t[(size_t)k * n_points + t] = weights[(size_t)order[t] * k_wgt + k];
            }
        R2S_HIP_TRY(hipMemcpyAsync(h->d_order, order.data(), sizeof(int) * (size_t)n_points, hipMemcpyHostToDevice, s));
        R2S_HIP_TRY(hipMemcpyAsync(h->d_widx, widx_t.data(), sizeof(int) * (size_t)n_points * k_wgt, hipMemcpyHostToDevice, s));
        R2S_HIP_TRY(hipMemcpyAsync(h->d_w, w_t.data(), sizeof(float) * (size_t)n_points * k_wgt, hipMemcpyHostToDevice, s));
    }
    R2S_HIP_TRY(hipStreamSynchronize(s));
    *out = h;
    return R2S_OK;
}

void r2s_skin_destroy(R2SSkin* h)
{
    if (!h) return;
    (void)hipDeviceSynchronize();
    void* ptrs[] = {h->d_rel, h->d_widx, h->d_w, h->d_flag, h->d_rec, h->d_rot, h->d_order};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete h;
}

int r2s_skin_interpolate_motions(R2SSkin* h, int32_t n_env, const float* bones, const float* motions, const float* xyz, float* xyz_out,
                                 r2s_stream_t stream_)
{
    return r2s_skin_interpolate_motions_strided(h, n_env, bones, motions, xyz, h ? 3 * (int64_t)h->P : 0, xyz_out, h ? 3 * (int64_t)h->P : 0, stream_);
}

int r2s_skin_interpolate_motions_strided(R2SSkin* h, int32_t n_env, const float* bones, const float* motions, const float* xyz, int64_t xyz_env_stride,
                                         float* xyz_out, int64_t out_env_stride, r2s_stream_t stream_)
{
    hipStream_t s = (hipStream_t)stream_;
    if (!h || n_env <= 0 || !bones || !motions || (h->P > 0 && (!xyz || !xyz_out))) return R2S_ERR_INVALID;
    if (h->P > 0 && (xyz_env_stride < 3 * (int64_t)h->P || out_env_stride < 3 * (int64_t)h->P) && n_env > 1) return R2S_ERR_INVALID;
    if (n_env > h->cap_env) {
        if (h->d_rec) (void)hipFree(h->d_rec);
        if (h->d_flag) (void)hipFree(h->d_flag);
        if (h->d_rot) (void)hipFree(h->d_rot);
        h->d_rec = nullptr; h->d_flag = nullptr; h->d_rot = nullptr; h->cap_env = 0;
        R2S_HIP_TRY(r2s::dev_malloc((void**)&h->d_rec, sizeof(BoneRec) * (size_t)n_env * h->N));
        R2S_HIP_TRY(r2s::dev_malloc((void**)&h->d_flag, sizeof(int) * (size_t)n_env));
        h->cap_env = n_env;
    }
    R2S_HIP_TRY(hipMemsetAsync(h->d_flag, 0, sizeof(int) * (size_t)n_env, s));
    hipLaunchKernelGGL(k_bone_fit, dim3((h->N + 255) / 256, n_env), dim3(256), 0, s, h->N, h->k_rel, h->d_rel, bones, motions, h->d_rec, h->d_flag);
    if (h->P > 0)
        hipLaunchKernelGGL(k_skin, dim3((unsigned)(8 * ((h->P + 255) / 256) * ((n_env + 7) / 8))), dim3(256), 0, s, n_env, h->N, h->P, h->k_wgt, h->d_order, h->d_w,
                           h->d_widx, h->d_rec, h->d_flag, xyz, (long long)xyz_env_stride, xyz_out, (long long)out_env_stride);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_skin_rotate_quats(R2SSkin* h, int32_t n_env, const float* quat, int64_t quat_env_stride, float* quat_out, int64_t out_env_stride,
                          r2s_stream_t stream_)
{
    if (!h || n_env <= 0 || n_env > h->cap_env || !h->d_rec || (h->P > 0 && (!quat || !quat_out))) return R2S_ERR_INVALID;
    if (h->P > 0)
        hipLaunchKernelGGL(k_skin_quat, dim3((h->P + 255) / 256, n_env), dim3(256), 0, (hipStream_t)stream_, h->N, h->P, h->k_wgt, h->d_order, h->d_w, h->d_widx,
                           h->d_rec, h->d_flag, quat, (long long)quat_env_stride, quat_out, (long long)out_env_stride);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_skin_debug(R2SSkin* h, const float** rotations, const int32_t** identity_flags)
{
    if (!h || !h->d_rec) return R2S_ERR_INVALID;
    if (rotations) { // unpack the rotations of the records into a dense [cap_env, N, 9] array
        if (!h->d_rot) R2S_HIP_TRY(r2s::dev_malloc((void**)&h->d_rot, sizeof(float) * 9 * (size_t)h->cap_env * h->N));
        R2S_HIP_TRY(hipMemcpy2D(h->d_rot, sizeof(float) * 9, h->d_rec, sizeof(BoneRec), sizeof(float) * 9, (size_t)h->cap_env * h->N, hipMemcpyDeviceToDevice));
        *rotations = h->d_rot;
    }
    if (identity_flags) *identity_flags = h->d_flag;
    return R2S_OK;
}

} // extern "C"
