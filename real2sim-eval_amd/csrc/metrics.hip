// Task-success predicates on the device (include/r2s_metrics.h): three small reductions over the particle state of
// every environment.  float64 where the reference scripts compute in float64 (numpy `dtype=float`).
#include "r2s_common.h"
#include "../../include/r2s_metrics.h"

namespace {

struct Box { double lo[3], hi[3]; };
struct Obb { double c[3], r[9], h[3]; };

__device__ __forceinline__ int wave_sum(int v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// _segment_plane_intersections_xz, calculate_success_rope.py:40-72, for one segment and one plane
__device__ __forceinline__ bool crosses(const double* p0, const double* p1, double y_plane, const Box& b, double eps)
{
    const double y0 = p0[1], y1 = p1[1], dy = y1 - y0;
    const bool parallel = fabs(dy) <= eps;                      // np.isclose(dy, 0.0, atol=eps): rtol * |0| = 0
    const double x_min = b.lo[0], x_max = b.hi[0], z_min = b.lo[2], z_max = b.hi[2];
    if (!parallel) {
        const double t = (y_plane - y0) / dy;
        if (!(t >= -eps && t <= 1.0 + eps)) return false;
        const double xi = p0[0] + t * (p1[0] - p0[0]), zi = p0[2] + t * (p1[2] - p0[2]);
        return xi >= x_min - eps && xi <= x_max + eps && zi >= z_min - eps && zi <= z_max + eps;
    }
    if (!(fabs(y0 - y_plane) <= eps)) return false;             // coplanar: np.isclose(y0 - y_plane, 0.0, atol=eps)
    const bool e0 = p0[0] >= x_min - eps && p0[0] <= x_max + eps && p0[2] >= z_min - eps && p0[2] <= z_max + eps;
    const bool e1 = p1[0] >= x_min - eps && p1[0] <= x_max + eps && p1[2] >= z_min - eps && p1[2] <= z_max + eps;
    return e0 || e1;
}

__global__ void __launch_bounds__(256) k_plane_crossings(int N, int S, const float* __restrict__ x, const int* __restrict__ springs, Box b, double eps,
                                                         int* __restrict__ counts)
{
#pragma clang fp contract(off)
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    int lo = 0, hi = 0;
    if (s < S) {
        const int i = springs[2 * s], j = springs[2 * s + 1];
        const float* a = x + ((size_t)e * N + i) * 3; const float* c = x + ((size_t)e * N + j) * 3;
        const double p0[3] = {(double)a[0], (double)a[1], (double)a[2]}, p1[3] = {(double)c[0], (double)c[1], (double)c[2]};
        lo = crosses(p0, p1, b.lo[1], b, eps);
        hi = crosses(p0, p1, b.hi[1], b, eps);
    }
    lo = wave_sum(lo); hi = wave_sum(hi);
    if ((threadIdx.x & 63) == 0) {
        if (lo) atomicAdd(counts + 2 * e, lo);
        if (hi) atomicAdd(counts + 2 * e + 1, hi);
    }
}

__global__ void __launch_bounds__(256) k_mse(int N, const float* __restrict__ x, const float* __restrict__ target, double* __restrict__ out)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    double v = 0.0;
    if (i < N) {
        const float* a = x + ((size_t)e * N + i) * 3; const float* t = target + (size_t)i * 3;
        const float dx = a[0] - t[0], dy = a[1] - t[1], dz = a[2] - t[2]; // float32 differences and squares like the script
        v = (double)((dx * dx + dy * dy) + dz * dz);
    }
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + e, v / (double)N);
}

__global__ void __launch_bounds__(256) k_in_obb(int N, const float* __restrict__ x, Obb o, int* __restrict__ count)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = blockIdx.y;
    int in = 0;
    if (i < N) {
        const float* a = x + ((size_t)e * N + i) * 3;
        const double d[3] = {(double)a[0] - o.c[0], (double)a[1] - o.c[1], (double)a[2] - o.c[2]};
        in = 1;
        for (int k = 0; k < 3; ++k) { // local coordinate along box axis k = column k of R
            const double l = (d[0] * o.r[0 * 3 + k] + d[1] * o.r[1 * 3 + k]) + d[2] * o.r[2 * 3 + k];
            in = in && fabs(l) <= o.h[k];
        }
    }
    in = wave_sum(in);
    if ((threadIdx.x & 63) == 0 && in) atomicAdd(count + e, in);
}

} // namespace

extern "C" {

int r2s_metric_plane_crossings(int32_t n_env, int32_t n_points, const float* x, int32_t n_springs, const int32_t* springs, const double* bbox_min,
                               const double* bbox_max, double eps, int32_t* counts, r2s_stream_t stream_)
{
    if (n_env <= 0 || n_points <= 0 || n_springs < 0 || !x || (n_springs > 0 && !springs) || !bbox_min || !bbox_max || !counts) return R2S_ERR_INVALID;
    for (int k = 0; k < 3; ++k) if (bbox_min[k] > bbox_max[k]) return R2S_ERR_INVALID; // "bbox min must be <= max"
    hipStream_t s = (hipStream_t)stream_;
    R2S_HIP_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * 2 * (size_t)n_env, s));
    if (n_springs == 0) return R2S_OK;
    Box b;
    for (int k = 0; k < 3; ++k) { b.lo[k] = bbox_min[k]; b.hi[k] = bbox_max[k]; }
    hipLaunchKernelGGL(k_plane_crossings, dim3((n_springs + 255) / 256, n_env), dim3(256), 0, s, n_points, n_springs, x, springs, b, eps, counts);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_metric_mse(int32_t n_env, int32_t n_points, const float* x, const float* x_target, double* mse, r2s_stream_t stream_)
{
    if (n_env <= 0 || n_points <= 0 || !x || !x_target || !mse) return R2S_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream_;
    R2S_HIP_TRY(hipMemsetAsync(mse, 0, sizeof(double) * (size_t)n_env, s));
    hipLaunchKernelGGL(k_mse, dim3((n_points + 255) / 256, n_env), dim3(256), 0, s, n_points, x, x_target, mse);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

int r2s_metric_points_in_obb(int32_t n_env, int32_t n_points, const float* x, const double* center, const double* R, const double* half_extent,
                             int32_t* count, r2s_stream_t stream_)
{
    if (n_env <= 0 || n_points <= 0 || !x || !center || !R || !half_extent || !count) return R2S_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream_;
    R2S_HIP_TRY(hipMemsetAsync(count, 0, sizeof(int32_t) * (size_t)n_env, s));
    Obb o;
    for (int k = 0; k < 3; ++k) { o.c[k] = center[k]; o.h[k] = half_extent[k]; }
    for (int k = 0; k < 9; ++k) o.r[k] = R[k];
    hipLaunchKernelGGL(k_in_obb, dim3((n_points + 255) / 256, n_env), dim3(256), 0, s, n_points, x, o, count);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}

} // extern "C"
