// Observation sink, device half (include/r2s_obs.h): planar float RGB frames -> interleaved 8-bit pixels, all frames of a
// batch in one launch.  One lane = 4 consecutive pixels of a row: three 16-byte coalesced loads (one per colour plane), one
// 12-byte store; 12 B read + 3 B written per pixel — a pure HBM stream (19.7 M pixels per 64-frame batch: ~0.3 GB).
#include "r2s_common.h"
#include "../../include/r2s_obs.h"

namespace {
#pragma clang fp contract(off)
__device__ __forceinline__ uint32_t u8(float c) { return (uint32_t)(fminf(fmaxf(c, 0.0f), 1.0f) * 255.0f); } // clamp, * 255 (f32), astype(uint8)

__global__ void __launch_bounds__(256) k_pack_u8(const float* __restrict__ color, long long hw, long long quads_per_frame, int bgr, uint8_t* __restrict__ out)
{
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (q >= quads_per_frame) return;
    const float* base = color + (size_t)f * 3 * hw;
    const long long p0 = q * 4;
    float r[4], g[4], b[4];
    if (p0 + 4 <= hw && (hw & 3) == 0) {
        const float4 R = *reinterpret_cast<const float4*>(base + p0), G = *reinterpret_cast<const float4*>(base + hw + p0),
                     B = *reinterpret_cast<const float4*>(base + 2 * hw + p0);
        r[0] = R.x; r[1] = R.y; r[2] = R.z; r[3] = R.w; g[0] = G.x; g[1] = G.y; g[2] = G.z; g[3] = G.w; b[0] = B.x; b[1] = B.y; b[2] = B.z; b[3] = B.w;
    } else {
        for (int k = 0; k < 4; ++k) { const long long p = p0 + k < hw ? p0 + k : hw - 1; r[k] = base[p]; g[k] = base[hw + p]; b[k] = base[2 * hw + p]; }
    }
    uint8_t px[12];
    for (int k = 0; k < 4; ++k) {
        px[3 * k] = (uint8_t)u8(bgr ? b[k] : r[k]); px[3 * k + 1] = (uint8_t)u8(g[k]); px[3 * k + 2] = (uint8_t)u8(bgr ? r[k] : b[k]);
    }
    uint8_t* o = out + ((size_t)f * hw + p0) * 3;
    if (p0 + 4 <= hw && (((size_t)f * hw + p0) * 3) % 4 == 0) {
        uint32_t w[3];
        for (int k = 0; k < 3; ++k) w[k] = px[4 * k] | (px[4 * k + 1] << 8) | (px[4 * k + 2] << 16) | ((uint32_t)px[4 * k + 3] << 24);
        uint32_t* o4 = reinterpret_cast<uint32_t*>(o);
        o4[0] = w[0]; o4[1] = w[1]; o4[2] = w[2];
    } else {
        for (int k = 0; k < 12 && p0 * 3 + k < hw * 3; ++k) o[k] = px[k];
    }
}
} // namespace

extern "C" int r2s_obs_pack_u8(const float* color, int32_t n_frames, int32_t height, int32_t width, int32_t bgr, uint8_t* out, r2s_stream_t stream)
{
    if (n_frames < 0 || height <= 0 || width <= 0 || (n_frames > 0 && (!color || !out))) return R2S_ERR_INVALID;
    if (n_frames == 0) return R2S_OK;
    const long long hw = (long long)height * width, quads = (hw + 3) / 4;
    hipLaunchKernelGGL(k_pack_u8, dim3((unsigned)((quads + 255) / 256), (unsigned)n_frames), dim3(256), 0, (hipStream_t)stream, color, hw, quads, bgr, out);
    R2S_HIP_TRY(hipGetLastError());
    return R2S_OK;
}
