// Gaussian-splatting rasteriser forward pass for MI355X (gfx950, wave64).
//
// Built from scratch against the behaviour of the reference CUDA rasteriser
//   third-party/diff-gaussian-rasterization-w-depth/cuda_rasterizer/{forward.cu,rasterizer_impl.cu,auxiliary.h}
// (forward only, median-depth output).  Pipeline for a batch of F frames (environment x camera view):
//
//   k_preprocess   one thread per (frame, Gaussian): cull, project, EWA conic, radius, tile rect, SH->RGB.
//                  Writes ONE packed 48-byte record per Gaussian (what compositing gathers later) and the key of the first sort:
//                  frame | depth bits | index inside the frame.
//   rocPRIM sort   radix_sort_keys on the (frame, depth bits) fields: 32 + log2(F) bits, stable; the index rides in the key.
//   rocPRIM scan   inclusive sum of tiles_touched read through that order (TilesInOrder), all frames at once.
//   k_emit_keys    (tile inside the frame u16, global Gaussian index) per overlapped tile, Gaussians walked in depth order.
//   k_bin_*        ONE-pass stable partition of the instances by (frame, tile) — chunk histograms, column scan, ranked scatter — so a
//                  tile's list stays in depth order and ties keep index order: the order of the reference's single 64-bit sort.  The scan
//                  of the histograms IS the per-tile [start, end) table.  (Frames of more than 2 048 tiles: two-pass rocPRIM sort on
//                  frame-extended 32-bit keys + k_tile_ranges, the form of rounds 1-5.)
//   k_tile_order   compositor workgroups longest list first (counting sort over length classes).
//   k_composite    one 256-thread workgroup per 16x16 tile, each of its 4 wavefronts owns an 8x8 pixel
//                  quadrant; 256 instance records per round are staged through LDS and broadcast-read.
//
// Reference lines are cited at each function.  No CUDA compatibility layer: wave64, HIP only.

#include "r2s_common.h"
#include <rocprim/rocprim.hpp>
#include "../../include/r2s_raster.h"
#include <vector>
#include <type_traits>

namespace {

constexpr int TILE = 16;         // BLOCK_X == BLOCK_Y, cuda_rasterizer/config.h:15-16
constexpr int TILE_THREADS = 256;

struct FrameDev {
    const float* view;
    const float* proj;
    const float* campos;
    const float* bg;
    const float* means3D;
    const float* shs;
    const float* colors;
    const float* opac;
    const float* scales;
    const float* rots;
    const float* cov3D;
    float* out_color;
    float* out_depth;
    int* radii;
    int P, D, M;
    uint32_t base; // index of this frame's first Gaussian in the per-batch geometry arrays
    float scale_mod, tan_fovx, tan_fovy, focal_x, focal_y, z_thr;
    int prefiltered;
};

// Packed per-Gaussian record read by the compositor: 3 x 16 B.
//   q0 = (px, py, conic_a, conic_b)   q1 = (conic_c, opacity, depth, r)   q2 = (g, b, 0, 0)
struct __attribute__((aligned(16))) GeomRec {
    float4 q0, q1, q2;
};

// ---- per-Gaussian math -------------------------------------------------------------------------

// ndc2Pix, auxiliary.h:41-44 (double-precision literals promote the expression to double).
__device__ __forceinline__ float ndc2pix(float v, int S)
{
    return (float)((((double)v + 1.0) * S - 1.0) * 0.5);
}

// getRect, auxiliary.h:46-56.
__device__ __forceinline__ void tile_rect(float px, float py, int r, int gx, int gy, uint32_t& x0, uint32_t& y0,
                                          uint32_t& x1, uint32_t& y1)
{
#pragma clang fp contract(off)
    x0 = (uint32_t)min(gx, max(0, (int)((px - (float)r) / (float)TILE)));
    y0 = (uint32_t)min(gy, max(0, (int)((py - (float)r) / (float)TILE)));
    x1 = (uint32_t)min(gx, max(0, (int)((px + (float)r + (float)TILE - 1.0f) / (float)TILE)));
    y1 = (uint32_t)min(gy, max(0, (int)((py + (float)r + (float)TILE - 1.0f) / (float)TILE)));
}

// Exact-output tile culling (optional; never used by the single-frame drop-in entry point).
// A (Gaussian, tile) instance can only change a pixel if alpha = min(0.99, o * exp(power)) >= 1/255 somewhere in the
// tile (forward.cu:350-352 skips it otherwise, with no side effect on C, T or the median depth).  power is minus half a
// positive-definite quadratic in d = mean - pixel; its maximum over the tile's pixel RECTANGLE (a superset of the pixel
// lattice) is attained at the mean if that is inside, else on one of the two rectangle edges facing the mean.  The
// instance is dropped only if that bound is 1 % below the threshold, so float rounding can never drop a live instance.
// rdy = -cb / cc and rdx = -cb / ca (the unconstrained minimiser's slope along an edge) are formed ONCE per Gaussian by the caller: two IEEE
// divisions per candidate tile were a third of this test's instructions, and the test is what k_preprocess and k_emit_keys are bound by.
// (The bound is conservative with 0.01 of slack in the log domain; how its arithmetic rounds never changes a pixel.)
__device__ __forceinline__ bool rect_can_contribute(float mx, float my, float ca, float cb, float cc, float rdy, float rdx, float log_thresh, float x_lo,
                                                    float x_hi, float y_lo, float y_hi)
{
    const bool in_x = mx >= x_lo && mx <= x_hi, in_y = my >= y_lo && my <= y_hi;
    if (in_x && in_y) return true;
    float qmin = 3.0e38f;
    if (!in_x) { // vertical edge facing the mean: dx fixed, minimise over dy = my - y, y in [y_lo, y_hi]
        const float dx = mx - (mx < x_lo ? x_lo : x_hi);
        const float dy_free = rdy * dx;
        const float dy = fminf(fmaxf(dy_free, my - y_hi), my - y_lo);
        qmin = fminf(qmin, ca * dx * dx + 2.f * cb * dx * dy + cc * dy * dy);
    }
    if (!in_y) {
        const float dy = my - (my < y_lo ? y_lo : y_hi);
        const float dx_free = rdx * dy;
        const float dx = fminf(fmaxf(dx_free, mx - x_hi), mx - x_lo);
        qmin = fminf(qmin, ca * dx * dx + 2.f * cb * dx * dy + cc * dy * dy);
    }
    return !(-0.5f * qmin < log_thresh - 0.01f); // NaN-safe: anything odd keeps the instance
}

__device__ __forceinline__ bool tile_can_contribute(float mx, float my, float ca, float cb, float cc, float rdy, float rdx, float log_thresh, int tx, int ty,
                                                    int W, int H)
{
    return rect_can_contribute(mx, my, ca, cb, cc, rdy, rdx, log_thresh, (float)(tx * TILE), (float)min(tx * TILE + TILE - 1, W - 1),
                               (float)(ty * TILE), (float)min(ty * TILE + TILE - 1, H - 1));
}

// computeColorFromSH, forward.cu:20-71.
__device__ __forceinline__ void sh_to_rgb(int idx, int deg, int M, const float* means, const float* campos,
                                          const float* shs, float rgb[3])
{
#pragma clang fp contract(off)
    constexpr float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    constexpr float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                             0.5462742152960396f};
    constexpr float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    const float* sh = shs + (size_t)idx * M * 3;
    float x = 0.f, y = 0.f, z = 0.f;
    if (deg > 0) {
        float dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
        float len = sqrtf(dx * dx + dy * dy + dz * dz);
        x = dx / len; y = dy / len; z = dz / len;
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float res = C0 * sh[ch];
        if (deg > 0) {
            res = res - C1 * y * sh[3 + ch] + C1 * z * sh[6 + ch] - C1 * x * sh[9 + ch];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + C2[0] * xy * sh[12 + ch] + C2[1] * yz * sh[15 + ch] +
                      C2[2] * (2.0f * zz - xx - yy) * sh[18 + ch] + C2[3] * xz * sh[21 + ch] +
                      C2[4] * (xx - yy) * sh[24 + ch];
                if (deg > 2) {
                    res = res + C3[0] * y * (3.0f * xx - yy) * sh[27 + ch] + C3[1] * xy * z * sh[30 + ch] +
                          C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + ch] +
                          C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + ch] +
                          C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + ch] + C3[5] * z * (xx - yy) * sh[42 + ch] +
                          C3[6] * x * (xx - 3.0f * yy) * sh[45 + ch];
                }
            }
        }
        res += 0.5f;
        rgb[ch] = res > 0.f ? res : 0.f; // clamp flags are backward-only and not produced
    }
}

// The candidate tiles of one Gaussian — its 3-sigma bounding rectangle in tiles, row-major like duplicateWithKeys walks them
// (rasterizer_impl.cu:97-108) — counted (k_preprocess, exact-output culling) or emitted (k_emit_keys).  One lane per Gaussian
// used to walk its own rectangle: with the wrist camera riding on the gripper a few splats close to the lens cover hundreds of
// tiles, and every wavefront waited for the largest rectangle among its 64 Gaussians (k_preprocess 0.25 -> 0.73 ms, k_emit_keys
// 0.32 -> 0.73 ms per 64 frames when the camera started to follow the gripper).  Now rectangles of up to RECT_SMALL_* tiles stay
// per lane; the larger ones are taken ONE AT A TIME BY THE WHOLE WAVEFRONT, 64 candidate tiles per trip: the count is a ballot,
// and the emitted run of a Gaussian is written with consecutive lanes at consecutive offsets (the per-lane walk scatters
// 4-byte stores: 8x the payload in HBM traffic, profiles/r2_pmc_summary.json).  Order and content of the lists are unchanged.
// Candidate tiles under the culling test (round 6): the reference's rectangle — the bounding square of 3 sigma of the LARGER axis (getRect) —
// cut down to the tiles that the bounding box of the alpha >= 1/255 ellipse can reach: half-widths sqrt(2 (ln(255 o) + margin) Sigma_xx / yy).
// tile_can_contribute drops every tile outside that box anyway (its bound is the same quadratic form, with 0.01 of slack in the log domain:
// here 0.02 and a pixel on top), so the survivors and their row-major order are unchanged; what shrinks is the walk — on the benchmark batch
// the square holds 63 M candidate tiles for 36 M survivors (elongated and faint splats) — and, on the binning path, the emission slots.
// Computed from the STORED record (k_preprocess and k_emit_keys must agree to the bit: contraction off).
__device__ __forceinline__ void cull_rect(float mx, float my, float ca, float cb, float cc, float opac, int gx, int gy, uint32_t& x0, uint32_t& y0,
                                          uint32_t& x1, uint32_t& y1)
{
#pragma clang fp contract(off)
    const float lr = logf(255.0f * opac) + 0.02f;
    const float det = ca * cc - cb * cb;
    if (!(det > 0.f) || !(lr == lr)) return; // an odd conic or opacity keeps the reference's rectangle (the test itself keeps anything odd)
    if (!(lr > 0.f)) { x1 = x0; y1 = y0; return; } // fainter than 1/255 at its centre
    const float hx = sqrtf(2.f * lr * cc / det) + 1.0f, hy = sqrtf(2.f * lr * ca / det) + 1.0f;
    const float ax = floorf((mx - hx - 15.f) * (1.0f / (float)TILE)), bx = floorf((mx + hx) * (1.0f / (float)TILE)) + 1.f;
    const float ay = floorf((my - hy - 15.f) * (1.0f / (float)TILE)), by = floorf((my + hy) * (1.0f / (float)TILE)) + 1.f;
    const uint32_t cx0 = (uint32_t)fminf(fmaxf(ax, 0.f), (float)gx), cx1 = (uint32_t)fminf(fmaxf(bx, 0.f), (float)gx);
    const uint32_t cy0 = (uint32_t)fminf(fmaxf(ay, 0.f), (float)gy), cy1 = (uint32_t)fminf(fmaxf(by, 0.f), (float)gy);
    x0 = max(x0, cx0); x1 = max(min(x1, cx1), x0);
    y0 = max(y0, cy0); y1 = max(min(y1, cy1), y0);
}

// Binning path (16-bit keys, round 6): a Gaussian's candidate tile k owns emission slot `offset + k`, whether it passes the culling test or
// not — a culled candidate leaves KEY_NONE there and the binning skips it.  So the slots follow from the RECTANGLES alone: k_preprocess does
// not walk the candidates to count the survivors (63 % of its instructions), and the emitting walk needs no rank among the survivors.  The
// instance count is the total of the tile histograms instead of the last scan element; the scan is the capacity the slots need (at most the
// reference's count, ~5 % above the culled one).
constexpr uint32_t KEY_NONE = 0xffffu;
struct RectJob {
    float mx, my, ca, cb, cc, lt, rdy, rdx; // rdy = -cb / cc, rdx = -cb / ca
    uint32_t x0, y0, x1, y1; // empty (x1 == x0) for lanes without a visible Gaussian
};
// measured on the benchmark scene (64 frames, wrist cameras on the grippers), per-lane limit 2 / 6 / 12 / 24 / 40 / 64 / 100 tiles:
// count 0.80 / 0.57 / 0.47 / 0.43 / 0.41 / 0.43 / 0.47 ms, emit 0.86 / 0.60 / 0.47 / 0.44 / 0.47 / 0.51 / 0.54 ms (all per lane: 0.73 / 0.73)
constexpr uint32_t RECT_SMALL_COUNT = 40, RECT_SMALL_EMIT = 24;
#ifdef R2S_SURV_MASKS // experiment (measured, NOT the build: see rect_walk_flat): the counting walk leaves survivor masks for the emitting walk
constexpr bool SURV_MASKS = true;
#else
constexpr bool SURV_MASKS = false;
#endif
// EMIT = false: returns the number of candidate tiles that can contribute (every lane of the wavefront must call it).
// EMIT = true: writes (tile key, value) of the surviving tiles from offset `off` on, clamped to `cap`; `test` = apply the culling
// test (off in the reference-exact mode, where every tile of the rectangle is an instance).
template <bool EMIT, typename KeyT = uint32_t>
__device__ __forceinline__ uint32_t rect_walk(const RectJob& j, int W, int H, uint32_t off, uint32_t cap, uint32_t tile_base, uint32_t val,
                                              KeyT* __restrict__ keys, uint32_t* __restrict__ vals, int gx = 0, bool test = true)
{
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t w = j.x1 - j.x0, n = w * (j.y1 - j.y0);
    uint32_t count = 0;
    constexpr bool SLOTS = EMIT && sizeof(KeyT) == 2; // binning path: candidate k of a Gaussian owns slot off + k (see rect_walk_flat)
    const bool big = n > (EMIT ? RECT_SMALL_EMIT : RECT_SMALL_COUNT);
    if (n > 0 && !big) {
        for (uint32_t y = j.y0; y < j.y1; ++y)
            for (uint32_t x = j.x0; x < j.x1; ++x) {
                const bool pass = !test || tile_can_contribute(j.mx, j.my, j.ca, j.cb, j.cc, j.rdy, j.rdx, j.lt, (int)x, (int)y, W, H);
                if (SLOTS) { if (off < cap) { keys[off] = pass ? (KeyT)(y * (uint32_t)gx + x) : (KeyT)KEY_NONE; if (pass) vals[off] = val; } ++off; }
                if (!pass) continue;
                if (EMIT && !SLOTS) { if (off < cap) { keys[off] = (KeyT)(tile_base + y * (uint32_t)gx + x); vals[off] = val; } ++off; }
                ++count;
            }
    }
    unsigned long long m = __builtin_amdgcn_ballot_w64(big);
    while (m) { // wave-uniform: one large rectangle per trip, all 64 lanes on it
        const int L = __builtin_ctzll(m);
        m &= m - 1;
        auto bf = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), L)); };
        auto bu = [&](uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, L); };
        const float mx = bf(j.mx), my = bf(j.my), ca = bf(j.ca), cb = bf(j.cb), cc = bf(j.cc), lt = bf(j.lt), rdy = bf(j.rdy), rdx = bf(j.rdx);
        const uint32_t x0 = bu(j.x0), y0 = bu(j.y0), wL = bu(w), nL = bu(n);
        uint32_t base = EMIT ? bu(off) : 0u;
        const uint32_t tb = EMIT ? bu(tile_base) : 0u, vL = EMIT ? bu(val) : 0u;
        uint32_t total = 0;
        for (uint32_t t0 = 0; t0 < nL; t0 += 64) {
            const uint32_t t = t0 + (uint32_t)lane;
            const uint32_t ry = t / wL, rx = t - ry * wL;
            const bool ok = t < nL && (!test || tile_can_contribute(mx, my, ca, cb, cc, rdy, rdx, lt, (int)(x0 + rx), (int)(y0 + ry), W, H));
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(ok);
            if (SLOTS) {
                const uint32_t pos = base + t;
                if (t < nL && pos < cap) { keys[pos] = ok ? (KeyT)((y0 + ry) * (uint32_t)gx + (x0 + rx)) : (KeyT)KEY_NONE; if (ok) vals[pos] = vL; }
            } else if (EMIT) {
                const uint32_t pos = base + (uint32_t)__builtin_popcountll(bal & ((1ull << lane) - 1ull));
                if (ok && pos < cap) { keys[pos] = (KeyT)(tb + (y0 + ry) * (uint32_t)gx + (x0 + rx)); vals[pos] = vL; }
                base += (uint32_t)__builtin_popcountll(bal);
            }
            total += (uint32_t)__builtin_popcountll(bal);
        }
        if (lane == L) count = total;
    }
    return count;
}

// The candidate tiles of a wavefront's 64 Gaussians, FLATTENED (round 3, late): rect_walk gives every Gaussian to one lane (or, the large
// ones, to the whole wavefront one after the other), so a wavefront runs as many trips as its largest small rectangle has tiles while
// most lanes idle — the counters put k_emit_keys at 215 M wavefront instructions for 37 M candidate tiles, nine times what the tiles
// need.  Here the 64 rectangles are laid end to end (prefix sum of their tile counts, parameters parked in the wavefront's LDS slice),
// and trip t hands candidate 64 t + lane to lane `lane`: a binary search of the prefix array names its Gaussian, the test runs on all 64
// lanes, and the passing tiles of a Gaussian find their rank from a ballot plus the running count of earlier trips.  Counts and emitted
// lists are the ones rect_walk produces (tile order inside a Gaussian is row-major either way).
struct WalkShared {      // one per wavefront
    uint32_t pre[64];    // inclusive prefix of the candidate counts
    float4 p0[64];       // mx, my, ca, cb
    float4 p1[64];       // cc, lt, rdy, rdx
    uint4 p2[64];        // x0, y0, w, first output slot (emit)
    uint2 p3[64];        // tile_base, value (emit)
    uint32_t cnt[64];    // passing tiles so far
    unsigned long long mask[64]; // which candidate tiles of the Gaussian pass (rectangles of up to 64 tiles; see k_preprocess)
    float inv_w[64];     // 1 / rectangle width (one IEEE division per Gaussian instead of one per candidate tile)
};
// Survivor masks (round 6, -DR2S_SURV_MASKS; measured and NOT taken): the culling test is ~70 of the 240 instructions of a trip, and the counting walk
// of k_preprocess already knows the answer the emitting walk of k_emit_keys asks for again.  With MASKS the counting walk leaves, for every Gaussian
// whose rectangle has at most 64 candidate tiles, a 64-bit word — bit k = candidate k passes — and the emitting walk reads the bit instead of testing
// (larger rectangles are tested again: a wavefront skips the test code unless one of its candidates needs it).  Lists and images identical; on the
// 64-frame benchmark batch k_preprocess 308 -> 334 us (the word's bookkeeping + 41 MB), k_emit_keys 315 -> 304 us: a loss of 15 us.
template <bool EMIT, typename KeyT = uint32_t, bool MASKS = false>
__device__ __forceinline__ uint32_t rect_walk_flat(WalkShared& ws, const RectJob& j, int W, int H, uint32_t off, uint32_t cap, uint32_t tile_base,
                                                   uint32_t val, KeyT* __restrict__ keys, uint32_t* __restrict__ vals, int gx, bool test,
                                                   unsigned long long mask = 0ull)
{
    const int lane = (int)(threadIdx.x & 63);
    const uint32_t w = j.x1 - j.x0, n = w * (j.y1 - j.y0);
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
    ws.pre[lane] = incl;
    ws.p0[lane] = make_float4(j.mx, j.my, j.ca, j.cb);
    ws.p1[lane] = make_float4(j.cc, j.lt, j.rdy, j.rdx);
    ws.p2[lane] = make_uint4(j.x0, j.y0, w, off);
    if (EMIT) ws.p3[lane] = make_uint2(tile_base, val);
    ws.cnt[lane] = 0u;
    ws.inv_w[lane] = 1.0f / (float)max(w, 1u);
    if (MASKS) ws.mask[lane] = EMIT ? mask : 0ull;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    for (uint32_t t0 = 0; t0 < total; t0 += 64) { // wave-uniform
        const uint32_t c = t0 + (uint32_t)lane;
        const bool active = c < total;
        int lo = 0, hi = 63; // the Gaussian of candidate c: the first lane whose inclusive prefix exceeds c
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi) >> 1;
            if (ws.pre[mid] > c) hi = mid; else lo = mid + 1;
        }
        const int g = min(lo, 63);
        const uint32_t start = g > 0 ? ws.pre[g - 1] : 0u;
        const float4 q0 = ws.p0[g], q1 = ws.p1[g];
        const uint4 q2 = ws.p2[g];
        const uint32_t k = c - start; // the k-th candidate tile of Gaussian g, row-major in its rectangle
        const uint32_t ry = (uint32_t)(((float)k + 0.5f) * ws.inv_w[g]), rx = k - ry * q2.z; // k < 2^11, w <= 2^7: exact
        const uint32_t pre_g = ws.pre[g];
        bool ok = active;
        if (test && active) {
            if (MASKS && EMIT && pre_g - start <= 64u) ok = (ws.mask[g] >> k) & 1ull;
            else ok = tile_can_contribute(q0.x, q0.y, q0.z, q0.w, q1.x, q1.z, q1.w, q1.y, (int)(q2.x + rx), (int)(q2.y + ry), W, H);
        }
        if (EMIT && sizeof(KeyT) == 2) { // binning path: the candidate's own slot, a culled one leaves KEY_NONE (no rank, no bookkeeping)
            const uint32_t pos = q2.w + k;
            if (active && pos < cap) {
                keys[pos] = ok ? (KeyT)((q2.y + ry) * (uint32_t)gx + (q2.x + rx)) : (KeyT)KEY_NONE;
                if (ok) vals[pos] = ws.p3[g].y; // (the value of an empty slot is never looked at)
            }
            continue;
        }
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(ok);
        // this Gaussian's candidates of this trip sit in lanes [seg, seg_end)
        const int seg = (int)max((int)start - (int)t0, 0), seg_end = (int)min(pre_g - t0, 64u);
        const unsigned long long upto = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        const unsigned long long from = (seg == 0) ? ~0ull : ~(~0ull >> (64 - seg));
        const uint32_t before = ws.cnt[g]; // passing tiles of g in earlier trips (read by every lane before any lane of this trip adds)
        __builtin_amdgcn_wave_barrier();
        if (EMIT) {
            const uint32_t pos = q2.w + before + (uint32_t)__builtin_popcountll(bal & upto & from);
            if (ok && pos < cap) {
                const uint2 q3 = ws.p3[g];
                keys[pos] = (KeyT)(q3.x + (q2.y + ry) * (uint32_t)gx + (q2.x + rx));
                vals[pos] = q3.y;
            }
        }
        if (active && lane == seg_end - 1) { // the last lane of the segment books the segment's passing tiles
            const unsigned long long segmask = from & ((seg_end == 64) ? ~0ull : (~0ull >> (64 - seg_end)));
            ws.cnt[g] = before + (uint32_t)__builtin_popcountll(bal & segmask);
            if (MASKS && !EMIT && pre_g - start <= 64u) // the segment's first lane holds candidate max(t0 - start, 0) of the Gaussian
                ws.mask[g] |= ((bal & segmask) >> seg) << (uint32_t)max((int)t0 - (int)start, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    return ws.cnt[lane];
}

// preprocessCUDA, forward.cu:156-257 with in_frustum (auxiliary.h:139-165), computeCov3D (forward.cu:118-152)
// and computeCov2D (forward.cu:74-113) written out as scalar formulas in GLM's evaluation order.
__global__ void __launch_bounds__(256) k_preprocess(const FrameDev* __restrict__ frames, int gx, int gy, int W, int H,
                                                    float* __restrict__ depths, int* __restrict__ radii_all,
                                                    GeomRec* __restrict__ geom, uint32_t* __restrict__ tiles_touched,
                                                    int* __restrict__ err_flag, int cull, int count_exact, uint64_t* __restrict__ gkeys, int idx_bits,
                                                    unsigned long long* __restrict__ surv)
{
#pragma clang fp contract(off)
    const FrameDev& fr = frames[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = idx < fr.P; // no early return: the tile count of large splats below is wave-cooperative
    const size_t g = (size_t)fr.base + (valid ? idx : 0);
    int radius_out = 0;
    uint32_t tiles = 0;
    uint32_t depth_bits = 0xffffffffu; // a Gaussian that emits nothing sorts behind its frame's visible ones (where it sorts changes no list)
    RectJob job = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u, 0u, 0u}; // candidate tiles still to be tested (exact-output culling)
    do {
        if (!valid) break;
        const float* vm = fr.view;
        const float* pm = fr.proj;
        const float p0 = fr.means3D[3 * idx], p1 = fr.means3D[3 * idx + 1], p2 = fr.means3D[3 * idx + 2];
        // in_frustum: only the near/z_threshold test is live (auxiliary.h:155)
        const float pvz = vm[2] * p0 + vm[6] * p1 + vm[10] * p2 + vm[14];
        if (pvz <= fr.z_thr) {
            if (fr.prefiltered) atomicOr(err_flag, 1);
            break;
        }
        const float hx = pm[0] * p0 + pm[4] * p1 + pm[8] * p2 + pm[12];
        const float hy = pm[1] * p0 + pm[5] * p1 + pm[9] * p2 + pm[13];
        const float hw = pm[3] * p0 + pm[7] * p1 + pm[11] * p2 + pm[15];
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float projx = hx * p_w, projy = hy * p_w;

        float c3[6];
        if (fr.cov3D) {
#pragma unroll
            for (int k = 0; k < 6; ++k) c3[k] = fr.cov3D[6 * (size_t)idx + k];
        } else {
            const float sx = fr.scale_mod * fr.scales[3 * idx], sy = fr.scale_mod * fr.scales[3 * idx + 1],
                        sz = fr.scale_mod * fr.scales[3 * idx + 2];
            const float r = fr.rots[4 * idx], x = fr.rots[4 * idx + 1], y = fr.rots[4 * idx + 2], z = fr.rots[4 * idx + 3];
            // columns of M = S * R (GLM column-major), M[c][r] = s_r * R[c][r]
            const float m00 = sx * (1.f - 2.f * (y * y + z * z)), m01 = sy * (2.f * (x * y - r * z)), m02 = sz * (2.f * (x * z + r * y));
            const float m10 = sx * (2.f * (x * y + r * z)), m11 = sy * (1.f - 2.f * (x * x + z * z)), m12 = sz * (2.f * (y * z - r * x));
            const float m20 = sx * (2.f * (x * z - r * y)), m21 = sy * (2.f * (y * z + r * x)), m22 = sz * (1.f - 2.f * (x * x + y * y));
            // Sigma = M^T M : Sigma[c][r] = dot(column r, column c)
            c3[0] = m00 * m00 + m01 * m01 + m02 * m02;
            c3[1] = m10 * m00 + m11 * m01 + m12 * m02;
            c3[2] = m20 * m00 + m21 * m01 + m22 * m02;
            c3[3] = m10 * m10 + m11 * m11 + m12 * m12;
            c3[4] = m20 * m10 + m21 * m11 + m22 * m12;
            c3[5] = m20 * m20 + m21 * m21 + m22 * m22;
        }

        // computeCov2D
        float tx = vm[0] * p0 + vm[4] * p1 + vm[8] * p2 + vm[12];
        float ty = vm[1] * p0 + vm[5] * p1 + vm[9] * p2 + vm[13];
        const float tz = pvz;
        const float limx = 1.3f * fr.tan_fovx, limy = 1.3f * fr.tan_fovy;
        tx = fminf(limx, fmaxf(-limx, tx / tz)) * tz;
        ty = fminf(limy, fmaxf(-limy, ty / tz)) * tz;
        const float J00 = fr.focal_x / tz, J02 = -(fr.focal_x * tx) / (tz * tz);
        const float J11 = fr.focal_y / tz, J12 = -(fr.focal_y * ty) / (tz * tz);
        // T = W * J, T[c][r]; W[k][r] = vm[4*r + k]  (W's columns are the rows of the view rotation)
        float T0[3], T1[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float w0 = vm[4 * r + 0], w1 = vm[4 * r + 1], w2 = vm[4 * r + 2];
            T0[r] = w0 * J00 + w2 * J02;
            T1[r] = w1 * J11 + w2 * J12;
        }
        // V[a][b], symmetric
        const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
        // A = T^T V^T : A[k][r] = T[r][0] V[0][k] + T[r][1] V[1][k] + T[r][2] V[2][k]   (rows r = 0,1 only)
        float A0[3], A1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            A0[k] = T0[0] * V[0][k] + T0[1] * V[1][k] + T0[2] * V[2][k];
            A1[k] = T1[0] * V[0][k] + T1[1] * V[1][k] + T1[2] * V[2][k];
        }
        float cov_xx = A0[0] * T0[0] + A0[1] * T0[1] + A0[2] * T0[2];
        const float cov_xy = A1[0] * T0[0] + A1[1] * T0[1] + A1[2] * T0[2];
        float cov_yy = A1[0] * T1[0] + A1[1] * T1[1] + A1[2] * T1[2];
        cov_xx += 0.3f;
        cov_yy += 0.3f;

        const float det = cov_xx * cov_yy - cov_xy * cov_xy;
        if (det == 0.0f) break;
        const float det_inv = 1.f / det;
        const float ca = cov_yy * det_inv, cb = -cov_xy * det_inv, cc = cov_xx * det_inv;
        const float mid = 0.5f * (cov_xx + cov_yy);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float lambda1 = mid + sq, lambda2 = mid - sq;
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
        const float pix = ndc2pix(projx, W), piy = ndc2pix(projy, H);
        uint32_t x0, y0, x1, y1;
        tile_rect(pix, piy, (int)my_radius, gx, gy, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) break;

        float rgb[3];
        if (fr.colors) {
            rgb[0] = fr.colors[3 * (size_t)idx]; rgb[1] = fr.colors[3 * (size_t)idx + 1]; rgb[2] = fr.colors[3 * (size_t)idx + 2];
        } else {
            sh_to_rgb(idx, fr.D, fr.M, fr.means3D, fr.campos, fr.shs, rgb);
        }
        depths[g] = pvz;
        depth_bits = __float_as_uint(pvz);
        GeomRec rec;
        rec.q0 = make_float4(pix, piy, ca, cb);
        rec.q1 = make_float4(cc, fr.opac[idx], pvz, rgb[0]);
        rec.q2 = make_float4(rgb[1], rgb[2], 0.f, 0.f);
        geom[g] = rec;
        radius_out = (int)my_radius;
        if (cull) cull_rect(pix, piy, ca, cb, cc, fr.opac[idx], gx, gy, x0, y0, x1, y1); // (a Gaussian whose every tile is culled keeps its radius — the reference reports it — but emits nothing)
        tiles = (y1 - y0) * (x1 - x0); // binning path: the emission slots (KEY_NONE); reference mode: the instances
        if (count_exact) { // radix-sort path with culling: the survivors are counted here
            job = {pix, piy, ca, cb, cc, logf(1.0f / (255.0f * fr.opac[idx])), -cb / cc, -cb / ca, x0, y0, x1, y1};
            tiles = 0;
        }
    } while (false);
    __shared__ WalkShared ws_s[4];
    if (count_exact) tiles = rect_walk_flat<false, uint32_t, SURV_MASKS>(ws_s[threadIdx.x >> 6], job, W, H, 0u, 0u, 0u, 0u, nullptr, nullptr, 0, true);
    if (!valid) return;
    if (SURV_MASKS && count_exact) surv[g] = ws_s[threadIdx.x >> 6].mask[threadIdx.x & 63];
    radii_all[g] = radius_out;
    if (fr.radii) fr.radii[idx] = radius_out;
    tiles_touched[g] = tiles;
    // key of the first sort (see below): frame | raw depth bits (unsigned, like the reference's key) | index inside the frame.  Only the upper two
    // fields are sorted on; the index rides in the key's low bits, so the sort moves 8 bytes per Gaussian instead of a key-value pair's 12.
    gkeys[g] = ((uint64_t)blockIdx.y << (32 + idx_bits)) | ((uint64_t)depth_bits << idx_bits) | (uint64_t)(uint32_t)idx;
}

// The reference sorts all instances once by (tile << 32 | depth bits) (rasterizer_impl.cu:70-111, :306-311).  The same
// order comes out of two stable sorts that move 3.5x fewer bytes: first the GAUSSIANS by (frame, depth bits) —
// (frame, depth) keys of k_preprocess — then their instances, emitted in that order, by tile id alone (17 bits for 64 frames x 1200 tiles
// instead of 48-49): inside a tile the stable second sort keeps the depth order, and equal depths keep ascending index.
// (the keys are written by k_preprocess; the scan of the tile counts reads them through the sorted order: TilesInOrder)
struct DepthOrder { // the Gaussians in (frame, depth, index) order: entry i of the sorted keys
    const uint64_t* keys;
    const FrameDev* frames;
    int idx_bits;
    __device__ __forceinline__ uint32_t frame(uint32_t i) const { return (uint32_t)(keys[i] >> (32 + idx_bits)); }
    __device__ __forceinline__ uint32_t gaussian(uint32_t i) const
    {
        const uint64_t k = keys[i];
        return frames[(uint32_t)(k >> (32 + idx_bits))].base + (uint32_t)(k & ((1ull << idx_bits) - 1ull));
    }
};
struct TilesInOrder {
    DepthOrder order;
    const uint32_t* tiles_touched;
    __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return tiles_touched[order.gaussian(i)]; }
};

// duplicateWithKeys, rasterizer_impl.cu:70-111, walking the Gaussians in (frame, depth) order; key = frame-extended tile id.
// Large rectangles (> RECT_SMALL_EMIT tiles) are walked by a whole wavefront.  Big batches (QUEUE = false): in place, by the wavefront
// that owns them — 80 k wavefronts, a few large rectangles each.  Small batches (one environment's frames: a few hundred wavefronts on a
// thousand SIMDs): a wavefront whose 64 Gaussians include many screen-filling splats IS the kernel (0.10 ms for 160 k instances), so
// k_emit_keys only QUEUES them and k_emit_big drains the queue, one rectangle per wavefront, over the whole chip (0.057 ms).  Where an
// instance lands is fixed by the scan of the tile counts, so the drain order does not matter.  (The queue for big batches too: 2.0 - 2.6
// ms instead of 0.40 for the 64-frame benchmark batch — several 100 k appends through one counter, even one atomic per wavefront.)
struct EmitJob { RectJob job; uint32_t off, tile_base, g; };
__device__ __forceinline__ EmitJob emit_job(uint32_t i, int gx, int gy, const DepthOrder& order,
                                            const int* __restrict__ radii_all, const GeomRec* __restrict__ geom,
                                            const uint32_t* __restrict__ offsets, int cull)
{
    EmitJob e = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u, 0u, 0u}, 0u, 0u, 0u};
    e.g = order.gaussian(i);
    const int r = radii_all[e.g];
    if (r > 0) {
        e.off = (i == 0) ? 0u : offsets[i - 1];
        const float4 q0 = geom[e.g].q0;
        const float4 q1 = geom[e.g].q1;
        uint32_t x0, y0, x1, y1;
        tile_rect(q0.x, q0.y, r, gx, gy, x0, y0, x1, y1);
        if (cull) cull_rect(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, gx, gy, x0, y0, x1, y1);
        e.tile_base = order.frame(i) * (uint32_t)(gx * gy);
        e.job = {q0.x, q0.y, q0.z, q0.w, q1.x, cull ? logf(1.0f / (255.0f * q1.y)) : 0.f, -q0.w / q1.x, -q0.w / q0.z, x0, y0, x1, y1};
    }
    return e;
}
template <bool QUEUE, typename KeyT>
__global__ void __launch_bounds__(256) k_emit_keys(const FrameDev* __restrict__ frames, uint32_t G, int gx, int gy, int W, int H,
                                                   const DepthOrder order,
                                                   const int* __restrict__ radii_all, const GeomRec* __restrict__ geom,
                                                   const uint32_t* __restrict__ offsets, KeyT* __restrict__ keys,
                                                   uint32_t* __restrict__ vals, int cull, uint32_t cap, int* __restrict__ overflow,
                                                   uint32_t* __restrict__ big_q, int* __restrict__ big_n, const unsigned long long* __restrict__ surv)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < G;
    if (valid && i == G - 1 && offsets[i] > cap) *overflow = 1; // sync-free mode: the scratch was sized from an earlier batch and is too small
    EmitJob e = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u, 0u, 0u}, 0u, 0u, 0u};
    if (valid) e = emit_job(i, gx, gy, order, radii_all, geom, offsets, cull);
    if (sizeof(KeyT) == 2) e.tile_base = 0u; // 16-bit keys (one-pass binning): the tile inside its frame; the frame is the chunk's
    const bool big = QUEUE && (e.job.x1 - e.job.x0) * (e.job.y1 - e.job.y0) > RECT_SMALL_EMIT;
    const unsigned long long bm = __builtin_amdgcn_ballot_w64(big);
    if (QUEUE && bm) { // one atomic per wavefront
        const int lane = (int)(threadIdx.x & 63), leader = __builtin_ctzll(bm);
        int base = 0;
        if (lane == leader) base = atomicAdd(big_n, __builtin_popcountll(bm));
        base = __builtin_amdgcn_readlane(base, leader);
        if (big) {
            big_q[base + __builtin_popcountll(bm & ((1ull << lane) - 1ull))] = i;
            e.job.x1 = e.job.x0; // nothing left for this lane here
        }
    }
    __shared__ WalkShared ws_s[4];
    unsigned long long mask = 0ull;
    if (SURV_MASKS && cull && valid) mask = surv[e.g];
    (void)rect_walk_flat<true, KeyT, SURV_MASKS>(ws_s[threadIdx.x >> 6], e.job, W, H, e.off, cap, e.tile_base, e.g, keys, vals, gx, cull != 0, mask);
}
template <typename KeyT>
__global__ void __launch_bounds__(256) k_emit_big(uint32_t G, int gx, int gy, int W, int H, const DepthOrder order, const int* __restrict__ radii_all,
                                                  const GeomRec* __restrict__ geom, const uint32_t* __restrict__ offsets,
                                                  KeyT* __restrict__ keys, uint32_t* __restrict__ vals, int cull, uint32_t cap,
                                                  const uint32_t* __restrict__ big_q, const int* __restrict__ big_n)
{
    const int n = *big_n;
    const int lane = (int)(threadIdx.x & 63);
    const int waves = (int)(gridDim.x * (blockDim.x >> 6));
    for (int q = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); q < n; q += waves) { // wave-uniform: one rectangle per trip
        EmitJob e = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0u, 0u, 0u, 0u}, 0u, 0u, 0u};
        if (lane == 0) e = emit_job(big_q[q], gx, gy, order, radii_all, geom, offsets, cull); // rect_walk broadcasts lane 0's rectangle
        if (sizeof(KeyT) == 2) e.tile_base = 0u;
        (void)rect_walk<true, KeyT>(e.job, W, H, e.off, cap, e.tile_base, e.g, keys, vals, gx, cull != 0);
    }
}

// identifyTileRanges, rasterizer_impl.cu:116-138.
// L comes from device memory (the last scan element, clamped to the scratch capacity): the sync-free mode launches it over
// the capacity without knowing the count on the host.
__global__ void __launch_bounds__(256) k_tile_ranges(const uint32_t* __restrict__ last_offset, uint32_t cap, const uint32_t* __restrict__ keys,
                                                     uint2* __restrict__ ranges)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t L = min(*last_offset, cap);
    if (i >= L) return;
    const uint32_t cur = keys[i];
    if (i == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = keys[i - 1];
        if (cur != prev) {
            ranges[prev].y = i;
            ranges[cur].x = i;
        }
    }
    if (i == L - 1) ranges[cur].y = L;
}

// ---- tile binning in ONE pass (round 6) -----------------------------------------------------------------------------------------------------
// Until round 6 the emitted (tile key, Gaussian) pairs went through rocPRIM's Onesweep radix sort — two passes of 9 bits over 35 M pairs,
// 0.61 - 0.63 ms per 64 frames at 0.22 of the HBM peak, the front end's largest stage — and k_tile_ranges then read the sorted keys back to find
// where each tile's list starts.  The emission order already is (frame, depth, index): what is left to do is a STABLE partition by tile id, and
// within a frame there are only gx * gy tiles (1 200 at 640 x 480: one 11-bit digit).  So the list is cut into chunks of BIN_CHUNK instances that
// never straddle a frame, and the partition is a counting sort with the (frame, tile) pair as its single digit:
//   k_bin_plan     frame boundaries of the emission order (from the scan of the tile counts) -> chunk table {frame, first instance, length}
//   k_bin_hist     per chunk: LDS histogram over the frame's tiles -> one row of hist[chunk][tile]
//   k_bin_colscan  per (frame, tile): exclusive prefix over the frame's chunks (in place) + the tile's total
//   k_tile_starts  exclusive scan of the totals over all (frame, tile) = the ranges array of identifyTileRanges (rasterizer_impl.cu:116-138),
//                  without reading a key
//   k_bin_scatter  per chunk: every wavefront ranks its 64 x BIN_STEPS consecutive instances — peers of a lane (same tile, same step) from one
//                  ballot per key bit, earlier steps from the wavefront's own LDS counters — the counters of the eight wavefronts are chained
//                  behind the chunk's prefix, and the Gaussian indices go to  start(tile) + prefix(chunk, tile) + rank.
// Every instance keeps its emission order inside its tile: point_list is the one the two sorts of the reference produce (and the one the
// rocPRIM path produced: tests/test_raster_gpu.py compares it with the oracle's).  Keys are read twice and never written again; the values are
// written once.  Frames with more than BIN_MAX_TILES tiles (or more than BIN_MAX_FRAMES frames) keep the radix sort.
constexpr int BIN_THREADS = 512, BIN_STEPS = 16, BIN_CHUNK = BIN_THREADS * BIN_STEPS;
constexpr int BIN_MAX_TILES = 2048, BIN_MAX_FRAMES = 1024;

__global__ void __launch_bounds__(1024) k_bin_plan(const FrameDev* __restrict__ frames, int F, const uint32_t* __restrict__ offsets, uint32_t cap,
                                                   uint32_t nb_max, uint4* __restrict__ desc, uint32_t* __restrict__ chunk_first)
{
    __shared__ uint32_t s_end[BIN_MAX_FRAMES + 1], s_first[BIN_MAX_FRAMES + 1];
    const int tid = (int)threadIdx.x;
    // frame f's Gaussians sit at [base, base + P) of the (frame, depth) order: its instances end where the scan says (clamped to the capacity of
    // the sync-free mode, like the emission)
    if (tid < F) { const FrameDev& fr = frames[tid]; s_end[tid + 1] = fr.P > 0 ? min(offsets[(size_t)fr.base + fr.P - 1], cap) : 0u; }
    if (tid == 0) s_end[0] = 0u;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0, nb = 0;
        for (int f = 0; f < F; ++f) {
            run = max(run, s_end[f + 1]); // a frame without Gaussians ends where its predecessor ends
            s_end[f + 1] = run;
            s_first[f] = nb;
            nb += (run - s_end[f] + BIN_CHUNK - 1) / BIN_CHUNK;
        }
        s_first[F] = nb;
    }
    __syncthreads();
    for (int f = tid; f <= F; f += 1024) chunk_first[f] = s_first[f];
    for (int f = tid; f < F; f += 1024) {
        const uint32_t start = s_end[f], cnt = s_end[f + 1] - start, first = s_first[f];
        for (uint32_t k = 0; k * BIN_CHUNK < cnt; ++k) desc[first + k] = make_uint4((uint32_t)f, start + k * BIN_CHUNK, min((uint32_t)BIN_CHUNK, cnt - k * BIN_CHUNK), 0u);
    }
    for (uint32_t c = s_first[F] + (uint32_t)tid; c < nb_max; c += 1024) desc[c] = make_uint4(0u, 0u, 0u, 0u); // workgroups past the last chunk leave
}

__global__ void __launch_bounds__(BIN_THREADS) k_bin_hist(const uint4* __restrict__ desc, int tiles, const uint16_t* __restrict__ keys, uint32_t* __restrict__ hist)
{
    extern __shared__ uint32_t s_bin[];
    const uint4 d = desc[blockIdx.x];
    if (d.z == 0) return;
    const int tid = (int)threadIdx.x;
    for (int t = tid; t < tiles; t += BIN_THREADS) s_bin[t] = 0u;
    __syncthreads();
    const uint16_t* k = keys + d.y;
    uint32_t kk[BIN_STEPS]; // the chunk's keys of this lane, all loads in flight
#pragma unroll
    for (int s = 0; s < BIN_STEPS; ++s) { const uint32_t i = (uint32_t)(s * BIN_THREADS + tid); kk[s] = i < d.z ? (uint32_t)k[i] : 0xffffffffu; }
#pragma unroll
    for (int s = 0; s < BIN_STEPS; ++s) if (kk[s] < (uint32_t)tiles) atomicAdd(&s_bin[kk[s]], 1u); // (KEY_NONE: a culled candidate's slot)
    __syncthreads();
    uint32_t* row = hist + (size_t)blockIdx.x * tiles;
    for (int t = tid; t < tiles; t += BIN_THREADS) row[t] = s_bin[t];
}

__global__ void __launch_bounds__(256) k_bin_colscan(int F, int tiles, const uint32_t* __restrict__ chunk_first, uint32_t* __restrict__ hist, uint32_t* __restrict__ totals,
                                                     uint32_t* __restrict__ local_start, uint32_t* __restrict__ group_sum)
{
    const uint32_t ft = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t run = 0;
    if (ft < (uint32_t)F * (uint32_t)tiles) {
        const uint32_t f = ft / (uint32_t)tiles, t = ft - f * (uint32_t)tiles;
        const uint32_t c0 = chunk_first[f], c1 = chunk_first[f + 1];
        uint32_t c = c0;
        for (; c + 8 <= c1; c += 8) { // eight rows in flight
            uint32_t v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = hist[(size_t)(c + k) * tiles + t];
#pragma unroll
            for (int k = 0; k < 8; ++k) { hist[(size_t)(c + k) * tiles + t] = run; run += v[k]; }
        }
        for (; c < c1; ++c) { const uint32_t v = hist[(size_t)c * tiles + t]; hist[(size_t)c * tiles + t] = run; run += v; }
        totals[ft] = run;
    }
    // the tile's list starts behind the lists of the 256 (frame, tile) pairs of this workgroup before it (here) and of the workgroups before (k_tile_starts)
    __shared__ uint32_t s_wave[4];
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    uint32_t incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = incl - run;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
    if (ft < (uint32_t)F * (uint32_t)tiles) local_start[ft] = base;
    if (threadIdx.x == 255) group_sum[blockIdx.x] = base + run;
}

// ranges[ft] = [start, start + total) of every (frame, tile) with instances, {0, 0} for the others (what the reference's zero-filled ranges hold for a
// tile no key names): the scan of the totals over all (frame, tile) pairs, finished per group of 256 — a group adds up the few hundred sums before it.
constexpr int TILE_ORDER_BITS = 12; // length classes of the compositor's workgroup order (k_tile_order): four instances per class, 16 380+ in the first
__device__ __forceinline__ uint32_t tile_len_class(uint32_t n) { return ((1u << TILE_ORDER_BITS) - 1u) - min(n >> 2, (1u << TILE_ORDER_BITS) - 1u); }
// One more list of class `c` (valid lanes): counted in the workgroup's LDS first — most tiles of a batch are empty or short and share a handful of
// classes, and 76 800 atomics on a handful of addresses take 0.18 ms — then one global atomic per class the workgroup has seen.
__device__ __forceinline__ void tile_class_count(uint32_t* s_cls, bool valid, uint32_t c, uint32_t* __restrict__ cls_count)
{
    constexpr uint32_t NC = 1u << TILE_ORDER_BITS;
    for (uint32_t k = threadIdx.x; k < NC; k += blockDim.x) s_cls[k] = 0u;
    __syncthreads();
    const bool first = valid && atomicAdd(&s_cls[c], 1u) == 0u;
    __syncthreads();
    if (first) atomicAdd(&cls_count[c], s_cls[c]);
}

__global__ void __launch_bounds__(256) k_tile_starts(uint32_t FT, const uint32_t* __restrict__ totals, const uint32_t* __restrict__ local_start,
                                                     const uint32_t* __restrict__ group_sum, uint2* __restrict__ ranges, uint32_t* __restrict__ cls_count,
                                                     uint32_t tiles, uint32_t* __restrict__ frame_start)
{
    __shared__ uint32_t s_part[256];
    uint32_t part = 0;
    for (uint32_t g = threadIdx.x; g < blockIdx.x; g += 256) part += group_sum[g];
    s_part[threadIdx.x] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_part[threadIdx.x] += s_part[threadIdx.x + o];
        __syncthreads();
    }
    const uint32_t ft = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n = 0;
    if (ft < FT) {
        n = totals[ft];
        const uint32_t st = s_part[0] + local_start[ft];
        ranges[ft] = n ? make_uint2(st, st + n) : make_uint2(0u, 0u);
        // instances before each frame and, behind the last, of the whole batch: the instance COUNT on the binning path (the scan of the
        // rectangles is only the capacity of the emission slots)
        if (ft % tiles == 0u) frame_start[ft / tiles] = st;
        if (ft == FT - 1u) frame_start[FT / tiles] = st + n;
    }
    __shared__ uint32_t s_cls[1u << TILE_ORDER_BITS];
    if (cls_count) tile_class_count(s_cls, ft < FT, tile_len_class(n), cls_count);
}

#ifdef R2S_BIN_WAVES
#define R2S_BIN_OCC __attribute__((amdgpu_waves_per_eu(R2S_BIN_WAVES, R2S_BIN_WAVES)))
#else
#define R2S_BIN_OCC
#endif
template <int KEY_BITS>
__global__ void __launch_bounds__(BIN_THREADS) R2S_BIN_OCC k_bin_scatter(const uint4* __restrict__ desc, int tiles, const uint16_t* __restrict__ keys,
                                                             const uint32_t* __restrict__ vals, const uint32_t* __restrict__ hist,
                                                             const uint2* __restrict__ ranges, uint32_t* __restrict__ out)
{
    extern __shared__ uint32_t s_bin[]; // [wavefront][tile]: instances of the tile this wavefront has ranked so far; then: where its next one goes
    const uint4 d = desc[blockIdx.x];
    if (d.z == 0) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int WAVES = BIN_THREADS / 64;
    const uint32_t kb = d.x * (uint32_t)tiles;
    // this lane's instances: 64 consecutive ones per step and wavefront, all loads in flight before the first is needed
    uint32_t pk[BIN_STEPS];
#ifndef R2S_BIN_LATE_VALS
    uint32_t vv[BIN_STEPS];
#endif
#pragma unroll
    for (int s = 0; s < BIN_STEPS; ++s) {
        const uint32_t i = (uint32_t)(wave * (64 * BIN_STEPS) + s * 64 + lane);
        const uint32_t key = i < d.z ? (uint32_t)keys[d.y + i] : KEY_NONE;
        const bool ok = key < (uint32_t)tiles; // (KEY_NONE: a culled candidate's slot)
        pk[s] = ok ? key : 0x80000000u;
#ifndef R2S_BIN_LATE_VALS
        vv[s] = i < d.z ? vals[d.y + i] : 0u;
#endif
    }
    for (int t = tid; t < WAVES * tiles; t += BIN_THREADS) s_bin[t] = 0u;
    // the peers of every instance — lanes of its step with the same tile — from one ballot per key bit: the sixteen steps are independent chains
    // (no LDS, no barrier: the compiler interleaves them).  pk = key | rank among the peers << 11 | number of peers << 17 | last peer << 24 | invalid << 31
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int s = 0; s < BIN_STEPS; ++s) {
        const bool ok = !(pk[s] >> 31);
        const uint32_t key = pk[s] & 0x7ffu;
        uint32_t plo = 0xffffffffu, phi = 0xffffffffu;
#pragma unroll
        for (int b = 0; b < KEY_BITS; ++b) {
            const int m = __builtin_amdgcn_sbfe((int)key, b, 1); // 0 or ~0
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(m != 0);
            plo &= ~((uint32_t)bal ^ (uint32_t)m);               // bit set: bal, clear: ~bal
            phi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)m);
        }
        unsigned long long peers = (((unsigned long long)phi << 32) | plo) & __builtin_amdgcn_ballot_w64(ok);
        if (!ok) peers = 0ull;
        const uint32_t r = (uint32_t)__builtin_popcountll(peers & lt), n = (uint32_t)__builtin_popcountll(peers);
        pk[s] = (pk[s] & 0x800007ffu) | (r << 11) | (n << 17) | ((peers >> lane) == 1ull ? 1u << 24 : 0u);
#ifdef R2S_BIN_GROUP // experiment: at most R2S_BIN_GROUP chains interleaved (registers)
        if ((s + 1) % R2S_BIN_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
#endif
    }
    __syncthreads();
    uint32_t* wh = s_bin + wave * tiles;
#pragma unroll
    for (int s = 0; s < BIN_STEPS; ++s) {
        const bool ok = !(pk[s] >> 31);
        const uint32_t key = pk[s] & 0x7ffu;
        const uint32_t before = wh[key]; // read by every peer before the last of them books the step (one wavefront: its LDS operations stay in order)
        __builtin_amdgcn_wave_barrier();
        if (ok && ((pk[s] >> 24) & 1u)) wh[key] = before + ((pk[s] >> 17) & 0x7fu);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        pk[s] = (pk[s] & 0x800007ffu) | ((before + ((pk[s] >> 11) & 0x3fu)) << 11); // rank among the wavefront's instances of the tile
    }
    __syncthreads();
    const uint32_t* row = hist + (size_t)blockIdx.x * tiles;
    for (int t = tid; t < tiles; t += BIN_THREADS) { // the tile's list start + the chunks before this one + the wavefronts before each
        uint32_t run = ranges[kb + t].x + row[t];
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { const uint32_t n = s_bin[w * tiles + t]; s_bin[w * tiles + t] = run; run += n; }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < BIN_STEPS; ++s)
#ifdef R2S_BIN_LATE_VALS // experiment: the values loaded behind the ranking (16 registers less)
        if (!(pk[s] >> 31)) out[wh[pk[s] & 0x7ffu] + ((pk[s] >> 11) & 0xfffffu)] = vals[d.y + (uint32_t)(wave * (64 * BIN_STEPS) + s * 64 + lane)];
#else
        if (!(pk[s] >> 31)) out[wh[pk[s] & 0x7ffu] + ((pk[s] >> 11) & 0xfffffu)] = vv[s];
#endif
}

// sync-free mode: the tail of the key array (instances the batch did not produce) sorts behind every real tile
__global__ void __launch_bounds__(256) k_fill_sentinel(const uint32_t* __restrict__ last_offset, uint32_t cap, uint32_t sentinel, uint32_t* __restrict__ keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap && i >= *last_offset) keys[i] = sentinel;
}

// k_bin_scatter with the chunk staged TILE-MAJOR through LDS before it leaves (round 6, experiment behind R2S_BIN_STAGED=1): the direct form
// stores every instance where its rank says — 64 lanes, 64 places — and 0.10 of its 0.22 ms is that scatter.  Here the ranks first place the
// values in an LDS copy of the chunk ordered by tile (chunk-local prefix over the tiles + the wavefront's offset inside the tile + rank),
// and consecutive lanes then store consecutive entries of that copy: a tile's run of the chunk (a hundred instances for the tiles under the
// object) leaves as whole lines.  16-bit wavefront counters, 74 KB of LDS per workgroup at 1 200 tiles (two workgroups per CU: the same
// four wavefronts per SIMD the registers allow); frames of more than BIN_STAGED_MAX_TILES tiles take the direct form.
constexpr int BIN_STAGED_MAX_TILES = 1400;
template <int KEY_BITS>
__global__ void __launch_bounds__(BIN_THREADS) k_bin_scatter_staged(const uint4* __restrict__ desc, int tiles, const uint16_t* __restrict__ keys,
                                                                    const uint32_t* __restrict__ vals, const uint32_t* __restrict__ hist,
                                                                    const uint2* __restrict__ ranges, uint32_t* __restrict__ out)
{
    extern __shared__ uint32_t s_raw[];
    constexpr int WAVES = BIN_THREADS / 64;
    uint32_t* s_val = s_raw;                                          // [BIN_CHUNK] the chunk's values, tile-major
    uint32_t* s_gb = s_val + BIN_CHUNK;                               // [tiles] where the chunk's run of the tile starts in point_list
    unsigned short* s_tile = (unsigned short*)(s_gb + tiles);         // [BIN_CHUNK] the tile of every staged value
    unsigned short* s_lpre = s_tile + BIN_CHUNK;                      // [tiles + 1] chunk-local exclusive prefix of the tile counts
    unsigned short* s_cnt = s_lpre + ((tiles + 2) & ~1);              // [WAVES][tiles] per-wavefront counters; then: the wavefront's offset inside the tile
    __shared__ uint32_t s_wsum[WAVES];
    const uint4 d = desc[blockIdx.x];
    if (d.z == 0) return;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t kb = d.x * (uint32_t)tiles;
    uint32_t pk[BIN_STEPS], vv[BIN_STEPS];
#pragma unroll
    for (int s = 0; s < BIN_STEPS; ++s) {
        const uint32_t i = (uint32_t)(wave * (64 * BIN_STEPS) + s * 64 + lane);
        const uint32_t key = i < d.z ? (uint32_t)keys[d.y + i] : KEY_NONE;
        pk[s] = key < (uint32_t)tiles ? key : 0x80000000u;
        vv[s] = i < d.z ? vals[d.y + i] : 0u;
    }
    for (int t = tid; t < WAVES * tiles; t += BIN_THREADS) s_cnt[t] = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int s = 0; s < BIN_STEPS; ++s) { // peers of every slot (see k_bin_scatter)
        const bool ok = !(pk[s] >> 31);
        const uint32_t key = pk[s] & 0x7ffu;
        uint32_t plo = 0xffffffffu, phi = 0xffffffffu;
#pragma unroll
        for (int b = 0; b < KEY_BITS; ++b) {
            const int m = __builtin_amdgcn_sbfe((int)key, b, 1);
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(m != 0);
            plo &= ~((uint32_t)bal ^ (uint32_t)m);
            phi &= ~((uint32_t)(bal >> 32) ^ (uint32_t)m);
        }
        unsigned long long peers = (((unsigned long long)phi << 32) | plo) & __builtin_amdgcn_ballot_w64(ok);
        if (!ok) peers = 0ull;
        const uint32_t r = (uint32_t)__builtin_popcountll(peers & lt), n = (uint32_t)__builtin_popcountll(peers);
        pk[s] = (pk[s] & 0x800007ffu) | (r << 11) | (n << 17) | ((peers >> lane) == 1ull ? 1u << 24 : 0u);
    }
    __syncthreads();
    unsigned short* wh = s_cnt + wave * tiles;
#pragma unroll
    for (int s = 0; s < BIN_STEPS; ++s) {
        const bool ok = !(pk[s] >> 31);
        const uint32_t key = pk[s] & 0x7ffu;
        const uint32_t before = wh[key];
        __builtin_amdgcn_wave_barrier();
        if (ok && ((pk[s] >> 24) & 1u)) wh[key] = (unsigned short)(before + ((pk[s] >> 17) & 0x7fu));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        pk[s] = (pk[s] & 0x800007ffu) | ((before + ((pk[s] >> 11) & 0x3fu)) << 11);
    }
    __syncthreads();
    // per tile: the wavefronts' offsets inside the tile's run; over the tiles: where the run starts in the staged chunk (consecutive tiles per thread)
    const int per = (tiles + BIN_THREADS - 1) / BIN_THREADS, t0 = tid * per, t1 = min(t0 + per, tiles);
    const uint32_t* row = hist + (size_t)blockIdx.x * tiles;
    uint32_t mine = 0;
    for (int t = t0; t < t1; ++t) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { const uint32_t n = s_cnt[w * tiles + t]; s_cnt[w * tiles + t] = (unsigned short)run; run += n; }
        s_gb[t] = ranges[kb + t].x + row[t];
        s_lpre[t] = (unsigned short)run; // the tile's count for now
        mine += run;
    }
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    for (int t = t0; t < t1; ++t) { const uint32_t n = s_lpre[t]; s_lpre[t] = (unsigned short)base; base += n; }
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) total += s_wsum[w];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < BIN_STEPS; ++s)
        if (!(pk[s] >> 31)) {
            const uint32_t key = pk[s] & 0x7ffu;
            const uint32_t lp = (uint32_t)s_lpre[key] + (uint32_t)wh[key] + ((pk[s] >> 11) & 0xfffffu);
            s_val[lp] = vv[s];
            s_tile[lp] = (unsigned short)key;
        }
    __syncthreads();
    for (uint32_t j = (uint32_t)tid; j < total; j += BIN_THREADS) {
        const uint32_t t = s_tile[j];
        out[s_gb[t] + (j - (uint32_t)s_lpre[t])] = s_val[j];
    }
}

// Workgroup order of the compositor: tiles sorted by the length of their instance list, longest first.  On the benchmark
// scene 5 % of the tiles (the object region) hold 85 % of the instances; started in tile order, the deep tiles of the last
// frames run alone at the end of the kernel (1.26 ms); started first, the short ones fill the gaps (0.97 ms).
// (Until round 6: a key per tile + a two-pass rocPRIM sort of 76 800 pairs — five launches, 45 us.  The order only has to be by length CLASS and
// nothing depends on the order inside a class, so it is a counting sort: the class counts come with the ranges (k_tile_starts; k_tile_classes
// behind the radix-sort path), every workgroup scans them for itself, and the slots inside a class are handed out by atomics.  Which of two
// equally long lists starts first may differ from run to run; no pixel depends on it.)
__global__ void __launch_bounds__(256) k_tile_classes(uint32_t n, const uint2* __restrict__ ranges, uint32_t* __restrict__ cls_count)
{
    __shared__ uint32_t s_cls[1u << TILE_ORDER_BITS];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint2 r = make_uint2(0u, 0u);
    if (i < n) r = ranges[i];
    tile_class_count(s_cls, i < n, tile_len_class(r.y - r.x), cls_count);
}
__global__ void __launch_bounds__(1024) k_tile_order(uint32_t n, const uint2* __restrict__ ranges, const uint32_t* __restrict__ cls_count,
                                                     uint32_t* __restrict__ cls_cursor, uint32_t* __restrict__ order)
{
    constexpr uint32_t NC = 1u << TILE_ORDER_BITS;
    __shared__ uint32_t s_start[NC]; // where the class starts in the order
    __shared__ uint32_t s_cnt[NC];   // lists of the class in this workgroup; then: the workgroup's first slot inside the class
    __shared__ uint32_t s_wave[16];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint2 r = make_uint2(0u, 0u);
    if (i < n) r = ranges[i];
    const uint32_t c = tile_len_class(r.y - r.x);
    // exclusive scan of the class counts: 4 consecutive classes per thread
    constexpr int PER = (int)(NC / 1024);
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { v[k] = cls_count[tid * PER + k]; sum += v[k]; s_cnt[tid * PER + k] = 0u; }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t base = incl - sum;
    for (int w = 0; w < wave; ++w) base += s_wave[w];
#pragma unroll
    for (int k = 0; k < PER; ++k) { s_start[tid * PER + k] = base; base += v[k]; }
    const uint32_t local = i < n ? atomicAdd(&s_cnt[c], 1u) : 1u;
    __syncthreads();
    if (local == 0u) s_cnt[c] = atomicAdd(&cls_cursor[c], s_cnt[c]); // one global atomic per class and workgroup
    __syncthreads();
    if (i < n) order[s_start[c] + s_cnt[c] + local] = i;
}

// renderCUDA, forward.cu:262-394.  One workgroup per 16x16 tile; wavefront w owns the 8x8 quadrant
// (w&1, w>>1) so that a whole-wave skip (all 64 pixels fail the alpha test) is likely for small splats.
// The per-pixel sequence of DECISIONS (power -> alpha -> test_T stop -> colour -> median depth) is forward.cu:339-380's; the
// arithmetic is not the reference's to the bit: the conic is pre-scaled by log2(e) once per staged instance, the quadratic is an
// FMA chain and the exponential is v_exp_f32 (2^x) — parity is by tolerance.  rgb and depth travel through LDS
// with the rest of the record instead of being re-read from global memory inside the pixel loop (forward.cu:362).
#if defined(R2S_COMP_STATS) && !defined(R2S_COMP_CXX)
#define R2S_COMP_CXX // the lane statistics are taken inside the C++ form of the blend
#endif
#ifdef R2S_COMP_STATS // instrumented build (scratch/comp_stats.py): lane efficiency of the compositor
__device__ unsigned long long g_comp_stats[4]; // wave iterations, hit lanes, iterations without a hit, lanes still alive
extern "C" int r2s_raster_debug_comp_stats(unsigned long long* out, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_comp_stats), 32);
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_comp_stats), z, 32); }
    return rc;
}
#endif
template <bool AUX>
__global__ void __launch_bounds__(TILE_THREADS) k_composite(const FrameDev* __restrict__ frames, int gx, int gy, int W, int H,
                                                            const uint2* __restrict__ ranges,
                                                            const uint32_t* __restrict__ point_list,
                                                            const GeomRec* __restrict__ geom, float* __restrict__ aux_T,
                                                            uint32_t* __restrict__ aux_n, const uint32_t* __restrict__ tile_order)
{
    const int tiles = gx * gy;
    const uint32_t ft = tile_order ? tile_order[blockIdx.x] : blockIdx.x; // longest instance lists first
    const int f = (int)(ft / (uint32_t)tiles);
    const int t = (int)ft - f * tiles;
    const int ty = t / gx, tx = t - ty * gx;
    const FrameDev& fr = frames[f];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int px = tx * TILE + (wave & 1) * 8 + (lane & 7);
    const int py = ty * TILE + (wave >> 1) * 8 + (lane >> 3);
    const bool inside = px < W && py < H;
    const float pfx = (float)px, pfy = (float)py;
    bool done = !inside;
#ifndef R2S_COMP_CXX
    unsigned done_i = inside ? 0u : 1u; // the asm blend keeps the flag in a VGPR (its conditions live in EXEC, not in SGPR masks)
#endif

    const uint2 range = ranges[ft];
    const int n = (int)(range.y - range.x);
    const int rounds = (n + TILE_THREADS - 1) / TILE_THREADS;

    // one 48-byte record per staged instance: (px, py, A, B) | (C, opacity, -, -) | (depth, r, g, b) — conic pre-scaled, see
    // below.  ONE address register serves the three broadcast reads (the instance index is scalar: every extra LDS address
    // is a v_mov), and (g, b) land in an even register pair, so the compiler's packed FMA needs no operand shuffles.
    // s_live[q][w]: which of the 64 instances staged by wavefront w can reach alpha >= 1/255 somewhere in quadrant q
    // (the same conservative bound as the tile culling, on the 8x8 pixel rectangle).  A quadrant's wavefront walks
    // only its set bits, so an instance costs nothing in the quadrants it cannot touch.
    // (Round 3 measured the stage double-buffered — round i + 1 staged BEFORE round i is consumed, one barrier per round instead of
    // two, so that a fast quadrant stages ahead while a slow one still blends: 25 KB of LDS, 6 instead of 8 wavefronts per SIMD,
    // 1.14 vs 0.98 ms per 64 frames on the benchmark scene, images identical.  The barriers are not what the kernel waits for.)
    constexpr int NBUF = 1;
    __shared__ float4 s_rec[NBUF][TILE_THREADS * 3];
    __shared__ unsigned long long s_live[NBUF][4][4];

    float T = 1.0f;
    uint32_t last_contributor = 0;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f;
    float D = 15.0f; // forward.cu:309

    auto stage = [&](int i, int buf) {
        const int progress = i * TILE_THREADS + tid;
        unsigned live4 = 0;
        if (progress < n) {
            const uint32_t id = point_list[range.x + progress];
            const GeomRec* rec = geom + id;
            const float4 a = rec->q0, b = rec->q1, c = rec->q2;
            // conic pre-scaled once per staged instance: power * log2(e) = A dx^2 + B dx dy + C dy^2 with
            // A = -0.5 a log2e, B = -b log2e, C = -0.5 c log2e, so the pixel loop is 7 plain VALU ops + one v_exp_f32
            constexpr float LOG2E = 1.4426950408889634f;
            s_rec[buf][3 * tid] = make_float4(a.x, a.y, -0.5f * LOG2E * a.z, -LOG2E * a.w);
            s_rec[buf][3 * tid + 1] = make_float4(-0.5f * LOG2E * b.x, b.y, 0.f, 0.f);
            s_rec[buf][3 * tid + 2] = make_float4(b.z, b.w, c.x, c.y);
            const float lt = logf(1.0f / (255.0f * b.y)), rdy = -a.w / b.x, rdx = -a.w / a.z;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float x_lo = (float)(tx * TILE + (qd & 1) * 8), y_lo = (float)(ty * TILE + (qd >> 1) * 8);
                live4 |= rect_can_contribute(a.x, a.y, a.z, a.w, b.x, rdy, rdx, lt, x_lo, fminf(x_lo + 7.f, (float)(W - 1)), y_lo,
                                             fminf(y_lo + 7.f, (float)(H - 1))) ? (1u << qd) : 0u;
            }
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const unsigned long long bal = __builtin_amdgcn_ballot_w64((live4 >> qd) & 1u);
            if (lane == 0) s_live[buf][qd][wave] = bal;
        }
    };
    auto consume = [&](int i, int buf) {
        const uint32_t base = (uint32_t)(i * TILE_THREADS);
        for (int sw = 0; sw < 4; ++sw) {
#ifndef R2S_COMP_CXX
            done = done_i != 0;
#endif
            if (__builtin_amdgcn_ballot_w64(!done) == 0) break; // whole quadrant finished (forward.cu:315 per block)
            // the live word is wave-uniform: keep it in SGPRs so the walk is s_ff1 / s_andn2 and a scalar branch
            const unsigned long long lv = s_live[buf][wave][sw];
            unsigned long long bits = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(lv >> 32)) << 32) |
                                      (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)lv);
            // The walk is software-pipelined by hand, two register sets alternating: the three LDS reads of the NEXT live instance are
            // issued before the current one is blended.  With eight wavefronts per SIMD (the 32-environment batches) other wavefronts
            // cover the ~100 cycles between a record's reads and its first use; a single environment's frames leave the deepest tile's
            // wavefronts alone on their SIMDs, and the serial read -> blend -> read chain WAS the kernel: 0.53 ms for two frames whose
            // longest list has 3 326 instances (0.54 -> 0.47 ms; 0.92 -> 0.89 ms per 64 frames of the benchmark).  (Taking the live instances in
            // PAIRS on top — the two alphas computed side by side behind scheduling barriers, then the two blends in list order, bit-identical
            // images, 18 more registers, a variant for batches under 4096 tiles — was measured too: 0.466 -> 0.453 ms.  The lone wavefront issues
            // ~one instruction per 8 cycles whatever the order: 22 VALU + 15 SALU per instance with mask results crossing between the two.)
            auto fetch = [&](int j, float4& a, float2& b, float4& c) {
                const float4* rec = s_rec[buf] + 3 * j;
                a = rec[0];
                b = make_float2(rec[1].x, rec[1].y);
                c = rec[2]; // depth, r, g, b — issued with the other two reads (one address register, no second v_mov)
            };
#ifndef R2S_COMP_CXX
            // The blend with its predicates in EXEC (round 3; -DR2S_COMP_CXX builds the C++ form below instead, which is the specification:
            // forward.cu:339-380's sequence of decisions).  v_cmpx narrows the active lanes — pixel not done -> power <= 0 -> alpha >= 1/255 ->
            // not ending —, the arithmetic runs on the lanes left, and the state of every other lane is simply not touched: instead of the
            // compiler's three compares into SGPR masks, six scalar mask operations and four selects per instance, whose VALU -> SALU -> VALU
            // hand-overs a lone wavefront pays in full.  Operation for operation the arithmetic of the C++ form (same products, same FMAs,
            // same operand values; (C1, C2) as two FMAs instead of one packed FMA), so the images are bit-identical (sha256 of the benchmark
            // batch and of the single-environment frames compared between the two builds).  0.88 -> 0.76 ms per 64 frames, 0.47 -> 0.37 ms
            // for one environment's two frames.
#define R2S_BLEND_ASM(EXTRA)                                                                                                            \
                    "s_mov_b64 %[sv], exec\n\t"                                                                                           \
                    "v_cmpx_eq_u32_e32 0, %[done]\n\t"                                                                                    \
                    "v_sub_f32_e32 %[t0], %[ax], %[pfx]\n\t"                                                                              \
                    "v_sub_f32_e32 %[t1], %[ay], %[pfy]\n\t"                                                                              \
                    "v_mul_f32_e32 %[t2], %[aw], %[t1]\n\t"                                                                               \
                    "v_mul_f32_e32 %[t3], %[t1], %[bx]\n\t"                                                                               \
                    "v_fmac_f32_e32 %[t2], %[az], %[t0]\n\t"                                                                              \
                    "v_mul_f32_e32 %[t3], %[t1], %[t3]\n\t"                                                                               \
                    "v_fmac_f32_e32 %[t3], %[t0], %[t2]\n\t" /* power (x log2 e) */                                                        \
                    "v_exp_f32_e32 %[t2], %[t3]\n\t"                                                                                      \
                    "v_cmpx_nlt_f32_e32 0, %[t3]\n\t"        /* keep !(power > 0); also the wait state a trans result needs before a */    \
                    "s_nop 0\n\t"                            /* non-trans consumer (lanes it drops computed an exponential nobody reads) */ \
                    "v_mul_f32_e32 %[t2], %[by], %[t2]\n\t"                                                                               \
                    "v_min_f32_e32 %[t2], 0x3f7d70a4, %[t2]\n\t" /* alpha = min(0.99, opacity * e) */                                     \
                    "v_cmpx_ngt_f32_e32 0x3b808081, %[t2]\n\t"   /* keep !(alpha < 1/255) */                                              \
                    "v_sub_f32_e32 %[t3], 1.0, %[t2]\n\t"                                                                                 \
                    "v_mul_f32_e32 %[t3], %[T], %[t3]\n\t"       /* test_T = T * (1 - alpha) */                                           \
                    "v_cmp_gt_f32_e32 vcc, 0x38d1b717, %[t3]\n\t" /* ends: test_T < 1e-4 */                                               \
                    "v_cndmask_b32_e64 %[done], %[done], 1, vcc\n\t"                                                                      \
                    "v_cmpx_ngt_f32_e32 0x38d1b717, %[t3]\n\t"   /* keep the lanes that blend */                                          \
                    "v_mul_f32_e32 %[t2], %[T], %[t2]\n\t"       /* w = alpha * T */                                                      \
                    "v_fmac_f32_e32 %[C0], %[cy], %[t2]\n\t"                                                                              \
                    "v_fmac_f32_e32 %[C1], %[cz], %[t2]\n\t"                                                                              \
                    "v_fmac_f32_e32 %[C2], %[cw], %[t2]\n\t"                                                                              \
                    "v_cmp_lt_f32_e32 vcc, 0.5, %[T]\n\t"        /* T > 0.5 */                                                            \
                    "v_cmp_gt_f32_e64 %[s2], 0.5, %[t3]\n\t"     /* test_T < 0.5 */                                                       \
                    "s_and_b64 vcc, vcc, %[s2]\n\t"                                                                                       \
                    "v_cndmask_b32_e32 %[D], %[D], %[cx], vcc\n\t" /* median depth: the blend that crosses T = 0.5 */                     \
                    "v_mov_b32_e32 %[T], %[t3]\n\t"                                                                                       \
                    EXTRA                                                                                                                  \
                    "s_mov_b64 exec, %[sv]"
            auto blend_one = [&](const float4 a, const float2 b, const float4 c, const int j) {
                float t0, t1, t2, t3;
                unsigned long long sv, s2;
                if (AUX) {
                    const uint32_t jid = base + (uint32_t)j + 1u; // as forward.cu:335,380
                    asm volatile(R2S_BLEND_ASM("v_mov_b32_e32 %[last], %[jid]\n\t")
                                 : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sv] "=&s"(sv), [s2] "=&s"(s2), [done] "+v"(done_i),
                                   [T] "+v"(T), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [D] "+v"(D), [last] "+v"(last_contributor)
                                 : [ax] "v"(a.x), [ay] "v"(a.y), [az] "v"(a.z), [aw] "v"(a.w), [bx] "v"(b.x), [by] "v"(b.y), [cx] "v"(c.x), [cy] "v"(c.y),
                                   [cz] "v"(c.z), [cw] "v"(c.w), [pfx] "v"(pfx), [pfy] "v"(pfy), [jid] "s"(jid)
                                 : "vcc");
                } else {
                    asm volatile(R2S_BLEND_ASM("")
                                 : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [sv] "=&s"(sv), [s2] "=&s"(s2), [done] "+v"(done_i),
                                   [T] "+v"(T), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [D] "+v"(D)
                                 : [ax] "v"(a.x), [ay] "v"(a.y), [az] "v"(a.z), [aw] "v"(a.w), [bx] "v"(b.x), [by] "v"(b.y), [cx] "v"(c.x), [cy] "v"(c.y),
                                   [cz] "v"(c.z), [cw] "v"(c.w), [pfx] "v"(pfx), [pfy] "v"(pfy)
                                 : "vcc");
                }
            };
#undef R2S_BLEND_ASM
#else
            auto blend_one = [&](const float4 a, const float2 b, const float4 c, const int j) {
                const float dx = a.x - pfx, dy = a.y - pfy;
                const float power = fmaf(dx, fmaf(a.z, dx, a.w * dy), (b.x * dy) * dy); // = log2(e) * forward.cu:342's power
                const float alpha = fminf(0.99f, b.y * __builtin_amdgcn_exp2f(power));
                // forward.cu:344-351 as one predicate: power > 0 and alpha < 1/255 skip the instance for this pixel
                const bool hit = !done && !(power > 0.0f) && !(alpha < 1.0f / 255.0f);
#ifdef R2S_COMP_STATS
                {
                    const unsigned long long hb = __builtin_amdgcn_ballot_w64(hit), ab = __builtin_amdgcn_ballot_w64(!done);
                    if (lane == 0) { atomicAdd(&g_comp_stats[0], 1ull); atomicAdd(&g_comp_stats[1], (unsigned long long)__builtin_popcountll(hb));
                                     atomicAdd(&g_comp_stats[2], hb == 0 ? 1ull : 0ull); atomicAdd(&g_comp_stats[3], (unsigned long long)__builtin_popcountll(ab)); }
                }
#endif
                // No "nobody hit -> next instance" shortcut here: the ballot + scalar branch it needs in every iteration cost more than
                // the blend instructions it saved in the 12 % of iterations without a hit (0.98 -> 0.80 ms without it, images identical).
                const float test_T = T * (1.f - alpha);         // forward.cu:353 as written (T - alpha T is one instruction less, an ulp off, and 0.01 ms)
                const bool ends = test_T < 0.0001f;            // forward.cu:354 ends the pixel; its complement is a mask operation, not a second
                const bool term = hit && ends;                  // compare (">= 0.0001f" next to "!(>= 0.0001f)" compiled to two: NaN semantics)
                done = done || term;
                const bool blend = hit != term;                 // = hit && !ends as a mask XOR (written with !ends the compiler compares again)
                const float w = blend ? alpha * T : 0.0f;
                C0 += c.y * w;
                C1 += c.z * w;
                C2 += c.w * w;
                D = (blend && T > 0.5f && test_T < 0.5f) ? c.x : D;
                T = blend ? test_T : T;
                if (AUX) last_contributor = blend ? base + (uint32_t)j + 1u : last_contributor; // as forward.cu:335,380
            };
#endif
            if (bits) {
                int j0 = sw * 64 + __builtin_ctzll(bits), j1 = 0;
                bits &= bits - 1;
                float4 a0, c0, a1, c1;
                float2 b0, b1;
                fetch(j0, a0, b0, c0);
                for (;;) {
                    // the next record is fetched UNCONDITIONALLY (record 0 of the word again when the walk is over): a fetch inside a branch
                    // leaves the wait in front of the blend with "everything outstanding" on the merged path, i.e. no prefetch at all
                    const bool more = bits != 0;
                    j1 = sw * 64 + (more ? __builtin_ctzll(bits) : 0);
                    bits &= bits - 1;
                    fetch(j1, a1, b1, c1);
                    __builtin_amdgcn_sched_barrier(0); // the reads of the next record stay IN FRONT of this record's blend
                    blend_one(a0, b0, c0, j0);
                    if (!more) break;
                    const bool more2 = bits != 0;
                    j0 = sw * 64 + (more2 ? __builtin_ctzll(bits) : 0);
                    bits &= bits - 1;
                    fetch(j0, a0, b0, c0);
                    __builtin_amdgcn_sched_barrier(0);
                    blend_one(a1, b1, c1, j1);
                    if (!more2) break;
                }
            }
        }
    };
    for (int i = 0; i < rounds; ++i) {
#ifndef R2S_COMP_CXX
        done = done_i != 0;
#endif
        if (__syncthreads_count(done) == TILE_THREADS) break;
        stage(i, 0);
        __syncthreads();
        consume(i, 0);
    }
    if (inside) {
        const size_t pix = (size_t)W * py + px;
        const size_t hw = (size_t)H * W;
        fr.out_color[pix] = C0 + T * fr.bg[0];
        fr.out_color[hw + pix] = C1 + T * fr.bg[1];
        fr.out_color[2 * hw + pix] = C2 + T * fr.bg[2];
        fr.out_depth[pix] = D;
        if (aux_T) aux_T[(size_t)f * hw + pix] = T;
        if (aux_n) aux_n[(size_t)f * hw + pix] = last_contributor;
    }
}

// getHigherMsb, rasterizer_impl.cu:35-50.
uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// Second sort: 17 key bits for 64 frames x 1200 tiles.  rocPRIM's default Onesweep takes 8 bits per pass = 3 passes
// over all instances; 9 bits per pass does it in 2.
#ifndef R2S_TILE_SORT_RADIX
#define R2S_TILE_SORT_RADIX 9
#endif
using TileSortConfig = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 12>, rocprim::kernel_config<512, 12>, R2S_TILE_SORT_RADIX,
                                        rocprim::block_radix_rank_algorithm::match>>;

#ifdef R2S_GAUSS_SORT_RADIX
using GaussSortConfig = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 8>, rocprim::kernel_config<512, 8>, R2S_GAUSS_SORT_RADIX,
                                        rocprim::block_radix_rank_algorithm::match>>;
#else
using GaussSortConfig = rocprim::default_config;
#endif

struct CallbackAlloc {
    r2s_alloc_fn fn[3];
    void* user[3];
};

} // namespace

struct R2SRasterCtx {
    r2s::DevBuf buf[3]; // geometry, binning, image
    CallbackAlloc cb{}; // when set, scratch comes from the caller (single-frame API)
    bool use_cb = false;
    FrameDev* d_frames = nullptr;
    FrameDev* h_frames = nullptr; // pinned
    int frames_cap = 0;
    uint64_t* h_read = nullptr; // pinned: [0] last offset, [1] error flag, [2] overflow flag (sync-free mode), [3] instance count of the binning path
    // sync-free mode (r2s_raster_ctx_set_async): the instance count is NOT read back between scan and emit; the binning
    // scratch is sized from the last known count x 1.25 and the count / error / overflow words of a call are read later
    bool async_mode = false;
    uint32_t L_cap = 0;          // capacity the binning scratch was sized for; 0 = not known yet (next call synchronises once)
    int64_t last_L = 0;          // most recent count the host has seen
    int64_t last_slots = 0;      // ... and the emission slots that batch needed (= the count, except on the binning path with culling: see KEY_NONE)
    bool pending_true = false;   // the pending sync-free batch reports its count in h_read[3] (binning path)
    hipEvent_t cnt_ev = nullptr; // synchronous calls: behind the copy of the binning path's count
    hipEvent_t done_ev = nullptr;
    bool pending = false;
    int overflows = 0;           // sync-free batches whose capacity was too small (their images miss instances)
    int late_error = 0;          // error of a sync-free call, reported by the next poll        // a sync-free call whose words have not been looked at yet
    bool timing = false;
    int cull = 0; // exact-output tile culling of instances (batched API option)
    bool bin_pass = getenv("R2S_RASTER_RADIX_SORT") == nullptr; // one-pass tile binning (k_bin_*) instead of the radix sort of the instances
    bool bin_staged = getenv("R2S_BIN_STAGED") != nullptr; // experiment: k_bin_scatter_staged
    bool tile_order = true; // longest-first workgroup order of the compositor (R2S_NO_TILE_ORDER at context creation: A/B knob)
    hipEvent_t ev[7] = {};
    bool ev_ok = false;
    float stage_ms[6] = {};
    float* aux_T = nullptr;
    uint32_t* aux_n = nullptr;
    R2SRasterDebug dbg{};
    std::vector<int64_t> per_frame;

    char* scratch(int which, size_t bytes)
    {
        if (use_cb) return cb.fn[which](cb.user[which], bytes);
        if (buf[which].reserve(bytes) != hipSuccess) return nullptr;
        return buf[which].p;
    }
};

namespace {

int ensure_frames(R2SRasterCtx* c, int n)
{
    if (n > c->frames_cap) {
        if (c->d_frames) (void)hipFree(c->d_frames);
        if (c->h_frames) (void)hipHostFree(c->h_frames);
        c->d_frames = nullptr;
        c->h_frames = nullptr;
        int cap = n < 64 ? 64 : n;
        R2S_HIP_TRY(r2s::dev_malloc((void**)&c->d_frames, sizeof(FrameDev) * cap));
        R2S_HIP_TRY(hipHostMalloc((void**)&c->h_frames, sizeof(FrameDev) * cap, hipHostMallocDefault));
        c->frames_cap = cap;
    }
    if (!c->h_read) R2S_HIP_TRY(hipHostMalloc((void**)&c->h_read, 64, hipHostMallocDefault));
    return R2S_OK;
}

int64_t forward_impl(R2SRasterCtx* c, const R2SGaussianSet* sets, int n_sets, const R2SRasterFrame* frames, int F, int W, int H,
                     int64_t* per_frame_out, hipStream_t stream)
{
    if (!c || F < 0 || W <= 0 || H <= 0 || (F > 0 && (!sets || !frames))) return R2S_ERR_INVALID;
    if (F == 0) return 0;
    int rc = ensure_frames(c, F);
    if (rc != R2S_OK) return rc;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    const int tiles = gx * gy;

    uint64_t total = 0;
    int maxP = 0;
    for (int f = 0; f < F; ++f) {
        const R2SRasterFrame& fr = frames[f];
        if (fr.set < 0 || fr.set >= n_sets) return R2S_ERR_INVALID;
        const R2SGaussianSet& s = sets[fr.set];
        if (s.P < 0 || !fr.viewmatrix || !fr.projmatrix || !fr.background || !fr.out_color || !fr.out_depth) return R2S_ERR_INVALID;
        if (s.P > 0 && (!s.means3D || !s.opacities)) return R2S_ERR_INVALID;
        if (s.P > 0 && !s.colors_precomp && (!s.shs || !fr.cam_pos)) return R2S_ERR_INVALID;
        if (s.P > 0 && !s.cov3D_precomp && (!s.scales || !s.rotations)) return R2S_ERR_INVALID;
        FrameDev& d = c->h_frames[f];
        d.view = fr.viewmatrix; d.proj = fr.projmatrix; d.campos = fr.cam_pos; d.bg = fr.background;
        d.means3D = s.means3D; d.shs = s.shs; d.colors = s.colors_precomp; d.opac = s.opacities;
        d.scales = s.scales; d.rots = s.rotations; d.cov3D = s.cov3D_precomp;
        d.out_color = fr.out_color; d.out_depth = fr.out_depth; d.radii = fr.radii;
        d.P = s.P; d.D = s.D; d.M = s.M;
        d.base = (uint32_t)total;
        d.scale_mod = s.scale_modifier;
        d.tan_fovx = fr.tan_fovx; d.tan_fovy = fr.tan_fovy;
        d.focal_y = H / (2.0f * fr.tan_fovy); // rasterizer_impl.cu:225-226
        d.focal_x = W / (2.0f * fr.tan_fovx);
        d.z_thr = fr.z_threshold;
        d.prefiltered = fr.prefiltered;
        total += (uint64_t)s.P;
        if (s.P > maxP) maxP = s.P;
    }
    if (total >= 0xFFFFFFFFull) return R2S_ERR_OVERFLOW;
    const size_t G = (size_t)total;

    if (c->timing && !c->ev_ok) {
        for (auto& e : c->ev) R2S_HIP_TRY(hipEventCreate(&e));
        c->ev_ok = true;
    }
    auto mark = [&](int k) { if (c->timing) (void)hipEventRecord(c->ev[k], stream); };

    // ---- geometry scratch (GeometryState, rasterizer_impl.h:30-45) ----
    size_t scan_bytes = 0;
    R2S_HIP_TRY(rocprim::inclusive_scan(nullptr, scan_bytes,
                                        rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0u), TilesInOrder{DepthOrder{nullptr, nullptr, 0}, nullptr}),
                                        (uint32_t*)nullptr, G ? G : 1, rocprim::plus<uint32_t>(), stream));
    // first sort: Gaussians by (frame, depth bits)
    unsigned fbits = 0;
    while ((1u << fbits) < (unsigned)F) ++fbits;
    // the key carries the Gaussian's index inside its frame below the sorted bits (keys only: 8 bytes per Gaussian and pass instead of 12)
    unsigned idx_bits = 1;
    while ((1ull << idx_bits) < (unsigned long long)maxP) ++idx_bits;
    if (fbits + 32u + idx_bits > 64u) return R2S_ERR_OVERFLOW; // frames x Gaussians per frame beyond 2^32: not a batch this library renders
    size_t gsort_bytes = 0;
    {
        rocprim::double_buffer<uint64_t> dk((uint64_t*)nullptr, (uint64_t*)nullptr);
        R2S_HIP_TRY(rocprim::radix_sort_keys<GaussSortConfig>(nullptr, gsort_bytes, dk, G ? G : 1, idx_bits, idx_bits + 32u + fbits, stream));
    }
    float* depths; int* radii_all; GeomRec* geom; uint32_t* tiles_touched; uint32_t* offsets; char* scan_tmp; int* err_flag; uint32_t* big_q;
    uint64_t *gkeys_a, *gkeys_b; char* gsort_tmp; unsigned long long* surv;
    {
        r2s::Carver sz(nullptr);
        sz.take<float>(G); sz.take<int>(G); sz.take<GeomRec>(G); sz.take<uint32_t>(G); sz.take<uint32_t>(G);
        sz.take<char>(scan_bytes); sz.take<int>(4); sz.take<uint32_t>(G);
        sz.take<uint64_t>(G); sz.take<uint64_t>(G); sz.take<char>(gsort_bytes); sz.take<unsigned long long>(SURV_MASKS && c->cull ? G : 0);
        char* p = c->scratch(0, sz.bytes());
        if (!p) return R2S_ERR_ALLOC;
        r2s::Carver cv(p);
        depths = cv.take<float>(G); radii_all = cv.take<int>(G); geom = cv.take<GeomRec>(G);
        tiles_touched = cv.take<uint32_t>(G); offsets = cv.take<uint32_t>(G);
        scan_tmp = cv.take<char>(scan_bytes); err_flag = cv.take<int>(4); big_q = cv.take<uint32_t>(G);
        gkeys_a = cv.take<uint64_t>(G); gkeys_b = cv.take<uint64_t>(G);
        gsort_tmp = cv.take<char>(gsort_bytes);
        surv = cv.take<unsigned long long>(SURV_MASKS && c->cull ? G : 0);
    }
    // ---- image scratch (ImageState: ranges; accum_alpha / n_contrib are backward-only) ----
    uint2* ranges;
    uint32_t *tl_order, *tl_cls;
    const size_t FT = (size_t)F * tiles;
    {
        r2s::Carver sz(nullptr);
        sz.take<uint2>(FT); sz.take<uint32_t>(FT); sz.take<uint32_t>(2u << TILE_ORDER_BITS);
        char* p = c->scratch(2, sz.bytes());
        if (!p) return R2S_ERR_ALLOC;
        r2s::Carver cv(p);
        ranges = cv.take<uint2>(FT);
        tl_order = cv.take<uint32_t>(FT);
        tl_cls = cv.take<uint32_t>(2u << TILE_ORDER_BITS); // class counts | class cursors
    }

    R2S_HIP_TRY(hipMemcpyAsync(c->d_frames, c->h_frames, sizeof(FrameDev) * F, hipMemcpyHostToDevice, stream));
    R2S_HIP_TRY(hipMemsetAsync(err_flag, 0, sizeof(int) * 4, stream));

    // Sync-free pass: possible once a capacity is known (the first call of the mode reads the count back like the reference).
    // A previous sync-free call is looked at first: its count updates the capacity, an overflow makes THIS call synchronise
    // and re-size (the overflowing batch itself lost its deepest instances: reported through r2s_raster_ctx_poll).
    bool sync_free = c->async_mode && c->L_cap > 0 && G > 0 && !c->timing && !per_frame_out;
    if (c->pending) {
        R2S_HIP_TRY(hipEventSynchronize(c->done_ev)); // long finished: it was recorded a whole env step ago
        c->pending = false;
        c->last_slots = (int64_t)(c->h_read[0] & 0xFFFFFFFFull);
        c->last_L = c->pending_true ? (int64_t)(c->h_read[3] & 0xFFFFFFFFull) : c->last_slots;
        if ((int)(c->h_read[1] & 0xFFFFFFFFull) != 0) c->late_error = R2S_ERR_PREFILTERED;
        if ((int)(c->h_read[2] & 0xFFFFFFFFull) != 0) { c->overflows++; c->L_cap = 0; sync_free = false; c->last_L = c->last_slots; } // (its histograms cover the slots that fitted: report what it needed)
        else if (c->L_cap > 0) {
            // follow the scene BOTH ways, one batch late: everything behind the emission (sentinel fill, tile sort, ranges) runs over the
            // capacity, not over the count, so slack is paid for on every step — 12.5 % + 4096 (a rollout's count moves by a percent or two
            // per step; a bigger jump is an overflow: reported, and the closed loop renders that step again).  Growing-only with 25 % of
            // slack had the tile sort work on 1.25 x the PEAK count of the episode.
            const uint64_t want = (uint64_t)c->last_slots + c->last_slots / 8 + 4096;
            c->L_cap = (uint32_t)std::min<uint64_t>(want, 0xFFFFFFF0ull);
        }
    }
    mark(0);
    uint32_t L = 0;
    // one-pass binning (k_bin_*) wherever a frame's tiles fit its LDS counters; otherwise (and with R2S_RASTER_RADIX_SORT set: A/B knob) the radix sort
    const bool bin_pass = c->bin_pass && !SURV_MASKS && tiles <= BIN_MAX_TILES && F <= BIN_MAX_FRAMES; // (the survivor-mask experiment needs the exact counts: radix path)
    // ... on which the emission slots follow from the rectangles (KEY_NONE): k_preprocess counts the survivors of the culling test only for the radix sort
    const int count_exact = (c->cull && !bin_pass) ? 1 : 0;
    DepthOrder order{nullptr, c->d_frames, (int)idx_bits};
    if (G > 0) {
        dim3 grid((maxP + 255) / 256, F);
        hipLaunchKernelGGL(k_preprocess, grid, dim3(256), 0, stream, c->d_frames, gx, gy, W, H, depths, radii_all, geom,
                           tiles_touched, err_flag, c->cull, count_exact, gkeys_a, (int)idx_bits, surv);
        mark(1);
        rocprim::double_buffer<uint64_t> dgk(gkeys_a, gkeys_b);
        R2S_HIP_TRY(rocprim::radix_sort_keys<GaussSortConfig>(gsort_tmp, gsort_bytes, dgk, G, idx_bits, idx_bits + 32u + fbits, stream));
        order.keys = dgk.current();
        // offsets[i] = instances of the first i+1 Gaussians in (frame, depth) order; frames stay contiguous
        R2S_HIP_TRY(rocprim::inclusive_scan(scan_tmp, scan_bytes,
                                            rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0u), TilesInOrder{order, tiles_touched}),
                                            offsets, G, rocprim::plus<uint32_t>(), stream));
        mark(2);
        R2S_HIP_TRY(hipMemcpyAsync(&c->h_read[0], offsets + (G - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        R2S_HIP_TRY(hipMemcpyAsync(&c->h_read[1], err_flag, sizeof(int), hipMemcpyDeviceToHost, stream));
        if (!sync_free) {
            // The reference's blocking read of the instance count (rasterizer_impl.cu:284), once per batch.
            R2S_HIP_TRY(hipStreamSynchronize(stream));
            L = (uint32_t)(c->h_read[0] & 0xFFFFFFFFull);
            if ((int)(c->h_read[1] & 0xFFFFFFFFull) != 0) return R2S_ERR_PREFILTERED;
            c->last_L = L; c->last_slots = L;
            if (c->async_mode) c->L_cap = (uint32_t)std::min<uint64_t>((uint64_t)L + L / 8 + 4096, 0xFFFFFFF0ull); // first call of the mode
        }
    } else {
        mark(1); mark(2);
    }

    // ---- binning scratch (BinningState, rasterizer_impl.h:56-67) ----
    // sized for the instance count, or — sync-free — for the capacity derived from an earlier batch (scenes move slowly:
    // the count changes by a fraction of a per cent per env step; 25 % headroom, and an overflow flag if it ever is not enough)
    const uint32_t cap = sync_free ? c->L_cap : L;
    const uint32_t bits = higher_msb((uint32_t)F * (uint32_t)tiles + (sync_free ? 1u : 0u)); // + 1: the sentinel key F * tiles
    uint32_t *keys_a = nullptr, *keys_b = nullptr;
    uint32_t *vals_a = nullptr, *vals_b = nullptr;
    const uint32_t* keys_sorted = nullptr;
    const uint32_t* vals_sorted = nullptr;
    bool ranges_done = false, count_pending = false;
    const uint32_t* frame_start_dev = nullptr;
    const bool want_order = cap > 0 && G > 0 && FT >= 4096 && c->tile_order; // compositor workgroups longest list first (big batches)
    if (want_order) R2S_HIP_TRY(hipMemsetAsync(tl_cls, 0, sizeof(uint32_t) * (2u << TILE_ORDER_BITS), stream));
    if (cap > 0 && G > 0) {
        rocprim::double_buffer<uint32_t> dk((uint32_t*)nullptr, (uint32_t*)nullptr);
        rocprim::double_buffer<uint32_t> dv((uint32_t*)nullptr, (uint32_t*)nullptr);
        size_t sort_bytes = 0;
        if (!bin_pass) R2S_HIP_TRY(rocprim::radix_sort_pairs<TileSortConfig>(nullptr, sort_bytes, dk, dv, (size_t)cap, 0u, bits, stream));
        const uint32_t nb_max = cap / BIN_CHUNK + (uint32_t)F; // every frame ends in at most one partial chunk
        uint32_t *bin_hist = nullptr, *bin_first = nullptr, *bin_totals = nullptr, *bin_local = nullptr, *bin_gsum = nullptr;
        uint32_t* bin_fstart = nullptr; // instances before each frame; [F]: of the batch
        uint4* bin_desc = nullptr;
        char* sort_tmp = nullptr;
        {
            r2s::Carver sz(nullptr);
            sz.take<uint32_t>(bin_pass ? (cap + 1) / 2 : cap); sz.take<uint32_t>(cap); sz.take<uint32_t>(cap);
            if (bin_pass) { sz.take<uint32_t>((size_t)nb_max * tiles); sz.take<uint4>(nb_max); sz.take<uint32_t>((size_t)F + 1); sz.take<uint32_t>(FT); sz.take<uint32_t>(FT); sz.take<uint32_t>((FT + 255) / 256); sz.take<uint32_t>((size_t)F + 1); }
            else { sz.take<uint32_t>(cap); sz.take<char>(sort_bytes); }
            char* p = c->scratch(1, sz.bytes());
            if (!p) return R2S_ERR_ALLOC;
            r2s::Carver cv(p);
            keys_a = cv.take<uint32_t>(bin_pass ? (cap + 1) / 2 : cap); vals_a = cv.take<uint32_t>(cap); vals_b = cv.take<uint32_t>(cap);
            if (bin_pass) {
                bin_hist = cv.take<uint32_t>((size_t)nb_max * tiles); bin_desc = cv.take<uint4>(nb_max); bin_first = cv.take<uint32_t>((size_t)F + 1);
                bin_totals = cv.take<uint32_t>(FT); bin_local = cv.take<uint32_t>(FT); bin_gsum = cv.take<uint32_t>((FT + 255) / 256); bin_fstart = cv.take<uint32_t>((size_t)F + 1);
            } else { keys_b = cv.take<uint32_t>(cap); sort_tmp = cv.take<char>(sort_bytes); }
        }

        if (sync_free && !bin_pass) // instances this batch does not produce: sentinel keys that sort behind every (frame, tile)
            hipLaunchKernelGGL(k_fill_sentinel, dim3((cap + 255) / 256), dim3(256), 0, stream, offsets + (G - 1), cap, (uint32_t)F * (uint32_t)tiles, keys_a);
        auto emit = [&](auto* keys) { // 16-bit keys (tile inside the frame) for the one-pass binning, frame-extended 32-bit keys for the radix sort
            using KeyT = std::remove_pointer_t<decltype(keys)>;
            if (G <= (size_t)1 << 18) { // small batch: large rectangles through the queue (see k_emit_keys)
                hipLaunchKernelGGL((k_emit_keys<true, KeyT>), dim3((unsigned)((G + 255) / 256)), dim3(256), 0, stream, c->d_frames, (uint32_t)G, gx, gy, W, H,
                                   order, radii_all, geom, offsets, keys, vals_a, c->cull, cap, err_flag + 1, big_q, err_flag + 2, surv);
                hipLaunchKernelGGL(k_emit_big<KeyT>, dim3(1024), dim3(256), 0, stream, (uint32_t)G, gx, gy, W, H,
                                   order, radii_all, geom, offsets, keys, vals_a, c->cull, cap, big_q, err_flag + 2);
            } else
                hipLaunchKernelGGL((k_emit_keys<false, KeyT>), dim3((unsigned)((G + 255) / 256)), dim3(256), 0, stream, c->d_frames, (uint32_t)G, gx, gy, W, H,
                                   order, radii_all, geom, offsets, keys, vals_a, c->cull, cap, err_flag + 1, big_q, err_flag + 2, surv);
        };
        const uint16_t* keys16 = reinterpret_cast<const uint16_t*>(keys_a);
        if (bin_pass) emit(reinterpret_cast<uint16_t*>(keys_a));
        else emit(keys_a);
        mark(3);
        if (bin_pass) {
            unsigned key_bits = 0;
            while ((1u << key_bits) < (unsigned)tiles) ++key_bits;
            hipLaunchKernelGGL(k_bin_plan, dim3(1), dim3(1024), 0, stream, c->d_frames, F, offsets, cap, nb_max, bin_desc, bin_first);
            hipLaunchKernelGGL(k_bin_hist, dim3(nb_max), dim3(BIN_THREADS), sizeof(uint32_t) * tiles, stream, bin_desc, tiles, keys16, bin_hist);
            hipLaunchKernelGGL(k_bin_colscan, dim3((unsigned)((FT + 255) / 256)), dim3(256), 0, stream, F, tiles, bin_first, bin_hist, bin_totals, bin_local, bin_gsum);
            hipLaunchKernelGGL(k_tile_starts, dim3((unsigned)((FT + 255) / 256)), dim3(256), 0, stream, (uint32_t)FT, bin_totals, bin_local, bin_gsum, ranges, want_order ? tl_cls : nullptr, (uint32_t)tiles, bin_fstart);
            R2S_HIP_TRY(hipMemcpyAsync(&c->h_read[3], bin_fstart + F, sizeof(uint32_t), hipMemcpyDeviceToHost, stream)); // the instance count
            if (!sync_free && c->cull) { // a synchronous call returns the count: wait for it (not for the kernels behind it) at the end
                if (!c->cnt_ev) R2S_HIP_TRY(hipEventCreateWithFlags(&c->cnt_ev, hipEventDisableTiming));
                R2S_HIP_TRY(hipEventRecord(c->cnt_ev, stream));
                count_pending = true;
            }
            frame_start_dev = bin_fstart;
            if (c->bin_staged && tiles <= BIN_STAGED_MAX_TILES) {
                const size_t st_lds = sizeof(uint32_t) * (BIN_CHUNK + tiles) + sizeof(unsigned short) * (BIN_CHUNK + ((tiles + 2) & ~1) + (size_t)(BIN_THREADS / 64) * tiles);
                static bool lds_ok = false; // more than the default 64 KB of dynamic LDS per workgroup: asked for once per process
                if (!lds_ok) {
                    R2S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_scatter_staged<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                    R2S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_scatter_staged<10>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                    R2S_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bin_scatter_staged<11>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                    lds_ok = true;
                }
                if (key_bits <= 9)
                    hipLaunchKernelGGL(k_bin_scatter_staged<9>, dim3(nb_max), dim3(BIN_THREADS), st_lds, stream, bin_desc, tiles, keys16, vals_a, bin_hist, ranges, vals_b);
                else if (key_bits == 10)
                    hipLaunchKernelGGL(k_bin_scatter_staged<10>, dim3(nb_max), dim3(BIN_THREADS), st_lds, stream, bin_desc, tiles, keys16, vals_a, bin_hist, ranges, vals_b);
                else
                    hipLaunchKernelGGL(k_bin_scatter_staged<11>, dim3(nb_max), dim3(BIN_THREADS), st_lds, stream, bin_desc, tiles, keys16, vals_a, bin_hist, ranges, vals_b);
            } else {
            const size_t sc_lds = sizeof(uint32_t) * tiles * (BIN_THREADS / 64);
            if (key_bits <= 9)
                hipLaunchKernelGGL(k_bin_scatter<9>, dim3(nb_max), dim3(BIN_THREADS), sc_lds, stream, bin_desc, tiles, keys16, vals_a, bin_hist, ranges, vals_b);
            else if (key_bits == 10)
                hipLaunchKernelGGL(k_bin_scatter<10>, dim3(nb_max), dim3(BIN_THREADS), sc_lds, stream, bin_desc, tiles, keys16, vals_a, bin_hist, ranges, vals_b);
            else
                hipLaunchKernelGGL(k_bin_scatter<11>, dim3(nb_max), dim3(BIN_THREADS), sc_lds, stream, bin_desc, tiles, keys16, vals_a, bin_hist, ranges, vals_b);
            }
            vals_sorted = vals_b;
            ranges_done = true;
        } else {
            rocprim::double_buffer<uint32_t> dkey(keys_a, keys_b);
            rocprim::double_buffer<uint32_t> dval(vals_a, vals_b);
            R2S_HIP_TRY(rocprim::radix_sort_pairs<TileSortConfig>(sort_tmp, sort_bytes, dkey, dval, (size_t)cap, 0u, bits, stream));
            keys_sorted = dkey.current();
            vals_sorted = dval.current();
        }
        mark(4);
    } else {
        mark(3); mark(4);
    }
    if (!ranges_done) {
        R2S_HIP_TRY(hipMemsetAsync(ranges, 0, sizeof(uint2) * (size_t)F * tiles, stream));
        if (cap > 0 && G > 0)
            hipLaunchKernelGGL(k_tile_ranges, dim3((cap + 255) / 256), dim3(256), 0, stream, offsets + (G - 1), cap, keys_sorted, ranges);
    }
    if (sync_free) { // count, culling error and overflow words of THIS call land in pinned memory behind the pipeline
        R2S_HIP_TRY(hipMemcpyAsync(&c->h_read[2], err_flag + 1, sizeof(int), hipMemcpyDeviceToHost, stream));
        L = cap;
    }
    const uint32_t* tile_order = nullptr;
    if (want_order) { // big batches: start the deepest tiles first
        if (!ranges_done) hipLaunchKernelGGL(k_tile_classes, dim3((unsigned)((FT + 255) / 256)), dim3(256), 0, stream, (uint32_t)FT, ranges, tl_cls);
        hipLaunchKernelGGL(k_tile_order, dim3((unsigned)((FT + 1023) / 1024)), dim3(1024), 0, stream, (uint32_t)FT, ranges, tl_cls, tl_cls + (1u << TILE_ORDER_BITS), tl_order);
        tile_order = tl_order;
    }
    mark(5);
    if (c->aux_T || c->aux_n) // the backward-only auxiliaries (final_T, n_contrib) cost two VALU instructions per blend: only on request
        hipLaunchKernelGGL(k_composite<true>, dim3((uint32_t)F * tiles), dim3(TILE_THREADS), 0, stream, c->d_frames, gx, gy, W, H, ranges,
                           vals_sorted, geom, c->aux_T, c->aux_n, tile_order);
    else
        hipLaunchKernelGGL(k_composite<false>, dim3((uint32_t)F * tiles), dim3(TILE_THREADS), 0, stream, c->d_frames, gx, gy, W, H, ranges,
                           vals_sorted, geom, c->aux_T, c->aux_n, tile_order);
    mark(6);
    R2S_HIP_TRY(hipGetLastError());
    if (sync_free) {
        if (!c->done_ev) R2S_HIP_TRY(hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming));
        R2S_HIP_TRY(hipEventRecord(c->done_ev, stream));
        c->pending = true;
        c->pending_true = ranges_done;
    }

    // per-frame instance counts: offsets at frame boundaries are not read back (that would add syncs);
    // the caller gets the total, and per-frame counts only in timing/debug mode.
    c->per_frame.assign(F, -1);
    if (c->timing) {
        R2S_HIP_TRY(hipStreamSynchronize(stream));
        for (int k = 0; k < 6; ++k) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, c->ev[k], c->ev[k + 1]);
            c->stage_ms[k] = ms;
        }
    }
    int64_t L_report = sync_free ? c->last_L : (int64_t)L;
    if (count_pending) { // binning path with culling: the count is the histograms' total, not the slots' scan
        R2S_HIP_TRY(hipEventSynchronize(c->cnt_ev));
        L_report = (int64_t)(c->h_read[3] & 0xFFFFFFFFull);
        c->last_L = L_report;
    }
    if (per_frame_out) {
        std::vector<uint32_t> ends(F, 0);
        if (frame_start_dev) { // binning path: the scan of the tile totals at the frame boundaries
            std::vector<uint32_t> fs((size_t)F + 1, 0);
            R2S_HIP_TRY(hipMemcpyAsync(fs.data(), frame_start_dev, sizeof(uint32_t) * ((size_t)F + 1), hipMemcpyDeviceToHost, stream));
            R2S_HIP_TRY(hipStreamSynchronize(stream));
            for (int f = 0; f < F; ++f) ends[f] = fs[(size_t)f + 1];
        } else {
            // one small D2H per frame boundary, only when asked for
            for (int f = 0; f < F; ++f) {
                const uint64_t endg = (uint64_t)c->h_frames[f].base + (uint64_t)c->h_frames[f].P;
                if (endg == 0) { ends[f] = 0; continue; }
                R2S_HIP_TRY(hipMemcpyAsync(&ends[f], offsets + (endg - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            }
            R2S_HIP_TRY(hipStreamSynchronize(stream));
        }
        uint32_t prev = 0;
        for (int f = 0; f < F; ++f) {
            per_frame_out[f] = (int64_t)(ends[f] - prev);
            prev = ends[f];
        }
    }
    c->dbg.total_gaussians = (int64_t)G;
    c->dbg.num_rendered = L_report;
    c->dbg.depths = depths;
    c->dbg.radii = radii_all;
    c->dbg.geom = reinterpret_cast<const float*>(geom);
    c->dbg.tiles_touched = tiles_touched;
    c->dbg.point_offsets = offsets;
    c->dbg.keys_sorted = keys_sorted;
    c->dbg.point_list = vals_sorted;
    c->dbg.ranges = reinterpret_cast<const uint32_t*>(ranges);
    return L_report; // sync-free: the most recent count the host has seen (an earlier batch's)
}

} // namespace

// ------------------------------------------------------------------------------------------------
extern "C" {

int r2s_raster_ctx_create(R2SRasterCtx** out)
{
    if (!out) return R2S_ERR_INVALID;
    *out = new (std::nothrow) R2SRasterCtx();
    if (*out && getenv("R2S_NO_TILE_ORDER")) (*out)->tile_order = false; // A/B knob, read once per context
    return *out ? R2S_OK : R2S_ERR_ALLOC;
}

void r2s_raster_ctx_destroy(R2SRasterCtx* c)
{
    if (!c) return;
    for (auto& b : c->buf) b.release();
    if (c->d_frames) (void)hipFree(c->d_frames);
    if (c->h_frames) (void)hipHostFree(c->h_frames);
    if (c->h_read) (void)hipHostFree(c->h_read);
    if (c->done_ev) (void)hipEventDestroy(c->done_ev);
    if (c->cnt_ev) (void)hipEventDestroy(c->cnt_ev);
    if (c->ev_ok) for (auto& e : c->ev) (void)hipEventDestroy(e);
    delete c;
}

size_t r2s_raster_ctx_scratch_bytes(const R2SRasterCtx* c)
{
    return c ? c->buf[0].cap + c->buf[1].cap + c->buf[2].cap : 0;
}

void r2s_raster_ctx_set_timing(R2SRasterCtx* c, int enable) { if (c) c->timing = enable != 0; }
void r2s_raster_ctx_set_async(R2SRasterCtx* c, int enable) { if (c) { c->async_mode = enable != 0; if (!enable) c->L_cap = 0; } }

int r2s_raster_ctx_poll(R2SRasterCtx* c, int wait, int64_t* num_rendered, int32_t* overflows)
{
    if (!c) return R2S_ERR_INVALID;
    if (c->pending && (wait || hipEventQuery(c->done_ev) == hipSuccess)) {
        R2S_HIP_TRY(hipEventSynchronize(c->done_ev));
        c->pending = false;
        c->last_slots = (int64_t)(c->h_read[0] & 0xFFFFFFFFull);
        c->last_L = c->pending_true ? (int64_t)(c->h_read[3] & 0xFFFFFFFFull) : c->last_slots;
        if ((int)(c->h_read[1] & 0xFFFFFFFFull) != 0) c->late_error = R2S_ERR_PREFILTERED;
        if ((int)(c->h_read[2] & 0xFFFFFFFFull) != 0) { c->overflows++; c->L_cap = 0; c->last_L = c->last_slots; }
    }
    if (num_rendered) *num_rendered = c->last_L;
    if (overflows) *overflows = c->overflows;
    const int rc = c->late_error;
    c->late_error = 0;
    return rc ? rc : (c->pending ? 1 : R2S_OK);
}
void r2s_raster_ctx_set_tile_culling(R2SRasterCtx* c, int enable) { if (c) c->cull = enable != 0; }
float r2s_raster_ctx_stage_ms(const R2SRasterCtx* c, int stage) { return (c && stage >= 0 && stage < 6) ? c->stage_ms[stage] : -1.f; }
void r2s_raster_ctx_set_aux(R2SRasterCtx* c, float* final_T, uint32_t* n_contrib) { if (c) { c->aux_T = final_T; c->aux_n = n_contrib; } }

int r2s_raster_ctx_debug(const R2SRasterCtx* c, R2SRasterDebug* out)
{
    if (!c || !out) return R2S_ERR_INVALID;
    *out = c->dbg;
    return R2S_OK;
}

int64_t r2s_raster_forward_batch(R2SRasterCtx* ctx, const R2SGaussianSet* sets, int n_sets, const R2SRasterFrame* frames, int n_frames,
                                 int width, int height, int64_t* num_rendered_per_frame, r2s_stream_t stream)
{
    if (!ctx) return R2S_ERR_INVALID;
    ctx->use_cb = false;
    return forward_impl(ctx, sets, n_sets, frames, n_frames, width, height, num_rendered_per_frame, (hipStream_t)stream);
}

int64_t r2s_raster_forward(r2s_alloc_fn geometry_buffer, void* geometry_user, r2s_alloc_fn binning_buffer, void* binning_user,
                           r2s_alloc_fn image_buffer, void* image_user, int P, int D, int M, const float* background, int width,
                           int height, const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                           const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                           const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                           int prefiltered, float z_threshold, float* out_color, float* out_depth, int* radii, r2s_stream_t stream)
{
    if (!geometry_buffer || !binning_buffer || !image_buffer) return R2S_ERR_INVALID;
    if (P == 0) return 0; // rasterize_points.cu:82
    // A per-thread context carries the pinned staging words; scratch itself comes from the callbacks.
    static thread_local R2SRasterCtx* tl = nullptr;
    if (!tl) {
        tl = new (std::nothrow) R2SRasterCtx();
        if (!tl) return R2S_ERR_ALLOC;
    }
    tl->use_cb = true;
    tl->cull = 0; // the drop-in entry point reproduces the reference's instance list exactly
    tl->cb.fn[0] = geometry_buffer; tl->cb.user[0] = geometry_user;
    tl->cb.fn[1] = binning_buffer; tl->cb.user[1] = binning_user;
    tl->cb.fn[2] = image_buffer; tl->cb.user[2] = image_user;
    R2SGaussianSet set{};
    set.P = P; set.D = D; set.M = M; set.scale_modifier = scale_modifier;
    set.means3D = means3D; set.shs = shs; set.colors_precomp = colors_precomp; set.opacities = opacities;
    set.scales = scales; set.rotations = rotations; set.cov3D_precomp = cov3D_precomp;
    R2SRasterFrame fr{};
    fr.set = 0; fr.prefiltered = prefiltered; fr.tan_fovx = tan_fovx; fr.tan_fovy = tan_fovy; fr.z_threshold = z_threshold;
    fr.viewmatrix = viewmatrix; fr.projmatrix = projmatrix; fr.cam_pos = cam_pos; fr.background = background;
    fr.out_color = out_color; fr.out_depth = out_depth; fr.radii = radii;
    return forward_impl(tl, &set, 1, &fr, 1, width, height, nullptr, (hipStream_t)stream);
}

} // extern "C"
