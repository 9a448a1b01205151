/*
 * r2s_raster.h — C ABI of the MI355X (gfx950) Gaussian-splatting rasteriser forward pass.
 *
 * Drop-in boundary for the reference's rasteriser core
 *   third-party/diff-gaussian-rasterization-w-depth/cuda_rasterizer/rasterizer.h:31-54
 *   (CudaRasterizer::Rasterizer::forward) as reached from
 *   rasterize_points.cu:36-117 (RasterizeGaussiansCUDA) / ext.cpp:16 (_C.rasterize_gaussians).
 *
 * Conventions shared by every entry point
 *   - all `const float*` / `float*` / `int*` data arguments are DEVICE pointers (HIP);
 *     scalars are passed by value; nothing here is a torch type;
 *   - the caller owns every in/out buffer; scratch memory is obtained through caller
 *     callbacks (single-frame API, like the reference's std::function<char*(size_t)>)
 *     or lives in an opaque grow-only context (batched API);
 *   - work is enqueued on `stream`; the call performs ONE blocking 8-byte device->host
 *     read (the instance count, the reference's rasterizer_impl.cu:284) and returns
 *     after the remaining kernels are enqueued (not completed);
 *   - return value: number of rendered Gaussian/tile instances (>= 0) or a negative
 *     R2S_ERR_* code.  No exceptions cross the ABI.  Not re-entrant per context.
 */
#ifndef R2S_RASTER_H
#define R2S_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R2S_OK 0
#define R2S_ERR_INVALID (-1)     /* bad argument (null pointer, negative size, ...)            */
#define R2S_ERR_HIP (-2)         /* a HIP runtime call failed; see r2s_last_error()           */
#define R2S_ERR_ALLOC (-3)       /* a scratch callback returned NULL / hipMalloc failed        */
#define R2S_ERR_PREFILTERED (-4) /* `prefiltered` set but a point was culled                  */
                                 /* (device __trap() in the reference, auxiliary.h:157-161)   */
#define R2S_ERR_OVERFLOW (-5)    /* more than 2^32-1 instances (offsets are uint32 like the ref) */

typedef void* r2s_stream_t; /* hipStream_t */

/* Scratch callback: return a device pointer to at least `bytes` bytes (grow-only buffer
 * owned by the caller).  Replaces std::function<char*(size_t)> of rasterizer.h:32-34 and
 * resizeFunctional of rasterize_points.cu:27-33. */
typedef char* (*r2s_alloc_fn)(void* user, size_t bytes);

/* Single frame — replaces CudaRasterizer::Rasterizer::forward (rasterizer.h:31-54),
 * argument for argument, plus the explicit stream.  `radii` may be NULL.
 * Outputs: out_color[3,H,W] planar, out_depth[1,H,W] (median depth, 15.0 where the ray never
 * crosses T=0.5), radii[P].  The caller zero-fills nothing: every pixel and radius is written.
 * P == 0 is handled one level up in the reference (rasterize_points.cu:82); here it returns 0
 * and writes nothing. */
int64_t r2s_raster_forward(
    r2s_alloc_fn geometry_buffer, void* geometry_user,
    r2s_alloc_fn binning_buffer, void* binning_user,
    r2s_alloc_fn image_buffer, void* image_user,
    int P, int D, int M,
    const float* background,
    int width, int height,
    const float* means3D,
    const float* shs,
    const float* colors_precomp,
    const float* opacities,
    const float* scales,
    float scale_modifier,
    const float* rotations,
    const float* cov3D_precomp,
    const float* viewmatrix,
    const float* projmatrix,
    const float* cam_pos,
    float tan_fovx, float tan_fovy,
    int prefiltered,
    float z_threshold,
    float* out_color,
    float* out_depth,
    int* radii,
    r2s_stream_t stream);

/* ---- batched frames (many environments x camera views in one pass) -------------------- */

/* One Gaussian cloud (one environment's scene).  Same meaning as the per-call arguments above. */
typedef struct R2SGaussianSet {
    int32_t P, D, M;
    float scale_modifier;
    const float* means3D;        /* [P,3]            */
    const float* shs;            /* [P,M,3] or NULL  */
    const float* colors_precomp; /* [P,3]   or NULL  */
    const float* opacities;      /* [P]              */
    const float* scales;         /* [P,3]   or NULL  */
    const float* rotations;      /* [P,4]   or NULL  */
    const float* cov3D_precomp;  /* [P,6]   or NULL  */
} R2SGaussianSet;

/* One camera view of one Gaussian set. */
typedef struct R2SRasterFrame {
    int32_t set; /* index into the `sets` array */
    int32_t prefiltered;
    float tan_fovx, tan_fovy, z_threshold;
    const float* viewmatrix; /* [16] device */
    const float* projmatrix; /* [16] device */
    const float* cam_pos;    /* [3]  device */
    const float* background; /* [3]  device */
    float* out_color;        /* [3,H,W] */
    float* out_depth;        /* [1,H,W] */
    int32_t* radii;          /* [P] or NULL */
} R2SRasterFrame;

typedef struct R2SRasterCtx R2SRasterCtx; /* opaque; owns grow-only device scratch */

int r2s_raster_ctx_create(R2SRasterCtx** out);
void r2s_raster_ctx_destroy(R2SRasterCtx* ctx);
/* Bytes of device scratch currently held by the context. */
size_t r2s_raster_ctx_scratch_bytes(const R2SRasterCtx* ctx);

/* All frames share width x height.  One preprocess / depth sort / scan / key-emit / tile binning /
 * composite pass covers every frame (instances are binned by (frame, tile); frames of more than 2 048
 * tiles take a radix sort on frame-extended tile keys instead, with identical lists).
 * `num_rendered_per_frame` (host, may be NULL) receives each frame's instance count.  Returns the total
 * instance count or a negative error (R2S_ERR_OVERFLOW also when log2(frames) + log2(largest set) > 32:
 * the first sort's key holds frame, depth bits and the Gaussian's index inside its frame in 64 bits). */
int64_t r2s_raster_forward_batch(
    R2SRasterCtx* ctx,
    const R2SGaussianSet* sets, int n_sets,
    const R2SRasterFrame* frames, int n_frames,
    int width, int height,
    int64_t* num_rendered_per_frame,
    r2s_stream_t stream);

/* Sync-free mode (off by default).  The reference reads the instance count back between the scan and the key emission
 * (a blocking cudaMemcpy, rasterizer_impl.cu:284) to size its binning buffer; r2s_raster_forward_batch normally does the
 * same once per batch.  With this mode on, only the FIRST batch does: later batches size the binning scratch from the last
 * known count + 12.5 % (+ 4096), bin over that capacity (the count is taken on the device where a kernel needs it), and return
 * without touching the host — the return value is then the most recent count the host has seen (an earlier batch's).
 * r2s_raster_ctx_poll(ctx, wait, &num_rendered, &overflows) looks at the last batch without blocking (wait = 0: returns 1
 * while it is still running) or blocking (wait = 1): its count, the number of batches so far whose capacity was too small
 * (such a batch lost its deepest instances; the next call re-sizes by synchronising once), and a deferred
 * R2S_ERR_PREFILTERED.  Not used by r2s_raster_forward (the drop-in returns the exact count like the reference). */
void r2s_raster_ctx_set_async(R2SRasterCtx* ctx, int enable);
int r2s_raster_ctx_poll(R2SRasterCtx* ctx, int wait, int64_t* num_rendered, int32_t* overflows);

/* Timing hooks for bench.py: HIP-event time (ms) of each stage of the LAST
 * r2s_raster_forward_batch call on this context, measured on the launch stream.
 * stage: 0 preprocess, 1 scan, 2 emit, 3 sort, 4 ranges, 5 composite.  Enabled by
 * r2s_raster_ctx_set_timing(ctx, 1) (adds event records + one sync at the end of the call). */
void r2s_raster_ctx_set_timing(R2SRasterCtx* ctx, int enable);
float r2s_raster_ctx_stage_ms(const R2SRasterCtx* ctx, int stage);

/* Exact-output tile culling (off by default).  The reference emits one instance for every tile of a Gaussian's
 * 3-sigma bounding square (duplicateWithKeys, rasterizer_impl.cu:70-111); with culling on, instances that provably
 * cannot reach alpha >= 1/255 at any pixel of the tile are not emitted.  out_color / out_depth are unchanged (such
 * instances are skipped per pixel at forward.cu:351 anyway); the returned instance count and the backward-only
 * n_contrib become smaller.  With culling on, the debug taps `tiles_touched` / `point_offsets` hold a Gaussian's emission
 * SLOTS (its candidate tiles: the reference's rectangle cut to the bounding box of its alpha >= 1/255 ellipse), not its
 * surviving instances; `num_rendered`, `point_list` and `ranges` are those of the survivors.  A sync-free batch that
 * overflowed reports the slots it needed.  Never applied by r2s_raster_forward. */
void r2s_raster_ctx_set_tile_culling(R2SRasterCtx* ctx, int enable);

/* Debug taps for parity tests: copies of the last batch call's intermediates
 * (device pointers valid until the next call on `ctx`). */
typedef struct R2SRasterDebug {
    int64_t total_gaussians;      /* sum over frames of P(frame) */
    int64_t num_rendered;
    const float* depths;          /* [total_gaussians] */
    const int32_t* radii;         /* [total_gaussians] */
    const float* geom;            /* [total_gaussians, 12]: x,y,conic_a,conic_b | conic_c,opacity,depth,r | g,b,0,0 */
    const uint32_t* tiles_touched;/* [total_gaussians] */
    const uint32_t* point_offsets;/* [total_gaussians] inclusive scan, Gaussians in (frame, depth) order */
    const uint32_t* keys_sorted;  /* [num_rendered] frame-extended tile id of every sorted instance */
    const uint32_t* point_list;   /* [num_rendered] (global Gaussian index = frame base + idx) */
    const uint32_t* ranges;       /* [n_frames*tiles, 2] */
} R2SRasterDebug;
int r2s_raster_ctx_debug(const R2SRasterCtx* ctx, R2SRasterDebug* out);

/* Optional per-pixel auxiliaries of the reference's ImageState (accum_alpha, n_contrib;
 * rasterizer_impl.h:47-54).  Backward-only in the reference; written only when enabled. */
void r2s_raster_ctx_set_aux(R2SRasterCtx* ctx, float* final_T /*[n_frames,H,W]*/, uint32_t* n_contrib);

/* Last HIP error string recorded by this library on the calling thread ("" if none). */
const char* r2s_last_error(void);
/* Raw device-to-device copy enqueued on `stream` (used by the Python host's debug taps). */
int r2s_memcpy_d2d(void* dst, const void* src, size_t bytes, r2s_stream_t stream);
/* Library / ABI version (major*100+minor). */
int r2s_version(void);

#ifdef __cplusplus
}
#endif
#endif /* R2S_RASTER_H */
