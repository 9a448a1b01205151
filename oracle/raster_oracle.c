/*
 * TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See raster_oracle_impl.inc for the header.
 * Builds the float32 oracle (r2s_oracle_raster_forward_f32) and its float64 shadow
 * (r2s_oracle_raster_forward_f64) from one restatement.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define R2S_BLOCK_X 16 /* cuda_rasterizer/config.h:15-16 */
#define R2S_BLOCK_Y 16

#define R2S_REAL float
#define R2S_SUFFIX _f32
#define R2S_SQRT sqrtf
#define R2S_EXP expf
#define R2S_CEIL ceilf
#include "raster_oracle_impl.inc"
#undef R2S_REAL
#undef R2S_SUFFIX
#undef R2S_SQRT
#undef R2S_EXP
#undef R2S_CEIL

#define R2S_REAL double
#define R2S_SUFFIX _f64
#define R2S_SQRT sqrt
#define R2S_EXP exp
#define R2S_CEIL ceil
#include "raster_oracle_impl.inc"

void r2s_oracle_set_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

int r2s_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* getHigherMsb, cuda_rasterizer/rasterizer_impl.cu:35-50 — number of tile-id bits in the sort key. */
uint32_t r2s_oracle_higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4;
    uint32_t step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb) msb += step;
        else msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}
