#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b)
{
    float x[8]; v2f y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = (v2f){x[i], x[i] + 1.f}; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) x[i] = fmaf(x[i], a, b);
            if (MODE == 1) y[i] = y[i] * a + b;                    // packed fma
            if (MODE == 2) x[i] = x[i] * a;                        // mul
            if (MODE == 3) x[i] = __builtin_amdgcn_rsqf(x[i]);     // transcendental
            if (MODE == 4) x[i] = x[i] + b;                        // add
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, float* d, int flops_per_op)
{
    const int blocks = 256 * 8, iters = 4096;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 16, 1.0001f, 0.5f);
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 8;   // wave-instructions
    printf("%-12s %.3f ms  %.2f T wave-instr/s  -> %.2f cycles/instr/SIMD @2.4GHz  %.1f TFLOP/s\n", name, ms, winstr / ms / 1e9,
           1024 * 2.4e9 / (winstr / (ms * 1e-3)), winstr * 64 * flops_per_op / ms / 1e9);
}
int main() { float* d; hipMalloc(&d, 256 * 8 * 256 * 4); run<0>("v_fma_f32", d, 2); run<1>("v_pk_fma_f32", d, 4); run<2>("v_mul_f32", d, 1); run<4>("v_add_f32", d, 1); run<3>("v_rsq_f32", d, 1); return 0; }
