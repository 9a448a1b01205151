// Micro-benchmark: cost of executing straight-line code once per wavefront (instruction fetch), vs the same work in a loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP256(x) REP16(REP16(x))
#define REP1K(x) REP4(REP256(x))
#define OP asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
template <int KINSTR, int TAG>
__global__ void __launch_bounds__(64) k_line(float* out, float b, float c)
{
    float a = (float)threadIdx.x + TAG;
    if (KINSTR >= 1) { REP1K(OP) }
    if (KINSTR >= 2) { REP1K(OP) }
    if (KINSTR >= 4) { REP1K(OP) REP1K(OP) }
    if (KINSTR >= 8) { REP1K(OP) REP1K(OP) REP1K(OP) REP1K(OP) }
    if (KINSTR >= 12) { REP1K(OP) REP1K(OP) REP1K(OP) REP1K(OP) }
    if (KINSTR >= 16) { REP1K(OP) REP1K(OP) REP1K(OP) REP1K(OP) }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
__global__ void __launch_bounds__(64) k_loop(float* out, float b, float c, int n)
{
    float a = (float)threadIdx.x;
    for (int i = 0; i < n; ++i) { REP16(OP) }
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
template <class F> float timeit(F f, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main()
{
    float* out; hipMalloc(&out, 4 * 64 * 4096);
    const int blocks = 128;
    printf("blocks %d (one wave each)\n", blocks);
    float base = timeit([&] { hipLaunchKernelGGL(k_loop, dim3(blocks), dim3(64), 0, 0, out, 1.f, 2.f, 0); }, 200);
    printf("empty-ish kernel: %.2f us per launch\n", base);
#define RUN(K) { float t = timeit([&] { hipLaunchKernelGGL((k_line<K, 0>), dim3(blocks), dim3(64), 0, 0, out, 1.f, 2.f); }, 200); \
                 float tl = timeit([&] { hipLaunchKernelGGL(k_loop, dim3(blocks), dim3(64), 0, 0, out, 1.f, 2.f, K * 1024 / 16); }, 200); \
                 float ta = timeit([&] { hipLaunchKernelGGL((k_line<K, 0>), dim3(blocks), dim3(64), 0, 0, out, 1.f, 2.f); hipLaunchKernelGGL((k_line<16, 1>), dim3(blocks), dim3(64), 0, 0, out, 1.f, 2.f); }, 200); \
                 printf("%2dk instr (%3d KB): straight-line alone %.2f us, same work in a loop %.2f us, alternating with another 64 KB kernel %.2f us (pair)\n", K, K * 4, t, tl, ta); }
    RUN(1) RUN(2) RUN(4) RUN(8) RUN(12) RUN(16)
    float t16b = timeit([&] { hipLaunchKernelGGL((k_line<16, 1>), dim3(blocks), dim3(64), 0, 0, out, 1.f, 2.f); }, 200);
    printf("the other 64 KB kernel alone: %.2f us\n", t16b);
    // many waves per CU: do followers hit?
    for (int b : {128, 1024, 8192}) {
        float t = timeit([&] { hipLaunchKernelGGL((k_line<8, 0>), dim3(b), dim3(64), 0, 0, out, 1.f, 2.f); hipLaunchKernelGGL((k_line<16, 1>), dim3(128), dim3(64), 0, 0, out, 1.f, 2.f); }, 100);
        printf("8k-instr kernel with %d blocks (+ the thrasher): %.2f us\n", b, t);
    }
    return 0;
}
