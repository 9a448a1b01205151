// Skeleton of a persistent env-step kernel: 960 workgroups x 512 threads, one (block, env) item each, 667 iterations of
// {compute delay -> publish 512 records -> flag -> wait 8 neighbour flags -> gather ~900 halo records}.  Measures us / iteration
// and checks every gathered record carries the expected epoch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

struct Args {
    u64* pub[2];        // [items][512][3]
    unsigned* epoch;    // [items]
    unsigned* err;      // [4]
    float* sink;
    int items, per_env, ndep, halo, iters, delay;
};

__global__ void __launch_bounds__(512) k_persist(Args a)
{
    extern __shared__ u64 lds[]; // halo staging
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int ipx = (a.items + 7) / 8;
    const int item = xcd * ipx + q;
    if (q >= ipx || item >= a.items) return;
    const int env = item / a.per_env, b = item % a.per_env;
    const int tid = threadIdx.x;
    float acc = tid * 1e-3f;
    for (int s = 1; s <= a.iters; ++s) {
        // "compute"
        for (int k = 0; k < a.delay; ++k) acc = fmaf(acc, 1.0001f, 0.5f);
        // publish own records (3 x 8 B per thread), tagged with the epoch
        u64* dst = a.pub[s & 1] + ((size_t)item * 512 + tid) * 3;
        const u64 tag = ((u64)s << 32) | (unsigned)tid;
        __hip_atomic_store(dst + 0, tag, RLX, AGENT);
        __hip_atomic_store(dst + 1, tag + 1, RLX, AGENT);
        __hip_atomic_store(dst + 2, tag + 2, RLX, AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.epoch + item, (unsigned)s, RLX, AGENT);
        // wait for the neighbours (same env, blocks b-ndep/2 .. b+ndep/2)
        if (tid < a.ndep) {
            int nb = b - a.ndep / 2 + tid; if (nb >= b) nb++;
            if (nb >= 0 && nb < a.per_env) {
                const unsigned* f = a.epoch + env * a.per_env + nb;
                unsigned spins = 0;
                while (__hip_atomic_load(f, RLX, AGENT) < (unsigned)s) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > 4000000u) { atomicExch(a.err, 1u); break; }
                }
            }
        }
        __syncthreads();
        // gather halo: `halo` records spread over the neighbours
        for (int h = tid; h < a.halo; h += 512) {
            int k = h % a.ndep; int nb = b - a.ndep / 2 + k; if (nb >= b) nb++;
            if (nb < 0 || nb >= a.per_env) continue;
            const int r = (h * 7) & 511;
            const u64* src = a.pub[s & 1] + ((size_t)(env * a.per_env + nb) * 512 + r) * 3;
            const u64 v0 = __hip_atomic_load(src, RLX, AGENT), v1 = __hip_atomic_load(src + 1, RLX, AGENT), v2 = __hip_atomic_load(src + 2, RLX, AGENT);
            if ((unsigned)(v0 >> 32) != (unsigned)s || (unsigned)(v1 >> 32) != (unsigned)s || (unsigned)(v2 >> 32) != (unsigned)s) atomicAdd(a.err + 1, 1u);
            lds[h] = v0 ^ v1 ^ v2;
        }
        __syncthreads();
        acc += (float)(lds[tid % (a.halo > 0 ? a.halo : 1)] & 1);
    }
    a.sink[item * 512 + tid] = acc;
}

int main(int argc, char** argv)
{
    Args a{};
    a.per_env = 30; const int envs = 32; a.items = envs * a.per_env; a.ndep = 8; a.halo = 922; a.iters = 667;
    a.delay = argc > 1 ? atoi(argv[1]) : 2000;
    for (int k = 0; k < 2; ++k) { hipMalloc(&a.pub[k], sizeof(u64) * 3 * 512 * a.items); hipMemset(a.pub[k], 0, sizeof(u64) * 3 * 512 * a.items); }
    hipMalloc(&a.epoch, 4 * a.items); hipMalloc(&a.err, 16); hipMalloc(&a.sink, 4 * 512 * a.items);
    const int grid = 8 * ((a.items + 7) / 8);
    int maxb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_persist, 512, 34416);
    printf("occupancy API: %d blocks/CU -> %d resident >= %d needed\n", maxb, maxb * 256, grid);
    if (maxb * 256 < grid) { printf("not resident, abort\n"); return 1; }
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(a.epoch, 0, 4 * a.items); hipMemset(a.err, 0, 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_persist, dim3(grid), dim3(512), 34416, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned err[4]; hipMemcpy(err, a.err, 16, hipMemcpyDeviceToHost);
        printf("delay %d: %.3f ms total, %.2f us / iteration, timeout %u, stale records %u\n", a.delay, ms, ms * 1e3 / a.iters, err[0], err[1]);
    }
    return 0;
}
