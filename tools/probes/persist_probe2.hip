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

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
#define SC (1 | 16) /* sc0 | sc1 on gfx94x/gfx950 buffer cache policy */
struct Args {
    int mode;
    long long* prof;    // [items][4] accumulated wall-clock ticks: publish, wait, gather, total
    u64* pub[2];        // [items][512][4]
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
    long long t_pub = 0, t_wait = 0, t_gat = 0;
    const long long t_begin = wall_clock64();
    __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.pub[0], 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.pub[1], 0, 0x7fffffff, 0x00020000);
    for (int s = 1; s <= a.iters; ++s) {
        const long long c0 = wall_clock64();
        // "compute"
        for (int k = 0; k < a.delay; ++k) acc = fmaf(acc, 1.0001f, 0.5f);
        // publish own records (3 x 8 B per thread), tagged with the epoch
        u64* dst = a.pub[s & 1] + ((size_t)item * 512 + tid) * 4;
        const u64 tag = ((u64)s << 32) | (unsigned)tid;
        if (a.mode == 0) {
            __hip_atomic_store(dst + 0, tag, RLX, AGENT);
            __hip_atomic_store(dst + 1, tag + 1, RLX, AGENT);
            __hip_atomic_store(dst + 2, tag + 2, RLX, AGENT);
        } else {
            const i4 v0 = {s, tid, 0, 1}, v1 = {s, tid, 2, 3};
            const int off = (item * 512 + tid) * 32;
            __builtin_amdgcn_raw_buffer_store_b128(v0, (s & 1) ? r1 : r0, off, 0, SC);
            __builtin_amdgcn_raw_buffer_store_b128(v1, (s & 1) ? r1 : r0, off + 16, 0, SC);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.epoch + item, (unsigned)s, RLX, AGENT);
        const long long c1 = wall_clock64();
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
        const long long c2 = wall_clock64();
        // gather halo: `halo` records spread over the neighbours
        for (int h = tid; h < a.halo; h += 512) {
            int k = h % a.ndep; int nb = b - a.ndep / 2 + k; if (nb >= b) nb++;
            if (nb < 0 || nb >= a.per_env) continue;
            const int r = (h * 7) & 511;
            const u64* src = a.pub[s & 1] + ((size_t)(env * a.per_env + nb) * 512 + r) * 4;
            if (a.mode == 0) {
                const u64 v0 = __hip_atomic_load(src, RLX, AGENT), v1 = __hip_atomic_load(src + 1, RLX, AGENT), v2 = __hip_atomic_load(src + 2, RLX, AGENT);
                if ((unsigned)(v0 >> 32) != (unsigned)s || (unsigned)(v1 >> 32) != (unsigned)s || (unsigned)(v2 >> 32) != (unsigned)s) atomicAdd(a.err + 1, 1u);
                lds[h] = v0 ^ v1 ^ v2;
            } else {
                const int off = ((env * a.per_env + nb) * 512 + r) * 32;
                const i4 v0 = __builtin_amdgcn_raw_buffer_load_b128((s & 1) ? r1 : r0, off, 0, SC);
                const i4 v1 = __builtin_amdgcn_raw_buffer_load_b128((s & 1) ? r1 : r0, off + 16, 0, SC);
                if (v0.x != s || v1.x != s) atomicAdd(a.err + 1, 1u);
                lds[h] = (u64)(unsigned)(v0.y + v1.w);
            }
        }
        __syncthreads();
        acc += (float)(lds[tid % (a.halo > 0 ? a.halo : 1)] & 1);
        const long long c3 = wall_clock64();
        t_pub += c1 - c0; t_wait += c2 - c1; t_gat += c3 - c2;
    }
    if (tid == 0) { a.prof[item * 4] = t_pub; a.prof[item * 4 + 1] = t_wait; a.prof[item * 4 + 2] = t_gat; a.prof[item * 4 + 3] = wall_clock64() - t_begin; a.prof[4 * a.items + item] = t_begin; }
    a.sink[item * 512 + tid] = acc;
}

int main(int argc, char** argv)
{
    Args a{};
    a.per_env = 30; const int envs = 32; a.items = envs * a.per_env; a.ndep = 8; a.halo = 922; a.iters = 667;
    a.delay = argc > 1 ? atoi(argv[1]) : 2000; a.mode = argc > 2 ? atoi(argv[2]) : 0; const int ldsb = argc > 3 ? atoi(argv[3]) : 34416;
    for (int k = 0; k < 2; ++k) { hipMalloc(&a.pub[k], sizeof(u64) * 4 * 512 * a.items); hipMemset(a.pub[k], 0, sizeof(u64) * 4 * 512 * a.items); }
    hipMalloc(&a.epoch, 4 * a.items); hipMalloc(&a.prof, 40 * a.items); hipMalloc(&a.err, 16); hipMalloc(&a.sink, 4 * 512 * a.items);
    const int grid = 8 * ((a.items + 7) / 8);
    int maxb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_persist, 512, 34416);
    printf("occupancy API: %d blocks/CU -> %d resident >= %d needed\n", maxb, maxb * 256, grid);
    if (maxb * 256 < grid) { printf("not resident, abort\n"); return 1; }
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(a.epoch, 0, 4 * a.items); hipMemset(a.err, 0, 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_persist, dim3(grid), dim3(512), ldsb, 0, a);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned err[4]; hipMemcpy(err, a.err, 16, hipMemcpyDeviceToHost);
        std::vector<long long> pr(5 * a.items); hipMemcpy(pr.data(), a.prof, 40 * a.items, hipMemcpyDeviceToHost);
        long long s0 = pr[4 * a.items], s1 = s0, eL = 0; for (int i = 0; i < a.items; ++i) { long long b = pr[4 * a.items + i]; s0 = b < s0 ? b : s0; s1 = b > s1 ? b : s1; long long e = b + pr[i * 4 + 3]; eL = e > eL ? e : eL; }
        { int early = 0; for (int i = 0; i < a.items; ++i) if (pr[4 * a.items + i] - s0 < 10000) ++early; printf("blocks started within 100 us: %d of %d; ", early, a.items); }
        printf("start spread %.1f us, first start -> last end %.1f us\n", (s1 - s0) * 0.01, (eL - s0) * 0.01);
        double m[4] = {0, 0, 0, 0}; for (int i = 0; i < a.items; ++i) for (int k = 0; k < 4; ++k) m[k] += pr[i * 4 + k] * 0.01 / a.iters / a.items;
        printf("mode %d delay %d: %.2f us / iteration (publish %.2f wait %.2f gather %.2f; sum %.2f), timeout %u, stale %u\n", a.mode, a.delay, ms * 1e3 / a.iters, m[0], m[1], m[2], m[3], err[0], err[1]);
    }
    return 0;
}
