// Shared host-side helpers of libr2s_hip (error capture, grow-only device buffers).
#pragma once
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

namespace r2s {

void set_last_error(const char* what, hipError_t e, const char* file, int line);
void set_last_error_msg(const char* msg);

#define R2S_HIP_TRY(expr)                                              \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) {                                        \
            ::r2s::set_last_error(#expr, _e, __FILE__, __LINE__);      \
            return R2S_ERR_HIP;                                        \
        }                                                              \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// hipMalloc for every device allocation of the library.  With R2S_POISON=1 in the environment (read once; a test facility) the new
// memory is filled with 0xFF bytes — NaN as float, -1 as index — so that code which relies on fresh allocations being zero, or
// reads entries it never wrote, fails loudly instead of working until freed memory is reused (tests/ run the GPU suite this way).
inline hipError_t dev_malloc(void** p, size_t bytes)
{
    static const bool poison = [] { const char* e = getenv("R2S_POISON"); return e && e[0] && e[0] != '0'; }();
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess && poison) e = hipMemset(*p, 0xFF, bytes);
    return e;
}

// Grow-only device buffer (the role torch's resize_ plays for the reference's scratch tensors).
struct DevBuf {
    char* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) {
            hipError_t e = hipFree(p); // implicit device sync: safe w.r.t. in-flight users
            p = nullptr;
            cap = 0;
            if (e != hipSuccess) return e;
        }
        size_t want = align_up(bytes + bytes / 4, 1 << 20);
        hipError_t e = dev_malloc((void**)&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Bump allocator over a byte chunk with 128-byte aligned sub-allocations
// (the reference's obtain(), rasterizer_impl.h:19-28).
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(char* b) : base(b) {}
    template <typename T>
    T* take(size_t count)
    {
        off = align_up(off, 128);
        T* r = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return r;
    }
    size_t bytes() const { return align_up(off, 128) + 128; }
};

} // namespace r2s
