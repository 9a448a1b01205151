// Error capture and version of libr2s_hip.
#include "r2s_common.h"
#include "../../include/r2s_raster.h"

namespace r2s {
static thread_local char g_err[512] = "";

void set_last_error(const char* what, hipError_t e, const char* file, int line)
{
    snprintf(g_err, sizeof(g_err), "%s failed: %s (%d) at %s:%d", what, hipGetErrorString(e), (int)e, file, line);
}
void set_last_error_msg(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }
} // namespace r2s

extern "C" const char* r2s_last_error(void) { return r2s::g_err; }
extern "C" int r2s_version(void) { return 100; }

// Raw device-to-device copy on `stream` (debug taps of the Python host; not on any hot path).
extern "C" int r2s_memcpy_d2d(void* dst, const void* src, size_t bytes, r2s_stream_t stream)
{
    if (bytes == 0) return R2S_OK;
    if (!dst || !src) return R2S_ERR_INVALID;
    R2S_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return R2S_OK;
}
