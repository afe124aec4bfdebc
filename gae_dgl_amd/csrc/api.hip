// Library-level entry points: version, error string, device info.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace gae {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
} // namespace gae

extern "C" int gae_version(void) { return GAE_VERSION; }

extern "C" const char *gae_last_error(void) { return gae::g_err; }

extern "C" int gae_device_info_get(int device, gae_device_info *out)
{
    GAE_REQUIRE(out != nullptr, GAE_E_NULL, "gae_device_info_get: out is NULL");
    hipDeviceProp_t p;
    GAE_HIP(hipGetDeviceProperties(&p, device));
    memset(out, 0, sizeof(*out));
    out->compute_units = p.multiProcessorCount;
    out->wavefront_size = p.warpSize;
    out->lds_bytes_per_cu = static_cast<int32_t>(p.maxSharedMemoryPerMultiProcessor);
    out->l2_bytes = p.l2CacheSize;
    out->hbm_bytes = static_cast<int64_t>(p.totalGlobalMem);
    out->clock_khz = p.clockRate;
    int gfx = 0;
    const char *a = strstr(p.gcnArchName, "gfx");
    if (a) gfx = atoi(a + 3);
    out->gfx_major_minor = gfx;
    strncpy(out->name, p.name, sizeof(out->name) - 1);
    return GAE_OK;
}
