// Library services: error text, launch counters, ABI version.
#include "p3d_common.h"

namespace p3d {
static thread_local char g_err[512] = "";
static std::atomic<uint64_t> g_launches[FAM_COUNT];

void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
void count_launch(int family) { g_launches[family].fetch_add(1, std::memory_order_relaxed); }
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(P3D_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return P3D_OK;
}
} // namespace p3d

extern "C" {
const char* p3d_last_error(void) { return p3d::g_err; }
int p3d_abi_version(void) { return 3; }
uint64_t p3d_launch_count(void) {
    uint64_t s = 0; for (int i = 0; i < p3d::FAM_COUNT; ++i) s += p3d::g_launches[i].load(); return s;
}
uint64_t p3d_launch_count_of(int which) {
    return (which >= 0 && which < p3d::FAM_COUNT) ? p3d::g_launches[which].load() : 0;
}
}
