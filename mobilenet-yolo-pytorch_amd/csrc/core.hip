// Error plumbing + version for libmnyolo.
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace mny {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return MNY_EHIP;
    }
    return MNY_OK;
}
}  // namespace mny

extern "C" int mny_version(void) { return 100; }
extern "C" const char* mny_last_error(void) { return mny::g_err; }
extern "C" int mny_max_parts(void) { return mny::kMaxParts; }
