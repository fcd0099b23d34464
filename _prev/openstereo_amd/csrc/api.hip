// C-ABI plumbing: version, target, thread-local error string.
#include "osa_common.h"
#include <cstring>

namespace osa {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace osa

extern "C" int osa_abi_version(void) { return OSA_ABI_VERSION; }
extern "C" const char* osa_last_error(void) { return osa::g_err; }
extern "C" const char* osa_target_arch(void) { return "gfx950"; }
