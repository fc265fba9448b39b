// Error reporting and ABI version of libmssr_hip.so.
#include "common.hpp"

namespace mssr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mssr

extern "C" const char* mssr_last_error(void) { return mssr::g_err; }
extern "C" int mssr_abi_version(void) { return 1; }
