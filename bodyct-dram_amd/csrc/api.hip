// Error reporting and ABI version of libdram_hip.so.
#include "common.h"
#include <string.h>

namespace dram {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace dram

extern "C" const char* dram_last_error(void) { return dram::g_err; }
extern "C" int dram_abi_version(void) { return 1; }
