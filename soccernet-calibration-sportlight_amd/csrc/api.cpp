// libsncal.so: version + thread-local error string.
#include "common.hpp"
#include "x3.hpp"
#include <cstring>

namespace sncal {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
LaunchEvents& launch_events() { static thread_local LaunchEvents e; return e; }
}  // namespace sncal

extern "C" int sncal_version(void) { return SNCAL_VERSION; }
extern "C" const char* sncal_last_error(void) { return sncal::g_err; }
extern "C" const char* sncal_x3_name(void) { return SNCAL_X3_NAME; }
