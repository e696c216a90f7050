// libspeecht_hip.so: version + thread-local error text.
#include "st_common.h"

namespace st {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace st

extern "C" {
int st_version(void) { return 100; }
const char* st_last_error(void) { return st::g_err; }
}
