#include "k3_common.h"
#include <cstring>
namespace k3 {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
}  // namespace k3
extern "C" const char *k3_last_error(void) { return k3::g_err; }
extern "C" int k3_version(void) { return 1; }
