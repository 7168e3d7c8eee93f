// libksmi: ABI version + thread-local error reporting.
#include <string.h>
#include "../../include/ksmi.h"
#include "errors.h"

static thread_local char g_err[512] = "";

int ksmi_fail(int code, const char* msg) {
  strncpy(g_err, msg ? msg : "unknown error", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
  return code;
}

int ksmi_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  char buf[400];
  snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  return ksmi_fail((int)e, buf);
}

extern "C" {
int ksmi_abi_version(void) { return KSMI_ABI_VERSION; }
const char* ksmi_last_error(void) { return g_err; }
}
