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

// kernels started by this thread's launchers since the last ksmi_last_kernels() (the __PRETTY_FUNCTION__ of ksmi_kernel_pretty<&kernel>)
static thread_local const char* g_notes[8];
static thread_local int g_nnotes = 0;
void ksmi_note_kernel(const char* pretty) {
  if (g_nnotes < 8) g_notes[g_nnotes] = pretty;
  ++g_nnotes;
}

extern "C" {
int ksmi_last_kernels(char* buf, int cap) {
  int n = g_nnotes < 8 ? g_nnotes : 8, pos = 0;
  if (buf && cap > 0) buf[0] = 0;
  for (int i = 0; i < n && buf; ++i) {
    // "... [K = &(anonymous namespace)::name<args>]" -> "name<args>", written the way profiles/summarize.py cleans rocprofv3's names
    const char* p = strstr(g_notes[i], "K = &");
    p = p ? p + 5 : g_notes[i];
    const char* e = strrchr(p, ']');
    int len = e ? (int)(e - p) : (int)strlen(p);
    if (i && pos < cap - 1) buf[pos++] = ';';
    for (int j = 0; j < len && pos < cap - 1;) {
      if (!strncmp(p + j, "(anonymous namespace)::", 23)) { j += 23; continue; }
      if (!strncmp(p + j, "unsigned short", 14)) { if (pos + 4 < cap) { memcpy(buf + pos, "bf16", 4); pos += 4; } j += 14; continue; }
      buf[pos++] = p[j++];
    }
    buf[pos] = 0;
  }
  g_nnotes = 0;
  return n;
}
int ksmi_abi_version(void) { return KSMI_ABI_VERSION; }
const char* ksmi_last_error(void) { return g_err; }
}
