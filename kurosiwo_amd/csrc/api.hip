// libksmi: ABI version + thread-local error reporting.
#include <string.h>
#include <stdlib.h>
#include <cxxabi.h>
#include <map>
#include <mutex>
#include <string>
#include <hip/hip_runtime.h>
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

// kernels started by this thread's launchers since the last ksmi_last_kernels() (host function pointers, resolved on demand)
static thread_local const void* g_notes[8];
static thread_local int g_nnotes = 0;
void ksmi_note_kernel(const void* fn) {
  if (g_nnotes < 8) g_notes[g_nnotes] = fn;
  ++g_nnotes;
}

// pointer -> "name<template args>" as rocprofv3 prints it after profiles/summarize.py's clean-up: demangled, without the return type,
// the "(anonymous namespace)::" qualifier and the parameter list; the 16-bit storage type spelled bf16
static const std::string& kernel_name(const void* fn) {
  static std::mutex mu;
  static std::map<const void*, std::string> cache;
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(fn);
  if (it != cache.end()) return it->second;
  std::string out = "?";
  const char* mangled = hipKernelNameRefByPtr(fn, nullptr);
  if (mangled) {
    int status = 0;
    char* dm = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
    std::string s = (status == 0 && dm) ? dm : mangled;
    free(dm);
    if (!s.compare(0, 5, "void ")) s.erase(0, 5);
    for (size_t p; (p = s.find("(anonymous namespace)::")) != std::string::npos;) s.erase(p, 23);
    for (size_t p; (p = s.find("unsigned short")) != std::string::npos;) s.replace(p, 14, "bf16");
    // drop the parameter list: the '(' that follows the closing '>' of the template arguments (or the first '(' of a non-template)
    int depth = 0;
    size_t cut = std::string::npos;
    for (size_t i = 0; i < s.size(); ++i) {
      if (s[i] == '<') ++depth;
      else if (s[i] == '>') --depth;
      else if (s[i] == '(' && depth == 0) { cut = i; break; }
    }
    if (cut != std::string::npos) s.erase(cut);
    out = s;
  }
  return cache.emplace(fn, out).first->second;
}

// ---- knobs: the test / probe switches that must be changeable WHILE a process runs (the launchers' start-up switches stay
// `static const ... getenv` reads: evaluated once).  A knob starts as the environment variable of its name read ONCE, at its first
// look-up; afterwards only ksmi_set_knob changes it -- no launcher calls getenv per launch.
static std::mutex g_knob_mu;
static std::map<std::string, std::pair<bool, std::string>> g_knobs;     // name -> (set?, value)

static std::pair<bool, std::string>& knob_slot(const char* name) {
  auto it = g_knobs.find(name);
  if (it == g_knobs.end()) {
    const char* e = getenv(name);
    it = g_knobs.emplace(name, std::make_pair(e != nullptr, std::string(e ? e : ""))).first;
  }
  return it->second;
}
int ksmi_knob_int(const char* name, int dflt) {
  std::lock_guard<std::mutex> lk(g_knob_mu);
  const auto& k = knob_slot(name);
  return k.first ? atoi(k.second.c_str()) : dflt;
}
bool ksmi_knob_is_set(const char* name) {
  std::lock_guard<std::mutex> lk(g_knob_mu);
  return knob_slot(name).first;
}
bool ksmi_knob_str(const char* name, char* out, int cap) {
  std::lock_guard<std::mutex> lk(g_knob_mu);
  const auto& k = knob_slot(name);
  if (!k.first || cap < 1) return false;
  strncpy(out, k.second.c_str(), (size_t)cap - 1);
  out[cap - 1] = 0;
  return true;
}

extern "C" {
int ksmi_set_knob(const char* name, const char* value) {
  if (!name || strncmp(name, "KSMI_", 5)) return ksmi_fail(KSMI_E_ARG, "set_knob: names start with KSMI_");
  std::lock_guard<std::mutex> lk(g_knob_mu);
  auto& k = knob_slot(name);
  k.first = value != nullptr;
  k.second = value ? value : "";
  return 0;
}
int ksmi_last_kernels(char* buf, int cap) {
  const int n = g_nnotes < 8 ? g_nnotes : 8;
  int pos = 0;
  if (buf && cap > 0) buf[0] = 0;
  for (int i = 0; i < n && buf; ++i) {
    const std::string& nm = kernel_name(g_notes[i]);
    if (i && pos < cap - 1) buf[pos++] = ';';
    for (size_t j = 0; j < nm.size() && pos < cap - 1; ++j) buf[pos++] = nm[j];
    buf[pos] = 0;
  }
  g_nnotes = 0;
  return n;
}
int ksmi_abi_version(void) { return KSMI_ABI_VERSION; }
const char* ksmi_last_error(void) { return g_err; }
}
