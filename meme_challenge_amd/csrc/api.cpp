// Error reporting / version entry points of libuniter_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include "common.h"

static thread_local char g_err[512] = "";

void uniter_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int uniter_abi_version(void) { return UNITER_ABI_VERSION; }
extern "C" const char* uniter_last_error(void) { return g_err; }
extern "C" const char* uniter_build_info(void) {
  return "libuniter_hip gfx950 fp32-mfma (built " __DATE__ " " __TIME__ ")";
}

// the "next launch" side channel of common.h: one per HOST THREAD, so that two threads driving two model handles (cross-
// validation folds in threads, a DataParallel-style caller) can never take each other's stamp slot or wave priority
thread_local unsigned long long* g_uniter_stamp_slot = nullptr;
thread_local int g_uniter_launch_prio = 0;
thread_local int g_uniter_cu_reserve = 0;
thread_local const unsigned char* g_uniter_drop_bits = nullptr;
