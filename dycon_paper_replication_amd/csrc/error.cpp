// Thread-local error string + version for libdycon_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/dycon_hip.h"

static thread_local char g_err[512] = "";

void dycon_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dycon_last_error(void) { return g_err; }
extern "C" int dycon_version(void) { return 100; }
