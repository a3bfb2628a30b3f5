// Library-level entry points: thread-local error text, ABI version.
#include "common.h"

#include <stdarg.h>

static thread_local char g_error[512] = "";

void mg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

extern "C" {

const char* mg_last_error(void) { return g_error; }

int mg_version(void) { return 1; }

const char* mg_build_arch(void) { return "gfx950"; }

}  // extern "C"
