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

int g_mg_tuning[MG_TUNING_KEYS] = {0};

extern "C" {

int mg_set_tuning(int key, int value) {
    MG_CHECK_ARG(key >= 0 && key < MG_TUNING_KEYS, "mg_set_tuning: key %d not in 0..%d", key, MG_TUNING_KEYS - 1);
    g_mg_tuning[key] = value;
    return MG_OK;
}

const char* mg_last_error(void) { return g_error; }

int mg_version(void) { return 1; }

const char* mg_build_arch(void) { return "gfx950"; }

}  // extern "C"
