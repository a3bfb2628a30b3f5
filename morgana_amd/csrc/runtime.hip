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

// Which (key, value) pairs the PRODUCT library takes: choices between kernels / plans that compute the same results.  The lab
// builds (make lab / diag: -DMG_EXPERIMENTS) take every value - measured-slower experiments and timing probes live there only.
static bool tuning_allowed(int key, int value) {
#ifdef MG_EXPERIMENTS
    (void)key;
    (void)value;
    return true;
#else
    switch (key) {
        case MG_TUNE_FORM: return value == 0 || value == 3 || value == 6 || value == 7 || value == 12 || value == 13 || value == 14 || value == 16;
        case MG_TUNE_GRU_HANDOFF: return value >= 0 && value <= 7;
        case MG_TUNE_PERSISTENT: return value >= 0 && value <= 2;
        case MG_TUNE_WGRAD_SPLITS: return value >= 0;
        case MG_TUNE_WGRAD_ORDER: return value >= 0 && value <= 3;
        case MG_TUNE_LSTM_BWD_STACK: return value >= 0 && value <= 3;
        case MG_TUNE_AB: return value == 0 || value == 65 || value == 66 || value == 86 || value == 87 || value == 88 || value == 89 || value == 90 || value == 91 || value == 92 || value == 93 || value == 94 || value == 95 || value == 96 || value == 97 || value == 98 || value == 99;
        default: return false;
    }
#endif
}

int mg_set_tuning(int key, int value) {
    MG_CHECK_ARG(key >= 0 && key < MG_TUNING_KEYS, "mg_set_tuning: key %d not in 0..%d", key, MG_TUNING_KEYS - 1);
    MG_CHECK_ARG(tuning_allowed(key, value),
                 "mg_set_tuning: key %d does not take value %d in the product library (experiments and timing probes are compiled into the "
                 "lab builds only: make -C morgana_amd/csrc lab)", key, value);
    g_mg_tuning[key] = value;
    return MG_OK;
}

const char* mg_last_error(void) { return g_error; }

int mg_version(void) { return 1; }

const char* mg_build_arch(void) { return "gfx950"; }

}  // extern "C"
