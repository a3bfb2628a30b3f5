// Library-level entry points: thread-local error text, ABI version.
#include "common.h"

#include <stdarg.h>
#include <string.h>

#include <thread>
#include <vector>

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

// The loader's host half: `count` utterance arrays laid back to back into one (pinned) staging buffer by `threads` host threads - the
// np.concatenate of data.collate_to_device (49 MB per C2 batch: 4-5 ms on one core, 80 ms when the destination is a fresh allocation).
// Pieces are dealt to the threads by bytes; a piece is one memcpy.  Pure host code: no HIP call, no stream.
int mg_host_pack(const void* const* srcs, const int64_t* bytes, int count, void* dst, int64_t dst_bytes, int threads) {
    MG_CHECK_ARG(srcs && bytes && dst && count > 0 && threads > 0 && threads <= 64, "mg_host_pack: bad arguments (count=%d threads=%d)", count,
                 threads);
    int64_t total = 0;
    for (int i = 0; i < count; ++i) {
        MG_CHECK_ARG(bytes[i] >= 0 && (bytes[i] == 0 || srcs[i]), "mg_host_pack: bad piece %d", i);
        total += bytes[i];
    }
    MG_CHECK_ARG(total <= dst_bytes, "mg_host_pack: %lld bytes do not fit the destination's %lld", (long long)total, (long long)dst_bytes);
    if (total == 0) return MG_OK;
    auto work = [&](int64_t lo, int64_t hi) {            // bytes [lo, hi) of the packed image
        int64_t at = 0;
        for (int i = 0; i < count && at < hi; ++i) {
            const int64_t b = bytes[i], p_lo = lo > at ? lo : at, p_hi = hi < at + b ? hi : at + b;
            if (p_hi > p_lo) memcpy((char*)dst + p_lo, (const char*)srcs[i] + (p_lo - at), (size_t)(p_hi - p_lo));
            at += b;
        }
    };
    int n = threads;
    if (total < (int64_t)n * (1 << 20)) n = (int)(total >> 20) + 1;      // at least 1 MB per thread
    if (n <= 1) {
        work(0, total);
        return MG_OK;
    }
    const int64_t per = ((total + n - 1) / n + 63) / 64 * 64;
    std::vector<std::thread> pool;
    for (int t = 1; t < n; ++t) pool.emplace_back(work, (int64_t)t * per < total ? (int64_t)t * per : total, (int64_t)(t + 1) * per < total ? (int64_t)(t + 1) * per : total);
    work(0, per < total ? per : total);
    for (auto& th : pool) th.join();
    return MG_OK;
}

}  // extern "C"
