/* ORACLE (C leg) - plain-C restatement of the integer / reduction parts of the reference hot path.
 * TEST INFRASTRUCTURE ONLY: built by oracle/Makefile into oracle/liboracle_c.so and loaded with ctypes from tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product never links or loads it.
 *
 * Follows ZackHodari/morgana:
 *   oracle_upsample_index   morgana/utils.py:198-220   (np.repeat per utterance into a -1 initialised int64 map)
 *   oracle_upsample_gather  morgana/utils.py:206-226   (index -1 gathers the appended zero row)
 *   oracle_masked_mse       morgana/losses.py:29-44    (per-utterance masked mean, then mean over (B, D))
 */
#include <stdint.h>
#include <string.h>

/* Returns Tmax (max_b sum_p dur[b,p]); if idx != NULL it must hold B*t_cap entries and t_cap >= Tmax. */
int64_t oracle_upsample_index(const int64_t* dur, int64_t B, int64_t P, int64_t t_cap, int64_t* idx, int64_t* n_frames)
{
    int64_t tmax = 0;
    for (int64_t b = 0; b < B; ++b) {
        int64_t total = 0;
        for (int64_t p = 0; p < P; ++p) total += dur[b * P + p];
        if (n_frames) n_frames[b] = total;
        if (total > tmax) tmax = total;
    }
    if (!idx) return tmax;
    if (t_cap < tmax) return -1;
    for (int64_t b = 0; b < B; ++b) {
        int64_t* row = idx + b * t_cap;
        int64_t t = 0;
        for (int64_t p = 0; p < P; ++p)
            for (int64_t k = 0; k < dur[b * P + p]; ++k) row[t++] = p;
        for (; t < t_cap; ++t) row[t] = -1;
    }
    return tmax;
}

void oracle_upsample_gather(const float* src, const int64_t* idx, int64_t B, int64_t P, int64_t T, int64_t F, float* out)
{
    for (int64_t b = 0; b < B; ++b)
        for (int64_t t = 0; t < T; ++t) {
            int64_t p = idx[b * T + t];
            float* dst = out + (b * T + t) * F;
            if (p < 0) memset(dst, 0, (size_t)F * sizeof(float));
            else memcpy(dst, src + (b * P + p) * F, (size_t)F * sizeof(float));
        }
}

/* seq_len may be NULL (divide by T).  grad may be NULL.  float accumulation, t-major like torch.sum(dim=1). */
float oracle_masked_mse(const float* pred, const float* tgt, const int64_t* seq_len, int64_t B, int64_t T, int64_t D,
                        float* grad)
{
    float total = 0.0f;
    for (int64_t b = 0; b < B; ++b) {
        int64_t n = seq_len ? seq_len[b] : T;
        float nf = (float)(n < T ? n : T);
        if (!seq_len) nf = (float)T;
        for (int64_t d = 0; d < D; ++d) {
            float s = 0.0f;
            for (int64_t t = 0; t < T; ++t) {
                float m = (t < n) ? 1.0f : 0.0f;
                float e = pred[(b * T + t) * D + d] - tgt[(b * T + t) * D + d];
                s += e * e * m;
                if (grad) grad[(b * T + t) * D + d] = 2.0f * e * m / (nf * (float)(B * D));
            }
            total += s / nf;
        }
    }
    return total / (float)(B * D);
}
