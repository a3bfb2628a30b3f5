/* ORACLE (C leg) under AddressSanitizer + UBSan - TEST INFRASTRUCTURE ONLY (`make -C oracle asan`).  Exercises oracle_c.c on the edge
 * cases the parity tests use (zero durations, t_cap == Tmax exactly, one-frame utterances, seq_len NULL / longer than T) with
 * heap buffers of exactly the documented sizes, so that any over-read or over-write is reported. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int64_t oracle_upsample_index(const int64_t* dur, int64_t B, int64_t P, int64_t t_cap, int64_t* idx, int64_t* n_frames);
void oracle_upsample_gather(const float* src, const int64_t* idx, int64_t B, int64_t P, int64_t T, int64_t F, float* out);
float oracle_masked_mse(const float* pred, const float* tgt, const int64_t* seq_len, int64_t B, int64_t T, int64_t D, float* grad);

int main(void) {
    const int64_t B = 5, P = 7, F = 3;
    int64_t* dur = malloc(sizeof(int64_t) * B * P);
    unsigned s = 12345u;
    for (int64_t i = 0; i < B * P; ++i) { s = s * 1664525u + 1013904223u; dur[i] = (s >> 24) % 4; }   /* zeros included */
    for (int64_t p = 0; p < P; ++p) dur[1 * P + p] = 0;                                              /* an empty utterance */
    int64_t* n_frames = malloc(sizeof(int64_t) * B);
    const int64_t tmax = oracle_upsample_index(dur, B, P, 0, NULL, n_frames);
    if (tmax <= 0) return 2;
    int64_t* idx = malloc(sizeof(int64_t) * B * tmax);                                               /* t_cap == Tmax exactly */
    if (oracle_upsample_index(dur, B, P, tmax, idx, n_frames) != tmax) return 3;
    if (oracle_upsample_index(dur, B, P, tmax - 1, idx, NULL) != -1) return 4;                       /* too small: refused */
    float* src = malloc(sizeof(float) * B * P * F);
    for (int64_t i = 0; i < B * P * F; ++i) src[i] = (float)i;
    float* out = malloc(sizeof(float) * B * tmax * F);
    oracle_upsample_gather(src, idx, B, P, tmax, F, out);
    for (int64_t b = 0; b < B; ++b)
        for (int64_t t = 0; t < tmax; ++t) {
            const int64_t p = idx[b * tmax + t];
            if ((t < n_frames[b]) != (p >= 0)) return 5;
            if (p >= 0 && out[(b * tmax + t) * F] != src[(b * P + p) * F]) return 6;
            if (p < 0 && out[(b * tmax + t) * F + F - 1] != 0.f) return 7;
        }
    const int64_t T = tmax, D = 2;
    float* pred = malloc(sizeof(float) * B * T * D);
    float* tgt = malloc(sizeof(float) * B * T * D);
    float* grad = malloc(sizeof(float) * B * T * D);
    for (int64_t i = 0; i < B * T * D; ++i) { pred[i] = 0.01f * (float)i; tgt[i] = 1.f; }
    int64_t* seq_len = malloc(sizeof(int64_t) * B);
    for (int64_t b = 0; b < B; ++b) seq_len[b] = 1 + b % T;
    seq_len[B - 1] = T + 3;                                                                          /* longer than the padded axis */
    const float l1 = oracle_masked_mse(pred, tgt, seq_len, B, T, D, grad);
    const float l2 = oracle_masked_mse(pred, tgt, NULL, B, T, D, NULL);
    if (!(l1 == l1) || !(l2 == l2)) return 8;
    free(dur); free(n_frames); free(idx); free(src); free(out); free(pred); free(tgt); free(grad); free(seq_len);
    printf("oracle_c asan check ok (Tmax %lld, losses %.6f %.6f)\n", (long long)tmax, l1, l2);
    return 0;
}
