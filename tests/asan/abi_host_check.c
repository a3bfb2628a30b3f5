/* Host-side AddressSanitizer pass over libmorgana_hip's C ABI (sanitizers run on the CPU build only).
 * Built by `make -C morgana_amd/csrc asan-check` against libmorgana_hip_asan.so, whose HOST code (argument validation, split /
 * workspace planning, error formatting, launch set-up) is instrumented; device code is not.  No GPU is needed: calls are made
 * with arguments that validation must reject (MG_EINVAL) or - with plausible shapes and fake, never dereferenced device
 * pointers - run up to the launch, which fails with MG_ELAUNCH on a box without a device.  The pass is "no ASan report and
 * every return code is one the header documents". */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "morgana_hip.h"

static int n_calls = 0, n_bad = 0;

static void expect_code(const char* what, int rc) {
    ++n_calls;
    if (rc != MG_OK && rc != MG_EINVAL && rc != MG_ELAUNCH && rc != MG_EWORKSPACE) {
        fprintf(stderr, "%s: undocumented return code %d\n", what, rc);
        ++n_bad;
    }
    if (rc != MG_OK) {
        const char* msg = mg_last_error();
        if (!msg || strlen(msg) == 0 || strlen(msg) > 4096) {
            fprintf(stderr, "%s: code %d without a sane error text\n", what, rc);
            ++n_bad;
        }
    }
}

int main(void) {
    /* fake device addresses: 16-byte aligned, never dereferenced on the host */
    void* d = (void*)(uintptr_t)0x7f0000001000ull;
    float* df = (float*)d;
    uint16_t* dh = (uint16_t*)d;
    int32_t* di = (int32_t*)d;
    int64_t* dl = (int64_t*)d;

    printf("arch %s version %d\n", mg_build_arch(), mg_version());
    expect_code("mg_set_tuning", mg_set_tuning(0, 0));
    expect_code("mg_set_tuning(bad)", mg_set_tuning(-5, 1) == 0 ? MG_OK : MG_EINVAL);

    /* every size helper over a sweep that includes zero, negative and near-overflow shapes */
    const int dims[] = {-1, 0, 1, 7, 64, 187, 512, 600, 1536, 65536, 2147483647};
    const int nd = (int)(sizeof(dims) / sizeof(dims[0]));
    size_t acc = 0;
    for (int i = 0; i < nd; ++i)
        for (int j = 0; j < nd; ++j) {
            const int a = dims[i], b = dims[j];
            acc += mg_masked_mse_workspace_bytes(a, b, 1) + mg_masked_mse_workspace_bytes(a, 1000, b);
            acc += mg_stream_loss_workspace_bytes(a, b, 199);
            acc += mg_pad_rows_colsum_workspace_bytes(a, b, 187);
            acc += mg_linear_wgrad_workspace_bytes((int64_t)a * 1000, b, 600) + mg_linear_wgrad_workspace_bytes(a, 512, b);
            acc += mg_linear_bwd_fused_workspace_bytes((int64_t)a * 1000, b, 600);
            acc += mg_f0_tail_workspace_bytes((int64_t)a * b) + mg_f0_l2tail_workspace_bytes((int64_t)a * b);
            acc += mg_gru_bwd_workspace_bytes(a, b) + mg_gru_persist_workspace_bytes(a, b);
            acc += mg_phone_target_stats_workspace_bytes(a, b);
            acc += (size_t)mg_gru_persist_supported(a, 1000, b) + (size_t)mg_gru_persist_f32_supported(a, 1000, b) +
                   (size_t)mg_lstm_persist_supported(a, 1000, b) + (size_t)mg_gru_small_supported(b);
            n_calls += 16;
        }
    acc += mg_metric_workspace_bytes();
    const int win_l[3] = {0, 1, 1}, win_u[3] = {0, 1, 1};
    acc += mg_mlpg_workspace_bytes(64, 1000, 60, 100, 3, win_l, win_u) + mg_mlpg_workspace_bytes(0, 0, 0, 0, 0, NULL, NULL);
    printf("size helpers ok (checksum %zu)\n", acc);

    /* rejected by validation: NULL pointers, non-positive or overflowing shapes, misaligned leading dimensions */
    expect_code("upsample_lengths(null)", mg_upsample_lengths(NULL, 0, 0, NULL, NULL, NULL));
    expect_code("upsample_index(null)", mg_upsample_index(NULL, 4, 4, 16, NULL, NULL, NULL));
    expect_code("upsample_index(P too large)", mg_upsample_index(dl, 4, 1 << 20, 16, dl, di, NULL));
    expect_code("upsample_index(overflow)", mg_upsample_index(dl, 1 << 20, 8192, 16, dl, di, NULL));
    expect_code("gather_rows_f32(null)", mg_gather_rows_f32(NULL, NULL, NULL, 10, 8, NULL));
    expect_code("gather_rows_bf16(ldo)", mg_gather_rows_bf16(df, di, dh, 10, 600, 7, NULL));
    expect_code("segment_index(null)", mg_segment_index(NULL, 1, 1, 1, 1, NULL, NULL, NULL));
    expect_code("scatter_rows(null)", mg_scatter_rows_f32(NULL, NULL, NULL, 4, 4, NULL));
    expect_code("frame_layout(null)", mg_frame_layout(NULL, 0, 0, 0, NULL, NULL, NULL, NULL));
    expect_code("frame_layout(total > B*T)", mg_frame_layout(dl, 4, 10, 41, di, di, di, NULL));
    expect_code("frame_layout(overflow)", mg_frame_layout(dl, 1 << 20, 1 << 12, 5, di, di, di, NULL));
    expect_code("pad_rows_colsum(small ws)", mg_pad_rows_colsum_f32(df, dl, 4, 100, 187, df, d, 16, NULL));
    expect_code("sequence_mask(null)", mg_sequence_mask(NULL, 0, 0, NULL, 0, 0, NULL));
    expect_code("masked_mse(small ws)", mg_masked_mse_f32(df, df, dl, 4, 100, 1, 1.f, df, df, d, 1, NULL));
    expect_code("masked_mse(null)", mg_masked_mse_f32(NULL, NULL, NULL, 0, 0, 0, 1.f, NULL, NULL, NULL, 0, NULL));
    expect_code("masked_bce(null)", mg_masked_bce_f32(NULL, NULL, NULL, 0, 0, 0, 1.f, NULL, NULL, NULL, 0, NULL));
    expect_code("normalise(null)", mg_normalise_f32(NULL, NULL, NULL, NULL, 0, 0, 0, NULL));
    expect_code("normalise(kind)", mg_normalise_f32(df, df, df, df, 100, 8, 99, NULL));
    expect_code("linear_fwd_f32(null)", mg_linear_fwd_f32(NULL, 0, NULL, 0, 0, NULL, NULL, 0, NULL, 0, 0, NULL));
    expect_code("linear_fwd_bf16(lda)", mg_linear_fwd_bf16(dh, 601, NULL, 4096, 600, dh, 640, df, 512, dh, 512, 1, 0, NULL));
    expect_code("linear_dgrad_bf16(null)", mg_linear_dgrad_bf16(NULL, 0, 0, 0, NULL, 0, 0, NULL, 0, NULL, 0, 0, NULL));
    expect_code("linear_wgrad_bf16(small ws)", mg_linear_wgrad_bf16(dh, 512, dh, 640, NULL, 256000, 512, 600, df, df, 0, d, 64, NULL));
    expect_code("linear_bwd_fused(null)", mg_linear_bwd_fused_bf16(NULL, 0, 0, NULL, 0, NULL, 0, NULL, 0, NULL, 0, 0, 0, NULL, NULL, 0, NULL, 0, NULL));
    expect_code("cast_pad_bf16(ld)", mg_cast_pad_bf16(df, 600, dh, 100, 64, 600, NULL));
    expect_code("f0_tail(null)", mg_f0_tail_bf16(NULL, 0, 0, NULL, NULL, NULL, NULL, NULL, NULL, 0, 0, 1.f, NULL, NULL, NULL, NULL, 0, NULL, 0, NULL));
    expect_code("f0_l2tail(null)", mg_f0_l2tail_bf16(NULL, 0, 0, NULL, 0, 0, NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0, 0, 1.f, NULL, NULL, NULL, 0, NULL,
                                                     0, NULL, 0, NULL));
    expect_code("f0_l2tail(shape)", mg_f0_l2tail_bf16(dh, 512, 256, dh, 512, 128, df, df, df, df, df, df, dl, 64, 1000, 1.f, df, df, dh, 128, df, 0, d,
                                                      1 << 20, NULL));
    expect_code("f0_l2tail(small ws)", mg_f0_l2tail_bf16(dh, 512, 512, dh, 512, 128, df, df, df, df, df, df, dl, 64, 1000, 1.f, df, df, dh, 128, df, 0, d,
                                                         64, NULL));
    { int ns = 0; expect_code("f0_l2tail_rows_slabs(null)", mg_f0_l2tail_rows_slabs_bf16(dh, 512, 512, dh, 512, 128, df, df, df, df, df, df, NULL, 21504,
                                                                                         1.f, df, dh, 128, d, 1 << 20, &ns, NULL)); }
    expect_code("expand_column_reduce(null)", mg_expand_column_reduce_f32(NULL, NULL, 0, NULL, NULL, 0, 0, NULL, 0, 0, 0, NULL, NULL));
    { float v[4] = {0.f, 0.f, 0.f, 0.f}; expect_code("store_pairs(n)", mg_store_pairs_f32(df, v, 99, NULL)); }
    { int ns = 0; long long st = 0; expect_code("linear_bwd_fused_slabs(null)", mg_linear_bwd_fused_slabs_bf16(NULL, 0, 0, NULL, 0, NULL, 0, NULL, 0, NULL, 0,
                                                                                                         0, 0, NULL, 0, &ns, (int64_t*)&st, NULL)); }
    { int ns = 0; long long st = 0; expect_code("linear_wgrad_dgrad(null)", mg_linear_wgrad_dgrad_bf16(NULL, 0, NULL, 0, 0, 0, 0, NULL, 0, NULL, 0, NULL, 0,
                                                                                               &ns, (int64_t*)&st, NULL)); }
    expect_code("slab_reduce(null)", mg_slab_reduce_f32(NULL, 0, 0, 0, NULL, 0, NULL));
    expect_code("phone_front(null)", mg_phone_front(NULL, 0, 0, 0, NULL, NULL, 0, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, 0, NULL));
    expect_code("phone_front_linear_fwd(null)", mg_phone_front_linear_fwd_bf16(NULL, 0, 0, 0, NULL, NULL, 0, NULL, NULL, 0, NULL, NULL, NULL, NULL, NULL, 0,
                                                                               NULL, 0, 0, 0, NULL, 0, NULL, 0, NULL, 0, 0, NULL));
    expect_code("linear_fwd_bf16(runs hint, bad act)", mg_linear_fwd_bf16(dh, 640, di, 4096, 600, dh, 640, df, 512, dh, 512, 0, MG_ACT_ROWS_RUNS | 7, NULL));
    expect_code("gru_fwd_f32(null)", mg_gru_fwd_f32(NULL, NULL, NULL, NULL, 0, 0, 0, NULL, NULL, NULL, NULL));
    expect_code("gru_fwd_bf16(H)", mg_gru_fwd_bf16(df, dh, 96, df, dl, 4, 10, 96, df, dh, df, df, NULL));
    expect_code("gru_fwd_persist(shape)", mg_gru_fwd_persist_bf16(df, dh, 512, df, dl, 4096, 10, 512, df, dh, df, df, d, 1 << 20, NULL));
    expect_code("gru_bwd_persist(null)", mg_gru_bwd_persist_bf16(NULL, NULL, NULL, NULL, NULL, 0, NULL, 0, 0, 0, NULL, NULL, NULL, NULL, NULL, NULL, 0, NULL));
    expect_code("lstm_fwd_f32(null)", mg_lstm_fwd_f32(NULL, NULL, NULL, NULL, 0, 0, 0, NULL, NULL, NULL, NULL, NULL));
    expect_code("adam(null)", mg_adam_step_f32(NULL, NULL, NULL, NULL, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0, 1.f, NULL));
    { float two[2]; mg_adam_scalars(0.01f, 0.9f, 0.999f, 1, two); if (!(two[0] > 0.f && two[1] > 0.f)) { fprintf(stderr, "mg_adam_scalars\n"); ++n_bad; } }
    expect_code("ema(null)", mg_ema_update_f32(NULL, NULL, 0, 0.5f, NULL));
    /* round 4 entry points: operand splits of precision 'bf16x3', dropout, the calibration loop, the GRU recurrence with a bf16 output copy */
    {
        mg_split3_desc sd[2];
        memset(sd, 0, sizeof(sd));
        expect_code("split3(count)", mg_split3_bf16(sd, 0, NULL));
        expect_code("split3(count big)", mg_split3_bf16(sd, MG_SPLIT3_MAX + 1, NULL));
        expect_code("split3(null desc)", mg_split3_bf16(sd, 2, NULL));
        sd[0].src = df; sd[0].rows = 21504; sd[0].cols = 600; sd[0].lds = 600; sd[0].dst = dh; sd[0].ldp = 640; sd[0].order = 0;
        sd[1] = sd[0]; sd[1].order = 2; sd[1].plane_rows = 22528;
        expect_code("split3(C2 table)", mg_split3_bf16(sd, 2, NULL));
        sd[1].order = 1; sd[1].transpose = 1; sd[1].plane_rows = 0;                       /* 21504 rows do not fit planes of 640 columns */
        expect_code("split3(transposed too tall)", mg_split3_bf16(sd, 2, NULL));
        sd[1].transpose = 0; sd[1].sig = df; sd[1].ldsig = 8;                             /* ldsig < cols */
        expect_code("split3(bad sig)", mg_split3_bf16(sd, 2, NULL));
    }
    expect_code("dropout(p)", mg_dropout(df, df, 100, 0, 1.5f, 1, 0, NULL, NULL));
    expect_code("dropout(ok)", mg_dropout(df, df, 1 << 20, 1, 0.25f, 0x1234567890abcdefull, 3, (const uint64_t*)d, NULL));
    expect_code("dropout_advance(null)", mg_dropout_advance(NULL, NULL, NULL));
    { uint32_t c[4] = {0, 0, 0, 0}, k[2] = {0, 0}, o[4]; mg_philox4x32_10(c, k, o); if (o[0] != 0x6627e8d5u) { fprintf(stderr, "mg_philox4x32_10\n"); ++n_bad; } }
    { double fl = 0.0; expect_code("calib_mfma(null)", mg_calib_mfma_bf16(NULL, NULL, 256, 10, &fl, NULL));
      expect_code("calib_mfma(ok)", mg_calib_mfma_bf16(dh, df, 256, 100, &fl, NULL)); }
    expect_code("phone_concat_layer(C)", mg_phone_concat_layer_bf16(df, 512, di, 64000, df, 17, df, 609, 600, df, 512, MG_ACT_SIGMOID, dh, 512, 0, NULL));
    expect_code("phone_concat_layer(ok)", mg_phone_concat_layer_bf16(df, 512, di, 64000, df, 9, df, 609, 600, df, 512, MG_ACT_SIGMOID, dh, 512, 0, NULL));
    expect_code("f0_tail_rows_f32(ldz)", mg_f0_tail_rows_f32(df, 130, df, df, df, df, df, df, NULL, 0, 0, 22528, df, df, 128, df, d, 1 << 24, NULL));
    expect_code("f0_tail_rows_f32(ok)", mg_f0_tail_rows_f32(df, 128, df, df, df, df, df, NULL, dl, 64, 352, 22528, df, df, 128, df, d, 1 << 24, NULL));
    expect_code("phone_mse_rows(ok)", mg_phone_mse_rows_f32(df, 1, df, df, 22528, df, df, NULL));
    expect_code("segment_sum_feat(small slabs)", mg_segment_sum_feat_bf16(dh, 512, di, 64000, di, di, 5000, 1024, 512, dh, 512, df, 9, d, 16, NULL));
    expect_code("feat_wgrad_reduce(ok)", mg_feat_wgrad_reduce(d, 9, 512, 512, df, 609, 600, 1, NULL));
    expect_code("gru_fwd_persist_out(shape)", mg_gru_fwd_persist_out_bf16(df, NULL, 0, dh, 512, df, dl, 64, 1000, 512, df, dh, df, dh, df, d, 16, NULL));

    /* plausible shapes: the host path runs its planning and set-up; without a device the launch itself fails (MG_ELAUNCH) */
    expect_code("upsample_index(C2)", mg_upsample_index(dl, 256, 80, 1000, NULL, di, NULL));
    expect_code("frame_layout(C5)", mg_frame_layout(dl, 64, 2000, 73600, di, di, di, NULL));
    expect_code("linear_fwd_bf16(C2 L1)", mg_linear_fwd_bf16(dh, 640, di, 256000, 600, dh, 640, df, 512, dh, 512, 1, 0, NULL));
    {
        const size_t ws = mg_linear_wgrad_workspace_bytes(256000, 512, 600);
        expect_code("linear_wgrad_bf16(C2 L1)", mg_linear_wgrad_bf16(dh, 512, dh, 640, di, 256000, 512, 600, df, df, 0, d, ws, NULL));
        const size_t ws2 = mg_linear_wgrad_workspace_bytes(21504, 512, 600);
        expect_code("linear_wgrad_bf16(phone rate)", mg_linear_wgrad_bf16(dh, 512, dh, 640, NULL, 21504, 512, 600, df, df, 1, d, ws2, NULL));
    }
    expect_code("linear_fwd_bf16(C2 L1, runs)", mg_linear_fwd_bf16(dh, 640, di, 256000, 600, dh, 640, df, 512, dh, 512, 0, 1 | MG_ACT_ROWS_RUNS, NULL));
    expect_code("f0_l2tail(C2)", mg_f0_l2tail_bf16(dh, 512, 512, dh, 512, 128, df, df, df, df, df, df, dl, 256, 1000, 1.f, df, df, dh, 128, df, 0, d,
                                                   mg_f0_l2tail_workspace_bytes(256000), NULL));
    expect_code("masked_mse(C5)", mg_masked_mse_f32(df, df, dl, 64, 2000, 187, 1.f, df, df, d, mg_masked_mse_workspace_bytes(64, 2000, 187), NULL));
    expect_code("gru_fwd_f32(C4)", mg_gru_fwd_f32(df, df, df, dl, 64, 3, 512, df, df, df, NULL));

    /* mg_host_pack is pure host code on real memory: ragged pieces (one empty) into an exactly-sized destination, every thread count -
     * an overrun or a torn range is ASan's to see; a destination that is one byte short must be refused */
    {
        enum { PIECES = 37 };
        const void* srcs[PIECES];
        int64_t bytes[PIECES], total = 0;
        for (int i = 0; i < PIECES; ++i) {
            bytes[i] = i == 5 ? 0 : (int64_t)(1 + (i * 7919) % 300000);
            unsigned char* p = (unsigned char*)malloc((size_t)bytes[i] + 1);
            for (int64_t k = 0; k < bytes[i]; ++k) p[k] = (unsigned char)(i + k);
            srcs[i] = p;
            total += bytes[i];
        }
        unsigned char* dst = (unsigned char*)malloc((size_t)total);
        const int threads[] = {1, 2, 3, 8, 64};
        for (int t = 0; t < 5; ++t) {
            memset(dst, 0xee, (size_t)total);
            expect_code("host_pack", mg_host_pack(srcs, bytes, PIECES, dst, total, threads[t]));
            int64_t at = 0;
            for (int i = 0; i < PIECES; ++i) {
                if (memcmp(dst + at, srcs[i], (size_t)bytes[i]) != 0) {
                    fprintf(stderr, "host_pack: piece %d differs with %d threads\n", i, threads[t]);
                    ++n_bad;
                }
                at += bytes[i];
            }
        }
        expect_code("host_pack(short)", mg_host_pack(srcs, bytes, PIECES, dst, total - 1, 4));
        expect_code("host_pack(threads)", mg_host_pack(srcs, bytes, PIECES, dst, total, 0));
        expect_code("host_pack(null)", mg_host_pack(NULL, bytes, PIECES, dst, total, 1));
        for (int i = 0; i < PIECES; ++i) free((void*)srcs[i]);
        free(dst);
    }

    printf("%d calls, %d unexpected results\n", n_calls, n_bad);
    return n_bad ? 1 : 0;
}
