// Fused backward of a  Linear(K -> N1) + Sigmoid -> Linear(N1 -> N2)  pair, bf16 mode: the weight gradient of the FIRST
// layer straight from the pre-activation gradient of the SECOND one,
//
//     dZ1 = (dZ2 W2) * H1 (1 - H1)          (never written to HBM)
//     dW1 = dZ1^T gather(X),   db1 = column sums of dZ1
//
// Reference: autograd of the README stack (README.rst:65-73 via morgana/utils.py:401-418): mm, sigmoid_backward, mm.
// The unfused path ran a dgrad GEMM that wrote dZ1 (262 MB at C2) and a wgrad GEMM that read it back; both were bound
// by that traffic (profiles/r1: 174 us + 193 us).  Here a workgroup owns 128 hidden units (columns of dZ1) and a slice
// of the frames; per 32-frame step
//   P1 (waves 0-3)  dZ1s^T[128 j, 32 m] = W2T_s[128 j, N2] . dZ2[32 m, N2]^T  on the matrix pipe (weights as A operand, so
//                   the lane holds a frame), times H1 (1 - H1) from an LDS tile, written as bf16 to the LDS tile the
//                   transposed reads of P2 expect;
//   P2 (all waves)  the 128 x 640 wgrad step of wgrad_big_kernel (gemm_bf16_big.hip), db1 through the ones fragment.
// X is double buffered (LDS-DMA two steps ahead of use), the small dZ2 / H1 tiles are refilled while P2 runs.
// Deterministic: split-M slabs + ordered reduce (shared with the unfused wgrad).
#include "common.h"
#include "slab_reduce.h"

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfv4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

MG_STAMP_DECL(g_stamps_fz);
#define FZ_ELEMS 16384
__device__ uint16_t g_fused_zero_row[FZ_ELEMS];

__device__ __forceinline__ void fglds16(const uint16_t* src, unsigned char* lds_wave_base) {
    const unsigned lds_off = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)lds_wave_base);
    const unsigned lds_uni = __builtin_amdgcn_readfirstlane(lds_off);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_uni)
                 : "memory");
}

#define F_BNT 128
#define F_BKT 640
#define F_N2 128
#define F_ROWS_MAX 4096
#define F_W2T 0                                   // 4 x [128 rows x 64 B]   (swizzled: chunk ^ ((row >> 2) & 3))
#define F_X (F_W2T + 32768)                       // 2 stages x [32 m][1280 B] (tr-swizzled)
#define F_DZ (F_X + 2 * 40960)                    // 4 x [32 rows x 64 B]     (swizzled like W2T)
#define F_H1 (F_DZ + 8192)                        // [32 m][256 B] plain rows
#define F_YS (F_H1 + 8192)                        // [32 m][256 B] tr-swizzled: the dZ1 tile
#define F_ROWS (F_YS + 8192)
#define F_LDS (F_ROWS + F_ROWS_MAX * 4)

__global__ __launch_bounds__(512) void wgrad_fused_kernel(const uint16_t* __restrict__ dZ2, int lddz, const uint16_t* __restrict__ W2T,
                                                          int ldwt, const uint16_t* __restrict__ H1, int ldh,
                                                          const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                          int64_t M, int N, int K, int m_chunk, float* __restrict__ slab,
                                                          float* __restrict__ bslab, int variant) {
    constexpr int TKT = 5;
    constexpr int PY = 256, PX = F_BKT * 2;
    constexpr int NX = 5;
    __shared__ __attribute__((aligned(16))) unsigned char smem[F_LDS];
    int* row_lds = reinterpret_cast<int*>(smem + F_ROWS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MG_STAMPS
    unsigned long long ts0, ts1 = 0, ts2, ts3, tr0, tr1, ta, tb, sum_wait_a = 0, sum_issue_x = 0, sum_p1 = 0, sum_wait_b = 0,
                       sum_issue_s = 0, sum_p2 = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    const int wn0 = (wave >> 2) * 64;
    const int wk0 = (wave & 3) * (TKT * 32);
    const int tiles_n = N / F_BNT;
    const int n0 = (blockIdx.x % tiles_n) * F_BNT;      // n tile fastest (XCD-grouped order and non-temporal streaming of
    const int s = blockIdx.x / tiles_n;                 // dZ2 / H1 were measured: both within noise, 351-364 us)
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t m_hi = min(M, m_lo + (int64_t)m_chunk);
    const int n_rows = m_hi > m_lo ? (int)(m_hi - m_lo) : 0;

    for (int i = tid; i < m_chunk; i += 512) {
        int r = -1;
        if (i < n_rows) r = rows ? rows[m_lo + i] : (int)(m_lo + i);
        row_lds[i] = r;
    }
    __syncthreads();

    // ---- one-time: this workgroup's 128 rows of W2^T, as 4 k-tiles of [128 x 64 B] ------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = wave * 4 + i;                       // piece 0..31
        const int kt = p >> 3;
        const int row = 16 * (p & 7) + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        fglds16(W2T + (size_t)(n0 + row) * ldwt + 32 * kt + 8 * c, smem + F_W2T + p * 1024);
    }

    // ---- per-step DMA slots ---------------------------------------------------------------------------------------------
    int x_row[NX], x_off[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int byte = (wave * NX + i) * 1024 + lane * 16;
        x_row[i] = byte / PX;
        const int cpos = (byte % PX) >> 4;
        x_off[i] = (cpos ^ ((x_row[i] & 3) << 2)) * 8;
    }
    auto issue_x = [&](int step) {
        unsigned char* st = smem + F_X + (step & 1) * 40960;
        int rr[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) rr[i] = row_lds[step * 32 + x_row[i]];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const uint16_t* p = (rr[i] >= 0) ? A + (size_t)rr[i] * lda + x_off[i] : g_fused_zero_row;
            fglds16(p, st + (wave * NX + i) * 1024);
        }
    };
    // dZ2 piece `wave`: k-tile kt = wave >> 1, rows 16 (wave & 1) ..; H1 piece `wave`: rows 4 wave .. 4 wave + 3
    const int dz_row = 16 * (wave & 1) + (lane >> 2);
    const int dz_col = 32 * (wave >> 1) + 8 * ((lane & 3) ^ ((dz_row >> 2) & 3));
    const int h1_row = 4 * wave + (lane >> 4);
    const int h1_col = n0 + 8 * (lane & 15);
    auto issue_small = [&](int step) {
        const int mz = step * 32 + dz_row, mh = step * 32 + h1_row;
        fglds16(mz < n_rows ? dZ2 + (size_t)(m_lo + mz) * lddz + dz_col : g_fused_zero_row, smem + F_DZ + wave * 1024);
        fglds16(mh < n_rows ? H1 + (size_t)(m_lo + mh) * ldh + h1_col : g_fused_zero_row, smem + F_H1 + wave * 1024);
    };

    f32x16 acc[2][TKT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TKT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool bias_free = bslab != nullptr && (wave & 3) == 3;        // K <= 608: last 32-column tile of k-wave 3 is padding
    const __bf16 one_bf = (__bf16)1.0f;
    const bfv8 ones = bfv8{one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf};

    const int n_steps = (n_rows + 31) / 32;
    const int li = lane & 15, g4 = lane >> 4;
    const int q = li >> 2, p4 = li & 3;
    const int cgrp = 16 * (g4 & 1) + 4 * p4;
    const int rbase = 8 * (g4 >> 1) + q;
    const int sw = q << 2;
    int yoff[2], xoff[TKT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = wn0 + i * 32 + cgrp;
        yoff[i] = F_YS + rbase * PY + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
    }
#pragma unroll
    for (int j = 0; j < TKT; ++j) {
        const int col = wk0 + j * 32 + cgrp;
        xoff[j] = rbase * PX + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
    }
    // P1 geometry (waves 0-3): hidden-unit tile = wave, lane = frame
    const int mi = lane & 31, lh = lane >> 5;
    const int a_row = 32 * (wave & 3) + mi;               // W2T row (hidden unit) this lane feeds as A operand
    const int a_base = F_W2T + a_row * 64;
    const int a_swz = (a_row >> 2) & 3;
    const int b_base = F_DZ + mi * 64;
    const int b_swz = (mi >> 2) & 3;

    if (n_steps > 0) {
        issue_x(0);
        issue_small(0);
    }
    for (int step = 0; step < n_steps; ++step) {
        MG_STAMP(ta);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");          // A: X(step), dZ2/H1(step), W2T landed
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_wait_a, tb, ta);
#ifdef MG_STAMPS
        if (step == 0) ts1 = tb;
#endif
        if (step + 1 < n_steps) issue_x(step + 1);
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_issue_x, ta, tb);
        if (wave < 4 && variant != 2) {
            // ---- P1: dZ1s^T tile = W2T_s . dZ2^T, 8 k-steps over the N2 = 128 outputs of layer 2 -------------------------
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int kt = ks >> 1, ch = 2 * (ks & 1) + lh;
                const bfv8 a = *reinterpret_cast<const bfv8*>(smem + a_base + kt * 8192 + ((ch ^ a_swz) << 4));
                const bfv8 b = *reinterpret_cast<const bfv8*>(smem + b_base + kt * 2048 + ((ch ^ b_swz) << 4));
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d, 0, 0, 0);
            }
            // register 4 g + e <-> hidden unit 32 wave + 8 g + 4 lh + e of frame mi; times H1 (1 - H1); to the Ys tile
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 4 * wave + g;                                   // 16-byte chunk of the 256-byte row
                const bfv4 hv = *reinterpret_cast<const bfv4*>(smem + F_H1 + mi * 256 + c * 16 + 8 * lh);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = (float)hv[e];
                    v[e] = d[4 * g + e] * h * (1.f - h);
                }
                const u32x2 pk = u32x2{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                       __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
                *reinterpret_cast<u32x2*>(smem + F_YS + mi * 256 + ((c ^ ((mi & 3) << 2)) << 4) + 8 * lh) = pk;
            }
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_p1, tb, ta);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");         // B: the dZ1 tile is complete
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_wait_b, ta, tb);
        if (step + 1 < n_steps) issue_small(step + 1);                           // dZ2 / H1 tiles are free again
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_issue_s, tb, ta);
        // ---- P2: dW1s += dZ1s^T . X ------------------------------------------------------------------------------------
        const unsigned char* xs = smem + F_X + (step & 1) * 40960;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bfv8 a[2], b[TKT];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const unsigned char* ad = smem + yoff[i] + ks * 16 * PY;
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PY));
                a[i] = bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TKT; ++j) {
                const unsigned char* ad = xs + xoff[j] + ks * 16 * PX;
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PX));
                b[j] = bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            if (bias_free) b[TKT - 1] = ones;
            if (variant == 3) continue;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TKT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_p2, ta, tb);
    }

#ifdef MG_STAMPS
    MG_STAMP(ts2);
#endif
    const int lr = lane & 31;
    float* out = slab + (size_t)s * N * K;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < TKT; ++j) {
            const int col = wk0 + j * 32 + lr;
            if (col >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wn0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < N) out[(size_t)row * K + col] = acc[i][j][r];
            }
        }
    }
    if (bias_free && lr == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wn0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < N) bslab[(size_t)s * N + row] = acc[i][TKT - 1][r];
            }
    }
#ifdef MG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG_STAMP(ts3);
    MG_STAMP_REAL(tr1);
    const int sb = blockIdx.x;
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 2, ts2);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 3, ts3);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 4, tr0);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 5, tr1);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 6, sum_wait_a);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 7, sum_issue_x);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 8, sum_p1);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 9, sum_wait_b);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 10, sum_issue_s);
    MG_STAMP_STORE(g_stamps_fz, sb, wave, lane, 11, sum_p2);
#endif
}


static void fused_plan(int64_t M, int N, int* S, int* m_chunk) {
    const int tiles_n = N / F_BNT;
    int64_t s = mg_ceil_div(256, tiles_n);
    int64_t chunk = mg_align_up((size_t)mg_ceil_div(M, s), 32);
    while (chunk > F_ROWS_MAX) {
        s *= 2;
        chunk = mg_align_up((size_t)mg_ceil_div(M, s), 32);
    }
    *S = (int)mg_ceil_div(M, chunk);
    *m_chunk = (int)chunk;
}

extern "C" {

size_t mg_linear_bwd_fused_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || N % F_BNT != 0) return 256;
    int S, chunk;
    fused_plan(M, N, &S, &chunk);
    return mg_align_up((size_t)S * ((size_t)N * K + N) * sizeof(float), 256);
}

int mg_linear_bwd_fused_bf16(const uint16_t* dZ2, int lddz, int N2, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh,
                             const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K, float* dW, float* db,
                             int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(dZ2 && W2T && H1 && A && dW && db && M > 0, "mg_linear_bwd_fused_bf16: null argument or empty batch");
    MG_CHECK_ARG(N2 == F_N2 && lddz >= F_N2 && ldwt >= F_N2 && lddz % 8 == 0 && ldwt % 8 == 0,
                 "mg_linear_bwd_fused_bf16: the second layer must have %d outputs (N2=%d lddz=%d ldwt=%d)", F_N2, N2, lddz, ldwt);
    MG_CHECK_ARG(N % F_BNT == 0 && ldh >= N && ldh % 8 == 0, "mg_linear_bwd_fused_bf16: hidden width %d must be a multiple of %d (ldh=%d)", N, F_BNT, ldh);
    MG_CHECK_ARG(lda == F_BKT && K > 512 && K <= F_BKT - 32, "mg_linear_bwd_fused_bf16: needs 512 < K <= 608 with lda = 640 (K=%d lda=%d)", K, lda);
    MG_CHECK_ARG((((uintptr_t)dZ2 | (uintptr_t)W2T | (uintptr_t)H1 | (uintptr_t)A) % 16) == 0, "mg_linear_bwd_fused_bf16: buffers must be 16-byte aligned");
    if (!workspace || workspace_bytes < mg_linear_bwd_fused_workspace_bytes(M, N, K)) {
        mg_set_error("mg_linear_bwd_fused_bf16: workspace of %zu bytes needed, got %zu", mg_linear_bwd_fused_workspace_bytes(M, N, K), workspace_bytes);
        return MG_EWORKSPACE;
    }
    int S, chunk;
    fused_plan(M, N, &S, &chunk);
    float* slab = (float*)workspace;
    float* bslab = slab + (size_t)S * N * K;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(wgrad_fused_kernel, dim3((unsigned)((N / F_BNT) * S)), dim3(512), 0, st, dZ2, lddz, W2T, ldwt, H1, ldh, A, lda, rows, M,
                       N, K, chunk, slab, bslab, g_mg_tuning[MG_TUNE_STAGGER]);
    MG_CHECK_LAUNCH("mg_linear_bwd_fused_bf16/main");
    const int64_t nk = (int64_t)N * K;
    mg_launch_slab_reduce(slab, nk, nk, S, dW, accumulate, st);
    mg_launch_slab_reduce(bslab, N, N, S, db, accumulate, st);
    MG_CHECK_LAUNCH("mg_linear_bwd_fused_bf16/reduce");
    return MG_OK;
}

}  // extern "C"

#ifdef MG_STAMPS
extern "C" int mg_diag_read_stamps_fz(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_fz), bytes < sizeof(g_stamps_fz) ? bytes : sizeof(g_stamps_fz), 0, hipMemcpyDeviceToHost);
}
#endif
