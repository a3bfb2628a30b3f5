// K2 (bf16 throughput mode), large-tile kernels for the shapes that dominate the step:
//   gemm_nt_big   C[M,N] = epi(A'[M,K] B[N,K]^T)    256 x {256,128} tiles, BK = 64      (forward, dgrad)
//   wgrad_big     dW[N,K] = dY[M,N]^T A'[M,K]        128 x {640,512} tiles over split M    (weight gradients)
// Reference: the nn.Linear + nn.Sigmoid stack of README.rst:65-73 (morgana/utils.py:401-418) and its backward.
//
// Why these shapes (measured on MI355X with the 128 x 128 kernels of gemm_bf16.hip, profiles/r1_*):
//   * a 128 x 128 x 32 tile moves 16 KB through L2->LDS per 1 MFLOP; at the ~30 B/clk a CU sustains from L2 that caps
//     the MFMA pipe below 50 %.  256 x 256 x 64 moves 64 KB per 8.4 MFLOP (2x the FLOP per byte).
//   * the 128 x 128 wgrad re-read dZ1 (262 MB) once per 128-wide K tile: FETCH_SIZE 0.71 GB (x2 on gfx950) per launch.
//     A 128 x 640 tile covers all of K, so dY is read exactly once; X (the gathered phone rows) comes from L2/MALL.
// Mechanics (cdna_hip_programming.md section 5): tiles are filled by global_load_lds_dwordx4 (LDS-DMA: no VGPR staging,
// no ds_write), two LDS stages, the next tile's DMA stays in flight across a raw s_barrier behind a COUNTED
// s_waitcnt vmcnt(N); the LDS image is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address and
// again on the fragment reads (the same involution on both sides).  The per-lane source address is also what fuses
// the upsample gather: a lane simply points at its phone row (or at a zero row for the -1 pad index).
// 512 threads = 8 waves, one workgroup per CU (2 waves per SIMD).
#include "common.h"

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfv4 __attribute__((ext_vector_type(4)));

#define EPI_BIAS 0
#define EPI_BIAS_SIGMOID 1
#define EPI_SIGMOID_GRAD 2

#define MG_ZERO_ELEMS 16384
__device__ uint16_t g_zero_row[MG_ZERO_ELEMS];   // zero-initialised: source of pad rows / out-of-range rows

__device__ __forceinline__ void glds16(const uint16_t* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }

#define WAIT_VM_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")
#define WAIT_LGKM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---------------------------------------------------------------------------------------------------------------------
// gemm_nt_big: BM = 256, BN in {256, 128}, BK = 64.  LDS rows are 128 bytes (8 chunks of 16 B); chunk c of tile row r is
// stored at chunk position c ^ ((r >> 1) & 7): conflict free for the 4 x 16 lane groups of ds_read_b128.
// Requires lda, ldb multiples of 64 with zero padding, N a multiple of BN... (checked by the launcher).
// ---------------------------------------------------------------------------------------------------------------------
template <int BN, int EPI>
__global__ __launch_bounds__(512) void gemm_nt_big_kernel(const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                          int64_t M, int K, const uint16_t* __restrict__ Bm, int ldb, int N,
                                                          const float* __restrict__ bias, const uint16_t* __restrict__ H, int ldh,
                                                          void* __restrict__ Cv, int ldc, int tiles_n, int c_f32) {
    constexpr int BM = 256;
    constexpr int WAVES_N = BN / 64;              // 4 or 2
    constexpr int WAVES_M = 8 / WAVES_N;          // 2 or 4
    constexpr int WM = BM / WAVES_M;              // 128 or 64
    constexpr int TM = WM / 32, TN = 2;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int GA = BM / 64;                   // 1 KB row groups (8 rows) per wave for A: 4
    constexpr int GB = BN / 64;                   // for B: 4 or 2
    constexpr int NL = GA + GB;                   // LDS-DMA instructions per wave per K tile
    constexpr int STG_LD = 68;                    // fp32 staging pitch (floats): 32 rows x 64 cols per wave and pass
    constexpr int STG_BYTES = 8 * 32 * STG_LD * 4;
    constexpr int LDS_BYTES = 2 * STAGE > STG_BYTES ? 2 * STAGE : STG_BYTES;

    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * 64;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM;
    const int n0 = (blockIdx.x % tiles_n) * BN;
    const int kc = lda < ldb ? lda : ldb;
    const int n_kt = kc / 64;

    // Per-lane DMA sources.  Group g covers tile rows 8g..8g+7; lane l writes LDS row 8g + (l>>3), chunk position l&7,
    // and therefore fetches source chunk (l&7) ^ ((row>>1)&7) of that row.
    const uint16_t* asrc[GA];
    const uint16_t* bsrc[GB];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int row = (wave * GA + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int64_t m = m0 + row;
        const uint16_t* p = g_zero_row;
        if (m < M) {
            if (rows) {
                const int r = rows[m];
                if (r >= 0) p = A + (size_t)r * lda;
            } else {
                p = A + (size_t)m * lda;
            }
        }
        asrc[i] = p + c * 8;
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
        const int row = (wave * GB + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        const int n = n0 + row;
        bsrc[i] = (n < N ? Bm + (size_t)n * ldb : g_zero_row) + c * 8;
    }

    auto issue = [&](int kt) {
        unsigned char* st = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int i = 0; i < GA; ++i) glds16(asrc[i] + kt * 64, st + (wave * GA + i) * 1024);
#pragma unroll
        for (int i = 0; i < GB; ++i) glds16(bsrc[i] + kt * 64, st + A_BYTES + (wave * GB + i) * 1024);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    const int ks_last = ((K - (n_kt - 1) * 64) + 15) / 16;       // k-steps of the last tile that hold real columns

    issue(0);
    for (int kt = 0; kt < n_kt; ++kt) {
        if (kt + 1 < n_kt) {
            issue(kt + 1);
            if (NL == 8) WAIT_VM_BARRIER(8); else WAIT_VM_BARRIER(6);
        } else {
            WAIT_VM_BARRIER(0);
        }
        const unsigned char* As = smem + (kt & 1) * STAGE;
        const unsigned char* Bs = As + A_BYTES;
        const int n_ks = (kt + 1 < n_kt) ? 4 : (ks_last < 4 ? ks_last : 4);
        for (int ks = 0; ks < n_ks; ++ks) {
            const int c = ks * 2 + lh;
            bfv8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm0 + i * 32 + lr;
                a[i] = *reinterpret_cast<const bfv8*>(As + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 32 + lr;
                b[j] = *reinterpret_cast<const bfv8*>(Bs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        WAIT_LGKM_BARRIER();     // every wave has its fragments in registers: the stage may be refilled
    }

    // Epilogue: per pass one 32 x 64 fp32 sub-tile of this wave goes through a private LDS slab so that the global
    // stores are whole 16-byte (bf16) or 32-byte (fp32) row pieces, 8 lanes per 128-byte row.
    float* stg = reinterpret_cast<float*>(smem) + wave * 32 * STG_LD;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * STG_LD + j * 32 + lr] = acc[i][j][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rl = it * 8 + (lane >> 3);
            const int cl = (lane & 7) * 8;
            const int64_t row = m0 + wm0 + i * 32 + rl;
            const int col = n0 + wn0 + cl;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&stg[rl * STG_LD + cl]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&stg[rl * STG_LD + cl + 4]);
            if (row >= M || col >= ldc) continue;
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            bfv8 hv;
            if (EPI == EPI_SIGMOID_GRAD) hv = *reinterpret_cast<const bfv8*>(H + (size_t)row * ldh + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = v[e];
                if (col + e >= N) x = 0.f;
                else if (EPI == EPI_BIAS) x += bias ? bias[col + e] : 0.f;
                else if (EPI == EPI_BIAS_SIGMOID) x = fast_sigmoid(x + (bias ? bias[col + e] : 0.f));
                else {
                    const float h = (float)hv[e];
                    x = x * h * (1.f - h);
                }
                v[e] = x;
            }
            if (c_f32) {
                float* crow = reinterpret_cast<float*>(Cv) + (size_t)row * ldc + col;
                *reinterpret_cast<f32x4*>(crow) = f32x4{v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>(crow + 4) = f32x4{v[4], v[5], v[6], v[7]};
            } else {
                bfv8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                *reinterpret_cast<bfv8*>(reinterpret_cast<uint16_t*>(Cv) + (size_t)row * ldc + col) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad_big: output tile 128 (n) x BKT (k), BKT = 64 * TKW (TKW = 10 -> 640, 8 -> 512); contraction over m in steps of 32.
// LDS tiles are straight row copies ([m][n], [m][k]); fragments come from ds_read_b64_tr_b16.  16-byte chunk c of
// tile row r sits at chunk position c ^ ((r & 3) << 2), i.e. the 64-byte blocks of a row are XORed with r & 3, so the 4
// rows x 64 bytes a half wave touches in one transposed read fall on all 64 banks.
// Row indices of the workgroup's whole m range are parked in LDS first, so the loop issues no VGPR-destination loads.
// ---------------------------------------------------------------------------------------------------------------------
#define WG_ROWS_MAX 8192

template <int TKW>
__global__ __launch_bounds__(512) void wgrad_big_kernel(const uint16_t* __restrict__ dY, int lddy, const uint16_t* __restrict__ A, int lda,
                                                        const int32_t* __restrict__ rows, int64_t M, int N, int K, int m_chunk,
                                                        float* __restrict__ slab, float* __restrict__ bslab) {
    constexpr int BNT = 128, BKT = 64 * TKW;
    constexpr int TKT = BKT / 4 / 32;             // 32-column MFMA tiles per wave along k (4 waves along k): 5 or 4
    constexpr int PY = BNT * 2, PX = BKT * 2;     // LDS row pitches in bytes: 256, 1280 / 1024
    constexpr int Y_BYTES = 32 * PY, X_BYTES = 32 * PX;
    constexpr int STAGE = Y_BYTES + X_BYTES;
    constexpr int NX = X_BYTES / 1024 / 8;        // X LDS-DMA instructions per wave per step: 5 or 4
    constexpr int NLW = 1 + NX;                   // + one for dY
    constexpr int LDS_BYTES = 2 * STAGE + WG_ROWS_MAX * 4;

    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    int* row_lds = reinterpret_cast<int*>(smem + 2 * STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn0 = (wave >> 2) * 64;             // 2 waves along n
    const int wk0 = (wave & 3) * (TKT * 32);      // 4 waves along k
    const int n0 = blockIdx.x * BNT;
    const int s = blockIdx.y;
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t m_hi = min(M, m_lo + (int64_t)m_chunk);
    const int n_rows = (int)(m_hi - m_lo);

    // Park source-row indices (identity when there is no gather; -1 = zero row) for the whole m range.
    for (int i = tid; i < m_chunk; i += 512) {
        int r = -1;
        if (i < n_rows) r = rows ? rows[m_lo + i] : (int)(m_lo + i);
        row_lds[i] = r;
    }
    __syncthreads();

    // DMA slots.  dY: one 1 KB piece (4 rows of 256 B) per wave.  X: NX pieces per wave, piece t = wave*NX + i covers
    // linear tile bytes [1024 t, 1024 t + 1024); a lane's byte decides its tile row and chunk position.
    const int y_row = wave * 4 + (lane >> 4);
    const int y_c = (lane & 15) ^ ((y_row & 3) << 2);
    int x_row[NX], x_off[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int byte = (wave * NX + i) * 1024 + lane * 16;
        x_row[i] = byte / PX;
        const int cpos = (byte % PX) >> 4;
        x_off[i] = (cpos ^ ((x_row[i] & 3) << 2)) * 8;
    }

    auto issue = [&](int step) {                 // rows [32 step, 32 step + 32) of this workgroup's range
        unsigned char* st = smem + (step & 1) * STAGE;
        {
            const int ml = step * 32 + y_row;
            const uint16_t* p = (ml < n_rows) ? dY + (size_t)(m_lo + ml) * lddy + n0 + y_c * 8 : g_zero_row;
            glds16(p, st + wave * 1024);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int r = row_lds[step * 32 + x_row[i]];
            const uint16_t* p = (r >= 0) ? A + (size_t)r * lda + x_off[i] : g_zero_row;
            glds16(p, st + Y_BYTES + (wave * NX + i) * 1024);
        }
    };

    f32x16 acc[2][TKT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TKT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum = 0.f;

    const int n_steps = (n_rows + 31) / 32;
    // transposed-read lane geometry (see tr_frag in gemm_bf16.hip)
    const int li = lane & 15, g = lane >> 4;
    const int q = li >> 2, p4 = li & 3;
    const int cgrp = 16 * (g & 1) + 4 * p4;       // column offset inside a 32-wide operand tile
    const int rbase = 8 * (g >> 1) + q;           // contraction row inside a 16-deep k-step

    if (n_steps > 0) issue(0);
    for (int step = 0; step < n_steps; ++step) {
        if (step + 1 < n_steps) {
            issue(step + 1);
            if (NLW == 6) WAIT_VM_BARRIER(6); else WAIT_VM_BARRIER(5);
        } else {
            WAIT_VM_BARRIER(0);
        }
        const unsigned char* Ys = smem + (step & 1) * STAGE;
        const unsigned char* Xs = Ys + Y_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int mrow = ks * 16 + rbase;
            const int sw = (mrow & 3) << 2;
            bfv8 a[2], b[TKT];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int col = wn0 + i * 32 + cgrp;
                const unsigned char* ad = Ys + mrow * PY + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PY));
                a[i] = bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TKT; ++j) {
                const int col = wk0 + j * 32 + cgrp;
                const unsigned char* ad = Xs + mrow * PX + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PX));
                b[j] = bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TKT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (bslab != nullptr && tid < BNT) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const int cpos = (tid >> 3) ^ ((r & 3) << 2);
                bsum += mg_bf2f(*reinterpret_cast<const uint16_t*>(Ys + r * PY + (cpos << 4) + ((tid & 7) << 1)));
            }
        }
        WAIT_LGKM_BARRIER();
    }

    const int lr = lane & 31, lh = lane >> 5;
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
    if (bslab != nullptr && tid < BNT && n0 + tid < N) bslab[(size_t)s * N + n0 + tid] = bsum;
}

// ---------------------------------------------------------------------------------------------------------------------
// Launch helpers used by the entry points in gemm_bf16.hip.  Each returns 1 if it launched, 0 if the shape does not
// qualify (the caller then uses the generic 128 x 128 kernels), negative on error.
// ---------------------------------------------------------------------------------------------------------------------
static bool big16(const void* p) { return ((uintptr_t)p % 16) == 0; }

int mg_try_nt_big(const uint16_t* A, int lda, const int32_t* rows, int64_t M, int K, const uint16_t* Bm, int ldb, int N,
                  const float* bias, const uint16_t* H, int ldh, void* C, int ldc, int c_f32, int epi, hipStream_t st) {
    if (M < 2048 || lda % 64 != 0 || ldb % 64 != 0 || lda > MG_ZERO_ELEMS - 64 || ldb > MG_ZERO_ELEMS - 64) return 0;
    if (N % 128 != 0 || ldc < N || ldc % 8 != 0 || !big16(A) || !big16(Bm) || !big16(C)) return 0;
    if (epi == EPI_SIGMOID_GRAD && (!H || ldh % 8 != 0 || ldh < N)) return 0;
    const bool wide = (N % 256 == 0);
    const int bn = wide ? 256 : 128;
    const int tiles_n = N / bn;
    const int64_t blocks = mg_ceil_div(M, 256) * tiles_n;
    if (blocks >= 2147483647LL) return 0;
    dim3 grid((unsigned)blocks), block(512);
#define LAUNCH_NT(BN_, EPI_) hipLaunchKernelGGL((gemm_nt_big_kernel<BN_, EPI_>), grid, block, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, H, ldh, C, ldc, tiles_n, c_f32)
    if (wide) {
        if (epi == EPI_BIAS) LAUNCH_NT(256, EPI_BIAS);
        else if (epi == EPI_BIAS_SIGMOID) LAUNCH_NT(256, EPI_BIAS_SIGMOID);
        else LAUNCH_NT(256, EPI_SIGMOID_GRAD);
    } else {
        if (epi == EPI_BIAS) LAUNCH_NT(128, EPI_BIAS);
        else if (epi == EPI_BIAS_SIGMOID) LAUNCH_NT(128, EPI_BIAS_SIGMOID);
        else LAUNCH_NT(128, EPI_SIGMOID_GRAD);
    }
#undef LAUNCH_NT
    return 1;
}

// Plan of the split over M for the wide wgrad: S slabs of m_chunk rows.  Returns 0 if the shape does not qualify.
int mg_wgrad_big_plan(int64_t M, int N, int K, int lda, int lddy, int* S_out, int* m_chunk_out) {
    if (M < 4096 || N % 128 != 0 || lddy < N || lddy % 8 != 0) return 0;
    if (!((lda == 640 && K > 512 && K <= 640) || (lda == 512 && K > 384 && K <= 512))) return 0;
    const int tiles_n = N / 128;
    int64_t S = mg_ceil_div(256, tiles_n);
    int64_t m_chunk = mg_align_up((size_t)mg_ceil_div(M, S), 32);
    while (m_chunk > WG_ROWS_MAX) {
        S *= 2;
        m_chunk = mg_align_up((size_t)mg_ceil_div(M, S), 32);
    }
    S = mg_ceil_div(M, m_chunk);
    if (S > 65535) return 0;
    *S_out = (int)S;
    *m_chunk_out = (int)m_chunk;
    return 1;
}

int mg_launch_wgrad_big(const uint16_t* dY, int lddy, const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K,
                        int S, int m_chunk, float* slab, float* bslab, hipStream_t st) {
    dim3 grid((unsigned)(N / 128), (unsigned)S), block(512);
    if (lda == 640)
        hipLaunchKernelGGL((wgrad_big_kernel<10>), grid, block, 0, st, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab);
    else
        hipLaunchKernelGGL((wgrad_big_kernel<8>), grid, block, 0, st, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab);
    return 1;
}
