// K2 (fp32 parity mode) - Linear forward / dgrad / wgrad on v_mfma_f32_32x32x2_f32.
//
// Reference: the nn.Linear + nn.Sigmoid stack of README.rst:65-73 run by SequentialWithRecurrent.forward
// (morgana/utils.py:401-418) and its autograd backward.  The f32-input MFMA is an exact k-ordered fp32 fma chain
// (64 FLOP/clk/SIMD, 157 TF/s peak), so this path carries the 1e-4 parity claim; the bf16 path (gemm_bf16.hip) is the
// throughput mode.
//
// Operand fragments of v_mfma_f32_32x32x2_f32 are ONE float per lane: lane l holds A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  Tiles are therefore staged
// k-major in LDS ([k][row], +2 padding) so that a fragment read is 32 consecutive floats per half wave
// (conflict free), and no transposed copy of any operand is ever needed:
//   fwd   Y = gather(A) W^T        A-tile [k][m] from A[m][k],  B-tile [k][n] from W[n][k]
//   dgrad dX = dY W                A-tile [n][m] from dY[m][n], B-tile [n][k] = straight copy of W rows
//   wgrad dW = dY^T gather(A)      contraction over m: both tiles are straight row copies ([m][n], [m][k])
// The layer-1 gather (upsample_to_repetitions) is fused into the A-tile loader: the (B*T, 600) frame-rate input
// never exists in HBM.
#include "common.h"
#include "slab_reduce.h"

#define GK 16  // contraction depth of one LDS tile

__device__ __forceinline__ f32x4 load4_guard(const float* p, int valid, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid >= 4 && vec) {
        v = *reinterpret_cast<const f32x4*>(p);
    } else {
        if (valid > 0) v.x = p[0];
        if (valid > 1) v.y = p[1];
        if (valid > 2) v.z = p[2];
        if (valid > 3) v.w = p[3];
    }
    return v;
}

#define EPI_BIAS 0
#define EPI_BIAS_SIGMOID 1
#define EPI_SIGMOID_GRAD 2

// C[M,N] = epi( A'[M,Kc] * B ), A' = rows ? A[rows[m]] : A[m]
//   B_KN == false: Bm is [N, Kc] row-major (C = A Bm^T);  B_KN == true: Bm is [Kc, N] row-major (C = A Bm).
template <int BM, int BN, int WAVES_M, int WAVES_N, bool B_KN, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                       int64_t M, int Kc, const float* __restrict__ Bm, int ldb, int N,
                                                       const float* __restrict__ bias, const float* __restrict__ H, int ldh,
                                                       float* __restrict__ C, int ldc, int tiles_n, int vec_a, int vec_b) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int LDA_S = BM + 2;
    constexpr int LDB_S = B_KN ? BN + 4 : BN + 2;
    constexpr int IT_A = (BM * 4 + 255) / 256;
    constexpr int IT_B = (BN * 4 + 255) / 256;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 MFMA tile");

    __shared__ __attribute__((aligned(16))) float As[GK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[GK * LDB_S];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM;
    const int wn0 = (wave % WAVES_N) * WN;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM;
    const int n0 = (blockIdx.x % tiles_n) * BN;

    // Per-thread A source rows (constant over the K loop).
    const float* a_ptr[IT_A];
    int a_row[IT_A], a_kq[IT_A];
#pragma unroll
    for (int i = 0; i < IT_A; ++i) {
        const int f = tid + 256 * i;
        a_row[i] = f >> 2;
        a_kq[i] = f & 3;
        a_ptr[i] = nullptr;
        const int64_t m = m0 + a_row[i];
        if (f < BM * 4 && m < M) {
            if (rows) {
                const int r = rows[m];
                if (r >= 0) a_ptr[i] = A + (size_t)r * lda;
            } else {
                a_ptr[i] = A + (size_t)m * lda;
            }
        }
    }

    f32x4 ra[IT_A], rb[IT_B];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < IT_A; ++i) {
            const int k = k0 + a_kq[i] * 4;
            ra[i] = a_ptr[i] ? load4_guard(a_ptr[i] + k, Kc - k, vec_a) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < IT_B; ++i) {
            const int f = tid + 256 * i;
            rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (f < BN * 4) {
                if (!B_KN) {
                    const int n = n0 + (f >> 2);
                    const int k = k0 + (f & 3) * 4;
                    if (n < N) rb[i] = load4_guard(Bm + (size_t)n * ldb + k, Kc - k, vec_b);
                } else {
                    const int kr = f / (BN / 4);
                    const int c = (f % (BN / 4)) * 4;
                    if (k0 + kr < Kc) rb[i] = load4_guard(Bm + (size_t)(k0 + kr) * ldb + n0 + c, N - (n0 + c), vec_b);
                }
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < IT_A; ++i) {
            const int f = tid + 256 * i;
            if (f < BM * 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) As[(a_kq[i] * 4 + j) * LDA_S + a_row[i]] = ra[i][j];
            }
        }
#pragma unroll
        for (int i = 0; i < IT_B; ++i) {
            const int f = tid + 256 * i;
            if (f < BN * 4) {
                if (!B_KN) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) Bs[((f & 3) * 4 + j) * LDB_S + (f >> 2)] = rb[i][j];
                } else {
                    const int kr = f / (BN / 4);
                    const int c = (f % (BN / 4)) * 4;
                    *reinterpret_cast<f32x4*>(&Bs[kr * LDB_S + c]) = rb[i];
                }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int n_kt = (Kc + GK - 1) / GK;
    load_tiles(0);
    store_tiles();
    __syncthreads();
    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < n_kt; ++kt) {
        if (kt + 1 < n_kt) load_tiles((kt + 1) * GK);
#pragma unroll
        for (int kk = 0; kk < GK / 2; ++kk) {
            const int k = 2 * kk + lh;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[k * LDA_S + wm0 + i * 32 + lr];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[k * LDB_S + wn0 + j * 32 + lr];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < n_kt) {
            store_tiles();
            __syncthreads();
        }
    }

    // Epilogue.  C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn0 + j * 32 + lr;
            if (col >= N) continue;
            const float bv = (EPI != EPI_SIGMOID_GRAD && bias) ? bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row >= M) continue;
                float v = acc[i][j][r];
                if (EPI == EPI_BIAS) v += bv;
                else if (EPI == EPI_BIAS_SIGMOID) v = mg_sigmoid(v + bv);
                else {
                    const float h = H[(size_t)row * ldh + col];
                    v = v * h * (1.f - h);
                }
                C[(size_t)row * ldc + col] = v;
            }
        }
    }
}

// Partial dW over rows [s*m_chunk, (s+1)*m_chunk): slab[s][n][k] = sum_m dY[m][n] * A'[m][k]; bslab[s][n] = sum_m dY[m][n].
template <int BNT, int BKT, int WAVES_N, int WAVES_K>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const float* __restrict__ dY, int lddy, const float* __restrict__ A, int lda,
                                                        const int32_t* __restrict__ rows, int64_t M, int N, int K, int64_t m_chunk,
                                                        float* __restrict__ slab, float* __restrict__ bslab, int tiles_k,
                                                        int vec_y, int vec_a) {
    constexpr int WN = BNT / WAVES_N, WK = BKT / WAVES_K;
    constexpr int TN = WN / 32, TK = WK / 32;
    constexpr int LDY_S = BNT + 4, LDX_S = BKT + 4;
    constexpr int IT_Y = (GK * BNT / 4 + 255) / 256;
    constexpr int IT_X = (GK * BKT / 4 + 255) / 256;
    static_assert(WAVES_N * WAVES_K == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) float Ys[GK * LDY_S];
    __shared__ __attribute__((aligned(16))) float Xs[GK * LDX_S];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn0 = (wave / WAVES_K) * WN;
    const int wk0 = (wave % WAVES_K) * WK;
    const int n0 = (blockIdx.x / tiles_k) * BNT;
    const int k0 = (blockIdx.x % tiles_k) * BKT;
    const int s = blockIdx.y;
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t m_hi = min(M, m_lo + m_chunk);
    const bool do_bias = (blockIdx.x % tiles_k) == 0 && bslab != nullptr;

    f32x4 ry[IT_Y], rx[IT_X];
    auto load_tiles = [&](int64_t mb) {
#pragma unroll
        for (int i = 0; i < IT_Y; ++i) {
            const int f = tid + 256 * i;
            ry[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (f < GK * BNT / 4) {
                const int r = f / (BNT / 4);
                const int c = (f % (BNT / 4)) * 4;
                const int64_t m = mb + r;
                if (m < m_hi) ry[i] = load4_guard(dY + (size_t)m * lddy + n0 + c, N - (n0 + c), vec_y);
            }
        }
#pragma unroll
        for (int i = 0; i < IT_X; ++i) {
            const int f = tid + 256 * i;
            rx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (f < GK * BKT / 4) {
                const int r = f / (BKT / 4);
                const int c = (f % (BKT / 4)) * 4;
                const int64_t m = mb + r;
                if (m < m_hi) {
                    int64_t src = m;
                    if (rows) src = rows[m];
                    if (src >= 0) rx[i] = load4_guard(A + (size_t)src * lda + k0 + c, K - (k0 + c), vec_a);
                }
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < IT_Y; ++i) {
            const int f = tid + 256 * i;
            if (f < GK * BNT / 4) *reinterpret_cast<f32x4*>(&Ys[(f / (BNT / 4)) * LDY_S + (f % (BNT / 4)) * 4]) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < IT_X; ++i) {
            const int f = tid + 256 * i;
            if (f < GK * BKT / 4) *reinterpret_cast<f32x4*>(&Xs[(f / (BKT / 4)) * LDX_S + (f % (BKT / 4)) * 4]) = rx[i];
        }
    };

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    if (m_lo < m_hi) {
        load_tiles(m_lo);
        store_tiles();
        __syncthreads();
        for (int64_t mb = m_lo; mb < m_hi; mb += GK) {
            const bool more = mb + GK < m_hi;
            if (more) load_tiles(mb + GK);
#pragma unroll
            for (int kk = 0; kk < GK / 2; ++kk) {
                const int k = 2 * kk + lh;
                float a[TN], b[TK];
#pragma unroll
                for (int i = 0; i < TN; ++i) a[i] = Ys[k * LDY_S + wn0 + i * 32 + lr];
#pragma unroll
                for (int j = 0; j < TK; ++j) b[j] = Xs[k * LDX_S + wk0 + j * 32 + lr];
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (do_bias && tid < BNT) {
#pragma unroll
                for (int r = 0; r < GK; ++r) bsum += Ys[r * LDY_S + tid];
            }
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }

    float* out = slab + (size_t)s * N * K;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int col = k0 + wk0 + j * 32 + lr;
            if (col >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wn0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < N) out[(size_t)row * K + col] = acc[i][j][r];
            }
        }
    }
    if (do_bias && tid < BNT && n0 + tid < N) bslab[(size_t)s * N + n0 + tid] = bsum;
}


static bool aligned16(const void* p) { return ((uintptr_t)p % 16) == 0; }

struct WgradPlan {
    int tiles_n, tiles_k, S;
    int64_t m_chunk;
    bool narrow;
};

static WgradPlan wgrad_plan(int64_t M, int N, int K) {
    WgradPlan p;
    p.narrow = N <= 32;
    const int bnt = p.narrow ? 32 : 128;
    p.tiles_n = (int)mg_ceil_div(N, bnt);
    p.tiles_k = (int)mg_ceil_div(K, 128);
    const int64_t tiles = (int64_t)p.tiles_n * p.tiles_k;
    int64_t S = mg_ceil_div(1024, tiles);
    // at least 512 rows per split where there are tiles enough to fill the chip; a one- or two-tile product (the 128 -> 32 -> 1 tail of
    // the README model at phone rate: M = 22,528) would otherwise run on M / 512 = 44 workgroups - 31-39 us for 0.2 GFLOP
    const int64_t max_s = mg_ceil_div(M, tiles <= 2 ? 128 : 512);
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    if (S > 65535) S = 65535;
    p.m_chunk = mg_align_up((size_t)mg_ceil_div(M, S), GK);
    p.S = (int)mg_ceil_div(M, p.m_chunk);
    if (p.S < 1) p.S = 1;
    return p;
}

extern "C" {

int mg_linear_fwd_f32(const float* A, int lda, const int32_t* rows, int64_t M, int K, const float* W, const float* bias,
                      int N, float* Y, int ldy, int act, void* stream) {
    MG_CHECK_ARG(A && W && Y && M >= 0 && K > 0 && N > 0 && lda >= K && ldy >= N, "mg_linear_fwd_f32: bad arguments (M=%lld K=%d N=%d lda=%d ldy=%d)",
                 (long long)M, K, N, lda, ldy);
    MG_CHECK_ARG(act == MG_ACT_NONE || act == MG_ACT_SIGMOID, "mg_linear_fwd_f32: unknown activation %d", act);
    if (M == 0) return MG_OK;
    const int vec_a = (lda % 4 == 0) && aligned16(A);
    const int vec_b = (K % 4 == 0) && aligned16(W);
    hipStream_t st = (hipStream_t)stream;
    if (N <= 32) {
        const int tn = (int)mg_ceil_div(N, 32);
        const int64_t blocks = mg_ceil_div(M, 128) * tn;
        MG_CHECK_ARG(blocks < 2147483647LL, "mg_linear_fwd_f32: grid too large");
        if (act == MG_ACT_SIGMOID)
            hipLaunchKernelGGL((gemm_f32_kernel<128, 32, 4, 1, false, EPI_BIAS_SIGMOID>), dim3((unsigned)blocks), dim3(256), 0, st, A, lda, rows, M, K, W, K, N, bias, nullptr, 0, Y, ldy, tn, vec_a, vec_b);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<128, 32, 4, 1, false, EPI_BIAS>), dim3((unsigned)blocks), dim3(256), 0, st, A, lda, rows, M, K, W, K, N, bias, nullptr, 0, Y, ldy, tn, vec_a, vec_b);
    } else {
        const int tn = (int)mg_ceil_div(N, 128);
        const int64_t blocks = mg_ceil_div(M, 128) * tn;
        MG_CHECK_ARG(blocks < 2147483647LL, "mg_linear_fwd_f32: grid too large");
        if (act == MG_ACT_SIGMOID)
            hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 2, 2, false, EPI_BIAS_SIGMOID>), dim3((unsigned)blocks), dim3(256), 0, st, A, lda, rows, M, K, W, K, N, bias, nullptr, 0, Y, ldy, tn, vec_a, vec_b);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 2, 2, false, EPI_BIAS>), dim3((unsigned)blocks), dim3(256), 0, st, A, lda, rows, M, K, W, K, N, bias, nullptr, 0, Y, ldy, tn, vec_a, vec_b);
    }
    MG_CHECK_LAUNCH("mg_linear_fwd_f32");
    return MG_OK;
}

int mg_linear_dgrad_f32(const float* dY, int64_t M, int N, const float* W, int K, const float* H, float* dX, void* stream) {
    MG_CHECK_ARG(dY && W && dX && M >= 0 && N > 0 && K > 0, "mg_linear_dgrad_f32: bad arguments (M=%lld N=%d K=%d)", (long long)M, N, K);
    if (M == 0) return MG_OK;
    const int vec_a = (N % 4 == 0) && aligned16(dY);
    const int vec_b = (K % 4 == 0) && aligned16(W);
    hipStream_t st = (hipStream_t)stream;
    // C[M,K] = dY[M,N] * W[N,K]: contraction over N, B operand given as [Kc=N, Ncols=K] row-major.
    if (K <= 32) {
        const int tn = (int)mg_ceil_div(K, 32);
        const int64_t blocks = mg_ceil_div(M, 128) * tn;
        if (H)
            hipLaunchKernelGGL((gemm_f32_kernel<128, 32, 4, 1, true, EPI_SIGMOID_GRAD>), dim3((unsigned)blocks), dim3(256), 0, st, dY, N, nullptr, M, N, W, K, K, nullptr, H, K, dX, K, tn, vec_a, vec_b);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<128, 32, 4, 1, true, EPI_BIAS>), dim3((unsigned)blocks), dim3(256), 0, st, dY, N, nullptr, M, N, W, K, K, nullptr, nullptr, 0, dX, K, tn, vec_a, vec_b);
    } else {
        const int tn = (int)mg_ceil_div(K, 128);
        const int64_t blocks = mg_ceil_div(M, 128) * tn;
        MG_CHECK_ARG(blocks < 2147483647LL, "mg_linear_dgrad_f32: grid too large");
        if (H)
            hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 2, 2, true, EPI_SIGMOID_GRAD>), dim3((unsigned)blocks), dim3(256), 0, st, dY, N, nullptr, M, N, W, K, K, nullptr, H, K, dX, K, tn, vec_a, vec_b);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 2, 2, true, EPI_BIAS>), dim3((unsigned)blocks), dim3(256), 0, st, dY, N, nullptr, M, N, W, K, K, nullptr, nullptr, 0, dX, K, tn, vec_a, vec_b);
    }
    MG_CHECK_LAUNCH("mg_linear_dgrad_f32");
    return MG_OK;
}

size_t mg_linear_wgrad_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 256;
    const WgradPlan p = wgrad_plan(M, N, K);
    int64_t S = p.S;
    if (N % 128 == 0) {   // the wide bf16 kernel (gemm_bf16_big.hip) may split further: size for the larger plan
        int64_t sb = mg_ceil_div(256, N / 128);
        int64_t chunk = mg_align_up((size_t)mg_ceil_div(M, sb), 32);
        while (chunk > 4096) {
            sb *= 2;
            chunk = mg_align_up((size_t)mg_ceil_div(M, sb), 32);
        }
        sb = mg_align_up((size_t)mg_ceil_div(M, chunk), 8);
        if (sb > S) S = sb;
        if (g_mg_tuning[MG_TUNE_WGRAD_SPLITS] > S) S = mg_align_up((size_t)g_mg_tuning[MG_TUNE_WGRAD_SPLITS], 8);   // experiment knob
    }
    return mg_align_up((size_t)S * ((size_t)N * K + (size_t)N) * sizeof(float), 256);
}

int mg_linear_wgrad_f32(const float* dY, const float* A, int lda, const int32_t* rows, int64_t M, int N, int K, float* dW,
                        float* db, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(dY && A && dW && M > 0 && N > 0 && K > 0 && lda >= K, "mg_linear_wgrad_f32: bad arguments (M=%lld N=%d K=%d lda=%d)",
                 (long long)M, N, K, lda);
    if (!workspace || workspace_bytes < mg_linear_wgrad_workspace_bytes(M, N, K)) {
        mg_set_error("mg_linear_wgrad_f32: workspace of %zu bytes needed, got %zu", mg_linear_wgrad_workspace_bytes(M, N, K), workspace_bytes);
        return MG_EWORKSPACE;
    }
    const WgradPlan p = wgrad_plan(M, N, K);
    float* slab = (float*)workspace;
    float* bslab = slab + (size_t)p.S * N * K;
    const int vec_y = (N % 4 == 0) && aligned16(dY);
    const int vec_a = (lda % 4 == 0) && aligned16(A);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(p.tiles_n * p.tiles_k), (unsigned)p.S);
    if (p.narrow)
        hipLaunchKernelGGL((wgrad_f32_kernel<32, 128, 1, 4>), grid, dim3(256), 0, st, dY, N, A, lda, rows, M, N, K, p.m_chunk, slab, db ? bslab : nullptr, p.tiles_k, vec_y, vec_a);
    else
        hipLaunchKernelGGL((wgrad_f32_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, dY, N, A, lda, rows, M, N, K, p.m_chunk, slab, db ? bslab : nullptr, p.tiles_k, vec_y, vec_a);
    MG_CHECK_LAUNCH("mg_linear_wgrad_f32/partial");
    const int64_t nk = (int64_t)N * K;
    mg_launch_slab_reduce(slab, nk, nk, p.S, dW, accumulate, st);
    MG_CHECK_LAUNCH("mg_linear_wgrad_f32/reduce");
    if (db) {
        mg_launch_slab_reduce(bslab, N, N, p.S, db, accumulate, st);
        MG_CHECK_LAUNCH("mg_linear_wgrad_f32/reduce_bias");
    }
    return MG_OK;
}

}  // extern "C"
