// K2 (bf16 throughput mode) - Linear forward / dgrad / wgrad on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).
//
// Reference: the nn.Linear + nn.Sigmoid stack of README.rst:65-73 run by SequentialWithRecurrent.forward
// (morgana/utils.py:401-418) and its autograd backward; the reference runs them as fp32 ATen addmm/sigmoid kernels.
//
// Operand maps of v_mfma_f32_32x32x16_bf16 (lane l, r = l&31, h = l>>5): A[row r][k = 8h+j], B[k = 8h+j][col r],
// j = 0..7 (one 16-byte fragment); C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
//
//   gemm_nt  C = A' B^T   both operands contraction-contiguous ([m][k], [n][k]): fragments are ds_read_b128 from
//            64-byte LDS rows whose 16-byte chunks are XOR-swizzled by (row>>2)&3 (conflict free for the 4x16 lane
//            groups of ds_read_b128).  Used for forward (B = W) and dgrad (A = dY, B = W^T).
//   gemm_tn  C = A^T B    both operands contraction-strided ([m][n], [m][k]): tiles are straight row copies and the
//            fragments come from ds_read_b64_tr_b16 (hardware transpose read); rows padded to 320 bytes so the 4 rows
//            a half wave touches land on disjoint banks.  Used for wgrad, split over M into fp32 slabs that are
//            summed in a fixed order (deterministic; no float atomics).
// The layer-1 gather (upsample_to_repetitions) is fused into the A-tile loaders through the `rows` array.
// Epilogues stage the fp32 accumulators through LDS so that every global store is a 16-byte lane (8 bf16).
#include "common.h"
#include "expand_reduce.h"
#include "phone_front.h"
#include "slab_reduce.h"

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfv4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define EPI_BIAS 0
#define EPI_BIAS_SIGMOID 1
#define EPI_SIGMOID_GRAD 2

#define NT_BK 32  // contraction depth per LDS tile (2 MFMA k-steps)

__device__ __forceinline__ u32x4 ldg16(const uint16_t* p) { return *reinterpret_cast<const u32x4*>(p); }

// C[M,N] (bf16, ldc) = epi( A'[M,Kc] * B[N,Kc]^T ).  lda/ldb multiples of 8, padding columns zero.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                           int64_t M, const uint16_t* __restrict__ Bm, int ldb, int N,
                                                           const float* __restrict__ bias, const uint16_t* __restrict__ H, int ldh,
                                                           void* __restrict__ Cv, int ldc, int tiles_n, int c_f32) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int IT_A = (BM * 4 + 255) / 256;
    constexpr int IT_B = (BN * 4 + 255) / 256;
    constexpr int STG_LD = WN + 4;                       // fp32 staging row pitch (floats)
    constexpr int TILE_BYTES = (BM + BN) * 64;
    constexpr int STG_BYTES = 4 * WM * STG_LD * 4;
    constexpr int LDS_BYTES = TILE_BYTES > STG_BYTES ? TILE_BYTES : STG_BYTES;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    unsigned char* As = smem;
    unsigned char* Bs = smem + BM * 64;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * WN;
    const int64_t m0 = (int64_t)(blockIdx.x / tiles_n) * BM;
    const int n0 = (blockIdx.x % tiles_n) * BN;
    const int kc = lda < ldb ? lda : ldb;                // columns beyond are zero in the shorter operand
    const int n_kt = (kc + NT_BK - 1) / NT_BK;

    const uint16_t* a_ptr[IT_A];
#pragma unroll
    for (int i = 0; i < IT_A; ++i) {
        const int f = tid + 256 * i;
        const int64_t m = m0 + (f >> 2);
        a_ptr[i] = nullptr;
        if (f < BM * 4 && m < M) {
            if (rows) {
                const int r = rows[m];
                if (r >= 0) a_ptr[i] = A + (size_t)r * lda;
            } else {
                a_ptr[i] = A + (size_t)m * lda;
            }
        }
    }
    const uint16_t* b_ptr[IT_B];
#pragma unroll
    for (int i = 0; i < IT_B; ++i) {
        const int f = tid + 256 * i;
        const int n = n0 + (f >> 2);
        b_ptr[i] = (f < BN * 4 && n < N) ? Bm + (size_t)n * ldb : nullptr;
    }

    u32x4 ra[IT_A], rb[IT_B];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < IT_A; ++i) {
            const int k = k0 + ((tid + 256 * i) & 3) * 8;
            ra[i] = (a_ptr[i] && k < lda) ? ldg16(a_ptr[i] + k) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int i = 0; i < IT_B; ++i) {
            const int k = k0 + ((tid + 256 * i) & 3) * 8;
            rb[i] = (b_ptr[i] && k < ldb) ? ldg16(b_ptr[i] + k) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < IT_A; ++i) {
            const int f = tid + 256 * i;
            if (f < BM * 4) {
                const int row = f >> 2, c = f & 3;
                *reinterpret_cast<u32x4*>(As + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < IT_B; ++i) {
            const int f = tid + 256 * i;
            if (f < BN * 4) {
                const int row = f >> 2, c = f & 3;
                *reinterpret_cast<u32x4*>(Bs + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = rb[i];
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

    const int lr = lane & 31, lh = lane >> 5;
    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int kt = 0; kt < n_kt; ++kt) {
        if (kt + 1 < n_kt) load_tiles((kt + 1) * NT_BK);
#pragma unroll
        for (int ks = 0; ks < NT_BK / 16; ++ks) {
            const int c = ks * 2 + lh;
            bfv8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm0 + i * 32 + lr;
                a[i] = *reinterpret_cast<const bfv8*>(As + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn0 + j * 32 + lr;
                b[j] = *reinterpret_cast<const bfv8*>(Bs + row * 64 + ((c ^ ((row >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < n_kt) {
            store_tiles();
            __syncthreads();
        }
    }

    // Epilogue: stage this wave's WM x WN fp32 tile in LDS, then 8 columns (16 bytes of bf16) per lane.
    float* stg = reinterpret_cast<float*>(smem) + wave * WM * STG_LD;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) stg[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * STG_LD + j * 32 + lr] = acc[i][j][r];
    __syncthreads();
    constexpr int CPR = WN / 8;          // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;        // rows per wave instruction
#pragma unroll
    for (int it = 0; it < WM / RPI; ++it) {
        const int rl = it * RPI + lane / CPR;
        const int cl = (lane % CPR) * 8;
        const int64_t row = m0 + wm0 + rl;
        const int col = n0 + wn0 + cl;
        if (row >= M || col >= ldc) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(&stg[rl * STG_LD + cl]);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(&stg[rl * STG_LD + cl + 4]);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        bfv8 hv;
        if (EPI == EPI_SIGMOID_GRAD) hv = *reinterpret_cast<const bfv8*>(H + (size_t)row * ldh + col);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = v[e];
            if (col + e >= N) x = 0.f;
            else if (EPI == EPI_BIAS) x += bias ? bias[col + e] : 0.f;
            else if (EPI == EPI_BIAS_SIGMOID) x = mg_sigmoid_fast(x + (bias ? bias[col + e] : 0.f));
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
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad: slab[s][n][k] = sum_{m in split s} dY[m][n] * A'[m][k];  bslab[s][n] = sum_m dY[m][n]
// ---------------------------------------------------------------------------------------------------------------------
#define TN_BM 32  // contraction rows per LDS tile (2 MFMA k-steps)

__device__ __forceinline__ bfv8 tr_frag(const unsigned char* tile, int pitch, int col0, int lane, int ks) {
    // 32x32x16 operand whose contraction index is the LDS row: lane group g = lane>>4 covers operand rows/cols
    // 16*(g&1) .. +15 and contraction rows 8*(g>>1) .. +7 of k-step ks; lane 4q+p of the group supplies the address
    // of row q, columns 4p..4p+3 of its 4x16 block (ds_read_b64_tr_b16), two blocks give the 8 contraction values.
    const int i = lane & 15, g = lane >> 4;
    const int q = i >> 2, p = i & 3;
    const int mrow = ks * 16 + 8 * (g >> 1) + q;
    const unsigned char* addr = tile + mrow * pitch + (col0 + 16 * (g & 1) + 4 * p) * 2;
    const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(addr));
    const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(addr + 4 * pitch));
    bfv8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <int BNT, int BKT, int WAVES_N, int WAVES_K>
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const uint16_t* __restrict__ dY, int lddy, const uint16_t* __restrict__ A, int lda,
                                                         const int32_t* __restrict__ rows, int64_t M, int N, int K, int64_t m_chunk,
                                                         float* __restrict__ slab, float* __restrict__ bslab, int tiles_k) {
    constexpr int WN = BNT / WAVES_N, WK = BKT / WAVES_K;
    constexpr int TN = WN / 32, TK = WK / 32;
    constexpr int PY = BNT * 2 + ((BNT * 2) % 256 == 0 ? 64 : 0);   // LDS row pitch in bytes
    constexpr int PX = BKT * 2 + ((BKT * 2) % 256 == 0 ? 64 : 0);
    constexpr int CY = BNT / 8, CX = BKT / 8;                      // 16-byte chunks per row
    constexpr int IT_Y = (TN_BM * CY + 255) / 256;
    constexpr int IT_X = (TN_BM * CX + 255) / 256;
    static_assert(WAVES_N * WAVES_K == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) unsigned char Ys[TN_BM * PY];
    __shared__ __attribute__((aligned(16))) unsigned char Xs[TN_BM * PX];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn0 = (wave / WAVES_K) * WN, wk0 = (wave % WAVES_K) * WK;
    const int n0 = (blockIdx.x / tiles_k) * BNT;
    const int k0 = (blockIdx.x % tiles_k) * BKT;
    const int s = blockIdx.y;
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t m_hi = min(M, m_lo + m_chunk);
    const bool do_bias = (blockIdx.x % tiles_k) == 0 && bslab != nullptr;

    u32x4 ry[IT_Y], rx[IT_X];
    auto load_tiles = [&](int64_t mb) {
#pragma unroll
        for (int i = 0; i < IT_Y; ++i) {
            const int f = tid + 256 * i;
            ry[i] = u32x4{0u, 0u, 0u, 0u};
            if (f < TN_BM * CY) {
                const int64_t m = mb + f / CY;
                const int c = n0 + (f % CY) * 8;
                if (m < m_hi && c < lddy) ry[i] = ldg16(dY + (size_t)m * lddy + c);
            }
        }
#pragma unroll
        for (int i = 0; i < IT_X; ++i) {
            const int f = tid + 256 * i;
            rx[i] = u32x4{0u, 0u, 0u, 0u};
            if (f < TN_BM * CX) {
                const int64_t m = mb + f / CX;
                const int c = k0 + (f % CX) * 8;
                if (m < m_hi && c < lda) {
                    int64_t src = m;
                    if (rows) src = rows[m];
                    if (src >= 0) rx[i] = ldg16(A + (size_t)src * lda + c);
                }
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < IT_Y; ++i) {
            const int f = tid + 256 * i;
            if (f < TN_BM * CY) *reinterpret_cast<u32x4*>(Ys + (f / CY) * PY + (f % CY) * 16) = ry[i];
        }
#pragma unroll
        for (int i = 0; i < IT_X; ++i) {
            const int f = tid + 256 * i;
            if (f < TN_BM * CX) *reinterpret_cast<u32x4*>(Xs + (f / CX) * PX + (f % CX) * 16) = rx[i];
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

    if (m_lo < m_hi) {
        load_tiles(m_lo);
        store_tiles();
        __syncthreads();
        for (int64_t mb = m_lo; mb < m_hi; mb += TN_BM) {
            const bool more = mb + TN_BM < m_hi;
            if (more) load_tiles(mb + TN_BM);
#pragma unroll
            for (int ks = 0; ks < TN_BM / 16; ++ks) {
                bfv8 a[TN], b[TK];
#pragma unroll
                for (int i = 0; i < TN; ++i) a[i] = tr_frag(Ys, PY, wn0 + i * 32, lane, ks);
#pragma unroll
                for (int j = 0; j < TK; ++j) b[j] = tr_frag(Xs, PX, wk0 + j * 32, lane, ks);
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (do_bias && tid < BNT) {
#pragma unroll
                for (int r = 0; r < TN_BM; ++r) bsum += mg_bf2f(*reinterpret_cast<const uint16_t*>(Ys + r * PY + tid * 2));
            }
            __syncthreads();
            if (more) {
                store_tiles();
                __syncthreads();
            }
        }
    }

    const int lr = lane & 31, lh = lane >> 5;
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


struct WgradPlanB {
    int tiles_n, tiles_k, S;
    int64_t m_chunk;
    bool narrow;
};

// Must match wgrad_plan() in gemm_f32.hip (shared workspace-size helper mg_linear_wgrad_workspace_bytes).
static WgradPlanB wgrad_plan_b(int64_t M, int N, int K) {
    WgradPlanB p;
    p.narrow = N <= 32;
    const int bnt = p.narrow ? 32 : 128;
    p.tiles_n = (int)mg_ceil_div(N, bnt);
    p.tiles_k = (int)mg_ceil_div(K, 128);
    const int64_t tiles = (int64_t)p.tiles_n * p.tiles_k;
    int64_t S = mg_ceil_div(1024, tiles);
    const int64_t max_s = mg_ceil_div(M, 512);
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    if (S > 65535) S = 65535;
    if (g_mg_tuning[MG_TUNE_WGRAD_ORDER] == 3 && g_mg_tuning[MG_TUNE_WGRAD_SPLITS] > 0) S = g_mg_tuning[MG_TUNE_WGRAD_SPLITS];   // experiment
    p.m_chunk = mg_align_up((size_t)mg_ceil_div(M, S), 32);
    p.S = (int)mg_ceil_div(M, p.m_chunk);
    if (p.S < 1) p.S = 1;
    return p;
}

static bool al16(const void* p) { return ((uintptr_t)p % 16) == 0; }

// large-tile kernels (gemm_bf16_big.hip)
int mg_try_nt_big(const uint16_t* A, int lda, const int32_t* rows, int64_t M, int K, const uint16_t* Bm, int ldb, int N,
                  const float* bias, const uint16_t* H, int ldh, const int32_t* h_rows, void* C, int ldc, int c_f32, int epi,
                  hipStream_t st);
int mg_try_nt_runs(const uint16_t* A, int lda, const int32_t* rows, int64_t M, int K, const uint16_t* Bm, int ldb, int N,
                   const float* bias, uint16_t* C, int ldc, int sigmoid, hipStream_t st);       // gemm_nt_runs.hip
int mg_wgrad_big_plan(int64_t M, int N, int K, int lda, int lddy, int* S_out, int* m_chunk_out);
int mg_launch_wgrad_big(const uint16_t* dY, int lddy, const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K,
                        int S, int m_chunk, float* slab, float* bslab, int64_t sstride, hipStream_t st, const int32_t* dy_rows = nullptr, int x3 = 0);
int mg_launch_nt_persist_x3(const PhoneFrontArgs* pf, const uint16_t* A, int lda, int64_t M, int K, const uint16_t* Bm, int ldb, int N,
                            const float* bias, uint16_t* C, int ldc, int epi, hipStream_t st, int* front_rode);
int mg_launch_nt_big_x3_f32(const uint16_t* A, int lda, int64_t M, int K, const uint16_t* Bm, int ldb, int N, const float* bias, float* C,
                            int ldc, int epi, hipStream_t st, int parts = 1);
int mg_launch_phone_front_gemm(const PhoneFrontArgs& pf, const uint16_t* A, int lda, int64_t M, int K, const uint16_t* Bm, int ldb, int N,
                               const float* bias, uint16_t* C, int ldc, int epi, hipStream_t st);
extern "C" int mg_phone_front_check(const int64_t* dur, int B, int P, int T, const float* target, int extra, const int32_t* rows32,
                                    const int32_t* rows_mapped, const int32_t* seg_start, const int32_t* seg_end, const float* ybar,
                                    const float* weight, const void* workspace, size_t workspace_bytes, const char* who);
int mg_launch_wgrad_dgrad_pair(const uint16_t* dY, int lddy, const uint16_t* A, int lda, int64_t M, int N, int K, const uint16_t* WT, int ldwt,
                               uint16_t* dX, int lddx, float* slab, int64_t sstride, size_t slab_floats, int* S_out, hipStream_t st,
                               const ExpandReduceArgs* rider, int x3 = 0, float* colsum = nullptr);

extern "C" {

int mg_linear_fwd_bf16(const uint16_t* A, int lda, const int32_t* rows, int64_t M, int K, const uint16_t* W, int ldw,
                       const float* bias, int N, void* Y, int ldy, int y_f32, int act, void* stream) {
    MG_CHECK_ARG(A && W && Y && M >= 0 && K > 0 && N > 0, "mg_linear_fwd_bf16: bad arguments (M=%lld K=%d N=%d)", (long long)M, K, N);
    MG_CHECK_ARG(lda >= K && ldw >= K && ldy >= N && lda % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0,
                 "mg_linear_fwd_bf16: lda=%d ldw=%d ldy=%d must be multiples of 8 and cover K=%d / N=%d", lda, ldw, ldy, K, N);
    MG_CHECK_ARG(al16(A) && al16(W) && al16(Y), "mg_linear_fwd_bf16: buffers must be 16-byte aligned");
    const int runs_hint = act & MG_ACT_ROWS_RUNS;
    act &= ~MG_ACT_ROWS_RUNS;
    MG_CHECK_ARG(act == MG_ACT_NONE || act == MG_ACT_SIGMOID, "mg_linear_fwd_bf16: unknown activation %d", act);
    if (M == 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    // rows made of runs (frame map of upsample_to_repetitions): the gathered operand staged once per distinct row, two workgroups per CU
    if (runs_hint && rows && !y_f32 && ldy == N && g_mg_tuning[MG_TUNE_FORM] != 14 &&
        mg_try_nt_runs(A, lda, rows, M, K, W, ldw, N, bias, (uint16_t*)Y, ldy, act == MG_ACT_SIGMOID, st) > 0) {
        MG_CHECK_LAUNCH("mg_linear_fwd_bf16/runs");
        return MG_OK;
    }
    if (ldy == N && mg_try_nt_big(A, lda, rows, M, K, W, ldw, N, bias, nullptr, 0, nullptr, Y, ldy, y_f32,
                                  act == MG_ACT_SIGMOID ? EPI_BIAS_SIGMOID : EPI_BIAS, st) > 0) {
        MG_CHECK_LAUNCH("mg_linear_fwd_bf16/big");
        return MG_OK;
    }
    if (N <= 32) {
        const int tn = (int)mg_ceil_div(ldy, 32);
        const int64_t blocks = mg_ceil_div(M, 128) * tn;
        MG_CHECK_ARG(blocks < 2147483647LL, "mg_linear_fwd_bf16: grid too large");
        if (act == MG_ACT_SIGMOID)
            hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 32, 4, 1, EPI_BIAS_SIGMOID>), dim3((unsigned)blocks), dim3(256), 0, st, A, lda, rows, M, W, ldw, N, bias, nullptr, 0, Y, ldy, tn, y_f32);
        else
            hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 32, 4, 1, EPI_BIAS>), dim3((unsigned)blocks), dim3(256), 0, st, A, lda, rows, M, W, ldw, N, bias, nullptr, 0, Y, ldy, tn, y_f32);
    } else {
        const int tn = (int)mg_ceil_div(ldy, 128);
        const int64_t blocks = mg_ceil_div(M, 128) * tn;
        MG_CHECK_ARG(blocks < 2147483647LL, "mg_linear_fwd_bf16: grid too large");
        if (act == MG_ACT_SIGMOID)
            hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 128, 2, 2, EPI_BIAS_SIGMOID>), dim3((unsigned)blocks), dim3(256), 0, st, A, lda, rows, M, W, ldw, N, bias, nullptr, 0, Y, ldy, tn, y_f32);
        else
            hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 128, 2, 2, EPI_BIAS>), dim3((unsigned)blocks), dim3(256), 0, st, A, lda, rows, M, W, ldw, N, bias, nullptr, 0, Y, ldy, tn, y_f32);
    }
    MG_CHECK_LAUNCH("mg_linear_fwd_bf16");
    return MG_OK;
}

int mg_linear_dgrad_bf16(const uint16_t* dY, int lddy, int64_t M, int N, const uint16_t* WT, int ldwt, int K,
                         const uint16_t* H, int ldh, void* dX, int lddx, int dx_f32, void* stream) {
    MG_CHECK_ARG(dY && WT && dX && M >= 0 && N > 0 && K > 0, "mg_linear_dgrad_bf16: bad arguments (M=%lld N=%d K=%d)", (long long)M, N, K);
    MG_CHECK_ARG(lddy >= N && ldwt >= N && lddx >= K && lddy % 8 == 0 && ldwt % 8 == 0 && lddx % 8 == 0 && (!H || (ldh >= K && ldh % 8 == 0)),
                 "mg_linear_dgrad_bf16: leading dimensions must be multiples of 8 and cover N=%d / K=%d (lddy=%d ldwt=%d lddx=%d ldh=%d)", N, K, lddy, ldwt, lddx, ldh);
    MG_CHECK_ARG(al16(dY) && al16(WT) && al16(dX) && (!H || al16(H)), "mg_linear_dgrad_bf16: buffers must be 16-byte aligned");
    if (M == 0) return MG_OK;
    hipStream_t st = (hipStream_t)stream;
    // C[M,K] = dY[M,N] * WT[K,N]^T : the NT kernel with contraction N, output width K.
    if (lddx == K && mg_try_nt_big(dY, lddy, nullptr, M, N, WT, ldwt, K, nullptr, H, ldh, nullptr, dX, lddx, dx_f32,
                                   H ? EPI_SIGMOID_GRAD : EPI_BIAS, st) > 0) {
        MG_CHECK_LAUNCH("mg_linear_dgrad_bf16/big");
        return MG_OK;
    }
    if (K <= 32) {
        const int tn = (int)mg_ceil_div(lddx, 32);
        const int64_t blocks = mg_ceil_div(M, 128) * tn;
        if (H)
            hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 32, 4, 1, EPI_SIGMOID_GRAD>), dim3((unsigned)blocks), dim3(256), 0, st, dY, lddy, nullptr, M, WT, ldwt, K, nullptr, H, ldh, dX, lddx, tn, dx_f32);
        else
            hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 32, 4, 1, EPI_BIAS>), dim3((unsigned)blocks), dim3(256), 0, st, dY, lddy, nullptr, M, WT, ldwt, K, nullptr, nullptr, 0, dX, lddx, tn, dx_f32);
    } else {
        const int tn = (int)mg_ceil_div(lddx, 128);
        const int64_t blocks = mg_ceil_div(M, 128) * tn;
        MG_CHECK_ARG(blocks < 2147483647LL, "mg_linear_dgrad_bf16: grid too large");
        if (H)
            hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 128, 2, 2, EPI_SIGMOID_GRAD>), dim3((unsigned)blocks), dim3(256), 0, st, dY, lddy, nullptr, M, WT, ldwt, K, nullptr, H, ldh, dX, lddx, tn, dx_f32);
        else
            hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, 128, 2, 2, EPI_BIAS>), dim3((unsigned)blocks), dim3(256), 0, st, dY, lddy, nullptr, M, WT, ldwt, K, nullptr, nullptr, 0, dX, lddx, tn, dx_f32);
    }
    MG_CHECK_LAUNCH("mg_linear_dgrad_bf16");
    return MG_OK;
}

// dX = (dY W) * H[h_rows] (1 - H[h_rows]): mg_linear_dgrad_bf16 with the sigmoid outputs read from a TABLE (the phone-rate first
// layer, csrc/phone_rate.hip: frame m's activation row is table row h_rows[m], >= 0).  Wide-tile kernel only.
int mg_linear_dgrad_gathered_bf16(const uint16_t* dY, int lddy, int64_t M, int N, const uint16_t* WT, int ldwt, int K,
                                  const uint16_t* H, int ldh, const int32_t* h_rows, void* dX, int lddx, int dx_f32, void* stream) {
    MG_CHECK_ARG(dY && WT && dX && H && h_rows && M > 0 && N > 0 && K > 0, "mg_linear_dgrad_gathered_bf16: bad arguments (M=%lld N=%d K=%d)",
                 (long long)M, N, K);
    MG_CHECK_ARG(lddy >= N && ldwt >= N && lddx >= K && lddy % 8 == 0 && ldwt % 8 == 0 && lddx % 8 == 0 && ldh >= K && ldh % 8 == 0,
                 "mg_linear_dgrad_gathered_bf16: leading dimensions must be multiples of 8 and cover N=%d / K=%d", N, K);
    MG_CHECK_ARG(al16(dY) && al16(WT) && al16(dX) && al16(H), "mg_linear_dgrad_gathered_bf16: buffers must be 16-byte aligned");
    if (!(lddx == K && mg_try_nt_big(dY, lddy, nullptr, M, N, WT, ldwt, K, nullptr, H, ldh, h_rows, dX, lddx, dx_f32, EPI_SIGMOID_GRAD,
                                     (hipStream_t)stream) > 0)) {
        mg_set_error("mg_linear_dgrad_gathered_bf16: shape outside the wide-tile kernel (M=%lld N=%d K=%d lddy=%d ldwt=%d)", (long long)M, N, K,
                     lddy, ldwt);
        return MG_EINVAL;
    }
    MG_CHECK_LAUNCH("mg_linear_dgrad_gathered_bf16");
    return MG_OK;
}

int mg_linear_wgrad_bf16(const uint16_t* dY, int lddy, const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N,
                         int K, float* dW, float* db, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(dY && A && dW && M > 0 && N > 0 && K > 0, "mg_linear_wgrad_bf16: bad arguments (M=%lld N=%d K=%d)", (long long)M, N, K);
    MG_CHECK_ARG(lddy >= N && lda >= K && lddy % 8 == 0 && lda % 8 == 0, "mg_linear_wgrad_bf16: lddy=%d lda=%d must be multiples of 8 covering N=%d / K=%d", lddy, lda, N, K);
    MG_CHECK_ARG(al16(dY) && al16(A), "mg_linear_wgrad_bf16: buffers must be 16-byte aligned");
    if (!workspace || workspace_bytes < mg_linear_wgrad_workspace_bytes(M, N, K)) {
        mg_set_error("mg_linear_wgrad_bf16: workspace of %zu bytes needed, got %zu", mg_linear_wgrad_workspace_bytes(M, N, K), workspace_bytes);
        return MG_EWORKSPACE;
    }
    WgradPlanB p = wgrad_plan_b(M, N, K);
    float* slab = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    int big_s = 0, big_chunk = 0;
    const bool big = g_mg_tuning[MG_TUNE_WGRAD_ORDER] != 3 && mg_wgrad_big_plan(M, N, K, lda, lddy, &big_s, &big_chunk) > 0;
    if (big) p.S = big_s;
    const int64_t nk = (int64_t)N * K;
    if (big) {
        // wide tiles: split s of the workspace = [N*K weight partials | N bias partials]; with db right behind dW (one flat
        // gradient buffer) a single reduce launch finishes both
        const int64_t sstride = nk + N;
        float* bslab = slab + nk;
        if ((size_t)big_s * (size_t)sstride * sizeof(float) > workspace_bytes) {      // never launch past the caller's workspace
            mg_set_error("mg_linear_wgrad_bf16: %d split slabs of %lld floats do not fit the %zu-byte workspace", big_s, (long long)sstride,
                         workspace_bytes);
            return MG_EWORKSPACE;
        }
        mg_launch_wgrad_big(dY, lddy, A, lda, rows, M, N, K, big_s, big_chunk, slab, db ? bslab : nullptr, sstride, st);
        MG_CHECK_LAUNCH("mg_linear_wgrad_bf16/partial");
        if (db && db == dW + nk) {
            mg_launch_slab_reduce(slab, sstride, sstride, p.S, dW, accumulate, st);
        } else {
            mg_launch_slab_reduce(slab, nk, sstride, p.S, dW, accumulate, st);
            if (db) mg_launch_slab_reduce(bslab, N, sstride, p.S, db, accumulate, st);
        }
        MG_CHECK_LAUNCH("mg_linear_wgrad_bf16/reduce");
        return MG_OK;
    }
    float* bslab = slab + (size_t)p.S * N * K;
    dim3 grid((unsigned)(p.tiles_n * p.tiles_k), (unsigned)p.S);
    if (p.narrow)
        hipLaunchKernelGGL((wgrad_bf16_kernel<32, 128, 1, 4>), grid, dim3(256), 0, st, dY, lddy, A, lda, rows, M, N, K, p.m_chunk, slab, db ? bslab : nullptr, p.tiles_k);
    else
        hipLaunchKernelGGL((wgrad_bf16_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, dY, lddy, A, lda, rows, M, N, K, p.m_chunk, slab, db ? bslab : nullptr, p.tiles_k);
    MG_CHECK_LAUNCH("mg_linear_wgrad_bf16/partial");
    mg_launch_slab_reduce(slab, nk, nk, p.S, dW, accumulate, st);
    MG_CHECK_LAUNCH("mg_linear_wgrad_bf16/reduce");
    if (db) {
        mg_launch_slab_reduce(bslab, N, N, p.S, db, accumulate, st);
        MG_CHECK_LAUNCH("mg_linear_wgrad_bf16/reduce_bias");
    }
    return MG_OK;
}

int mg_linear_wgrad_slabs_bf16(const uint16_t* dY, int lddy, const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K,
                               void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream) {
    MG_CHECK_ARG(dY && A && workspace && n_slabs && stride && M > 0 && N > 0 && K > 0, "mg_linear_wgrad_slabs_bf16: bad arguments (M=%lld N=%d K=%d)",
                 (long long)M, N, K);
    MG_CHECK_ARG(lddy >= N && lda >= K && lddy % 8 == 0 && lda % 8 == 0, "mg_linear_wgrad_slabs_bf16: lddy=%d lda=%d must be multiples of 8 covering N=%d / K=%d",
                 lddy, lda, N, K);
    MG_CHECK_ARG(al16(dY) && al16(A) && al16(workspace), "mg_linear_wgrad_slabs_bf16: buffers must be 16-byte aligned");
    int big_s = 0, big_chunk = 0;
    MG_CHECK_ARG(mg_wgrad_big_plan(M, N, K, lda, lddy, &big_s, &big_chunk) > 0,
                 "mg_linear_wgrad_slabs_bf16: M=%lld N=%d K=%d lda=%d is not a wide-tile shape (use mg_linear_wgrad_bf16)", (long long)M, N, K, lda);
    const int64_t nk = (int64_t)N * K, sstride = nk + N;
    if ((size_t)big_s * (size_t)sstride * sizeof(float) > workspace_bytes) {
        mg_set_error("mg_linear_wgrad_slabs_bf16: %d split slabs of %lld floats do not fit the %zu-byte workspace", big_s, (long long)sstride,
                     workspace_bytes);
        return MG_EWORKSPACE;
    }
    float* slab = (float*)workspace;
    mg_launch_wgrad_big(dY, lddy, A, lda, rows, M, N, K, big_s, big_chunk, slab, slab + nk, sstride, (hipStream_t)stream);
    MG_CHECK_LAUNCH("mg_linear_wgrad_slabs_bf16");
    *n_slabs = big_s;
    *stride = sstride;
    return MG_OK;
}

// mg_linear_wgrad_bf16 with BOTH operands gathered: dW = sum_m dY[dy_rows[m]]^T A[rows[m]] (rows == NULL: A[dy_rows[m]]), M = the number of
// index pairs.  The weight gradients of a recurrent layer over the valid frames of a ragged batch only (the reference packs them away:
// morgana/utils.py:366-385 pack_padded_sequence; here the recurrences write padded (B, T) arrays and the GEMM skips the padding).
// Wide-tile shapes with lda == 512 only (MG_EINVAL otherwise: the caller keeps the padded form).
int mg_linear_wgrad_rows_bf16(const uint16_t* dY, int lddy, const int32_t* dy_rows, const uint16_t* A, int lda, const int32_t* rows, int64_t M,
                              int N, int K, float* dW, float* db, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(dY && dy_rows && A && dW && M > 0 && N > 0 && K > 0, "mg_linear_wgrad_rows_bf16: bad arguments (M=%lld N=%d K=%d)", (long long)M, N, K);
    MG_CHECK_ARG(lddy >= N && lda >= K && lddy % 8 == 0 && lda % 8 == 0,
                 "mg_linear_wgrad_rows_bf16: lddy=%d lda=%d must be multiples of 8 covering N=%d / K=%d", lddy, lda, N, K);
    MG_CHECK_ARG(al16(dY) && al16(A), "mg_linear_wgrad_rows_bf16: buffers must be 16-byte aligned");
    int big_s = 0, big_chunk = 0;
    MG_CHECK_ARG(lda == 512 && mg_wgrad_big_plan(M, N, K, lda, lddy, &big_s, &big_chunk) > 0,
                 "mg_linear_wgrad_rows_bf16: M=%lld N=%d K=%d lda=%d is not a 512-wide wide-tile shape", (long long)M, N, K, lda);
    const int64_t nk = (int64_t)N * K, sstride = nk + N;
    if (!workspace || (size_t)big_s * (size_t)sstride * sizeof(float) > workspace_bytes) {
        mg_set_error("mg_linear_wgrad_rows_bf16: %d split slabs of %lld floats do not fit the %zu-byte workspace", big_s, (long long)sstride,
                     workspace_bytes);
        return MG_EWORKSPACE;
    }
    float* slab = (float*)workspace;
    float* bslab = slab + nk;
    hipStream_t st = (hipStream_t)stream;
    MG_CHECK_ARG(mg_launch_wgrad_big(dY, lddy, A, lda, rows ? rows : dy_rows, M, N, K, big_s, big_chunk, slab, db ? bslab : nullptr, sstride, st,
                                     dy_rows) > 0,
                 "mg_linear_wgrad_rows_bf16: M=%lld N=%d takes the half-width tiles, which have no gathered-dY form", (long long)M, N);
    MG_CHECK_LAUNCH("mg_linear_wgrad_rows_bf16/partial");
    if (db && db == dW + nk) {
        mg_launch_slab_reduce(slab, sstride, sstride, big_s, dW, accumulate, st);
    } else {
        mg_launch_slab_reduce(slab, nk, sstride, big_s, dW, accumulate, st);
        if (db) mg_launch_slab_reduce(bslab, N, sstride, big_s, db, accumulate, st);
    }
    MG_CHECK_LAUNCH("mg_linear_wgrad_rows_bf16/reduce");
    return MG_OK;
}

// Weight gradient (slabs, as mg_linear_wgrad_slabs_bf16 without a gather) and dX = (dY W) * A (1 - A) of ONE Linear + the Sigmoid
// below it, A being both the layer's input and that sigmoid's output: one grid for both where the shapes allow
// (mg_launch_wgrad_dgrad_pair), otherwise the two launches.
int mg_linear_wgrad_dgrad_bf16(const uint16_t* dY, int lddy, const uint16_t* A, int lda, int64_t M, int N, int K, const uint16_t* WT, int ldwt,
                               uint16_t* dX, int lddx, void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream) {
    MG_CHECK_ARG(dY && A && WT && dX && workspace && n_slabs && stride && M > 0 && N > 0 && K > 0,
                 "mg_linear_wgrad_dgrad_bf16: bad arguments (M=%lld N=%d K=%d)", (long long)M, N, K);
    MG_CHECK_ARG(lddy >= N && lda >= K && ldwt >= N && lddx >= K && lddy % 8 == 0 && lda % 8 == 0 && ldwt % 8 == 0 && lddx % 8 == 0,
                 "mg_linear_wgrad_dgrad_bf16: leading dimensions must be multiples of 8 and cover N=%d / K=%d (lddy=%d lda=%d ldwt=%d lddx=%d)", N, K,
                 lddy, lda, ldwt, lddx);
    MG_CHECK_ARG(al16(dY) && al16(A) && al16(WT) && al16(dX) && al16(workspace), "mg_linear_wgrad_dgrad_bf16: buffers must be 16-byte aligned");
    const int64_t nk = (int64_t)N * K, sstride = nk + N;
    int S = 0;
    if (mg_launch_wgrad_dgrad_pair(dY, lddy, A, lda, M, N, K, WT, ldwt, dX, lddx, (float*)workspace, sstride, workspace_bytes / sizeof(float), &S,
                                   (hipStream_t)stream, nullptr) > 0) {
        MG_CHECK_LAUNCH("mg_linear_wgrad_dgrad_bf16/pair");
        *n_slabs = S;
        *stride = sstride;
        return MG_OK;
    }
    const int rc = mg_linear_wgrad_slabs_bf16(dY, lddy, A, lda, nullptr, M, N, K, workspace, workspace_bytes, n_slabs, stride, stream);
    if (rc != MG_OK) return rc;
    return mg_linear_dgrad_bf16(dY, lddy, M, N, WT, ldwt, K, A, lda, dX, lddx, 0, stream);
}

// mg_linear_wgrad_dgrad_bf16 with mg_expand_column_reduce_f32 riding at the end of its grid (the repeated prediction and the ordered
// sum of the fused tail's slabs: two jobs of the step's forward that nothing reads before the update); where the one-grid form does
// not take the shape, the launches one after the other.  The same results as the separate calls, bit for bit.
int mg_linear_wgrad_dgrad_expand_bf16(const uint16_t* dY, int lddy, const uint16_t* A, int lda, int64_t M, int N, int K, const uint16_t* WT,
                                      int ldwt, uint16_t* dX, int lddx, void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride,
                                      const float* table, const int32_t* rows, int64_t frames, float* out, const void* stats_workspace, int R,
                                      int extra, const float* tail_slab, int64_t tail_n, int64_t tail_stride, int tail_S, float* tail_dst,
                                      int loss_only, void* stream) {
    MG_CHECK_ARG(dY && A && WT && dX && workspace && n_slabs && stride && M > 0 && N > 0 && K > 0,
                 "mg_linear_wgrad_dgrad_expand_bf16: bad arguments (M=%lld N=%d K=%d)", (long long)M, N, K);
    MG_CHECK_ARG(lddy >= N && lda >= K && ldwt >= N && lddx >= K && lddy % 8 == 0 && lda % 8 == 0 && ldwt % 8 == 0 && lddx % 8 == 0,
                 "mg_linear_wgrad_dgrad_expand_bf16: leading dimensions must be multiples of 8 and cover N=%d / K=%d (lddy=%d lda=%d ldwt=%d lddx=%d)",
                 N, K, lddy, lda, ldwt, lddx);
    MG_CHECK_ARG(al16(dY) && al16(A) && al16(WT) && al16(dX) && al16(workspace), "mg_linear_wgrad_dgrad_expand_bf16: buffers must be 16-byte aligned");
    MG_CHECK_ARG(table && rows && out && stats_workspace && tail_slab && tail_dst && frames > 0 && R > 0 && extra >= 0 && tail_n > 0 &&
                 tail_stride >= tail_n && tail_S > 0,
                 "mg_linear_wgrad_dgrad_expand_bf16: bad rider arguments (frames=%lld n=%lld S=%d)", (long long)frames, (long long)tail_n, tail_S);
    const int64_t nk = (int64_t)N * K, sstride = nk + N;
    const ExpandReduceArgs xr{table, rows, frames, out, (const float*)stats_workspace, (int)(mg_ceil_div(R, 16) + mg_ceil_div(extra, 4)),
                              tail_slab, tail_n, tail_stride, tail_S, tail_dst, loss_only ? (tail_n - 1) / 16 : 0};
    int S = 0;
    if (mg_launch_wgrad_dgrad_pair(dY, lddy, A, lda, M, N, K, WT, ldwt, dX, lddx, (float*)workspace, sstride, workspace_bytes / sizeof(float), &S,
                                   (hipStream_t)stream, &xr) > 0) {
        MG_CHECK_LAUNCH("mg_linear_wgrad_dgrad_expand_bf16/pair");
        *n_slabs = S;
        *stride = sstride;
        return MG_OK;
    }
    // (loss_only: the separate launch sums every element - more than asked for, the same values where both write)
    int rc = mg_expand_column_reduce_f32(table, rows, frames, out, stats_workspace, R, extra, tail_slab, tail_n, tail_stride, tail_S, tail_dst, stream);
    if (rc != MG_OK) return rc;
    return mg_linear_wgrad_dgrad_bf16(dY, lddy, A, lda, M, N, K, WT, ldwt, dX, lddx, workspace, workspace_bytes, n_slabs, stride, stream);
}

// mg_phone_front (frame map + per-phone loss statistics) and mg_linear_fwd_bf16 of the phone table's first layer - two launches that
// read nothing of each other - as ONE grid where the GEMM leaves CUs idle (mg_launch_phone_front_gemm), otherwise one after the other.
int mg_phone_front_linear_fwd_bf16(const int64_t* dur, int B, int P, int T, const float* target, const int64_t* seq_len, int extra,
                                   int32_t* rows32, int32_t* rows_mapped, int pad_row, int32_t* seg_start, int32_t* seg_end, float* ybar,
                                   float* weight, void* workspace, size_t workspace_bytes, const uint16_t* A, int lda, int64_t M, int K,
                                   const uint16_t* W, int ldw, const float* bias, int N, uint16_t* Y, int ldy, int act, void* stream) {
    int rc = mg_phone_front_check(dur, B, P, T, target, extra, rows32, rows_mapped, seg_start, seg_end, ybar, weight, workspace, workspace_bytes,
                                  "mg_phone_front_linear_fwd_bf16");
    if (rc != MG_OK) return rc;
    MG_CHECK_ARG(A && W && Y && M > 0 && N > 0 && K > 0 && (act == MG_ACT_NONE || act == MG_ACT_SIGMOID),
                 "mg_phone_front_linear_fwd_bf16: bad GEMM arguments (M=%lld N=%d K=%d act=%d)", (long long)M, N, K, act);
    PhoneFrontArgs pf{dur, target, seq_len, B, P, T, extra, rows32, rows_mapped, pad_row, seg_start, seg_end, ybar, weight, (float*)workspace, 0, 0};
    if (bias && lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 &&
        mg_launch_phone_front_gemm(pf, A, lda, M, K, W, ldw, N, bias, Y, ldy, act == MG_ACT_SIGMOID ? EPI_BIAS_SIGMOID : EPI_BIAS,
                                   (hipStream_t)stream) > 0) {
        MG_CHECK_LAUNCH("mg_phone_front_linear_fwd_bf16/one grid");
        return MG_OK;
    }
    rc = mg_phone_front(dur, B, P, T, target, seq_len, extra, rows32, rows_mapped, pad_row, seg_start, seg_end, ybar, weight, workspace,
                        workspace_bytes, stream);
    if (rc != MG_OK) return rc;
    return mg_linear_fwd_bf16(A, lda, nullptr, M, K, W, ldw, bias, N, Y, ldy, 0, act, stream);
}

// ---- pair planes: the fused step of precision mode 'bf16x3' (include/morgana_hip.h; kernels in gemm_bf16_big.hip) -------------------
int mg_phone_front_linear_fwd_x3(const int64_t* dur, int B, int P, int T, const float* target, const int64_t* seq_len, int extra,
                                 int32_t* rows32, int32_t* rows_mapped, int pad_row, int32_t* seg_start, int32_t* seg_end, float* ybar,
                                 float* weight, void* workspace, size_t workspace_bytes, const uint16_t* A, int lda, int64_t M, int K,
                                 const uint16_t* W, int ldw, const float* bias, int N, uint16_t* Y, int ldy, int act, void* stream) {
    MG_CHECK_ARG(A && W && Y && bias && M > 0 && N > 0 && K > 0 && (act == MG_ACT_NONE || act == MG_ACT_SIGMOID),
                 "mg_phone_front_linear_fwd_x3: bad GEMM arguments (M=%lld N=%d K=%d act=%d)", (long long)M, N, K, act);
    MG_CHECK_ARG(lda % 128 == 0 && ldw % 128 == 0 && lda / 2 >= K && ldw / 2 >= K && ldy == 2 * N && N % 256 == 0 && al16(A) && al16(W) && al16(Y),
                 "mg_phone_front_linear_fwd_x3: pair planes need lda=%d, ldw=%d multiples of 128 with planes >= K=%d, ldy=%d == 2 N, N=%d a multiple of 256, 16-byte aligned buffers",
                 lda, ldw, K, ldy, N);
    PhoneFrontArgs pf{};
    if (dur) {
        const int rc = mg_phone_front_check(dur, B, P, T, target, extra, rows32, rows_mapped, seg_start, seg_end, ybar, weight, workspace,
                                            workspace_bytes, "mg_phone_front_linear_fwd_x3");
        if (rc != MG_OK) return rc;
        pf = PhoneFrontArgs{dur, target, seq_len, B, P, T, extra, rows32, rows_mapped, pad_row, seg_start, seg_end, ybar, weight, (float*)workspace, 0, 0};
    }
    // does the front ride in the GEMM's grid?  Asked first (nothing is launched when it does not): the front must precede nothing of
    // the GEMM, but a launch order of "front, then GEMM" keeps the two-launch form the one of mg_phone_front_linear_fwd_bf16
    int rode = 0;
    const int epi = act == MG_ACT_SIGMOID ? EPI_BIAS_SIGMOID : EPI_BIAS;
    const int launched = mg_launch_nt_persist_x3(dur ? &pf : nullptr, A, lda, M, K, W, ldw, N, bias, Y, ldy, epi, (hipStream_t)stream, &rode);
    if (launched <= 0) {
        mg_set_error("mg_phone_front_linear_fwd_x3: shape outside the wide-tile kernel (M=%lld N=%d K=%d lda=%d ldw=%d)", (long long)M, N, K, lda, ldw);
        return MG_EINVAL;
    }
    MG_CHECK_LAUNCH("mg_phone_front_linear_fwd_x3");
    if (dur && !rode)
        return mg_phone_front(dur, B, P, T, target, seq_len, extra, rows32, rows_mapped, pad_row, seg_start, seg_end, ybar, weight, workspace,
                              workspace_bytes, stream);
    return MG_OK;
}

int mg_linear_fwd_x3_f32(const uint16_t* A, int lda, int64_t M, int K, const uint16_t* W, int ldw, const float* bias, int N, float* Y,
                         int ldy, int act, int parts, void* stream) {
    MG_CHECK_ARG(A && W && Y && M > 0 && N > 0 && K > 0 && (act == MG_ACT_NONE || act == MG_ACT_SIGMOID),
                 "mg_linear_fwd_x3_f32: bad arguments (M=%lld N=%d K=%d act=%d)", (long long)M, N, K, act);
    MG_CHECK_ARG(parts == 1 || (parts == 3 && act == MG_ACT_NONE), "mg_linear_fwd_x3_f32: parts=%d (1, or 3 without an activation)", parts);
    MG_CHECK_ARG(lda % 128 == 0 && ldw % 128 == 0 && lda / 2 >= K && ldw / 2 >= K && ldy == N && N % 128 == 0 && al16(A) && al16(W) && al16(Y),
                 "mg_linear_fwd_x3_f32: pair planes need lda=%d, ldw=%d multiples of 128 with planes >= K=%d, ldy=%d == N, N=%d a multiple of 128, 16-byte aligned buffers",
                 lda, ldw, K, ldy, N);
    if (mg_launch_nt_big_x3_f32(A, lda, M, K, W, ldw, N, bias, Y, ldy, act == MG_ACT_SIGMOID ? EPI_BIAS_SIGMOID : EPI_BIAS, (hipStream_t)stream, parts) <= 0) {
        mg_set_error("mg_linear_fwd_x3_f32: shape outside the wide-tile kernel (M=%lld N=%d K=%d lda=%d ldw=%d)", (long long)M, N, K, lda, ldw);
        return MG_EINVAL;
    }
    MG_CHECK_LAUNCH("mg_linear_fwd_x3_f32");
    return MG_OK;
}

size_t mg_linear_wgrad_dgrad_x3_colsum_floats(int64_t M, int K) { return M > 0 && K > 0 ? (size_t)(2 * mg_ceil_div(M, 256)) * (size_t)K : 0; }

int mg_linear_wgrad_dgrad_x3(const uint16_t* dY, int lddy, const uint16_t* A, int lda, int64_t M, int N, int K, const uint16_t* WT, int ldwt,
                             uint16_t* dX, int lddx, void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride, float* colsum,
                             size_t colsum_floats, int* n_colsum, void* stream) {
    MG_CHECK_ARG(dY && A && WT && dX && workspace && n_slabs && stride && colsum && n_colsum && M > 0 && N > 0 && K > 0,
                 "mg_linear_wgrad_dgrad_x3: bad arguments (M=%lld N=%d K=%d)", (long long)M, N, K);
    MG_CHECK_ARG(lddy % 128 == 0 && lda % 128 == 0 && ldwt % 128 == 0 && lddy / 2 >= N && lda / 2 >= K && ldwt / 2 >= N && lddx == 2 * K,
                 "mg_linear_wgrad_dgrad_x3: pair planes need lddy=%d lda=%d ldwt=%d multiples of 128 covering N=%d / K=%d and lddx=%d == 2 K", lddy, lda,
                 ldwt, N, K, lddx);
    MG_CHECK_ARG(al16(dY) && al16(A) && al16(WT) && al16(dX) && al16(workspace) && al16(colsum), "mg_linear_wgrad_dgrad_x3: buffers must be 16-byte aligned");
    MG_CHECK_ARG(colsum_floats >= mg_linear_wgrad_dgrad_x3_colsum_floats(M, K), "mg_linear_wgrad_dgrad_x3: colsum holds %zu floats, %zu needed",
                 colsum_floats, mg_linear_wgrad_dgrad_x3_colsum_floats(M, K));
    const int64_t nk = (int64_t)N * K, sstride = nk + N;
    int S = 0;
    if (mg_launch_wgrad_dgrad_pair(dY, lddy, A, lda, M, N, K, WT, ldwt, dX, lddx, (float*)workspace, sstride, workspace_bytes / sizeof(float), &S,
                                   (hipStream_t)stream, nullptr, 1, colsum) <= 0) {
        mg_set_error("mg_linear_wgrad_dgrad_x3: M=%lld N=%d K=%d is not the 512 -> 128 pair shape (or the workspace of %zu bytes is too small)",
                     (long long)M, N, K, workspace_bytes);
        return MG_EINVAL;
    }
    MG_CHECK_LAUNCH("mg_linear_wgrad_dgrad_x3");
    *n_slabs = S;
    *stride = sstride;
    *n_colsum = (int)(2 * mg_ceil_div(M, 256));
    return MG_OK;
}

int mg_linear_wgrad_slabs_x3(const uint16_t* dY, int lddy, const uint16_t* A, int lda, int64_t M, int N, int K, void* workspace,
                             size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream) {
    MG_CHECK_ARG(dY && A && workspace && n_slabs && stride && M > 0 && N > 0 && K > 0, "mg_linear_wgrad_slabs_x3: bad arguments (M=%lld N=%d K=%d)",
                 (long long)M, N, K);
    MG_CHECK_ARG(lddy % 128 == 0 && lda % 128 == 0 && lddy / 2 >= N && lda / 2 >= K, "mg_linear_wgrad_slabs_x3: pair planes need lddy=%d lda=%d multiples of 128 covering N=%d / K=%d",
                 lddy, lda, N, K);
    MG_CHECK_ARG(al16(dY) && al16(A) && al16(workspace), "mg_linear_wgrad_slabs_x3: buffers must be 16-byte aligned");
    int big_s = 0, big_chunk = 0;
    MG_CHECK_ARG(mg_wgrad_big_plan(M, N, K, lda / 2, lddy / 2, &big_s, &big_chunk) > 0,
                 "mg_linear_wgrad_slabs_x3: M=%lld N=%d K=%d plane=%d is not a wide-tile shape", (long long)M, N, K, lda / 2);
    const int64_t nk = (int64_t)N * K, sstride = nk + N;
    if ((size_t)big_s * (size_t)sstride * sizeof(float) > workspace_bytes) {
        mg_set_error("mg_linear_wgrad_slabs_x3: %d split slabs of %lld floats do not fit the %zu-byte workspace", big_s, (long long)sstride, workspace_bytes);
        return MG_EWORKSPACE;
    }
    MG_CHECK_ARG(mg_launch_wgrad_big(dY, lddy, A, lda, nullptr, M, N, K, big_s, big_chunk, (float*)workspace, nullptr, sstride, (hipStream_t)stream, nullptr, 1) > 0,
                 "mg_linear_wgrad_slabs_x3: no pair-plane tile program for M=%lld N=%d K=%d", (long long)M, N, K);
    MG_CHECK_LAUNCH("mg_linear_wgrad_slabs_x3");
    *n_slabs = big_s;
    *stride = sstride;
    return MG_OK;
}

// dst[0 .. count) (+)= the ordered sum of n_slabs slabs (stride floats apart): the library's slab reduce as a launch of its own, for a
// caller that took split-M slabs (mg_linear_wgrad_slabs_bf16, mg_linear_wgrad_dgrad_bf16) and needs the finished gradient before
// the update - a rank of a data-parallel job, whose all-reduce comes first.  A slab is [N*K weights | N bias sums], so with the bias
// gradient stored right behind the weight gradient one launch finishes both.  Bit for bit mg_linear_wgrad_bf16's own reduce.
int mg_slab_reduce_f32(const float* slab, int n_slabs, int64_t stride, int64_t count, float* dst, int accumulate, void* stream) {
    MG_CHECK_ARG(slab && dst && n_slabs >= 1 && count > 0 && stride >= count, "mg_slab_reduce_f32: bad arguments (n_slabs=%d stride=%lld count=%lld)",
                 n_slabs, (long long)stride, (long long)count);
    mg_launch_slab_reduce(slab, count, stride, n_slabs, dst, accumulate, (hipStream_t)stream);
    MG_CHECK_LAUNCH("mg_slab_reduce_f32");
    return MG_OK;
}

}  // extern "C"
