// K2 (bf16 throughput mode), large-tile kernels for the shapes that dominate the step:
//   gemm_nt_big   C[M,N] = epi(A'[M,K] B[N,K]^T)    256 x {256,128} tiles, BK = 32, 4 LDS stages   (forward, dgrad)
//   wgrad_big     dW[N,K] = dY[M,N]^T A'[M,K]        128 x {640,512} tiles over split M, 3 stages   (weight gradients)
// Reference: the nn.Linear + nn.Sigmoid stack of README.rst:65-73 (morgana/utils.py:401-418) and its backward.
//
// Why these shapes (measured on MI355X with the 128 x 128 kernels of gemm_bf16.hip, profiles/):
//   * a 128 x 128 x 32 tile moves 16 KB through L2->LDS per 1 MFLOP, a 256 x 256 tile half of that per FLOP;
//   * the 128 x 128 wgrad re-read dZ1 (262 MB) once per 128-wide K tile: FETCH_SIZE 0.71 GB (x2 on gfx950) per launch.
//     A 128 x 640 tile covers all of K, so dY is read exactly once; X (the gathered phone rows) comes from L2/MALL.
// Mechanics (cdna_hip_programming.md section 5): tiles are filled by global_load_lds_dwordx4 (LDS-DMA: no VGPR staging,
// no ds_write), several LDS stages, the next stages' DMA stays in flight across a raw s_barrier behind a COUNTED
// s_waitcnt vmcnt(N) (one barrier per stage); the LDS image is lane-linear, so the bank swizzle is applied to the per-lane
// SOURCE address and again on the fragment reads (the same involution on both sides).  The per-lane source address is
// also what fuses the upsample gather: a lane simply points at its phone row (or at a zero row for the -1 pad index).
// 512 threads = 8 waves, one workgroup per CU (2 waves per SIMD).
// Tried and dropped (all within +-8 % on the layer-1 forward shape): 128 x 256 tiles x 2 workgroups per CU,
// 128 x 128 x 3 per CU, 3 instead of 4 stages, two 64-deep stages, XCD-grouped wgrad block order, non-temporal streams.
#include "common.h"
#include "phone_front.h"
#include "expand_reduce.h"

#include <type_traits>

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfv4 __attribute__((ext_vector_type(4)));

#define EPI_BIAS 0
#define EPI_BIAS_SIGMOID 1
#define EPI_SIGMOID_GRAD 2

#define MG_ZERO_ELEMS 16384
__device__ uint16_t g_zero_row[MG_ZERO_ELEMS];   // zero-initialised: source of pad rows / out-of-range rows

// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes land at LDS byte (lds_wave_base + 16 l).  Issued from inline asm on
// purpose: hipcc then keeps no record of a pending LDS write, so it neither drains vmcnt in front of the fragment reads
// (it does for ds_read_b64_tr_b16 behind the builtin form: one s_waitcnt vmcnt(0) per step, measured) nor in front of
// __syncthreads; completion is tracked by the counted waits of WAIT_VM_BARRIER below.  M0 is saved and restored inside
// the statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16(const uint16_t* src, unsigned char* lds_wave_base) {
    const unsigned lds_off = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)lds_wave_base);
    const unsigned lds_uni = __builtin_amdgcn_readfirstlane(lds_off);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_uni)
                 : "memory");
}


#define WAIT_VM_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ")\n\ts_barrier" ::: "memory")
#define WAIT_LGKM_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---------------------------------------------------------------------------------------------------------------------
// gemm_nt_big: BM = 256, BN in {256, 128}, BK = 32, FOUR LDS stages (three tiles of LDS-DMA in flight, one barrier per
// stage).  LDS rows are 64 bytes (4 chunks of 16 B); chunk c of tile row r is stored at chunk position c ^ ((r>>2)&3):
// conflict free for the 4 x 16 lane groups of ds_read_b128.
// The MFMA is issued with the operands swapped (weights as A, frames as B), so the accumulator tile is C^T: the lane
// holds one output ROW (frame) and its registers hold 4-column groups of it.  The epilogue therefore runs entirely in
// registers (bias / sigmoid / sigmoid-grad per register, bf16 pack, v_permlane32_swap to widen to 16-byte stores): no
// LDS staging and no per-element index arithmetic (the first version spent ~45 % of its cycles there, profiles/r1).
// Requires lda, ldb multiples of 64 (zero padded), N % BN == 0, ldc == N, (SIGMOID_GRAD) ldh >= N.
// ---------------------------------------------------------------------------------------------------------------------
MG_STAMP_DECL(g_stamps_nt);

// The tile program is a device function on a caller-provided LDS block and block id, so that two independent launches can share one
// grid (wgrad_dgrad_pair_kernel below); gemm_nt_big_kernel is the plain launch of it.
#define NT_BIG_LDS(BN_) (((BN_) == 256 ? 4 : 6) * (256 * 64 + (BN_) * 64))
// X3 (precision mode 'bf16x3', fused step - see the section "pair planes" at the end of this file): both operands are PAIR PLANES
// [hi | lo] (row stride lda / ldb, planes lda / 2 / ldb / 2 apart; K = the contraction length of ONE plane), and the k loop runs three
// times over the plane: (A hi, B hi), (A hi, B lo), (A lo, B hi) - three bf16 products into one fp32 accumulator per fp32 product.
// EPI_SIGMOID_GRAD then reads H as pair planes too (h = hi + lo), writes dX as pair planes (ldc = row stride, planes ldc / 2 apart)
// and leaves the column sums of the fp32 dX (the bias gradient of the layer below) in `colsum`: one 64-column strip per wave,
// slab (tile_m * WAVES_M + wave row) of `N` floats.
template <int BN, int EPI, int X3 = 0>
__device__ __forceinline__ void gemm_nt_big_body(unsigned char* __restrict__ smem, const unsigned block_id, const uint16_t* __restrict__ A,
                                                 int lda, const int32_t* __restrict__ rows, int64_t M, int K, const uint16_t* __restrict__ Bm,
                                                 int ldb, int N, const float* __restrict__ bias_in, const uint16_t* __restrict__ H, int ldh,
                                                 const int32_t* __restrict__ h_rows, void* __restrict__ Cv_in, int ldc, int tiles_m, int tiles_n,
                                                 int c_f32, float* __restrict__ colsum = nullptr) {
    const float* bias = bias_in;
    void* Cv = Cv_in;
    constexpr int BM = 256;
    constexpr int WAVES_N = BN / 64;              // 4 or 2
    constexpr int WAVES_M = 8 / WAVES_N;          // 2 or 4
    constexpr int WM = BM / WAVES_M;              // 128 or 64
    constexpr int TM = WM / 32, TN = 2;
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int GA = BM / 16 / 8;               // 1 KB pieces (16 rows) per wave for A: 2
    constexpr int GB = BN / 16 / 8;               // for B: 2 or 1
    constexpr int NL = GA + GB;                   // LDS-DMA instructions per wave per stage: 4 or 3
    // Stages: 4 x 32 KB for the square tile; 6 x 24 KB for the 128-wide one, whose shapes (K >> N: layer-2 forward) stream A
    // from HBM - five tiles in flight per CU cover the HBM latency at ~6 TB/s, three do not (87 -> see DESIGN.md).
    constexpr int NS = (BN == 256) ? 4 : 6;

    static_assert(NS * STAGE == NT_BIG_LDS(BN), "LDS size helper");

    // XCD-aware tile order: blocks b and b + 8 share an XCD (its L2); give them the N tiles of ONE M tile so the A rows
    // and the H tile are fetched into that L2 once.
    // X3 == 2 (fp32 output, EPI_BIAS): the three passes as three SETS of workgroups - the grid is three times the tiles, set p runs
    // pass p over the plane and writes its partial sums to C + p * M * ldc (the bias rides in set 0); the consumer adds the three
    // partial results.  A few-tile GEMM (the 128-wide layer of the phone table: 84 tiles on 256 CUs) then fills the chip with chains a
    // third as long.
    unsigned tile_id = block_id;
    int my_pass = 0;
    if (X3 == 2) {
        const unsigned per_pass = (unsigned)((tiles_m + 7) / 8) * 8u * (unsigned)tiles_n;
        my_pass = (int)(block_id / per_pass);
        tile_id = block_id - (unsigned)my_pass * per_pass;
    }
    const int xcd = tile_id & 7, jj = tile_id >> 3;
    const int tile_n = jj % tiles_n;
    const int tile_m = (jj / tiles_n) * 8 + xcd;
    if (tile_m >= tiles_m) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MG_STAMPS
    unsigned long long ts0, ts1 = 0, ts2, ts3, tr0, tr1, ta, tb, sum_wait = 0, sum_issue = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * 64;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    const int n_kp = (K + 31) / 32;               // stages that hold real columns (lda, ldb >= 64 * ceil(K / 64))
    const int n_kt = X3 == 1 ? 3 * n_kp : n_kp;   // X3: three passes over the plane (X3 == 2: this workgroup's one)
    const int a_lo = lda >> 1, b_lo = ldb >> 1;   // X3: element offset of an operand's lo plane
    if (X3 == 2) {
        if (my_pass > 0) bias = nullptr;
        Cv = reinterpret_cast<float*>(Cv) + (size_t)my_pass * (size_t)M * ldc;
    }

    // Per-lane DMA sources.  Piece g covers tile rows 16g..16g+15; lane l writes row 16g + (l>>2), chunk position l&3,
    // and therefore fetches source chunk (l&3) ^ ((row>>2)&3) of that row.
    const uint16_t* asrc[GA];
    const uint16_t* bsrc[GB];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int row = (wave * GA + i) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
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
        const int row = (wave * GB + i) * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        const int n = n0 + row;
        bsrc[i] = (n < N ? Bm + (size_t)n * ldb : g_zero_row) + c * 8;
    }

    auto issue = [&](int kt) {
        unsigned char* st = smem + (kt % NS) * STAGE;
        int ka = kt * 32, kb = kt * 32;
        if (X3) {                                  // pass 0: (hi, hi), 1: (hi, lo), 2: (lo, hi)
            // (pass fastest: the second read of an operand's k-tile follows the first by a stage or two and is served by the XCD's L2)
            const int pass = X3 == 2 ? my_pass : kt % 3, w = (X3 == 2 ? kt : kt / 3) * 32;
            ka = w + (pass == 2 ? a_lo : 0);
            kb = w + (pass == 1 ? b_lo : 0);
        }
#pragma unroll
        for (int i = 0; i < GA; ++i) glds16(asrc[i] + ka, st + (wave * GA + i) * 1024);
#pragma unroll
        for (int i = 0; i < GB; ++i) glds16(bsrc[i] + kb, st + A_BYTES + (wave * GB + i) * 1024);
    };

    // acc[i][j] holds C^T of the (i, j) 32 x 32 sub-tile: lane&31 = frame row, registers = output columns.
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    // fragment byte offsets inside a stage for k-step 0; k-step 1 is the same offset XOR 32
    int aoff[TM], boff[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm0 + i * 32 + lr;
        aoff[i] = row * 64 + ((lh ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn0 + j * 32 + lr;
        boff[j] = A_BYTES + row * 64 + ((lh ^ ((row >> 2) & 3)) << 4);
    }

#pragma unroll
    for (int p = 0; p < NS - 1; ++p)
        if (p < n_kt) issue(p);
    for (int kt = 0; kt < n_kt; ++kt) {
        MG_STAMP(ta);
        // tiles issued so far: min(n_kt, kt + NS - 1); tile kt must have landed: min(n_kt - 1 - kt, NS - 2) tiles may stay
        // in flight, NL LDS-DMA instructions each (vmcnt takes an immediate, hence the ladder)
        {
            const int fly = min(n_kt - 1 - kt, NS - 2);
            if (NL == 4) {
                if (fly >= 2) WAIT_VM_BARRIER(8); else if (fly == 1) WAIT_VM_BARRIER(4); else WAIT_VM_BARRIER(0);
            } else if (NS == 4) {
                if (fly >= 2) WAIT_VM_BARRIER(6); else if (fly == 1) WAIT_VM_BARRIER(3); else WAIT_VM_BARRIER(0);
            } else {
                if (fly >= 4) WAIT_VM_BARRIER(12); else if (fly == 3) WAIT_VM_BARRIER(9); else if (fly == 2) WAIT_VM_BARRIER(6);
                else if (fly == 1) WAIT_VM_BARRIER(3); else WAIT_VM_BARRIER(0);
            }
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_wait, tb, ta);
#ifdef MG_STAMPS
        if (kt == 0) ts1 = tb;
#endif
        // the stage refilled here is the one every wave finished reading before this barrier
        if (kt + NS - 1 < n_kt) issue(kt + NS - 1);
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_issue, ta, tb);
        const unsigned char* st = smem + (kt % NS) * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bfv8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bfv8*>(st + (aoff[i] ^ (ks << 5)));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bfv8*>(st + (boff[j] ^ (ks << 5)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (EPI == EPI_SIGMOID_GRAD) acc[i][j] = mg_mfma_32x32x16(a[i], b[j], acc[i][j]);
                    else acc[i][j] = mg_mfma_32x32x16(b[j], a[i], acc[i][j]);
                }
        }
    }

#ifdef MG_STAMPS
    MG_STAMP(ts2);
    // the diagnostic build ends with the epilogue's stores drained, so that its stamp covers them
#endif
    if (EPI == EPI_SIGMOID_GRAD) {
        // Memory-bound variant (dX = (dY W) * H (1 - H), K small): plain C layout, fp32 sub-tiles staged through LDS so that
        // H is read and dX written as whole 16-byte row pieces (8 lanes per 128-byte row segment).
        constexpr int STG_LD = 68;
        WAIT_LGKM_BARRIER();                                   // all waves are done with the tile stages
        float* stg = reinterpret_cast<float*>(smem) + wave * 32 * STG_LD;
        float csum[8];                             // X3: this lane's share of the column sums of dX (its 8 columns, its rows)
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lh) * STG_LD + j * 32 + lr] = acc[i][j][r];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl = it * 8 + (lane >> 3);
                const int cl = (lane & 7) * 8;
                const int64_t row = m0 + wm0 + i * 32 + rl;
                const int col = n0 + wn0 + cl;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(&stg[rl * STG_LD + cl]);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(&stg[rl * STG_LD + cl + 4]);
                if (row >= M) continue;
                // H row: the output row itself, or (phone-rate first layer) the table row its frame was gathered from
                const int64_t hrow = h_rows ? (int64_t)h_rows[row] : row;
                const bfv8 hv = *reinterpret_cast<const bfv8*>(H + (size_t)hrow * ldh + col);
                float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (X3) {
                    // pair planes: h = hi + lo (exact in fp32), dX split again; the bias gradient of the layer below = the column
                    // sums of the fp32 values, in a fixed order (row blocks i, it ascending per lane; lanes and waves below)
                    const bfv8 hl = *reinterpret_cast<const bfv8*>(H + (size_t)hrow * ldh + (ldh >> 1) + col);
                    bfv8 o_hi, o_lo;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
#pragma clang fp contract(off)
                        const float h = (float)hv[e] + (float)hl[e];
                        const float x = v[e] * h * (1.f - h);
                        csum[e] += x;
                        o_hi[e] = (__bf16)x;
                        o_lo[e] = (__bf16)(x - (float)o_hi[e]);
                    }
                    uint16_t* crow = reinterpret_cast<uint16_t*>(Cv) + (size_t)row * ldc + col;
                    *reinterpret_cast<bfv8*>(crow) = o_hi;
                    *reinterpret_cast<bfv8*>(crow + (ldc >> 1)) = o_lo;
                    continue;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float h = (float)hv[e];
                    v[e] = v[e] * h * (1.f - h);
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
        if (X3 && colsum) {
            // the eight row lanes of a column chunk (lane >> 3), added in a fixed butterfly; one slab per (M tile, wave row)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                csum[e] += __shfl_xor(csum[e], 8, 64);
                csum[e] += __shfl_xor(csum[e], 16, 64);
                csum[e] += __shfl_xor(csum[e], 32, 64);
            }
            if (lane < 8) {
                float* dst = colsum + ((size_t)tile_m * WAVES_M + (wave / WAVES_N)) * N + n0 + wn0 + lane * 8;
                *reinterpret_cast<f32x4*>(dst) = f32x4{csum[0], csum[1], csum[2], csum[3]};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{csum[4], csum[5], csum[6], csum[7]};
            }
        }
        return;
    }

    // Accumulator layout of sub-tile (i, j): this lane's frame row m = m0 + wm0 + 32 i + (lane & 31); register 4g + e is
    // column n0 + wn0 + 32 j + 8 g + 4 (lane >> 5) + e.
    float bv[TN][16];
    if (EPI != EPI_SIGMOID_GRAD) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[j][4 * g + e] = bias ? bias[n0 + wn0 + j * 32 + 8 * g + 4 * lh + e] : 0.f;
    }
    if (!c_f32) {
        // bf16 output: rows leave through a wave-private LDS patch so that a store instruction writes whole 128-byte row
        // segments (8 lanes x 16 B).  Stored straight from the accumulator layout (lane = row) every lane of a store hits a
        // different row: 16-byte transactions, measured 16-25k cycles per 256 x 256 tile against ~33k for its whole main loop.
        constexpr int SP = 144;                                // patch row pitch: 64 columns x 2 B + 16 B of bank spread
        WAIT_LGKM_BARRIER();                                   // every wave is done with the tile stages
        unsigned char* patch = smem + wave * (32 * SP);
        const int prow = lane >> 3, pchunk = lane & 7;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[i][j][4 * g + e] + bv[j][4 * g + e];
                        if (EPI == EPI_BIAS_SIGMOID) x = mg_sigmoid_fast(x);
                        v[e] = x;
                    }
                    typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
                    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
                    const u32x2_t pk = u32x2_t{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                               __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
                    *reinterpret_cast<u32x2_t*>(patch + lr * SP + (j * 32 + 8 * g + 4 * lh) * 2) = pk;
                }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                const int rl = it * 8 + prow;
                const u32x4_t o = *reinterpret_cast<const u32x4_t*>(patch + rl * SP + pchunk * 16);
                const int64_t m = m0 + wm0 + i * 32 + rl;
                if (m < M) *reinterpret_cast<u32x4_t*>(reinterpret_cast<uint16_t*>(Cv) + (size_t)m * ldc + n0 + wn0 + pchunk * 8) = o;
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t m = m0 + wm0 + i * 32 + lr;
        const bool live = m < M;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            // fp32 output (a wide last layer): straight from the registers, 16 bytes per lane and 4-column group
            if (!live) continue;
            float* crow = reinterpret_cast<float*>(Cv) + (size_t)m * ldc + n0 + wn0 + j * 32 + 4 * lh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[i][j][4 * g + e] + bv[j][4 * g + e];
                    if (EPI == EPI_BIAS_SIGMOID) x = mg_sigmoid_fast(x);
                    v[e] = x;
                }
                *reinterpret_cast<f32x4*>(crow + 8 * g) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
    }
#ifdef MG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG_STAMP(ts3);
    MG_STAMP_REAL(tr1);
    const int sb = block_id;
    MG_STAMP_STORE(g_stamps_nt, sb, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_nt, sb, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_nt, sb, wave, lane, 2, ts2);
    MG_STAMP_STORE(g_stamps_nt, sb, wave, lane, 3, ts3);
    MG_STAMP_STORE(g_stamps_nt, sb, wave, lane, 4, tr0);
    MG_STAMP_STORE(g_stamps_nt, sb, wave, lane, 5, tr1);
    MG_STAMP_STORE(g_stamps_nt, sb, wave, lane, 6, sum_wait);
    MG_STAMP_STORE(g_stamps_nt, sb, wave, lane, 7, sum_issue);
#endif
}

template <int BN, int EPI, int X3 = 0>
__global__ __launch_bounds__(512) void gemm_nt_big_kernel(const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                          int64_t M, int K, const uint16_t* __restrict__ Bm, int ldb, int N,
                                                          const float* __restrict__ bias, const uint16_t* __restrict__ H, int ldh,
                                                          const int32_t* __restrict__ h_rows, void* __restrict__ Cv, int ldc, int tiles_m,
                                                          int tiles_n, int c_f32) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[NT_BIG_LDS(BN)];
    gemm_nt_big_body<BN, EPI, X3>(smem, blockIdx.x, A, lda, rows, M, K, Bm, ldb, N, bias, H, ldh, h_rows, Cv, ldc, tiles_m, tiles_n, c_f32);
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm_nt_persist: the same tile program as gemm_nt_big for the bias / bias + sigmoid epilogues with bf16 output, as ONE
// resident workgroup per CU that walks its tiles while the LDS-DMA stream runs on across tile boundaries.
// Why (in-kernel stamps of the per-tile kernel): every tile began with an empty ring - "entry -> first stage
// landed" was 11 % of the layer-1 forward and 22 % of the layer-2 forward, every CU bursting 96-120 KB at HBM at once -
// and ended with its stores draining while nothing was in flight.  Here the first NS - 1 stages of the next tile are
// issued during the last k-steps of the current one, so the matrix pipe restarts right behind the epilogue, whose stores
// drain under the next tile's loop.
// vmcnt is one in-order counter for LDS-DMA and stores: behind a tile boundary the waits let the epilogue's NST stores ride
// along (allowance = in-flight stages * NL + NST for the first NS - 1 k-steps of a tile, see the ladder below).
// The row indices of all of the workgroup's tiles are parked in LDS up front (<= 8 tiles), so the loop issues no
// VGPR-destination load.  The epilogue's row patch is the stage consumed last (BN = 256: exactly 32 KB) or a region of its
// own (BN = 128).
// ---------------------------------------------------------------------------------------------------------------------
#define NTP_MAX_TILES(BN_) ((BN_) == 256 ? 8 : 7)  // tiles a workgroup parks row indices for (1 KB each); the 128-wide form is at the 160 KB LDS limit
__device__ uint16_t g_ntp_sink[64 * 8];            // where the stores of rows past M go: every wave issues exactly NST stores
MG_STAMP_DECL(g_stamps_ntp);

#define WAIT_VM_LGKM_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define NTP_WAIT_CASE(N) case N: WAIT_VM_LGKM_BARRIER(N); break;

// BK: contraction depth of one LDS stage.  32 (64-byte LDS rows, 4 stages of 32 KB) for the square tile, whose operands sit in L2;
// 64 (128-byte rows, 3 stages of 48 KB) for the 128-wide tile, whose A operand streams from HBM: a DMA instruction then fetches whole
// 128-byte lines (8 rows x 128 B instead of 16 rows x 64 B) - with half lines the layer-2 forward read its 262 MB of H1 at 4.5 TB/s,
// with whole lines (timing probe on this kernel) 20 % faster.  The square tile gains nothing from it (same probe: 180 -> 177 us).
// STAG (cdna_hip_programming.md section 5, the 8-phase template's wave-group stagger): waves 4-7 run half a k-step behind waves 0-3.
// A k-step is two intervals separated by barriers: the leading group reads its fragments from LDS (and issues its share of the
// LDS-DMA) in the first and multiplies in the second; the lagging group multiplies the PREVIOUS step's fragments in the first and reads
// in the second, so that on every SIMD one wave feeds the matrix pipe while the other one occupies LDS.  MEASURED SLOWER than both
// groups in lockstep on this kernel (layer-1 forward 203 vs 180 us, stamps: an interval takes ~1 000 cycles, not the 512 of its 16
// MFMAs - the reading group's 4 LDS-DMA pieces + 12 ds_read_b128 take that long to issue, MI355X_MICROARCH.md "LDS-DMA piece issue
// cost"), so it is kept as an experiment only (MG_TUNE_FORM = 4).  The groups fall back into step at every tile end.
// PIPE (square tile, 32-deep stages, both wave groups in lockstep): the k-step as a half-step software pipeline.  The fragments of a
// stage are two sets (MFMA steps ks = 0 / 1, 6 reads and 8 MFMAs each); a set is read half a step before it is multiplied, and the
// stage boundary (counted vmcnt + barrier, then the LDS-DMA of the stage NS ahead into the slot just vacated) sits BETWEEN the two
// halves:   [8 MFMA (kt, 0)] [wait: stage kt+1 landed; barrier] [DMA stage kt+NS -> slot kt; read (kt+1, 0)] [8 MFMA (kt, 1)] [read (kt+1, 1)]
// so that no MFMA ever waits for an LDS read issued in its own half and the DMA issue (60-185 cycles per piece) runs beside MFMAs.
// The slot of a stage is refilled right behind the barrier that follows its last read, so the epilogue's row patch cannot live in the
// ring: it is a region of its own with 64-byte rows (32 rows x 32 columns per wave and pass, 16 KB in all).
// MODE 2 = SPREAD (experiment): the stage order of the default, but the four LDS-DMA pieces of the stage NS - 1 ahead are issued one at
// a time between the MFMAs, at MFMA slots 4 q + off with off different for the eight waves' SIMD partners (0..3): the address path
// takes a 1 KB piece in at ~32 B/clk, and a wave stands at the issue while the queue is full - with all 32 pieces of a stage issued
// at once that is up to 1 000 cycles per k-step in which no wave of the workgroup issues an MFMA.
// MODE 5 = ROLE (timing probe, results garbage): can the MFMAs and the LDS-DMA of DIFFERENT waves overlap?  Waves 0-3 issue no DMA and
// read no fragments but run every MFMA twice; waves 4-7 issue every piece twice, read their fragments and run no MFMA; every wait is
// vmcnt(0).  Same DMA bytes and MFMA count per k-step as the real kernel.
template <int BN, int BK, int MODE>
constexpr int ntp_lds_bytes() {
    constexpr int STAGE = 256 * BK * 2 + BN * BK * 2;
    constexpr int NS = (BK == 64) ? (BN == 256 ? 2 : 3) : ((BN == 256) ? 4 : 5);
    constexpr int PATCH = NS * STAGE + NTP_MAX_TILES(BN) * 256 * 4;
    constexpr bool kPatchInRing = MODE != 1 && STAGE >= 8 * 32 * 128;
    return PATCH + (kPatchInRing ? 0 : (MODE == 1 ? 8 * 32 * 64 : 8 * 32 * 128)) + BN * 4;
}

// The tile program on a caller-provided LDS block, as workgroup `block_id` of `n_blocks` (gemm_nt_persist_kernel: the plain launch;
// phone_front_gemm_kernel: behind the blocks of an unrelated small job).
// X3: pair-plane operands and a pair-plane output (see gemm_nt_big_body): A [M, lda] = [hi | lo] planes lda / 2 apart, B likewise, K =
// the contraction length of one plane, three passes (hi, hi), (hi, lo), (lo, hi); C [M, ldc] = [hi | lo] of the fp32 result (N columns
// per plane, planes ldc / 2 apart): twice the epilogue stores per tile.
template <int BN, int EPI, bool STAG, int BK, int MODE = 0, int BMV = 256, int X3 = 0>
__device__ __forceinline__ void gemm_nt_persist_body(unsigned char* __restrict__ smem, const unsigned block_id, const unsigned n_blocks,
                                                     const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows, int64_t M,
                                                     int K, const uint16_t* __restrict__ Bm, int ldb, int N, const float* __restrict__ bias,
                                                     uint16_t* __restrict__ C, int ldc, int tiles_m, int tiles_n, int probe) {
    constexpr bool PIPE = MODE == 1, ROLE = MODE == 5, SPREAD = MODE == 2;
    constexpr int BM = 256;
    constexpr int WAVES_N = BN / 64;              // 4 or 2
    constexpr int WAVES_M = 8 / WAVES_N;          // 2 or 4
    // BMV: rows of a tile that hold output (256, or 192 = the square tile cut to three quarters so that the 21 504-row phone table of C2
    // makes 224 one-tile workgroups instead of 168: the LDS stage keeps its 256 row slots, the last 64 are fed from the zero row and
    // never read).  The GEMM alone: 30.4 -> 24.3 us at that shape.
    static_assert(BMV == 256 || (BMV == 192 && BN == 256 && MODE == 0 && !STAG), "192-row tiles: the plain square form only");
    constexpr int WM = BMV / WAVES_M;             // 128, 96 or 64
    constexpr int TM = WM / 32, TN = 2;
    constexpr int ROWB = BK * 2;                  // bytes of a stage row: 64 or 128
    constexpr int KS = BK / 16;                   // 16-deep MFMA steps per stage: 2 or 4
    constexpr int LPR = ROWB / 16;                // lanes (16-byte chunks) per row: 4 or 8
    constexpr int RPP = 64 / LPR;                 // rows per 1 KB DMA piece: 16 or 8
    // X3 == 2 ("native" pair planes): ONE ring slot holds the k-tile of all four planes - [A hi | B hi] and behind them [A lo | B lo],
    // each in the layout of a plain stage - and the step multiplies three times from it: four tile fills per k-tile where the three
    // passes of X3 == 1 make six.  Two slots of twice the size in the LDS of the four-slot ring.
    constexpr bool NATIVE = X3 == 2;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int PLANE = A_BYTES + B_BYTES;      // one plane pair's images
    constexpr int STAGE = NATIVE ? 2 * PLANE : PLANE;      // 32 KB (BN 256, BK 32), 24 KB (BN 128, BK 32) or 48 KB (BN 128, BK 64); native: twice
    constexpr int GA = BM / RPP / 8;              // 1 KB pieces per wave for A: 2 or 4
    constexpr int GB = BN / RPP / 8;              // for B: 2, 1 or 2
    constexpr int NL = (GA + GB) * (NATIVE ? 2 : 1);       // LDS-DMA instructions per wave per stage: 4, 3 or 6
    // (BN 256 with 64-deep stages: two slots of 64 KB - a DMA row is a whole 128-byte line, half the line requests of two 32-deep stages)
    constexpr int NS = NATIVE ? 2 : ((BK == 64) ? (BN == 256 ? 2 : 3) : ((BN == 256) ? 4 : 5));
    constexpr int NST = TM * 4 * (X3 ? 2 : 1);    // epilogue stores per wave per tile: 16 or 8 (X3: both planes)
    static_assert(!X3 || (MODE == 0 && !STAG && BN == 256 && (BK == 32 || X3 == 1)), "X3: the plain square form only (native: 32-deep)");
    constexpr int ROWTAB = NS * STAGE;            // int32[MAXT][256]
    constexpr int MAXT = NTP_MAX_TILES(BN);
    constexpr int PATCH = ROWTAB + MAXT * BM * 4;
    constexpr int SP = 128;                       // patch row: 64 columns x 2 B, 16-byte chunk c of row r at c ^ (r & 7)
    static_assert(!PIPE || (BN == 256 && BK == 32 && !STAG), "PIPE: square tile, 32-deep stages, lockstep");
    constexpr bool kPatchInRing = !PIPE && STAGE >= 8 * 32 * SP;     // the epilogue's row patch fits the ring slot consumed last
    constexpr int SPP = 64;                       // PIPE: patch row = 32 columns x 2 B, chunk c of row r at c ^ ((r >> 1) & 3)
    constexpr int BIAS_OFF = PATCH + (kPatchInRing ? 0 : (PIPE ? 8 * 32 * SPP : 8 * 32 * SP));
    constexpr int LDS_BYTES = BIAS_OFF + BN * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

    static_assert(LDS_BYTES == ntp_lds_bytes<BN, BK, MODE>(), "LDS size helper");
    int* rowtab = reinterpret_cast<int*>(smem + ROWTAB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MG_STAMPS
    unsigned long long ts0, ts1 = 0, ts2, ts3, tr0, tr1, ta, tb, sum_wait = 0, sum_epi = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    const int wm0 = (wave / WAVES_N) * WM, wn0 = (wave % WAVES_N) * 64;
    const int n_kp = (K + BK - 1) / BK;           // stages that hold real columns (lda, ldb >= 64 * ceil(K / 64))
    const int n_kt = X3 == 1 ? 3 * n_kp : n_kp;   // X3 == 1: three passes over the plane
    const int a_lo = lda >> 1, b_lo = ldb >> 1;   // X3: element offset of an operand's lo plane

    // Virtual block v = blockIdx.x + i gridDim.x (gridDim.x a multiple of 8) keeps the XCD-aware order of gemm_nt_big: the N
    // tiles of one M tile go to blocks 8 apart, which share an XCD and its L2.
    auto tile_of = [&](int i, int& tile_m, int& tile_n) {
        const int v = (int)(block_id + i * n_blocks);
        const int xcd = v & 7, jj = v >> 3;
        tile_n = jj % tiles_n;
        tile_m = (jj / tiles_n) * 8 + xcd;
    };
    int n_my = 0;                                 // the launcher makes (gridDim.x / 8) a multiple of tiles_n: one N tile per workgroup
    for (int i = 0; i < MAXT; ++i) {
        int tm, tn;
        tile_of(i, tm, tn);
        if (tm < tiles_m) n_my = i + 1;
    }
    // park the source rows of every tile of this workgroup (-1: zero row)
    for (int e = tid; e < n_my * BM; e += 512) {
        int tm, tn;
        tile_of(e / BM, tm, tn);
        const int64_t m = (int64_t)tm * BMV + (e % BM);
        int r = -1;
        if (m < M && (e % BM) < BMV) r = rows ? rows[m] : (int)m;
        rowtab[e] = r;
    }
    __syncthreads();
    if (n_my == 0) return;

    // chunk swizzle of a stage row (the same involution on the DMA source and on the fragment reads): conflict free for the four 16-lane
    // groups of ds_read_b128 - 64-byte rows: chunk ^ ((row >> 2) & 3); 128-byte rows: chunk ^ ((row >> 1) & 7)
    auto swz = [](int row) { return BK == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7); };
    // issue cursor: tile i_t, k-tile i_k, ring slot i_s, per-lane sources of that tile
    const uint16_t* asrc[GA];
    const uint16_t* bsrc[GB];
    auto set_sources = [&](int i) {
        int tm, tn;
        tile_of(i, tm, tn);
#pragma unroll
        for (int g = 0; g < GA; ++g) {
            const int row = (wave * GA + g) * RPP + lane / LPR;
            const int c = (lane % LPR) ^ swz(row);
            const int r = rowtab[i * BM + row];
            asrc[g] = (r >= 0 ? A + (size_t)r * lda : g_zero_row) + c * 8;
        }
#pragma unroll
        for (int g = 0; g < GB; ++g) {
            const int row = (wave * GB + g) * RPP + lane / LPR;
            const int c = (lane % LPR) ^ swz(row);
            const int n = tn * BN + row;
            bsrc[g] = (n < N ? Bm + (size_t)n * ldb : g_zero_row) + c * 8;
        }
    };
    int i_t = 0, i_k = 0, i_s = 0, i_p = 0, i_w = 0;          // (X3: pass i_p, k-tile i_w of the plane; i_k counts both)
    set_sources(0);
    auto issue_next = [&](int reps = 1) {         // next stage of the stream, if any is left
        if (i_t >= n_my) return;
        unsigned char* st = smem + i_s * STAGE;
        const int ka = X3 == 1 ? i_w * BK + (i_p == 2 ? a_lo : 0) : i_k * BK;
        const int kb = X3 == 1 ? i_w * BK + (i_p == 1 ? b_lo : 0) : i_k * BK;
        if (NATIVE) {
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int g = 0; g < GA; ++g) glds16(asrc[g] + ka + pl * a_lo, st + pl * PLANE + (wave * GA + g) * 1024);
#pragma unroll
                for (int g = 0; g < GB; ++g) glds16(bsrc[g] + kb + pl * b_lo, st + pl * PLANE + A_BYTES + (wave * GB + g) * 1024);
            }
        } else
        // probe bits (timing experiments, results garbage): 1 = no A pieces, 2 = no B pieces, 4 = no fragment reads, 8 = no MFMAs, 16 = no epilogue
        for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
        for (int g = 0; g < GA; ++g)
            if (!(probe & 1)) glds16(asrc[g] + ka, st + (wave * GA + g) * 1024);
#pragma unroll
        for (int g = 0; g < GB; ++g)
            if (!(probe & 2)) glds16(bsrc[g] + kb, st + A_BYTES + (wave * GB + g) * 1024);
        }
        i_s = (i_s + 1 == NS) ? 0 : i_s + 1;
        if (X3 == 1 && ++i_p == 3) {              // pass fastest: an operand's k-tile is read again while the XCD's L2 still holds it
            i_p = 0;
            ++i_w;
        }
        if (++i_k == n_kt) {
            i_k = 0;
            i_p = 0;
            i_w = 0;
            if (++i_t < n_my) set_sources(i_t);
        }
    };

    auto issue_piece = [&](int q) {               // SPREAD: piece q of the next stage; advance_stage() after the last
        if (i_t >= n_my) return;
        unsigned char* st = smem + i_s * STAGE;
        if (q < GA) glds16(asrc[q < GA ? q : 0] + i_k * BK, st + (wave * GA + q) * 1024);
        else glds16(bsrc[q < GA ? 0 : q - GA] + i_k * BK, st + A_BYTES + (wave * GB + (q - GA)) * 1024);
    };
    auto advance_stage = [&]() {
        if (i_t >= n_my) return;
        i_s = (i_s + 1 == NS) ? 0 : i_s + 1;
        if (++i_k == n_kt) {
            i_k = 0;
            if (++i_t < n_my) set_sources(i_t);
        }
    };
    const int spread_off = (probe & 32) ? 0 : ((wave + (wave >> 2)) & 3);     // partners on a SIMD (w, w + 4) differ by one slot

    const int lr = lane & 31, lh = lane >> 5;
    // byte offset of this lane's fragment of MFMA step ks inside a stage: row base + ((2 ks + lh) ^ swz(row)) * 16
    int abase[TM], bbase[TN], asw[TM], bsw[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wm0 + i * 32 + lr;
        abase[i] = row * ROWB;
        asw[i] = lh ^ swz(row);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn0 + j * 32 + lr;
        bbase[j] = A_BYTES + row * ROWB;
        bsw[j] = lh ^ swz(row);
    }

    // The workgroup's bias slice (one N tile per workgroup) is parked in LDS next to the row table and read back in accumulator order
    // (register 4 q + e <-> column 32 j + 8 q + 4 lh + e) by each epilogue: held in 32 registers for the whole kernel, as it first was,
    // it pushed the staggered loop (fragments live across the barrier) over the 256-VGPR limit.  A VGPR-destination global load inside
    // the tile loop is not an option either: hipcc drains vmcnt - the whole prefetched ring and the stores - in front of its first use.
    float* bias_lds = reinterpret_cast<float*>(smem + BIAS_OFF);
    {
        int tm, tn;
        tile_of(0, tm, tn);
        for (int c = tid; c < BN; c += 512) bias_lds[c] = (bias && tn * BN + c < N) ? bias[tn * BN + c] : 0.f;
        __syncthreads();                          // plain barrier: no LDS-DMA is in flight yet
    }

#pragma unroll
    for (int p = 0; p < (PIPE ? NS : NS - 1); ++p) issue_next(ROLE ? (wave < 4 ? 0 : 2) : 1);       // PIPE fills every slot: slot g % NS is refilled during step g

    // All fragment reads of a stage first, then its MFMAs (pinned with sched_group_barrier): left alone hipcc reads two to four
    // fragments at a time with an lgkmcnt(0) in front of every MFMA group - six exposed LDS round trips per stage.
    const bool lag = (STAG || ROLE) && wave >= 4; // wave-uniform
    bfv8 fa[KS][TM], fb[KS][TN];
    if (ROLE) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[ks][i] = bfv8{};
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[ks][j] = bfv8{};
        }
    }
    auto read_frags = [&](const unsigned char* st) {
        if (probe & 4) return;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bfv8*>(st + abase[i] + ((asw[i] ^ (2 * ks)) << 4));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const bfv8*>(st + bbase[j] + ((bsw[j] ^ (2 * ks)) << 4));
        }
        __builtin_amdgcn_sched_group_barrier(0x100, KS * (TM + TN), 0);
    };
    auto mfma_all = [&](f32x16 (&acc)[TM][TN]) {
        if (probe & 8) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[ks][i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[ks][j]));
            }
            return;
        }
        if (STAG) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = mg_mfma_32x32x16(fb[ks][j], fa[ks][i], acc[i][j]);
        __builtin_amdgcn_sched_group_barrier(0x008, KS * TM * TN * MG_MFMA_PER_TILE, 0);
        if (STAG) __builtin_amdgcn_s_setprio(0);
    };

    auto read_half = [&](const unsigned char* st, int ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bfv8*>(st + abase[i] + ((asw[i] ^ (2 * ks)) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const bfv8*>(st + bbase[j] + ((bsw[j] ^ (2 * ks)) << 4));
    };
    auto mfma_half = [&](f32x16 (&acc)[TM][TN], int ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = mg_mfma_32x32x16(fb[ks][j], fa[ks][i], acc[i][j]);
    };

    const int g_total = n_my * n_kt;
    int g = 0, c_s = 0;                           // global stage counter, its ring slot
    if (PIPE) {
        // stage 0 landed (up to NS - 1 younger stages may stay in flight), then both fragment sets of it
        switch (min(g_total - 1, NS - 1) * NL) {
            NTP_WAIT_CASE(4) NTP_WAIT_CASE(8) NTP_WAIT_CASE(12)
            default: WAIT_VM_LGKM_BARRIER(0); break;
        }
        read_half(smem, 0);
        read_half(smem, 1);
    }
    // The tile loop exists twice, once per wave group (LAG = the group that runs half a k-step behind), selected once by a wave-uniform
    // branch: with the group tested inside one loop body the register allocator has to reconcile the two schedules at every merge
    // point and spills (425-569 VGPRs of scratch measured); as two independent loops each stays within 256.
    auto run_tiles = [&](auto lag_c) {
    constexpr bool LAG = decltype(lag_c)::value;
    for (int ti = 0; ti < n_my; ++ti) {
        int tile_m, tile_n;
        tile_of(ti, tile_m, tile_n);
        const int64_t m0 = (int64_t)tile_m * BMV;
        const int n0 = tile_n * BN;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        if (PIPE) {
            for (int kt = 0; kt < n_kt; ++kt, ++g) {
                const bool more = g + 1 < g_total;
                mfma_half(acc, 0);
                const unsigned char* st_next = smem + ((g + 1) % NS) * STAGE;
                if (more) {
                    // stage g + 1 must have landed; issued so far: stages up to g + NS - 1, so g + 2 and g + 3 may stay in flight, and
                    // the previous tile's NST stores while they are younger than stage g + 1's DMA (first NS - 1 steps of a tile)
                    const int allow = max(0, min(g_total - 2 - g, NS - 2)) * NL + ((ti > 0 && kt < NS - 1) ? NST : 0);
                    MG_STAMP(ta);
                    __builtin_amdgcn_sched_barrier(0);
                    switch (allow) {
                        NTP_WAIT_CASE(0) NTP_WAIT_CASE(4) NTP_WAIT_CASE(8) NTP_WAIT_CASE(16) NTP_WAIT_CASE(20) NTP_WAIT_CASE(24)
                        default: WAIT_VM_LGKM_BARRIER(0); break;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    MG_STAMP(tb);
                    MG_STAMP_ADD(sum_wait, tb, ta);
#ifdef MG_STAMPS
                    if (g == 0) ts1 = tb;
#endif
                    issue_next();                  // stage g + NS into the slot every wave has just finished reading
                    read_half(st_next, 0);
                }
                mfma_half(acc, 1);
                if (more) read_half(st_next, 1);
            }
        } else
        for (int kt = 0; kt < n_kt; ++kt, ++g) {
            MG_STAMP(ta);
            // stages issued so far: min(g_total, g + NS - 1); stage g must have landed; younger than it are
            // min(g_total - 1 - g, NS - 2) stages and, for the first NS - 1 k-steps behind a tile boundary, the NST stores.
            // The wait also retires this wave's LDS reads (lgkmcnt): the lagging group comes here straight from its fragment reads,
            // and the slot they came from is refilled right behind this barrier.
            {
                const int allow = ROLE ? (LAG ? min(g_total - 1 - g, NS - 2) * 2 * NL : 0) : min(g_total - 1 - g, NS - 2) * (((probe & 3) == 3) ? 0 : ((probe & 3) ? NL / 2 : NL)) + ((ti > 0 && kt < NS - 1 && !(probe & 16)) ? NST : 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(probe & 64))                // probe bit 64: no wait, no barrier (only meaningful without DMA and reads)
                switch (allow) {
                    NTP_WAIT_CASE(0) NTP_WAIT_CASE(3) NTP_WAIT_CASE(4) NTP_WAIT_CASE(6) NTP_WAIT_CASE(8) NTP_WAIT_CASE(9)
                    NTP_WAIT_CASE(11) NTP_WAIT_CASE(12) NTP_WAIT_CASE(14) NTP_WAIT_CASE(16) NTP_WAIT_CASE(17) NTP_WAIT_CASE(20)
                    NTP_WAIT_CASE(24) NTP_WAIT_CASE(28) NTP_WAIT_CASE(32) NTP_WAIT_CASE(36) NTP_WAIT_CASE(40)
                    default: WAIT_VM_LGKM_BARRIER(0); break;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            MG_STAMP(tb);
            MG_STAMP_ADD(sum_wait, tb, ta);
#ifdef MG_STAMPS
            if (g == 0) ts1 = tb;
#endif
            const unsigned char* st = smem + c_s * STAGE;
            if (SPREAD) {
                static_assert(!SPREAD || (NL == 4 && KS * TM * TN == 16), "SPREAD: 4 pieces over 16 MFMAs");
                read_frags(st);
#pragma unroll
                for (int mi = 0; mi < 16; ++mi) {
                    if (spread_off == (mi & 3)) issue_piece(mi >> 2);      // the slot every wave finished with before this barrier
                    const int ks = mi / (TM * TN), i = (mi / TN) % TM, j = mi % TN;
                    acc[i][j] = mg_mfma_32x32x16(fb[ks][j], fa[ks][i], acc[i][j]);
                }
                advance_stage();
            } else if (ROLE) {
                if (LAG) {
                    issue_next(2);
                    read_frags(st);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[ks][i]));
#pragma unroll
                        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[ks][j]));
                    }
                } else {
                    mfma_all(acc);
                    mfma_all(acc);
                }
            } else if (NATIVE) {
                issue_next();                     // the other slot: every wave finished with it before this barrier
                // one 16-deep step at a time (both planes' fragments of a step: as many registers as a plain stage's)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    bfv8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        ah[i] = *reinterpret_cast<const bfv8*>(st + abase[i] + ((asw[i] ^ (2 * ks)) << 4));
                        al[i] = *reinterpret_cast<const bfv8*>(st + PLANE + abase[i] + ((asw[i] ^ (2 * ks)) << 4));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        bh[j] = *reinterpret_cast<const bfv8*>(st + bbase[j] + ((bsw[j] ^ (2 * ks)) << 4));
                        bl[j] = *reinterpret_cast<const bfv8*>(st + PLANE + bbase[j] + ((bsw[j] ^ (2 * ks)) << 4));
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            acc[i][j] = mg_mfma_32x32x16(bh[j], ah[i], acc[i][j]);
                            acc[i][j] = mg_mfma_32x32x16(bl[j], ah[i], acc[i][j]);
                            acc[i][j] = mg_mfma_32x32x16(bh[j], al[i], acc[i][j]);
                        }
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * TM * TN * MG_MFMA_PER_TILE, 0);
                }
            } else if (!STAG) {
                issue_next();                     // refills the slot every wave finished with before this barrier
                read_frags(st);
                mfma_all(acc);
            } else {
                if (!LAG) {
                    issue_next();
                    read_frags(st);
                } else if (kt > 0) {
                    mfma_all(acc);                // the previous step's fragments
                }
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                if (!LAG) {
                    mfma_all(acc);
                } else {
                    issue_next();
                    read_frags(st);
                }
            }
            if (kt + 1 < n_kt) c_s = (c_s + 1 == NS) ? 0 : c_s + 1;
        }

        // ---- epilogue: bias (+ sigmoid), bf16, whole 128-byte row segments through the LDS patch ---------------------------
        MG_STAMP(ta);
        if (probe & 16) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[i][j][r]));
            WAIT_LGKM_BARRIER();
            c_s = (c_s + 1 == NS) ? 0 : c_s + 1;
            continue;
        }
        if (PIPE) {
            // wave-private patch outside the ring: no barrier; 32 rows x 32 columns per pass, rows leave as 64-byte segments
            unsigned char* patch = smem + PATCH + wave * (32 * SPP);
            const int prow = lane >> 2, pchunk = lane & 3;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 bq = *reinterpret_cast<const f32x4*>(bias_lds + wn0 + j * 32 + 8 * q + 4 * lh);
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float x = acc[i][j][4 * q + e] + bq[e];
                            if (EPI == EPI_BIAS_SIGMOID) x = mg_sigmoid_fast(x);
                            v[e] = x;
                        }
                        typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
                        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
                        const u32x2_t pk = u32x2_t{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                                   __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
                        *reinterpret_cast<u32x2_t*>(patch + lr * SPP + ((q ^ ((lr >> 1) & 3)) << 4) + 8 * lh) = pk;
                    }
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                        const int rl = it * 16 + prow;
                        const u32x4_t o = *reinterpret_cast<const u32x4_t*>(patch + rl * SPP + ((pchunk ^ ((rl >> 1) & 3)) << 4));
                        const int64_t m = m0 + wm0 + i * 32 + rl;
                        uint16_t* dst = (m < M) ? C + (size_t)m * ldc + n0 + wn0 + j * 32 + pchunk * 8 : g_ntp_sink + lane * 8;
                        *reinterpret_cast<u32x4_t*>(dst) = o;     // unconditional: the counted vmcnt waits rely on NST stores per wave
                    }
                }
            }
            MG_STAMP(tb);
            MG_STAMP_ADD(sum_epi, tb, ta);
            continue;
        }
        __builtin_amdgcn_sched_barrier(0);
        WAIT_LGKM_BARRIER();                      // every wave is done with the last stage (the lagging group has just read it)
        __builtin_amdgcn_sched_barrier(0);
        if (STAG && LAG) mfma_all(acc);           // its last multiply runs while the leading group starts the epilogue
        unsigned char* patch = (kPatchInRing ? smem + c_s * STAGE : smem + PATCH) + wave * (32 * SP);
        c_s = (c_s + 1 == NS) ? 0 : c_s + 1;
        const int prow = lane >> 3, pchunk = lane & 7;
        f32x4 bv[TN][4];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[j][q] = *reinterpret_cast<const f32x4*>(bias_lds + wn0 + j * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
            u32x2_t pk_lo[X3 ? TN : 1][4];                                    // X3: the lo plane's values, written in a second pass
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[i][j][4 * q + e] + bv[j][q][e];
                        if (EPI == EPI_BIAS_SIGMOID) x = mg_sigmoid_fast(x);
                        v[e] = x;
                    }
                    const bfv2 h01 = bfv2{(__bf16)v[0], (__bf16)v[1]}, h23 = bfv2{(__bf16)v[2], (__bf16)v[3]};
                    const u32x2_t pk = u32x2_t{__builtin_bit_cast(unsigned int, h01), __builtin_bit_cast(unsigned int, h23)};
                    if (X3) {
                        const bfv2 l01 = bfv2{(__bf16)(v[0] - (float)h01[0]), (__bf16)(v[1] - (float)h01[1])};
                        const bfv2 l23 = bfv2{(__bf16)(v[2] - (float)h23[0]), (__bf16)(v[3] - (float)h23[1])};
                        pk_lo[X3 ? j : 0][q] = u32x2_t{__builtin_bit_cast(unsigned int, l01), __builtin_bit_cast(unsigned int, l23)};
                    }
                    const int chunk = 4 * j + q;                              // columns 32 j + 8 q .. + 7 of the 64-wide strip
                    *reinterpret_cast<u32x2_t*>(patch + lr * SP + ((chunk ^ (lr & 7)) << 4) + 8 * lh) = pk;
                }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                const int rl = it * 8 + prow;
                const u32x4_t o = *reinterpret_cast<const u32x4_t*>(patch + rl * SP + ((pchunk ^ (rl & 7)) << 4));
                const int64_t m = m0 + wm0 + i * 32 + rl;
                uint16_t* dst = (m < M) ? C + (size_t)m * ldc + n0 + wn0 + pchunk * 8 : g_ntp_sink + lane * 8;
                *reinterpret_cast<u32x4_t*>(dst) = o;     // unconditional: the counted vmcnt waits rely on NST stores per wave
            }
            if (X3) {
                // the lo plane through the same wave-private patch (a wave's LDS operations complete in order)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<u32x2_t*>(patch + lr * SP + (((4 * j + q) ^ (lr & 7)) << 4) + 8 * lh) = pk_lo[X3 ? j : 0][q];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                    const int rl = it * 8 + prow;
                    const u32x4_t o = *reinterpret_cast<const u32x4_t*>(patch + rl * SP + ((pchunk ^ (rl & 7)) << 4));
                    const int64_t m = m0 + wm0 + i * 32 + rl;
                    uint16_t* dst = (m < M) ? C + (size_t)m * ldc + (ldc >> 1) + n0 + wn0 + pchunk * 8 : g_ntp_sink + lane * 8;
                    *reinterpret_cast<u32x4_t*>(dst) = o;
                }
            }
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_epi, tb, ta);
    }
    };
    if (lag) run_tiles(std::true_type{}); else run_tiles(std::false_type{});
#ifdef MG_STAMPS
    MG_STAMP(ts2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG_STAMP(ts3);
    MG_STAMP_REAL(tr1);
    const int sb = block_id;
    MG_STAMP_STORE(g_stamps_ntp, sb, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_ntp, sb, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_ntp, sb, wave, lane, 2, ts2);
    MG_STAMP_STORE(g_stamps_ntp, sb, wave, lane, 3, ts3);
    MG_STAMP_STORE(g_stamps_ntp, sb, wave, lane, 4, tr0);
    MG_STAMP_STORE(g_stamps_ntp, sb, wave, lane, 5, tr1);
    MG_STAMP_STORE(g_stamps_ntp, sb, wave, lane, 6, sum_wait);
    MG_STAMP_STORE(g_stamps_ntp, sb, wave, lane, 7, sum_epi);
#endif
}

template <int BN, int EPI, bool STAG, int BK, int MODE = 0>
__global__ __launch_bounds__(512) void gemm_nt_persist_kernel(const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                              int64_t M, int K, const uint16_t* __restrict__ Bm, int ldb, int N,
                                                              const float* __restrict__ bias, uint16_t* __restrict__ C, int ldc,
                                                              int tiles_m, int tiles_n, int probe) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[ntp_lds_bytes<BN, BK, MODE>()];
    gemm_nt_persist_body<BN, EPI, STAG, BK, MODE>(smem, blockIdx.x, gridDim.x, A, lda, rows, M, K, Bm, ldb, N, bias, C, ldc, tiles_m, tiles_n,
                                                  probe);
}

// The front of the phone-rate step (csrc/phone_front.h: frame map + per-phone loss statistics, one job per utterance) in the grid of
// the first layer's GEMM, which reads none of its outputs: blocks [0, side_blocks) run the jobs, the blocks behind them are the persistent tile program.  The GEMM of
// C2's phone table has 168 tiles, one per workgroup and CU: the jobs take CUs it leaves idle and two launch boundaries disappear.
// side_blocks is a multiple of 8 (the tile order derives a block's XCD from its id modulo 8).
template <int EPI, int BMV = 256, int X3 = 0, int BK = 32>
__global__ __launch_bounds__(512) void phone_front_gemm_kernel(unsigned side_blocks, int wave_ints, PhoneFrontArgs pf, const uint16_t* __restrict__ A, int lda,
                                                               int64_t M, int K, const uint16_t* __restrict__ Bm, int ldb, int N,
                                                               const float* __restrict__ bias, uint16_t* __restrict__ C, int ldc, int tiles_m,
                                                               int tiles_n) {
    static_assert(ntp_lds_bytes<256, BK, 0>() == ntp_lds_bytes<256, 32, 0>(), "the front's jobs were sized for this LDS block");
    __shared__ __attribute__((aligned(16))) unsigned char smem[ntp_lds_bytes<256, BK, 0>()];
    if (blockIdx.x < side_blocks) {
        if (pf.lds_ints > 0) {
            if (wave_ints > 0)       // wave jobs (phone_front.h): the block's eight waves each work off jobs of their own, no barrier
                phone_front_wave_jobs(pf, (int)blockIdx.x * 8 + (int)(threadIdx.x >> 6), (int)side_blocks * 8,
                                      reinterpret_cast<int*>(smem) + (threadIdx.x >> 6) * wave_ints);
            else
                phone_front_block<512>(pf, blockIdx.x, side_blocks, reinterpret_cast<int*>(smem));
        }
        return;                                          // lds_ints == 0: timing probe, the rider's blocks leave at once
    }
    if (pf.probe & 8) return;
    gemm_nt_persist_body<256, EPI, false, BK, 0, BMV, X3>(smem, blockIdx.x - side_blocks, gridDim.x - side_blocks, A, lda, nullptr, M, K, Bm, ldb, N, bias, C,
                                                     ldc, tiles_m, tiles_n, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// wgrad_big: output tile 128 (n) x BKT (k), BKT = 64 * TKW (TKW = 10 -> 640, 8 -> 512); contraction over m in steps of 32,
// THREE LDS stages (two steps of LDS-DMA in flight, one barrier per step).
// LDS tiles are straight row copies ([m][n], [m][k]); fragments come from ds_read_b64_tr_b16.  16-byte chunk c of
// tile row r sits at chunk position c ^ ((r & 3) << 2), i.e. the 64-byte blocks of a row are XORed with r & 3, so the 4
// rows x 64 bytes a half wave touches in one transposed read fall on all 64 banks.
// Row indices of the workgroup's whole m range are parked in LDS first, so the loop issues no VGPR-destination loads.
// ---------------------------------------------------------------------------------------------------------------------
#define WG_ROWS_MAX 4096
#define WG_STAGES(TKW_) (((TKW_) == 5 || (TKW_) == 4) ? 4 : 3)        // ring depth: the half-width tile has the LDS for a fourth stage
MG_STAMP_DECL(g_stamps_wg);

// X tile row pitch: the k columns of the tile, rounded up to whole groups of 16 chunks (the chunk swizzle XORs bits 2-3 of the chunk index)
#define WG_BIG_PX(TKW_) ((64 * (TKW_) * 2 + 255) / 256 * 256)
#define WG_BIG_LDS(TKW_) (WG_STAGES(TKW_) * (32 * 256 + 32 * WG_BIG_PX(TKW_)) + WG_ROWS_MAX * 4)
#define WG_STAGES_NW(TKW_, NW_) (((NW_) == 2 && (TKW_) == 5) ? 3 : WG_STAGES(TKW_))          // 256 x 320: 40 KB stages
#define WG_BIG_LDS_NW(TKW_, NW_) (WG_STAGES_NW(TKW_, NW_) * (32 * 256 * (NW_) + 32 * WG_BIG_PX(TKW_)) + WG_ROWS_MAX * 4)      // NW_ x 128 n columns
// DYR: the dY operand is gathered too - row m of the product is dY[dy_rows[m]] (x) A[rows[m]]: the valid frames of a ragged batch picked
// out of the padded (B, T) arrays a recurrence writes (morgana/utils.py:366-385 packs them away), so that the layer's weight
// gradients multiply sum_b T_b rows instead of B T.  A second parked index table: 16 KB more LDS, a kernel of its own.
// (NW = 2 with TKW = 5: 256 x 320, the same for the 640-wide operand - 288 lines per step where the 128 x 640 tile asks for 384.)
// NW = 2 (with TKW = 4): the SQUARE tile 256 (n) x 256 (k), two k halves per n tile, 4 waves along n x 2 along k - the wave's 64 x 128 part
// and its fragment reads are those of the 128 x 512 tile, the stage holds 256 + 256 instead of 128 + 512 operand columns: a fifth
// fewer 128-byte lines per step for the same products, and the loop is bound by the lines a CU can request (~8 cycles each;
// profiles/r4_notes_falsified_kernel_ideas.txt).  Same number of tiles per split as the 128 x 512 form, so the same plan.
// X3 (pair planes, see gemm_nt_big_body): dY [M, lddy] and A [M, lda] are [hi | lo] pairs (planes lddy / 2 / lda / 2 apart); the
// workgroup walks its row range three times - (dY hi, A hi), (dY lo, A hi), (dY hi, A lo) - into the same accumulators.  No bias sums
// here (the column sums of the fp32 gradient come from the kernel that produced it).
template <int TKW, bool DYR = false, int NW = 1, int X3 = 0>
__device__ __forceinline__ void wgrad_big_body(unsigned char* __restrict__ smem, const unsigned block_id, const uint16_t* __restrict__ dY,
                                               int lddy, const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows, int64_t M,
                                               int N, int K, int m_chunk, float* __restrict__ slab, float* __restrict__ bslab, int64_t sstride,
                                               int xcd_group, const int32_t* __restrict__ dy_rows = nullptr) {
    static_assert(!X3 || !DYR, "X3: no gathered dY form");
    // TKW = 10 / 8: the workgroup's tile is 128 x 640 / 128 x 512 (all k columns of the operand), 2 waves along n x 4 along k.
    // TKW = 5: 128 x 320 - HALF the k columns, two k halves per n tile (KH = 2), 4 waves along n x 2 along k: the same five k tiles per
    // wave with one n tile instead of two.  Twice the tiles per split means two thirds of the splits for a full chip (8 x 32 instead of
    // 4 x 48 workgroups at the phone-rate rows of C2): a third less slab traffic and half the epilogue per workgroup.
    static_assert(NW == 1 || (NW == 2 && (TKW == 4 || TKW == 5)), "the 256-wide n tile goes with a k half (256 or 320 columns)");
    constexpr int BNT = 128 * NW, BKT = 64 * TKW;
    constexpr int KH = (TKW == 5 || TKW == 4) ? 2 : 1;    // k tiles per operand row (TKW = 4: 128 x 256, the half of TKW = 8's tile)
    constexpr int WK = KH == 2 ? 2 : 4;           // waves along k
    constexpr int TNW = BNT / 32 / (8 / WK);      // 32-row MFMA tiles per wave along n: 2 or 1
    constexpr int TKT = BKT / WK / 32;            // 32-column MFMA tiles per wave along k: 5 or 4
    constexpr int PY = BNT * 2, PX = WG_BIG_PX(TKW);      // LDS row pitches in bytes: 256; 1280 / 1024 / 768 (640 of it in use)
    constexpr int XC = BKT * 2 / 16;              // 16-byte chunks of an X row that hold operand columns
    constexpr int Y_BYTES = 32 * PY, X_BYTES = 32 * PX;
    constexpr int STAGE = Y_BYTES + X_BYTES;
    constexpr int NX = X_BYTES / 1024 / 8;        // X LDS-DMA instructions per wave per step: 5, 4, 3 or 2
    constexpr int NY = Y_BYTES / 1024 / 8;        // dY: 1 (2 for the 256-wide n tile)
    constexpr int NLW = NY + NX;
    static_assert(X_BYTES % 8192 == 0, "whole pieces per wave");
    // X3 == 2 ("native" pair planes): ONE ring slot holds the step's rows of all four planes - [dY hi | A hi] and behind them
    // [dY lo | A lo], each in the layout of a plain stage - and the step multiplies three times from it: 14 lines per row where the
    // three walks of X3 == 1 request 21 (the loop is bound by the lines a CU can request, profiles/r4_notes_falsified_kernel_ideas.txt).
    // Two slots of twice the size in the LDS of the four-slot ring.
    constexpr bool NATIVE = X3 == 2;
    static_assert(!NATIVE || WG_STAGES_NW(TKW, NW) == 4, "native pair planes: the tiles whose ring has four slots");
    constexpr int NSTG = NATIVE ? 2 : WG_STAGES_NW(TKW, NW);
    constexpr int SLOT = NATIVE ? 2 * STAGE : STAGE;
    constexpr int LDS_BYTES = NSTG * SLOT + WG_ROWS_MAX * 4 * (DYR ? 2 : 1);

    static_assert(LDS_BYTES == WG_BIG_LDS_NW(TKW, NW) + (DYR ? WG_ROWS_MAX * 4 : 0) && LDS_BYTES <= 160 * 1024, "LDS size helper");
    int* row_lds = reinterpret_cast<int*>(smem + NSTG * SLOT);
    int* dyrow_lds = row_lds + WG_ROWS_MAX;       // DYR only

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MG_STAMPS
    unsigned long long ts0, ts1 = 0, ts2, ts3, tr0, tr1, ta, tb, sum_wait = 0, sum_issue = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    const int wn0 = (wave / WK) * (TNW * 32);     // 2 (4) waves along n
    const int wk0 = (wave % WK) * (TKT * 32);     // 4 (2) waves along k
    // Block order: n tile fastest, then split (measured: grouping the n tiles of a split on one XCD so that they share the
    // gathered X rows through its L2 was 15 % SLOWER, 249 vs 217 us on the C2 layer-1 shape).
    // xcd_group: blocks b, b + 8, b + 16, ... share an XCD (and its L2): give them the n tiles of ONE split, so the split's X rows
    // leave HBM once (phone-rate shapes, where X is a streamed table rather than an L2-resident set of gathered rows).  The number
    // of splits is a multiple of 8 (mg_wgrad_big_plan).
    const int tiles = (N / BNT) * KH;             // workgroups per split: n tiles x k halves, k half fastest
    int tile, s;
    if (xcd_group) {
        tile = (block_id >> 3) % tiles;
        s = (block_id / (8 * tiles)) * 8 + (block_id & 7);
    } else {
        tile = block_id % tiles;
        s = block_id / tiles;
    }
    const int n0 = (tile / KH) * BNT;
    const int kh = tile % KH, k0 = kh * BKT;       // this workgroup's k columns: [k0, k0 + BKT)
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t m_hi = min(M, m_lo + (int64_t)m_chunk);
    const int n_rows = m_hi > m_lo ? (int)(m_hi - m_lo) : 0;     // trailing splits may be empty: they write zero slabs

    // Park source-row indices (identity when there is no gather; -1 = zero row) for the whole m range.
    for (int i = tid; i < m_chunk; i += 512) {
        int r = -1;
        if (i < n_rows) r = rows ? rows[m_lo + i] : (int)(m_lo + i);
        row_lds[i] = r;
        if (DYR) dyrow_lds[i] = i < n_rows ? dy_rows[m_lo + i] : -1;
    }
    __syncthreads();

    // DMA slots.  dY: one 1 KB piece (4 rows of 256 B) per wave.  X: NX pieces per wave, piece t = wave*NX + i covers
    // linear tile bytes [1024 t, 1024 t + 1024); a lane's byte decides its tile row and chunk position.
    int y_row[NY], y_c[NY];
#pragma unroll
    for (int i = 0; i < NY; ++i) {                // (one piece of a 128-wide tile: row wave * 4 + lane / 16, chunk lane % 16 swizzled)
        const int byte = (wave * NY + i) * 1024 + lane * 16;
        y_row[i] = byte / PY;
        y_c[i] = ((byte % PY) >> 4) ^ ((y_row[i] & 3) << 2);
    }
    int x_row[NX], x_off[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int byte = (wave * NX + i) * 1024 + lane * 16;
        x_row[i] = byte / PX;
        const int cpos = (byte % PX) >> 4;
        const int src = cpos ^ ((x_row[i] & 3) << 2);         // source chunk that belongs at this position
        x_off[i] = src < XC ? k0 + src * 8 : -1;              // (-1: a position of the pitch's padding - filled from the zero row)
    }

    const int n_steps1 = (n_rows + 31) / 32;     // steps of one walk over the workgroup's rows
    auto issue = [&](int step_all) {             // rows [32 step, 32 step + 32) of this workgroup's range (X3: step = step_all / 3, pass step_all % 3)
        unsigned char* st = smem + (step_all % NSTG) * SLOT;
        int step = step_all, y_plane = 0, a_plane = 0;
        if (NATIVE) {
            // both planes of both operands, the lo planes' images one plain stage further
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int i = 0; i < NY; ++i) {
                    const int ml = step * 32 + y_row[i];
                    const uint16_t* p = (ml < n_rows) ? dY + (size_t)(m_lo + ml) * lddy + pl * (lddy >> 1) + n0 + y_c[i] * 8 : g_zero_row;
                    glds16(p, st + pl * STAGE + (wave * NY + i) * 1024);
                }
                int rn[NX];
#pragma unroll
                for (int i = 0; i < NX; ++i) rn[i] = row_lds[step * 32 + x_row[i]];
#pragma unroll
                for (int i = 0; i < NX; ++i) {
                    const uint16_t* p = (rn[i] >= 0 && x_off[i] >= 0) ? A + (size_t)rn[i] * lda + pl * (lda >> 1) + x_off[i] : g_zero_row;
                    glds16(p, st + pl * STAGE + Y_BYTES + (wave * NX + i) * 1024);
                }
            }
            return;
        }
        if (X3) {                                 // pass fastest: the rows of a step are read again while the XCD's L2 still holds them
            step = step_all / 3;
            const int pass = step_all - 3 * step;
            y_plane = pass == 1 ? (lddy >> 1) : 0;
            a_plane = pass == 2 ? (lda >> 1) : 0;
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            if (DYR) {
                const int ry = dyrow_lds[step * 32 + y_row[i]];
                const uint16_t* p = ry >= 0 ? dY + (size_t)ry * lddy + n0 + y_c[i] * 8 : g_zero_row;
                glds16(p, st + (wave * NY + i) * 1024);
            } else {
                const int ml = step * 32 + y_row[i];
                const uint16_t* p = (ml < n_rows) ? dY + (size_t)(m_lo + ml) * lddy + y_plane + n0 + y_c[i] * 8 : g_zero_row;
                glds16(p, st + (wave * NY + i) * 1024);
            }
        }
        int rr[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) rr[i] = row_lds[step * 32 + x_row[i]];     // all index reads first: one lgkmcnt wait
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const uint16_t* p = (rr[i] >= 0 && x_off[i] >= 0) ? A + (size_t)rr[i] * lda + a_plane + x_off[i] : g_zero_row;
            glds16(p, st + Y_BYTES + (wave * NX + i) * 1024);
        }
    };

    f32x16 acc[TNW][TKT];
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TKT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // Bias gradient db[n] = sum_m dY[m][n] rides on the matrix pipe: dY^T times a block of ones.  With K <= BKT - 32 the
    // last 32-column tile of the k-wave 3 is pure padding, so its B fragment is replaced by ones (no extra MFMA, no
    // extra registers); otherwise k-wave 0 carries two extra accumulators (TKW == 8 has the registers for it).
    // (TKW == 5: the padding tile, if there is one, is the last tile of the UPPER k half; the lower half's workgroups form no sums.)
    constexpr bool kExtraBias = (TKW == 8 || TKW == 4);
    const bool free_tile = !kExtraBias && K <= KH * BKT - 32;
    const bool bias_free = bslab != nullptr && free_tile && kh == KH - 1 && (wave % WK) == WK - 1;
    const bool bias_extra = bslab != nullptr && kExtraBias && kh == 0 && (wave % WK) == 0;
    const bool bias_valu = bslab != nullptr && !kExtraBias && !free_tile && kh == 0;
    f32x16 accb[kExtraBias ? TNW : 1];
#pragma unroll
    for (int i = 0; i < (kExtraBias ? TNW : 1); ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
    const __bf16 one_bf = (__bf16)1.0f;
    const bfv8 ones = bfv8{one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf};
    float bsum = 0.f;

    const int n_steps = X3 == 1 ? 3 * n_steps1 : n_steps1;
    // transposed-read lane geometry (see tr_frag in gemm_bf16.hip)
    const int li = lane & 15, g = lane >> 4;
    const int q = li >> 2, p4 = li & 3;
    const int cgrp = 16 * (g & 1) + 4 * p4;       // column offset inside a 32-wide operand tile
    const int rbase = 8 * (g >> 1) + q;           // contraction row inside a 16-deep k-step
    // byte offsets of this lane's transposed reads for k-step 0 (k-step 1 = + 16 rows); row & 3 == q for both
    const int sw = q << 2;
    int yoff[TNW], xoff[TKT];
#pragma unroll
    for (int i = 0; i < TNW; ++i) {
        const int col = wn0 + i * 32 + cgrp;
        yoff[i] = rbase * PY + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
    }
#pragma unroll
    for (int j = 0; j < TKT; ++j) {
        const int col = wk0 + j * 32 + cgrp;
        xoff[j] = Y_BYTES + rbase * PX + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
    }

#pragma unroll
    for (int p = 0; p < NSTG - 1; ++p)
        if (p < n_steps) issue(p);
    for (int step = 0; step < n_steps; ++step) {
        MG_STAMP(ta);
        // stage `step` must have landed; up to NSTG - 2 younger stages (NLW LDS-DMA instructions each) may stay in flight
        if (NATIVE) {
            WAIT_VM_BARRIER(0);                   // two slots: nothing younger is in flight when a step starts
        } else if (NSTG == 4 && step + 2 < n_steps) {
            if (NLW == 4) WAIT_VM_BARRIER(8); else WAIT_VM_BARRIER(6);       // (NSTG == 4: the half-width tiles, NLW 4 or 3)
        } else if (step + 1 < n_steps) {
            if (NLW == 6) WAIT_VM_BARRIER(6); else if (NLW == 5) WAIT_VM_BARRIER(5); else if (NLW == 4) WAIT_VM_BARRIER(4); else WAIT_VM_BARRIER(3);
        } else {
            WAIT_VM_BARRIER(0);
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_wait, tb, ta);
#ifdef MG_STAMPS
        if (step == 0) ts1 = tb;
#endif
        if (step + NSTG - 1 < n_steps) issue(step + NSTG - 1);        // refills the stage every wave finished reading before this barrier
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_issue, ta, tb);
        const unsigned char* st = smem + (step % NSTG) * SLOT;
        if constexpr (NATIVE) {
            // (dY hi, A hi), (dY lo, A hi), (dY hi, A lo) from the one slot: the lo planes' images lie one plain stage behind the hi ones
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bfv8 a[2][TNW], b[2][TKT];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int i = 0; i < TNW; ++i) {
                        const unsigned char* ad = st + pl * STAGE + yoff[i] + ks * 16 * PY;
                        const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
                        const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PY));
                        a[pl][i] = bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
#pragma unroll
                    for (int j = 0; j < TKT; ++j) {
                        const unsigned char* ad = st + pl * STAGE + xoff[j] + ks * 16 * PX;
                        const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
                        const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PX));
                        b[pl][j] = bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
                }
#pragma unroll
                for (int i = 0; i < TNW; ++i)
#pragma unroll
                    for (int j = 0; j < TKT; ++j) {
                        acc[i][j] = mg_mfma_32x32x16(a[0][i], b[0][j], acc[i][j]);
                        acc[i][j] = mg_mfma_32x32x16(a[1][i], b[0][j], acc[i][j]);
                        acc[i][j] = mg_mfma_32x32x16(a[0][i], b[1][j], acc[i][j]);
                    }
            }
            continue;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bfv8 a[TNW], b[TKT];
#pragma unroll
            for (int i = 0; i < TNW; ++i) {
                const unsigned char* ad = st + yoff[i] + ks * 16 * PY;
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PY));
                a[i] = bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < TKT; ++j) {
                const unsigned char* ad = st + xoff[j] + ks * 16 * PX;
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PX));
                b[j] = bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            if (bias_free) b[TKT - 1] = ones;
#pragma unroll
            for (int i = 0; i < TNW; ++i)
#pragma unroll
                for (int j = 0; j < TKT; ++j) acc[i][j] = mg_mfma_32x32x16(a[i], b[j], acc[i][j]);
            if constexpr (kExtraBias) {
                if (bias_extra) {
#pragma unroll
                    for (int i = 0; i < TNW; ++i) accb[i] = mg_mfma_32x32x16(a[i], ones, accb[i]);
                }
            }
        }
        if (bias_valu && tid < BNT) {
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
                const int cpos = (tid >> 3) ^ ((r & 3) << 2);
                bsum += mg_bf2f(*reinterpret_cast<const uint16_t*>(st + r * PY + (cpos << 4) + ((tid & 7) << 1)));
            }
        }
    }

#ifdef MG_STAMPS
    MG_STAMP(ts2);
#endif
    const int lr = lane & 31, lh = lane >> 5;
    float* out = slab + (size_t)s * sstride;        // split s: [N*K weights | bias sums behind them when bslab = slab + N*K]
#pragma unroll
    for (int i = 0; i < TNW; ++i) {
#pragma unroll
        for (int j = 0; j < TKT; ++j) {
            const int col = k0 + wk0 + j * 32 + lr;
            if (col >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wn0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < N) out[(size_t)row * K + col] = acc[i][j][r];
            }
        }
    }
    if (bias_valu && tid < BNT && n0 + tid < N) bslab[(size_t)s * sstride + n0 + tid] = bsum;
    if ((bias_free || bias_extra) && lr == 0) {            // every column of the ones-product holds the same sums
#pragma unroll
        for (int i = 0; i < TNW; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + wn0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v = bias_free ? acc[i][TKT - 1][r] : accb[kExtraBias ? i : 0][r];
                if (row < N) bslab[(size_t)s * sstride + row] = v;
            }
        }
    }
#ifdef MG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG_STAMP(ts3);
    MG_STAMP_REAL(tr1);
    const int sb = block_id;
    MG_STAMP_STORE(g_stamps_wg, sb, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_wg, sb, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_wg, sb, wave, lane, 2, ts2);
    MG_STAMP_STORE(g_stamps_wg, sb, wave, lane, 3, ts3);
    MG_STAMP_STORE(g_stamps_wg, sb, wave, lane, 4, tr0);
    MG_STAMP_STORE(g_stamps_wg, sb, wave, lane, 5, tr1);
    MG_STAMP_STORE(g_stamps_wg, sb, wave, lane, 6, sum_wait);
    MG_STAMP_STORE(g_stamps_wg, sb, wave, lane, 7, sum_issue);
#endif
}

template <int TKW, int NW = 1, int X3 = 0>
__global__ __launch_bounds__(512) void wgrad_big_kernel(const uint16_t* __restrict__ dY, int lddy, const uint16_t* __restrict__ A, int lda,
                                                        const int32_t* __restrict__ rows, int64_t M, int N, int K, int m_chunk,
                                                        float* __restrict__ slab, float* __restrict__ bslab, int64_t sstride, int xcd_group) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[WG_BIG_LDS_NW(TKW, NW)];
    wgrad_big_body<TKW, false, NW, X3>(smem, blockIdx.x, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab, sstride, xcd_group);
}

template <int TKW, int NW = 1>
__global__ __launch_bounds__(512) void wgrad_big_rows_kernel(const uint16_t* __restrict__ dY, int lddy, const int32_t* __restrict__ dy_rows,
                                                             const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows, int64_t M,
                                                             int N, int K, int m_chunk, float* __restrict__ slab, float* __restrict__ bslab,
                                                             int64_t sstride, int xcd_group) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[WG_BIG_LDS_NW(TKW, NW) + WG_ROWS_MAX * 4];
    wgrad_big_body<TKW, true, NW>(smem, blockIdx.x, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab, sstride, xcd_group, dy_rows);
}

// Two INDEPENDENT launches of a backward pass in one grid: the weight gradient of a layer (blocks [0, wg_blocks): dW = dY^T A as split-M
// slabs) and the dgrad + sigmoid backward towards the layer below (the blocks behind them: dX = (dY W) * H (1 - H)).  Both read dY; at
// the phone-rate row count of C2 the first fills 96 CUs for 18 us and the second 168 for 16 us, one after the other; as parallel
// branches of the step's HIP graph they were SLOWER (fork and join nodes), as one grid they simply share the chip.  wg_blocks must be a
// multiple of 8 (both tile programs derive the XCD of a block from its id modulo 8).
// With `riders` > 0 the grid ends in that many RIDER blocks (expand_reduce.h): the repeated prediction and the ordered sum of the fused
// tail's slabs, two small jobs of the step's forward that nothing needs before the update - as a launch of their own they cost the
// step 5.8 us plus a kernel boundary; here they start on the CUs the dgrad tiles free first (blocks are dispatched in id order) and
// end about when the weight-gradient blocks do.
template <int TKW, int BN, int X3 = 0>
__global__ __launch_bounds__(512) void wgrad_dgrad_pair_kernel(unsigned wg_blocks, const uint16_t* __restrict__ dY, int lddy,
                                                               const uint16_t* __restrict__ A, int lda, int64_t M, int N, int K, int m_chunk,
                                                               float* __restrict__ slab, float* __restrict__ bslab, int64_t sstride,
                                                               int xcd_group, const uint16_t* __restrict__ WT, int ldwt,
                                                               const uint16_t* __restrict__ H, int ldh, void* __restrict__ dX, int lddx,
                                                               int tiles_m, int tiles_n, unsigned nt_blocks, ExpandReduceArgs xr, int riders,
                                                               float* __restrict__ colsum) {
    constexpr int LDS = WG_BIG_LDS(TKW) > NT_BIG_LDS(BN) ? WG_BIG_LDS(TKW) : NT_BIG_LDS(BN);
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS];
    if (blockIdx.x < wg_blocks)
        wgrad_big_body<TKW, false, 1, (X3 && WG_STAGES(TKW) == 4) ? 2 : X3>(smem, blockIdx.x, dY, lddy, A, lda, nullptr, M, N, K, m_chunk, slab, bslab, sstride,
                                                                            xcd_group);       // (native pair planes where the ring has four slots)
    else if (blockIdx.x < wg_blocks + nt_blocks)
        // C[M, K] = dY[M, N] WT[K, N]^T with the sigmoid-grad epilogue: contraction N, output width K (as mg_linear_dgrad_bf16)
        gemm_nt_big_body<BN, EPI_SIGMOID_GRAD, X3>(smem, blockIdx.x - wg_blocks, dY, lddy, nullptr, M, N, WT, ldwt, K, nullptr, H, ldh, nullptr, dX,
                                                   lddx, tiles_m, tiles_n, 0, colsum);
    else
        mg_expand_reduce_rider<512>(xr, (int)(blockIdx.x - wg_blocks - nt_blocks), riders, smem);
}


// ---------------------------------------------------------------------------------------------------------------------
// Launch helpers used by the entry points in gemm_bf16.hip.  Each returns 1 if it launched, 0 if the shape does not
// qualify (the caller then uses the generic 128 x 128 kernels), negative on error.
// ---------------------------------------------------------------------------------------------------------------------
static bool big16(const void* p) { return ((uintptr_t)p % 16) == 0; }

int mg_try_nt_big(const uint16_t* A, int lda, const int32_t* rows, int64_t M, int K, const uint16_t* Bm, int ldb, int N,
                  const float* bias, const uint16_t* H, int ldh, const int32_t* h_rows, void* C, int ldc, int c_f32, int epi,
                  hipStream_t st) {
    if (M < 2048 || lda % 64 != 0 || ldb % 64 != 0 || lda > MG_ZERO_ELEMS - 64 || ldb > MG_ZERO_ELEMS - 64) return 0;
    if (N % 128 != 0 || ldc < N || ldc % 8 != 0 || !big16(A) || !big16(Bm) || !big16(C)) return 0;
    if (epi == EPI_SIGMOID_GRAD && (!H || ldh % 8 != 0 || ldh < N)) return 0;
    if (ldc != N || lda < (K + 63) / 64 * 64 || ldb < (K + 63) / 64 * 64) return 0;
    // Tile width: 256 where N allows it - unless that leaves most of the chip idle.  At the phone-rate row counts of the recurrent
    // models (RNN_SPSS at C4 / C5: 6,144 table rows = 24 M tiles) a 512-wide layer is 48 workgroups of 256 x 256 on 256 CUs, each a
    // chain of K / 32 dependent k-steps; 128-wide tiles are twice the workgroups at the same chain length.
    // MG_TUNE_AB (A/B): 97 = 256-wide whenever N allows (the rule before), 99 = 128-wide always, 98 = leave row counts under 32,768
    // to the 128 x 128 kernel of gemm_bf16.hip.
    const int ab = g_mg_tuning[MG_TUNE_AB];
    if (ab == 98 && M < 32768) return 0;
    bool wide = (N % 256 == 0);
    if (wide && ab != 97 && (ab == 99 || mg_ceil_div(M, 256) * (N / 256) < 128)) wide = false;
    const int bn = wide ? 256 : 128;
    const int tiles_n = N / bn;
    const int64_t tiles_m = mg_ceil_div(M, 256);
    const int64_t blocks = mg_ceil_div(tiles_m, 8) * 8 * tiles_n;
    if (blocks >= 2147483647LL || tiles_m >= 2147483647LL) return 0;
    // Persistent form: bias / bias + sigmoid with bf16 output, at least one ring of k-tiles per tile, a stores-per-row pattern
    // that needs every store of a row in range (M is arbitrary: rows past the end are skipped per lane).
    const int n_kt = (K + 31) / 32;
    if (!c_f32 && epi != EPI_SIGMOID_GRAD && n_kt >= 5 && M < 2147483647LL && 32 % tiles_n == 0 && g_mg_tuning[MG_TUNE_FORM] != 6) {
        int64_t g = 256;                                             // one resident workgroup per CU
        while (mg_ceil_div(blocks, g) > NTP_MAX_TILES(bn)) g += 256;    // more tiles than a workgroup parks rows for: more groups
        if (g > blocks) g = mg_ceil_div(blocks, 8 * tiles_n) * 8 * tiles_n;
        dim3 pgrid((unsigned)g), pblock(512);
#define LAUNCH_NTP(BN_, EPI_, STAG_, BK_) hipLaunchKernelGGL((gemm_nt_persist_kernel<BN_, EPI_, STAG_, BK_>), pgrid, pblock, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, (uint16_t*)C, ldc, (int)tiles_m, tiles_n, probe)
        const bool deep = g_mg_tuning[MG_TUNE_FORM] != 3;        // 128-wide tile: 64-deep stages (whole 128-byte lines per DMA row); 3 = 32-deep
#ifdef MG_EXPERIMENTS
        // Lab builds only (make lab / diag): measured alternatives of the square tile and its timing probes (results garbage for the
        // probes) - MG_TUNE_FORM 4 = wave groups half a k-step apart, 2 = half-step software pipeline, 5 / 11 = DMA pieces spread between
        // the MFMAs, 9 / 10 / 12 / 13 = role-split probes, 32 + mask / 256 + mask = parts of the k-step switched off.
        const int form = g_mg_tuning[MG_TUNE_FORM];
        const int probe = form >= 256 ? form - 256 : form >= 32 ? form - 32 : 0;
        if (wide && (form == 5 || form == 11)) {
            const int pr = form == 11 ? 32 : 0;
            if (epi == EPI_BIAS) hipLaunchKernelGGL((gemm_nt_persist_kernel<256, EPI_BIAS, false, 32, 2>), pgrid, pblock, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, (uint16_t*)C, ldc, (int)tiles_m, tiles_n, pr);
            else hipLaunchKernelGGL((gemm_nt_persist_kernel<256, EPI_BIAS_SIGMOID, false, 32, 2>), pgrid, pblock, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, (uint16_t*)C, ldc, (int)tiles_m, tiles_n, pr);
            return 1;
        }
        if (wide && (form == 9 || form == 10 || form == 12 || form == 13)) {
            hipLaunchKernelGGL((gemm_nt_persist_kernel<256, EPI_BIAS_SIGMOID, false, 32, 5>), pgrid, pblock, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, (uint16_t*)C, ldc, (int)tiles_m, tiles_n, form == 9 ? 16 : form == 10 ? 24 : form == 12 ? 23 : 23 + 64);
            return 1;
        }
        if (wide && form == 2) {
            if (epi == EPI_BIAS) hipLaunchKernelGGL((gemm_nt_persist_kernel<256, EPI_BIAS, false, 32, 1>), pgrid, pblock, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, (uint16_t*)C, ldc, (int)tiles_m, tiles_n, probe);
            else hipLaunchKernelGGL((gemm_nt_persist_kernel<256, EPI_BIAS_SIGMOID, false, 32, 1>), pgrid, pblock, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, (uint16_t*)C, ldc, (int)tiles_m, tiles_n, probe);
            return 1;
        }
        if (wide && form == 4) {
            if (epi == EPI_BIAS) LAUNCH_NTP(256, EPI_BIAS, true, 32); else LAUNCH_NTP(256, EPI_BIAS_SIGMOID, true, 32);
            return 1;
        }
#else
        const int probe = 0;
#endif
        if (wide) {
            if (epi == EPI_BIAS) LAUNCH_NTP(256, EPI_BIAS, false, 32); else LAUNCH_NTP(256, EPI_BIAS_SIGMOID, false, 32);
        } else if (deep) {
            if (epi == EPI_BIAS) LAUNCH_NTP(128, EPI_BIAS, false, 64); else LAUNCH_NTP(128, EPI_BIAS_SIGMOID, false, 64);
        } else {
            if (epi == EPI_BIAS) LAUNCH_NTP(128, EPI_BIAS, false, 32); else LAUNCH_NTP(128, EPI_BIAS_SIGMOID, false, 32);
        }
#undef LAUNCH_NTP
        return 1;
    }
    dim3 grid((unsigned)blocks), block(512);
#define LAUNCH_NT(BN_, EPI_) hipLaunchKernelGGL((gemm_nt_big_kernel<BN_, EPI_>), grid, block, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, H, ldh, h_rows, C, ldc, (int)tiles_m, tiles_n, c_f32)
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

// The 128 x 320 tile form (wgrad_big_body<5>: two k halves per n tile): phone-rate row counts of a 640-wide operand with 4+ n tiles
static bool wgrad_ksplit(int64_t M, int N, int lda) {
    if (g_mg_tuning[MG_TUNE_AB] == 92) return false;                                       // 92: A/B, the full-width tiles
    if (lda == 640) return M <= 32768 && N / 128 >= 4;
    // N = 128 at phone-rate row counts only: at M = 256 000 the half-width tiles were 34 us per step SLOWER (0.607 vs 0.573 ms)
    return lda == 512 && N == 128 && M <= 32768 && g_mg_tuning[MG_TUNE_AB] != 93;          // 93: A/B, 128 x 512 tiles for this shape
}

// Plan of the split over M for the wide wgrad: S slabs of m_chunk rows.  Returns 0 if the shape does not qualify.
int mg_wgrad_big_plan(int64_t M, int N, int K, int lda, int lddy, int* S_out, int* m_chunk_out) {
    if (M < 4096 || N % 128 != 0 || lddy < N || lddy % 8 != 0) return 0;
    if (!((lda == 640 && K > 512 && K <= 640) || (lda == 512 && K > 384 && K <= 512))) return 0;
    const int tiles_n = N / 128;
    int64_t S = mg_ceil_div(256, tiles_n);
    // phone-rate shapes (M ~ 2e4 rows): 48 splits x 4 n-tiles = 192 workgroups beat 64 x 4 (45.6 vs 49.7 us with the reduce at
    // M = 21 504, N = 512, K = 600: a third less partial-slab traffic); never more splits than the workspace was sized for
    if (M <= 32768 && tiles_n >= 4) S = 192 / tiles_n;
    if (wgrad_ksplit(M, N, lda)) {
        if (lda == 640) S = 256 / (2 * tiles_n);     // 8 tiles per split: 32 splits fill the chip, a third less slab traffic
        else S = 48;                                 // N = 128: two tiles per split - 96 workgroups with half the slabs of 96 splits
    }
    if (M <= 32768 && tiles_n == 1) S = 96;                  // N = 128 at the same M: 24.2 vs 27.7 us (224 splits write 59 MB of slabs)
    // N = 128 at frame-rate row counts (M = 256 000, K = 512): 192 splits of ~1 334 rows beat one split per CU - a quarter less slab
    // traffic (50 vs 67 MB written and read again by the reduce): 89.5 -> 80 us with the reduce (sweep 144 .. 512, scripts/kbench.py wgrad2)
    if (M > 32768 && tiles_n == 1) S = 192;
    // An n-tile count that does not divide the chip (N = 1536, a GRU's three gates: 12 tiles) at the row count of a whole batch of
    // frames: rounded up to a multiple of 8 the splits overshoot one workgroup per CU (24 x 12 = 288) and the 32 left over run as a second
    // round - 231 us for the 64 000 x 1536 x 512 recurrent weight gradient of C4 where 21 splits (252 workgroups, no XCD grouping)
    // take one round.  The largest split count that still fits one round, unaligned.
    bool one_round = false;
    if (M > 32768 && tiles_n > 1 && !wgrad_ksplit(M, N, lda) && (int64_t)mg_align_up((size_t)S, 8) * tiles_n > 256 && tiles_n <= 128) {
        S = 256 / tiles_n;
        one_round = true;
    }
    // experiments: a split count for every plan (< 1000), for the one-n-tile plans only (1000 + S) or for the plans of 4+ n tiles (2000 + S)
    const int ts = g_mg_tuning[MG_TUNE_WGRAD_SPLITS];
    if (ts > 0 && ts < 1000) S = ts;
    if (ts > 1000 && ts < 2000 && tiles_n == 1) S = ts - 1000;
    if (ts > 2000 && tiles_n >= 4) S = ts - 2000;
    int64_t m_chunk = mg_align_up((size_t)mg_ceil_div(M, S), 32);
    while (m_chunk > WG_ROWS_MAX) {  // row indices of a workgroup's range live in LDS
        S *= 2;
        m_chunk = mg_align_up((size_t)mg_ceil_div(M, S), 32);
    }
    S = mg_ceil_div(M, m_chunk);
    if (!one_round) S = mg_align_up((size_t)S, 8);             // multiple of 8: one split per XCD and round
    if (S > 65535) return 0;
    *S_out = (int)S;
    *m_chunk_out = (int)m_chunk;
    return 1;
}

// dy_rows != nullptr: the dY operand gathered as well (wgrad_big_rows_kernel).  The 128 x 512 tile form only (lda == 512: the 640-wide
// tile leaves no LDS for the second index table); returns 0 without launching for other shapes.
int mg_launch_wgrad_big(const uint16_t* dY, int lddy, const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K,
                        int S, int m_chunk, float* slab, float* bslab, int64_t sstride, hipStream_t st, const int32_t* dy_rows, int x3) {
    if (x3) {
        // pair planes (wgrad_big_body X3): the plan was made for the PLANE widths lda / 2, lddy / 2; no bias sums, no gathers
        if (dy_rows || rows || bslab) return 0;
        const int pa = lda / 2;
        const bool ks = wgrad_ksplit(M, N, pa);
        dim3 grid((unsigned)((N / 128) * (ks ? 2 : 1) * S)), block(512);
        const int xg = (S % 8 == 0) ? 1 : 0;
#define LAUNCH_WG3(TKW_, NW_, X3_) hipLaunchKernelGGL((wgrad_big_kernel<TKW_, NW_, X3_>), grid, block, 0, st, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab, sstride, xg)
        // X3_ = 2: the native form (one slot holds all four planes' rows) where the tile's ring has the LDS for it; MG_TUNE_AB 90 (A/B):
        // three walks over the rows everywhere
        const bool walks = g_mg_tuning[MG_TUNE_AB] == 90;
        if (!ks && pa == 512 && N % 256 == 0) { if (walks) LAUNCH_WG3(4, 2, 1); else LAUNCH_WG3(4, 2, 2); }
        else if (ks && pa == 512) { if (walks) LAUNCH_WG3(4, 1, 1); else LAUNCH_WG3(4, 1, 2); }
        else if (ks) { if (walks) LAUNCH_WG3(5, 1, 1); else LAUNCH_WG3(5, 1, 2); }
        else if (pa == 640 && N % 256 == 0) LAUNCH_WG3(5, 2, 1);
        else if (pa == 640) LAUNCH_WG3(10, 1, 1);
        else LAUNCH_WG3(8, 1, 1);
#undef LAUNCH_WG3
        return 1;
    }
    if (dy_rows) {
        if (lda != 512 || wgrad_ksplit(M, N, lda)) return 0;
        dim3 grid((unsigned)((N / 128) * S)), block(512);
        const int xg = g_mg_tuning[MG_TUNE_WGRAD_ORDER] == 2 && S % 8 == 0 ? 1 : 0;
        if (N % 256 == 0 && g_mg_tuning[MG_TUNE_AB] != 91)            // the square tile (see below): both index tables fill the LDS exactly
            hipLaunchKernelGGL((wgrad_big_rows_kernel<4, 2>), grid, block, 0, st, dY, lddy, dy_rows, A, lda, rows, M, N, K, m_chunk, slab, bslab,
                               sstride, xg);
        else
            hipLaunchKernelGGL((wgrad_big_rows_kernel<8>), grid, block, 0, st, dY, lddy, dy_rows, A, lda, rows, M, N, K, m_chunk, slab, bslab,
                               sstride, xg);
        return 1;
    }
    const bool ksplit = wgrad_ksplit(M, N, lda);
    dim3 grid((unsigned)((N / 128) * (ksplit ? 2 : 1) * S)), block(512);
    // block order: the n tiles of a split on one XCD when X is streamed (no gather: a table read once), n tile fastest when X is a
    // gathered, L2-resident set of rows (measured 15 % slower grouped on the C2 frame-rate shape)
    int xcd_group = (rows == nullptr && S % 8 == 0) ? 1 : 0;
    if (g_mg_tuning[MG_TUNE_WGRAD_ORDER] == 1) xcd_group = 0;
    if (g_mg_tuning[MG_TUNE_WGRAD_ORDER] == 2 && S % 8 == 0) xcd_group = 1;
    // a 512-wide operand with an even number of n tiles: the square 256 x 256 tile (N / 256 n tiles x two k halves = the N / 128 tiles
    // per split the plan counted): 121 vs 143 us at 64 000 x 1536 x 512, 150-160 vs 166-182 at 64 000 x 2048 x 512
    if (!ksplit && lda == 512 && N % 256 == 0 && g_mg_tuning[MG_TUNE_AB] != 91)                       // 91: A/B, the 128 x 512 tiles
        hipLaunchKernelGGL((wgrad_big_kernel<4, 2>), grid, block, 0, st, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab, sstride, xcd_group);
    else if (ksplit && lda == 512)
        hipLaunchKernelGGL((wgrad_big_kernel<4>), grid, block, 0, st, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab, sstride, xcd_group);
    else if (ksplit)
        hipLaunchKernelGGL((wgrad_big_kernel<5>), grid, block, 0, st, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab, sstride, xcd_group);
    else if (lda == 640 && N % 256 == 0 && g_mg_tuning[MG_TUNE_AB] != 91)
        hipLaunchKernelGGL((wgrad_big_kernel<5, 2>), grid, block, 0, st, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab, sstride, xcd_group);
    else if (lda == 640)
        hipLaunchKernelGGL((wgrad_big_kernel<10>), grid, block, 0, st, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab, sstride, xcd_group);
    else
        hipLaunchKernelGGL((wgrad_big_kernel<8>), grid, block, 0, st, dY, lddy, A, lda, rows, M, N, K, m_chunk, slab, bslab, sstride, xcd_group);
    return 1;
}

// The phone-rate front (phone_front.h) in the grid of the persistent wide-tile GEMM C = act(A W^T + b) (phone_front_gemm_kernel).
// Qualifies when the GEMM takes mg_try_nt_big's persistent 256-wide form without a gather and leaves at least 32 CUs idle, and
// the GEMM's LDS block holds a job's scan.  Returns 1 if it launched, 0 otherwise (the caller runs the two launches).
int mg_launch_phone_front_gemm(const PhoneFrontArgs& pf, const uint16_t* A, int lda, int64_t M, int K, const uint16_t* Bm, int ldb, int N,
                               const float* bias, uint16_t* C, int ldc, int epi, hipStream_t st) {
    if (g_mg_tuning[MG_TUNE_AB] == 66 || g_mg_tuning[MG_TUNE_FORM] != 0) return 0;      // A/B: the separate launches
    if (M < 2048 || M >= 2147483647LL || lda % 64 != 0 || ldb % 64 != 0 || lda > MG_ZERO_ELEMS - 64 || ldb > MG_ZERO_ELEMS - 64) return 0;
    if (N % 256 != 0 || ldc != N || !big16(A) || !big16(Bm) || !big16(C)) return 0;
    if (lda < (K + 63) / 64 * 64 || ldb < (K + 63) / 64 * 64 || (K + 31) / 32 < 5) return 0;
    if (epi != EPI_BIAS && epi != EPI_BIAS_SIGMOID) return 0;
    const int tiles_n = N / 256;
    if (32 % tiles_n != 0) return 0;
    constexpr int LDS_INTS = ntp_lds_bytes<256, 32, 0>() / 4;
    // The front's jobs as WAVE jobs where eight of them fit the block's LDS (phone_front.h): 32 rider blocks then work off C2's 512
    // jobs under the GEMM, which can take 192-row tiles - 224 one-tile workgroups instead of 168 (one tile per workgroup and at least
    // 32 CUs left over either way).  MG_TUNE_AB 94: block jobs and 256-row tiles (the form of round 2).
    const int64_t wave_ints_need = phone_front_wave_ints(pf.B, pf.P, pf.T, pf.extra);
    const int wave_ints = (g_mg_tuning[MG_TUNE_AB] != 94 && 8 * wave_ints_need <= LDS_INTS) ? (int)wave_ints_need : 0;
    int bmv = 256;
    if (wave_ints > 0 && g_mg_tuning[MG_TUNE_AB] != 95 && mg_ceil_div(mg_ceil_div(M, 192), 8) * 8 * tiles_n <= 224) bmv = 192;
    const int64_t tiles_m = mg_ceil_div(M, bmv);
    const int64_t blocks = mg_ceil_div(tiles_m, 8) * 8 * tiles_n;
    if (blocks > 224) return 0;
    const int64_t g = mg_ceil_div(blocks, 8 * tiles_n) * 8 * tiles_n;
    const int side = (int)((256 - g) / 8 * 8);
    if (side < 32) return 0;
    if (phone_front_lds_ints(pf.B, pf.P, pf.T, pf.extra) > LDS_INTS) return 0;
    PhoneFrontArgs a = pf;
    a.lds_ints = LDS_INTS;
#ifdef MG_EXPERIMENTS      // lab builds only: 67 = idle riders (the front's outputs are not written), 70 + mask = parts of the rider off
    if (g_mg_tuning[MG_TUNE_AB] == 67) a.lds_ints = 0;
    if (g_mg_tuning[MG_TUNE_AB] >= 70 && g_mg_tuning[MG_TUNE_AB] < 86) a.probe = g_mg_tuning[MG_TUNE_AB] - 70;
#endif
    dim3 grid((unsigned)(side + g)), block(512);
#define LAUNCH_PFG(EPI_, BMV_, BK_)                                                                                                             \
    hipLaunchKernelGGL((phone_front_gemm_kernel<EPI_, BMV_, 0, BK_>), grid, block, 0, st, (unsigned)side, wave_ints, a, A, lda, M, K, Bm, ldb, N, bias, C, \
                       ldc, (int)tiles_m, tiles_n)
    // MG_TUNE_AB 88 (A/B, round 5): 64-deep stages in two 64 KB slots - a DMA row is then a whole 128-byte line, half the line requests
    // of two 32-deep stages for the same bytes.  MEASURED EQUAL (C2 step 0.1041 vs 0.1050 ms, box noise): the front's intake follows the
    // bytes, not the request count; the pair-plane form with 64-deep three-pass stages (87) was 4 % slower than the native 32-deep one.
    const bool deep64 = g_mg_tuning[MG_TUNE_AB] == 88 && lda % 64 == 0 && ldb % 64 == 0;
    if (epi == EPI_BIAS) {
        if (bmv == 192) { if (deep64) LAUNCH_PFG(EPI_BIAS, 192, 64); else LAUNCH_PFG(EPI_BIAS, 192, 32); }
        else { if (deep64) LAUNCH_PFG(EPI_BIAS, 256, 64); else LAUNCH_PFG(EPI_BIAS, 256, 32); }
    } else {
        if (bmv == 192) { if (deep64) LAUNCH_PFG(EPI_BIAS_SIGMOID, 192, 64); else LAUNCH_PFG(EPI_BIAS_SIGMOID, 192, 32); }
        else { if (deep64) LAUNCH_PFG(EPI_BIAS_SIGMOID, 256, 64); else LAUNCH_PFG(EPI_BIAS_SIGMOID, 256, 32); }
    }
#undef LAUNCH_PFG
    return 1;
}

// The layer's weight gradient (slabs) and the dgrad + sigmoid backward towards the layer below in ONE grid (wgrad_dgrad_pair_kernel).
// Qualifies when both are wide-tile shapes with a 128-wide dY (one n tile per split), A = H is the [M, 512] activation table, and
// the blocks of both fit the chip at once: blocks are handed to the XCDs round robin and every workgroup here owns a CU, so an XCD
// (32 CUs) takes ceil(tiles_m / 8) * tiles_n dgrad tiles plus S / 8 splits - S shrinks from the plan's to what is left (fewer, longer
// splits; a 33rd workgroup on an XCD would wait for a whole tile program to finish).  Returns 1 and the split plan, or 0.
// x3 != 0: every operand and dX are pair planes (leading dimensions = two planes); `colsum` then receives the column sums of the fp32 dX
// as 2 ceil(M / 256) slabs of K floats (gemm_nt_big_body X3), and the weight-gradient slabs carry no bias sums.
int mg_launch_wgrad_dgrad_pair(const uint16_t* dY, int lddy, const uint16_t* A, int lda, int64_t M, int N, int K, const uint16_t* WT, int ldwt,
                               uint16_t* dX, int lddx, float* slab, int64_t sstride, size_t slab_floats, int* S_out, hipStream_t st,
                               const ExpandReduceArgs* rider, int x3, float* colsum) {
    if (g_mg_tuning[MG_TUNE_AB] == 65) return 0;                   // A/B: the two launches
    int S = 0, m_chunk = 0;
    const int pl = x3 ? 2 : 1;                                       // planes per row
    if (x3 && (lddy % 2 || lda % 2 || ldwt % 2 || lddx % 2 || !colsum)) return 0;
    const int lda_p = lda / pl, lddy_p = lddy / pl, ldwt_p = ldwt / pl;
    if (N != 128 || K != 512 || lda_p != 512 || lddx != pl * K || mg_wgrad_big_plan(M, N, K, lda_p, lddy_p, &S, &m_chunk) <= 0) return 0;
    if (lddy_p % 64 != 0 || ldwt_p % 64 != 0 || lddy > MG_ZERO_ELEMS - 64 || ldwt > MG_ZERO_ELEMS - 64 || lddy_p < 128 || ldwt_p < 128) return 0;
    if (!big16(dY) || !big16(A) || !big16(WT) || !big16(dX)) return 0;
    const int tiles_n = K / 256;
    const int64_t tiles_m = mg_ceil_div(M, 256);
    // Blocks go to the XCDs round robin and every workgroup here owns a CU: XCD x takes the dgrad tiles of the M tiles x, x + 8, ...
    // and the splits b = x (mod 8), of which the trailing ones are empty when M is not a multiple of the chunk (they write a zero slab
    // and leave).  The largest split count, a multiple of 8 from the plan's down, with at most 32 working blocks on every XCD (a 33rd
    // would wait for a whole tile program to finish).  Measured at C2's M = 21 504: 80 splits; 88 fit too when the splits are handed
    // out from the last one down - the four empty ones then land on the XCDs with a dgrad tile more - and were 2.4 us per step SLOWER
    // (every CU busy to the end, 8 more slabs for the update kernel to sum).
    const bool ksplit = wgrad_ksplit(M, N, lda_p);
    const int kt = ksplit ? 2 : 1;                                    // workgroups per split
    auto fits = [&](int S_, int chunk_) {
        const int64_t s_real = mg_ceil_div(M, chunk_);
        for (int x = 0; x < 8; ++x) {
            const int64_t nt_x = (tiles_m > x ? mg_ceil_div(tiles_m - x, 8) : 0) * tiles_n;
            int64_t wg_x = 0;
            for (int b = x; b < S_; b += 8) wg_x += b < s_real ? kt : 0;        // the k halves of a split sit on one XCD
            if (nt_x + wg_x > 32) return false;
        }
        return true;
    };
    bool found = false;
    for (int cand = S; cand >= 48 / kt && !found; cand -= 8) {
        const int chunk_ = cand == S ? m_chunk : (int)mg_align_up((size_t)mg_ceil_div(M, cand), 32);
        if (chunk_ > WG_ROWS_MAX) break;
        const int S_ = (int)mg_align_up((size_t)mg_ceil_div(M, chunk_), 8);
        if (S_ > S || !fits(S_, chunk_)) continue;
        found = true;
        S = S_;
        m_chunk = chunk_;
    }
    if (!found) return 0;
    if ((size_t)S * (size_t)sstride > slab_floats) return 0;
    const unsigned nt_blocks = (unsigned)(mg_ceil_div(tiles_m, 8) * 8 * tiles_n);
    // riders: 2,048 frames (four per thread) and at most two 16-element chunks of the slab sum each
    ExpandReduceArgs xr{};
    int riders = 0;
    if (rider) {
        xr = *rider;
        const int64_t by_frames = mg_ceil_div(xr.M, 2048), by_chunks = mg_ceil_div(mg_ceil_div(xr.n, 16) - xr.first_chunk, 2);
        riders = (int)(by_frames > by_chunks ? by_frames : by_chunks);
        if (riders < 1) riders = 1;
        if (riders > 1024) riders = 1024;
    }
    if (x3) {
        // pair planes: no bias slabs (the sums of the split values would count hi twice); riders as in the bf16 form
        if (ksplit)
            hipLaunchKernelGGL((wgrad_dgrad_pair_kernel<4, 256, 1>), dim3((unsigned)(2 * S) + nt_blocks + riders), dim3(512), 0, st, (unsigned)(2 * S),
                               dY, lddy, A, lda, M, N, K, m_chunk, slab, (float*)nullptr, sstride, 1, WT, ldwt, A, lda, (void*)dX, lddx, (int)tiles_m,
                               tiles_n, nt_blocks, xr, riders, colsum);
        else
            hipLaunchKernelGGL((wgrad_dgrad_pair_kernel<8, 256, 1>), dim3((unsigned)S + nt_blocks + riders), dim3(512), 0, st, (unsigned)S, dY, lddy,
                               A, lda, M, N, K, m_chunk, slab, (float*)nullptr, sstride, 1, WT, ldwt, A, lda, (void*)dX, lddx, (int)tiles_m, tiles_n,
                               nt_blocks, xr, riders, colsum);
        *S_out = S;
        return 1;
    }
    if (ksplit)
        hipLaunchKernelGGL((wgrad_dgrad_pair_kernel<4, 256>), dim3((unsigned)(2 * S) + nt_blocks + riders), dim3(512), 0, st, (unsigned)(2 * S), dY,
                           lddy, A, lda, M, N, K, m_chunk, slab, slab + (int64_t)N * K, sstride, 1, WT, ldwt, A, lda, (void*)dX, lddx, (int)tiles_m,
                           tiles_n, nt_blocks, xr, riders, (float*)nullptr);
    else
        hipLaunchKernelGGL((wgrad_dgrad_pair_kernel<8, 256>), dim3((unsigned)S + nt_blocks + riders), dim3(512), 0, st, (unsigned)S, dY, lddy, A, lda,
                           M, N, K, m_chunk, slab, slab + (int64_t)N * K, sstride, 1, WT, ldwt, A, lda, (void*)dX, lddx, (int)tiles_m, tiles_n,
                           nt_blocks, xr, riders, (float*)nullptr);
    *S_out = S;
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Pair planes: the fused step of precision mode 'bf16x3' (csrc/split3.hip has the arithmetic).  Every bf16 operand of the step is a
// [hi | lo] pair of planes in one row (row stride = two planes): the phone table (written once by the loader), the weights and
// W2^T (kept current by the update kernel), H1 (written split by the first layer's epilogue), dZ2 (by the fp32 tail), dZ1 (by the
// dgrad epilogue).  The tile programs above run the contraction three times over the plane pairs instead of reading three-plane
// buffers: a third less operand memory, and no split pass anywhere in the step.
// ---------------------------------------------------------------------------------------------------------------------
// C [M, 2 N] = split(act(A W^T + b)), A [M, 2 pa], W [N, 2 pw]; with `pf` the phone-rate front rides in the grid where the GEMM leaves
// CUs idle (*front_rode = 1), as mg_launch_phone_front_gemm.  Returns 1 if it launched, 0 if the shape does not qualify.
int mg_launch_nt_persist_x3(const PhoneFrontArgs* pf, const uint16_t* A, int lda, int64_t M, int K, const uint16_t* Bm, int ldb, int N,
                            const float* bias, uint16_t* C, int ldc, int epi, hipStream_t st, int* front_rode) {
    if (front_rode) *front_rode = 0;
    if (M < 1 || M >= 2147483647LL || lda % 128 != 0 || ldb % 128 != 0 || lda > MG_ZERO_ELEMS - 64 || ldb > MG_ZERO_ELEMS - 64) return 0;
    const int pa = lda / 2, pb = ldb / 2;
    if (N % 256 != 0 || ldc != 2 * N || !big16(A) || !big16(Bm) || !big16(C)) return 0;
    if (pa < (K + 63) / 64 * 64 || pb < (K + 63) / 64 * 64 || (K + 31) / 32 < 2) return 0;
    if (epi != EPI_BIAS && epi != EPI_BIAS_SIGMOID) return 0;
    const int tiles_n = N / 256;
    if (32 % tiles_n != 0) return 0;
    constexpr int LDS_INTS = ntp_lds_bytes<256, 32, 0>() / 4;
    PhoneFrontArgs a{};
    int side = 0, wave_ints = 0, bmv = 256;
    int64_t g = 0, tiles_m = 0;
    if (pf && g_mg_tuning[MG_TUNE_AB] != 66) {
        const int64_t wave_ints_need = phone_front_wave_ints(pf->B, pf->P, pf->T, pf->extra);
        wave_ints = (8 * wave_ints_need <= LDS_INTS) ? (int)wave_ints_need : 0;
        if (wave_ints > 0 && mg_ceil_div(mg_ceil_div(M, 192), 8) * 8 * tiles_n <= 224) bmv = 192;
        tiles_m = mg_ceil_div(M, bmv);
        const int64_t blocks = mg_ceil_div(tiles_m, 8) * 8 * tiles_n;
        g = mg_ceil_div(blocks, 8 * tiles_n) * 8 * tiles_n;
        side = (int)((256 - g) / 8 * 8);
        if (blocks <= 224 && side >= 32 && phone_front_lds_ints(pf->B, pf->P, pf->T, pf->extra) <= LDS_INTS) {
            a = *pf;
            a.lds_ints = LDS_INTS;
            if (front_rode) *front_rode = 1;
        } else {
            side = 0;
        }
    }
    if (side == 0) {
        // the GEMM alone: one resident workgroup per CU walking its tiles (mg_try_nt_big's persistent form)
        bmv = 256;
        wave_ints = 0;
        tiles_m = mg_ceil_div(M, 256);
        const int64_t blocks = mg_ceil_div(tiles_m, 8) * 8 * tiles_n;
        g = 256;
        while (mg_ceil_div(blocks, g) > NTP_MAX_TILES(256)) g += 256;
        if (g > blocks) g = mg_ceil_div(blocks, 8 * tiles_n) * 8 * tiles_n;
    }
    dim3 grid((unsigned)(side + g)), block(512);
#define LAUNCH_PFG3(EPI_, BMV_, X3_, BK_)                                                                                                        \
    hipLaunchKernelGGL((phone_front_gemm_kernel<EPI_, BMV_, X3_, BK_>), grid, block, 0, st, (unsigned)side, wave_ints, a, A, lda, M, K, Bm, ldb, N, bias, C, \
                       ldc, (int)tiles_m, tiles_n)
    // X3_ = 2: the native form (one ring slot holds the 32-deep k-tile of all four planes); MG_TUNE_AB 89 (A/B): three passes over the
    // plane, 32-deep; 87 (A/B): three passes with 64-deep stages (whole-line DMA rows, 1.5x the bytes of the native form)
    const int ab3 = g_mg_tuning[MG_TUNE_AB];
    if (ab3 == 87) {
        if (epi == EPI_BIAS) { if (bmv == 192) LAUNCH_PFG3(EPI_BIAS, 192, 1, 64); else LAUNCH_PFG3(EPI_BIAS, 256, 1, 64); }
        else { if (bmv == 192) LAUNCH_PFG3(EPI_BIAS_SIGMOID, 192, 1, 64); else LAUNCH_PFG3(EPI_BIAS_SIGMOID, 256, 1, 64); }
        return 1;
    }
    const bool passes = ab3 == 89;
    if (epi == EPI_BIAS) {
        if (bmv == 192) { if (passes) LAUNCH_PFG3(EPI_BIAS, 192, 1, 32); else LAUNCH_PFG3(EPI_BIAS, 192, 2, 32); }
        else { if (passes) LAUNCH_PFG3(EPI_BIAS, 256, 1, 32); else LAUNCH_PFG3(EPI_BIAS, 256, 2, 32); }
    } else {
        if (bmv == 192) { if (passes) LAUNCH_PFG3(EPI_BIAS_SIGMOID, 192, 1, 32); else LAUNCH_PFG3(EPI_BIAS_SIGMOID, 192, 2, 32); }
        else { if (passes) LAUNCH_PFG3(EPI_BIAS_SIGMOID, 256, 1, 32); else LAUNCH_PFG3(EPI_BIAS_SIGMOID, 256, 2, 32); }
    }
#undef LAUNCH_PFG3
    return 1;
}

// C fp32 [M, N] = act(A W^T + b) from pair-plane operands (the 128-wide layer in front of the exact-fp32 tail).  1 = launched.
// parts == 3 (epi == EPI_BIAS only): C is [3, M, N] - the three passes' partial sums from three sets of workgroups (gemm_nt_big_body
// X3 == 2); the consumer adds them.
int mg_launch_nt_big_x3_f32(const uint16_t* A, int lda, int64_t M, int K, const uint16_t* Bm, int ldb, int N, const float* bias, float* C,
                            int ldc, int epi, hipStream_t st, int parts) {
    if (parts == 3) {
        if (M < 1 || lda % 128 != 0 || ldb % 128 != 0 || lda > MG_ZERO_ELEMS - 64 || ldb > MG_ZERO_ELEMS - 64 || epi != EPI_BIAS) return 0;
        if (N % 128 != 0 || ldc != N || !big16(A) || !big16(Bm) || !big16(C) || lda / 2 < (K + 63) / 64 * 64 || ldb / 2 < (K + 63) / 64 * 64) return 0;
        const int64_t tm = mg_ceil_div(M, 256);
        const int tn = N / 128;
        const int64_t per_pass = mg_ceil_div(tm, 8) * 8 * tn;
        if (3 * per_pass >= 2147483647LL) return 0;
        hipLaunchKernelGGL((gemm_nt_big_kernel<128, EPI_BIAS, 2>), dim3((unsigned)(3 * per_pass)), dim3(512), 0, st, A, lda, (const int32_t*)nullptr, M, K,
                           Bm, ldb, N, bias, (const uint16_t*)nullptr, 0, (const int32_t*)nullptr, (void*)C, ldc, (int)tm, tn, 1);
        return 1;
    }
    if (parts != 1) return 0;
    if (M < 1 || lda % 128 != 0 || ldb % 128 != 0 || lda > MG_ZERO_ELEMS - 64 || ldb > MG_ZERO_ELEMS - 64) return 0;
    const int pa = lda / 2, pb = ldb / 2;
    if (N % 128 != 0 || ldc != N || !big16(A) || !big16(Bm) || !big16(C)) return 0;
    if (pa < (K + 63) / 64 * 64 || pb < (K + 63) / 64 * 64) return 0;
    if (epi != EPI_BIAS && epi != EPI_BIAS_SIGMOID) return 0;
    const int64_t tiles_m = mg_ceil_div(M, 256);
    const bool wide = N % 256 == 0 && tiles_m * (N / 256) >= 128;
    const int bn = wide ? 256 : 128, tiles_n = N / bn;
    const int64_t blocks = mg_ceil_div(tiles_m, 8) * 8 * tiles_n;
    if (blocks >= 2147483647LL) return 0;
    dim3 grid((unsigned)blocks), block(512);
#define LAUNCH_NT3(BN_, EPI_) hipLaunchKernelGGL((gemm_nt_big_kernel<BN_, EPI_, 1>), grid, block, 0, st, A, lda, (const int32_t*)nullptr, M, K, Bm, ldb, N, bias, (const uint16_t*)nullptr, 0, (const int32_t*)nullptr, (void*)C, ldc, (int)tiles_m, tiles_n, 1)
    if (wide) {
        if (epi == EPI_BIAS) LAUNCH_NT3(256, EPI_BIAS); else LAUNCH_NT3(256, EPI_BIAS_SIGMOID);
    } else {
        if (epi == EPI_BIAS) LAUNCH_NT3(128, EPI_BIAS); else LAUNCH_NT3(128, EPI_BIAS_SIGMOID);
    }
#undef LAUNCH_NT3
    return 1;
}

#ifdef MG_STAMPS
extern "C" int mg_diag_read_stamps_ntp(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_ntp), bytes < sizeof(g_stamps_ntp) ? bytes : sizeof(g_stamps_ntp), 0, hipMemcpyDeviceToHost);
}
extern "C" int mg_diag_read_stamps_wg(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_wg), bytes < sizeof(g_stamps_wg) ? bytes : sizeof(g_stamps_wg), 0, hipMemcpyDeviceToHost);
}
extern "C" int mg_diag_read_stamps_nt(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_nt), bytes < sizeof(g_stamps_nt) ? bytes : sizeof(g_stamps_nt), 0, hipMemcpyDeviceToHost);
}
#endif
