// The README F0Model's layers 2-4 with the masked MSE and their backward down to dZ2, precision mode 'bf16x3', on table rows that each
// stand for a group of frames - ONE launch, the pair-plane counterpart of f0_l2tail_kernel (l2tail_bf16.hip):
//     Z2 = H1 W2^T + b2   three bf16 MFMA products per fp32 product on [hi | lo] pair planes (csrc/split3.hip), fp32 accumulators, the
//                         128-wide result kept ON CHIP (LDS) - it is needed by nothing but the tail
//     h2 = sigmoid(Z2);  h3 = sigmoid(h2 W3^T + b3);  p = h3 W4^T + b4;  loss = sum_m weight[m] (p[m] - ybar[m])^2         exact fp32
//     dp = 2 weight (p - ybar);  dz3 = dp W4 * h3 (1 - h3);  dW4, db4, dW3 = dz3^T h2, db3;  dZ2 = (dz3 W3) * h2 (1 - h2);  db2 = colsum(dZ2)
// Reference: nn.Linear(512, 128) -> nn.Sigmoid -> nn.Linear(128, 32) -> nn.Sigmoid -> nn.Linear(32, 1) of README.rst:65-73 run by
// SequentialWithRecurrent.forward (morgana/utils.py:401-418), losses.mse (morgana/losses.py:29-51) in its per-phone form
// (csrc/phone_rate.hip) and their autograd backward.
//
// Why one launch (round 5, profiles/r5x3_*): as two launches - mg_linear_fwd_x3_f32 + mg_f0_tail_rows_x3 - the pair took 32 + 30 us of
// the 0.21 ms step at C2's 21 504 table rows; the GEMM is 84 tiles of a 48-step chain that reads H1 in half lines (64-byte row pieces),
// and the tail re-reads Z2 from HBM.  Here a workgroup owns <= 96 consecutive rows: it streams its rows of H1 (both planes, whole
// 128-byte lines, once) and all of W2's pair (256 KB, from the XCD's L2) through a two-slot LDS ring by LDS-DMA, every slot holding the
// 64-deep k-tile of all four planes (three MFMA sets per slot), leaves Z2 + b2 in LDS and runs the exact-fp32 tail of tail_f32.hip on it.
// Phase 2 is f0_tail_rows_f32_kernel<1>'s tile program (same fragment conventions, same slab layout and order of sums).
#include "common.h"
#include "slab_reduce.h"

#define LX_K2 512                           // contraction length of layer 2 (one plane)
#define LX_N2 128
#define LX_N3 32
#define LX_ROWS 96                          // rows of a workgroup's block (three 32-row MFMA tiles)
#define LX_BK 64                            // k-tile: 128-byte LDS rows, whole lines per DMA row
#define LX_SLAB 4292                        // MG_F0_TAIL_X3_SLAB: db2 128 | dW3 4096 | db3 32 | dW4 32 | db4 | loss | 2 pad
#define LX_MAX_BLOCKS 256
#define LX_A_BYTES (LX_ROWS * 128)          // one plane's rows of a k-tile: 12 KB
#define LX_B_BYTES (LX_N2 * 128)            // 16 KB
#define LX_SLOT (2 * LX_A_BYTES + 2 * LX_B_BYTES)      // [A hi | A lo | B hi | B lo] = 56 KB
#define LX_ZLD 132                          // Z2 / h2 row pitch in floats
#define LX_LDS (2 * LX_SLOT)                // 112 KB: the ring; phase 2 lives in the same bytes

typedef __bf16 lx_bfv8 __attribute__((ext_vector_type(8)));


__device__ __forceinline__ f32x4 lx_mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

#define LX_WAIT_VM_BARRIER() asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory")

__global__ __launch_bounds__(256) void f0_l2tail_x3_kernel(const uint16_t* __restrict__ H1, int ldh, const uint16_t* __restrict__ W2, int ldw,
                                                           const float* __restrict__ b2, const float* __restrict__ W3,
                                                           const float* __restrict__ b3, const float* __restrict__ W4,
                                                           const float* __restrict__ b4, const float* __restrict__ ybar,
                                                           const float* __restrict__ weight, int64_t M, int rows_per, int n_blocks,
                                                           float* __restrict__ pred, uint16_t* __restrict__ dZ2, int lddz,
                                                           float* __restrict__ slab) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LX_LDS];
    // phase 2's carve-up of the ring's bytes
    float* zbuf = reinterpret_cast<float*>(smem);                                        // [96][132]: Z2 + b2 of the block
    float (*th)[16][LX_ZLD] = reinterpret_cast<float (*)[16][LX_ZLD]>(smem + 51200);      // per wave: the tile's h2 [row][k]
    float (*t3)[16][LX_N3 + 4] = reinterpret_cast<float (*)[16][LX_N3 + 4]>(smem + 84992);      // per wave: the tile's dz3 [row][j]
    float* acc_w = reinterpret_cast<float*>(smem + 94208);                               // [32 * 128]
    float (*acc_s)[68] = reinterpret_cast<float (*)[68]>(smem + 110592);
    float (*acc_c)[LX_N2] = reinterpret_cast<float (*)[LX_N2]>(smem + 111680);
    static_assert(96 * LX_ZLD * 4 <= 51200 && 51200 + 4 * 16 * LX_ZLD * 4 <= 84992 && 84992 + 4 * 16 * 36 * 4 <= 94208 &&
                      94208 + 16384 <= 110592 && 110592 + 4 * 68 * 4 <= 111680 && 111680 + 4 * LX_N2 * 4 <= LX_LDS,
                  "phase 2 fits the ring's bytes");

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- the tail's resident operands (tail_f32.hip): both layouts of W3 in registers, its partial sums of dW3 ----------------------
    f32x4 w3a[2][8], w3b[8][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) w3a[jt][kb] = *reinterpret_cast<const f32x4*>(W3 + (size_t)(jt * 16 + li) * LX_N2 + 16 * kb + 4 * q);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int e = 0; e < 4; ++e) w3b[kt][jb][e] = W3[(size_t)(16 * jb + 4 * q + e) * LX_N2 + kt * 16 + li];
    const float b3v[2] = {b3[li], b3[16 + li]}, w4v[2] = {W4[li], W4[16 + li]};
    const float b4v = b4[0];
    f32x4 dw3acc[2][8];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) dw3acc[jt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db3p[2] = {0.f, 0.f}, dw4p[2] = {0.f, 0.f}, db4p = 0.f, lossp = 0.f;
    float dbz[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) dbz[kt] = 0.f;

    // ---- phase 1 geometry: LDS-DMA pieces of 1 KB = 8 rows x 128 B; chunk c of tile row r is stored at position c ^ ((r >> 1) & 7) ----
    // (the swizzle of gemm_nt_persist_body's 64-deep stages: conflict free for the 16-lane groups of ds_read_b128).  A: 12 pieces per
    // plane (wave w: pieces w, w + 4, w + 8), B: 16 per plane (w, w + 4, w + 8, w + 12).
    const int prow = lane >> 3, pch = lane & 7;
    const int a_lo = ldh >> 1, b_lo = ldw >> 1;
    const uint16_t* bsrc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = (wave + 4 * g) * 8 + prow;                 // output column n of layer 2
        bsrc[g] = W2 + (size_t)row * ldw + ((pch ^ ((row >> 1) & 7)) << 3);
    }
    // fragment geometry: this wave multiplies the 32 output columns [32 wave, 32 wave + 32) for all three 32-row tiles
    const int lr = lane & 31, lh = lane >> 5;
    int abase[3], asw[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = 32 * i + lr;
        abase[i] = row * 128;
        asw[i] = lh ^ ((row >> 1) & 7);
    }
    const int brow = 32 * wave + lr;
    const int bbase = 2 * LX_A_BYTES + brow * 128, bsw = lh ^ ((brow >> 1) & 7);
    float b2v[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) b2v[g][e] = b2[32 * wave + 8 * g + 4 * lh + e];

    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int64_t row_lo = (int64_t)blk * rows_per;
        const int n_rows = (int)((row_lo + rows_per <= M ? (int64_t)rows_per : M - row_lo));       // > 0: the host sizes n_blocks so
        const uint16_t* asrc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const int row = (wave + 4 * g) * 8 + prow;
            // (rows past the block's end fetch valid rows nobody uses - the row clamped into H1 - not one shared zero row: every
            // workgroup reading the same few lines of one L2 channel is a hot spot, profiles/r5_kbench_l2tail_wide.txt)
            const int64_t mr = row_lo + row < M ? row_lo + row : M - 1;
            asrc[g] = H1 + (size_t)mr * ldh + ((pch ^ ((row >> 1) & 7)) << 3);
        }
        auto issue = [&](int kt) {
            unsigned char* st = smem + (kt & 1) * LX_SLOT;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                for (int g = 0; g < 3; ++g) mg_glds16(asrc[g] + kt * LX_BK + pl * a_lo, st + pl * LX_A_BYTES + (wave + 4 * g) * 1024);
#pragma unroll
                for (int g = 0; g < 4; ++g) mg_glds16(bsrc[g] + kt * LX_BK + pl * b_lo, st + 2 * LX_A_BYTES + pl * LX_B_BYTES + (wave + 4 * g) * 1024);
            }
        };
        f32x16 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        __syncthreads();                              // the previous block's phase 2 is done with the ring's bytes
        issue(0);
        constexpr int N_KT = LX_K2 / LX_BK;
        for (int kt = 0; kt < N_KT; ++kt) {
            LX_WAIT_VM_BARRIER();                     // slot kt landed; every wave is done with the other slot
            if (kt + 1 < N_KT) issue(kt + 1);
            const unsigned char* st = smem + (kt & 1) * LX_SLOT;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                lx_bfv8 ah[3], al[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    ah[i] = *reinterpret_cast<const lx_bfv8*>(st + abase[i] + ((asw[i] ^ (2 * ks)) << 4));
                    al[i] = *reinterpret_cast<const lx_bfv8*>(st + LX_A_BYTES + abase[i] + ((asw[i] ^ (2 * ks)) << 4));
                }
                const lx_bfv8 bh = *reinterpret_cast<const lx_bfv8*>(st + bbase + ((bsw ^ (2 * ks)) << 4));
                const lx_bfv8 bl = *reinterpret_cast<const lx_bfv8*>(st + LX_B_BYTES + bbase + ((bsw ^ (2 * ks)) << 4));
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    // weights as the A operand: the lane holds one output ROW, its registers 4-column groups of it (gemm_nt_big_body)
                    acc[i] = mg_mfma_32x32x16(bh, ah[i], acc[i]);
                    acc[i] = mg_mfma_32x32x16(bl, ah[i], acc[i]);
                    acc[i] = mg_mfma_32x32x16(bh, al[i], acc[i]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave is done with the ring: phase 2 takes its bytes
        // Z2 + b2 into LDS: register 4 g + e of tile i is column 32 wave + 8 g + 4 lh + e of row 32 i + lr
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<f32x4*>(&zbuf[(32 * i + lr) * LX_ZLD + 32 * wave + 8 * g + 4 * lh]) =
                    f32x4{acc[i][4 * g] + b2v[g][0], acc[i][4 * g + 1] + b2v[g][1], acc[i][4 * g + 2] + b2v[g][2], acc[i][4 * g + 3] + b2v[g][3]};
        __syncthreads();

        // ---- phase 2: the exact-fp32 tail on the block's 16-row tiles (tail_f32.hip: phases A-D) ----------------------------------------
        const int tiles = (n_rows + 15) / 16;
        for (int tile = wave; tile < tiles; tile += 4) {
            const int r0 = tile * 16;
            const int64_t row0 = row_lo + r0;
            const bool va = r0 + li < n_rows;
            f32x4 h2a[8];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const f32x4 z = *reinterpret_cast<const f32x4*>(&zbuf[(r0 + li) * LX_ZLD + 16 * kb + 4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) h2a[kb][e] = va ? mg_sigmoid(z[e]) : 0.f;
                *reinterpret_cast<f32x4*>(&th[wave][li][16 * kb + 4 * q]) = h2a[kb];
            }
            f32x4 z3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kb = 0; kb < 8; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int jt = 0; jt < 2; ++jt) z3[jt] = lx_mfma4(h2a[kb][e], w3a[jt][kb][e], z3[jt]);
            const int rc = r0 + 4 * q;               // C layout: row 4 q + r, unit jt 16 + li
            float yb[4], wt[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool v = rc + r < n_rows;
                yb[r] = v ? ybar[row_lo + rc + r] : 0.f;
                wt[r] = v ? weight[row_lo + rc + r] : 0.f;
            }
            f32x4 h3[2], dz3[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h3[jt][r] = mg_sigmoid(z3[jt][r] + b3v[jt]);
            float dp[4];
            f32x4 pv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = mg_row16_sum(w4v[0] * h3[0][r] + w4v[1] * h3[1][r]) + b4v;
                pv[r] = p;
                const float w = wt[r];
                const float d = w > 0.f ? p - yb[r] : 0.f;
                dp[r] = 2.f * w * d;
                if (li == 0) {
                    lossp += (w * d) * d;
                    db4p += dp[r];
                }
            }
            if (li == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rc + r < n_rows) pred[row_lo + rc + r] = pv[r];
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s = h3[jt][r];
                    dz3[jt][r] = dp[r] * w4v[jt] * s * (1.f - s);
                    db3p[jt] += dz3[jt][r];
                    dw4p[jt] += dp[r] * s;
                    t3[wave][4 * q + r][jt * 16 + li] = dz3[jt][r];
                }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x4 dz3a[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) dz3a[jb] = *reinterpret_cast<const f32x4*>(&t3[wave][li][16 * jb + 4 * q]);
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                f32x4 dh = {0.f, 0.f, 0.f, 0.f}, hc;
#pragma unroll
                for (int r = 0; r < 4; ++r) hc[r] = th[wave][4 * q + r][kt * 16 + li];
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dh = lx_mfma4(dz3a[jb][e], w3b[kt][jb][e], dh);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rc + r < n_rows) {
                        // (lo from the ROUNDED product, as mg_f0_tail_rows_x3)
#pragma clang fp contract(off)
                        const float x = dh[r] * hc[r] * (1.f - hc[r]);
                        const uint16_t hi = mg_f2bf(x);
                        uint16_t* dst = dZ2 + (size_t)(row_lo + rc + r) * lddz + kt * 16 + li;
                        dst[0] = hi;
                        dst[lddz >> 1] = mg_f2bf(x - mg_bf2f(hi));
                        dbz[kt] += x;
                    }
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dw3acc[jt][kt] = lx_mfma4(dz3[jt][e], hc[e], dw3acc[jt][kt]);
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the next tile overwrites th / t3
            (void)row0;
        }
    }

    // ---- the workgroup's slab: waves added in wave order (tail_f32.hip) -----------------------------------------------------------------
    __syncthreads();
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (jt * 16 + 4 * q + r) * LX_N2 + kt * 16 + li;
                        acc_w[idx] = (wv == 0 ? 0.f : acc_w[idx]) + dw3acc[jt][kt][r];
                    }
        }
        __syncthreads();
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        float a = db3p[jt], b = dw4p[jt];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (q == 0) {
            acc_s[wave][jt * 16 + li] = a;
            acc_s[wave][32 + jt * 16 + li] = b;
        }
    }
    {
        float a = db4p, b = lossp;                  // non-zero in the lanes li == 0 only
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (lane == 0) {
            acc_s[wave][64] = a;
            acc_s[wave][65] = b;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        float a = dbz[kt];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (q == 0) acc_c[wave][kt * 16 + li] = a;
    }
    __syncthreads();
    float* out = slab + (size_t)blockIdx.x * LX_SLAB;
    if (tid < LX_N2) out[tid] = ((acc_c[0][tid] + acc_c[1][tid]) + acc_c[2][tid]) + acc_c[3][tid];
    out += LX_N2;
    for (int i = tid; i < LX_N3 * LX_N2; i += 256) out[i] = acc_w[i];
    if (tid < 68) out[LX_N3 * LX_N2 + tid] = tid < 66 ? ((acc_s[0][tid] + acc_s[1][tid]) + acc_s[2][tid]) + acc_s[3][tid] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same launch with EIGHT waves per workgroup (two per SIMD, <= 256 registers each): the block's six 16-row tiles of the tail phase
// are then ONE round of waves (four waves took two rounds: the tail phase was ~25 of the launch's 37 us), and phase 1 runs on
// v_mfma_f32_16x16x32_bf16 - 6 x 8 tiles of 16 x 16, wave w owns row tiles 3 (w / 4) .. + 2 and column tiles 2 (w % 4), + 1.
// W3's second layout (the B operand of dh = dz3 W3) is read from an LDS copy of W3 instead of 64 registers; the tile's h2 overwrites its
// Z2 rows in place (a tile's rows belong to one wave).  Same slab layout; the sums of a workgroup's waves are taken in two groups of
// four (each in wave order) and added - another order than the four-wave kernel's, equally fixed.
// ---------------------------------------------------------------------------------------------------------------------------------
#define LXW_W3L_OFF LX_LDS                              // W3 [32][132] f32 behind the ring: it must survive phase 1
#define LXW_LDS (LX_LDS + LX_N3 * LX_ZLD * 4)

__global__ __launch_bounds__(512) void f0_l2tail_x3w_kernel(const uint16_t* __restrict__ H1, int ldh, const uint16_t* __restrict__ W2, int ldw,
                                                            const float* __restrict__ b2, const float* __restrict__ W3,
                                                            const float* __restrict__ b3, const float* __restrict__ W4,
                                                            const float* __restrict__ b4, const float* __restrict__ ybar,
                                                            const float* __restrict__ weight, int64_t M, int rows_per, int n_blocks,
                                                            float* __restrict__ pred, uint16_t* __restrict__ dZ2, int lddz,
                                                            float* __restrict__ slab) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LXW_LDS];
    float* zbuf = reinterpret_cast<float*>(smem);                                        // [96][132]: Z2 + b2, then h2 in place
    float (*t3)[16][LX_N3 + 4] = reinterpret_cast<float (*)[16][LX_N3 + 4]>(smem + 51200);      // per wave: the tile's dz3 [row][j]
    float (*acc_w)[LX_N3 * LX_N2] = reinterpret_cast<float (*)[LX_N3 * LX_N2]>(smem + 69632);    // [2 wave groups][32 * 128]
    float (*acc_s)[68] = reinterpret_cast<float (*)[68]>(smem + 102400);
    float (*acc_c)[LX_N2] = reinterpret_cast<float (*)[LX_N2]>(smem + 104576);
    float* w3l = reinterpret_cast<float*>(smem + LXW_W3L_OFF);
    static_assert(96 * LX_ZLD * 4 <= 51200 && 51200 + 8 * 16 * 36 * 4 <= 69632 && 69632 + 2 * 16384 <= 102400 && 102400 + 8 * 68 * 4 <= 104576 &&
                      104576 + 8 * LX_N2 * 4 <= LX_LDS && LXW_LDS <= 160 * 1024,
                  "phase 2 fits the ring's bytes");

    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    for (int i = tid; i < LX_N3 * LX_N2; i += 512) w3l[(i >> 7) * LX_ZLD + (i & 127)] = W3[i];      // both operand layouts of W3 come from this copy
    const float b3v[2] = {b3[li], b3[16 + li]}, w4v[2] = {W4[li], W4[16 + li]};
    const float b4v = b4[0];
    f32x4 dw3acc[2][8];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) dw3acc[jt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db3p[2] = {0.f, 0.f}, dw4p[2] = {0.f, 0.f}, db4p = 0.f, lossp = 0.f;
    float dbz[8];
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) dbz[kt] = 0.f;

    // ---- phase 1 geometry.  The slot's 56 pieces of 1 KB (8 rows x 128 B) in image order [A hi 12 | A lo 12 | B hi 16 | B lo 16]:
    // wave w issues pieces w, w + 8, ..., w + 48.
    const int prow = lane >> 3, pch = lane & 7;
    const int a_lo = ldh >> 1, b_lo = ldw >> 1;
    int pc_lds[7], pc_row[7], pc_kind[7];                      // LDS byte offset, row of the operand, 0 / 1 = A hi / lo, 2 / 3 = B hi / lo
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int p = wave + 8 * j;
        const int kind = p < 12 ? 0 : p < 24 ? 1 : p < 40 ? 2 : 3;
        const int pin = kind == 0 ? p : kind == 1 ? p - 12 : kind == 2 ? p - 24 : p - 40;
        pc_kind[j] = kind;
        pc_row[j] = pin * 8 + prow;
        pc_lds[j] = (kind == 0 ? 0 : kind == 1 ? LX_A_BYTES : kind == 2 ? 2 * LX_A_BYTES : 2 * LX_A_BYTES + LX_B_BYTES) + pin * 1024;
    }
    const int rt0 = 3 * (wave >> 2), ct0 = 2 * (wave & 3);
    int aoff[3], boff[2];                                      // fragment byte offsets of k-step 0, chunk q: + ((4 ks) ^ ...) handled below
    int asw[3], bsw[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int row = 16 * (rt0 + i) + li;
        aoff[i] = row * 128;
        asw[i] = (row >> 1) & 7;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 16 * (ct0 + j) + li;
        boff[j] = 2 * LX_A_BYTES + row * 128;
        bsw[j] = (row >> 1) & 7;
    }
    f32x4 b2v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b2v[j] = *reinterpret_cast<const f32x4*>(b2 + 16 * (ct0 + j) + 4 * q);
    __syncthreads();                                           // w3l is written

    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int64_t row_lo = (int64_t)blk * rows_per;
        const int n_rows = (int)((row_lo + rows_per <= M ? (int64_t)rows_per : M - row_lo));
        const uint16_t* src[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int row = pc_row[j], sw = (pch ^ ((row >> 1) & 7)) << 3;
            if (pc_kind[j] < 2) {
                // (rows past the block's end fetch valid rows nobody uses, not one shared zero row: see the four-wave kernel)
                const int64_t mr = row_lo + row < M ? row_lo + row : M - 1;
                src[j] = H1 + (size_t)mr * ldh + (pc_kind[j] == 1 ? a_lo : 0) + sw;
            }
            else
                src[j] = W2 + (size_t)row * ldw + (pc_kind[j] == 3 ? b_lo : 0) + sw;
        }
        auto issue = [&](int kt) {
            unsigned char* st = smem + (kt & 1) * LX_SLOT;
#pragma unroll
            for (int j = 0; j < 7; ++j) mg_glds16(src[j] + kt * LX_BK, st + pc_lds[j]);
        };
        f32x4 acc[3][2];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();                              // the previous block's phase 2 is done with the ring's bytes
        issue(0);
        constexpr int N_KT = LX_K2 / LX_BK;
        for (int kt = 0; kt < N_KT; ++kt) {
            LX_WAIT_VM_BARRIER();                     // slot kt landed; every wave is done with the other slot
            if (kt + 1 < N_KT) issue(kt + 1);
            const unsigned char* st = smem + (kt & 1) * LX_SLOT;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                lx_bfv8 ah[3], al[3], bh[2], bl[2];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int o = aoff[i] + (((4 * ks + q) ^ asw[i]) << 4);
                    ah[i] = *reinterpret_cast<const lx_bfv8*>(st + o);
                    al[i] = *reinterpret_cast<const lx_bfv8*>(st + LX_A_BYTES + o);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int o = boff[j] + (((4 * ks + q) ^ bsw[j]) << 4);
                    bh[j] = *reinterpret_cast<const lx_bfv8*>(st + o);
                    bl[j] = *reinterpret_cast<const lx_bfv8*>(st + LX_B_BYTES + o);
                }
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        // weights as the A operand: the lane holds columns 16 ct + 4 q .. + 3 of frame row 16 rt + li
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // every wave is done with the ring: phase 2 takes its bytes
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<f32x4*>(&zbuf[(16 * (rt0 + i) + li) * LX_ZLD + 16 * (ct0 + j) + 4 * q]) = acc[i][j] + b2v[j];
        __syncthreads();

        // ---- phase 2: one 16-row tile per wave (tail_f32.hip: phases A-D; h2 in place of the tile's Z2 rows, W3's B layout from LDS) ---
        const int tiles = (n_rows + 15) / 16;
        for (int tile = wave; tile < tiles; tile += 8) {
            // (lane geometry re-derived from an opaque copy of the lane id: hipcc otherwise hoists every address of this phase out of the
            // block loop, where they stay live across phase 1 - 44 registers of scratch)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int li = lane_o & 15, q = lane_o >> 4;
            const int r0 = tile * 16;
            const bool va = r0 + li < n_rows;
            f32x4 h2a[8];
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                float* zp = &zbuf[(r0 + li) * LX_ZLD + 16 * kb + 4 * q];
                const f32x4 z = *reinterpret_cast<const f32x4*>(zp);
#pragma unroll
                for (int e = 0; e < 4; ++e) h2a[kb][e] = va ? mg_sigmoid(z[e]) : 0.f;
                *reinterpret_cast<f32x4*>(zp) = h2a[kb];
            }
            f32x4 z3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const f32x4 wa0 = *reinterpret_cast<const f32x4*>(&w3l[li * LX_ZLD + 16 * kb + 4 * q]);
                const f32x4 wa1 = *reinterpret_cast<const f32x4*>(&w3l[(16 + li) * LX_ZLD + 16 * kb + 4 * q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    z3[0] = lx_mfma4(h2a[kb][e], wa0[e], z3[0]);
                    z3[1] = lx_mfma4(h2a[kb][e], wa1[e], z3[1]);
                }
            }
            const int rc = r0 + 4 * q;
            float yb[4], wt[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool v = rc + r < n_rows;
                yb[r] = v ? ybar[row_lo + rc + r] : 0.f;
                wt[r] = v ? weight[row_lo + rc + r] : 0.f;
            }
            f32x4 h3[2], dz3[2];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) h3[jt][r] = mg_sigmoid(z3[jt][r] + b3v[jt]);
            float dp[4];
            f32x4 pv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = mg_row16_sum(w4v[0] * h3[0][r] + w4v[1] * h3[1][r]) + b4v;
                pv[r] = p;
                const float w = wt[r];
                const float d = w > 0.f ? p - yb[r] : 0.f;
                dp[r] = 2.f * w * d;
                if (li == 0) {
                    lossp += (w * d) * d;
                    db4p += dp[r];
                }
            }
            if (li == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rc + r < n_rows) pred[row_lo + rc + r] = pv[r];
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s_ = h3[jt][r];
                    dz3[jt][r] = dp[r] * w4v[jt] * s_ * (1.f - s_);
                    db3p[jt] += dz3[jt][r];
                    dw4p[jt] += dp[r] * s_;
                    t3[wave][4 * q + r][jt * 16 + li] = dz3[jt][r];
                }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x4 dz3a[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) dz3a[jb] = *reinterpret_cast<const f32x4*>(&t3[wave][li][16 * jb + 4 * q]);
#pragma unroll
            for (int kt = 0; kt < 8; ++kt) {
                f32x4 dh = {0.f, 0.f, 0.f, 0.f}, hc;
#pragma unroll
                for (int r = 0; r < 4; ++r) hc[r] = zbuf[(r0 + 4 * q + r) * LX_ZLD + kt * 16 + li];
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dh = lx_mfma4(dz3a[jb][e], w3l[(16 * jb + 4 * q + e) * LX_ZLD + kt * 16 + li], dh);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (rc + r < n_rows) {
#pragma clang fp contract(off)
                        const float x = dh[r] * hc[r] * (1.f - hc[r]);
                        const uint16_t hi = mg_f2bf(x);
                        uint16_t* dst = dZ2 + (size_t)(row_lo + rc + r) * lddz + kt * 16 + li;
                        dst[0] = hi;
                        dst[lddz >> 1] = mg_f2bf(x - mg_bf2f(hi));
                        dbz[kt] += x;
                    }
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dw3acc[jt][kt] = lx_mfma4(dz3[jt][e], hc[e], dw3acc[jt][kt]);
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }

    // ---- the workgroup's slab: two groups of four waves, each added in wave order, then the groups ---------------------------------------
    __syncthreads();
    const int grp = wave >> 2;
    for (int wv = 0; wv < 4; ++wv) {
        if ((wave & 3) == wv) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (jt * 16 + 4 * q + r) * LX_N2 + kt * 16 + li;
                        acc_w[grp][idx] = (wv == 0 ? 0.f : acc_w[grp][idx]) + dw3acc[jt][kt][r];
                    }
        }
        __syncthreads();
    }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        float a = db3p[jt], b = dw4p[jt];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (q == 0) {
            acc_s[wave][jt * 16 + li] = a;
            acc_s[wave][32 + jt * 16 + li] = b;
        }
    }
    {
        float a = db4p, b = lossp;
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (lane == 0) {
            acc_s[wave][64] = a;
            acc_s[wave][65] = b;
        }
    }
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
        float a = dbz[kt];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        if (q == 0) acc_c[wave][kt * 16 + li] = a;
    }
    __syncthreads();
    float* out = slab + (size_t)blockIdx.x * LX_SLAB;
    if (tid < LX_N2) {
        float t_ = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t_ += acc_c[w][tid];
        out[tid] = t_;
    }
    out += LX_N2;
    for (int i = tid; i < LX_N3 * LX_N2; i += 512) out[i] = acc_w[0][i] + acc_w[1][i];
    if (tid < 68) {
        float t_ = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t_ += acc_s[w][tid];
        out[LX_N3 * LX_N2 + tid] = tid < 66 ? t_ : 0.f;
    }
}

// rows per block and number of blocks: blocks of <= 96 rows, one round of at most 256 workgroups when the rows allow it (every CU a
// block), whole rounds otherwise
static void lx_plan(int64_t M, int* rows_per, int* n_blocks, int* grid) {
    int64_t nb = mg_ceil_div(M, LX_ROWS);
    if (nb < LX_MAX_BLOCKS) {
        nb = mg_ceil_div(M, 16);
        if (nb > LX_MAX_BLOCKS) nb = LX_MAX_BLOCKS;
    } else {
        nb = mg_ceil_div(nb, LX_MAX_BLOCKS) * LX_MAX_BLOCKS;
    }
    int64_t rp = mg_ceil_div(M, nb);
    nb = mg_ceil_div(M, rp);                          // no empty blocks
    *rows_per = (int)rp;
    *n_blocks = (int)nb;
    *grid = (int)(nb < LX_MAX_BLOCKS ? nb : LX_MAX_BLOCKS);
}

extern "C" {

size_t mg_f0_l2tail_x3_workspace_bytes(int64_t M) {
    if (M <= 0) return 256;
    int rp, nb, grid;
    lx_plan(M, &rp, &nb, &grid);
    return mg_align_up((size_t)grid * LX_SLAB * sizeof(float), 256);
}

int mg_f0_l2tail_x3(const uint16_t* H1, int ldh, const uint16_t* W2, int ldw, const float* b2, const float* W3, const float* b3,
                    const float* W4, const float* b4, const float* ybar, const float* weight, int64_t M, float* pred, uint16_t* dZ2,
                    int lddz, float* grads_out, void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream) {
    MG_CHECK_ARG(H1 && W2 && b2 && W3 && b3 && W4 && b4 && ybar && weight && pred && dZ2 && n_slabs && stride && M > 0 && M < 2147483647LL,
                 "mg_f0_l2tail_x3: bad arguments (M=%lld)", (long long)M);
    MG_CHECK_ARG(ldh == 2 * LX_K2 && ldw == 2 * LX_K2 && lddz >= 2 * LX_N2 && lddz % 16 == 0,
                 "mg_f0_l2tail_x3: pair planes of the 512 -> 128 layer need ldh=%d == ldw=%d == 1024 and lddz=%d two planes of >= 128 (multiple of 16)",
                 ldh, ldw, lddz);
    MG_CHECK_ARG(((uintptr_t)H1 % 16) == 0 && ((uintptr_t)W2 % 16) == 0 && ((uintptr_t)W3 % 16) == 0 && ((uintptr_t)dZ2 % 16) == 0 &&
                     (!grads_out || ((uintptr_t)grads_out % 16) == 0),
                 "mg_f0_l2tail_x3: H1, W2, W3, dZ2 and grads_out must be 16-byte aligned");
    static_assert(LX_SLAB == MG_F0_TAIL_X3_SLAB, "header and kernel disagree");
    if (!workspace || workspace_bytes < mg_f0_l2tail_x3_workspace_bytes(M) || ((uintptr_t)workspace % 16) != 0) {
        mg_set_error("mg_f0_l2tail_x3: 16-byte aligned workspace of %zu bytes needed, got %zu", mg_f0_l2tail_x3_workspace_bytes(M), workspace_bytes);
        return MG_EWORKSPACE;
    }
    int rp, nb, grid;
    lx_plan(M, &rp, &nb, &grid);
    hipStream_t st = (hipStream_t)stream;
    // eight waves per workgroup (one round of tail tiles); MG_TUNE_AB 86 (A/B): the four-wave kernel
    if (g_mg_tuning[MG_TUNE_AB] == 86)
        hipLaunchKernelGGL(f0_l2tail_x3_kernel, dim3((unsigned)grid), dim3(256), 0, st, H1, ldh, W2, ldw, b2, W3, b3, W4, b4, ybar, weight, M, rp, nb,
                           pred, dZ2, lddz, (float*)workspace);
    else
        hipLaunchKernelGGL(f0_l2tail_x3w_kernel, dim3((unsigned)grid), dim3(512), 0, st, H1, ldh, W2, ldw, b2, W3, b3, W4, b4, ybar, weight, M, rp, nb,
                           pred, dZ2, lddz, (float*)workspace);
    MG_CHECK_LAUNCH("mg_f0_l2tail_x3");
    *n_slabs = grid;
    *stride = LX_SLAB;
    if (grads_out) {
        mg_launch_slab_reduce((const float*)workspace, LX_SLAB, LX_SLAB, grid, grads_out, 0, st);
        MG_CHECK_LAUNCH("mg_f0_l2tail_x3/reduce");
    }
    return MG_OK;
}

}  // extern "C"
