// The README F0Model's layers 2-4 with the masked MSE and their backward down to dZ2, bf16 mode, for LARGE row counts (the reference's
// order of operations: every frame a row, M = 256 000 at BASELINE config C2) - the wide form of f0_l2tail_kernel (l2tail_bf16.hip):
//     H2 = sigmoid(H1 W2^T + b2)   H3 = sigmoid(H2 W3^T + b3)   pred = H3 w4 + b4   L = masked MSE(pred, target, seq_len)
//     dpred, dZ3 = (dpred w4) * H3 (1 - H3), dZ2 = (dZ3 W3) * H2 (1 - H2), dW3 = dZ3^T H2, db3, dW4 = dpred^T H3, db4
// Reference: nn.Linear(512, 128) -> nn.Sigmoid -> nn.Linear(128, 32) -> nn.Sigmoid -> nn.Linear(32, 1) of README.rst:65-73 run by
// SequentialWithRecurrent.forward (morgana/utils.py:401-418), losses.mse (morgana/losses.py:29-51) and their autograd backward.
//
// Why another form (round 5).  f0_l2tail_kernel keeps W2 in LDS (128 KB) and gives ONE wave per SIMD a 32-frame tile from its first
// MFMA to its last store: 85 us at C2 against a 41-48 us stream of H1, every phase of the tile in one in-order instruction stream
// (26 k cycles per tile for ~7 k of VALU, 5.6 k of MFMA).  A second wave per SIMD needs the per-wave state in 256 registers beside
// that LDS image - it does not fit.  Here the operands swap places:
//   * W2 lives in REGISTERS, spread over the workgroup: wave w of eight holds the 16 units [16 w, 16 w + 16) of all 512 inputs as the
//     A fragments of v_mfma_f32_16x16x32_bf16 (64 registers per lane, loaded once per launch);
//   * H1 streams through a six-slot LDS ring by LDS-DMA (64-deep k-tiles of a 128-frame block, 16 KB per slot, whole 128-byte lines),
//     rolling across blocks: five to six tiles are in flight while the tail phase of the block before runs;
//   * phase 1 (all waves): H2^T block = W2 slice x the block's 128 frames, + b2, sigmoid, bf16, into LDS ([128 frames][128 units]);
//   * phase 2 (wave w = frames 16 w .. 16 w + 15 of the block): the tail on MFMAs from LDS - Z3 (16x16x32), the loss terms, dZ3 through a
//     per-wave patch into the A layout, dH2 (16x16x32 against an LDS image of W3^T), dZ2 written over the tile's H2 in place and
//     stored as whole 16-byte row pieces, dW3 on v_mfma_f32_16x16x16_bf16 straight from the C-layout registers (frames as the
//     contraction index: no transposes).
// Two waves per SIMD with <= 256 registers each.  Same arithmetic as f0_l2tail_kernel up to the order of the sums (H2 and dZ3 rounded
// to bf16 in front of the products that consume them, everything else fp32); same slab layout; deterministic; passes the kernel's
// parity test (tests/test_gpu_parity.py::test_l2tail_kernel_vs_numpy_and_unfused_pair, variant 69).
//
// EXPERIMENT (lab builds only; round 5; MG_TUNE_AB = 69).  MEASURED EQUAL TO SLOWER and kept as evidence - the product path is
// f0_l2tail_kernel at every size.  scripts/probe/kbench_l2tail_wide.py at C2 (M = 256 000, stand-alone launches incl. the slab reduce,
// profiles/r5_kbench_l2tail_wide.txt): this form 100.3 us, f0_l2tail_kernel 100.9 us; in the frame-rate step (first version of phase
// 2) 0.547-0.556 against 0.510-0.513 ms.  Its probes: the H1 stream alone 47-49 us (as predicted), phase 1 without the stream 43.5 us
// (the eight waves each read the whole block from LDS: 1 MB per block and CU, the LDS pipe's 8.2 k cycles; + 16 384 sigmoids per block
// = 3 k cycles of every SIMD's transcendental issue; + 9 barriers), phase 2 without the stream 40 us (first version, frames in the
// registers and 2-byte LDS accesses: 50 us), the two together 98 us: the compute side of this pass is TWICE its stream in either
// form - the sigmoids of H2, the tail's VALU work and the barriers do not shrink with the operand layout, and with everything
// resident the second wave per SIMD has the same VALU to share.  Halving the fragment reads (32-unit slices: 128 registers of W2
// per lane) needs dW3 moved out of the per-wave registers (a slice per wave over all frames, two more block barriers); sized at
// ~76 us, compute-bound still - not built.  At phone-rate row counts (21 504 rows: one round of 224 blocks of 96 rows, NFT = 6;
// MG_L2TAIL_WIDE=1 in the lab build) 20.6 us against f0_l2tail_kernel's 17.2: the step 0.1067 against 0.1025 ms - the one block per CU
// pays the prologue (W2 slice, W3 images) and eight barrier steps for 96 rows.
#ifdef MG_EXPERIMENTS
#include "common.h"
#include "slab_reduce.h"

typedef __bf16 lw_bfv8 __attribute__((ext_vector_type(8)));
typedef __bf16 lw_bfv4 __attribute__((ext_vector_type(4)));
typedef __bf16 lw_bfv2 __attribute__((ext_vector_type(2)));
typedef short lw_s4 __attribute__((ext_vector_type(4)));
typedef unsigned int lw_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int lw_u32x2 __attribute__((ext_vector_type(2)));

#define LW_K 512
#define LW_N2 128
#define LW_N3 32
#define LW_ROWS 128                               // frames of a block
#define LW_BK 64                                  // k-tile: 128-byte rows in LDS
#define LW_NKT (LW_K / LW_BK)
#define LW_NS 6                                   // ring slots
#define LW_SLOT (LW_ROWS * 128)                   // 16 KB: chunk c of row r at position c ^ ((r >> 1) & 7)
#define LW_H2S (LW_NS * LW_SLOT)                  // [128 frames][256 B] bf16: chunk c of row r at position c ^ (r & 15)
#define LW_W3S (LW_H2S + LW_ROWS * 256)           // W3 bf16 [32 j][256 B], the same image
#define LW_W3T (LW_W3S + LW_N3 * 256)             // W3^T bf16 [128 k][80 B] (32 j + padding: rows 80 bytes apart)
#define LW_T3 (LW_W3T + LW_N2 * 80)               // per wave: dZ3 of its tile [16 frames][80 B]
#define LW_LDS (LW_T3 + 8 * 16 * 80)              // 159,744 B
#define LW_SLAB (LW_N3 * LW_N2 + LW_N3 + LW_N3 + 2)       // dW3 | db3 | dW4 | db4 | loss: f0_l2tail_kernel's slab
#define LW_SLAB_STRIDE ((LW_SLAB + 3) / 4 * 4)


// PROBE (lab builds only, timing experiments, results garbage): 1 = no phase 2, 2 = no fragment reads / MFMAs in phase 1, 4 = no stream (tiles
// are not fetched), 8 = phase 2 without its dZ2 / pred stores.
// NFT = 16-frame tiles of a block (8: blocks of up to 128 rows; 6: up to 96 - the rows of a one-round launch at phone-rate row counts)
template <int PROBE, int NFT>
__global__ __launch_bounds__(512) void f0_l2tail_wide_kernel(const uint16_t* __restrict__ H1, int ldh1, const uint16_t* __restrict__ W2, int ldw2,
                                                             const float* __restrict__ b2, const float* __restrict__ W3,
                                                             const float* __restrict__ b3, const float* __restrict__ W4,
                                                             const float* __restrict__ b4, const float* __restrict__ target,
                                                             const int64_t* __restrict__ seq_len, int64_t M, int B, int T, float grad_scale,
                                                             float* __restrict__ pred, uint16_t* __restrict__ dZ2, int lddz,
                                                             float* __restrict__ slab, const float* __restrict__ row_weight, int rev,
                                                             int n_blocks, int rows_per) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LW_LDS];
    static_assert(LW_LDS <= 160 * 1024, "one workgroup per CU");
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- one-time: both LDS images of W3 (bf16), this wave's slice of W2 and the small vectors in registers ----------------------------
    for (int i = tid; i < LW_N3 * LW_N2; i += 512) {
        const int j = i >> 7, k = i & 127;
        const uint16_t v = mg_f2bf(W3[i]);
        *reinterpret_cast<uint16_t*>(smem + LW_W3S + j * 256 + ((((k >> 3) ^ (j & 15)) << 4) | ((k & 7) << 1))) = v;
        *reinterpret_cast<uint16_t*>(smem + LW_W3T + k * 80 + j * 2) = v;
    }
    lw_bfv8 w2f[16];                                  // A fragments: unit 16 wave + li, inputs 32 ks + 8 q .. + 7
    {
        const uint16_t* wp = W2 + (size_t)(16 * wave + li) * ldw2 + 8 * q;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) w2f[ks] = *reinterpret_cast<const lw_bfv8*>(wp + 32 * ks);
    }
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(b2 + 16 * wave + 4 * q);      // units 16 wave + 4 q + r of the C layout
    f32x4 b3q[2], w4q[2];                             // units 16 jb + 4 q + r of layer 3 (the C layout of phase 2)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
        b3q[jb] = *reinterpret_cast<const f32x4*>(b3 + 16 * jb + 4 * q);
        w4q[jb] = *reinterpret_cast<const f32x4*>(W4 + 16 * jb + 4 * q);
    }
    const float b4v = b4[0];
    f32x4 dw3acc[2][8];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) dw3acc[jb][kb] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 db3p[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dw4p[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float db4p = 0.f, lossp = 0.f;

    // ---- the stream: tile qt = 8 it + kt is k-tile kt of this workgroup's it-th block, in ring slot qt % LW_NS.  A slot = 16 pieces of
    // 1 KB (8 rows x 128 B); wave w issues pieces w and w + 8.  Tiles past the last block are fetched all the same (the issue count per
    // step stays constant, so the counted waits below hold to the end).
    const int prow = lane >> 3, pch = lane & 7;
    auto block_row0 = [&](int it) -> int64_t {
        const int64_t blk = (int64_t)blockIdx.x + (int64_t)it * gridDim.x;
        if (blk >= n_blocks) return -1;
        return (rev ? (int64_t)n_blocks - 1 - blk : blk) * rows_per;          // blocks of rows_per <= 128 rows (the rest of a slot: zero rows)
    };
    const int n_pieces = (wave + 8 < 2 * NFT) ? 2 : 1;          // pieces w and w + 8 of the 2 NFT pieces that hold the block's tiles
    auto issue = [&](int qt) {
        const int it = qt >> 3, kt = qt & 7;
        const int64_t r0 = block_row0(it);
        // tiles past the last block fetch VALID rows nobody uses (this workgroup's first block again): the issue count stays constant
        const int base = (int)(r0 >= 0 ? r0 : block_row0(0));
        unsigned char* slot = smem + (qt % LW_NS) * LW_SLOT;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i >= n_pieces) break;
            const int piece = wave + 8 * i, row = piece * 8 + prow;
            int m = base + row;                           // rows past the block's own: the next block's (or the last row) - valid, unused
            if (m > (int)M - 1) m = (int)M - 1;
            if (!(PROBE & 4)) mg_glds16(H1 + (size_t)m * ldh1 + kt * LW_BK + ((pch ^ ((row >> 1) & 7)) << 3), slot + piece * 1024);
        }
    };
    const int n_iters = (int)((n_blocks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the one-time loads are in: from here on vmcnt counts the stream only
    __syncthreads();                                       // the W3 images are written
#pragma unroll 1
    for (int qt0 = 0; qt0 < LW_NS; ++qt0) issue(qt0);

    // fragment geometry of phase 1: B operand = H1[frame 16 ft + li][64 kt + 32 ks + 8 q .. + 7]
    int qt = 0;
    for (int it = 0; it < n_iters; ++it) {
        const int64_t row0 = block_row0(it);
        const int n_rows = (int)(row0 + rows_per <= M ? (int64_t)rows_per : M - row0);      // rows of this block (> 0)
        f32x4 acc[NFT];
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) acc[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < LW_NKT; ++kt, ++qt) {
            // tile qt has landed when at most the four tiles behind it are still in flight (two pieces per tile and wave); then every
            // wave is done with tile qt - 1, whose slot takes tile qt + LW_NS - 1 (step 0: issued at the start of the tail before)
            if (n_pieces == 2)
                asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            if (kt > 0) issue(qt + LW_NS - 1);
            const unsigned char* slot = smem + (qt % LW_NS) * LW_SLOT;
#pragma unroll
            for (int ks = 0; ks < ((PROBE & 2) ? 0 : 2); ++ks) {
                lw_bfv8 bf[NFT];
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) {
                    const int row = 16 * ft + li;
                    bf[ft] = *reinterpret_cast<const lw_bfv8*>(slot + row * 128 + (((4 * ks + q) ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int ft = 0; ft < NFT; ++ft) acc[ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[2 * kt + ks], bf[ft], acc[ft], 0, 0, 0);
            }
        }
        // H2 of the block into LDS: this lane holds units 16 wave + 4 q + r (r = 0..3) of frame 16 ft + li
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
            const int row = 16 * ft + li;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = mg_sigmoid_fast(acc[ft][r] + b2v[r]);
            const lw_u32x2 pk = lw_u32x2{__builtin_bit_cast(unsigned int, lw_bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                         __builtin_bit_cast(unsigned int, lw_bfv2{(__bf16)v[2], (__bf16)v[3]})};
            *reinterpret_cast<lw_u32x2*>(smem + LW_H2S + row * 256 + ((((2 * wave + (q >> 1)) ^ (row & 15)) << 4) | ((q & 1) << 3))) = pk;
        }
        __syncthreads();                                   // the block's H2 is complete; every wave is done with the block's last tile
        issue(qt + LW_NS - 1);                             // (qt = first tile of the next block) into the slot of the tile just finished

        // ---- phase 2: this wave's 16 frames.  Every product with the FRAMES as the B side: the lane (li, q) then holds four consecutive
        // units (4 q + r) of ITS frame li, so the per-frame scalars are computed once per lane and every LDS access of the phase is an
        // 8- or 16-byte piece of a row (the first version had the frames in the registers: 2-byte accesses, ~1 500 VALU instructions per
        // tile - 50 us of the launch). ----------------------------------------------------------------------------------------------------
        if (!(PROBE & 1) && 16 * wave < n_rows) {
            int lane_o = lane;                              // (addresses of this phase re-derived from an opaque lane id: not hoisted across phase 1)
            asm volatile("" : "+v"(lane_o));
            const int li = lane_o & 15, q = lane_o >> 4;
            const int r_t = 16 * wave;                      // first row of the tile inside the block
            const int64_t m = row0 + r_t + li;              // this lane's frame
            const bool live = r_t + li < n_rows;            // (rows past the block's end belong to the next block)
            const int64_t mm = live ? m : M - 1;
            const float tg = target[mm];
            float s1, cw, lw;
            if (row_weight) {
                s1 = row_weight[mm];
                cw = 2.f * grad_scale;
                lw = 1.f;
            } else {
                const unsigned mu = (unsigned)mm, bb = mu / (unsigned)T, tt = mu - bb * (unsigned)T;
                int64_t nb = seq_len ? seq_len[bb] : (int64_t)T;
                if (nb > T) nb = T;
                if (nb < 0) nb = 0;
                s1 = (int64_t)tt < nb ? 1.f : 0.f;
                const float inv = 1.f / ((float)nb * (float)B);           // n_b == 0 -> inf: 0 * inf = NaN, as the reference
                cw = 2.f * grad_scale * inv;
                lw = inv;
            }
            // Z3^T = W3 H2^T: A = W3 rows (units of layer 3), B = H2 rows (frames); D: unit 16 jb + 4 q + r of frame li
            f32x4 z3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const lw_bfv8 b = *reinterpret_cast<const lw_bfv8*>(smem + LW_H2S + (r_t + li) * 256 + (((4 * ks + q) ^ li) << 4));
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const lw_bfv8 a = *reinterpret_cast<const lw_bfv8*>(smem + LW_W3S + (16 * jb + li) * 256 + (((4 * ks + q) ^ li) << 4));
                    z3[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, z3[jb], 0, 0, 0);
                }
            }
            f32x4 h3[2];
            float ph = 0.f;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h3[jb][r] = mg_sigmoid_fast(z3[jb][r] + b3q[jb][r]);
                    ph += w4q[jb][r] * h3[jb][r];
                }
            ph += __shfl_xor(ph, 16, 64);
            ph += __shfl_xor(ph, 32, 64);
            const float pv = ph + b4v;
            const float e = pv - tg;
            const float dp = live ? (e * s1) * cw : 0.f;
            if (q == 0) {
                lossp += live ? (e * e * s1) * lw : 0.f;
                db4p += dp;
            }
            unsigned char* t3 = smem + LW_T3 + wave * (16 * 80);
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                float dz[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s_ = h3[jb][r];
                    dz[r] = dp * w4q[jb][r] * s_ * (1.f - s_);
                    db3p[jb][r] += dz[r];
                    dw4p[jb][r] += dp * s_;
                }
                *reinterpret_cast<lw_u32x2*>(t3 + li * 80 + (16 * jb + 4 * q) * 2) =
                    lw_u32x2{__builtin_bit_cast(unsigned int, lw_bfv2{(__bf16)dz[0], (__bf16)dz[1]}),
                             __builtin_bit_cast(unsigned int, lw_bfv2{(__bf16)dz[2], (__bf16)dz[3]})};
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // dH2^T = W3^T dZ3^T: A = W3[j = 8 q .. + 7][k = 16 kb + li] (the W3^T image), B = dZ3[frame li][j = 8 q .. + 7] (the patch);
            // D: unit 16 kb + 4 q + r of frame li.  dW3 += dZ3^T H2 with the frames as contraction index (16x16x16): both operands by
            // transposed reads - A: dZ3[frames 4 q .. + 3][16 jb + li] from the patch, B: H2[frames 4 q .. + 3][16 kb + li].
            const lw_bfv8 bt = *reinterpret_cast<const lw_bfv8*>(t3 + li * 80 + q * 16);
            const int ta = li >> 2, tp = li & 3;            // transposed reads: this lane supplies row 4 q + ta, columns 4 tp .. + 3 of the block
            lw_s4 dza[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
                dza[jb] = __builtin_bit_cast(lw_s4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                                                        (__attribute__((address_space(3))) lw_bfv4*)(t3 + (4 * q + ta) * 80 + (16 * jb + 4 * tp) * 2)));
            const int rt_row = r_t + 4 * q + ta;            // (rt_row & 15 = 4 q + ta)
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                const lw_bfv8 af = *reinterpret_cast<const lw_bfv8*>(smem + LW_W3T + (16 * kb + li) * 80 + q * 16);
                const f32x4 dh = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bt, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const lw_s4 hB = __builtin_bit_cast(lw_s4, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) lw_bfv4*)(
                                                               smem + LW_H2S + rt_row * 256 + ((((2 * kb + (tp >> 1)) ^ (4 * q + ta)) << 4) | ((tp & 1) << 3)))));
                lw_u32x2* hp = reinterpret_cast<lw_u32x2*>(smem + LW_H2S + (r_t + li) * 256 + ((((2 * kb + (q >> 1)) ^ li) << 4) | ((q & 1) << 3)));
                const lw_u32x2 hw = *hp;                    // H2[frame li][16 kb + 4 q .. + 3]
                const float h[4] = {__uint_as_float(hw[0] << 16), __uint_as_float(hw[0] & 0xffff0000u), __uint_as_float(hw[1] << 16),
                                    __uint_as_float(hw[1] & 0xffff0000u)};
                float x[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = dh[r] * h[r] * (1.f - h[r]);
                // dZ2 takes the place of H2 (LDS operations of a wave execute in order: the transposed read above has the old values)
                *hp = lw_u32x2{__builtin_bit_cast(unsigned int, lw_bfv2{(__bf16)x[0], (__bf16)x[1]}),
                               __builtin_bit_cast(unsigned int, lw_bfv2{(__bf16)x[2], (__bf16)x[3]})};
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) dw3acc[jb][kb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dza[jb], hB, dw3acc[jb][kb], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
            // The tiles in flight (and this phase's scalar loads) land BEFORE the stores go out: loads return in order among themselves,
            // but stores may retire out of order with them - with every load older than the stores complete, "at most 8 outstanding" at the
            // next block's steps can only mean stores and younger tiles.
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (live && !(PROBE & 8)) {
                if (q == 0) pred[m] = pv;
                // the tile's dZ2 rows as 16-byte pieces: lane (li, q) stores chunks q, q + 4, q + 8, q + 12 of row li
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = 4 * i + q;
                    const lw_u32x4 v = *reinterpret_cast<const lw_u32x4*>(smem + LW_H2S + (r_t + li) * 256 + ((c ^ li) << 4));
                    *reinterpret_cast<lw_u32x4*>(dZ2 + (size_t)m * lddz + 8 * c) = v;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                       // nobody needs the ring any more: it takes the waves' sums

    // ---- the workgroup's slab: two groups of four waves, each added in wave order, then the groups (as f0_l2tail_x3w_kernel) --------------
    float (*acc_w)[LW_N3 * LW_N2] = reinterpret_cast<float (*)[LW_N3 * LW_N2]>(smem);      // [2][4096]
    float (*acc_s)[68] = reinterpret_cast<float (*)[68]>(smem + 2 * LW_N3 * LW_N2 * 4);     // [8][68]
    const int grp = wave >> 2;
    for (int wv = 0; wv < 4; ++wv) {
        if ((wave & 3) == wv) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int kb = 0; kb < 8; ++kb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (jb * 16 + 4 * q + r) * LW_N2 + kb * 16 + li;
                        acc_w[grp][idx] = (wv == 0 ? 0.f : acc_w[grp][idx]) + dw3acc[jb][kb][r];
                    }
        }
        __syncthreads();
    }
    // the small sums: over the wave's frames (the 16 lanes of a group: DPP row sum), then per wave into LDS
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = mg_row16_sum(db3p[jb][r]), b = mg_row16_sum(dw4p[jb][r]);
            if (li == 0) {
                acc_s[wave][16 * jb + 4 * q + r] = a;
                acc_s[wave][32 + 16 * jb + 4 * q + r] = b;
            }
        }
    {
        const float a = mg_row16_sum(db4p), b = mg_row16_sum(lossp);      // non-zero in the lanes q == 0 only
        if (lane == 0) {
            acc_s[wave][64] = a;
            acc_s[wave][65] = b;
        }
    }
    __syncthreads();
    float* out = slab + (size_t)blockIdx.x * LW_SLAB_STRIDE;
    for (int i = tid; i < LW_N3 * LW_N2; i += 512) out[i] = acc_w[0][i] + acc_w[1][i];
    if (tid < 66) {
        float t_ = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t_ += acc_s[w][tid];
        out[LW_N3 * LW_N2 + tid] = t_;
    }
}

// blocks of 128 frames over at most 256 workgroups (one per CU); the slab count the caller's workspace is sized for
int mg_launch_f0_l2tail_wide(const uint16_t* H1, int ldh1, const uint16_t* W2, int ldw2, const float* b2, const float* W3, const float* b3,
                             const float* W4, const float* b4, const float* target, const int64_t* seq_len, int64_t M, int B, int T,
                             float grad_scale, float* pred, uint16_t* dZ2, int lddz, float* slab, const float* row_weight, int rev, int grid,
                             hipStream_t st) {
    // blocks of at most 128 rows; one round of at most 256 workgroups when the rows allow it (every CU a block: rows_per = ceil(M / 256)
    // rounded up to whole 16-row tiles), whole rounds of 128-row blocks otherwise
    int rows_per = LW_ROWS;
    if (mg_ceil_div(M, LW_ROWS) < grid) rows_per = (int)(mg_ceil_div(mg_ceil_div(M, grid), 16) * 16);
    const int n_blocks = (int)mg_ceil_div(M, rows_per);
    if (n_blocks < grid) grid = n_blocks;
#define LW_LAUNCH_N(P_, N_)                                                                                                                      \
    hipLaunchKernelGGL((f0_l2tail_wide_kernel<P_, N_>), dim3((unsigned)grid), dim3(512), 0, st, H1, ldh1, W2, ldw2, b2, W3, b3, W4, b4, target, seq_len, \
                       M, B, T, grad_scale, pred, dZ2, lddz, slab, row_weight, rev, n_blocks, rows_per)
#define LW_LAUNCH(P_)                \
    do {                             \
        if (rows_per <= 96)          \
            LW_LAUNCH_N(P_, 6);      \
        else                         \
            LW_LAUNCH_N(P_, 8);      \
    } while (0)
    switch (g_mg_tuning[MG_TUNE_AB]) {                  // 101.. = timing probes (results garbage)
        case 101: LW_LAUNCH(1); break;
        case 102: LW_LAUNCH(2); break;
        case 103: LW_LAUNCH(3); break;
        case 104: LW_LAUNCH(4); break;
        case 105: LW_LAUNCH(5); break;
        case 106: LW_LAUNCH(6); break;
        case 108: LW_LAUNCH(8); break;
        default: LW_LAUNCH(0); break;
    }
#undef LW_LAUNCH
#undef LW_LAUNCH_N
    return grid;                                        // slabs written
}
#endif  // MG_EXPERIMENTS
