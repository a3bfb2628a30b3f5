// Second hidden layer TOGETHER with the fused tail of the README F0Model (README.rst:65-73 layers 2-4 + morgana/losses.py:29-51),
// bf16 throughput mode:
//
//     H2 = sigmoid(H1 W2^T + b2)    H3 = sigmoid(H2 W3^T + b3)    pred = H3 w4 + b4    L = masked MSE(pred, target, seq_len)
//     dpred = dL/dpred   dZ3 = (dpred w4) * H3 (1 - H3)   dZ2 = (dZ3 W3) * H2 (1 - H2)
//     dW3 = dZ3^T H2, db3, dW4 = dpred^T H3, db4
//
// in ONE pass over H1 (M x 512 bf16).  The unfused pair (gemm_nt_persist<128> then f0_tail_kernel, tail_bf16.hip) wrote H2
// (65 MB at C2) and read it back: 68 + 43 us for 40 GFLOP, both kernels HBM / latency bound.  H2 is needed by nothing else - the
// backward of layer 2 wants dZ2 and H1 only - so here it never leaves the registers:
//   * one 256-thread workgroup per CU, ONE wave per SIMD with up to 512 registers; W2 (128 x 512 bf16 = 128 KB) is resident
//     in LDS for the whole launch, 16-byte chunk c of row n at c ^ (n & 15) (conflict free for ds_read_b128's 16-lane groups);
//   * a wave owns 32-frame tiles.  H2^T = W2 . H1^T with the WEIGHTS as the A operand (v_mfma_f32_32x32x16_bf16, 4 blocks of 32
//     units x 32 k-steps): the lane then holds ONE frame, so its slice of H1 comes straight from global memory into the B
//     operand registers - 32 x 16 bytes of the frame's own row, no LDS staging, no transposition - and the next tile's rows are
//     requested while this tile multiplies (two half-tile register buffers of 64 VGPRs: 16-32 KB in flight per wave);
//     the contraction order inside a k-step is permuted so that a lane reads 64 contiguous bytes per 128-byte line
//     (lane half lh takes bytes [64 lh, 64 lh + 64) of each line; W2's fragment addresses follow the same permutation);
//   * the accumulator layout of H2^T (lane = frame, registers = units) IS the B-operand layout of the next product up to a
//     permutation of the contraction index, so Z3^T = W3 . H2^T takes the bf16-rounded H2 registers as they are (W3 fragments
//     pre-permuted in LDS, cdna_hip_programming.md section 3), and H2 (1 - H2) is formed from the same registers;
//   * everything after that is the tail kernel's chain (same arithmetic, same order; tail_bf16.hip): sigmoid, the 32-wide dot
//     with w4, the masked MSE term, dZ3, dH2^T = W3^T . dZ3^T, dZ2 stored as 16-byte row pieces, dW3 += dZ3^T . H2 through
//     ds_read_b64_tr_b16 (the H2 tile goes through a 2 KB per-wave LDS patch in four 32-unit slices).
// Partial sums are reduced lane -> wave -> workgroup in a fixed order and finished by an ordered slab reduce (no atomics).
#include "common.h"
#include <stdlib.h>
#include "slab_reduce.h"

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfv4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define LT_K 512
#define LT_N2 128
#define LT_N3 32
#define LT_SLAB (LT_N3 * LT_N2 + LT_N3 + LT_N3 + 2)      // dW3 | db3 | dW4 | db4 | loss   (the tail kernel's slab)
#define LT_SLAB_STRIDE ((LT_SLAB + 3) / 4 * 4)           // floats between two workgroups' slabs: rows a consumer can read 16 bytes at a time

// LDS map (bytes)
#define LT_W2 0                                          // 128 rows x 1024 B, chunk c of row n at c ^ (n & 15)
#define LT_ZF (LT_N2 * LT_K * 2)                         // A fragments of Z3^T = W3 . H2^T: [8 k-steps][64 lanes][16 B]
#define LT_W3P (LT_ZF + 8 * 64 * 16)                     // A fragments of dH2^T = W3^T . dZ3^T: [4 kt][2 s][64 lanes][16 B]
#define LT_B2 (LT_W3P + 8 * 64 * 16)                     // b2 f32[128]
#define LT_WAVE0 (LT_B2 + LT_N2 * 4)                     // per wave: 2 KB patch (dZ3 tile, then the four H2 slices)
#define LT_WAVE_BYTES 2048
#define LT_LDS (LT_WAVE0 + 4 * LT_WAVE_BYTES)            // 156,160 B: one workgroup per CU

MG_STAMP_DECL(g_stamps_lt);
__device__ unsigned int g_lt_sink[512];                   // where the stores of rows past M go (no branch around a store)

// 16 x 16-bit transposed fragment of a [32 rows][64 B] patch: lane gets column 16 (g & 1) + (lane & 15)... as the 32x32x16 A / B
// operand wants it (row = contraction index).  swz: 16-byte chunk c of row m stored at c ^ ((m >> 2) & 3).
__device__ __forceinline__ bfv8 lt_tr_frag(const unsigned char* tile, int lane, int ks, bool swz) {
    const int i = lane & 15, g = lane >> 4;
    const int q = i >> 2, p = i & 3;
    const int mrow = ks * 16 + 8 * (g >> 1) + q;
    const int col = 16 * (g & 1) + 4 * p;
    const int c = col >> 3, in = (col & 7) << 1;
    const int c_lo = swz ? (c ^ ((mrow >> 2) & 3)) : c;
    const int c_hi = swz ? (c ^ (((mrow + 4) >> 2) & 3)) : c;
    const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(tile + mrow * 64 + ((c_lo << 4) | in)));
    const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(tile + (mrow + 4) * 64 + ((c_hi << 4) | in)));
    return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// PROBE (timing experiments through MG_TUNE_AB, results garbage): 1 = no H1 loads inside the loop, 2 = no tail (steps 2-9),
// 4 = no layer-2 MFMAs, 8 = no sigmoid on H2, 16 = no steps 8-9, 32 = no step 9.  The product kernel is PROBE = 0.
template <int PROBE>
__global__ __launch_bounds__(256, 1) void f0_l2tail_kernel(const uint16_t* __restrict__ H1, int ldh1, const uint16_t* __restrict__ W2,
                                                           int ldw2, const float* __restrict__ b2, const float* __restrict__ W3,
                                                           const float* __restrict__ b3, const float* __restrict__ W4,
                                                           const float* __restrict__ b4, const float* __restrict__ target,
                                                           const int64_t* __restrict__ seq_len, int64_t M, int B, int T,
                                                           float grad_scale, float* __restrict__ pred, uint16_t* __restrict__ dZ2,
                                                           int lddz, float* __restrict__ slab, const float* __restrict__ row_weight, int rev) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LT_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mi = lane & 31, lh = lane >> 5;
#ifdef MG_STAMPS
    unsigned long long ts0, ts1, ts2, ts3, tr0, tr1, ta = 0, tb = 0, tc = 0, tp1 = 0, tp2 = 0, tp3 = 0, tp4 = 0, tp5 = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif

    // The loads of the one-time tables go out first, the first tile's rows and scalars behind them and the LDS-DMA of W2 last (vector
    // memory returns in issue order).
    // The two permuted W3 fragment tables are gathered straight from W3 (fp32 [32][128], 16 KB, L2 resident): per thread two
    // fragments of each table - for a Z3 fragment two runs of four consecutive floats, for a dH2 fragment eight floats of one column.
    // (They were built in LDS from a staged bf16 copy: sixteen loads, a cast pass, a barrier and 32 two-byte LDS reads per thread -
    // 5,000 cycles of the 19,000-cycle prologue, stamps of round 3.)
    f32x4 w3a[2][2];
    float w3b[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f = tid + 256 * i, l = f & 63, r = l & 31, h = l >> 5, g = (f >> 6) & 7;
        // Z3 fragment [st = g][lane l]: element j = W3[r][16 g + 8 (j >> 2) + 4 h + (j & 3)]
        w3a[i][0] = *reinterpret_cast<const f32x4*>(W3 + r * LT_N2 + 16 * g + 4 * h);
        w3a[i][1] = *reinterpret_cast<const f32x4*>(W3 + r * LT_N2 + 16 * g + 8 + 4 * h);
        // dH2 fragment [kt = g >> 1][s = g & 1][lane l]: element j = W3[16 (g & 1) + 8 (j >> 2) + 4 h + (j & 3)][32 (g >> 1) + r]
#pragma unroll
        for (int j = 0; j < 8; ++j) w3b[i][j] = W3[(16 * (g & 1) + 8 * (j >> 2) + 4 * h + (j & 3)) * LT_N2 + 32 * (g >> 1) + r];
    }
    const float b2v = tid < LT_N2 ? b2[tid] : 0.f;

    unsigned char* patch = smem + LT_WAVE0 + wave * LT_WAVE_BYTES;

    // register r of a C^T tile <-> unit offset 8 (r >> 2) + 4 lh + (r & 3)
    float b3v[16], w4v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = 8 * (r >> 2) + 4 * lh + (r & 3);
        b3v[r] = b3[n];
        w4v[r] = W4[n];
    }
    const float b4v = b4[0];

    f32x16 acc_w3[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_w3[kt][r] = 0.f;
    float dw4p[16], db3p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dw4p[r] = db3p[r] = 0.f;
    float db4p = 0.f, lossp = 0.f;

    // W2 fragment of MFMA step j = 4 a + c, unit block blk: row 32 blk + mi, 16-byte chunk 8 a + 4 lh + c (this lane's B slot of
    // step j holds H1[frame][64 a + 32 lh + 8 c .. + 7]) at position chunk ^ (mi & 15):  bits 0-3 of the chunk are (8 (a & 1) + c)
    // ^ (4 lh) ^ (mi & 15), bits 4-5 are a >> 1.
    const int w2_lane = LT_W2 + mi * (LT_K * 2);
    const int xl = (mi & 15) ^ (4 * lh);

    const int64_t n_tiles = (M + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    // rev: the tiles are walked from the END of H1 (tile ordinal t -> rows of tile n_tiles - 1 - t): the layer-1 forward wrote H1 in
    // ascending order, so its last rows are the ones still held by L2 / the memory-side cache when this launch starts.
    auto first_row = [&](int64_t t) -> int64_t { return (rev ? (t < n_tiles ? n_tiles - 1 - t : 0) : t) * 32; };
    auto load_half = [&](u32x4 (&dst)[16], int64_t tile, int half) {
        int64_t mm = first_row(tile) + mi;
        if (mm > M - 1) mm = M - 1;                         // rows past the end: any valid row (their results are discarded)
        const uint16_t* hp = H1 + (size_t)mm * ldh1 + 256 * half + 32 * lh;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[4 * a + c] = *reinterpret_cast<const u32x4*>(hp + 64 * a + 8 * c);
    };
    auto load_half_p = [&](u32x4 (&dst)[16], int64_t tile, int half) {
        if (!(PROBE & 1)) load_half(dst, tile, half);
    };

    // Per-frame scalars of the loss, fetched one tile ahead and IN FRONT of that tile's H1 rows: vector memory returns in issue
    // order, so a target load issued inside the tail would wait for the whole prefetch of the next tile.
    //   row_weight mode: s1 = weight;  seq_len mode: s1 = [t < n_b], s2 = (float) n_b
    auto load_scalars = [&](int64_t tile_, float& tg, float& s1, float& s2) {
        int64_t mm = first_row(tile_) + mi;
        if (mm > M - 1) mm = M - 1;
        tg = target[mm];
        if (row_weight) {
            s1 = row_weight[mm];
            s2 = 0.f;
        } else {
            const unsigned mu = (unsigned)mm, b = mu / (unsigned)T, t = mu - b * (unsigned)T;     // M < 2^31 (checked by the launcher)
            int64_t nb = seq_len ? seq_len[b] : (int64_t)T;
            if (nb > T) nb = T;
            if (nb < 0) nb = 0;
            s1 = (int64_t)t < nb ? 1.f : 0.f;
            s2 = (float)nb;
        }
    };

    f32x16 acc[4];
    auto bias_acc = [&]() {                                // register 4 q + e of block blk <-> unit 32 blk + 8 q + 4 lh + e
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(smem + LT_B2 + (32 * blk + 8 * q + 4 * lh) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[blk][4 * q + e] = bq[e];
            }
    };
    float* const pred_sink = reinterpret_cast<float*>(g_lt_sink) + lane;
    uint16_t* const dz_sink = reinterpret_cast<uint16_t*>(g_lt_sink) + lane * 8;

    // dZ2 / pred of a tile are stored at the top of the NEXT iteration, in front of that iteration's prefetch: issued in the tail
    // they queue behind the 32 row loads of the next tile (vector memory issues in order) and the wave stands at the store until
    // those have gone out - the tail then no longer overlaps the stream it was meant to hide.
    u32x4 dzst[8];
    uint16_t* dzp = dz_sink;
    float* predp = pred_sink;
    float pst = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) dzst[i] = u32x4{0u, 0u, 0u, 0u};

    u32x4 ha[16], hb[16];
    float tg_n = 0.f, s1_n = 0.f, s2_n = 0.f;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < n_tiles) {
        load_scalars(tile, tg_n, s1_n, s2_n);
        load_half(ha, tile, 0);
        load_half(hb, tile, 1);
    }
    // ---- one-time: W2 (bf16 as it is), the two permuted W3 fragment tables, b2 ------------------------------------------
    // W2 goes to LDS by LDS-DMA, one row (64 lanes x 16 B) per instruction and 32 rows per wave, all in flight at once with no
    // register staging: lane p of row n fetches the chunk that belongs at position p, c = p ^ (n & 15).  (A copy loop of load ->
    // store pairs cost ~15 us per launch, batches of eight loads in front of eight stores still four trips to L2 - and at the
    // phone-rate row count the whole kernel is one tile per wave behind this prologue.)
    {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int n = wave * 32 + i;
            mg_glds16(W2 + (size_t)n * ldw2 + ((lane ^ (n & 15)) << 3), smem + LT_W2 + n * (LT_K * 2));
        }
        MG_STAMP(tp1);
        if (tid < LT_N2) *reinterpret_cast<float*>(smem + LT_B2 + tid * 4) = b2v;
        MG_STAMP(tp2);
        MG_STAMP(tp3);
        // fragment f = tid + 256 i -> Z3 table [st = 2 blk + s][lane]; fragment 512 + f -> dH2 table [kt][s][lane]
        typedef __bf16 bfv2_ __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 256 * i;
            auto pk = [](float a, float b) { return __builtin_bit_cast(unsigned int, bfv2_{(__bf16)a, (__bf16)b}); };
            *reinterpret_cast<u32x4*>(smem + LT_ZF + f * 16) =
                u32x4{pk(w3a[i][0][0], w3a[i][0][1]), pk(w3a[i][0][2], w3a[i][0][3]), pk(w3a[i][1][0], w3a[i][1][1]), pk(w3a[i][1][2], w3a[i][1][3])};
            *reinterpret_cast<u32x4*>(smem + LT_ZF + (512 + f) * 16) =
                u32x4{pk(w3b[i][0], w3b[i][1]), pk(w3b[i][2], w3b[i][3]), pk(w3b[i][4], w3b[i][5]), pk(w3b[i][6], w3b[i][7])};
        }
        MG_STAMP(tp4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows of W2 have landed (the compiler does not count the DMAs)
        MG_STAMP(tp5);
        __syncthreads();                                   // W2, the fragment tables and b2 are in place for every wave
    }

    bias_acc();
    MG_STAMP(ts1);

    for (; tile < n_tiles; tile += stride) {
#ifdef MG_STAMPS
        if (ta == 0) MG_STAMP(ta);
#endif
        const int64_t m = first_row(tile) + mi;
        const bool live = m < M;
        const float tg = tg_n, s1 = s1_n, s2 = s2_n;
        load_scalars(tile + stride, tg_n, s1_n, s2_n);
        if (!(PROBE & 64)) {
            *predp = pst;
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(dzp + 32 * (i >> 1) + 16 * (i & 1)) = dzst[i];
            __builtin_amdgcn_sched_barrier(0);
        }

        // (1) H2^T = W2 . H1^T + b2 (the accumulators start from the bias, loaded at the end of the previous tile)
        // W2 fragments are read LT_AHEAD MFMA groups (4 MFMAs = one k-step over the four unit blocks) ahead of their use and the order
        // is pinned: left alone hipcc puts every ds_read_b128 right in front of its MFMA with an lgkmcnt(0) between them - 128 exposed
        // LDS round trips per tile on a wave that has no partner on its SIMD.
        auto w2frag = [&](int j, int blk) -> bfv8 {
            const int a = j >> 2, c = j & 3;
            return *reinterpret_cast<const bfv8*>(smem + w2_lane + blk * 32 * (LT_K * 2) + (a >> 1) * 256 + ((xl ^ (8 * (a & 1) + c)) << 4));
        };
        constexpr int LT_AHEAD = 2;
        bfv8 wf[LT_AHEAD + 1][4];
#pragma unroll
        for (int j = 0; j < LT_AHEAD; ++j) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) wf[j][blk] = w2frag(j, blk);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        }
        const int64_t next = tile + stride;                  // past the last tile: rows clamp to M - 1, the values are never used
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            if (j + LT_AHEAD < 32) {
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) wf[(j + LT_AHEAD) % (LT_AHEAD + 1)][blk] = w2frag(j + LT_AHEAD, blk);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
            const u32x4 hv = j < 16 ? ha[j & 15] : hb[j & 15];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                if (PROBE & 4) {
                    asm volatile("" ::"v"(wf[j % (LT_AHEAD + 1)][blk]), "v"(hv));
                    continue;
                }
                acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j % (LT_AHEAD + 1)][blk], __builtin_bit_cast(bfv8, hv), acc[blk], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (PROBE & 256) {
                // EXPERIMENT (lab builds, MG_TUNE_AB = 68; results valid): rolling prefetch - the register step j has just read takes the
                // NEXT tile's step j at once, so that the row loads go out all through the MFMA phase instead of as two bulk issues
                // at j = 15 and j = 31 (the probes say the phases add: loads 25 + MFMAs 9 + tail 29 us of a 92 us launch).  MEASURED
                // SLOWER: 99.9 against 89.8 us in kbench_l2tail.py, the C2 frame-rate step 0.5150-0.5158 against 0.5126-0.5137 ms
                int64_t mm = first_row(next) + mi;
                if (mm > M - 1) mm = M - 1;
                const uint16_t* hp = H1 + (size_t)mm * ldh1 + 256 * (j >> 4) + 32 * lh + 64 * ((j & 15) >> 2) + 8 * (j & 3);
                if (j < 16) ha[j & 15] = *reinterpret_cast<const u32x4*>(hp);
                else hb[j & 15] = *reinterpret_cast<const u32x4*>(hp);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            } else if (j == 15) {                            // the registers just consumed take the next tile's first half
                load_half_p(ha, next, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 16, 0);
            }
        }
#ifdef MG_STAMPS
        if (tb == 0) MG_STAMP(tb);
#endif
        if (!(PROBE & 256)) load_half_p(hb, next, 1);
        if (PROBE & 2) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(acc[blk][r]));
            continue;
        }

        // (2) sigmoid, bf16: hq[2 blk + s] = registers 8 s .. 8 s + 7 of block blk = B fragment of Z3's k-step 2 blk + s.
        //     The W3 fragments of step 3 are requested first: their LDS latency passes under the sigmoids.
        bfv8 zf[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) zf[st] = *reinterpret_cast<const bfv8*>(smem + LT_ZF + (st * 64 + lane) * 16);
        u32x4 hq[8];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            unsigned int w[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (PROBE & 8) ? acc[blk][4 * q + e] : mg_sigmoid_fast(acc[blk][4 * q + e]);
                w[2 * q] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]});
                w[2 * q + 1] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]});
            }
            hq[2 * blk] = u32x4{w[0], w[1], w[2], w[3]};
            hq[2 * blk + 1] = u32x4{w[4], w[5], w[6], w[7]};
        }

        // (3) Z3^T = W3 . H2^T; the W3^T fragments of step 8 are requested behind it
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 8; ++st) z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf[st], __builtin_bit_cast(bfv8, hq[st]), z, 0, 0, 0);
        bfv8 w3p[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) w3p[st] = *reinterpret_cast<const bfv8*>(smem + LT_W3P + (st * 64 + lane) * 16);

        // (4) sigmoid, prediction
        float h3[16];
        float ph = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            h3[r] = mg_sigmoid_fast(z[r] + b3v[r]);
            ph += h3[r] * w4v[r];
        }
        // ph of the other lane half (the other 16 units of this frame): v_permlane32_swap instead of a trip through LDS
        const auto phs = __builtin_amdgcn_permlane32_swap(__float_as_uint(ph), __float_as_uint(ph), false, false);
        const float p = (__uint_as_float(phs[0]) + __uint_as_float(phs[1])) + b4v;

        // (5) masked MSE of this frame: tail_bf16.hip (5) without branches.  row_weight mode: a = weight, c = 2 g, l = 1;
        //     seq_len mode: a = [t < n_b], inv = 1 / (n_b B) (n_b == 0 -> inf: 0 * inf = NaN, as the reference), c = 2 g inv, l = inv;
        //     dpred = (e a) c, loss term = (e e a) l: the same products in the same order as the branches of the tail kernel
        const float inv = 1.f / (s2 * (float)B);
        const float cw = row_weight ? 2.f * grad_scale : 2.f * grad_scale * inv;
        const float lw = row_weight ? 1.f : inv;
        const float e = p - tg;
        const float dpred = live ? (e * s1) * cw : 0.f;
        if (lh == 0) {
            lossp += live ? (e * e * s1) * lw : 0.f;
            db4p += dpred;
        }
        predp = (live && lh == 0) ? pred + m : pred_sink;
        pst = p;

        // (6) backward of layer 4 and of the layer-3 sigmoid
        float dz3[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dz3[r] = dpred * w4v[r] * h3[r] * (1.f - h3[r]);
            dw4p[r] += dpred * h3[r];
            db3p[r] += dz3[r];
        }
        // (7) dZ3 as bf16: registers 8 s .. 8 s + 7 are the B fragment of k-step s; the row-major copy goes to the patch
        u32x4 dzf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            unsigned int w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[q] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)dz3[8 * s + 2 * q], (__bf16)dz3[8 * s + 2 * q + 1]});
            dzf[s] = u32x4{w[0], w[1], w[2], w[3]};
            *reinterpret_cast<u32x2*>(patch + mi * 64 + (16 * s + 4 * lh) * 2) = u32x2{w[0], w[1]};
            *reinterpret_cast<u32x2*>(patch + mi * 64 + (16 * s + 8 + 4 * lh) * 2) = u32x2{w[2], w[3]};
        }
        bfv8 dzt[2];                                         // dZ3^T fragments of the dW3 product (contraction over frames)
#pragma unroll
        for (int s = 0; s < 2; ++s) dzt[s] = lt_tr_frag(patch, lane, s, false);

        // (8) per 32-unit block kt of layer 2: dH2^T = W3^T . dZ3^T, dZ2 = dH2 * H2 (1 - H2) written as 16-byte row pieces;
        // (9) dW3[:, block kt] += dZ3^T . H2[:, block kt] with the H2 slice read back transposed from the patch (LDS operations of a
        //     wave execute in order: the slice store cannot overtake the transposed reads of the previous occupant)
        if (PROBE & 16) {
            asm volatile("" ::"v"(dzt[0]), "v"(dzt[1]), "v"(dzf[0]), "v"(dzf[1]));
            continue;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3p[kt * 2 + s], __builtin_bit_cast(bfv8, dzf[s]), d, 0, 0, 0);
            }
            unsigned int pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // H2 of units 32 kt + 8 g + 4 lh + e = registers 4 g + e of block kt = words 2 (g & 1), 2 (g & 1) + 1 of hq[2 kt + (g >> 1)]
                const unsigned int w0 = hq[2 * kt + (g >> 1)][2 * (g & 1)], w1 = hq[2 * kt + (g >> 1)][2 * (g & 1) + 1];
                const float h[4] = {__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16),
                                    __uint_as_float(w1 & 0xffff0000u)};
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = d[4 * g + e] * h[e] * (1.f - h[e]);
                pk[g][0] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]});
                pk[g][1] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]});
                if (!(PROBE & 32)) *reinterpret_cast<u32x2*>(patch + mi * 64 + ((g ^ ((mi >> 2) & 3)) << 4) + 8 * lh) = u32x2{w0, w1};
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                dzst[2 * kt + (g >> 1)] = u32x4{r0[0], r1[0], r0[1], r1[1]};          // row m, columns 32 kt + 8 g + 8 lh ..
            }
#pragma unroll
            for (int s = 0; s < (PROBE & 32 ? 0 : 2); ++s) {
                const bfv8 bq = lt_tr_frag(patch, lane, s, true);
                acc_w3[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dzt[s], bq, acc_w3[kt], 0, 0, 0);
            }
        }
        dzp = live ? dZ2 + (size_t)m * lddz + 8 * lh : dz_sink;
        bias_acc();                                         // the next tile's accumulators (their registers were free until here)
#ifdef MG_STAMPS
        if (tc == 0) MG_STAMP(tc);
#endif
    }
    MG_STAMP(ts2);
    *predp = pst;                                          // the last tile's results
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4*>(dzp + 32 * (i >> 1) + 16 * (i & 1)) = dzst[i];

    // ---- reduction: lanes (frames) -> wave -> workgroup, fixed order (as tail_bf16.hip) -----------------------------------
    // over the 32 lanes of a half: the halves of 16 through the crossbar, the rest on the DPP path (mg_row16_sum: bit for bit the xor
    // butterfly 8, 4, 2, 1 - 128 of the 160 cross-lane moves of a wave leave the LDS pipe)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dw4p[r] += __shfl_xor(dw4p[r], 16, 64);
        db3p[r] += __shfl_xor(db3p[r], 16, 64);
        dw4p[r] = mg_row16_sum(dw4p[r]);
        db3p[r] = mg_row16_sum(db3p[r]);
    }
    db4p = mg_wave_sum(db4p);
    lossp = mg_wave_sum(lossp);

    __syncthreads();                                   // every wave is done with W2: its region takes the four waves' sums
    float* red = reinterpret_cast<float*>(smem + LT_W2);               // [4 waves][LT_SLAB]
    float* mine = red + wave * LT_SLAB;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * lh;             // plain C layout: row = hidden unit n
            const int k = 32 * kt + mi;                                //                 col = input feature k
            mine[n * LT_N2 + k] = acc_w3[kt][r];
        }
    if (mi == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 8 * (r >> 2) + 4 * lh + (r & 3);
            mine[LT_N3 * LT_N2 + n] = db3p[r];
            mine[LT_N3 * LT_N2 + LT_N3 + n] = dw4p[r];
        }
    }
    if (lane == 0) {
        mine[LT_N3 * LT_N2 + 2 * LT_N3] = db4p;
        mine[LT_N3 * LT_N2 + 2 * LT_N3 + 1] = lossp;
    }
    __syncthreads();
    float* out = slab + (size_t)blockIdx.x * LT_SLAB_STRIDE;
    for (int e = tid; e < LT_SLAB; e += 256) out[e] = ((red[e] + red[LT_SLAB + e]) + red[2 * LT_SLAB + e]) + red[3 * LT_SLAB + e];
#ifdef MG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG_STAMP(ts3);
    MG_STAMP_REAL(tr1);
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 2, ts2);
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 3, ts3);
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 4, tr0);
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 5, tr1);
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 6, tb - ta);      // first tile: loop top -> layer-2 MFMAs issued
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 7, tc - tb);      // first tile: the tail (sigmoid .. dW3)
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 8, tp1 - ts0);    // prologue: every load and the W2 DMA issued
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 9, tp2 - tp1);    //           W3 landed, cast, staged
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 10, tp3 - tp2);   //           barrier
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 11, tp4 - tp3);   //           fragment tables
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 12, tp5 - tp4);   //           rest of the DMA / first tile
    MG_STAMP_STORE(g_stamps_lt, blockIdx.x, wave, lane, 13, ts1 - tp5);   //           barrier + bias
#endif
}

#ifdef MG_EXPERIMENTS      // lab builds only (make lab / diag): the role-split experiment below, measured slower
// ---------------------------------------------------------------------------------------------------------------------
// The same pass with the work of a tile split between the TWO waves of a SIMD (512 threads, 2 waves per SIMD, 256 registers each).
// Why: in the kernel above a wave owns a tile from the first MFMA to the last store; one wave per SIMD (it needs ~490 registers: a
// tile's H1 rows + the next tile's prefetch, the layer-2 accumulators, dW3's accumulators, the tail's temporaries) leaves every LDS
// round trip, every dependent MFMA and the lone-wave VALU issue rate (half the pipe's) exposed: 96 us at C2 against 48 us for the
// H1 stream alone (profiles/r2_kbench_l2tail.txt: the tail is 13,000 cycles per tile for ~5,000 of issue).  Here
//   * waves 0-3 (PRODUCERS) stream H1, run the 128 layer-2 MFMAs of a tile, bias + sigmoid + bf16, and hand the tile's H2 over
//     in the B-operand register layout (8 x 16 bytes per lane: the consumer's lane l takes what the producer's lane l stored);
//   * waves 4-7 (CONSUMERS) run the tail on it (steps 3-9 above: 24 MFMAs, the transcendentals, dZ2 / pred stores, dW3 / db3 /
//     dW4 / db4 / loss accumulators).  Producer w and consumer w + 4 sit on the same SIMD: the consumer's VALU / LDS work passes
//     under the producer's MFMAs.
//   * the hand-off goes through a 3-slot ring per pair in GLOBAL memory (8 KB per slot; the 24 MB of all pairs stay in L2 / MALL):
//     LDS has no room (W2 fills 128 of its 160 KB).  Plain stores by the producer (written through to this XCD's L2), sc1 loads
//     (L1 bypass) by the consumer on the same CU; the pair's `ready` / `done` counters live in LDS.  The producer publishes tile
//     n - 1 when it is about to store tile n: vector memory retires in order, so s_waitcnt vmcnt(32) - the 32 row loads issued
//     since - says those stores are complete without draining the prefetch.  Spins are bounded; a wait that gives up makes the loss NaN.
// MEASURED SLOWER and kept as an experiment (MG_TUNE_AB = 64; correct: the parity tests pass on it): 107 us against 100 at C2,
// 26.3 against 25.2 at the phone-rate rows.  The split moves the sigmoid of H2 to the producer, but the consumer still owns the tail's
// dependent chain (8 + 2 + 2 dependent MFMAs, three LDS round trips, the loads of the hand-off: ~10,000 cycles per tile of which
// ~4,000 are issue) and has ONE tile in flight; a second tile in flight per consumer does not fit 256 registers.
// ---------------------------------------------------------------------------------------------------------------------
#define LT_FLAGS (LT_WAVE0 + 4 * LT_WAVE_BYTES)          // int ready[4], done[4], error
#define LT_LDS2 (LT_FLAGS + 64)
#define LT_RING 3                                        // hand-off slots per producer / consumer pair
#define LT_SLOT_VEC (8 * 64)                             // u32x4 per slot
#define LT_SPIN_LIMIT (1 << 22)

__global__ __launch_bounds__(512, 2) void f0_l2tail_split_kernel(const uint16_t* __restrict__ H1, int ldh1, const uint16_t* __restrict__ W2,
                                                                 int ldw2, const float* __restrict__ b2, const float* __restrict__ W3,
                                                                 const float* __restrict__ b3, const float* __restrict__ W4,
                                                                 const float* __restrict__ b4, const float* __restrict__ target,
                                                                 const int64_t* __restrict__ seq_len, int64_t M, int B, int T,
                                                                 float grad_scale, float* __restrict__ pred, uint16_t* __restrict__ dZ2,
                                                                 int lddz, float* __restrict__ slab, const float* __restrict__ row_weight,
                                                                 u32x4* __restrict__ xbuf) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LT_LDS2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mi = lane & 31, lh = lane >> 5;
    const int q = wave & 3;
    volatile int* flags = reinterpret_cast<volatile int*>(smem + LT_FLAGS);       // [0..3] ready, [4..7] done, [8] error

    // ---- one-time (all 8 waves): W2, the two permuted W3 fragment tables, b2 (as the kernel above) -----------------------------
    {
        unsigned short* w3s = reinterpret_cast<unsigned short*>(smem + LT_WAVE0);
        float w3v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w3v[i] = W3[tid + 512 * i];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            u32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 512 * (8 * r + i), n = e >> 6, c = e & 63;
                v[i] = *reinterpret_cast<const u32x4*>(W2 + (size_t)n * ldw2 + c * 8);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 512 * (8 * r + i), n = e >> 6, c = e & 63;
                *reinterpret_cast<u32x4*>(smem + LT_W2 + n * (LT_K * 2) + ((c ^ (n & 15)) << 4)) = v[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) w3s[tid + 512 * i] = mg_f2bf(w3v[i]);
        if (tid < LT_N2) *reinterpret_cast<float*>(smem + LT_B2 + tid * 4) = b2[tid];
        if (tid < 16) flags[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + 512 * i, l = f & 63, r = l & 31, h = l >> 5, g = (f >> 6) & 7;
            unsigned short el[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int u = 8 * (j >> 2) + 4 * h + (j & 3);
                el[j] = (f < 512) ? w3s[r * LT_N2 + 16 * g + u] : w3s[(16 * (g & 1) + u) * LT_N2 + 32 * (g >> 1) + r];
            }
            *reinterpret_cast<u32x4*>(smem + LT_ZF + f * 16) = u32x4{el[0] | ((unsigned)el[1] << 16), el[2] | ((unsigned)el[3] << 16),
                                                                   el[4] | ((unsigned)el[5] << 16), el[6] | ((unsigned)el[7] << 16)};
        }
        __syncthreads();
    }

    const int64_t n_tiles = (M + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    const int64_t tile0 = (int64_t)blockIdx.x * 4 + q;
    u32x4* const xq = xbuf + ((size_t)blockIdx.x * 4 + q) * (LT_RING * LT_SLOT_VEC) + lane;
    bool timed_out = false;

    if (wave < 4) {
        // =============================================== producer ===============================================
        const int w2_lane = LT_W2 + mi * (LT_K * 2);
        const int xl = (mi & 15) ^ (4 * lh);
        auto load_half = [&](u32x4 (&dst)[16], int64_t tile, int half) {
            int64_t mm = tile * 32 + mi;
            if (mm > M - 1) mm = M - 1;
            const uint16_t* hp = H1 + (size_t)mm * ldh1 + 256 * half + 32 * lh;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) dst[4 * a + c] = *reinterpret_cast<const u32x4*>(hp + 64 * a + 8 * c);
        };
        f32x16 acc[4];
        auto bias_acc = [&]() {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(smem + LT_B2 + (32 * blk + 8 * qq + 4 * lh) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[blk][4 * qq + e] = bq[e];
                }
        };
        u32x4 ha[16], hb[16];
        if (tile0 < n_tiles) {
            load_half(ha, tile0, 0);
            load_half(hb, tile0, 1);
        }
        int n = 0;                                           // ordinal of the tile inside this pair's sequence
        for (int64_t tile = tile0; tile < n_tiles; tile += stride, ++n) {
            bias_acc();
            auto w2frag = [&](int j, int blk) -> bfv8 {
                const int a = j >> 2, c = j & 3;
                return *reinterpret_cast<const bfv8*>(smem + w2_lane + blk * 32 * (LT_K * 2) + (a >> 1) * 256 + ((xl ^ (8 * (a & 1) + c)) << 4));
            };
            bfv8 wf[2][4];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) wf[0][blk] = w2frag(0, blk);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            const int64_t next = tile + stride;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (j + 1 < 32) {
#pragma unroll
                    for (int blk = 0; blk < 4; ++blk) wf[(j + 1) & 1][blk] = w2frag(j + 1, blk);
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                }
                const u32x4 hv = j < 16 ? ha[j & 15] : hb[j & 15];
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
                    acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j & 1][blk], __builtin_bit_cast(bfv8, hv), acc[blk], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (j == 15) {
                    load_half(ha, next, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 16, 0);
                }
            }
            load_half(hb, next, 1);
            asm volatile("" ::: "memory");                   // the 32 row loads of this phase stay behind the stores of the last tile
            // sigmoid, bf16: hq[2 blk + s] = registers 8 s .. 8 s + 7 of block blk
            u32x4 hq[8];
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                unsigned int w[8];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mg_sigmoid_fast(acc[blk][4 * qq + e]);
                    w[2 * qq] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]});
                    w[2 * qq + 1] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]});
                }
                hq[2 * blk] = u32x4{w[0], w[1], w[2], w[3]};
                hq[2 * blk + 1] = u32x4{w[4], w[5], w[6], w[7]};
            }
            // publish tile n - 1: its stores are older than the 32 row loads issued since
            asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            if (lane == 0) flags[q] = n;
            // slot n % LT_RING is free once the consumer has loaded tile n - LT_RING
            if (n >= LT_RING) {
                int spins = 0;
                while (flags[4 + q] < n - LT_RING + 1) {
                    if (++spins > LT_SPIN_LIMIT) {
                        timed_out = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            u32x4* dst = xq + (n % LT_RING) * LT_SLOT_VEC;
#pragma unroll
            for (int st = 0; st < 8; ++st) dst[st * 64] = hq[st];
            asm volatile("" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) flags[q] = n;                         // every tile published
        if (timed_out && lane == 0) flags[8] = 1;
    } else {
        // =============================================== consumer ===============================================
        unsigned char* patch = smem + LT_WAVE0 + q * LT_WAVE_BYTES;
        const float b4v = b4[0];
        f32x16 acc_w3[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_w3[kt][r] = 0.f;
        float dw4p[16], db3p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dw4p[r] = db3p[r] = 0.f;
        float db4p = 0.f, lossp = 0.f;
        float* const pred_sink = reinterpret_cast<float*>(g_lt_sink) + lane;
        uint16_t* const dz_sink = reinterpret_cast<uint16_t*>(g_lt_sink) + lane * 8;

        int n = 0;
        for (int64_t tile = tile0; tile < n_tiles; tile += stride, ++n) {
            const int64_t m = tile * 32 + mi;
            const bool live = m < M;
            // per-frame scalars of the loss (requested before the wait for the tile)
            float tg, s1, s2;
            {
                int64_t mm = m;
                if (mm > M - 1) mm = M - 1;
                tg = target[mm];
                if (row_weight) {
                    s1 = row_weight[mm];
                    s2 = 0.f;
                } else {
                    const unsigned mu = (unsigned)mm, b = mu / (unsigned)T, t = mu - b * (unsigned)T;
                    int64_t nb = seq_len ? seq_len[b] : (int64_t)T;
                    if (nb > T) nb = T;
                    if (nb < 0) nb = 0;
                    s1 = (int64_t)t < nb ? 1.f : 0.f;
                    s2 = (float)nb;
                }
            }
            {
                int spins = 0;
                while (flags[q] < n + 1) {
                    if (++spins > LT_SPIN_LIMIT) {
                        timed_out = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            u32x4 hq[8];
            {
                const u32x4* src = xq + (n % LT_RING) * LT_SLOT_VEC;
#pragma unroll
                for (int st = 0; st < 8; ++st) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(hq[st]) : "v"(src + st * 64) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int st = 0; st < 8; ++st) asm volatile("" : "+v"(hq[st]));
                if (lane == 0) flags[4 + q] = n + 1;         // the slot may be refilled
            }

            // (3) Z3^T = W3 . H2^T
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const bfv8 a = *reinterpret_cast<const bfv8*>(smem + LT_ZF + (st * 64 + lane) * 16);
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bfv8, hq[st]), z, 0, 0, 0);
            }
            // (4) sigmoid, prediction (b3 / w4 per register <-> unit 8 (r >> 2) + 4 lh + (r & 3), read as 16-byte groups)
            float h3[16], w4r[16];
            float ph = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b3 + 8 * g + 4 * lh);
                const f32x4 ww = *reinterpret_cast<const f32x4*>(W4 + 8 * g + 4 * lh);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    w4r[4 * g + e] = ww[e];
                    h3[4 * g + e] = mg_sigmoid_fast(z[4 * g + e] + bb[e]);
                    ph += h3[4 * g + e] * ww[e];
                }
            }
            const auto phs = __builtin_amdgcn_permlane32_swap(__float_as_uint(ph), __float_as_uint(ph), false, false);
            const float p = (__uint_as_float(phs[0]) + __uint_as_float(phs[1])) + b4v;
            // (5) masked MSE of this frame (as above)
            const float inv = 1.f / (s2 * (float)B);
            const float cw = row_weight ? 2.f * grad_scale : 2.f * grad_scale * inv;
            const float lw = row_weight ? 1.f : inv;
            const float e = p - tg;
            const float dpred = live ? (e * s1) * cw : 0.f;
            if (lh == 0) {
                lossp += live ? (e * e * s1) * lw : 0.f;
                db4p += dpred;
            }
            *((live && lh == 0) ? pred + m : pred_sink) = p;
            // (6) backward of layer 4 and of the layer-3 sigmoid
            float dz3[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dz3[r] = dpred * w4r[r] * h3[r] * (1.f - h3[r]);
                dw4p[r] += dpred * h3[r];
                db3p[r] += dz3[r];
            }
            // (7) dZ3 as bf16
            u32x4 dzf[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                unsigned int w[4];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
                    w[qq] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)dz3[8 * s + 2 * qq], (__bf16)dz3[8 * s + 2 * qq + 1]});
                dzf[s] = u32x4{w[0], w[1], w[2], w[3]};
                *reinterpret_cast<u32x2*>(patch + mi * 64 + (16 * s + 4 * lh) * 2) = u32x2{w[0], w[1]};
                *reinterpret_cast<u32x2*>(patch + mi * 64 + (16 * s + 8 + 4 * lh) * 2) = u32x2{w[2], w[3]};
            }
            bfv8 dzt[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) dzt[s] = lt_tr_frag(patch, lane, s, false);
            // (8), (9)
            uint16_t* const dzrow = live ? dZ2 + (size_t)m * lddz + 8 * lh : dz_sink;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bfv8 a = *reinterpret_cast<const bfv8*>(smem + LT_W3P + ((kt * 2 + s) * 64 + lane) * 16);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bfv8, dzf[s]), d, 0, 0, 0);
                }
                unsigned int pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const unsigned int w0 = hq[2 * kt + (g >> 1)][2 * (g & 1)], w1 = hq[2 * kt + (g >> 1)][2 * (g & 1) + 1];
                    const float h[4] = {__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16),
                                        __uint_as_float(w1 & 0xffff0000u)};
                    float v[4];
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) v[e2] = d[4 * g + e2] * h[e2] * (1.f - h[e2]);
                    pk[g][0] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]});
                    pk[g][1] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]});
                    *reinterpret_cast<u32x2*>(patch + mi * 64 + ((g ^ ((mi >> 2) & 3)) << 4) + 8 * lh) = u32x2{w0, w1};
                }
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                    const auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                    *reinterpret_cast<u32x4*>(dzrow + 32 * kt + 8 * g) = u32x4{r0[0], r1[0], r0[1], r1[1]};
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bfv8 bq = lt_tr_frag(patch, lane, s, true);
                    acc_w3[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dzt[s], bq, acc_w3[kt], 0, 0, 0);
                }
            }
        }
        if (timed_out && lane == 0) flags[8] = 1;

        // ---- reduction: lanes (frames) -> wave, fixed order ------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) {
                dw4p[r] += __shfl_xor(dw4p[r], off, 64);
                db3p[r] += __shfl_xor(db3p[r], off, 64);
            }
        }
        db4p = mg_wave_sum(db4p);
        lossp = mg_wave_sum(lossp);
        __builtin_amdgcn_s_barrier();                      // (A) every producer is past its last W2 read: see below
        float* mine = reinterpret_cast<float*>(smem + LT_W2) + q * LT_SLAB;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int k = 32 * kt + mi;
                mine[nn * LT_N2 + k] = acc_w3[kt][r];
            }
        if (mi == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int nn = 8 * (r >> 2) + 4 * lh + (r & 3);
                mine[LT_N3 * LT_N2 + nn] = db3p[r];
                mine[LT_N3 * LT_N2 + LT_N3 + nn] = dw4p[r];
            }
        }
        if (lane == 0) {
            mine[LT_N3 * LT_N2 + 2 * LT_N3] = db4p;
            mine[LT_N3 * LT_N2 + 2 * LT_N3 + 1] = lossp;
        }
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();            // (A) pairs with the consumers' barrier: the W2 region becomes the sums
    __syncthreads();
    const float* red = reinterpret_cast<const float*>(smem + LT_W2);
    float* out = slab + (size_t)blockIdx.x * LT_SLAB_STRIDE;
    const bool failed = flags[8] != 0;                 // a wait gave up: the results are invalid and the loss says so (NaN)
    for (int e = tid; e < LT_SLAB; e += 512) {
        const float v = ((red[e] + red[LT_SLAB + e]) + red[2 * LT_SLAB + e]) + red[3 * LT_SLAB + e];
        out[e] = (failed && e == LT_SLAB - 1) ? __builtin_nanf("") : v;
    }
}
#endif  // MG_EXPERIMENTS


#ifdef MG_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------------
// EXPERIMENT (lab builds only; round 4; MG_TUNE_AB = 67): the same pass on the 16x16x32 MFMA shape.  MEASURED EQUAL TO SLOWER - in the
// C2 frame-rate step 0.5153-0.5186 ms against 0.5126-0.5137 for the kernel above (same box, three alternating runs), 98.4 against
// 89.8 us in scripts/kbench_l2tail.py - and kept as evidence: the hypothesis it tests is false.  The hypothesis: f0_l2tail_kernel is a stream (48 us of H1 at C2, 53 with the layer-2
// MFMAs) plus a DEPENDENT CHAIN per 32-frame tile that nothing covers on a wave without a partner (13,000 cycles for ~5,000 of
// issue: eight dependent 32x32x16 MFMAs for Z3, per 32-unit slice two more dependent ones in front of the dZ2 arithmetic, three LDS
// round trips).  A second wave per SIMD does not fit (256 registers: DESIGN.md R4.1).  On 16 x 16 blocks the same tile is TWO
// independent 16-frame sub-tiles per wave iteration - a lane owns frame f of each and 4 consecutive units of every 16-unit block -
// so every step of the chain exists twice, independently (the scheduler interleaves them), and the chains themselves get
// shorter: Z3 is 2 unit blocks x 4 dependent k-steps (it was 8 dependent MFMAs), dH2^T = W3^T dZ3^T is ONE 32-deep MFMA per
// 16-unit block, all eight independent (K = the 32 units of Z3 is exactly one k-step), and dW3 += dZ3^T H2 contracts over the 32
// frames of both sub-tiles in one k-step per block.  Same LDS map, same W2 image and swizzle (conflict free for the 16-row A
// operand: chunk (4 s + g) ^ u over u = 0..15, g = 0..3 hits 16 different 16-byte bank groups per ds_read_b128 lane group), same
// W2 bytes per FLOP (a fragment feeds both sub-tiles), same arithmetic up to the summation order inside the dot products.
//   lane l = (f, g) = (l & 15, l >> 4); C^T block of 16 units x 16 frames: register r <-> unit 16 ub + 4 g + r, frame f.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bfv8 l16_tr_frag(const unsigned char* tile, int lane, int cb, bool swz) {
    // 16x16x32 operand whose contraction index is the LDS row (32 rows x 64 B): lane group G = lane >> 4 takes contraction rows
    // 8 G .. 8 G + 7 (two 4 x 16 blocks, ds_read_b64_tr_b16 each), lane i = lane & 15 receives column 16 cb + i; lane 4 q + p of a
    // group supplies the address of row q, columns 4 p .. 4 p + 3 of the block.
    const int i = lane & 15, G = lane >> 4;
    const int q = i >> 2, p = i & 3;
    const int row0 = 8 * G + q;
    const int col = 16 * cb + 4 * p;
    const int c = col >> 3, in = (col & 7) << 1;
    const int c_lo = swz ? (c ^ ((row0 >> 2) & 3)) : c;
    const int c_hi = swz ? (c ^ (((row0 + 4) >> 2) & 3)) : c;
    const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(tile + row0 * 64 + ((c_lo << 4) | in)));
    const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(tile + (row0 + 4) * 64 + ((c_hi << 4) | in)));
    return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256, 1) void f0_l2tail16_kernel(const uint16_t* __restrict__ H1, int ldh1, const uint16_t* __restrict__ W2,
                                                             int ldw2, const float* __restrict__ b2, const float* __restrict__ W3,
                                                             const float* __restrict__ b3, const float* __restrict__ W4,
                                                             const float* __restrict__ b4, const float* __restrict__ target,
                                                             const int64_t* __restrict__ seq_len, int64_t M, int B, int T,
                                                             float grad_scale, float* __restrict__ pred, uint16_t* __restrict__ dZ2,
                                                             int lddz, float* __restrict__ slab, const float* __restrict__ row_weight, int rev) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LT_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = lane & 15, g = lane >> 4;

    // ---- the two W3 fragment tables, gathered straight from W3 (fp32 [32][128]): two fragments of each per thread -----------------
    f32x4 w3a[2][2];
    float w3b[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int fr = tid + 256 * i, l = fr & 63, u = l & 15, gg = l >> 4, q = (fr >> 6) & 7;
        // Z3 fragment [ob = q >> 2][s3 = q & 3][lane l]: element j = W3[16 ob + u][32 s3 + (j < 4 ? 4 gg + j : 16 + 4 gg + j - 4)]
        const float* zr = W3 + (16 * (q >> 2) + u) * LT_N2 + 32 * (q & 3) + 4 * gg;
        w3a[i][0] = *reinterpret_cast<const f32x4*>(zr);
        w3a[i][1] = *reinterpret_cast<const f32x4*>(zr + 16);
        // dH2 fragment [ub = q][lane l]: element j = W3[j < 4 ? 4 gg + j : 16 + 4 gg + j - 4][16 ub + u]
#pragma unroll
        for (int j = 0; j < 8; ++j) w3b[i][j] = W3[((j < 4 ? 0 : 12) + 4 * gg + j) * LT_N2 + 16 * q + u];
    }
    const float b2v = tid < LT_N2 ? b2[tid] : 0.f;
    unsigned char* patch = smem + LT_WAVE0 + wave * LT_WAVE_BYTES;

    float b3v[2][4], w4v[2][4];                            // [ob][r] <-> unit 16 ob + 4 g + r of the 32-wide layer
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b3v[ob][r] = b3[16 * ob + 4 * g + r];
            w4v[ob][r] = W4[16 * ob + 4 * g + r];
        }
    const float b4v = b4[0];

    f32x4 acc_w3[2][8];                                    // dW3 block (nb, ubk): row 16 nb + 4 g + r, column 16 ubk + f
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc_w3[nb][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dw4p[2][4], db3p[2][4];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) dw4p[ob][r] = db3p[ob][r] = 0.f;
    float db4p = 0.f, lossp = 0.f;

    const int64_t n_tiles = (M + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 4;
    auto first_row = [&](int64_t t) -> int64_t { return (rev ? (t < n_tiles ? n_tiles - 1 - t : 0) : t) * 32; };
    // H1 operand registers: k-steps 0-7 of both sub-tiles in h0, 8-15 in h1 (entry 2 s' + t); a lane's 16 bytes of k-step s are
    // bytes [64 s + 16 g, + 16) of its frame's row - the four lanes of a frame read 64 contiguous bytes per instruction
    auto load_half = [&](u32x4 (&dst)[16], int64_t tile_, int half) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            int64_t mm = first_row(tile_) + 16 * t + f;
            if (mm > M - 1) mm = M - 1;                     // rows past the end: any valid row (their results are discarded)
            const uint16_t* hp = H1 + (size_t)mm * ldh1 + 256 * half + 8 * g;
#pragma unroll
            for (int s = 0; s < 8; ++s) dst[2 * s + t] = *reinterpret_cast<const u32x4*>(hp + 32 * s);
        }
    };
    auto load_scalars = [&](int64_t tile_, int t, float& tg, float& s1, float& s2) {
        int64_t mm = first_row(tile_) + 16 * t + f;
        if (mm > M - 1) mm = M - 1;
        tg = target[mm];
        if (row_weight) {
            s1 = row_weight[mm];
            s2 = 0.f;
        } else {
            const unsigned mu = (unsigned)mm, b = mu / (unsigned)T, tt = mu - b * (unsigned)T;     // M < 2^31 (checked by the launcher)
            int64_t nb = seq_len ? seq_len[b] : (int64_t)T;
            if (nb > T) nb = T;
            if (nb < 0) nb = 0;
            s1 = (int64_t)tt < nb ? 1.f : 0.f;
            s2 = (float)nb;
        }
    };

    f32x4 acc[2][8];
    auto bias_acc = [&]() {
#pragma unroll
        for (int ub = 0; ub < 8; ++ub) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(smem + LT_B2 + (16 * ub + 4 * g) * 4);
            acc[0][ub] = bq;
            acc[1][ub] = bq;
        }
    };
    float* const pred_sink = reinterpret_cast<float*>(g_lt_sink) + lane;
    uint16_t* const dz_sink = reinterpret_cast<uint16_t*>(g_lt_sink) + lane * 8;

    // dZ2 / pred of a tile are stored at the top of the NEXT iteration, in front of that iteration's prefetch (as the kernel above)
    u32x4 dzst[2][4];
    uint16_t* dzp[2] = {dz_sink, dz_sink};
    float* predp[2] = {pred_sink, pred_sink};
    float pst[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) dzst[t][j] = u32x4{0u, 0u, 0u, 0u};

    u32x4 h0[16], h1[16];
    float tg_n[2] = {0.f, 0.f}, s1_n[2] = {0.f, 0.f}, s2_n[2] = {0.f, 0.f};
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile < n_tiles) {
        load_scalars(tile, 0, tg_n[0], s1_n[0], s2_n[0]);
        load_scalars(tile, 1, tg_n[1], s1_n[1], s2_n[1]);
        load_half(h0, tile, 0);
        load_half(h1, tile, 1);
    }
    // ---- one-time: W2 by LDS-DMA (position p of row n holds chunk p ^ (n & 15)), b2, the fragment tables -------------------------
    {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int n = wave * 32 + i;
            mg_glds16(W2 + (size_t)n * ldw2 + ((lane ^ (n & 15)) << 3), smem + LT_W2 + n * (LT_K * 2));
        }
        if (tid < LT_N2) *reinterpret_cast<float*>(smem + LT_B2 + tid * 4) = b2v;
        typedef __bf16 bfv2_ __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int fr = tid + 256 * i;
            auto pk = [](float a, float b) { return __builtin_bit_cast(unsigned int, bfv2_{(__bf16)a, (__bf16)b}); };
            *reinterpret_cast<u32x4*>(smem + LT_ZF + fr * 16) =
                u32x4{pk(w3a[i][0][0], w3a[i][0][1]), pk(w3a[i][0][2], w3a[i][0][3]), pk(w3a[i][1][0], w3a[i][1][1]), pk(w3a[i][1][2], w3a[i][1][3])};
            *reinterpret_cast<u32x4*>(smem + LT_ZF + (512 + fr) * 16) =
                u32x4{pk(w3b[i][0], w3b[i][1]), pk(w3b[i][2], w3b[i][3]), pk(w3b[i][4], w3b[i][5]), pk(w3b[i][6], w3b[i][7])};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's rows of W2 have landed (the compiler does not count the DMAs)
        __syncthreads();
    }
    bias_acc();

    // W2 fragment of k-step s, unit block ub: row 16 ub + f, chunk 4 s + g at position (4 s + g) ^ f
    const int w2_lane = LT_W2 + f * (LT_K * 2);
    const int xlow = (g ^ (f & 3)) << 4, fh = f >> 2;
    auto w2frag = [&](int s, int ub) -> bfv8 {
        return *reinterpret_cast<const bfv8*>(smem + w2_lane + ub * 16 * (LT_K * 2) + ((s ^ fh) << 6) + xlow);
    };

    for (; tile < n_tiles; tile += stride) {
        int64_t m[2];
        bool live[2];
        float tg[2], s1[2], s2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            m[t] = first_row(tile) + 16 * t + f;
            live[t] = m[t] < M;
            tg[t] = tg_n[t], s1[t] = s1_n[t], s2[t] = s2_n[t];
            load_scalars(tile + stride, t, tg_n[t], s1_n[t], s2_n[t]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            *predp[t] = pst[t];
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(dzp[t] + 32 * j) = dzst[t][j];
        }
        __builtin_amdgcn_sched_barrier(0);

        // (1) H2^T = W2 . H1^T + b2 for both sub-tiles: a W2 fragment feeds two MFMAs; fragments read one k-step ahead
        const int64_t next = tile + stride;
        bfv8 wf[2][8];
#pragma unroll
        for (int ub = 0; ub < 8; ++ub) wf[0][ub] = w2frag(0, ub);
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (s + 1 < 16) {
#pragma unroll
                for (int ub = 0; ub < 8; ++ub) wf[(s + 1) & 1][ub] = w2frag(s + 1, ub);
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
            }
            const u32x4 hv0 = s < 8 ? h0[2 * (s & 7)] : h1[2 * (s & 7)];
            const u32x4 hv1 = s < 8 ? h0[2 * (s & 7) + 1] : h1[2 * (s & 7) + 1];
#pragma unroll
            for (int ub = 0; ub < 8; ++ub) {
                acc[0][ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s & 1][ub], __builtin_bit_cast(bfv8, hv0), acc[0][ub], 0, 0, 0);
                acc[1][ub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s & 1][ub], __builtin_bit_cast(bfv8, hv1), acc[1][ub], 0, 0, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            if (s == 7) {                                    // the registers just consumed take the next tile's first k-half
                load_half(h0, next, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 16, 0);
            }
        }
        load_half(h1, next, 1);

        // (2) sigmoid, bf16: hq[t][ub] = the lane's 4 units of block ub; B fragment of Z3's k-step s3 = blocks 2 s3, 2 s3 + 1
        bfv8 zf[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) zf[q] = *reinterpret_cast<const bfv8*>(smem + LT_ZF + (q * 64 + lane) * 16);
        u32x2 hq[2][8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ub = 0; ub < 8; ++ub) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = mg_sigmoid_fast(acc[t][ub][e]);
                hq[t][ub] = u32x2{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                  __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
            }

        // (3) Z3^T = W3 . H2^T: per sub-tile two unit blocks x four k-steps
        f32x4 z[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) z[t][ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s3 = 0; s3 < 4; ++s3)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const u32x4 bv = u32x4{hq[t][2 * s3][0], hq[t][2 * s3][1], hq[t][2 * s3 + 1][0], hq[t][2 * s3 + 1][1]};
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
                    z[t][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[ob * 4 + s3], __builtin_bit_cast(bfv8, bv), z[t][ob], 0, 0, 0);
            }
        bfv8 w3p[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) w3p[q] = *reinterpret_cast<const bfv8*>(smem + LT_W3P + (q * 64 + lane) * 16);

        // (4)-(7) per sub-tile: sigmoid, prediction (sum over the four lanes of a frame: xor 16, xor 32), loss term, dZ3
        float dz3[2][2][4];
        u32x4 dzf[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float h3[2][4];
            float ph = 0.f;
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h3[ob][r] = mg_sigmoid_fast(z[t][ob][r] + b3v[ob][r]);
                    ph += h3[ob][r] * w4v[ob][r];
                }
            ph += __shfl_xor(ph, 16, 64);
            ph += __shfl_xor(ph, 32, 64);
            const float p = ph + b4v;
            const float inv = 1.f / (s2[t] * (float)B);
            const float cw = row_weight ? 2.f * grad_scale : 2.f * grad_scale * inv;
            const float lw = row_weight ? 1.f : inv;
            const float e = p - tg[t];
            const float dpred = live[t] ? (e * s1[t]) * cw : 0.f;
            if (g == 0) {
                lossp += live[t] ? (e * e * s1[t]) * lw : 0.f;
                db4p += dpred;
            }
            predp[t] = (live[t] && g == 0) ? pred + m[t] : pred_sink;
            pst[t] = p;
            unsigned int w[4];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dz3[t][ob][r] = dpred * w4v[ob][r] * h3[ob][r] * (1.f - h3[ob][r]);
                    dw4p[ob][r] += dpred * h3[ob][r];
                    db3p[ob][r] += dz3[t][ob][r];
                }
                w[2 * ob] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)dz3[t][ob][0], (__bf16)dz3[t][ob][1]});
                w[2 * ob + 1] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)dz3[t][ob][2], (__bf16)dz3[t][ob][3]});
            }
            dzf[t] = u32x4{w[0], w[1], w[2], w[3]};       // B fragment of dH2's one k-step: units 4 g + j | 16 + 4 g + j
            // row-major copy for the transposed reads of dW3: row 16 t + f, units 4 g .. (block 0) at byte 8 g, block 1 at 32 + 8 g
            *reinterpret_cast<u32x2*>(patch + (16 * t + f) * 64 + 8 * g) = u32x2{w[0], w[1]};
            *reinterpret_cast<u32x2*>(patch + (16 * t + f) * 64 + 32 + 8 * g) = u32x2{w[2], w[3]};
        }
        bfv8 dzt[2];                                        // dZ3^T fragments of the dW3 product (contraction over the 32 frames)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) dzt[nb] = l16_tr_frag(patch, lane, nb, false);

        // (8) dH2^T = W3^T . dZ3^T (one MFMA per 16-unit block), dZ2 = dH2 * H2 (1 - H2); pairs of blocks trade halves between the
        //     lanes g, g ^ 1 of a frame (v_permlane16_swap) so that every lane stores 16 bytes: g even -> block 2 j, units 8 (g / 2) ..,
        //     g odd -> block 2 j + 1;   (9) per 32-unit slice: dW3[:, slice] += dZ3^T . H2[:, slice] through the patch
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            unsigned int pk[2][2][2];                        // [t][block of the pair][word]
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int ub = 2 * kt + c;
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3p[ub], __builtin_bit_cast(bfv8, dzf[t]), zero, 0, 0, 0);
                    const unsigned int w0 = hq[t][ub][0], w1 = hq[t][ub][1];
                    const float h[4] = {__uint_as_float(w0 << 16), __uint_as_float(w0 & 0xffff0000u), __uint_as_float(w1 << 16),
                                        __uint_as_float(w1 & 0xffff0000u)};
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = d[e] * h[e] * (1.f - h[e]);
                    pk[t][c][0] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]});
                    pk[t][c][1] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]});
                    // H2 slice row 16 t + f: block c of the slice = columns 16 c + 4 g .. -> chunk 2 c + (g >> 1), half g & 1
                    const int row = 16 * t + f;
                    *reinterpret_cast<u32x2*>(patch + row * 64 + (((2 * c + (g >> 1)) ^ ((row >> 2) & 3)) << 4) + 8 * (g & 1)) = u32x2{w0, w1};
                }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const auto r0 = __builtin_amdgcn_permlane16_swap(pk[t][0][0], pk[t][1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(pk[t][0][1], pk[t][1][1], false, false);
                dzst[t][kt] = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bfv8 bq = l16_tr_frag(patch, lane, c, true);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc_w3[nb][2 * kt + c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dzt[nb], bq, acc_w3[nb][2 * kt + c], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)                          // 16 bytes per store: block 2 j + (g & 1), units 8 (g >> 1) .. + 7; j -> + 32 columns
            dzp[t] = live[t] ? dZ2 + (size_t)m[t] * lddz + 16 * (g & 1) + 8 * (g >> 1) : dz_sink;
        bias_acc();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        *predp[t] = pst[t];                                  // the last tile's results
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(dzp[t] + 32 * j) = dzst[t][j];
    }

    // ---- reduction: the 16 frames of a lane row on the DPP path, then wave -> workgroup in a fixed order --------------------------
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dw4p[ob][r] = mg_row16_sum(dw4p[ob][r]);
            db3p[ob][r] = mg_row16_sum(db3p[ob][r]);
        }
    db4p = mg_wave_sum(db4p);
    lossp = mg_wave_sum(lossp);

    __syncthreads();                                   // every wave is done with W2: its region takes the four waves' sums
    float* red = reinterpret_cast<float*>(smem + LT_W2);               // [4 waves][LT_SLAB]
    float* mine = red + wave * LT_SLAB;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(16 * nb + 4 * g + r) * LT_N2 + 16 * k + f] = acc_w3[nb][k][r];
    if (f == 0) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mine[LT_N3 * LT_N2 + 16 * ob + 4 * g + r] = db3p[ob][r];
                mine[LT_N3 * LT_N2 + LT_N3 + 16 * ob + 4 * g + r] = dw4p[ob][r];
            }
    }
    if (lane == 0) {
        mine[LT_N3 * LT_N2 + 2 * LT_N3] = db4p;
        mine[LT_N3 * LT_N2 + 2 * LT_N3 + 1] = lossp;
    }
    __syncthreads();
    float* out = slab + (size_t)blockIdx.x * LT_SLAB_STRIDE;
    for (int e = tid; e < LT_SLAB; e += 256) out[e] = ((red[e] + red[LT_SLAB + e]) + red[2 * LT_SLAB + e]) + red[3 * LT_SLAB + e];
}

#endif  // MG_EXPERIMENTS (f0_l2tail16_kernel)

#ifdef MG_EXPERIMENTS
// l2tail_wide.hip (lab builds, MG_TUNE_AB = 69 and its probes 101-108): the pass with W2 in registers and H1 through an LDS ring
int mg_launch_f0_l2tail_wide(const uint16_t* H1, int ldh1, const uint16_t* W2, int ldw2, const float* b2, const float* W3, const float* b3,
                             const float* W4, const float* b4, const float* target, const int64_t* seq_len, int64_t M, int B, int T,
                             float grad_scale, float* pred, uint16_t* dZ2, int lddz, float* slab, const float* row_weight, int rev, int grid,
                             hipStream_t st);
#endif

static int l2tail_blocks(int64_t M) {
    int64_t blocks = mg_ceil_div(mg_ceil_div(M, 32), 4);
    if (blocks > 256) blocks = 256;                    // one resident workgroup per CU (156 KB of LDS each)
    return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" {

static size_t l2tail_slab_bytes(int64_t M) { return mg_align_up((size_t)l2tail_blocks(M) * LT_SLAB_STRIDE * sizeof(float), 256); }
// slabs of the workgroups' sums (lab builds: + the hand-off ring of the role-split experiment, 3 slots x 8 KB per wave pair)
size_t mg_f0_l2tail_workspace_bytes(int64_t M) {
#ifdef MG_EXPERIMENTS
    return l2tail_slab_bytes(M) + (size_t)l2tail_blocks(M) * 4 * LT_RING * LT_SLOT_VEC * sizeof(u32x4);
#else
    return l2tail_slab_bytes(M);
#endif
}

static int f0_l2tail_launch(const char* name, const uint16_t* H1, int ldh1, int K2, const uint16_t* W2, int ldw2, int N2, const float* b2,
                            const float* W3, const float* b3, const float* W4, const float* b4, const float* target,
                            const int64_t* seq_len, const float* row_weight, int64_t M, int B, int T, float grad_scale, float* pred,
                            float* loss, uint16_t* dZ2, int lddz, float* grads, int accumulate, void* workspace, size_t workspace_bytes,
                            void* stream, int* slabs_out = nullptr) {
    MG_CHECK_ARG(H1 && W2 && b2 && W3 && b3 && W4 && b4 && target && pred && (slabs_out || (loss && grads)) && dZ2 && M > 0 && M < 2147483647LL,
                 "%s: bad arguments (M=%lld)", name, (long long)M);
    MG_CHECK_ARG(K2 == LT_K && N2 == LT_N2 && ldh1 >= LT_K && ldh1 % 8 == 0 && ldw2 >= LT_K && ldw2 % 8 == 0 && lddz >= LT_N2 && lddz % 8 == 0,
                 "%s: needs a 512 -> 128 layer (K2=%d N2=%d ldh1=%d ldw2=%d lddz=%d)", name, K2, N2, ldh1, ldw2, lddz);
    MG_CHECK_ARG((((uintptr_t)H1 | (uintptr_t)W2 | (uintptr_t)dZ2) % 16) == 0, "%s: H1 / W2 / dZ2 must be 16-byte aligned", name);
    if (!workspace || workspace_bytes < mg_f0_l2tail_workspace_bytes(M)) {
        mg_set_error("%s: workspace of %zu bytes needed, got %zu", name, mg_f0_l2tail_workspace_bytes(M), workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int blocks = l2tail_blocks(M);
    float* slab = (float*)workspace;
    // Large H1 (more than the 8 L2s hold): walk the tiles from the END - the rows the layer-1 forward wrote last are the ones L2 and
    // the memory-side cache still hold (C2 at frame rate: the step 0.5043 -> 0.5008 ms in a same-box A/B; at phone-rate row counts
    // everything is resident either way and the forward walk is 1 us faster).  MG_TUNE_AB = 96: the forward walk at every size (A/B).
    const int rev = (M >= 65536 && g_mg_tuning[MG_TUNE_AB] != 96) ? 1 : 0;
#define LT_LAUNCH(P_) hipLaunchKernelGGL(f0_l2tail_kernel<P_>, dim3(blocks), dim3(256), 0, st, H1, ldh1, W2, ldw2, b2, W3, b3, W4, b4, target, seq_len, M, B, T, grad_scale, pred, dZ2, lddz, slab, row_weight, rev)
#ifdef MG_EXPERIMENTS
    // lab builds only: MG_TUNE_AB 64 = the producer / consumer role split (same results, measured slower); 1 .. 32 = the product
    // kernel with parts switched off (timing probes, results garbage)
    static const int wide_env = getenv("MG_L2TAIL_WIDE") ? atoi(getenv("MG_L2TAIL_WIDE")) : 0;      // lab: the wide form with every other switch at its default
    if (g_mg_tuning[MG_TUNE_AB] == 69 || g_mg_tuning[MG_TUNE_AB] > 100 || (wide_env && g_mg_tuning[MG_TUNE_AB] == 0)) {          // the wide form and its timing probes
        // (up to 256 slabs: the lab build's workspace holds them - the role split's ring lies behind the slab area)
        blocks = mg_launch_f0_l2tail_wide(H1, ldh1, W2, ldw2, b2, W3, b3, W4, b4, target, seq_len, M, B, T, grad_scale, pred, dZ2, lddz, slab,
                                          row_weight, rev, 256, st);
    } else if (g_mg_tuning[MG_TUNE_AB] == 64) {
        u32x4* xbuf = reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(workspace) + l2tail_slab_bytes(M));
        hipLaunchKernelGGL(f0_l2tail_split_kernel, dim3(blocks), dim3(512), 0, st, H1, ldh1, W2, ldw2, b2, W3, b3, W4, b4, target, seq_len, M, B, T,
                           grad_scale, pred, dZ2, lddz, slab, row_weight, xbuf);
    } else
    switch (g_mg_tuning[MG_TUNE_AB]) {
        case 1: LT_LAUNCH(1); break;
        case 2: LT_LAUNCH(2); break;
        case 3: LT_LAUNCH(3); break;
        case 4: LT_LAUNCH(4); break;
        case 6: LT_LAUNCH(6); break;
        case 7: LT_LAUNCH(7); break;
        case 8: LT_LAUNCH(8); break;
        case 16: LT_LAUNCH(16); break;
        case 32: LT_LAUNCH(32); break;
        case 17: LT_LAUNCH(17); break;
        case 68: LT_LAUNCH(256); break;
        case 67:
            hipLaunchKernelGGL(f0_l2tail16_kernel, dim3(blocks), dim3(256), 0, st, H1, ldh1, W2, ldw2, b2, W3, b3, W4, b4, target, seq_len, M, B, T,
                               grad_scale, pred, dZ2, lddz, slab, row_weight, rev);
            break;
        default: LT_LAUNCH(0); break;
    }
#else
    LT_LAUNCH(0);
#endif
#undef LT_LAUNCH
    MG_CHECK_LAUNCH(name);
    if (slabs_out) {                                   // the caller sums the slabs (mg_expand_column_reduce_f32)
        *slabs_out = blocks;
        return MG_OK;
    }
    // grads = [dW3 (32*128) | db3 (32) | dW4 (32) | db4 (1)]; the loss is the last slab entry
    const int n_grads = LT_SLAB - 1;
    if (loss == grads + n_grads && !accumulate) {
        mg_launch_slab_reduce(slab, LT_SLAB, LT_SLAB_STRIDE, blocks, grads, 0, st);   // loss stored right behind the gradients
    } else {
        mg_launch_slab_reduce(slab, n_grads, LT_SLAB_STRIDE, blocks, grads, accumulate, st);
        mg_launch_slab_reduce(slab + (LT_SLAB - 1), 1, LT_SLAB_STRIDE, blocks, loss, 0, st);
    }
    MG_CHECK_LAUNCH(name);
    return MG_OK;
}

int mg_f0_l2tail_bf16(const uint16_t* H1, int ldh1, int K2, const uint16_t* W2, int ldw2, int N2, const float* b2, const float* W3,
                      const float* b3, const float* W4, const float* b4, const float* target, const int64_t* seq_len, int B, int T,
                      float grad_scale, float* pred, float* loss, uint16_t* dZ2, int lddz, float* grads, int accumulate, void* workspace,
                      size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(B > 0 && T > 0, "mg_f0_l2tail_bf16: bad arguments (B=%d T=%d)", B, T);
    return f0_l2tail_launch("mg_f0_l2tail_bf16", H1, ldh1, K2, W2, ldw2, N2, b2, W3, b3, W4, b4, target, seq_len, nullptr, (int64_t)B * T, B, T,
                            grad_scale, pred, loss, dZ2, lddz, grads, accumulate, workspace, workspace_bytes, stream);
}

// mg_f0_l2tail_bf16 without its reduce launch: the workgroups' sums stay in `workspace` as *n_slabs slabs (mg_f0_l2tail_slab_stride()
// floats apart; 32*128 + 32 + 32 + 2 floats used: dW3 | db3 | dW4 | db4 | loss) for the update kernel's plan to sum (gradients: a slab
// source; loss: mg_adam_tail) - a step captured whole into a HIP graph, where nothing reads the loss before the update has run.
int mg_f0_l2tail_slabs_bf16(const uint16_t* H1, int ldh1, int K2, const uint16_t* W2, int ldw2, int N2, const float* b2, const float* W3,
                            const float* b3, const float* W4, const float* b4, const float* target, const int64_t* seq_len, int B, int T,
                            float grad_scale, float* pred, uint16_t* dZ2, int lddz, void* workspace, size_t workspace_bytes, int* n_slabs,
                            void* stream) {
    MG_CHECK_ARG(B > 0 && T > 0 && n_slabs, "mg_f0_l2tail_slabs_bf16: bad arguments (B=%d T=%d)", B, T);
    return f0_l2tail_launch("mg_f0_l2tail_slabs_bf16", H1, ldh1, K2, W2, ldw2, N2, b2, W3, b3, W4, b4, target, seq_len, nullptr, (int64_t)B * T, B, T,
                            grad_scale, pred, nullptr, dZ2, lddz, nullptr, 0, workspace, workspace_bytes, stream, n_slabs);
}

int mg_f0_l2tail_rows_bf16(const uint16_t* H1, int ldh1, int K2, const uint16_t* W2, int ldw2, int N2, const float* b2, const float* W3,
                           const float* b3, const float* W4, const float* b4, const float* target, const float* row_weight, int64_t M,
                           float grad_scale, float* pred, float* loss, uint16_t* dZ2, int lddz, float* grads, int accumulate,
                           void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(row_weight, "mg_f0_l2tail_rows_bf16: row_weight is null");
    return f0_l2tail_launch("mg_f0_l2tail_rows_bf16", H1, ldh1, K2, W2, ldw2, N2, b2, W3, b3, W4, b4, target, nullptr, row_weight, M, 1,
                            (int)(M < 2147483647LL ? M : 1), grad_scale, pred, loss, dZ2, lddz, grads, accumulate, workspace,
                            workspace_bytes, stream);
}

// mg_f0_l2tail_rows_bf16 without its reduce launch: the workgroups' sums stay in `workspace` as *n_slabs slabs of 32*128 + 32 + 32 + 2
// floats (dW3 | db3 | dW4 | db4 | loss) from its start, for mg_expand_column_reduce_f32 to sum in the same launch that repeats the
// prediction.
int mg_f0_l2tail_rows_slabs_bf16(const uint16_t* H1, int ldh1, int K2, const uint16_t* W2, int ldw2, int N2, const float* b2, const float* W3,
                                 const float* b3, const float* W4, const float* b4, const float* target, const float* row_weight, int64_t M,
                                 float grad_scale, float* pred, uint16_t* dZ2, int lddz, void* workspace, size_t workspace_bytes, int* n_slabs,
                                 void* stream) {
    MG_CHECK_ARG(row_weight && n_slabs, "mg_f0_l2tail_rows_slabs_bf16: null argument");
    return f0_l2tail_launch("mg_f0_l2tail_rows_slabs_bf16", H1, ldh1, K2, W2, ldw2, N2, b2, W3, b3, W4, b4, target, nullptr, row_weight, M, 1,
                            (int)(M < 2147483647LL ? M : 1), grad_scale, pred, nullptr, dZ2, lddz, nullptr, 0, workspace, workspace_bytes, stream,
                            n_slabs);
}

// floats between two slabs of mg_f0_l2tail_rows_slabs_bf16 (a multiple of 4: 16-byte loads for whoever sums them)
int64_t mg_f0_l2tail_slab_stride(void) { return LT_SLAB_STRIDE; }

#ifdef MG_STAMPS
int mg_diag_read_stamps_lt(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_lt), bytes < sizeof(g_stamps_lt) ? bytes : sizeof(g_stamps_lt), 0, hipMemcpyDeviceToHost);
}
#endif

}  // extern "C"
