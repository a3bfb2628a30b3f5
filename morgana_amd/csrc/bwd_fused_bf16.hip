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
//   P1 (all waves)  dZ1s^T[128 j, 32 m] = W2T_s[128 j, N2] . dZ2[32 m, N2]^T  on v_mfma_f32_16x16x32_bf16: wave w owns hidden
//                   units 16 w .. 16 w + 15 (weights as A operand, so a lane holds 4 consecutive units of one frame),
//                   times H1 (1 - H1) from an LDS tile, written as bf16 to the LDS tile the transposed reads of P2 expect.
//                   (First version: 32x32x16 on waves 0-3 with the other four idle and an H1 tile read 16 ways conflicted -
//                   stamped at 2,100 cycles per step, 37 % of the kernel.)
//   P2 (all waves)  the 128 x 640 wgrad step of wgrad_big_kernel (gemm_bf16_big.hip), db1 through the ones fragment.
// X is double buffered (LDS-DMA two steps ahead of use), the small dZ2 / H1 tiles are refilled while P2 runs.
// Deterministic: split-M slabs + ordered reduce (shared with the unfused wgrad).
#include "common.h"
#include "slab_reduce.h"

#include <type_traits>

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
#define F_GROUPS 16                               // X ring: 16 groups of 4 source rows (5 KB each)
#define F_W2T 0                                   // 4 x [128 rows x 64 B]   (swizzled: chunk ^ ((row >> 2) & 3))
#define F_X (F_W2T + 32768)                       // ring of 64 source rows x 1280 B (tr-swizzled by row & 3)
#define F_DZ (F_X + F_GROUPS * 5120)              // 4 x [32 rows x 64 B]     (swizzled like W2T)
#define F_H1 (F_DZ + 8192)                        // [32 m][256 B], chunk c of row m at position c ^ (m & 15)
#define F_YS (F_H1 + 8192)                        // [32 m][256 B] tr-swizzled: the dZ1 tile
#define F_RUNROW (F_YS + 8192)                    // int32[4096]: source row of each run of equal consecutive row indices
#define F_SLOT (F_RUNROW + F_ROWS_MAX * 4)        // uint16[4096]: run ordinal of each frame of this workgroup's range
#define F_LDS (F_SLOT + F_ROWS_MAX * 2)           // 163,840 B: all of the CU's LDS

// The gathered operand X is staged by RUNS, not by frames: consecutive frames of one phone point at the same source row
// (utils.upsample_to_repetitions repeats each phone row dur times; 12.5 frames per phone at C2), so a 32-frame step touches
// about 3 distinct rows.  The ring holds each distinct row once and the transposed fragment reads address it through a
// per-frame slot table; the matrix work is unchanged (every frame is still multiplied), only the L2 -> LDS traffic shrinks:
// 40 KB -> ~5 KB per step.  (Frame-wise staging was stamped at 30 % of the kernel's wave-cycles in LDS-DMA issue: 56 KB per
// step is what a CU's texture-address path delivers in the step's 1,536 matrix cycles.)  Arbitrary `rows` still work - every
// change of index starts a run, the worst case (all frames distinct) stages exactly what the frame-wise version did.
__global__ __launch_bounds__(512) void wgrad_fused_kernel(const uint16_t* __restrict__ dZ2, int lddz, const uint16_t* __restrict__ W2T,
                                                          int ldwt, const uint16_t* __restrict__ H1, int ldh,
                                                          const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                          int64_t M, int N, int K, int m_chunk, float* __restrict__ slab,
                                                          float* __restrict__ bslab, int64_t sstride) {
    constexpr int TKT = 5;
    constexpr int PY = 256, PX = F_BKT * 2;
    constexpr int NX = 5;
    __shared__ __attribute__((aligned(16))) unsigned char smem[F_LDS];
    int* run_row = reinterpret_cast<int*>(smem + F_RUNROW);
    unsigned short* slot_lds = reinterpret_cast<unsigned short*>(smem + F_SLOT);

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

    // ---- one-time: runs of equal consecutive source rows over this workgroup's frames ---------------------------------------
    // Thread t owns frames 8 t .. 8 t + 7 (m_chunk <= 4096); frames past n_rows count as pad frames (row -1: the zero row).
    int n_runs;
    {
        auto src_row = [&](int i) -> int { return (i < n_rows) ? (rows ? rows[m_lo + i] : (int)(m_lo + i)) : -1; };
        int r[8], flag[8], cnt = 0;
        int prev = tid ? src_row(8 * tid - 1) : 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            r[e] = src_row(8 * tid + e);
            flag[e] = (tid == 0 && e == 0) ? 1 : (r[e] != prev);
            prev = r[e];
            cnt += flag[e];
        }
        int incl = cnt;                                   // inclusive scan over the 512 threads: wave scan + wave totals in LDS
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        int* wtot = reinterpret_cast<int*>(smem + F_X);   // the ring is not in use yet
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int base = incl - cnt;
        for (int w = 0; w < wave; ++w) base += wtot[w];
        int total = 0;
        for (int w = 0; w < 8; ++w) total += wtot[w];
        n_runs = __builtin_amdgcn_readfirstlane(total);
        int ord = base - 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ord += flag[e];
            if (8 * tid + e < F_ROWS_MAX) slot_lds[8 * tid + e] = (unsigned short)ord;
            if (flag[e]) run_row[ord] = r[e];
        }
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

    // ---- DMA slots ----------------------------------------------------------------------------------------------------------
    // A group = 4 consecutive runs = 5 KB = 5 pieces; lane l of piece i covers byte 1024 i + 16 l of the group: run x_row[i],
    // 16-byte chunk cpos of its 1280-byte row, stored at chunk position cpos ^ ((run & 3) << 2)  (run & 3 == x_row[i]).
    int x_row[NX], x_off[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int byte = i * 1024 + lane * 16;
        x_row[i] = byte / PX;
        const int cpos = (byte % PX) >> 4;
        x_off[i] = (cpos ^ ((x_row[i] & 3) << 2)) * 16;
    }
    // Source select without a branch: rows < 0 (pad frames) and rows past the end read the zero row.
    const unsigned long long zero_ptr = (unsigned long long)g_fused_zero_row;
    const unsigned long long a_ptr = (unsigned long long)A;
    const unsigned lda_bytes = (unsigned)lda * 2u;
    auto pick = [&](unsigned long long p, int neg) -> const uint16_t* {      // neg < 0 selects the zero row
        const unsigned long long mask = (unsigned long long)(long long)(neg >> 31);
        return (const uint16_t*)((p & ~mask) | (zero_ptr & mask));
    };
    auto issue_group = [&](int g) {                       // called by one wave: runs 4 g .. 4 g + 3 into ring slot g % 16
        unsigned char* st = smem + F_X + (g & (F_GROUPS - 1)) * 5120;
        int rr[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int run = 4 * g + x_row[i];
            rr[i] = run_row[min(run, F_ROWS_MAX - 1)];
            if (run >= n_runs) rr[i] = -1;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i)
            fglds16(pick(a_ptr + (unsigned long long)(unsigned)rr[i] * lda_bytes + (unsigned)x_off[i], rr[i]), st + i * 1024);
    };
    int g_issued = 0;                                     // groups 0 .. g_issued - 1 are in flight or landed (wave-uniform)
    auto issue_upto = [&](int g_last) {
        for (int g = g_issued; g <= g_last; ++g)
            if ((g & 7) == wave) issue_group(g);
        if (g_last >= g_issued) g_issued = g_last + 1;
    };
    auto first_group = [&](int step) -> int { return __builtin_amdgcn_readfirstlane((int)slot_lds[step * 32]) >> 2; };
    auto last_group = [&](int step) -> int { return __builtin_amdgcn_readfirstlane((int)slot_lds[step * 32 + 31]) >> 2; };

    // dZ2 piece `wave`: k-tile kt = wave >> 1, rows 16 (wave & 1) ..; H1 piece `wave`: rows 4 wave .. 4 wave + 3, 16-byte chunk c of
    // row m stored at chunk position c ^ (m & 15) (P1 reads one 8-byte group per frame: 16 frames then cover all banks)
    const int dz_row = 16 * (wave & 1) + (lane >> 2);
    const int dz_col = 32 * (wave >> 1) + 8 * ((lane & 3) ^ ((dz_row >> 2) & 3));
    const int h1_row = 4 * wave + (lane >> 4);
    const int h1_col = n0 + 8 * ((lane & 15) ^ (h1_row & 15));
    auto issue_small = [&](int step) {
        const int mz = step * 32 + dz_row, mh = step * 32 + h1_row;
        fglds16(pick((unsigned long long)(dZ2 + (size_t)(m_lo + mz) * lddz + dz_col), n_rows - 1 - mz), smem + F_DZ + wave * 1024);
        fglds16(pick((unsigned long long)(H1 + (size_t)(m_lo + mh) * ldh + h1_col), n_rows - 1 - mh), smem + F_H1 + wave * 1024);
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
    const int rbase = 8 * (g4 >> 1) + q;                  // this lane's frame inside a 16-frame k-step (and + 4)
    const int sw = q << 2;
    int yoff[2], xk[TKT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = wn0 + i * 32 + cgrp;
        yoff[i] = F_YS + rbase * PY + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
    }
#pragma unroll
    for (int j = 0; j < TKT; ++j) {
        const int col = wk0 + j * 32 + cgrp;
        xk[j] = F_X + (((col >> 3) << 4) | ((col & 7) << 1));          // byte offset inside a ring row before the swizzle
    }
    // P1 geometry: wave w owns hidden units 16 w .. 16 w + 15; 16x16x32 MFMA, A = W2T rows (unit l15, k chunk lq), B = dZ2 rows
    // (frame l15 of half t, k chunk lq); D: this lane holds units 16 w + 4 lq .. + 3 of frame 16 t + l15.
    const int l15 = lane & 15, lq = lane >> 4;
    const int a_row = 16 * wave + l15;
    const int a_off = F_W2T + a_row * 64 + ((lq ^ ((a_row >> 2) & 3)) << 4);        // + 8192 per k-tile
    int b_off[2], h_off[2], y_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = 16 * t + l15;
        const int c = 2 * wave + (lq >> 1);                                           // 16-byte chunk of the 256-byte row
        b_off[t] = F_DZ + m * 64 + ((lq ^ ((m >> 2) & 3)) << 4);                      // + 2048 per k-tile
        h_off[t] = F_H1 + m * 256 + ((c ^ (m & 15)) << 4) + 8 * (lq & 1);
        y_off[t] = F_YS + m * 256 + ((c ^ ((m & 3) << 2)) << 4) + 8 * (lq & 1);
    }

    if (n_steps > 0) {
        issue_upto(last_group(0));
        issue_small(0);
    }
    for (int step = 0; step < n_steps; ++step) {
        MG_STAMP(ta);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");          // A: what was issued has landed; the ring rows,
        MG_STAMP(tb);                                                          //    dZ2 / H1 tiles of the last step are free
        MG_STAMP_ADD(sum_wait_a, tb, ta);
#ifdef MG_STAMPS
        if (step == 0) ts1 = tb;
#endif
        {
            const int need = last_group(step);
            if (need >= g_issued) {                       // only when a step and its successor span more than the ring holds
                issue_upto(need);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (step + 1 < n_steps) issue_upto(min(last_group(step + 1), first_group(step) + F_GROUPS - 1));
        }
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_issue_x, ta, tb);
        {
            // ---- P1: dZ1s^T = W2T_s . dZ2^T over the N2 = 128 outputs of layer 2 (4 k-steps), all operand reads up front ------
            bfv8 a[4], b[2][4];
            bfv4 hv[2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const bfv8*>(smem + a_off + ks * 8192);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) b[t][ks] = *reinterpret_cast<const bfv8*>(smem + b_off[t] + ks * 2048);
#pragma unroll
            for (int t = 0; t < 2; ++t) hv[t] = *reinterpret_cast<const bfv4*>(smem + h_off[t]);
            f32x4 d[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int t = 0; t < 2; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b[t][ks], d[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = (float)hv[t][e];
                    v[e] = d[t][e] * h * (1.f - h);
                }
                const u32x2 pk = u32x2{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                       __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
                *reinterpret_cast<u32x2*>(smem + y_off[t]) = pk;
            }
        }
        // ring addresses of this lane's four frames of the step: row (slot & 63), swizzle (slot & 3) << 6 on the byte offset
        int xbase[4], xswz[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int sl = slot_lds[step * 32 + rbase + 4 * (f & 1) + 16 * (f >> 1)];
            xbase[f] = (sl & (4 * F_GROUPS - 1)) * PX;
            xswz[f] = (sl & 3) << 6;
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
                const unsigned char* alo = smem + xbase[2 * ks] + (xk[j] ^ xswz[2 * ks]);
                const unsigned char* ahi = smem + xbase[2 * ks + 1] + (xk[j] ^ xswz[2 * ks + 1]);
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(alo));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ahi));
                b[j] = bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            if (bias_free) b[TKT - 1] = ones;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TKT; ++j) acc[i][j] = mg_mfma_32x32x16(a[i], b[j], acc[i][j]);
        }
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_p2, ta, tb);
    }

#ifdef MG_STAMPS
    MG_STAMP(ts2);
#endif
    const int lr = lane & 31, lh = lane >> 5;
    float* out = slab + (size_t)s * sstride;        // split s: [N*K weights | N bias sums], sstride floats apart
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
                if (row < N) bslab[(size_t)s * sstride + row] = acc[i][TKT - 1][r];
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


// ---------------------------------------------------------------------------------------------------------------------
// Pipelined form for gathered inputs (rows != NULL).  Same math and the same run staging as above, restructured after the
// in-kernel stamps of the first form (per 32-frame step: 1,536 matrix cycles, 4,600 elapsed - two barriers, LDS round trips
// of the group bookkeeping and P1's operand reads all on the critical path):
//   * the W2^T fragments of P1 are loop invariant: 16 VGPRs per lane, read once from global memory (no LDS tile);
//   * dZ2 / H1 tiles are triple buffered and fetched three steps ahead (vmcnt(2) at the barrier leaves the newest pair in
//     flight: two iterations to arrive from HBM), the dZ1 tile is double buffered: P1 of step + 1 runs beside P2 of step and
//     ONE barrier per step is left;
//   * waves 0-3 run P1 then P2, waves 4-7 P2 then P1: the two waves of a SIMD are in different phases, one's LDS latencies
//     sit under the other's matrix work;
//   * group bounds and the frame -> ring-row slots of the NEXT step are read before the barrier, so no LDS round trip
//     follows it;
//   * the dZ2 tile swizzle is conflict free for the 16x16x32 operand read (chunk ^ 2 ((row >> 3) & 1)).
// Ring: 10 groups = 40 source rows.  With every frame distinct a step can need 9 groups, so the ring still holds any single
// step; it just cannot prefetch then (identity inputs use the kernel above).
// ---------------------------------------------------------------------------------------------------------------------
#define P_GROUPS 10
#define P_X 0                                     // ring of 40 source rows x 1280 B (tr-swizzled by row & 3)
#define P_DZ (P_X + P_GROUPS * 5120)              // 3 buffers x (4 k-tiles x [32 rows x 64 B]), chunk ^ 2 ((row >> 3) & 1)
#define P_H1 (P_DZ + 3 * 8192)                    // 3 buffers x [32 m][256 B], chunk c of row m at position c ^ (m & 15)
#define P_YS (P_H1 + 3 * 8192)                    // 2 buffers x [32 m][256 B] tr-swizzled: the dZ1 tile
#define P_RUNROW (P_YS + 2 * 8192)
#define P_SLOT (P_RUNROW + F_ROWS_MAX * 4)
#define P_LDS (P_SLOT + F_ROWS_MAX * 2 + 64)      // + a pad row of the slot table for the look-ahead past the last step
MG_STAMP_DECL(g_stamps_fp);

__global__ __launch_bounds__(512) void wgrad_fused_pipe_kernel(const uint16_t* __restrict__ dZ2, int lddz, const uint16_t* __restrict__ W2T,
                                                               int ldwt, const uint16_t* __restrict__ H1, int ldh,
                                                               const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                               int64_t M, int N, int K, int m_chunk, int n_splits,
                                                               float* __restrict__ slab, float* __restrict__ bslab, int64_t sstride) {
    constexpr int TKT = 5;
    constexpr int PY = 256, PX = F_BKT * 2;
    constexpr int NX = 5;
    __shared__ __attribute__((aligned(16))) unsigned char smem[P_LDS];
    int* run_row = reinterpret_cast<int*>(smem + P_RUNROW);
    unsigned short* slot_lds = reinterpret_cast<unsigned short*>(smem + P_SLOT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MG_STAMPS
    unsigned long long ts0, ts1 = 0, ts2, ts3, tr0, tr1, ta, tb, sum_wait_a = 0, sum_issue_x = 0, sum_body = 0, sum_p2 = 0, sum_first = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    const int wn0 = (wave >> 2) * 64;
    const int wk0 = (wave & 3) * (TKT * 32);
    const int tiles_n = N / F_BNT;
    // Blocks b, b + 8, b + 16, ... share an XCD and its L2: the tiles_n workgroups of one frame range go there, so the range's
    // dZ2 tiles leave HBM once instead of once per n tile (FETCH_SIZE x 2 of the n-tile-fastest order: 619 MB per launch
    // against 327 MB of dZ2 + H1).
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int n0 = (jq % tiles_n) * F_BNT;
    const int s = (jq / tiles_n) * 8 + xcd;
    if (s >= n_splits) return;
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t m_hi = min(M, m_lo + (int64_t)m_chunk);
    const int n_rows = m_hi > m_lo ? (int)(m_hi - m_lo) : 0;

    // ---- one-time: runs of equal consecutive source rows over this workgroup's frames ---------------------------------------
    // Thread t owns frames 8 t .. 8 t + 7 (m_chunk <= 4096); frames past n_rows count as pad frames (row -1: the zero row).
    int n_runs;
    {
        int r[9], flag[8], cnt = 0;
#pragma unroll
        for (int e = 0; e < 9; ++e) {                     // r[0] is the frame before this thread's first one
            const int i = 8 * tid + e - 1;
            const int64_t m = m_lo + min(max(i, 0), max(n_rows - 1, 0));
            const int v = rows[min(m, M - 1)];
            r[e] = (i >= 0 && i < n_rows) ? v : -1;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            flag[e] = (tid == 0 && e == 0) ? 1 : (r[e + 1] != r[e]);
            cnt += flag[e];
        }
        int incl = cnt;                                   // inclusive scan over the 512 threads: wave scan + wave totals in LDS
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        int* wtot = reinterpret_cast<int*>(smem + P_X);   // the ring is not in use yet
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int base = incl - cnt;
        for (int w = 0; w < wave; ++w) base += wtot[w];
        int total = 0;
        for (int w = 0; w < 8; ++w) total += wtot[w];
        n_runs = __builtin_amdgcn_readfirstlane(total);
        int ord = base - 1;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ord += flag[e];
            slot_lds[8 * tid + e] = (unsigned short)ord;
            if (flag[e]) run_row[ord] = r[e + 1];
        }
        if (tid < 32) slot_lds[F_ROWS_MAX + tid] = 0;     // pad row: look-ahead reads past the last step
    }
    __syncthreads();

    // ---- DMA slots ----------------------------------------------------------------------------------------------------------
    // A group = 4 consecutive runs = 5 KB = 5 pieces; lane l of piece i covers byte 1024 i + 16 l of the group: run x_row[i],
    // 16-byte chunk cpos of its 1280-byte row, stored at chunk position cpos ^ ((run & 3) << 2)  (run & 3 == x_row[i]).
    auto x_slot = [&](int i, int& xr, int& xo) {
        const int byte = i * 1024 + lane * 16;
        xr = byte / PX;
        const int cpos = (byte - xr * PX) >> 4;
        xo = (cpos ^ ((xr & 3) << 2)) * 16;
    };
    // Source select without a branch: rows < 0 (pad frames) and rows past the end read the zero row.
    const unsigned long long zero_ptr = (unsigned long long)g_fused_zero_row;
    const unsigned long long a_ptr = (unsigned long long)A;
    const unsigned lda_bytes = (unsigned)lda * 2u;
    auto pick = [&](unsigned long long p, int neg) -> const uint16_t* {      // neg < 0 selects the zero row
        const unsigned long long mask = (unsigned long long)(long long)(neg >> 31);
        return (const uint16_t*)((p & ~mask) | (zero_ptr & mask));
    };
    auto issue_group = [&](int g) {                       // called by one wave: runs 4 g .. 4 g + 3 into ring slot g % P_GROUPS
        unsigned char* st = smem + P_X + (g % P_GROUPS) * 5120;
        int rr[NX], xo[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            int xr;
            x_slot(i, xr, xo[i]);
            const int run = 4 * g + xr;
            rr[i] = run_row[min(run, F_ROWS_MAX - 1)];
            if (run >= n_runs) rr[i] = -1;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i)
            fglds16(pick(a_ptr + (unsigned long long)(unsigned)rr[i] * lda_bytes + (unsigned)xo[i], rr[i]), st + i * 1024);
    };
    int g_issued = 0;                                     // groups 0 .. g_issued - 1 are in flight or landed (wave-uniform)
    auto issue_upto = [&](int g_last) {
        for (int g = g_issued; g <= g_last; ++g)
            if ((g & 7) == wave) issue_group(g);
        if (g_last >= g_issued) g_issued = g_last + 1;
    };
    auto group_of = [&](int frame) -> int { return __builtin_amdgcn_readfirstlane((int)slot_lds[frame]) >> 2; };

    // dZ2 piece `wave`: k-tile kt = wave >> 1, rows 16 (wave & 1) ..; H1 piece `wave`: rows 4 wave .. 4 wave + 3
    const int dz_row = 16 * (wave & 1) + (lane >> 2);
    const int dz_col = 32 * (wave >> 1) + 8 * ((lane & 3) ^ (((dz_row >> 3) & 1) << 1));
    const int h1_row = 4 * wave + (lane >> 4);
    const int h1_col = n0 + 8 * ((lane & 15) ^ (h1_row & 15));
    auto issue_small = [&](int step, int buf) {           // tiles of `step` into buffer buf (= step % 3)
        const int mz = step * 32 + dz_row, mh = step * 32 + h1_row;
        fglds16(pick((unsigned long long)(dZ2 + (size_t)(m_lo + mz) * lddz + dz_col), n_rows - 1 - mz),
                smem + P_DZ + buf * 8192 + wave * 1024);
        fglds16(pick((unsigned long long)(H1 + (size_t)(m_lo + mh) * ldh + h1_col), n_rows - 1 - mh),
                smem + P_H1 + buf * 8192 + wave * 1024);
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
    const int rbase = 8 * (g4 >> 1) + q;                  // this lane's frame inside a 16-frame k-step (and + 4)
    const int sw = q << 2;
    int yoff[2], xk[TKT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = wn0 + i * 32 + cgrp;
        yoff[i] = P_YS + rbase * PY + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
    }
#pragma unroll
    for (int j = 0; j < TKT; ++j) {
        const int col = wk0 + j * 32 + cgrp;
        xk[j] = P_X + (((col >> 3) << 4) | ((col & 7) << 1));          // byte offset inside a ring row before the swizzle
    }
    // P1 geometry: wave w owns hidden units 16 w .. 16 w + 15; 16x16x32 MFMA, A = W2T rows (unit l15, k chunk lq), B = dZ2 rows
    // (frame l15 of half t, k chunk lq); D: this lane holds units 16 w + 4 lq .. + 3 of frame 16 t + l15.
    const int l15 = lane & 15, lq = lane >> 4;
    bfv8 w2[4];                                           // loop invariant: W2^T[n0 + 16 w + l15][32 ks + 8 lq ..]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        w2[ks] = *reinterpret_cast<const bfv8*>(W2T + (size_t)(n0 + 16 * wave + l15) * ldwt + 32 * ks + 8 * lq);
    // Land them here and re-define the registers behind the wait: otherwise hipcc, which cannot see across the loop's back
    // edge that these loads are long complete, puts s_waitcnt vmcnt(3..0) in front of P1's MFMAs inside the loop and drains
    // the LDS-DMA prefetches every step.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(w2[ks]));
    // frame m = 16 t + l15: the t = 1 offsets are the t = 0 ones plus a constant (the swizzles only look at m & 15 / (m >> 3) & 1)
    const int c16 = 2 * wave + (lq >> 1);                                             // 16-byte chunk of the 256-byte row
    const int b_off0 = P_DZ + l15 * 64 + ((lq ^ (((l15 >> 3) & 1) << 1)) << 4);       // + 1024 t + 2048 per k-tile
    const int h_off0 = P_H1 + l15 * 256 + ((c16 ^ l15) << 4) + 8 * (lq & 1);          // + 4096 t
    const int y_off0 = P_YS + l15 * 256 + ((c16 ^ ((l15 & 3) << 2)) << 4) + 8 * (lq & 1);

    // P1 of one step: dZ2 / H1 tiles of buffer tb3 (of 3), dZ1 tile to buffer yb2 (of 2).
    auto p1 = [&](int tb3, int yb2) {
        const int bo = tb3 * 8192, yo = yb2 * 8192;
        bfv8 b[2][4];
        bfv4 hv[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) b[t][ks] = *reinterpret_cast<const bfv8*>(smem + bo + b_off0 + t * 1024 + ks * 2048);
#pragma unroll
        for (int t = 0; t < 2; ++t) hv[t] = *reinterpret_cast<const bfv4*>(smem + bo + h_off0 + t * 4096);
        f32x4 d[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int t = 0; t < 2; ++t) d[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[ks], b[t][ks], d[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (float)hv[t][e];
                v[e] = d[t][e] * h * (1.f - h);
            }
            const u32x2 pk = u32x2{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                   __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
            *reinterpret_cast<u32x2*>(smem + yo + y_off0 + t * 4096) = pk;
        }
    };
    // P2 of one step: the dZ1 tile of buffer yb2 against the ring rows named by the four slot values of this lane.
    auto p2 = [&](int yb2, const unsigned (&slp)[2]) {     // slp: the four 16-bit slot values, packed in pairs
        int xbase[4], xswz[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int sl = (f & 1) ? (int)(slp[f >> 1] >> 16) : (int)(slp[f >> 1] & 0xffffu);
            const int g = sl >> 2;
            const int gm = g - P_GROUPS * ((g * 6554) >> 16);                 // g % 10 for g < 16384
            xbase[f] = (gm * 4 + (sl & 3)) * PX;
            xswz[f] = (sl & 3) << 6;
        }
        const int yb = yb2 * 8192;
        // Fragment reads run two MFMA pairs ahead of their use (a rolling window instead of "all 14 reads, wait, 10 MFMAs" per
        // k-step: with two waves per SIMD the LDS latency at the head of each k-step was exposed twice per step).
        auto rd_a = [&](int ks, int i) -> bfv8 {
            const unsigned char* ad = smem + yb + yoff[i] + ks * 16 * PY;
            const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
            const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PY));
            return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto rd_b = [&](int ks, int j) -> bfv8 {
            if (bias_free && j == TKT - 1) return ones;
            const unsigned char* alo = smem + xbase[2 * ks] + (xk[j] ^ xswz[2 * ks]);
            const unsigned char* ahi = smem + xbase[2 * ks + 1] + (xk[j] ^ xswz[2 * ks + 1]);
            const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(alo));
            const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ahi));
            return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        bfv8 a[2][2], b[2 * TKT];
        a[0][0] = rd_a(0, 0);
        a[0][1] = rd_a(0, 1);
        b[0] = rd_b(0, 0);
        b[1] = rd_b(0, 1);
        auto mm = [&](auto tc) {                           // t = ks * TKT + j, a compile-time constant
            constexpr int t = decltype(tc)::value;
            constexpr int ks = t / TKT, j = t % TKT;
            if constexpr (t + 2 < 2 * TKT) b[t + 2] = rd_b((t + 2) / TKT, (t + 2) % TKT);
            if constexpr (t == 2) {
                a[1][0] = rd_a(1, 0);
                a[1][1] = rd_a(1, 1);
            }
            acc[0][j] = mg_mfma_32x32x16(a[ks][0], b[t], acc[0][j]);
            acc[1][j] = mg_mfma_32x32x16(a[ks][1], b[t], acc[1][j]);
            constexpr int n_reads = (t + 2 < 2 * TKT ? 2 : 0) + (t == 2 ? 4 : 0);
            if constexpr (n_reads > 0) __builtin_amdgcn_sched_group_barrier(0x100, n_reads, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MG_MFMA_PER_TILE, 0);
        };
        mm(std::integral_constant<int, 0>{});
        mm(std::integral_constant<int, 1>{});
        mm(std::integral_constant<int, 2>{});
        mm(std::integral_constant<int, 3>{});
        mm(std::integral_constant<int, 4>{});
        mm(std::integral_constant<int, 5>{});
        mm(std::integral_constant<int, 6>{});
        mm(std::integral_constant<int, 7>{});
        mm(std::integral_constant<int, 8>{});
        mm(std::integral_constant<int, 9>{});
    };
    auto read_slots = [&](int step, unsigned (&slp)[2]) {  // frames of the step this lane feeds to the transposed reads
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned lo = slot_lds[step * 32 + rbase + 16 * h], hi = slot_lds[step * 32 + rbase + 4 + 16 * h];
            slp[h] = lo | (hi << 16);
        }
    };

    // ---- prologue: ring rows of steps 0 and 1, tiles of steps 0, 1, 2, P1 of step 0 -----------------------------------------
    unsigned sl_cur[2] = {0u, 0u};
    int lg0 = 0, lg1 = 0, lg2 = 0, fg0 = 0;               // last group of step, step + 1, step + 2; first group of step
    if (n_steps > 0) {
        lg0 = group_of(31);
        lg1 = group_of(min(1, n_steps - 1) * 32 + 31);
        lg2 = group_of(min(2, n_steps - 1) * 32 + 31);
        issue_upto(lg0);
        issue_small(0, 0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        p1(0, 0);
        issue_upto(min(lg1, fg0 + P_GROUPS - 1));
        issue_small(1, 1);                                // past the end: zero rows (pick), never read
        issue_small(2, 2);
        read_slots(0, sl_cur);
    }
    int tb_next = 1;                                      // tile buffer of step + 1 (= (step + 1) % 3)
    for (int step = 0; step < n_steps; ++step) {
        MG_STAMP(ta);
        // ring rows of step + 1 and tiles of step + 1 landed (the pair of step + 2 may still fly), dZ1(step) and the
        // look-ahead reads complete; the buffers of the last step are free
        asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_wait_a, tb, ta);
#ifdef MG_STAMPS
        if (step == 0) ts1 = tb;
#endif
        if (lg0 >= g_issued) {                            // only when a step and its successor span more than the ring holds
            issue_upto(lg0);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // Fetches of this iteration (ring rows of step + 2, tiles of step + 3): the lower half of the waves issues them now,
        // the upper half after its matrix work - 16+ LDS-DMA instructions issued by all waves at once behind the barrier
        // serialise in the texture-address path (~45 cycles each, stamped) while every matrix pipe waits.
        const int ring_last = min(lg2, fg0 + P_GROUPS - 1);
        const int tb_new = tb_next == 0 ? 2 : tb_next - 1;              // (step + 3) % 3 == step % 3
        if (wave < 4) {
            if (step + 1 < n_steps) issue_upto(ring_last);
            issue_small(step + 3, tb_new);
        }
        // look-ahead (consumed after the next barrier): bounds of the coming steps, slots of step + 1
        const int la_last = (int)slot_lds[min(step + 3, n_steps - 1) * 32 + 31];
        const int la_first = (int)slot_lds[min(step + 1, n_steps - 1) * 32];
        unsigned sl_next[2];
        read_slots(step + 1, sl_next);                    // step + 1 == n_steps reads the pad row
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_issue_x, ta, tb);
#ifdef MG_STAMPS
        unsigned long long tc, td;
#endif
        if (wave < 4 && step + 1 < n_steps) p1(tb_next, (step + 1) & 1);
        MG_STAMP(tc);
        p2(step & 1, sl_cur);
        MG_STAMP(td);
        if (wave >= 4) {
            if (step + 1 < n_steps) p1(tb_next, (step + 1) & 1);
            if (step + 1 < n_steps) issue_upto(ring_last);
            issue_small(step + 3, tb_new);
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_body, tb, ta);
        MG_STAMP_ADD(sum_p2, td, tc);
        MG_STAMP_ADD(sum_first, tc, ta);
        lg0 = lg1;
        lg1 = lg2;
        lg2 = __builtin_amdgcn_readfirstlane(la_last) >> 2;
        fg0 = __builtin_amdgcn_readfirstlane(la_first) >> 2;
        sl_cur[0] = sl_next[0];
        sl_cur[1] = sl_next[1];
        tb_next = tb_next == 2 ? 0 : tb_next + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tiles fetched past the end

#ifdef MG_STAMPS
    MG_STAMP(ts2);
#endif
    const int lr = lane & 31, lh = lane >> 5;
    float* out = slab + (size_t)s * sstride;        // split s: [N*K weights | N bias sums], sstride floats apart
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
                if (row < N) bslab[(size_t)s * sstride + row] = acc[i][TKT - 1][r];
            }
    }
#ifdef MG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG_STAMP(ts3);
    MG_STAMP_REAL(tr1);
    const int sb = blockIdx.x;
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 2, ts2);
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 3, ts3);
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 4, tr0);
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 5, tr1);
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 6, sum_wait_a);
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 7, sum_issue_x);
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 8, sum_body);
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 9, sum_p2);
    MG_STAMP_STORE(g_stamps_fp, sb, wave, lane, 10, sum_first);
#endif
}


#ifdef MG_EXPERIMENTS      // lab builds only (make lab / diag): the one-wave-per-SIMD experiment, measured slower
// ---------------------------------------------------------------------------------------------------------------------
// "Solo" form of the pipelined kernel: ONE wave per SIMD (256 threads, up to 512 registers per lane), P1 of step + 1 woven into P2 of
// step inside one instruction stream.  Why (profiles/r2_stamps_fused.txt, cycles per 32-frame step of the 8-wave kernel, ~4,050 for
// 2 x 768 matrix cycles per SIMD): every wave spends ~950 cycles in P1 - an LDS round trip, two 4-deep MFMA chains, ~25 dependent
// VALU instructions and an LDS store for 128 matrix cycles - and ~700 issuing its share of the DMA, phases in which it feeds nothing
// to the matrix pipe; its SIMD partner covers only part of that because it has the same phases, and at 254 registers (160 of them
// accumulators) there is no room to interleave P1 with P2 inside a wave.  With one wave per SIMD the budget doubles: a wave owns all
// 128 rows x 160 columns of the tile (320 accumulator registers), runs P1 for 32 of the 128 hidden units (W2^T fragments resident: 32
// registers), and P1's reads, its 16 small MFMAs, its VALU tail and its LDS stores are placed between the groups of P2's 40 MFMAs,
// whose operands are read two groups ahead: the chain of P1 never stands alone.  Same LDS layout, run staging, DMA ring, slabs and
// reduce as the kernel above; 4 tr-reads fewer per MFMA than two waves per SIMD (36 per 40 MFMAs against 28 per 20).
// MEASURED SLOWER - an experiment (MG_TUNE_FORM = 8), correct (dW, db bit-identical to the kernel above, also for row maps
// without runs), not the product path: 492 us against 262 at C2.  320 accumulator registers leave 192 of the 256 VGPRs (the
// accumulator file holds 256) for everything else and hipcc spills 60 of them; and the probe with a 128 x 512 tile (TKT = 4: 256
// accumulators, no spill, 4/5 of the matrix work; MG_TUNE_FORM = 9, results incomplete) still takes 304 us - a lone in-order wave
// whose LDS counter is shared by the operand reads of P2, the reads and stores of P1 and the look-ahead stands at s_waitcnt lgkmcnt
// between every group; without a hand-placed stream the partner wave of the 8-wave form hides more than the wider budget buys.
// ---------------------------------------------------------------------------------------------------------------------
// TKT = 4 is a TIMING PROBE (the tile then covers 512 of the 640 columns: results incomplete): what the form does without spills.
template <int TKT>
__global__ __launch_bounds__(256, 1) void wgrad_fused_solo_kernel(const uint16_t* __restrict__ dZ2, int lddz, const uint16_t* __restrict__ W2T,
                                                                  int ldwt, const uint16_t* __restrict__ H1, int ldh,
                                                                  const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                                  int64_t M, int N, int K, int m_chunk, int n_splits,
                                                                  float* __restrict__ slab, float* __restrict__ bslab, int64_t sstride) {
    constexpr int PY = 256, PX = F_BKT * 2;
    constexpr int NX = 5;
    __shared__ __attribute__((aligned(16))) unsigned char smem[P_LDS];
    int* run_row = reinterpret_cast<int*>(smem + P_RUNROW);
    unsigned short* slot_lds = reinterpret_cast<unsigned short*>(smem + P_SLOT);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk0 = wave * (TKT * 32);
    const int tiles_n = N / F_BNT;
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int n0 = (jq % tiles_n) * F_BNT;
    const int s = (jq / tiles_n) * 8 + xcd;
    if (s >= n_splits) return;
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t m_hi = min(M, m_lo + (int64_t)m_chunk);
    const int n_rows = m_hi > m_lo ? (int)(m_hi - m_lo) : 0;

    // ---- one-time: runs of equal consecutive source rows over this workgroup's frames (thread t owns frames 16 t .. 16 t + 15) ----
    int n_runs;
    {
        int r[17], flag[16], cnt = 0;
#pragma unroll
        for (int e = 0; e < 17; ++e) {                    // r[0] is the frame before this thread's first one
            const int i = 16 * tid + e - 1;
            const int64_t m = m_lo + min(max(i, 0), max(n_rows - 1, 0));
            const int v = rows[min(m, M - 1)];
            r[e] = (i >= 0 && i < n_rows) ? v : -1;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            flag[e] = (tid == 0 && e == 0) ? 1 : (r[e + 1] != r[e]);
            cnt += flag[e];
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        int* wtot = reinterpret_cast<int*>(smem + P_X);   // the ring is not in use yet
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int base = incl - cnt;
        for (int w = 0; w < wave; ++w) base += wtot[w];
        int total = 0;
        for (int w = 0; w < 4; ++w) total += wtot[w];
        n_runs = __builtin_amdgcn_readfirstlane(total);
        int ord = base - 1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            ord += flag[e];
            slot_lds[16 * tid + e] = (unsigned short)ord;
            if (flag[e]) run_row[ord] = r[e + 1];
        }
        if (tid < 32) slot_lds[F_ROWS_MAX + tid] = 0;     // pad row: look-ahead reads past the last step
    }
    __syncthreads();

    // ---- DMA slots (as the kernel above) ------------------------------------------------------------------------------------------
    auto x_slot = [&](int i, int& xr, int& xo) {
        const int byte = i * 1024 + lane * 16;
        xr = byte / PX;
        const int cpos = (byte - xr * PX) >> 4;
        xo = (cpos ^ ((xr & 3) << 2)) * 16;
    };
    const unsigned long long zero_ptr = (unsigned long long)g_fused_zero_row;
    const unsigned long long a_ptr = (unsigned long long)A;
    const unsigned lda_bytes = (unsigned)lda * 2u;
    auto pick = [&](unsigned long long p, int neg) -> const uint16_t* {      // neg < 0 selects the zero row
        const unsigned long long mask = (unsigned long long)(long long)(neg >> 31);
        return (const uint16_t*)((p & ~mask) | (zero_ptr & mask));
    };
    auto issue_group = [&](int g) {                       // called by one wave: runs 4 g .. 4 g + 3 into ring slot g % P_GROUPS
        unsigned char* st = smem + P_X + (g % P_GROUPS) * 5120;
        int rr[NX], xo[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            int xr;
            x_slot(i, xr, xo[i]);
            const int run = 4 * g + xr;
            rr[i] = run_row[min(run, F_ROWS_MAX - 1)];
            if (run >= n_runs) rr[i] = -1;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i)
            fglds16(pick(a_ptr + (unsigned long long)(unsigned)rr[i] * lda_bytes + (unsigned)xo[i], rr[i]), st + i * 1024);
    };
    int g_issued = 0;
    auto issue_upto = [&](int g_last) {
        for (int g = g_issued; g <= g_last; ++g)
            if ((g & 3) == wave) issue_group(g);
        if (g_last >= g_issued) g_issued = g_last + 1;
    };
    auto group_of = [&](int frame) -> int { return __builtin_amdgcn_readfirstlane((int)slot_lds[frame]) >> 2; };

    // small tiles: dZ2 piece pp (k-tile pp >> 1, rows 16 (pp & 1) ..) and H1 piece pp (rows 4 pp .. 4 pp + 3), pp = wave and wave + 4
    int dz_row[2], dz_col[2], h1_row[2], h1_col[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int pp = wave + 4 * u;
        dz_row[u] = 16 * (pp & 1) + (lane >> 2);
        dz_col[u] = 32 * (pp >> 1) + 8 * ((lane & 3) ^ (((dz_row[u] >> 3) & 1) << 1));
        h1_row[u] = 4 * pp + (lane >> 4);
        h1_col[u] = n0 + 8 * ((lane & 15) ^ (h1_row[u] & 15));
    }
    auto issue_small = [&](int step, int buf) {           // tiles of `step` into buffer buf (= step % 3): 4 pieces per wave
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pp = wave + 4 * u;
            const int mz = step * 32 + dz_row[u], mh = step * 32 + h1_row[u];
            fglds16(pick((unsigned long long)(dZ2 + (size_t)(m_lo + mz) * lddz + dz_col[u]), n_rows - 1 - mz),
                    smem + P_DZ + buf * 8192 + pp * 1024);
            fglds16(pick((unsigned long long)(H1 + (size_t)(m_lo + mh) * ldh + h1_col[u]), n_rows - 1 - mh),
                    smem + P_H1 + buf * 8192 + pp * 1024);
        }
    };

    f32x16 acc[4][TKT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TKT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool bias_free = bslab != nullptr && wave == 3;               // K <= 608: the last 32-column tile of k-wave 3 is padding
    const __bf16 one_bf = (__bf16)1.0f;
    const bfv8 ones = bfv8{one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf, one_bf};

    const int n_steps = (n_rows + 31) / 32;
    const int li = lane & 15, g4 = lane >> 4;
    const int q = li >> 2, p4 = li & 3;
    const int cgrp = 16 * (g4 & 1) + 4 * p4;
    const int rbase = 8 * (g4 >> 1) + q;                  // this lane's frame inside a 16-frame k-step (and + 4)
    const int sw = q << 2;
    int yoff[4], xk[TKT];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = i * 32 + cgrp;
        yoff[i] = P_YS + rbase * PY + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
    }
#pragma unroll
    for (int j = 0; j < TKT; ++j) {
        const int col = wk0 + j * 32 + cgrp;
        xk[j] = P_X + (((col >> 3) << 4) | ((col & 7) << 1));
    }
    // P1 geometry: wave w owns hidden units 32 w .. 32 w + 31 as two 16-unit blocks ub; 16x16x32 MFMA, A = W2T rows (unit l15, k chunk
    // lq), B = dZ2 rows (frame l15 of half t, k chunk lq); D: this lane holds units 32 w + 16 ub + 4 lq .. + 3 of frame 16 t + l15.
    const int l15 = lane & 15, lq = lane >> 4;
    bfv8 w2[2][4];
#pragma unroll
    for (int ub = 0; ub < 2; ++ub)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            w2[ub][ks] = *reinterpret_cast<const bfv8*>(W2T + (size_t)(n0 + 32 * wave + 16 * ub + l15) * ldwt + 32 * ks + 8 * lq);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // landed here, re-defined behind the wait (see the kernel above)
#pragma unroll
    for (int ub = 0; ub < 2; ++ub)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(w2[ub][ks]));
    const int b_off0 = P_DZ + l15 * 64 + ((lq ^ (((l15 >> 3) & 1) << 1)) << 4);       // + 1024 t + 2048 per k-tile
    int h_off0[2], y_off0[2];
#pragma unroll
    for (int ub = 0; ub < 2; ++ub) {
        const int c16 = 4 * wave + 2 * ub + (lq >> 1);                                // 16-byte chunk of the 256-byte row
        h_off0[ub] = P_H1 + l15 * 256 + ((c16 ^ l15) << 4) + 8 * (lq & 1);             // + 4096 t
        y_off0[ub] = P_YS + l15 * 256 + ((c16 ^ ((l15 & 3) << 2)) << 4) + 8 * (lq & 1);
    }

    // P1 of one 16-unit block in two pieces (operand reads + MFMAs; the VALU tail + LDS store), so that the step loop can place them
    // between P2's MFMA groups; the dZ2 fragments are read per block (twice per step: 16 ds_read_b128) to keep few registers live
    f32x4 pd[2];
    auto p1_mfma = [&](int ub, int tb3) {
        const int bo = tb3 * 8192;
        bfv8 pb[2][2];                                     // k-tile ks + 1 is read while k-tile ks multiplies
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            pb[0][t] = *reinterpret_cast<const bfv8*>(smem + bo + b_off0 + t * 1024);
            pd[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
#pragma unroll
                for (int t = 0; t < 2; ++t) pb[(ks + 1) & 1][t] = *reinterpret_cast<const bfv8*>(smem + bo + b_off0 + t * 1024 + (ks + 1) * 2048);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) pd[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[ub][ks], pb[ks & 1][t], pd[t], 0, 0, 0);
        }
    };
    auto p1_finish = [&](int ub, int tb3, int yb2) {
        const int bo = tb3 * 8192, yo = yb2 * 8192;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bfv4 hv = *reinterpret_cast<const bfv4*>(smem + bo + h_off0[ub] + t * 4096);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (float)hv[e];
                v[e] = pd[t][e] * h * (1.f - h);
            }
            const u32x2 pk = u32x2{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                   __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
            *reinterpret_cast<u32x2*>(smem + yo + y_off0[ub] + t * 4096) = pk;
        }
    };
    auto read_slots = [&](int step, unsigned (&slp)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned lo = slot_lds[step * 32 + rbase + 16 * h], hi = slot_lds[step * 32 + rbase + 4 + 16 * h];
            slp[h] = lo | (hi << 16);
        }
    };

    // ---- prologue: ring rows of steps 0 and 1, tiles of steps 0, 1, 2, P1 of step 0 -----------------------------------------
    unsigned sl_cur[2] = {0u, 0u};
    int lg0 = 0, lg1 = 0, lg2 = 0, fg0 = 0;
    if (n_steps > 0) {
        lg0 = group_of(31);
        lg1 = group_of(min(1, n_steps - 1) * 32 + 31);
        lg2 = group_of(min(2, n_steps - 1) * 32 + 31);
        issue_upto(lg0);
        issue_small(0, 0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        p1_mfma(0, 0);
        p1_finish(0, 0, 0);
        p1_mfma(1, 0);
        p1_finish(1, 0, 0);
        issue_upto(min(lg1, fg0 + P_GROUPS - 1));
        issue_small(1, 1);                                // past the end: zero rows (pick), never read
        issue_small(2, 2);
        read_slots(0, sl_cur);
    }
    int tb_next = 1;                                      // tile buffer of step + 1 (= (step + 1) % 3)
    for (int step = 0; step < n_steps; ++step) {
        // ring rows and tiles of step + 1 landed (the 4 pieces of step + 2's tiles may still fly), dZ1(step) and the look-ahead reads
        // complete; the buffers of the last step are free
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (lg0 >= g_issued) {                            // only when a step and its successor span more than the ring holds
            issue_upto(lg0);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const int ring_last = min(lg2, fg0 + P_GROUPS - 1);
        const int tb_new = tb_next == 0 ? 2 : tb_next - 1;              // (step + 3) % 3 == step % 3
        if (step + 1 < n_steps) issue_upto(ring_last);
        issue_small(step + 3, tb_new);
        const int la_last = (int)slot_lds[min(step + 3, n_steps - 1) * 32 + 31];
        const int la_first = (int)slot_lds[min(step + 1, n_steps - 1) * 32];
        unsigned sl_next[2];
        read_slots(step + 1, sl_next);                    // step + 1 == n_steps reads the pad row
        const bool more = step + 1 < n_steps;             // P1 of the next step (wave-uniform)

        // ---- P2 of this step with P1 of the next one between its MFMA groups ---------------------------------------------------
        int xbase[4], xswz[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int sl = (f & 1) ? (int)(sl_cur[f >> 1] >> 16) : (int)(sl_cur[f >> 1] & 0xffffu);
            const int g = sl >> 2;
            const int gm = g - P_GROUPS * ((g * 6554) >> 16);                 // g % 10 for g < 16384
            xbase[f] = (gm * 4 + (sl & 3)) * PX;
            xswz[f] = (sl & 3) << 6;
        }
        const int yb = (step & 1) * 8192, ynext = (step + 1) & 1;
        auto rd_a = [&](int ks, int i) -> bfv8 {
            const unsigned char* ad = smem + yb + yoff[i] + ks * 16 * PY;
            const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
            const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PY));
            return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto rd_b = [&](int ks, int j) -> bfv8 {
            if (bias_free && j == TKT - 1) return ones;
            const unsigned char* alo = smem + xbase[2 * ks] + (xk[j] ^ xswz[2 * ks]);
            const unsigned char* ahi = smem + xbase[2 * ks + 1] + (xk[j] ^ xswz[2 * ks + 1]);
            const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(alo));
            const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ahi));
            return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        bfv8 fa[4], fbq[2 * TKT];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = rd_a(0, i);
        fbq[0] = rd_b(0, 0);
        fbq[1] = rd_b(0, 1);
#pragma unroll
        for (int t = 0; t < 2 * TKT; ++t) {               // group t: k-step t / TKT, column block t % TKT, 4 MFMAs
            const int j = t % TKT;
            if (t + 2 < 2 * TKT) fbq[t + 2] = rd_b((t + 2) / TKT, (t + 2) % TKT);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = mg_mfma_32x32x16(fa[i], fbq[t], acc[i][j]);
            if (t == TKT - 1) {                           // the dZ1 fragments of the second k-step take the registers of the first:
#pragma unroll                                            // their LDS latency passes under P1's second block right behind
                for (int i = 0; i < 4; ++i) fa[i] = rd_a(1, i);
            }
            if (more) {                                   // P1 of the next step, one 16-unit block at a time
                if (t == 0) p1_mfma(0, tb_next);
                if (t == 3) p1_finish(0, tb_next, ynext);
                if (t == 4) p1_mfma(1, tb_next);
                if (t == 7) p1_finish(1, tb_next, ynext);
            }
            __builtin_amdgcn_sched_barrier(0);            // groups stay in this order: hoisted operand reads cost registers that are not there
        }
        lg0 = lg1;
        lg1 = lg2;
        lg2 = __builtin_amdgcn_readfirstlane(la_last) >> 2;
        fg0 = __builtin_amdgcn_readfirstlane(la_first) >> 2;
        sl_cur[0] = sl_next[0];
        sl_cur[1] = sl_next[1];
        tb_next = tb_next == 2 ? 0 : tb_next + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tiles fetched past the end

    const int lr = lane & 31, lh = lane >> 5;
    float* out = slab + (size_t)s * sstride;        // split s: [N*K weights | N bias sums], sstride floats apart
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < TKT; ++j) {
            const int col = wk0 + j * 32 + lr;
            if (col >= K) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < N) out[(size_t)row * K + col] = acc[i][j][r];
            }
        }
    }
    if (bias_free && lr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (row < N) bslab[(size_t)s * sstride + row] = acc[i][TKT - 1][r];
            }
    }
}


#endif  // MG_EXPERIMENTS

void mg_launch_fused64(int nbt, const uint16_t* dZ2, int lddz, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh, const uint16_t* A,
                       int lda, const int32_t* rows, int64_t M, int N, int K, int m_chunk, int n_splits, float* slab, float* bslab,
                       int64_t sstride, hipStream_t st);

void mg_launch_fused3(const uint16_t* dZ2, int lddz, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh, const uint16_t* A, int lda,
                      const int32_t* rows, int64_t M, int N, int K, int m_chunk, int n_splits, float* slab, float* bslab, int64_t sstride,
                      float* slab2, int64_t sstride2, hipStream_t st);

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

static int fused_launch(const char* name, const uint16_t* dZ2, int lddz, int N2, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh,
                        const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K, void* workspace, size_t workspace_bytes,
                        int* S_out, int64_t* sstride_out, hipStream_t st) {
    MG_CHECK_ARG(dZ2 && W2T && H1 && A && M > 0, "%s: null argument or empty batch", name);
    MG_CHECK_ARG(N2 == F_N2 && lddz >= F_N2 && ldwt >= F_N2 && lddz % 8 == 0 && ldwt % 8 == 0,
                 "%s: the second layer must have %d outputs (N2=%d lddz=%d ldwt=%d)", name, F_N2, N2, lddz, ldwt);
    MG_CHECK_ARG(N % F_BNT == 0 && ldh >= N && ldh % 8 == 0, "%s: hidden width %d must be a multiple of %d (ldh=%d)", name, N, F_BNT, ldh);
    MG_CHECK_ARG(lda == F_BKT && K > 512 && K <= F_BKT - 32, "%s: needs 512 < K <= 608 with lda = 640 (K=%d lda=%d)", name, K, lda);
    MG_CHECK_ARG((((uintptr_t)dZ2 | (uintptr_t)W2T | (uintptr_t)H1 | (uintptr_t)A) % 16) == 0, "%s: buffers must be 16-byte aligned", name);
    if (!workspace || workspace_bytes < mg_linear_bwd_fused_workspace_bytes(M, N, K)) {
        mg_set_error("%s: workspace of %zu bytes needed, got %zu", name, mg_linear_bwd_fused_workspace_bytes(M, N, K), workspace_bytes);
        return MG_EWORKSPACE;
    }
    int S, chunk;
    fused_plan(M, N, &S, &chunk);
    // split s of the workspace: [N*K weight partials | N bias partials]
    const int64_t nk = (int64_t)N * K, sstride = nk + N;
    float* slab = (float*)workspace;
    float* bslab = slab + nk;
    // MG_TUNE_FORM (same results every way): 0 = 64-frame steps, tiles three steps ahead (bwd_fused64_bf16.hip); 12 = the same with
    // tiles two steps ahead and the larger ring; 13 = the 32-frame-step kernel it replaced (its bit-exact reference in the tests);
    // 7 = the single-buffered kernel (also what inputs without a row map take)
    const int form = g_mg_tuning[MG_TUNE_FORM];
#ifdef MG_EXPERIMENTS
    // lab builds only: 15 = the woven single-stream experiment, 100 + mask = timing probes of the 64-frame kernel (results garbage),
    // 8 / 9 = the one-wave-per-SIMD experiment and its 128 x 512 probe
    if (rows && (form == 15 || form >= 100)) {
        mg_launch_fused64(form == 15 ? 1 : form, dZ2, lddz, W2T, ldwt, H1, ldh, A, lda, rows, M, N, K, chunk, S, slab, bslab, sstride, st);
    } else if (rows && form == 9) {
        hipLaunchKernelGGL(wgrad_fused_solo_kernel<4>, dim3((unsigned)((N / F_BNT) * mg_align_up((size_t)S, 8))), dim3(256), 0, st, dZ2, lddz, W2T,
                           ldwt, H1, ldh, A, lda, rows, M, N, K, chunk, S, slab, bslab, sstride);
    } else if (rows && form == 8) {
        hipLaunchKernelGGL(wgrad_fused_solo_kernel<5>, dim3((unsigned)((N / F_BNT) * mg_align_up((size_t)S, 8))), dim3(256), 0, st, dZ2, lddz, W2T,
                           ldwt, H1, ldh, A, lda, rows, M, N, K, chunk, S, slab, bslab, sstride);
    } else
#endif
    if (rows && (form == 0 || form == 12))
        mg_launch_fused64(form == 12 ? 2 : 3, dZ2, lddz, W2T, ldwt, H1, ldh, A, lda, rows, M, N, K, chunk, S, slab, bslab, sstride, st);
    else if (rows && form != 7)
        hipLaunchKernelGGL(wgrad_fused_pipe_kernel, dim3((unsigned)((N / F_BNT) * mg_align_up((size_t)S, 8))), dim3(512), 0, st, dZ2, lddz, W2T,
                           ldwt, H1, ldh, A, lda, rows, M, N, K, chunk, S, slab, bslab, sstride);
    else
        hipLaunchKernelGGL(wgrad_fused_kernel, dim3((unsigned)((N / F_BNT) * S)), dim3(512), 0, st, dZ2, lddz, W2T, ldwt, H1, ldh, A, lda, rows, M,
                           N, K, chunk, slab, bslab, sstride);
    MG_CHECK_LAUNCH(name);
    *S_out = S;
    *sstride_out = sstride;
    return MG_OK;
}

size_t mg_linear_bwd_fused2_workspace_bytes(int64_t M, int N, int K) {
    if (M <= 0 || N <= 0 || N % F_BNT != 0) return 256;
    int S, chunk;
    fused_plan(M, N, &S, &chunk);
    return mg_align_up((size_t)S * (((size_t)N * K + N) + ((size_t)F_N2 * N + F_N2)) * sizeof(float), 256);
}

int mg_linear_bwd_fused2_slabs_bf16(const uint16_t* dZ2, int lddz, int N2, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh,
                                    const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K, void* workspace,
                                    size_t workspace_bytes, int* n_slabs, int64_t* stride1, int64_t* offset2, int64_t* stride2, void* stream) {
    const char* name = "mg_linear_bwd_fused2_slabs_bf16";
    MG_CHECK_ARG(n_slabs && stride1 && offset2 && stride2, "%s: null output argument", name);
    MG_CHECK_ARG(dZ2 && W2T && H1 && A && rows && M > 0, "%s: null argument (the row map is required) or empty batch", name);
    MG_CHECK_ARG(N2 == F_N2 && lddz >= F_N2 && ldwt >= F_N2 && lddz % 8 == 0 && ldwt % 8 == 0,
                 "%s: the second layer must have %d outputs (N2=%d lddz=%d ldwt=%d)", name, F_N2, N2, lddz, ldwt);
    MG_CHECK_ARG(N % F_BNT == 0 && ldh >= N && ldh % 8 == 0, "%s: hidden width %d must be a multiple of %d (ldh=%d)", name, N, F_BNT, ldh);
    MG_CHECK_ARG(lda == F_BKT && K > 512 && K <= F_BKT - 32, "%s: needs 512 < K <= 608 with lda = 640 (K=%d lda=%d)", name, K, lda);
    MG_CHECK_ARG((((uintptr_t)dZ2 | (uintptr_t)W2T | (uintptr_t)H1 | (uintptr_t)A) % 16) == 0, "%s: buffers must be 16-byte aligned", name);
    if (!workspace || workspace_bytes < mg_linear_bwd_fused2_workspace_bytes(M, N, K)) {
        mg_set_error("%s: workspace of %zu bytes needed, got %zu", name, mg_linear_bwd_fused2_workspace_bytes(M, N, K), workspace_bytes);
        return MG_EWORKSPACE;
    }
    int S, chunk;
    fused_plan(M, N, &S, &chunk);
    // [S x (N*K weight partials | N bias partials)] then [S x (128*N second-layer weight partials | 128 bias partials)]
    const int64_t nk = (int64_t)N * K, s1 = nk + N, s2 = (int64_t)F_N2 * N + F_N2;
    float* slab = (float*)workspace;
    float* slab2 = slab + (int64_t)S * s1;
    mg_launch_fused3(dZ2, lddz, W2T, ldwt, H1, ldh, A, lda, rows, M, N, K, chunk, S, slab, slab + nk, s1, slab2, s2, (hipStream_t)stream);
    MG_CHECK_LAUNCH(name);
    *n_slabs = S;
    *stride1 = s1;
    *offset2 = (int64_t)S * s1;
    *stride2 = s2;
    return MG_OK;
}

int mg_linear_bwd_fused_bf16(const uint16_t* dZ2, int lddz, int N2, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh,
                             const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K, float* dW, float* db,
                             int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(dW && db, "mg_linear_bwd_fused_bf16: null gradient buffer");
    hipStream_t st = (hipStream_t)stream;
    int S = 0;
    int64_t sstride = 0;
    const int rc = fused_launch("mg_linear_bwd_fused_bf16", dZ2, lddz, N2, W2T, ldwt, H1, ldh, A, lda, rows, M, N, K, workspace, workspace_bytes,
                                &S, &sstride, st);
    if (rc != MG_OK) return rc;
    // when db sits right behind dW (one flat gradient buffer) a single reduce launch finishes both
    const int64_t nk = (int64_t)N * K;
    float* slab = (float*)workspace;
    float* bslab = slab + nk;
    if (db == dW + nk) {
        mg_launch_slab_reduce(slab, sstride, sstride, S, dW, accumulate, st);
    } else {
        mg_launch_slab_reduce(slab, nk, sstride, S, dW, accumulate, st);
        mg_launch_slab_reduce(bslab, N, sstride, S, db, accumulate, st);
    }
    MG_CHECK_LAUNCH("mg_linear_bwd_fused_bf16/reduce");
    return MG_OK;
}

int mg_linear_bwd_fused_slabs_bf16(const uint16_t* dZ2, int lddz, int N2, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh,
                                   const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K, void* workspace,
                                   size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream) {
    MG_CHECK_ARG(n_slabs && stride, "mg_linear_bwd_fused_slabs_bf16: null output argument");
    return fused_launch("mg_linear_bwd_fused_slabs_bf16", dZ2, lddz, N2, W2T, ldwt, H1, ldh, A, lda, rows, M, N, K, workspace, workspace_bytes,
                        n_slabs, stride, (hipStream_t)stream);
}

}  // extern "C"

#ifdef MG_STAMPS
extern "C" int mg_diag_read_stamps_fp(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_fp), bytes < sizeof(g_stamps_fp) ? bytes : sizeof(g_stamps_fp), 0, hipMemcpyDeviceToHost);
}
extern "C" int mg_diag_read_stamps_fz(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_fz), bytes < sizeof(g_stamps_fz) ? bytes : sizeof(g_stamps_fz), 0, hipMemcpyDeviceToHost);
}
#endif
