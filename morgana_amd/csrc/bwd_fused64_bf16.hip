// Fused backward of a  Linear(K -> N1) + Sigmoid -> Linear(N1 -> 128)  pair, bf16 mode, for gathered inputs (rows != NULL), in
// 64-FRAME steps:
//
//     dZ1 = (dZ2 W2) * H1 (1 - H1)          (never written to HBM)
//     dW1 = dZ1^T gather(X, rows),   db1 = column sums of dZ1
//
// Reference: autograd of the README stack (README.rst:65-73 via morgana/utils.py:401-418): mm, sigmoid_backward, mm, with the gather
// of morgana/utils.py:175-228 (upsample_to_repetitions) fused into the operand loads.  Same math, tile (a workgroup owns 128 hidden
// units x all 640 input columns for a slice of the frames), run staging, slabs and ordered reduce as wgrad_fused_pipe_kernel
// (bwd_fused_bf16.hip), which remains the bit-exact reference of this kernel in the tests.  What changed, and why (in-kernel stamps
// of that kernel, profiles/r2_stamps_fused.txt: of ~4,050 cycles per 32-frame step only 768 are matrix cycles per wave - 756 at the
// barrier, 729 issuing 2.6 LDS-DMA pieces with a run-table lookup in front of each, 946 in P1's dependent chain):
//   * a step is 64 frames: the barrier, the DMA issue block and P1's chain latency are paid once per TWICE the matrix work (P2 is
//     4 k-steps of 10 MFMAs per wave, P1 two rounds of 8 small MFMAs with both rounds' chains interleaved);
//   * no run tables in LDS: every wave derives the run structure of a step from the step's 64 row indices itself (they arrive by
//     LDS-DMA one step ahead; lane = frame: one ballot of "differs from the previous frame" IS the table - a frame's run ordinal is
//     a population count under that mask, in SGPRs), so the per-frame slot lookups of P2 are register arithmetic, the run-row
//     lookups in front of the DMA pieces one private round trip per step, and the 24 KB of tables pay for the larger tiles; runs
//     are numbered per step (the run that crosses a step boundary is staged again);
//   * the ring of staged source rows is allocated by whole groups of 4 runs per step; a step with more runs than the ring holds is
//     multiplied in passes (frames outside the pass read a zero row), a step whose successor does not fit beside it fetches on demand.
// Deterministic: split-M slabs + ordered reduce (shared with the other weight-gradient kernels).
#include "common.h"
#include "slab_reduce.h"

#include <type_traits>

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfv4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes land at LDS byte address (lds_addr + 16 l); lds_addr = the wave-uniform LDS address
// (an integer: casting a generic pointer back to the LDS address space inside the step loop made hipcc emit a null check that does
// not assemble).  M0 saved / restored inside the statement; completion counted by vmcnt.
__device__ __forceinline__ void wglds16(const void* src, unsigned lds_addr) {
    const unsigned lds_uni = __builtin_amdgcn_readfirstlane(lds_addr);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_uni)
                 : "memory");
}

// The same with a wave-uniform base and a 32-bit per-lane byte offset (no 64-bit address arithmetic in vector registers).
__device__ __forceinline__ void wglds16_off(const void* base_uniform, unsigned byte_off, unsigned lds_addr) {
    const unsigned lds_uni = __builtin_amdgcn_readfirstlane(lds_addr);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(base_uniform), "s"(lds_uni)
                 : "memory");
}

MG_STAMP_DECL(g_stamps_f64);
MG_STAMP_DECL(g_stamps_f3);

#define W_BNT 128                                 // hidden units per workgroup
#define W_BKT 640                                 // padded input width
#define W_N2 128
#define W_F 64                                    // frames per step
#define W_GROUP 5120                              // a group = 4 staged source rows of 1280 B = 5 LDS-DMA pieces

template <int NBT>
struct W64Map {                                   // LDS map (bytes); NBT = buffers of the dZ2 / H1 tiles (fetched NBT steps ahead of P1)
    static constexpr int NG = NBT == 3 ? 5 : 8;   // ring groups
    static constexpr int X = 0;                   // ring of 4 NG source rows x 1280 B (tr-swizzled by row & 3)
    static constexpr int ZROW = X + NG * W_GROUP; // 1280 B of zeros: what frames outside a pass read
    static constexpr int ONES = ZROW + 1280;      // 1280 B of bf16 1.0: the "operand row" of the bias sums
    static constexpr int DZ = ONES + 1280;        // NBT x (4 k-tiles x [64 rows x 64 B]), chunk ^ 2 ((row >> 3) & 1)
    static constexpr int H1 = DZ + NBT * 16384;   // NBT x [64 m][256 B], chunk c of row m at position c ^ (m & 15)
    static constexpr int YS = H1 + NBT * 16384;   // 2 x [64 m][256 B] tr-swizzled: the dZ1 tile
    static constexpr int RUNROW = YS + 2 * 16384; // 8 waves x int32[64]: source row of each run of the step being fetched
    static constexpr int ROWS = RUNROW + 8 * 256; // 3 x int32[64]: the row indices of a step (LDS-DMA by wave 0, one step ahead)
    static constexpr int LDS = ROWS + 3 * 256;
};

// PROBE (lab builds only, -DMG_EXPERIMENTS; results garbage): timing experiments on parts of the step - 1 = every P2 operand read
// from the zero row, 2 = no operand reads in P2 at all, 4 = no P1, 8 = no tile / staged-row DMA, 16 = P2 reads two pairs ahead instead of four.
template <int NBT, int PROBE = 0>
__global__ __launch_bounds__(512) void wgrad_fused64_kernel(const uint16_t* __restrict__ dZ2, int lddz, const uint16_t* __restrict__ W2T,
                                                            int ldwt, const uint16_t* __restrict__ H1, int ldh,
                                                            const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                            int64_t M, int N, int K, int m_chunk, int n_splits,
                                                            float* __restrict__ slab, float* __restrict__ bslab, int64_t sstride) {
    using L = W64Map<NBT>;
    static_assert(L::LDS <= 160 * 1024, "one workgroup per CU");
    constexpr int NG = L::NG;
    constexpr int TKT = 5;
    constexpr int PY = 256, PX = W_BKT * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::LDS];
    const unsigned smem_lds = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MG_STAMPS
    unsigned long long ts0, ts1 = 0, ts2, ts3, tr0, tr1, ta, tb, sum_wait = 0, sum_fetch = 0, sum_p1 = 0, sum_p2 = 0, sum_scan = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    const int wn0 = (wave >> 2) * 64;
    const int wk0 = (wave & 3) * (TKT * 32);
    const int tiles_n = N / W_BNT;
    // Blocks b, b + 8, b + 16, ... share an XCD and its L2: the tiles_n workgroups of one frame range go there (dZ2 leaves HBM once).
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int n0 = (jq % tiles_n) * W_BNT;
    const int s = (jq / tiles_n) * 8 + xcd;
    if (s >= n_splits) return;
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t m_hi = min(M, m_lo + (int64_t)m_chunk);
    const int n_rows = m_hi > m_lo ? (int)(m_hi - m_lo) : 0;
    const int n_steps = (n_rows + W_F - 1) / W_F;

    if (tid < 80) *reinterpret_cast<uint4*>(smem + L::ZROW + tid * 16) = uint4{0u, 0u, 0u, 0u};
    if (tid >= 128 && tid < 208) *reinterpret_cast<uint4*>(smem + L::ONES + (tid - 128) * 16) = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    const unsigned long long a_ptr = (unsigned long long)A;
    const unsigned lda_bytes = (unsigned)lda * 2u;

    // ---- run structure of a step: every wave derives it from the step's 64 row indices (lane = frame) ---------------------------------
    // The indices of step u arrive by LDS-DMA (wave 0, issued one step ahead, in front of that step's other pieces: the step barrier's
    // counted wait covers it) in slot u % 3 (it stays readable while step u is the current one); frames past the range count as pad frames (row -1: the zero row).
    // (Per-lane constants are re-derived from an opaque copy of the lane id inside every phase: left to itself hipcc hoists them out
    // of the step loop, where they stay live across P1 and P2 - 160 accumulator registers leave no room for that - and spills.)
    auto opaque_lane = [&]() -> int {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return ln;
    };
    auto issue_rows = [&](int u) {                        // wave 0 only
        const int lane = opaque_lane();
        const int f = u * W_F + lane;
        int64_t m = m_lo + min(f, max(n_rows - 1, 0));
        if (m > M - 1) m = M - 1;
        const int32_t* src = rows + m;
        const unsigned lds_uni = __builtin_amdgcn_readfirstlane(smem_lds + L::ROWS + (u % 3) * 256);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(lds_uni)
                     : "memory");
    };
    auto read_rows = [&](int u) -> int {
        const int lane = opaque_lane();
        const int v = *reinterpret_cast<const int*>(smem + L::ROWS + (u % 3) * 256 + lane * 4);
        return (u * W_F + lane < n_rows) ? v : -1;
    };
    // masks (bit = frame) of "a run starts at this frame" and of "a pad frame" (row < 0: its operand row is the zero row); every wave
    // computes the same values
    struct StepMasks {
        unsigned lo, hi, plo, phi;
    };
    auto scan_mask = [&](int rr) -> StepMasks {
        const int lane = opaque_lane();
        const int prev = __shfl_up(rr, 1, 64);
        const bool flag = (lane == 0) || (rr != prev);
        const unsigned long long mk = __ballot(flag), pk = __ballot(rr < 0);
        StepMasks m;
        m.lo = __builtin_amdgcn_readfirstlane((unsigned)mk);
        m.hi = __builtin_amdgcn_readfirstlane((unsigned)(mk >> 32));
        m.plo = __builtin_amdgcn_readfirstlane((unsigned)pk);
        m.phi = __builtin_amdgcn_readfirstlane((unsigned)(pk >> 32));
        return m;
    };
    auto runs_of = [&](const StepMasks& m) -> int { return __builtin_popcount(m.lo) + __builtin_popcount(m.hi); };
    auto groups_of = [&](const StepMasks& m) -> int { return (runs_of(m) + 3) >> 2; };
    int* my_runrow = reinterpret_cast<int*>(smem + L::RUNROW + wave * 256);

    // DMA of staged rows: local groups [g_first, g_first + g_cnt) of the step whose rows are `rr` (lane = frame) with masks mk, into
    // ring groups (rp + 0 .. g_cnt - 1) % NG.  Issued by waves 0-3 only (they fetch at the head of a step: a whole step of slack; the
    // upper waves fetch behind their matrix work, which would leave these rows none); piece pi of the span goes to wave (pi + rot) & 3.
    // Runs of pad frames and the unused runs of the last group fetch table row 0: valid memory that no frame reads (P2 sends pad
    // frames to the zero row in LDS).
    auto issue_x = [&](int rr, const StepMasks& mk, int rp, int g_first, int g_cnt, int rot) {
        const int n_pieces = 5 * g_cnt;
        int pi = (wave - rot) & 3;
        if (wave >= 4 || pi >= n_pieces) return;          // wave-uniform: nothing for this wave
        if ((PROBE & 8) && rot > 0) return;
        const int lane = opaque_lane();
        const bool flag = ((lane < 32 ? (mk.lo >> lane) : (mk.hi >> (lane - 32))) & 1u) != 0u;
        const int ord = (int)__builtin_amdgcn_mbcnt_hi(mk.hi, __builtin_amdgcn_mbcnt_lo(mk.lo, 0u)) + (flag ? 1 : 0) - 1;
        if (flag) my_runrow[ord] = rr;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int n_runs = runs_of(mk);
        for (; pi < n_pieces; pi += 4) {
            const int gl = pi / 5, i = pi - 5 * gl;       // group of the span, piece of the group (wave-uniform)
            const int byte = i * 1024 + lane * 16;
            const int xr = byte >= 3 * PX ? 3 : byte >= 2 * PX ? 2 : byte >= PX ? 1 : 0;
            const int cpos = (byte - xr * PX) >> 4;
            const int xo = (cpos ^ ((xr & 3) << 2)) * 16;
            const int run = 4 * (g_first + gl) + xr;
            int src = my_runrow[min(run, 63)];
            if (run >= n_runs || src < 0) src = 0;
            int rg = rp + gl;
            rg -= (rg >= NG) ? NG : 0;
            wglds16((const void*)(a_ptr + (unsigned long long)(unsigned)src * lda_bytes + (unsigned)xo), smem_lds + L::X + rg * W_GROUP + i * 1024);
        }
    };

    // The 16 + 16 pieces of a step's dZ2 / H1 tiles (dZ2 piece p: k-tile p >> 2, rows 16 (p & 3) ..; H1 piece p: rows 4 p .. 4 p + 3).
    // NBT == 3: every wave takes pieces `wave` and `wave + 8` of both (the upper waves behind their matrix work: two more steps until
    // P1 wants them); NBT == 2: waves 0-3 take pieces `wave` + 4 h at the head of the step (one step until P1 wants them).
    // Wave-uniform bases (the range's first row) + 32-bit byte offsets; frames past the range re-read its last row (P1 zeroes them).
    const uint16_t* dz_base = dZ2 + (size_t)m_lo * lddz;
    const uint16_t* h1_base = H1 + (size_t)m_lo * ldh + n0;
    const int last_row = max(n_rows - 1, 0);
    auto issue_small = [&](int step, int buf) {
        constexpr int PW = NBT == 3 ? 2 : 4, PS = NBT == 3 ? 8 : 4;
        if (NBT != 3 && wave >= 4) return;
        if ((PROBE & 8) && step > 2) return;
        const int lane = opaque_lane();
#pragma unroll
        for (int h = 0; h < PW; ++h) {
            const int p = wave + PS * h;
            const int dz_row = 16 * (p & 3) + (lane >> 2);
            const int dz_col = 32 * (p >> 2) + 8 * ((lane & 3) ^ (((dz_row >> 3) & 1) << 1));
            const int mz = min(step * W_F + dz_row, last_row);
            wglds16_off(dz_base, (unsigned)(mz * lddz + dz_col) * 2u, smem_lds + L::DZ + buf * 16384 + p * 1024);
        }
#pragma unroll
        for (int h = 0; h < PW; ++h) {
            const int p = wave + PS * h;
            const int h1_row = 4 * p + (lane >> 4);
            const int h1_col = 8 * ((lane & 15) ^ (h1_row & 15));
            const int mh = min(step * W_F + h1_row, last_row);
            wglds16_off(h1_base, (unsigned)(mh * ldh + h1_col) * 2u, smem_lds + L::H1 + buf * 16384 + p * 1024);
        }
    };

    f32x16 acc[2][TKT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TKT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool bias_free = bslab != nullptr && (wave & 3) == 3;        // K <= 608: last 32-column tile of k-wave 3 is padding

    // P1 geometry: wave w owns hidden units 16 w .. 16 w + 15; 16x16x32 MFMA, A = W2T rows (unit l15, k chunk lq), B = dZ2 rows
    // (frame l15 of block t, k chunk lq); D: this lane holds units 16 w + 4 lq .. + 3 of frame 16 t + l15.
    const int l15 = lane & 15, lq = lane >> 4;
    bfv8 w2[4];                                           // loop invariant: W2^T[n0 + 16 w + l15][32 ks + 8 lq ..]
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        w2[ks] = *reinterpret_cast<const bfv8*>(W2T + (size_t)(n0 + 16 * wave + l15) * ldwt + 32 * ks + 8 * lq);
    // Land them here and re-define the registers behind the wait (hipcc cannot see across the loop's back edge that these loads are
    // long complete and would drain the LDS-DMA prefetches in front of P1's MFMAs every step).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(w2[ks]));
    // P1 of step u: dZ2 / H1 tiles of buffer tb, dZ1 tile to buffer yb2 (of 2); two rounds of two 16-frame blocks.  Frames past the
    // range (the last step's tail) give dZ1 = 0: their tile rows are copies of the range's last row.
    auto p1 = [&](int u, int tb, int yb2) {
        if (PROBE & 4) return;
        const int lane = opaque_lane();
        const int l15 = lane & 15, lq = lane >> 4;
        const int c16 = 2 * wave + (lq >> 1);                                             // 16-byte chunk of the 256-byte row
        const int b_off0 = L::DZ + l15 * 64 + ((lq ^ (((l15 >> 3) & 1) << 1)) << 4);       // + 1024 t + 4096 per k-tile
        const int h_off0 = L::H1 + l15 * 256 + ((c16 ^ l15) << 4) + 8 * (lq & 1);          // + 4096 t
        const int y_off0 = L::YS + l15 * 256 + ((c16 ^ ((l15 & 3) << 2)) << 4) + 8 * (lq & 1);
        const int bo = tb * 16384, yo = yb2 * 16384;
        const int f_left = n_rows - u * W_F - l15;        // frame 16 t + l15 of the step is inside the range iff 16 t < f_left
        // The four 16-frame blocks are four independent accumulation chains, streamed k-tile by k-tile: an accumulator comes round
        // every fourth MFMA (no dependent-issue stall), the operand reads run four MFMAs ahead in a ring of four fragments, and the
        // tiles' latency is paid once per step (as two rounds of two blocks - read 8, multiply 8, tail - it was paid twice, with the
        // 4-deep chains' latency and the VALU tail in line behind it: 1,700-2,000 cycles per step for 256 matrix cycles).
        bfv8 b[4];
        bfv4 hv[4];
        f32x4 d[4];
        auto rd = [&](int i) -> bfv8 {                     // operand of MFMA i = 4 ks + t
            return *reinterpret_cast<const bfv8*>(smem + bo + b_off0 + (i & 3) * 1024 + (i >> 2) * 4096);
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = rd(i);
#pragma unroll
        for (int t = 0; t < 4; ++t) hv[t] = *reinterpret_cast<const bfv4*>(smem + bo + h_off0 + t * 4096);
#pragma unroll
        for (int t = 0; t < 4; ++t) d[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            d[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[i >> 2], b[i & 3], d[i & 3], 0, 0, 0);
            if (i + 4 < 16) b[i & 3] = rd(i + 4);
        }
        const bool tail_step = f_left + l15 < W_F;        // wave-uniform: the range ends inside this step
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (float)hv[t][e];
                v[e] = d[t][e] * h * (1.f - h);
            }
            if (tail_step && !(16 * t < f_left)) v[0] = v[1] = v[2] = v[3] = 0.f;
            const u32x2 pk = u32x2{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                   __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
            *reinterpret_cast<u32x2*>(smem + yo + y_off0 + t * 4096) = pk;
        }
    };

    // P2 of one step and pass: the dZ1 tile of buffer yb2 against the ring rows of the frames' runs, as two halves of 32 frames (two
    // 16-frame k-steps each).  A frame's run ordinal is the population count of the run-start mask up to and including its bit, - 1;
    // local group gl = (ordinal >> 2) - g_first sits in ring group (rp + gl) % NG, runs outside [g_first, g_first + NG) read the zero
    // row, and so do pad frames (row index < 0).  with_bias: the pass that carries the bias sums.
    auto p2 = [&](int yb2, const StepMasks& mk4, int rp, int g_first, bool with_bias) {
        const int lane = opaque_lane();
        const int li = lane & 15, g4 = lane >> 4;
        const int q = li >> 2, p4 = li & 3;
        const int cgrp = 16 * (g4 & 1) + 4 * p4;
        const int rbase = 8 * (g4 >> 1) + q;                  // this lane's frame inside a 16-frame k-step (and + 4)
        const int sw = q << 2;
        int yoff[2], xk[TKT];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int col = wn0 + i * 32 + cgrp;
            yoff[i] = L::YS + rbase * PY + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
        }
#pragma unroll
        for (int j = 0; j < TKT; ++j) {
            const int col = wk0 + j * 32 + cgrp;
            xk[j] = ((col >> 3) << 4) | ((col & 7) << 1);                  // byte offset inside a ring row before the swizzle
        }
        const int pc_lo = __builtin_popcount(mk4.lo);
        const int bias_row = with_bias ? L::ONES : L::ZROW;   // bias_free waves: the last column block multiplies dZ1 by ones
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int yb = yb2 * 16384 + half * 32 * PY;
            const unsigned mk = half ? mk4.hi : mk4.lo, pad = half ? mk4.phi : mk4.plo;
            const int pc0 = half ? pc_lo - 1 : -1;
            int xb[4], xs[4];                             // ring byte offset and swizzle of this lane's 4 frames of the half: 2 ks + hi
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const int f = 16 * (fi >> 1) + rbase + 4 * (fi & 1);
                const int o = pc0 + __builtin_popcount(mk & ((2u << f) - 1u));
                const int gl = (o >> 2) - g_first;
                int rg = rp + gl;
                rg -= (rg >= NG) ? NG : 0;
                const bool in_pass = (unsigned)gl < (unsigned)NG && ((pad >> f) & 1u) == 0u;
                xb[fi] = in_pass ? L::X + (rg * 4 + (o & 3)) * PX : L::ZROW;
                xs[fi] = in_pass ? (o & 3) << 6 : 0;
            }
            bfv8 probe_frag;                               // PROBE & 2: stands for every operand (its value never matters)
            if (PROBE & 2) asm volatile("" : "=v"(probe_frag));
            auto rd_a = [&](int ks, int i) -> bfv8 {
                if (PROBE & 2) return probe_frag;
                const unsigned char* ad = smem + yb + yoff[i] + ks * 16 * PY;
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PY));
                return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            auto rd_b = [&](int ks, int j) -> bfv8 {
                if (PROBE & 2) return probe_frag;
                const bool bias_blk = (bias_free && j == TKT - 1) || (PROBE & 1);
                const unsigned char* alo = smem + (bias_blk ? bias_row : xb[2 * ks]) + (xk[j] ^ xs[2 * ks]);
                const unsigned char* ahi = smem + (bias_blk ? bias_row : xb[2 * ks + 1]) + (xk[j] ^ xs[2 * ks + 1]);
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(alo));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ahi));
                return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            // Fragment reads run four MFMA pairs ahead of their use (a rolling window instead of "all 14 reads, wait, 10 MFMAs" per
            // k-step; two pairs ahead measured 13 us slower at C2: the tr-reads take longer than two pairs' 128 matrix cycles to return
            // while the other waves and the DMA keep the LDS busy).
            bfv8 a[2][2], b[2 * TKT];
            a[0][0] = rd_a(0, 0);
            a[0][1] = rd_a(0, 1);
            constexpr int W = (PROBE & 16) ? 2 : 4;        // pairs the operand reads run ahead
            constexpr int TA = (PROBE & 16) ? 2 : 1;       // the pair at which the next k-step's dZ1 fragments are read
            b[0] = rd_b(0, 0);
            b[1] = rd_b(0, 1);
            if (W == 4) {
                b[2] = rd_b(0, 2);
                b[3] = rd_b(0, 3);
            }
            auto mm = [&](auto tc) {                       // t = ks * TKT + j, a compile-time constant
                constexpr int t = decltype(tc)::value;
                constexpr int ks = t / TKT, j = t % TKT;
                if constexpr (t + W < 2 * TKT) b[t + W] = rd_b((t + W) / TKT, (t + W) % TKT);
                if constexpr (t == TA) {
                    a[1][0] = rd_a(1, 0);
                    a[1][1] = rd_a(1, 1);
                }
                acc[0][j] = mg_mfma_32x32x16(a[ks][0], b[t], acc[0][j]);
                acc[1][j] = mg_mfma_32x32x16(a[ks][1], b[t], acc[1][j]);
                constexpr int n_reads = (PROBE & 2) ? 0 : (t + W < 2 * TKT ? 2 : 0) + (t == TA ? 4 : 0);
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
        }
    };

    // ---- per-step state (wave-uniform unless noted) ----------------------------------------------------------------------------------
    // cur: the step P2 works on; nxt: the step whose staged rows are fetched meanwhile.  Groups = 4 runs; rp = ring group of a step's
    // local group 0; have = are the first min(groups, NG) groups of the step in the ring (fetched ahead)?
    StepMasks mk_cur = {0u, 0u, 0u, 0u};
    int rp_cur = 0;
    bool have_cur = true;

    if (n_steps > 0) {
        // prologue: row indices of steps 0 and 1, staged rows and tiles of step 0 waited for, P1 of step 0; then the tiles of steps
        // 1 .. NBT - 1 in flight (past the end: copies of the last row, never used)
        if (wave == 0) {
            issue_rows(0);
            issue_rows(1);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        mk_cur = scan_mask(read_rows(0));
        issue_x(read_rows(0), mk_cur, 0, 0, min(groups_of(mk_cur), NG), 0);
        issue_small(0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        p1(0, 0, 0);
#pragma unroll
        for (int u = 1; u < NBT; ++u) issue_small(u, u);
    }
    int tb_next = 1 % NBT;                                // tile buffer of step + 1
    for (int step = 0; step < n_steps; ++step) {
        // Landed: staged rows of this step (when fetched ahead), tiles of step + 1, row indices of step + 1; dZ1(step) complete; the
        // buffers of the last step are free.  In flight with NBT == 3: the tile pieces of step + 2 (this wave's newest four).
        MG_STAMP(ta);
        if (NBT == 3)
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_wait, tb, ta);
#ifdef MG_STAMPS
        if (step == 0) ts1 = tb;
#endif
        const bool more = step + 1 < n_steps;
        const int g_cur = groups_of(mk_cur);
        if (!have_cur) {                                  // this step's rows did not fit beside its predecessor's: fetch them now
            rp_cur = 0;
            issue_x(read_rows(step), mk_cur, 0, 0, min(g_cur, NG), step);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // the next step: its run structure, whether its first groups fit in the ring beside this step's
        const StepMasks mk_nxt = scan_mask(read_rows(step + 1));      // past the last step: all pad frames, unused
        const int g_nxt = groups_of(mk_nxt);
        const bool pref = more && g_cur <= NG && g_cur + min(g_nxt, NG) <= NG;
        int rp_nxt = rp_cur + g_cur;
        rp_nxt -= (rp_nxt >= NG) ? NG : 0;
        const int tb_new = tb_next == 0 ? NBT - 1 : tb_next - 1;        // (step + NBT) % NBT == step % NBT
        // Fetches of this iteration: the row indices of step + 2 (wave 0), the staged rows of step + 1, the tiles of step + NBT.  The
        // lower half of the waves issues them now, the upper half after its matrix work: issued by all waves at once behind the
        // barrier they serialise in the texture-address path while every matrix pipe waits.
        auto fetch = [&]() {
            if (wave == 0) issue_rows(step + 2);
            if (pref) issue_x(read_rows(step + 1), mk_nxt, rp_nxt, 0, min(g_nxt, NG), step);
            issue_small(step + NBT, tb_new);
        };
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_scan, ta, tb);
        if (wave < 4) {
            fetch();
            MG_STAMP(tb);
            MG_STAMP_ADD(sum_fetch, tb, ta);
            if (more) p1(step + 1, tb_next, (step + 1) & 1);
            MG_STAMP(ta);
            MG_STAMP_ADD(sum_p1, ta, tb);
        }
        // P2; a step with more runs than the ring holds takes the remaining groups in passes through the whole ring (rare, and never
        // beside a prefetch: pref is false for such a step)
        for (int gf = 0;; gf += NG) {
            if (gf > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue_x(read_rows(step), mk_cur, 0, gf, min(g_cur - gf, NG), step);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            p2(step & 1, mk_cur, gf ? 0 : rp_cur, gf, gf == 0);
            if (gf + NG >= g_cur) break;
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_p2, tb, ta);
        if (wave >= 4) {
            if (more) p1(step + 1, tb_next, (step + 1) & 1);
            MG_STAMP(ta);
            MG_STAMP_ADD(sum_p1, ta, tb);
            fetch();
            MG_STAMP(tb);
            MG_STAMP_ADD(sum_fetch, tb, ta);
        }
        mk_cur = mk_nxt;
        rp_cur = pref ? rp_nxt : 0;
        have_cur = pref || !more;
        tb_next = tb_next + 1 == NBT ? 0 : tb_next + 1;
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
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 2, ts2);
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 3, ts3);
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 4, tr0);
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 5, tr1);
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 6, sum_wait);
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 7, sum_fetch);
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 8, sum_p1);
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 9, sum_p2);
    MG_STAMP_STORE(g_stamps_f64, sb, wave, lane, 10, sum_scan);
#endif
}

// -------------------------------------------------------------------------------------------------------------------------------------
// The same kernel WITH THE SECOND LAYER'S WEIGHT GRADIENT (P3):  dW2 = dZ2^T H1,  db2 = column sums of dZ2.
// Every workgroup already stages what that product needs - the full dZ2 tile and its own 128 columns of H1 - so the stand-alone
// launch that re-read H1 and dZ2 from HBM (wgrad_big_kernel<8> + its slab reduce: 72 + 10 us and 378 + 51 MB per step at C2, the
// launch HBM bound at 5.3 TB/s) goes: one pass over H1 less.  What it took (the 64-frame kernel above stood at 228 of 256 registers):
//   * P3's 128 x 128 block per workgroup is 32 more accumulator registers per lane; the loop-invariant W2^T fragments of P1 (16
//     registers) move to LDS (32 KB; P1 reads them beside its dZ2 fragments), which leaves two tile buffers (tiles fetched two steps
//     ahead) and a 5-group ring - the W64P3Map below;
//   * the dZ2 and H1 tiles take the dual-use image of cdna_hip_programming.md T10 (256-byte rows, chunk ^ (((row & 3) << 2) |
//     ((row >> 2) & 3))): P1 still reads rows (ds_read_b128 / 8-byte pieces), P3 reads the same tiles TRANSPOSED
//     (ds_read_b64_tr_b16: frames become the contraction index) - no second copy;
//   * P3 of step + 1 runs behind P1 of step + 1 on the same tiles: wave w owns dZ2 columns 32 (w >> 1) .. + 31 against H1 columns
//     64 (w & 1) .. + 63 of the workgroup's block (two 32 x 32 accumulators, 8 MFMAs and 24 transposed reads per step); the bias
//     gradient is summed from the dZ2^T fragments the even waves hold anyway (8 values per lane and k-step);
//   * frames past the range (the last step of a range that is not a multiple of 64) are zeroed in the dZ2 tile before P1 / P3 read it.
// Slabs: dW1 | db1 as above, and per frame range a second slab [128 x N | 128] of which the workgroup writes its 128 columns (the
// db2 part by the first column block only).  Same results as the two launches up to the summation order of dW2 (frames in ranges
// of m_chunk instead of the stand-alone plan's).
// PROBE (lab builds only): as above.
// -------------------------------------------------------------------------------------------------------------------------------------
struct W64P3Map {
    static constexpr int NG = 5;
    static constexpr int X = 0;                   // ring of 20 source rows x 1280 B (tr-swizzled by row & 3)
    static constexpr int ZROW = X + NG * W_GROUP;
    static constexpr int ONES = ZROW + 1280;
    static constexpr int DZ = ONES + 1280;        // 2 x [64 frames][256 B], dual-use image: chunk c of row m at c ^ sw(m)
    static constexpr int H1 = DZ + 2 * 16384;     // 2 x [64 frames][256 B], the same image
    static constexpr int YS = H1 + 2 * 16384;     // 2 x [64 m][256 B] tr-swizzled: the dZ1 tile
    static constexpr int W2T = YS + 2 * 16384;    // 4 k-tiles x [128 rows x 64 B], chunk ^ 2 ((row >> 3) & 1): this block's rows of W2^T
    static constexpr int RUNROW = W2T + 32768;    // 8 waves x int32[64]
    static constexpr int ROWS = RUNROW + 8 * 256; // 3 x int32[64]
    static constexpr int LDS = ROWS + 3 * 256;
};
__device__ __forceinline__ int w64_sw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }      // the dual-use image's chunk swizzle

template <int PROBE = 0>
__global__ __launch_bounds__(512) void wgrad_fused3_kernel(const uint16_t* __restrict__ dZ2, int lddz, const uint16_t* __restrict__ W2T,
                                                            int ldwt, const uint16_t* __restrict__ H1, int ldh,
                                                            const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                            int64_t M, int N, int K, int m_chunk, int n_splits,
                                                            float* __restrict__ slab, float* __restrict__ bslab, int64_t sstride,
                                                            float* __restrict__ slab2, int64_t sstride2) {
    constexpr int NBT = 2;
    using L = W64P3Map;
    static_assert(L::LDS <= 160 * 1024, "one workgroup per CU");
    constexpr int NG = L::NG;
    constexpr int TKT = 5;
    constexpr int PY = 256, PX = W_BKT * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::LDS];
    const unsigned smem_lds = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MG_STAMPS
    unsigned long long ts0, ts1 = 0, ts2, ts3, tr0, tr1, ta, tb, sum_wait = 0, sum_fetch = 0, sum_p1 = 0, sum_p2 = 0, sum_scan = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    const int wn0 = (wave >> 2) * 64;
    const int wk0 = (wave & 3) * (TKT * 32);
    const int tiles_n = N / W_BNT;
    // Blocks b, b + 8, b + 16, ... share an XCD and its L2: the tiles_n workgroups of one frame range go there (dZ2 leaves HBM once).
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int n0 = (jq % tiles_n) * W_BNT;
    const int s = (jq / tiles_n) * 8 + xcd;
    if (s >= n_splits) return;
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t m_hi = min(M, m_lo + (int64_t)m_chunk);
    const int n_rows = m_hi > m_lo ? (int)(m_hi - m_lo) : 0;
    const int n_steps = (n_rows + W_F - 1) / W_F;

    if (tid < 80) *reinterpret_cast<uint4*>(smem + L::ZROW + tid * 16) = uint4{0u, 0u, 0u, 0u};
    if (tid >= 128 && tid < 208) *reinterpret_cast<uint4*>(smem + L::ONES + (tid - 128) * 16) = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    const unsigned long long a_ptr = (unsigned long long)A;
    const unsigned lda_bytes = (unsigned)lda * 2u;

    // ---- one-time: this workgroup's 128 rows of W2^T, 4 k-tiles of [128 rows x 64 B] (32 pieces, 4 per wave) -------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = wave * 4 + i;
        const int kt = p >> 3;
        const int row = 16 * (p & 7) + (lane >> 2);
        const int c = (lane & 3) ^ (((row >> 3) & 1) << 1);
        wglds16(W2T + (size_t)(n0 + row) * ldwt + 32 * kt + 8 * c, smem_lds + L::W2T + p * 1024);
    }

    // ---- run structure of a step: every wave derives it from the step's 64 row indices (lane = frame) ---------------------------------
    // The indices of step u arrive by LDS-DMA (wave 0, issued one step ahead, in front of that step's other pieces: the step barrier's
    // counted wait covers it) in slot u % 3 (it stays readable while step u is the current one); frames past the range count as pad frames (row -1: the zero row).
    // (Per-lane constants are re-derived from an opaque copy of the lane id inside every phase: left to itself hipcc hoists them out
    // of the step loop, where they stay live across P1 and P2 - 160 accumulator registers leave no room for that - and spills.)
    auto opaque_lane = [&]() -> int {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return ln;
    };
    auto issue_rows = [&](int u) {                        // wave 0 only
        const int lane = opaque_lane();
        const int f = u * W_F + lane;
        int64_t m = m_lo + min(f, max(n_rows - 1, 0));
        if (m > M - 1) m = M - 1;
        const int32_t* src = rows + m;
        const unsigned lds_uni = __builtin_amdgcn_readfirstlane(smem_lds + L::ROWS + (u % 3) * 256);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(lds_uni)
                     : "memory");
    };
    auto read_rows = [&](int u) -> int {
        const int lane = opaque_lane();
        const int v = *reinterpret_cast<const int*>(smem + L::ROWS + (u % 3) * 256 + lane * 4);
        return (u * W_F + lane < n_rows) ? v : -1;
    };
    // masks (bit = frame) of "a run starts at this frame" and of "a pad frame" (row < 0: its operand row is the zero row); every wave
    // computes the same values
    struct StepMasks {
        unsigned lo, hi, plo, phi;
    };
    auto scan_mask = [&](int rr) -> StepMasks {
        const int lane = opaque_lane();
        const int prev = __shfl_up(rr, 1, 64);
        const bool flag = (lane == 0) || (rr != prev);
        const unsigned long long mk = __ballot(flag), pk = __ballot(rr < 0);
        StepMasks m;
        m.lo = __builtin_amdgcn_readfirstlane((unsigned)mk);
        m.hi = __builtin_amdgcn_readfirstlane((unsigned)(mk >> 32));
        m.plo = __builtin_amdgcn_readfirstlane((unsigned)pk);
        m.phi = __builtin_amdgcn_readfirstlane((unsigned)(pk >> 32));
        return m;
    };
    auto runs_of = [&](const StepMasks& m) -> int { return __builtin_popcount(m.lo) + __builtin_popcount(m.hi); };
    auto groups_of = [&](const StepMasks& m) -> int { return (runs_of(m) + 3) >> 2; };
    int* my_runrow = reinterpret_cast<int*>(smem + L::RUNROW + wave * 256);

    // DMA of staged rows: local groups [g_first, g_first + g_cnt) of the step whose rows are `rr` (lane = frame) with masks mk, into
    // ring groups (rp + 0 .. g_cnt - 1) % NG.  Issued by waves 0-3 only (they fetch at the head of a step: a whole step of slack; the
    // upper waves fetch behind their matrix work, which would leave these rows none); piece pi of the span goes to wave (pi + rot) & 3.
    // Runs of pad frames and the unused runs of the last group fetch table row 0: valid memory that no frame reads (P2 sends pad
    // frames to the zero row in LDS).
    auto issue_x = [&](int rr, const StepMasks& mk, int rp, int g_first, int g_cnt, int rot) {
        const int n_pieces = 5 * g_cnt;
        int pi = (wave - rot) & 3;
        if (wave >= 4 || pi >= n_pieces) return;          // wave-uniform: nothing for this wave
        if ((PROBE & 8) && rot > 0) return;
        const int lane = opaque_lane();
        const bool flag = ((lane < 32 ? (mk.lo >> lane) : (mk.hi >> (lane - 32))) & 1u) != 0u;
        const int ord = (int)__builtin_amdgcn_mbcnt_hi(mk.hi, __builtin_amdgcn_mbcnt_lo(mk.lo, 0u)) + (flag ? 1 : 0) - 1;
        if (flag) my_runrow[ord] = rr;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int n_runs = runs_of(mk);
        for (; pi < n_pieces; pi += 4) {
            const int gl = pi / 5, i = pi - 5 * gl;       // group of the span, piece of the group (wave-uniform)
            const int byte = i * 1024 + lane * 16;
            const int xr = byte >= 3 * PX ? 3 : byte >= 2 * PX ? 2 : byte >= PX ? 1 : 0;
            const int cpos = (byte - xr * PX) >> 4;
            const int xo = (cpos ^ ((xr & 3) << 2)) * 16;
            const int run = 4 * (g_first + gl) + xr;
            int src = my_runrow[min(run, 63)];
            if (run >= n_runs || src < 0) src = 0;
            int rg = rp + gl;
            rg -= (rg >= NG) ? NG : 0;
            wglds16((const void*)(a_ptr + (unsigned long long)(unsigned)src * lda_bytes + (unsigned)xo), smem_lds + L::X + rg * W_GROUP + i * 1024);
        }
    };

    // The 16 + 16 pieces of a step's dZ2 / H1 tiles (piece p = frames 4 p .. 4 p + 3 of the dual-use image: lane l fills position l & 15
    // of frame 4 p + (l >> 4) with source chunk (l & 15) ^ sw(frame)); waves 0-3 take pieces `wave` + 4 h of both at the head of the step.
    // Wave-uniform bases (the range's first row) + 32-bit byte offsets; frames past the range re-read its last row (zeroed / masked later).
    const uint16_t* dz_base = dZ2 + (size_t)m_lo * lddz;
    const uint16_t* h1_base = H1 + (size_t)m_lo * ldh + n0;
    const int last_row = max(n_rows - 1, 0);
    auto issue_small = [&](int step, int buf) {
        if (wave >= 4) return;
        if ((PROBE & 8) && step > 2) return;
        const int lane = opaque_lane();
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int p = wave + 4 * h;
            const int fr = 4 * p + (lane >> 4);
            const int col = 8 * ((lane & 15) ^ w64_sw(fr));
            const int m = min(step * W_F + fr, last_row);
            wglds16_off(dz_base, (unsigned)(m * lddz + col) * 2u, smem_lds + L::DZ + buf * 16384 + p * 1024);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int p = wave + 4 * h;
            const int fr = 4 * p + (lane >> 4);
            const int col = 8 * ((lane & 15) ^ w64_sw(fr));
            const int m = min(step * W_F + fr, last_row);
            wglds16_off(h1_base, (unsigned)(m * ldh + col) * 2u, smem_lds + L::H1 + buf * 16384 + p * 1024);
        }
    };
    // The range ends inside step u: the tile rows past it are copies of the last row - zero them in the dZ2 tile (P3 and the bias sums
    // read it; P1 masks its own results).  All threads; the caller puts barriers around it.
    auto zero_tail = [&](int u, int buf) {
        const int valid = n_rows - u * W_F;               // frames of step u inside the range
        const int fr = tid >> 3;
        if (fr >= valid) {
            uint4* d = reinterpret_cast<uint4*>(smem + L::DZ + buf * 16384 + fr * 256 + (tid & 7) * 32);
            d[0] = uint4{0u, 0u, 0u, 0u};
            d[1] = uint4{0u, 0u, 0u, 0u};
        }
    };

    f32x16 acc[2][TKT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TKT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool bias_free = bslab != nullptr && (wave & 3) == 3;        // K <= 608: last 32-column tile of k-wave 3 is padding
    f32x16 acc3[2];                                       // P3: dZ2 columns 32 (wave >> 1) .. x H1 columns 64 (wave & 1) + 32 i ..
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[i][r] = 0.f;
    float db2p = 0.f;                                     // even waves: sum over frames of dZ2[., 32 (wave >> 1) + (lane & 31)], this lane's k half

    // P1 geometry: wave w owns hidden units 16 w .. 16 w + 15; 16x16x32 MFMA, A = W2T rows from LDS (unit l15, k chunk lq), B = dZ2 rows
    // (frame l15 of block t, k chunk lq); D: this lane holds units 16 w + 4 lq .. + 3 of frame 16 t + l15.
    // P1 of step u: dZ2 / H1 tiles of buffer tb, dZ1 tile to buffer yb2 (of 2); two rounds of two 16-frame blocks.  Frames past the
    // range (the last step's tail) give dZ1 = 0: their tile rows are copies of the range's last row.
    auto p1 = [&](int u, int tb, int yb2) {
        if (PROBE & 4) return;
        const int lane = opaque_lane();
        const int l15 = lane & 15, lq = lane >> 4;
        const int c16 = 2 * wave + (lq >> 1);                                             // 16-byte chunk of the 256-byte row
        // dual-use image: frame m = 16 t + l15 has sw(m) = sw(l15) (16 t changes neither m & 3 nor (m >> 2) & 3)
        const int swm = w64_sw(l15);
        const int b_row0 = L::DZ + l15 * 256;                                             // + 4096 t; chunk 4 ks + lq at (4 ks + lq) ^ swm
        const int h_off0 = L::H1 + l15 * 256 + ((c16 ^ swm) << 4) + 8 * (lq & 1);          // + 4096 t
        const int y_off0 = L::YS + l15 * 256 + ((c16 ^ swm) << 4) + 8 * (lq & 1);      // the dZ1 tile in the dual-use image as well
        const int a_row = 16 * wave + l15;
        const int w_off0 = L::W2T + a_row * 64 + ((lq ^ (((a_row >> 3) & 1) << 1)) << 4);  // + 8192 per k-tile
        const int bo = tb * 16384, yo = yb2 * 16384;
        const int f_left = n_rows - u * W_F - l15;        // frame 16 t + l15 of the step is inside the range iff 16 t < f_left
        // The four 16-frame blocks are four independent accumulation chains, streamed k-tile by k-tile: an accumulator comes round
        // every fourth MFMA (no dependent-issue stall), the operand reads run four MFMAs ahead in a ring of four fragments, and the
        // tiles' latency is paid once per step (as two rounds of two blocks - read 8, multiply 8, tail - it was paid twice, with the
        // 4-deep chains' latency and the VALU tail in line behind it: 1,700-2,000 cycles per step for 256 matrix cycles).
        bfv8 b[4], wf[2];
        bfv4 hv[4];
        f32x4 d[4];
        auto rd = [&](int i) -> bfv8 {                     // operand of MFMA i = 4 ks + t: frame block t, chunk 4 ks + lq
            return *reinterpret_cast<const bfv8*>(smem + bo + b_row0 + (i & 3) * 4096 + ((((i >> 2) * 4 + lq) ^ swm) << 4));
        };
        wf[0] = *reinterpret_cast<const bfv8*>(smem + w_off0);
        wf[1] = *reinterpret_cast<const bfv8*>(smem + w_off0 + 8192);
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = rd(i);
#pragma unroll
        for (int t = 0; t < 4; ++t) hv[t] = *reinterpret_cast<const bfv4*>(smem + bo + h_off0 + t * 4096);
#pragma unroll
        for (int t = 0; t < 4; ++t) d[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            d[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(i >> 2) & 1], b[i & 3], d[i & 3], 0, 0, 0);
            if (i + 4 < 16) b[i & 3] = rd(i + 4);
            if ((i & 3) == 3 && (i >> 2) + 2 < 4) wf[(i >> 2) & 1] = *reinterpret_cast<const bfv8*>(smem + w_off0 + ((i >> 2) + 2) * 8192);
        }
        const bool tail_step = f_left + l15 < W_F;        // wave-uniform: the range ends inside this step
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (float)hv[t][e];
                v[e] = d[t][e] * h * (1.f - h);
            }
            if (tail_step && !(16 * t < f_left)) v[0] = v[1] = v[2] = v[3] = 0.f;
            const u32x2 pk = u32x2{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                   __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
            *reinterpret_cast<u32x2*>(smem + yo + y_off0 + t * 4096) = pk;
        }
    };

    // P2 of one step and pass: the dZ1 tile of buffer yb2 against the ring rows of the frames' runs, as two halves of 32 frames (two
    // 16-frame k-steps each).  A frame's run ordinal is the population count of the run-start mask up to and including its bit, - 1;
    // local group gl = (ordinal >> 2) - g_first sits in ring group (rp + gl) % NG, runs outside [g_first, g_first + NG) read the zero
    // row, and so do pad frames (row index < 0).  with_bias: the pass that carries the bias sums.
    auto p2 = [&](int yb2, const StepMasks& mk4, int rp, int g_first, bool with_bias) {
        const int lane = opaque_lane();
        const int li = lane & 15, g4 = lane >> 4;
        const int q = li >> 2, p4 = li & 3;
        const int cgrp = 16 * (g4 & 1) + 4 * p4;
        const int rbase = 8 * (g4 >> 1) + q;                  // this lane's frame inside a 16-frame k-step (and + 4)
        // dZ1 tile (dual-use image: chunk ^ sw(row); the 8-byte stores of P1 were 8-way bank conflicted under the former (row & 3) << 2
        // swizzle - 16 lanes, 16 rows, two distinct 128-byte phases): frames rbase and rbase + 4 of a k-step differ in (row >> 2) & 3
        int yoff[2][2], xk[TKT];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int col = wn0 + i * 32 + cgrp;
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int row = rbase + 4 * hi;
                yoff[i][hi] = L::YS + row * PY + ((((col >> 3) ^ w64_sw(row)) << 4) | ((col & 7) << 1));
            }
        }
#pragma unroll
        for (int j = 0; j < TKT; ++j) {
            const int col = wk0 + j * 32 + cgrp;
            xk[j] = ((col >> 3) << 4) | ((col & 7) << 1);                  // byte offset inside a ring row before the swizzle
        }
        const int pc_lo = __builtin_popcount(mk4.lo);
        const int bias_row = with_bias ? L::ONES : L::ZROW;   // bias_free waves: the last column block multiplies dZ1 by ones
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int yb = yb2 * 16384 + half * 32 * PY;
            const unsigned mk = half ? mk4.hi : mk4.lo, pad = half ? mk4.phi : mk4.plo;
            const int pc0 = half ? pc_lo - 1 : -1;
            int xb[4], xs[4];                             // ring byte offset and swizzle of this lane's 4 frames of the half: 2 ks + hi
#pragma unroll
            for (int fi = 0; fi < 4; ++fi) {
                const int f = 16 * (fi >> 1) + rbase + 4 * (fi & 1);
                const int o = pc0 + __builtin_popcount(mk & ((2u << f) - 1u));
                const int gl = (o >> 2) - g_first;
                int rg = rp + gl;
                rg -= (rg >= NG) ? NG : 0;
                const bool in_pass = (unsigned)gl < (unsigned)NG && ((pad >> f) & 1u) == 0u;
                xb[fi] = in_pass ? L::X + (rg * 4 + (o & 3)) * PX : L::ZROW;
                xs[fi] = in_pass ? (o & 3) << 6 : 0;
            }
            bfv8 probe_frag;                               // PROBE & 2: stands for every operand (its value never matters)
            if (PROBE & 2) asm volatile("" : "=v"(probe_frag));
            auto rd_a = [&](int ks, int i) -> bfv8 {
                if (PROBE & 2) return probe_frag;
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(smem + yb + yoff[i][0] + ks * 16 * PY));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(smem + yb + yoff[i][1] + ks * 16 * PY));
                return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            auto rd_b = [&](int ks, int j) -> bfv8 {
                if (PROBE & 2) return probe_frag;
                const bool bias_blk = (bias_free && j == TKT - 1) || (PROBE & 1);
                const unsigned char* alo = smem + (bias_blk ? bias_row : xb[2 * ks]) + (xk[j] ^ xs[2 * ks]);
                const unsigned char* ahi = smem + (bias_blk ? bias_row : xb[2 * ks + 1]) + (xk[j] ^ xs[2 * ks + 1]);
                const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(alo));
                const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ahi));
                return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };
            // Fragment reads run four MFMA pairs ahead of their use (a rolling window instead of "all 14 reads, wait, 10 MFMAs" per
            // k-step; two pairs ahead measured 13 us slower at C2: the tr-reads take longer than two pairs' 128 matrix cycles to return
            // while the other waves and the DMA keep the LDS busy).
            bfv8 a[2][2], b[2 * TKT];
            a[0][0] = rd_a(0, 0);
            a[0][1] = rd_a(0, 1);
            constexpr int W = (PROBE & 16) ? 2 : 4;        // pairs the operand reads run ahead
            constexpr int TA = (PROBE & 16) ? 2 : 1;       // the pair at which the next k-step's dZ1 fragments are read
            b[0] = rd_b(0, 0);
            b[1] = rd_b(0, 1);
            if (W == 4) {
                b[2] = rd_b(0, 2);
                b[3] = rd_b(0, 3);
            }
            auto mm = [&](auto tc) {                       // t = ks * TKT + j, a compile-time constant
                constexpr int t = decltype(tc)::value;
                constexpr int ks = t / TKT, j = t % TKT;
                if constexpr (t + W < 2 * TKT) b[t + W] = rd_b((t + W) / TKT, (t + W) % TKT);
                if constexpr (t == TA) {
                    a[1][0] = rd_a(1, 0);
                    a[1][1] = rd_a(1, 1);
                }
                acc[0][j] = mg_mfma_32x32x16(a[ks][0], b[t], acc[0][j]);
                acc[1][j] = mg_mfma_32x32x16(a[ks][1], b[t], acc[1][j]);
                constexpr int n_reads = (PROBE & 2) ? 0 : (t + W < 2 * TKT ? 2 : 0) + (t == TA ? 4 : 0);
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
        }
    };

    // P3 of a step on its tiles (buffer tb): acc3[i] += dZ2^T[32 nb .. + 31, frames] . H1[frames, 32 (cb0 + i) .. + 31], both operands by
    // transposed reads of the dual-use tiles.  Lane 16 g4 + 4 q + p4 supplies (frame 16 ks + 8 (g4 >> 1) + q (+ 4), columns cg .. cg + 3)
    // and receives column 16 (g4 & 1) + (lane & 15) of those four frames: the 32x32x16 operand with the frames as contraction index.
    auto p3 = [&](int tb) {
        if (PROBE & 32) return;
        const int lane = opaque_lane();
        const int li = lane & 15, g4 = lane >> 4;
        const int q = li >> 2, p4 = li & 3;
        const int nb = wave >> 1, cb0 = 2 * (wave & 1);
        const int rb = 8 * (g4 >> 1) + q;
        const int cg = 16 * (g4 & 1) + 4 * p4;
        // frame r = 16 ks + rb + 4 hi: sw(r) does not depend on ks (16 ks changes neither r & 3 nor (r >> 2) & 3), so an operand's
        // address is one per-lane base per (operand, hi) + the k-step as an immediate offset (4096 ks)
        int ad[3][2];
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int r = rb + 4 * hi, swr = w64_sw(r);
            const int cols[3] = {32 * nb + cg, 32 * cb0 + cg, 32 * (cb0 + 1) + cg};
#pragma unroll
            for (int o = 0; o < 3; ++o)
                ad[o][hi] = (o == 0 ? L::DZ : L::H1) + tb * 16384 + r * 256 + ((((cols[o] >> 3) ^ swr) << 4) | ((cols[o] & 7) << 1));
        }
        auto rd = [&](int o, int ks) -> bfv8 {
            const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(smem + ad[o][0] + ks * 4096));
            const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(smem + ad[o][1] + ks * 4096));
            return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        const bfv2 one2 = bfv2{(__bf16)1.0f, (__bf16)1.0f};
        bfv8 a[2], b0[2], b1[2];
        a[0] = rd(0, 0);
        b0[0] = rd(1, 0);
        b1[0] = rd(2, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) {
                a[(ks + 1) & 1] = rd(0, ks + 1);
                b0[(ks + 1) & 1] = rd(1, ks + 1);
                b1[(ks + 1) & 1] = rd(2, ks + 1);
            }
            acc3[0] = mg_mfma_32x32x16(a[ks & 1], b0[ks & 1], acc3[0]);
            acc3[1] = mg_mfma_32x32x16(a[ks & 1], b1[ks & 1], acc3[1]);
            if ((wave & 1) == 0) {                         // the bias gradient from the dZ2^T fragment (this lane: one column, 8 frames)
                const bfv8 v = a[ks & 1];
                db2p = __builtin_amdgcn_fdot2_f32_bf16(bfv2{v[0], v[1]}, one2, db2p, false);
                db2p = __builtin_amdgcn_fdot2_f32_bf16(bfv2{v[2], v[3]}, one2, db2p, false);
                db2p = __builtin_amdgcn_fdot2_f32_bf16(bfv2{v[4], v[5]}, one2, db2p, false);
                db2p = __builtin_amdgcn_fdot2_f32_bf16(bfv2{v[6], v[7]}, one2, db2p, false);
            }
        }
    };

    // ---- per-step state (wave-uniform unless noted) ----------------------------------------------------------------------------------
    // cur: the step P2 works on; nxt: the step whose staged rows are fetched meanwhile.  Groups = 4 runs; rp = ring group of a step's
    // local group 0; have = are the first min(groups, NG) groups of the step in the ring (fetched ahead)?
    StepMasks mk_cur = {0u, 0u, 0u, 0u};
    int rp_cur = 0;
    bool have_cur = true;

    if (n_steps > 0) {
        // prologue: row indices of steps 0 and 1, staged rows and tiles of step 0 waited for, P1 of step 0; then the tiles of steps
        // 1 .. NBT - 1 in flight (past the end: copies of the last row, never used)
        if (wave == 0) {
            issue_rows(0);
            issue_rows(1);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        mk_cur = scan_mask(read_rows(0));
        issue_x(read_rows(0), mk_cur, 0, 0, min(groups_of(mk_cur), NG), 0);
        issue_small(0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (W_F > n_rows) {                               // the range ends inside step 0
            zero_tail(0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        p1(0, 0, 0);
        p3(0);
#pragma unroll
        for (int u = 1; u < NBT; ++u) issue_small(u, u);
    }
    int tb_next = 1 % NBT;                                // tile buffer of step + 1
    for (int step = 0; step < n_steps; ++step) {
        // Landed: staged rows of this step (when fetched ahead), tiles of step + 1, row indices of step + 1; dZ1(step) complete; the
        // buffers of the last step are free.  In flight with NBT == 3: the tile pieces of step + 2 (this wave's newest four).
        MG_STAMP(ta);
        if (NBT == 3)
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_wait, tb, ta);
#ifdef MG_STAMPS
        if (step == 0) ts1 = tb;
#endif
        const bool more = step + 1 < n_steps;
        if (more && (step + 2) * W_F > n_rows) {          // the range ends inside step + 1, whose tiles P1 / P3 read in this step
            zero_tail(step + 1, tb_next);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const int g_cur = groups_of(mk_cur);
        if (!have_cur) {                                  // this step's rows did not fit beside its predecessor's: fetch them now
            rp_cur = 0;
            issue_x(read_rows(step), mk_cur, 0, 0, min(g_cur, NG), step);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // The next step's run structure (whether its first groups fit in the ring beside this step's) comes from its row indices in
        // LDS: the read goes out now, the scan - a cross-lane shift, two ballots: all latency - runs once the wave has done something
        // else (waves 0-3: the tile pieces, which need nothing of it; waves 4-7: P2), so the round trip is not paid at the head.
        const int rv = read_rows(step + 1);               // past the last step: all pad frames, unused
        StepMasks mk_nxt = {0u, 0u, 0u, 0u};
        int g_nxt = 0, rp_nxt = 0;
        bool pref = false;
        auto scan_next = [&]() {
            mk_nxt = scan_mask(rv);
            g_nxt = groups_of(mk_nxt);
            pref = more && g_cur <= NG && g_cur + min(g_nxt, NG) <= NG;
            rp_nxt = rp_cur + g_cur;
            rp_nxt -= (rp_nxt >= NG) ? NG : 0;
        };
        const int tb_new = tb_next == 0 ? NBT - 1 : tb_next - 1;        // (step + NBT) % NBT == step % NBT
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_scan, ta, tb);
        if (wave < 4) {
            // fetches of this iteration: the row indices of step + 2 (wave 0), the tiles of step + 2, the staged rows of step + 1
            if (wave == 0) issue_rows(step + 2);
            issue_small(step + NBT, tb_new);
            scan_next();
            if (pref) issue_x(rv, mk_nxt, rp_nxt, 0, min(g_nxt, NG), step);
            MG_STAMP(tb);
            MG_STAMP_ADD(sum_fetch, tb, ta);
            if (more) {
                p1(step + 1, tb_next, (step + 1) & 1);
                p3(tb_next);
            }
            MG_STAMP(ta);
            MG_STAMP_ADD(sum_p1, ta, tb);
        }
        // P2; a step with more runs than the ring holds takes the remaining groups in passes through the whole ring (rare, and never
        // beside a prefetch: pref is false for such a step)
        for (int gf = 0;; gf += NG) {
            if (gf > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue_x(read_rows(step), mk_cur, 0, gf, min(g_cur - gf, NG), step);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            p2(step & 1, mk_cur, gf ? 0 : rp_cur, gf, gf == 0);
            if (gf + NG >= g_cur) break;
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_p2, tb, ta);
        if (wave >= 4) {
            if (more) {
                p1(step + 1, tb_next, (step + 1) & 1);
                p3(tb_next);
            }
            MG_STAMP(ta);
            MG_STAMP_ADD(sum_p1, ta, tb);
            scan_next();
            MG_STAMP(tb);
            MG_STAMP_ADD(sum_fetch, tb, ta);
        }
        mk_cur = mk_nxt;
        rp_cur = pref ? rp_nxt : 0;
        have_cur = pref || !more;
        tb_next = tb_next + 1 == NBT ? 0 : tb_next + 1;
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
    {
        // the second layer's slab of this frame range: [128 x N weights | 128 bias sums]; this workgroup's 128 columns of it
        float* out2 = slab2 + (size_t)s * sstride2;
        const int nb = wave >> 1, cb0 = 2 * (wave & 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * nb + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out2[(size_t)row * N + n0 + 32 * (cb0 + i) + lr] = acc3[i][r];
            }
        const float both = db2p + __shfl_xor(db2p, 32, 64);
        if ((wave & 1) == 0 && n0 == 0 && lh == 0) out2[(size_t)W_N2 * N + 32 * nb + lr] = both;
    }
#ifdef MG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MG_STAMP(ts3);
    MG_STAMP_REAL(tr1);
    const int sb = blockIdx.x;
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 2, ts2);
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 3, ts3);
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 4, tr0);
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 5, tr1);
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 6, sum_wait);
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 7, sum_fetch);
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 8, sum_p1);
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 9, sum_p2);
    MG_STAMP_STORE(g_stamps_f3, sb, wave, lane, 10, sum_scan);
#endif
}


#ifdef MG_EXPERIMENTS      // lab builds only (make lab / diag)
// -------------------------------------------------------------------------------------------------------------------------------------
// WOVEN form.  Stamps of the kernel above at C2 (profiles/r3_stamps_fused64.txt, cycles per 64-frame step of ~7,800 with 3,072 matrix
// cycles per SIMD): a wave spends ~550 at the barrier, ~550 deriving the next step's runs (an LDS read, a cross-lane shift, a ballot: all
// latency), 500-1,550 issuing its DMA pieces, 1,700-2,000 in P1 (two rounds of an LDS round trip, two 4-deep chains of small MFMAs, ~25
// dependent VALU instructions, an LDS store) and 3,300-4,300 in P2 - phases that follow each other inside a wave, each mostly latency,
// while its SIMD partner is in another such phase: the two fill 40 % of the matrix pipe.  Here ONE instruction stream per wave carries
// all of it: P2 of the step is 20 slots of two 32x32x16 MFMAs, and into the slots go, by slot number,
//   * P1 of the NEXT step, one 16-frame block per five slots: the block's operand reads one chain link ahead, one small MFMA per slot
//     (the chain's latency passes under the slot's large MFMAs), the VALU tail and the LDS store in the slot behind the chain;
//   * the fetches: the tile pieces of step + 2 in slots 0-3, the row indices in slot 0, the run scan in slot 4, the staged rows of
//     step + 1 (waves 0-3) from slot 6 on;
//   * P2's own operand reads two slots ahead, the dZ1 fragments and the frames' ring addresses of the next k-step three slots ahead.
// Every slot ends in a scheduling fence, so the order above is the order of the binary and hipcc's counted lgkmcnt waits follow from it.
// To make room (the weave keeps P1's and P2's operands live together) the W2^T block of P1 lives in LDS instead of 16 registers per
// lane, which leaves two tile buffers (tiles fetched two steps ahead) and a 5-group ring.  Same results as the kernels above.
// -------------------------------------------------------------------------------------------------------------------------------------
struct W64WMap {
    static constexpr int NG = 5;
    static constexpr int X = 0;                   // ring of 20 source rows x 1280 B (tr-swizzled by row & 3)
    static constexpr int ZROW = X + NG * W_GROUP;
    static constexpr int ONES = ZROW + 1280;
    static constexpr int DZ = ONES + 1280;        // 2 x (4 k-tiles x [64 rows x 64 B]), chunk ^ 2 ((row >> 3) & 1)
    static constexpr int H1 = DZ + 2 * 16384;     // 2 x [64 m][256 B], chunk c of row m at position c ^ (m & 15)
    static constexpr int YS = H1 + 2 * 16384;     // 2 x [64 m][256 B] tr-swizzled: the dZ1 tile
    static constexpr int W2T = YS + 2 * 16384;    // 4 k-tiles x [128 rows x 64 B], chunk ^ 2 ((row >> 3) & 1): this block's rows of W2^T
    static constexpr int RUNROW = W2T + 32768;    // 8 waves x int32[64]
    static constexpr int ROWS = RUNROW + 8 * 256; // 3 x int32[64]
    static constexpr int LDS = ROWS + 3 * 256;
};

MG_STAMP_DECL(g_stamps_f64w);

__global__ __launch_bounds__(512) void wgrad_fused64w_kernel(const uint16_t* __restrict__ dZ2, int lddz, const uint16_t* __restrict__ W2T,
                                                             int ldwt, const uint16_t* __restrict__ H1, int ldh,
                                                             const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                             int64_t M, int N, int K, int m_chunk, int n_splits,
                                                             float* __restrict__ slab, float* __restrict__ bslab, int64_t sstride) {
    using L = W64WMap;
    static_assert(L::LDS <= 160 * 1024, "one workgroup per CU");
    constexpr int NG = L::NG;
    constexpr int TKT = 5;
    constexpr int PY = 256, PX = W_BKT * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[L::LDS];
    const unsigned smem_lds = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MG_STAMPS
    unsigned long long ts0, ts1 = 0, ts2, ts3, tr0, tr1, ta, tb, sum_wait = 0, sum_stream = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    const int wn0 = (wave >> 2) * 64;
    const int wk0 = (wave & 3) * (TKT * 32);
    const int tiles_n = N / W_BNT;
    const int xcd = blockIdx.x & 7, jq = blockIdx.x >> 3;
    const int n0 = (jq % tiles_n) * W_BNT;
    const int s = (jq / tiles_n) * 8 + xcd;
    if (s >= n_splits) return;
    const int64_t m_lo = (int64_t)s * m_chunk;
    const int64_t m_hi = min(M, m_lo + (int64_t)m_chunk);
    const int n_rows = m_hi > m_lo ? (int)(m_hi - m_lo) : 0;
    const int n_steps = (n_rows + W_F - 1) / W_F;

    if (tid < 80) *reinterpret_cast<uint4*>(smem + L::ZROW + tid * 16) = uint4{0u, 0u, 0u, 0u};
    if (tid >= 128 && tid < 208) *reinterpret_cast<uint4*>(smem + L::ONES + (tid - 128) * 16) = uint4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};

    const unsigned long long a_ptr = (unsigned long long)A;
    const unsigned lda_bytes = (unsigned)lda * 2u;
    auto opaque_lane = [&]() -> int {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        return ln;
    };

    // ---- one-time: this workgroup's 128 rows of W2^T, 4 k-tiles of [128 rows x 64 B] (32 pieces, 4 per wave) -------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = wave * 4 + i;
        const int kt = p >> 3;
        const int row = 16 * (p & 7) + (lane >> 2);
        const int c = (lane & 3) ^ (((row >> 3) & 1) << 1);
        wglds16(W2T + (size_t)(n0 + row) * ldwt + 32 * kt + 8 * c, smem_lds + L::W2T + p * 1024);
    }

    auto issue_rows = [&](int u) {                        // wave 0 only: the row indices of step u into slot u % 3
        const int lane = opaque_lane();
        const int f = u * W_F + lane;
        int64_t m = m_lo + min(f, max(n_rows - 1, 0));
        if (m > M - 1) m = M - 1;
        const int32_t* src = rows + m;
        const unsigned lds_uni = __builtin_amdgcn_readfirstlane(smem_lds + L::ROWS + (u % 3) * 256);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(src), "s"(lds_uni)
                     : "memory");
    };
    auto read_rows_raw = [&](int u) -> int {
        const int lane = opaque_lane();
        return *reinterpret_cast<const int*>(smem + L::ROWS + (u % 3) * 256 + lane * 4);
    };
    auto fix_rows = [&](int u, int v) -> int {
        const int lane = opaque_lane();
        return (u * W_F + lane < n_rows) ? v : -1;
    };
    struct StepMasks {
        unsigned lo, hi, plo, phi;
    };
    auto scan_mask = [&](int rr) -> StepMasks {
        const int lane = opaque_lane();
        const int prev = __shfl_up(rr, 1, 64);
        const bool flag = (lane == 0) || (rr != prev);
        const unsigned long long mk = __ballot(flag), pk = __ballot(rr < 0);
        StepMasks m;
        m.lo = __builtin_amdgcn_readfirstlane((unsigned)mk);
        m.hi = __builtin_amdgcn_readfirstlane((unsigned)(mk >> 32));
        m.plo = __builtin_amdgcn_readfirstlane((unsigned)pk);
        m.phi = __builtin_amdgcn_readfirstlane((unsigned)(pk >> 32));
        return m;
    };
    auto runs_of = [&](const StepMasks& m) -> int { return __builtin_popcount(m.lo) + __builtin_popcount(m.hi); };
    auto groups_of = [&](const StepMasks& m) -> int { return (runs_of(m) + 3) >> 2; };
    int* my_runrow = reinterpret_cast<int*>(smem + L::RUNROW + wave * 256);

    // staged rows: the private run -> source row table of a step (waves 0-3), then its pieces one at a time
    auto x_table = [&](int rr, const StepMasks& mk) {
        const int lane = opaque_lane();
        const bool flag = ((lane < 32 ? (mk.lo >> lane) : (mk.hi >> (lane - 32))) & 1u) != 0u;
        const int ord = (int)__builtin_amdgcn_mbcnt_hi(mk.hi, __builtin_amdgcn_mbcnt_lo(mk.lo, 0u)) + (flag ? 1 : 0) - 1;
        if (flag) my_runrow[ord] = rr;
    };
    auto x_piece = [&](int pi, int n_runs, int rp, int g_first) {      // piece pi of the span of groups that starts at g_first
        const int lane = opaque_lane();
        const int gl = pi / 5, i = pi - 5 * gl;
        const int byte = i * 1024 + lane * 16;
        const int xr = byte >= 3 * PX ? 3 : byte >= 2 * PX ? 2 : byte >= PX ? 1 : 0;
        const int cpos = (byte - xr * PX) >> 4;
        const int xo = (cpos ^ ((xr & 3) << 2)) * 16;
        const int run = 4 * (g_first + gl) + xr;
        int src = my_runrow[min(run, 63)];
        if (run >= n_runs || src < 0) src = 0;
        int rg = rp + gl;
        rg -= (rg >= NG) ? NG : 0;
        wglds16((const void*)(a_ptr + (unsigned long long)(unsigned)src * lda_bytes + (unsigned)xo), smem_lds + L::X + rg * W_GROUP + i * 1024);
    };
    auto issue_x = [&](int rr, const StepMasks& mk, int rp, int g_first, int g_cnt, int rot) {      // the slow paths: all pieces at once
        const int n_pieces = 5 * g_cnt;
        int pi = (wave - rot) & 3;
        if (wave >= 4 || pi >= n_pieces) return;
        x_table(rr, mk);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int n_runs = runs_of(mk);
        for (; pi < n_pieces; pi += 4) x_piece(pi, n_runs, rp, g_first);
    };

    // tile pieces: h = 0, 1: dZ2 pieces wave + 8 h (piece p: k-tile p >> 2, rows 16 (p & 3) ..), h = 2, 3: H1 pieces wave + 8 (h - 2)
    const uint16_t* dz_base = dZ2 + (size_t)m_lo * lddz;
    const uint16_t* h1_base = H1 + (size_t)m_lo * ldh + n0;
    const int last_row = max(n_rows - 1, 0);
    auto tile_piece = [&](int step, int buf, int h) {
        const int lane = opaque_lane();
        const int p = wave + 8 * (h & 1);
        if (h < 2) {
            const int dz_row = 16 * (p & 3) + (lane >> 2);
            const int dz_col = 32 * (p >> 2) + 8 * ((lane & 3) ^ (((dz_row >> 3) & 1) << 1));
            const int mz = min(step * W_F + dz_row, last_row);
            wglds16_off(dz_base, (unsigned)(mz * lddz + dz_col) * 2u, smem_lds + L::DZ + buf * 16384 + p * 1024);
        } else {
            const int h1_row = 4 * p + (lane >> 4);
            const int h1_col = 8 * ((lane & 15) ^ (h1_row & 15));
            const int mh = min(step * W_F + h1_row, last_row);
            wglds16_off(h1_base, (unsigned)(mh * ldh + h1_col) * 2u, smem_lds + L::H1 + buf * 16384 + p * 1024);
        }
    };

    f32x16 acc[2][TKT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TKT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool bias_free = bslab != nullptr && (wave & 3) == 3;        // K <= 608: last 32-column tile of k-wave 3 is padding

    // One step's stream: P2 of `step` (buffer yb2, masks mk4, ring base rp, pass g_first) and, when weave is set, P1 of step + 1
    // (tiles tb -> dZ1 buffer yb2 ^ 1) and the fetches.  State handed back: the next step's masks (scanned in slot 4).
    struct Next {
        StepMasks mk;
        int g, rp;
        bool pref;
    };
    auto stream = [&](int step, int yb2, const StepMasks& mk4, int rp, int g_first, int g_cur, bool weave, bool more, int tb, Next& nx) {
        const int lane = opaque_lane();
        const int li = lane & 15, g4 = lane >> 4;
        const int q = li >> 2, p4 = li & 3;
        const int cgrp = 16 * (g4 & 1) + 4 * p4;
        const int rbase = 8 * (g4 >> 1) + q;              // this lane's frame inside a 16-frame k-step (and + 4)
        const int sw = q << 2;
        int yoff[2], xk[TKT];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int col = wn0 + i * 32 + cgrp;
            yoff[i] = L::YS + yb2 * 16384 + rbase * PY + ((((col >> 3) ^ sw) << 4) | ((col & 7) << 1));
        }
#pragma unroll
        for (int j = 0; j < TKT; ++j) {
            const int col = wk0 + j * 32 + cgrp;
            xk[j] = ((col >> 3) << 4) | ((col & 7) << 1);
        }
        const int bias_row = weave || g_first == 0 ? L::ONES : L::ZROW;       // the pass that carries the bias sums (pass 0)
        const int pc_lo = __builtin_popcount(mk4.lo);
        // ring byte offset and swizzle of this lane's two frames of k-step ks (0..3): index [ks & 1][hi]
        int xb[2][2], xs[2][2];
        auto frame_addr = [&](int ks) {
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int f = 16 * (ks & 1) + rbase + 4 * hi;          // bit inside the half's mask
                const unsigned mk = ks >= 2 ? mk4.hi : mk4.lo, pad = ks >= 2 ? mk4.phi : mk4.plo;
                const int o = (ks >= 2 ? pc_lo - 1 : -1) + __builtin_popcount(mk & ((2u << f) - 1u));
                const int gl = (o >> 2) - g_first;
                int rg = rp + gl;
                rg -= (rg >= NG) ? NG : 0;
                const bool in_pass = (unsigned)gl < (unsigned)NG && ((pad >> f) & 1u) == 0u;
                xb[ks & 1][hi] = in_pass ? L::X + (rg * 4 + (o & 3)) * PX : L::ZROW;
                xs[ks & 1][hi] = in_pass ? (o & 3) << 6 : 0;
            }
        };
        auto rd_a = [&](int ks, int i) -> bfv8 {
            const unsigned char* ad = smem + yoff[i] + ks * 16 * PY;
            const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad));
            const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ad + 4 * PY));
            return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };
        auto rd_b = [&](int ks, int j) -> bfv8 {
            const bool bias_blk = bias_free && j == TKT - 1;
            const unsigned char* alo = smem + (bias_blk ? bias_row : xb[ks & 1][0]) + (xk[j] ^ xs[ks & 1][0]);
            const unsigned char* ahi = smem + (bias_blk ? bias_row : xb[ks & 1][1]) + (xk[j] ^ xs[ks & 1][1]);
            const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(alo));
            const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(ahi));
            return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        };

        // ---- P1 of step + 1: this wave's 16 hidden units, one 16-frame block k at a time --------------------------------------------
        const int l15 = lane & 15, lq = lane >> 4;
        const int c16 = 2 * wave + (lq >> 1);
        const int a_row = 16 * wave + l15;
        const int w_off0 = L::W2T + a_row * 64 + ((lq ^ (((a_row >> 3) & 1) << 1)) << 4);               // + 8192 per k-tile
        const int b_off0 = L::DZ + tb * 16384 + l15 * 64 + ((lq ^ (((l15 >> 3) & 1) << 1)) << 4);       // + 1024 k + 4096 per k-tile
        const int h_off0 = L::H1 + tb * 16384 + l15 * 256 + ((c16 ^ l15) << 4) + 8 * (lq & 1);          // + 4096 k
        const int y_off0 = L::YS + (yb2 ^ 1) * 16384 + l15 * 256 + ((c16 ^ ((l15 & 3) << 2)) << 4) + 8 * (lq & 1);
        const int f_left = n_rows - (step + 1) * W_F - l15;
        bfv8 pw[2], pb[2];
        bfv4 hv;
        f32x4 pd = {0.f, 0.f, 0.f, 0.f};
        auto p1_read = [&](int k, int c) {
            pw[c & 1] = *reinterpret_cast<const bfv8*>(smem + w_off0 + c * 8192);
            pb[c & 1] = *reinterpret_cast<const bfv8*>(smem + b_off0 + k * 1024 + c * 4096);
        };
        auto p1_mfma = [&](int c) {
            if (c == 0) pd = f32x4{0.f, 0.f, 0.f, 0.f};
            pd = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pw[c & 1], pb[c & 1], pd, 0, 0, 0);
        };
        auto p1_hv = [&](int k) { hv = *reinterpret_cast<const bfv4*>(smem + h_off0 + k * 4096); };
        auto p1_finish = [&](int k) {
            float v[4];
            const bool inside = 16 * k < f_left;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float h = (float)hv[e];
                v[e] = inside ? pd[e] * h * (1.f - h) : 0.f;
            }
            const u32x2 pk = u32x2{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                   __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
            *reinterpret_cast<u32x2*>(smem + y_off0 + k * 4096) = pk;
        };
        const bool do_p1 = weave && more;

        // ---- fetch state -------------------------------------------------------------------------------------------------------------
        int rv = 0;                                       // raw row index of frame `lane` of step + 1
        int x_np = 0, x_nruns = 0;                        // pieces / runs of the staged rows of step + 1 (0: nothing to fetch ahead)

        bfv8 a[2][2], b[4];                               // dZ1 fragments of k-step ks in a[ks & 1]; operand ring b[t & 3]
        frame_addr(0);
        a[0][0] = rd_a(0, 0);
        a[0][1] = rd_a(0, 1);
        b[0] = rd_b(0, 0);
        b[1] = rd_b(0, 1);
        if (weave) rv = read_rows_raw(step + 1);
        __builtin_amdgcn_sched_barrier(0);

        auto slot = [&](auto tc) {
            constexpr int t = decltype(tc)::value;
            constexpr int ks = t / TKT, j = t % TKT;
            // reads: P2's operand of slot t + 2; the dZ1 fragments of the next k-step; P1's chain operands one link ahead
            if constexpr (t + 2 < 4 * TKT) b[(t + 2) & 3] = rd_b((t + 2) / TKT, (t + 2) % TKT);
            if constexpr (j == 2 && ks + 1 < 4) {
                a[(ks + 1) & 1][0] = rd_a(ks + 1, 0);
                a[(ks + 1) & 1][1] = rd_a(ks + 1, 1);
            }
            constexpr int k1 = t / 5, c1 = t % 5;          // P1: block k1, position c1 inside the block's five slots
            if (do_p1) {
                if constexpr (c1 < 4) p1_read(k1, c1);
                if constexpr (c1 == 3) p1_hv(k1);
                if constexpr (c1 == 0 && k1 > 0) p1_finish(k1 - 1);
                if constexpr (c1 >= 1) p1_mfma(c1 - 1);
            }
            // fetches
            if (weave) {
                if constexpr (t == 0) {
                    if (wave == 0) issue_rows(step + 2);
                }
                if constexpr (t < 4) tile_piece(step + 2, step & 1, t);
                if constexpr (t == 4) {
                    nx.mk = scan_mask(fix_rows(step + 1, rv));
                    nx.g = groups_of(nx.mk);
                    nx.pref = more && g_cur <= NG && g_cur + min(nx.g, NG) <= NG;
                    nx.rp = rp + g_cur;
                    nx.rp -= (nx.rp >= NG) ? NG : 0;
                    x_np = (nx.pref && wave < 4) ? 5 * min(nx.g, NG) : 0;
                    x_nruns = runs_of(nx.mk);
                }
                if constexpr (t == 5) {
                    if (x_np > 0) x_table(fix_rows(step + 1, rv), nx.mk);
                }
                if constexpr (t >= 6 && t < 13) {
                    const int pi = ((wave - step) & 3) + 4 * (t - 6);
                    if (pi < x_np) x_piece(pi, x_nruns, nx.rp, 0);
                }
            }
            // P2
            if constexpr (j == 1 && ks + 1 < 4) frame_addr(ks + 1);
            acc[0][j] = mg_mfma_32x32x16(a[ks & 1][0], b[t & 3], acc[0][j]);
            acc[1][j] = mg_mfma_32x32x16(a[ks & 1][1], b[t & 3], acc[1][j]);
            __builtin_amdgcn_sched_barrier(0);
        };
        slot(std::integral_constant<int, 0>{});
        slot(std::integral_constant<int, 1>{});
        slot(std::integral_constant<int, 2>{});
        slot(std::integral_constant<int, 3>{});
        slot(std::integral_constant<int, 4>{});
        slot(std::integral_constant<int, 5>{});
        slot(std::integral_constant<int, 6>{});
        slot(std::integral_constant<int, 7>{});
        slot(std::integral_constant<int, 8>{});
        slot(std::integral_constant<int, 9>{});
        slot(std::integral_constant<int, 10>{});
        slot(std::integral_constant<int, 11>{});
        slot(std::integral_constant<int, 12>{});
        slot(std::integral_constant<int, 13>{});
        slot(std::integral_constant<int, 14>{});
        slot(std::integral_constant<int, 15>{});
        slot(std::integral_constant<int, 16>{});
        slot(std::integral_constant<int, 17>{});
        slot(std::integral_constant<int, 18>{});
        slot(std::integral_constant<int, 19>{});
        if (do_p1) p1_finish(3);                          // (slots 6-12 take 7 x 4 = 28 staged-row pieces; a span has at most 5 NG = 25)
    };

    // ---- per-step state --------------------------------------------------------------------------------------------------------------
    StepMasks mk_cur = {0u, 0u, 0u, 0u};
    int rp_cur = 0;
    bool have_cur = true;
    if (n_steps > 0) {
        // prologue: row indices of steps 0 and 1, staged rows and tiles of step 0 (and W2^T) waited for, P1 of step 0 as the
        // un-woven chain, tiles of step 1 in flight
        if (wave == 0) {
            issue_rows(0);
            issue_rows(1);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        const int r0 = fix_rows(0, read_rows_raw(0));
        mk_cur = scan_mask(r0);
        issue_x(r0, mk_cur, 0, 0, min(groups_of(mk_cur), NG), 0);
#pragma unroll
        for (int h = 0; h < 4; ++h) tile_piece(0, 0, h);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const int lane = opaque_lane();
            const int l15 = lane & 15, lq = lane >> 4;
            const int c16 = 2 * wave + (lq >> 1);
            const int a_row = 16 * wave + l15;
            const int w_off0 = L::W2T + a_row * 64 + ((lq ^ (((a_row >> 3) & 1) << 1)) << 4);
            const int b_off0 = L::DZ + l15 * 64 + ((lq ^ (((l15 >> 3) & 1) << 1)) << 4);
            const int h_off0 = L::H1 + l15 * 256 + ((c16 ^ l15) << 4) + 8 * (lq & 1);
            const int y_off0 = L::YS + l15 * 256 + ((c16 ^ ((l15 & 3) << 2)) << 4) + 8 * (lq & 1);
            const int f_left = n_rows - l15;
            for (int k = 0; k < 4; ++k) {
                f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bfv8 w = *reinterpret_cast<const bfv8*>(smem + w_off0 + c * 8192);
                    const bfv8 bb = *reinterpret_cast<const bfv8*>(smem + b_off0 + k * 1024 + c * 4096);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, bb, d, 0, 0, 0);
                }
                const bfv4 hv = *reinterpret_cast<const bfv4*>(smem + h_off0 + k * 4096);
                float v[4];
                const bool inside = 16 * k < f_left;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = (float)hv[e];
                    v[e] = inside ? d[e] * h * (1.f - h) : 0.f;
                }
                const u32x2 pk = u32x2{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                       __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
                *reinterpret_cast<u32x2*>(smem + y_off0 + k * 4096) = pk;
            }
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) tile_piece(1, 1, h);
    }
    for (int step = 0; step < n_steps; ++step) {
        // Landed: staged rows of this step (when fetched ahead), tiles and row indices of step + 1; dZ1(step) complete; the buffers of
        // the last step are free.
        MG_STAMP(ta);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_wait, tb, ta);
#ifdef MG_STAMPS
        if (step == 0) ts1 = tb;
#endif
        const bool more = step + 1 < n_steps;
        const int g_cur = groups_of(mk_cur);
        if (!have_cur) {                                  // this step's rows did not fit beside its predecessor's: fetch them now
            rp_cur = 0;
            issue_x(fix_rows(step, read_rows_raw(step)), mk_cur, 0, 0, min(g_cur, NG), step);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        Next nx = {{0u, 0u, 0u, 0u}, 0, 0, false};
        for (int gf = 0;; gf += NG) {
            if (gf > 0) {                                 // more runs than the ring holds: the remaining groups in passes (rare)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue_x(fix_rows(step, read_rows_raw(step)), mk_cur, 0, gf, min(g_cur - gf, NG), step);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            Next scratch = nx;
            stream(step, step & 1, mk_cur, gf ? 0 : rp_cur, gf, g_cur, gf == 0, more, (step + 1) & 1, gf == 0 ? nx : scratch);
            if (gf + NG >= g_cur) break;
        }
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_stream, ta, tb);
        mk_cur = nx.mk;
        rp_cur = nx.pref ? nx.rp : 0;
        have_cur = nx.pref || !more;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tiles fetched past the end
#ifdef MG_STAMPS
    MG_STAMP(ts2);
#endif

    const int lr = lane & 31, lh = lane >> 5;
    float* out = slab + (size_t)s * sstride;
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
    MG_STAMP_STORE(g_stamps_f64w, sb, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_f64w, sb, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_f64w, sb, wave, lane, 2, ts2);
    MG_STAMP_STORE(g_stamps_f64w, sb, wave, lane, 3, ts3);
    MG_STAMP_STORE(g_stamps_f64w, sb, wave, lane, 4, tr0);
    MG_STAMP_STORE(g_stamps_f64w, sb, wave, lane, 5, tr1);
    MG_STAMP_STORE(g_stamps_f64w, sb, wave, lane, 6, sum_wait);
    MG_STAMP_STORE(g_stamps_f64w, sb, wave, lane, 7, sum_stream);
#endif
}

#endif  // MG_EXPERIMENTS

// Launch helper of mg_linear_bwd_fused2_slabs_bf16 (bwd_fused_bf16.hip): P1 + P2 + P3 in one launch.
void mg_launch_fused3(const uint16_t* dZ2, int lddz, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh, const uint16_t* A, int lda,
                      const int32_t* rows, int64_t M, int N, int K, int m_chunk, int n_splits, float* slab, float* bslab, int64_t sstride,
                      float* slab2, int64_t sstride2, hipStream_t st) {
    const dim3 grid((unsigned)((N / W_BNT) * mg_align_up((size_t)n_splits, 8))), block(512);
    hipLaunchKernelGGL(wgrad_fused3_kernel<0>, grid, block, 0, st, dZ2, lddz, W2T, ldwt, H1, ldh, A, lda, rows, M, N, K, m_chunk, n_splits, slab,
                       bslab, sstride, slab2, sstride2);
}

// Launch helper used by fused_launch (bwd_fused_bf16.hip): the 64-frame-step kernel on the plan of the 32-frame one (same frame
// ranges, same slabs - the results are bit-identical).  nbt: 3 = tiles fetched three steps ahead (5 ring groups), 2 = two (8 groups); lab builds: 1 = the woven experiment, 100 + mask = probes.
void mg_launch_fused64(int nbt, const uint16_t* dZ2, int lddz, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh, const uint16_t* A,
                       int lda, const int32_t* rows, int64_t M, int N, int K, int m_chunk, int n_splits, float* slab, float* bslab,
                       int64_t sstride, hipStream_t st) {
    const dim3 grid((unsigned)((N / W_BNT) * mg_align_up((size_t)n_splits, 8))), block(512);
#ifdef MG_EXPERIMENTS
#define W64_PROBE_CASE(P)                                                                                                                  \
    case 100 + P:                                                                                                                          \
        hipLaunchKernelGGL((wgrad_fused64_kernel<3, P>), grid, block, 0, st, dZ2, lddz, W2T, ldwt, H1, ldh, A, lda, rows, M, N, K, m_chunk,   \
                           n_splits, slab, bslab, sstride);                                                                                \
        return;
    switch (nbt) {
        W64_PROBE_CASE(1) W64_PROBE_CASE(2) W64_PROBE_CASE(4) W64_PROBE_CASE(6) W64_PROBE_CASE(8) W64_PROBE_CASE(12) W64_PROBE_CASE(14)
        W64_PROBE_CASE(16)
        default: break;
    }
    if (nbt == 1) {
        hipLaunchKernelGGL(wgrad_fused64w_kernel, grid, block, 0, st, dZ2, lddz, W2T, ldwt, H1, ldh, A, lda, rows, M, N, K, m_chunk, n_splits, slab,
                           bslab, sstride);
        return;
    }
#endif
    if (nbt == 2)
        hipLaunchKernelGGL(wgrad_fused64_kernel<2>, grid, block, 0, st, dZ2, lddz, W2T, ldwt, H1, ldh, A, lda, rows, M, N, K, m_chunk, n_splits, slab,
                           bslab, sstride);
    else
        hipLaunchKernelGGL(wgrad_fused64_kernel<3>, grid, block, 0, st, dZ2, lddz, W2T, ldwt, H1, ldh, A, lda, rows, M, N, K, m_chunk, n_splits, slab,
                           bslab, sstride);
}

#ifdef MG_STAMPS
#ifdef MG_EXPERIMENTS
extern "C" int mg_diag_read_stamps_f64w(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_f64w), bytes < sizeof(g_stamps_f64w) ? bytes : sizeof(g_stamps_f64w), 0, hipMemcpyDeviceToHost);
}
#endif
extern "C" int mg_diag_read_stamps_f3(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_f3), bytes < sizeof(g_stamps_f3) ? bytes : sizeof(g_stamps_f3), 0, hipMemcpyDeviceToHost);
}
extern "C" int mg_diag_read_stamps_f64(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_f64), bytes < sizeof(g_stamps_f64) ? bytes : sizeof(g_stamps_f64), 0, hipMemcpyDeviceToHost);
}
#endif
