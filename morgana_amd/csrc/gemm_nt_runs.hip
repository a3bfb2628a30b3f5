// K2 (bf16 throughput mode), first layer of a stack whose input is upsample_to_repetitions(lab, dur) (morgana/utils.py:175-228
// feeding README.rst:65-73 via utils.py:401-418):   C[M, N] = epi(gather(A, rows)[M, K] B[N, K]^T + bias),   bf16 out.
//
// Every frame row is still multiplied (the reference's order of operations), but the A operand is STAGED BY RUNS: consecutive frames
// of one phone point at the same source row (12.5 frames per phone at C2), so a 256-frame tile has ~21 distinct rows.  The LDS
// stage holds each distinct row of the tile once (64 slots) and the A fragments are read through a per-frame slot table (lanes of
// one run read the same address: a broadcast).  What that buys, measured on gemm_nt_persist<256> (gemm_bf16_big.hip, DESIGN.md
// section 4): its 256 x 256 x 32 stage is 32 LDS-DMA pieces per k-step, which is what a CU's texture-address path takes in during the
// k-step's 1,024 matrix cycles (LDS-DMA alone ran at 75 us, MFMAs alone 90-98 us, the kernel 181 us) - the feed, not the matrix
// pipe, was the ceiling.  With A staged by runs the feed is the B tile only, so the tile can be HALVED along N at the same feed per
// FLOP:  256 x 128 tiles, 256-thread workgroups, TWO workgroups per CU - one wave of each on every SIMD, and the two run
// unsynchronised: one's epilogue (bias + sigmoid + bf16 + stores: a quarter of the old kernel, during which no wave of the
// workgroup multiplies) passes under the other's k-steps, and a barrier wait of one is issue time for the other.
//   * ring: 4 stages of [64 run slots + a zero slot | 128 weight rows] x 64 B, LDS-DMA with the stream running on across tile
//     boundaries (as gemm_nt_persist), one counted vmcnt + barrier per k-step, 3 DMA pieces per wave and k-step (it was 4);
//   * arbitrary `rows` still work: a tile with more than 64 runs is multiplied in passes of 64 runs (rows outside the pass read
//     the zero slot), all-distinct rows cost four passes - the launcher sends inputs without a row map to gemm_nt_persist;
//   * the accumulator tile is C^T (weights as the MFMA A operand), epilogue through a wave-private LDS patch to 16-byte stores.
#include "common.h"

#include <type_traits>

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#define NR_EPI_BIAS 0
#define NR_EPI_BIAS_SIGMOID 1
#define NR_ZERO_ELEMS 16384
__device__ uint16_t g_nr_zero_row[NR_ZERO_ELEMS];      // source of pad rows / unused run slots / out-of-range weight rows
__device__ uint16_t g_nr_sink[64 * 8];                  // where the stores of rows past M go: every wave issues exactly NST stores

__device__ __forceinline__ void nr_glds16(const uint16_t* src, unsigned char* lds_wave_base) {
    const unsigned lds_off = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)lds_wave_base);
    const unsigned lds_uni = __builtin_amdgcn_readfirstlane(lds_off);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_uni)
                 : "memory");
}

#define NR_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define NR_WAIT_CASE(N) case N: NR_WAIT(N); break;

#define NR_MAXT 8
#define NR_BM 256
#define NR_BN 128
#define NR_RA 64                                        // run slots of a stage
#define NR_A_BYTES 4352                                 // 64 run slots + 4 zero slots (the first of them is what invalid rows read)
#define NR_B_BYTES (NR_BN * 64)
#define NR_STAGE (NR_A_BYTES + NR_B_BYTES)              // 12,544
#define NR_NS 4
#define NR_SLOT (NR_NS * NR_STAGE)                      // uint8 [MAXT][256]: run ordinal of each tile row
#define NR_RUNROW (NR_SLOT + NR_MAXT * NR_BM)           // int32 [MAXT][256]: source row of each run
#define NR_NRUNS (NR_RUNROW + NR_MAXT * NR_BM * 4)      // int32 [MAXT]
#define NR_PATCH (NR_NRUNS + 64)                        // 4 waves x [32 rows x 128 B]
#define NR_BIAS (NR_PATCH + 4 * 4096)
#define NR_LDS (NR_BIAS + NR_BN * 4)                    // 77,376 B: two workgroups per CU

// PF (prefetch): the fragments of k-step s + 1 are read while k-step s multiplies - the four weight fragments into a second register
// set at the head of the step, each frame fragment into its own registers right behind the four MFMAs that used it last - so the
// MFMAs of a step never wait for its LDS reads (they did at the head of every step: 12 reads, then 32 MFMAs).  The step's wait then
// covers the NEXT stage as well (one stage in flight instead of two); the first step of a tile reads as before.
// PROBE (lab builds only; results garbage): 1 = no epilogue, 2 = no LDS-DMA behind the first ring fill, 4 = no fragment reads,
// 8 = no MFMAs, 16 = every store of the epilogue goes to the L2-resident sink (no HBM write stream), 32 = non-temporal stores
// (results valid), 64 = eight start phases (results valid).
template <int EPI, bool PF = false, int PROBE = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt_runs_kernel(const uint16_t* __restrict__ A, int lda, const int32_t* __restrict__ rows,
                                                              int64_t M, int K, const uint16_t* __restrict__ Bm, int ldb, int N,
                                                              const float* __restrict__ bias, uint16_t* __restrict__ C, int ldc,
                                                              int tiles_m, int tiles_n) {
    constexpr int BK = 32, TMB = 8, TNB = 4, NS = NR_NS, NL = 3, NST = 16, SP = 128;      // wave tile: 8 x 4 blocks of 16 x 16
    static_assert(NR_LDS <= 80 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) unsigned char smem[NR_LDS];
    unsigned char* slot8 = smem + NR_SLOT;
    int* run_row = reinterpret_cast<int*>(smem + NR_RUNROW);
    int* nruns = reinterpret_cast<int*>(smem + NR_NRUNS);
    float* bias_lds = reinterpret_cast<float*>(smem + NR_BIAS);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave >> 1) * 128, wn0 = (wave & 1) * 64;
    const int n_kt = (K + BK - 1) / BK;

    // Virtual block v = blockIdx.x + i gridDim.x (gridDim.x a multiple of 8 tiles_n): the N tiles of one M tile go to blocks 8 apart,
    // which share an XCD and its L2; a workgroup keeps ONE N tile (its bias slice is parked once).
    auto tile_of = [&](int i, int& tile_m, int& tile_n) {
        const int v = blockIdx.x + i * gridDim.x;
        const int xcd = v & 7, jj = v >> 3;
        tile_n = jj % tiles_n;
        tile_m = (jj / tiles_n) * 8 + xcd;
    };
    int n_my = 0;
    for (int i = 0; i < NR_MAXT; ++i) {
        int tm, tn;
        tile_of(i, tm, tn);
        if (tm < tiles_m) n_my = i + 1;
    }
    if (n_my == 0) return;
    if (PROBE & 64) {                                  // probe: eight start phases an eighth of a tile period apart (results valid)
        const int phase = ((blockIdx.x >> 3) + 4 * (blockIdx.x >> 8)) & 7;
        for (int i = 0; i < phase; ++i) __builtin_amdgcn_s_sleep(80);
    }

    // ---- one-time: runs of equal consecutive source rows of every tile (wave w takes tiles w, w + 4: 4 rows per lane, one wave scan) ---
    for (int i = wave; i < n_my; i += 4) {
        int tm, tn;
        tile_of(i, tm, tn);
        int r[4], flag[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t m = (int64_t)tm * NR_BM + 4 * lane + e;
            r[e] = (tm < tiles_m && m < M) ? (rows ? rows[m] : (int)m) : -1;
        }
        const int prev = __shfl_up(r[3], 1, 64);
        flag[0] = (lane == 0) || (r[0] != prev);
        int cnt = flag[0];
#pragma unroll
        for (int e = 1; e < 4; ++e) {
            flag[e] = r[e] != r[e - 1];
            cnt += flag[e];
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        int ord = incl - cnt - 1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ord += flag[e];
            slot8[i * NR_BM + 4 * lane + e] = (unsigned char)ord;
            if (flag[e]) run_row[i * NR_BM + ord] = r[e];
        }
        if (lane == 63) nruns[i] = incl;
    }
    // the zero slots of every stage (never a DMA target), the bias slice
    if (tid < 16 * NS) *reinterpret_cast<u32x4_t*>(smem + (tid >> 4) * NR_STAGE + NR_RA * 64 + (tid & 15) * 16) = u32x4_t{0u, 0u, 0u, 0u};
    {
        int tm, tn;
        tile_of(0, tm, tn);
        for (int c = tid; c < NR_BN; c += 256) bias_lds[c] = (bias && tn * NR_BN + c < N) ? bias[tn * NR_BN + c] : 0.f;
    }
    __syncthreads();                                   // plain barrier: no LDS-DMA is in flight yet

    auto n_pass = [&](int i) -> int { return __builtin_amdgcn_readfirstlane((nruns[i] + NR_RA - 1) >> 6); };
    int g_total = 0;
    for (int i = 0; i < n_my; ++i) g_total += n_pass(i) * n_kt;

    // chunk swizzle of a 64-byte stage row (the same involution on the DMA source and on the fragment reads)
    // conflict free for ds_read_b128 when a lane reads chunk (lane >> 4) of row base + (lane & 15) (the 16x16x32 operand read)
    auto swz = [](int row) { return (0 - (row >> 2)) & 3; };
    // ---- issue cursor: tile i_t, pass i_p, k-tile i_k, ring slot i_s ---------------------------------------------------------
    const uint16_t* asrc;
    const uint16_t* bsrc[2];
    int i_t = 0, i_p = 0, i_k = 0, i_s = 0, i_np = 0;
    auto set_sources = [&](int i, int p) {
        int tm, tn;
        tile_of(i, tm, tn);
        {
            const int s = 16 * wave + (lane >> 2);                 // run slot of this lane's A piece (piece = wave)
            const int c = (lane & 3) ^ swz(s);
            const int run = NR_RA * p + s;
            const int r = (run < nruns[i]) ? run_row[i * NR_BM + run] : -1;
            asrc = (r >= 0 ? A + (size_t)r * lda : g_nr_zero_row) + c * 8;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int row = (wave * 2 + g) * 16 + (lane >> 2);
            const int c = (lane & 3) ^ swz(row);
            const int n = tn * NR_BN + row;
            bsrc[g] = (n < N ? Bm + (size_t)n * ldb : g_nr_zero_row) + c * 8;
        }
    };
    i_np = n_pass(0);
    set_sources(0, 0);
    int issued = 0;
    auto issue_next = [&]() {                          // next stage of the stream, if any is left
        if (i_t >= n_my) return;
        if ((PROBE & 2) && issued >= NS - 1) {             // probe: the cursor moves, nothing is fetched
            i_s = (i_s + 1 == NS) ? 0 : i_s + 1;
            if (++i_k == n_kt) {
                i_k = 0;
                if (++i_p == i_np) {
                    i_p = 0;
                    if (++i_t < n_my) i_np = n_pass(i_t);
                }
            }
            return;
        }
        ++issued;
        unsigned char* st = smem + i_s * NR_STAGE;
        nr_glds16(asrc + i_k * BK, st + wave * 1024);
        nr_glds16(bsrc[0] + i_k * BK, st + NR_A_BYTES + (wave * 2) * 1024);
        nr_glds16(bsrc[1] + i_k * BK, st + NR_A_BYTES + (wave * 2 + 1) * 1024);
        i_s = (i_s + 1 == NS) ? 0 : i_s + 1;
        if (++i_k == n_kt) {
            i_k = 0;
            if (++i_p == i_np) {
                i_p = 0;
                if (++i_t < n_my) i_np = n_pass(i_t);
            }
            if (i_t < n_my) set_sources(i_t, i_p);
        }
    };

    // v_mfma_f32_16x16x32_bf16 (one MFMA takes a whole 32-deep stage row; at equal matrix cycles the chip holds a higher clock under
    // this shape than under 32x32x16: 173 -> 151 us measured on this kernel with a timing probe, MI355X_MICROARCH.md DVFS item 7).
    // A operand = 16 weight rows, B operand = 16 frames: lane l reads chunk l >> 4 of row base + (l & 15).
    const int l16 = lane & 15, lg = lane >> 4;
    int aoff[TMB], boff[TNB];                          // byte offset of this lane's fragment inside a stage
#pragma unroll
    for (int j = 0; j < TNB; ++j) {
        const int row = wn0 + j * 16 + l16;
        boff[j] = NR_A_BYTES + row * 64 + ((lg ^ swz(row)) << 4);
    }
    auto set_frag_rows = [&](int ti, int p) {
#pragma unroll
        for (int i = 0; i < TMB; ++i) {
            const int rel = (int)slot8[ti * NR_BM + wm0 + i * 16 + l16] - NR_RA * p;
            const int s = ((unsigned)rel < (unsigned)NR_RA) ? rel : NR_RA;          // outside this pass: the zero slot
            aoff[i] = s * 64 + ((lg ^ swz(s)) << 4);
        }
    };

#pragma unroll
    for (int p = 0; p < NS - 1; ++p) issue_next();

    bfv8 fa[TMB], fb[2][TNB];
    auto read_frags = [&](const unsigned char* st) {
#pragma unroll
        for (int i = 0; i < TMB; ++i) fa[i] = *reinterpret_cast<const bfv8*>(st + aoff[i]);
#pragma unroll
        for (int j = 0; j < TNB; ++j) fb[0][j] = *reinterpret_cast<const bfv8*>(st + boff[j]);
        __builtin_amdgcn_sched_group_barrier(0x100, TMB + TNB, 0);
    };

    int g = 0, c_s = 0;
    for (int ti = 0; ti < n_my; ++ti) {
        int tile_m, tile_n;
        tile_of(ti, tile_m, tile_n);
        const int64_t m0 = (int64_t)tile_m * NR_BM;
        const int n0 = tile_n * NR_BN;
        f32x4 acc[TMB][TNB];                           // block (i, j): frames 16 i + (lane & 15), units 16 j + 4 (lane >> 4) + r
#pragma unroll
        for (int i = 0; i < TMB; ++i)
#pragma unroll
            for (int j = 0; j < TNB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        const int np = n_pass(ti);
        for (int p = 0; p < np; ++p) {
            set_frag_rows(ti, p);
            // one k-step; CUR (compile time) = which weight-fragment set holds this step's operands when PF is on
            auto kstep = [&](int kt, auto curc) {
                constexpr int CUR = decltype(curc)::value;
                // Stage g must have landed - and with PF stage g + 1 too, whose fragments this step reads ahead.  Issued so far:
                // min(g_total, g + NS - 1) stages; the previous tile's NST stores sit in the queue behind the stages issued before them.
                const int ahead = PF ? 1 : 0;
                const int young = max(min(g_total - 1 - ahead - g, NS - 2 - ahead), 0);
                // probe 128 (results garbage): the previous tile's stores are never waited for - the counted wait lets NST more
                // operations stay in flight at every k-step, so it can pass before its stage has landed
                const int allow = young * NL + ((ti > 0 && ((PROBE & 128) || (p == 0 && kt < NS - 1 - ahead))) ? NST : 0);
                __builtin_amdgcn_sched_barrier(0);
                switch ((PROBE & ~128) ? 0 : allow) {
                    NR_WAIT_CASE(3) NR_WAIT_CASE(6) NR_WAIT_CASE(16) NR_WAIT_CASE(19) NR_WAIT_CASE(22)
                    default: NR_WAIT(0); break;
                }
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* st = smem + c_s * NR_STAGE;
                issue_next();                          // refills the slot every wave finished with before this barrier
                c_s = (c_s + 1 == NS) ? 0 : c_s + 1;
                if (!PF) {
                    if (!(PROBE & 4) || g == 0) read_frags(st);
                    if (!(PROBE & 8)) {
#pragma unroll
                        for (int i = 0; i < TMB; ++i)
#pragma unroll
                            for (int j = 0; j < TNB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[0][j], fa[i], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, TMB * TNB, 0);
                    }
                } else {
                    if (kt == 0) {                     // first step of a tile / pass: nothing was read ahead
#pragma unroll
                        for (int i = 0; i < TMB; ++i) fa[i] = *reinterpret_cast<const bfv8*>(st + aoff[i]);
#pragma unroll
                        for (int j = 0; j < TNB; ++j) fb[CUR][j] = *reinterpret_cast<const bfv8*>(st + boff[j]);
                    }
                    const bool more = kt + 1 < n_kt;   // wave-uniform: the next stage belongs to this tile and pass
                    const unsigned char* nx = smem + c_s * NR_STAGE;
                    if (more) {
#pragma unroll
                        for (int j = 0; j < TNB; ++j) fb[CUR ^ 1][j] = *reinterpret_cast<const bfv8*>(nx + boff[j]);
                    }
#pragma unroll
                    for (int i = 0; i < TMB; ++i) {
#pragma unroll
                        for (int j = 0; j < TNB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[CUR][j], fa[i], acc[i][j], 0, 0, 0);
                        if (more) fa[i] = *reinterpret_cast<const bfv8*>(nx + aoff[i]);
                        __builtin_amdgcn_sched_group_barrier(0x008, TNB, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                ++g;
            };
            for (int kt = 0; kt < n_kt; kt += 2) {
                kstep(kt, std::integral_constant<int, 0>{});
                if (kt + 1 < n_kt) kstep(kt + 1, std::integral_constant<int, 1>{});
            }
        }

        // ---- epilogue: bias (+ sigmoid), bf16, whole 128-byte row segments through the wave's LDS patch ------------------------
        if (PROBE & 1) {                               // probe: the accumulators stay "used", nothing else happens
#pragma unroll
            for (int i = 0; i < TMB; ++i)
#pragma unroll
                for (int j = 0; j < TNB; ++j) asm volatile("" ::"v"(acc[i][j]));
            continue;
        }
        unsigned char* patch = smem + NR_PATCH + wave * 4096;
        const int prow = lane >> 3, pchunk = lane & 7;
        f32x4 bv[TNB];
#pragma unroll
        for (int j = 0; j < TNB; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bias_lds + wn0 + j * 16 + 4 * lg);
#pragma unroll
        for (int p = 0; p < TMB / 2; ++p) {            // a pass = 32 frames = two frame blocks through the 32-row patch
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int rl = 16 * ii + l16;
#pragma unroll
                for (int j = 0; j < TNB; ++j) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc[2 * p + ii][j][e] + bv[j][e];
                        if (EPI == NR_EPI_BIAS_SIGMOID) x = mg_sigmoid_fast(x);
                        v[e] = x;
                    }
                    const u32x2_t pk = u32x2_t{__builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]}),
                                               __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]})};
                    const int chunk = 2 * j + (lg >> 1);                      // columns 16 j + 4 lg .. + 3 of the 64-wide strip
                    *reinterpret_cast<u32x2_t*>(patch + rl * SP + ((chunk ^ (rl & 7)) << 4) + 8 * (lg & 1)) = pk;
                }
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl = it * 8 + prow;
                const u32x4_t o = *reinterpret_cast<const u32x4_t*>(patch + rl * SP + ((pchunk ^ (rl & 7)) << 4));
                const int64_t m = m0 + wm0 + p * 32 + rl;
                uint16_t* dst = (m < M && !(PROBE & 16)) ? C + (size_t)m * ldc + n0 + wn0 + pchunk * 8 : g_nr_sink + lane * 8;
                if (PROBE & 32)
                    __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t*>(dst));
                else
                    *reinterpret_cast<u32x4_t*>(dst) = o; // unconditional: the counted vmcnt waits rely on NST stores per wave
            }
        }
    }
}

// Launch helper used by mg_try_nt_big (gemm_bf16_big.hip): 1 if it launched, 0 if the shape does not qualify.
int mg_try_nt_runs(const uint16_t* A, int lda, const int32_t* rows, int64_t M, int K, const uint16_t* Bm, int ldb, int N,
                   const float* bias, uint16_t* C, int ldc, int sigmoid, hipStream_t st) {
    if (!rows || M < 2048 || M >= 2147483647LL || N % NR_BN != 0 || ldc != N) return 0;
    if (lda % 64 != 0 || ldb % 64 != 0 || lda > NR_ZERO_ELEMS - 64 || ldb > NR_ZERO_ELEMS - 64) return 0;
    if (lda < (K + 63) / 64 * 64 || ldb < (K + 63) / 64 * 64 || (K + 31) / 32 < 5) return 0;
    const int tiles_n = N / NR_BN;
    const int64_t tiles_m = mg_ceil_div(M, NR_BM);
    if (64 % tiles_n != 0) return 0;                   // gridDim.x / 8 a multiple of tiles_n
    const int64_t blocks = mg_ceil_div(tiles_m, 8) * 8 * tiles_n;
    int64_t g = 512;                                   // two resident workgroups per CU
    while (mg_ceil_div(blocks, g) > NR_MAXT) g += 512;
    if (g > blocks) g = blocks;
    if (g >= 2147483647LL) return 0;
    dim3 grid((unsigned)g), block(256);
#ifdef MG_EXPERIMENTS
    if (sigmoid && g_mg_tuning[MG_TUNE_FORM] >= 200 && g_mg_tuning[MG_TUNE_FORM] < 456) {      // lab builds: timing probes 200 + mask
#define NR_PROBE_CASE(P)                                                                                                                  \
    case 200 + P:                                                                                                                          \
        hipLaunchKernelGGL((gemm_nt_runs_kernel<NR_EPI_BIAS_SIGMOID, false, P>), grid, block, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, C, ldc, \
                           (int)tiles_m, tiles_n);                                                                                         \
        return 1;
        switch (g_mg_tuning[MG_TUNE_FORM]) {
            NR_PROBE_CASE(1) NR_PROBE_CASE(2) NR_PROBE_CASE(3) NR_PROBE_CASE(4) NR_PROBE_CASE(7) NR_PROBE_CASE(8) NR_PROBE_CASE(11) NR_PROBE_CASE(15) NR_PROBE_CASE(16) NR_PROBE_CASE(32) NR_PROBE_CASE(24) NR_PROBE_CASE(40) NR_PROBE_CASE(64) NR_PROBE_CASE(96) NR_PROBE_CASE(65) NR_PROBE_CASE(128)
            default: break;
        }
    }
#endif
    const bool pf = g_mg_tuning[MG_TUNE_FORM] == 16;   // A/B: fragments read one k-step ahead
    if (sigmoid && pf)
        hipLaunchKernelGGL((gemm_nt_runs_kernel<NR_EPI_BIAS_SIGMOID, true>), grid, block, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, C, ldc,
                           (int)tiles_m, tiles_n);
    else if (sigmoid)
        hipLaunchKernelGGL((gemm_nt_runs_kernel<NR_EPI_BIAS_SIGMOID>), grid, block, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, C, ldc, (int)tiles_m,
                           tiles_n);
    else if (pf)
        hipLaunchKernelGGL((gemm_nt_runs_kernel<NR_EPI_BIAS, true>), grid, block, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, C, ldc, (int)tiles_m,
                           tiles_n);
    else
        hipLaunchKernelGGL((gemm_nt_runs_kernel<NR_EPI_BIAS>), grid, block, 0, st, A, lda, rows, M, K, Bm, ldb, N, bias, C, ldc, (int)tiles_m, tiles_n);
    return 1;
}
