// The LSTM recurrence in fp32 parity mode as ONE launch per direction (reference: torch.nn.LSTM behind RecurrentCuDNNWrapper,
// morgana/utils.py:345-393; the reference's shipped acoustic model stacks 8 of them, models/RNN_SPSS.py:36-37).
//
// The scheme of gru_persist.hip's fp32 kernels carried over to four gates: the batch is cut into 8 independent groups of
// R = ceil(B / 8) <= 16 items; a group is served by H / 16 workgroups (slots), slot s owning hidden units [16 s, 16 s + 16) of all
// four gates with its slice of W_hh RESIDENT IN REGISTERS for all T steps (128 VGPRs per lane at H = 512), exact-fp32 products on
// v_mfma_f32_16x16x4_f32.  Workgroups hand the state (forward: h_t, 64 bytes per item and slot; backward: the four gate
// gradients, 256 bytes per item and slot) to each other through an L2-resident ring with per-slot flags - protocol, placement
// independence and the bounded waits exactly as in gru_persist.hip / persist_common.h.
//
// Bit-identical to the launch-per-step kernels of lstm.hip (which is how the hand-off is tested): wave w owns the 16-deep contraction
// blocks w, w + 4, ... in the same order, the four waves' partial sums are added in the same order, and the cell arithmetic is the
// shared lstm_cell.h (contraction pinned off).  What it removes is the launch per time step: 11 us per step forward, 15 us backward
// at LSTM-512 (the shipped 8-layer acoustic model in fp32 mode: 99 ms per training step, of which ~60 ms were step launches).
//
// saved[b, t, :] past an item's length: gate values of the frozen state up to the group's longest sequence, zeros beyond it
// (unspecified by contract: include/morgana_hip.h, K3).  The backward treats such steps as inactive and emits exactly 0.
#include "lstm_cell.h"
#include "persist_common.h"

template <int KS>          // KS = H / 64 contraction blocks per wave
__global__ __launch_bounds__(256) void lstm_fwd_persist_f32_kernel(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                                   const float* __restrict__ b_hh, const int64_t* __restrict__ seq_len,
                                                                   int B, int T, int H, int R, float* __restrict__ hstate,
                                                                   float* __restrict__ cstate, float* __restrict__ out,
                                                                   float* __restrict__ saved, unsigned* sync, float* ring, int force_sc1) {
    __shared__ float red[4][4][GT * GT];
    __shared__ __attribute__((aligned(16))) float hb[GT][GT];
    __shared__ float res[6][GT * GT];              // out, i, f, g, o, c of the step for waves 2 and 3 (h itself goes through hb)
    __shared__ int s_abort, s_xcd;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int n_slots = H / GT;
    // which (group, slot) this workgroup serves: a ticket of the XCD it finds itself on (persist_common.h), not its block index
    const int claim = gp_claim_slot((gu32*)sync + GP_TICKET_OFFSET, n_slots, tid, &s_xcd, force_sc1 & 2);
    if (claim < 0) return;
    const int group = claim / GP_SLOTS, slot = claim % GP_SLOTS;
    const int row0 = group * R;
    const int nrows = min(R, B - row0);
    if (slot >= n_slots || nrows <= 0) return;
    const int j0 = slot * GT;
    gu32* flags = (gu32*)sync + group * GP_SLOTS;
    gu32* status = (gu32*)sync + GP_FLAG_WORDS;
    if (tid == 0) s_abort = 0;
    const int one_xcd = (force_sc1 & 1) ? 0 : gp_group_on_one_xcd((gu32*)sync + GP_GROUPS * GP_SLOTS + group * GP_SLOTS, slot, n_slots, tid, &s_xcd);
    if (one_xcd < 0) {
        if (tid == 0) __hip_atomic_store(status, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    int gmax = 0;
    for (int r = 0; r < nrows; ++r) {
        const int64_t n = seq_len ? seq_len[row0 + r] : (int64_t)T;
        gmax = max(gmax, (int)(n < T ? n : T));
    }
    // W_hh fragments: block i of this wave = columns 16 (wave + 4 i) + 4 q .. + 3 of rows j0 + li of the four gates
    f32x4 fw[4][KS];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float* wg = w_hh + ((size_t)g * H + j0 + li) * H + 4 * q;
#pragma unroll
        for (int i = 0; i < KS; ++i) fw[g][i] = *reinterpret_cast<const f32x4*>(wg + 16 * (wave + 4 * i));
    }
    // ring: [2 (epoch parity)][8 groups][H / 16 slots][R items][16 units] f32: a contraction block IS a slot's tile
    const unsigned par_bytes = (unsigned)(GP_GROUPS * n_slots * R * 64);
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, (int)(2 * par_bytes), 0x00020000);
    const unsigned rd_base = (unsigned)(((group * n_slots + wave) * R) * 64 + 16 * q);
    const unsigned rd_blk = (unsigned)(4 * R * 64);
    const unsigned wr_base = (unsigned)(((group * n_slots + slot) * R) * 64);

    const int bl = tid >> 4, jl = tid & 15;
    const int j = j0 + jl;
    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = b_hh[g * H + j];
    const bool mine = bl < nrows;
    const int b = row0 + (mine ? bl : 0);
    float hprev = hstate[((size_t)b * (T + 1)) * H + j];
    float cprev = cstate[((size_t)b * (T + 1)) * H + j];
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
    hb[bl][jl] = hprev;
    const float* xp = xproj + (size_t)b * T * 4 * H + j;
    float xg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xg[g] = xp[g * H];
    __syncthreads();

    auto publish = [&](int e) {
        if (wave == 0) {
            const int rrow = lane >> 2, piece = lane & 3;
            if (rrow < nrows) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(&hb[rrow][4 * piece]);
                const unsigned off = (e & 1) * par_bytes + wr_base + (unsigned)(rrow * 64 + piece * 16);
                if (one_xcd)
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 0);
                else
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) gp_store_flag(flags + slot, (unsigned)(e + 1), one_xcd);
        }
    };
    publish(0);

    for (int t = 0; t < gmax; ++t) {
        if (wave == 0 && !gp_wait_flags(flags, n_slots, (unsigned)(t + 1), lane)) s_abort = 1;
        gp_lds_barrier();
        if (s_abort) {
            if (tid == 0) __hip_atomic_store(status, 9u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        u32x4 raw[KS];
        {
            const unsigned off = (t & 1) * par_bytes + rd_base + (unsigned)((li < nrows ? li : 0) * 64);
#pragma unroll
            for (int i = 0; i < KS; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off + i * rd_blk, 0, 16);
        }
        // next step's input projection: requested behind the hand-off loads (in-order memory queue), consumed a step later
        const int t1 = t + 1 < T ? t + 1 : t;
        float xg1[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xg1[g] = xp[(size_t)t1 * 4 * H + g * H];
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            union { u32x4 u; f32x4 f; } a;
            a.u = raw[i];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[e], fw[g][i][e], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = (4 * q + r) * GT + li;
#pragma unroll
            for (int g = 0; g < 4; ++g) red[wave][g][e] = acc[g][r];
        }
        gp_lds_barrier();
        {
            const int e = bl * GT + jl;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) pre[g] = mg_lstm_pre(xg[g], red[0][g][e], red[1][g][e], red[2][g][e], red[3][g][e], bh[g]);
            const mg_lstm_cell_out c = mg_lstm_cell_exact(pre[0], pre[1], pre[2], pre[3], cprev);
            const bool active = t < len;
            hprev = active ? c.h : hprev;
            cprev = active ? c.c : cprev;
            hb[bl][jl] = hprev;
            res[0][e] = active ? c.h : 0.f;
            res[1][e] = c.i;
            res[2][e] = c.f;
            res[3][e] = c.g;
            res[4][e] = c.o;
            res[5][e] = cprev;
        }
        gp_lds_barrier();
        publish(t + 1);
        if (wave >= 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int e = (tid - 128) + 128 * half, rb = e >> 4, cj = j0 + (e & 15);
                if (rb < nrows) {
                    const int bb = row0 + rb;
                    const size_t row = (size_t)bb * T + t;
                    const size_t nxt = ((size_t)bb * (T + 1) + t + 1) * H + cj;
                    hstate[nxt] = hb[rb][e & 15];
                    cstate[nxt] = res[5][e];
                    out[row * H + cj] = res[0][e];
                    float* sv = saved + row * 4 * H + cj;
                    sv[0] = res[1][e];
                    sv[H] = res[2][e];
                    sv[2 * H] = res[3][e];
                    sv[3 * H] = res[4][e];
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[g] = xg1[g];
    }
    // beyond the group's longest sequence: states frozen, outputs zero, gate values (never read by the backward there) zero
    for (int t = gmax; t < T; ++t) {
        if (mine) {
            const size_t row = (size_t)b * T + t;
            const size_t nxt = ((size_t)b * (T + 1) + t + 1) * H + j;
            hstate[nxt] = hprev;
            cstate[nxt] = cprev;
            out[row * H + j] = 0.f;
            float* sv = saved + row * 4 * H;
            sv[j] = 0.f;
            sv[H + j] = 0.f;
            sv[2 * H + j] = 0.f;
            sv[3 * H + j] = 0.f;
        }
    }
}

// Backward recurrence.  Slot s owns d h[:, 16 s .. 16 s + 16): per step t (gmax - 1 down to 0, then the t = -1 pass that yields
// dh0 / dc0) it needs dgates_{t+1} of the whole group (R x 4H fp32: the hand-off, written by all slots in the step before),
// contracts it with its 16 columns of W_hh (4H x 16 fp32 = KS x 4 floats per lane, resident), applies the cell derivatives and
// publishes its 4 x 16 columns of dgates_t.  The elementwise carries (d loss / d h_{t-1} of finished items, d loss / d c_{t-1})
// stay in registers of the thread that owns the element.  Flag of a slot = gmax - t once row t is published.
// Ring: [2 (t parity)][8 groups][H / 16 slots][4 gates][R items][16 units] f32: contraction block (gate, slot) is one R x 64-byte
// tile.  The 4H-deep contraction is 4H / 64 = KS blocks per wave, requested in two rounds of KS / 2 (the fragments of W_hh take
// 128 VGPRs, a full round of hand-off tiles would take 128 more).
template <int KS>          // KS = 4 H / 64 contraction blocks per wave (even)
__global__ __launch_bounds__(256) void lstm_bwd_persist_f32_kernel(const float* __restrict__ grad_out, const float* __restrict__ grad_hn,
                                                                   const float* __restrict__ grad_cn, const float* __restrict__ cstate,
                                                                   const float* __restrict__ saved, const float* __restrict__ w_hh,
                                                                   const int64_t* __restrict__ seq_len, int B, int T, int H, int R,
                                                                   float* __restrict__ dgates, float* __restrict__ dh0,
                                                                   float* __restrict__ dc0, unsigned* sync, float* ring, int force_sc1) {
    __shared__ float red[4][GT * GT];
    __shared__ __attribute__((aligned(16))) float pub[4][GT][GT];   // di, df, dg, do tiles on their way to the ring and to dgates
    __shared__ int s_abort, s_xcd;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;
    const int n_slots = H / GT;
    // which (group, slot) this workgroup serves: a ticket of the XCD it finds itself on (persist_common.h), not its block index
    const int claim = gp_claim_slot((gu32*)sync + GP_TICKET_OFFSET, n_slots, tid, &s_xcd, force_sc1 & 2);
    if (claim < 0) return;
    const int group = claim / GP_SLOTS, slot = claim % GP_SLOTS;
    const int row0 = group * R;
    const int nrows = min(R, B - row0);
    if (slot >= n_slots || nrows <= 0) return;
    const int j0 = slot * GT;
    const int G = 4 * H;
    gu32* flags = (gu32*)sync + group * GP_SLOTS;
    gu32* status = (gu32*)sync + GP_FLAG_WORDS;
    if (tid == 0) s_abort = 0;
    const int one_xcd = (force_sc1 & 1) ? 0 : gp_group_on_one_xcd((gu32*)sync + GP_GROUPS * GP_SLOTS + group * GP_SLOTS, slot, n_slots, tid, &s_xcd);
    if (one_xcd < 0) {
        if (tid == 0) __hip_atomic_store(status, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    int gmax = 0;
    for (int r = 0; r < nrows; ++r) {
        const int64_t n = seq_len ? seq_len[row0 + r] : (int64_t)T;
        gmax = max(gmax, (int)(n < T ? n : T));
    }
    // block i of this wave = gate rows 16 (wave + 4 i) + 4 q + e of W_hh [4H, H]: W_hh[that row][j0 + li] as the B operand; its A
    // tile sits in the ring at (slot = block % n_slots, gate = block / n_slots) - wave-uniform, kept in scalar registers
    float fb[KS][4];
    unsigned rd_off[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        const int blk = wave + 4 * i, gate = blk / n_slots, sl = blk - gate * n_slots;
#pragma unroll
        for (int e = 0; e < 4; ++e) fb[i][e] = w_hh[(size_t)(16 * blk + 4 * q + e) * H + j0 + li];
        rd_off[i] = (unsigned)((((group * n_slots + sl) * 4 + gate) * R) * 64);
    }
    const unsigned par_bytes = (unsigned)(GP_GROUPS * n_slots * R * 256);
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, (int)(2 * par_bytes), 0x00020000);
    const unsigned wr_base = (unsigned)(((group * n_slots + slot) * R) * 256);

    const int bl = tid >> 4, jl = tid & 15;
    const int j = j0 + jl;
    const bool mine = bl < nrows;
    const int b = row0 + (mine ? bl : 0);
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
    float carry_h = (mine && grad_hn) ? grad_hn[(size_t)b * H + j] : 0.f;
    float carry_c = (mine && grad_cn) ? grad_cn[(size_t)b * H + j] : 0.f;
    const float* p_sv = saved + (size_t)b * T * 4 * H + j;
    const float* p_c = cstate + (size_t)b * (T + 1) * H + j;
    const float* p_g = grad_out ? grad_out + (size_t)b * T * H + j : nullptr;
    // steps at or beyond the group's longest sequence: every item of the group is inactive, the gate gradients are exactly zero
    for (int t = T - 1; t >= gmax; --t) {
        if (mine) {
            float* dgp = dgates + ((size_t)b * T + t) * G + j;
            dgp[0] = 0.f;
            dgp[H] = 0.f;
            dgp[2 * H] = 0.f;
            dgp[3 * H] = 0.f;
        }
    }
    const int t0 = gmax > 0 ? gmax - 1 : 0;
    float s_i = p_sv[(size_t)t0 * 4 * H], s_f = p_sv[(size_t)t0 * 4 * H + H], s_g = p_sv[(size_t)t0 * 4 * H + 2 * H],
          s_o = p_sv[(size_t)t0 * 4 * H + 3 * H], c_prev = p_c[(size_t)t0 * H], c_new = p_c[(size_t)(t0 + 1) * H],
          gout = p_g ? p_g[(size_t)t0 * H] : 0.f;
    __syncthreads();

    for (int t = gmax - 1; t >= -1; --t) {
        const bool need_mm = t + 1 < gmax;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (need_mm) {
            if (wave == 0 && !gp_wait_flags(flags, n_slots, (unsigned)(gmax - t - 1), lane)) s_abort = 1;
            gp_lds_barrier();
            if (s_abort) {
                if (tid == 0) __hip_atomic_store(status, 10u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            const unsigned off = ((t + 1) & 1) * par_bytes + (unsigned)((li < nrows ? li : 0) * 64 + 16 * q);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                u32x4 raw[KS / 2];
#pragma unroll
                for (int i = 0; i < KS / 2; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off + rd_off[half * (KS / 2) + i], 0, 16);
#pragma unroll
                for (int i = 0; i < KS / 2; ++i) {
                    union { u32x4 u; f32x4 f; } a;
                    a.u = raw[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[e], fb[half * (KS / 2) + i][e], acc, 0, 0, 0);
                }
            }
        }
        // the next step's saved gates, cell states and output gradient: requested here, consumed a step later
        const int t1 = t > 0 ? t - 1 : 0;
        const float s_i1 = p_sv[(size_t)t1 * 4 * H], s_f1 = p_sv[(size_t)t1 * 4 * H + H], s_g1 = p_sv[(size_t)t1 * 4 * H + 2 * H],
                    s_o1 = p_sv[(size_t)t1 * 4 * H + 3 * H], c_prev1 = p_c[(size_t)t1 * H], c_new1 = p_c[(size_t)(t1 + 1) * H],
                    gout1 = p_g ? p_g[(size_t)t1 * H] : 0.f;
        if (need_mm) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(4 * q + r) * GT + li] = acc[r];
            gp_lds_barrier();
        }
        {
            const int e = bl * GT + jl;
            // t + 1 >= gmax: the per-step kernel adds a zero accumulator here; carry + 0 is carry (also for -0: results compare equal)
            const float dh_state = need_mm ? mg_lstm_dstate(carry_h, red[0][e], red[1][e], red[2][e], red[3][e]) : carry_h;
            const float dc_state = carry_c;
            if (t < 0) {
                if (mine) {
                    dh0[(size_t)b * H + j] = dh_state;
                    dc0[(size_t)b * H + j] = dc_state;
                }
                break;
            }
            float di = 0.f, df = 0.f, dg = 0.f, d_o = 0.f, ch = dh_state, cc = dc_state;
            if (mine && t < len) {
                const mg_lstm_cell_grad cg = mg_lstm_cell_bwd(dh_state, dc_state, gout, s_i, s_f, s_g, s_o, c_prev, c_new);
                di = cg.di;
                df = cg.df;
                dg = cg.dg;
                d_o = cg.d_o;
                ch = 0.f;
                cc = cg.cc;
            }
            carry_h = ch;
            carry_c = cc;
            pub[0][bl][jl] = di;
            pub[1][bl][jl] = df;
            pub[2][bl][jl] = dg;
            pub[3][bl][jl] = d_o;
        }
        gp_lds_barrier();
        if (wave == 0) {
            // the slot's tile: 4 gates x R items x 64 bytes, contiguous in the ring; piece p = (gate, item, quarter)
            for (int pc = lane; pc < 16 * R; pc += 64) {
                const int gate = pc / (4 * R), rem = pc - gate * 4 * R, rrow = rem >> 2, piece = rem & 3;
                if (rrow < nrows) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(&pub[gate][rrow][4 * piece]);
                    const unsigned off = (t & 1) * par_bytes + wr_base + (unsigned)(pc * 16);
                    if (one_xcd)
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 16);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) gp_store_flag(flags + slot, (unsigned)(gmax - t), one_xcd);
        }
        if (wave >= 2) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int e = (tid - 128) + 128 * half, rb = e >> 4, cc = e & 15;
                if (rb < nrows) {
                    float* dgp = dgates + ((size_t)(row0 + rb) * T + t) * G + j0 + cc;
                    dgp[0] = pub[0][rb][cc];
                    dgp[H] = pub[1][rb][cc];
                    dgp[2 * H] = pub[2][rb][cc];
                    dgp[3 * H] = pub[3][rb][cc];
                }
            }
        }
        // pub is rewritten two barriers later (the matmul's and the cell's), behind which waves 0 and 2-3 have read it
        s_i = s_i1; s_f = s_f1; s_g = s_g1; s_o = s_o1; c_prev = c_prev1; c_new = c_new1; gout = gout1;
    }
}

extern "C" {

int mg_lstm_persist_f32_supported(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0 || g_mg_tuning[MG_TUNE_PERSISTENT] == 1) return 0;
    if (H % 64 != 0 || H > 512 || H < 256) return 0;
    if (mg_ceil_div(B, GP_GROUPS) > 16) return 0;                           // one 16-row MFMA tile per group
    return gp_device_holds((long)GP_GROUPS * (H / GT));
}

static int lpf_prepare(const char* what, int B, int T, int H, void* workspace, size_t workspace_bytes, hipStream_t st) {
    MG_CHECK_ARG(mg_lstm_persist_f32_supported(B, T, H), "%s: unsupported shape (B=%d T=%d H=%d)", what, B, T, H);
    MG_CHECK_ARG(workspace && ((uintptr_t)workspace % 16) == 0, "%s: workspace must be 16-byte aligned", what);
    if (workspace_bytes < mg_gru_persist_workspace_bytes(B, H)) {
        mg_set_error("%s: workspace of %zu bytes needed, got %zu", what, mg_gru_persist_workspace_bytes(B, H), workspace_bytes);
        return MG_EWORKSPACE;
    }
    if (hipMemsetAsync(workspace, 0, (size_t)GP_FLAG_WORDS * sizeof(unsigned), st) != hipSuccess) {
        mg_set_error("%s: memset failed", what);
        return MG_ELAUNCH;
    }
    return MG_OK;
}

int mg_lstm_fwd_persist_f32(const float* xproj, const float* w_hh, const float* b_hh, const int64_t* seq_len, int B, int T, int H,
                            float* hstate, float* cstate, float* out, float* saved, void* workspace, size_t workspace_bytes,
                            void* stream) {
    MG_CHECK_ARG(xproj && w_hh && b_hh && hstate && cstate && out && saved, "mg_lstm_fwd_persist_f32: null argument");
    MG_CHECK_ARG(((uintptr_t)w_hh % 16) == 0, "mg_lstm_fwd_persist_f32: w_hh must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int rc = lpf_prepare("mg_lstm_fwd_persist_f32", B, T, H, workspace, workspace_bytes, st);
    if (rc != MG_OK) return rc;
    const int R = (int)mg_ceil_div(B, GP_GROUPS);
    const unsigned grid = (unsigned)(GP_GROUPS * (H / GT));
    float* ring = (float*)((char*)workspace + GP_RING_OFFSET);
    const int force = g_mg_tuning[MG_TUNE_GRU_HANDOFF];
#define LPF_FWD(KS) hipLaunchKernelGGL((lstm_fwd_persist_f32_kernel<KS>), dim3(grid), dim3(256), 0, st, xproj, w_hh, b_hh, seq_len, B, T, H, R, hstate, cstate, out, saved, (unsigned*)workspace, ring, force)
    switch (H / 64) {
        case 4: LPF_FWD(4); break;
        case 5: LPF_FWD(5); break;
        case 6: LPF_FWD(6); break;
        case 7: LPF_FWD(7); break;
        default: LPF_FWD(8); break;
    }
    MG_CHECK_LAUNCH("mg_lstm_fwd_persist_f32");
    return MG_OK;
}

int mg_lstm_bwd_persist_f32(const float* grad_out, const float* grad_hn, const float* grad_cn, const float* cstate, const float* saved,
                            const float* w_hh, const int64_t* seq_len, int B, int T, int H, float* dgates, float* dh0, float* dc0,
                            void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(cstate && saved && w_hh && dgates && dh0 && dc0, "mg_lstm_bwd_persist_f32: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int rc = lpf_prepare("mg_lstm_bwd_persist_f32", B, T, H, workspace, workspace_bytes, st);
    if (rc != MG_OK) return rc;
    const int R = (int)mg_ceil_div(B, GP_GROUPS);
    const unsigned grid = (unsigned)(GP_GROUPS * (H / GT));
    float* ring = (float*)((char*)workspace + GP_RING_OFFSET);
    const int force = g_mg_tuning[MG_TUNE_GRU_HANDOFF];
#define LPF_BWD(KS) hipLaunchKernelGGL((lstm_bwd_persist_f32_kernel<KS>), dim3(grid), dim3(256), 0, st, grad_out, grad_hn, grad_cn, cstate, saved, w_hh, seq_len, B, T, H, R, dgates, dh0, dc0, (unsigned*)workspace, ring, force)
    switch (H / 64) {
        case 4: LPF_BWD(16); break;
        case 5: LPF_BWD(20); break;
        case 6: LPF_BWD(24); break;
        case 7: LPF_BWD(28); break;
        default: LPF_BWD(32); break;
    }
    MG_CHECK_LAUNCH("mg_lstm_bwd_persist_f32");
    return MG_OK;
}

}  // extern "C"
