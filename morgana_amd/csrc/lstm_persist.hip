// K4p - the LSTM recurrence as ONE launch per direction (throughput mode, bf16 matmul operands), the scheme of gru_persist.hip
// applied to torch.nn.LSTM behind RecurrentCuDNNWrapper (reference: morgana/utils.py:345-393, models/RNN_SPSS.py:36-37; gate
// order i, f, g, o).  The batch is cut into 8 independent groups of R = ceil(B / 8) items; a group is served by H / 16
// workgroups, each owning 16 hidden units of all four gates with its 64 rows of W_hh (bf16: 64 VGPRs per lane at H = 512)
// resident for all T steps; h_t and c_t of an element stay in registers of the thread that owns it.  Hand-off of the bf16
// state between the workgroups of a group, the same-XCD / write-through forms, the L2-resident ring, the raw LDS barriers and
// the rule that nothing slow may be queued in front of wave 0's poll, loads and drain: see gru_persist.hip.
// Arithmetic: bf16 operands for the two per-step products, fp32 accumulation, fp32 cell with the v_exp_f32 / v_rcp_f32
// sigmoid and tanh; states, gate values and all outputs fp32.  The fp32 parity mode stays on lstm.hip's per-step kernels.
#include "persist_common.h"

__device__ __forceinline__ float lp_tanh_fast(float x) { return 2.f * mg_sigmoid_fast(2.f * x) - 1.f; }

template <int MT, int KS>
__global__ __launch_bounds__(256) void lstm_fwd_persist_kernel(const float* __restrict__ xproj, const uint16_t* __restrict__ w_bf, int ldw,
                                                               const float* __restrict__ b_hh, const int64_t* __restrict__ seq_len,
                                                               int B, int T, int H, int R, float* __restrict__ hstate,
                                                               float* __restrict__ cstate, uint16_t* __restrict__ hstate_bf,
                                                               float* __restrict__ out, float* __restrict__ saved, unsigned* sync,
                                                               uint16_t* ring, int force_sc1) {
    __shared__ float red[4][4][MT][GT * GT];
    __shared__ __attribute__((aligned(16))) uint16_t hb[MT][GT][GT];
    __shared__ float res[MT][7][GT * GT];          // h, c, out, i, f, g, o of the step for waves 2 and 3, which store them
    __shared__ int s_abort, s_xcd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    // W_hh fragments of this slot (4 gates x 16 units), for the whole launch
    const int kbase = wave * (H / 4) + 8 * q;
    gbf8 fw[4][KS];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const uint16_t* wp = w_bf + ((size_t)g * H + j0 + li) * ldw + kbase;
#pragma unroll
        for (int i = 0; i < KS; ++i) fw[g][i] = *reinterpret_cast<const gbf8*>(wp + 32 * i);
    }
    // ring of h tiles: [2 (epoch parity)][8 groups][H / 16 slots][R items][16 units] bf16 (as in gru_persist.hip)
    const unsigned par_bytes = (unsigned)(GP_GROUPS * n_slots * R * 32);
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, (int)(2 * par_bytes), 0x00020000);
    const unsigned rd_base = (unsigned)(((group * n_slots + (kbase >> 4)) * R) * 32 + 16 * (q & 1));
    const unsigned rd_kstep = (unsigned)(2 * R * 32);
    const unsigned wr_base = (unsigned)(((group * n_slots + slot) * R) * 32);

    const int bl = tid >> 4, jl = tid & 15;
    const int j = j0 + jl;
    float bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bh[g] = b_hh[g * H + j];
    float hprev[MT], cprev[MT], xg[MT][4];
    int len[MT];
    bool mine[MT];
    const float* xp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        mine[m] = 16 * m + bl < nrows;
        const int b = row0 + (mine[m] ? 16 * m + bl : 0);
        hprev[m] = hstate[((size_t)b * (T + 1)) * H + j];
        cprev[m] = cstate[((size_t)b * (T + 1)) * H + j];
        len[m] = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
        hb[m][bl][jl] = mg_f2bf(hprev[m]);
        xp[m] = xproj + (size_t)b * T * 4 * H + j;
#pragma unroll
        for (int g = 0; g < 4; ++g) xg[m][g] = xp[m][g * H];          // step 0; step t + 1's are requested during step t
    }
    __syncthreads();

    // wave 0 publishes the tile in hb as epoch e (state h_e), then raises the slot's flag to e + 1; wave 1 writes the bf16 shadow
    auto publish = [&](int e) {
        if (wave <= 1 && lane < 2 * 16 * MT) {
            const int rrow = lane >> 1, half = lane & 1;
            if (rrow < nrows) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(&hb[rrow >> 4][rrow & 15][8 * half]);
                if (wave == 0) {
                    const unsigned off = (e & 1) * par_bytes + wr_base + (unsigned)(rrow * 32 + half * 16);
                    if (one_xcd)
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 0);
                    else
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 16);
                } else if (e > 0) {
                    *reinterpret_cast<u32x4*>(hstate_bf + ((size_t)(row0 + rrow) * (T + 1) + e) * H + j0 + 8 * half) = v;
                }
            }
        }
        if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) gp_store_flag(flags + slot, (unsigned)(e + 1), one_xcd);
        }
    };
    publish(0);

    for (int t = 0; t < gmax; ++t) {
        if (wave == 0 && !gp_wait_flags(flags, n_slots, (unsigned)(t + 1), lane)) s_abort = 1;
        gp_lds_barrier();
        if (s_abort) {
            if (tid == 0) __hip_atomic_store(status, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        u32x4 raw[MT][KS];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool valid = 16 * m + li < nrows;
            const unsigned off = (t & 1) * par_bytes + rd_base + (unsigned)((valid ? 16 * m + li : 0) * 32);
#pragma unroll
            for (int i = 0; i < KS; ++i) raw[m][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off + i * rd_kstep, 0, 16);
        }
        float xg1[MT][4];
        const int t1 = t + 1 < T ? t + 1 : t;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) xg1[m][g] = xp[m][(size_t)t1 * 4 * H + g * H];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const gbf8 a = as_bf8(raw[m][i]);
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, fw[g][i], acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = (4 * q + r) * GT + li;
#pragma unroll
                for (int g = 0; g < 4; ++g) red[wave][g][m][e] = acc[g][r];
            }
        }
        gp_lds_barrier();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int e = bl * GT + jl;
            float pre[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pre[g] = xg[m][g] + (((red[0][g][m][e] + red[1][g][m][e]) + (red[2][g][m][e] + red[3][g][m][e])) + bh[g]);
            const float ig = mg_sigmoid_fast(pre[0]), fg = mg_sigmoid_fast(pre[1]), gg = lp_tanh_fast(pre[2]), og = mg_sigmoid_fast(pre[3]);
            const float cnew = fg * cprev[m] + ig * gg;
            const float hnew = og * lp_tanh_fast(cnew);
            const bool active = t < len[m];
            hprev[m] = active ? hnew : hprev[m];
            cprev[m] = active ? cnew : cprev[m];
            hb[m][bl][jl] = mg_f2bf(hprev[m]);
            res[m][0][e] = hprev[m];
            res[m][1][e] = cprev[m];
            res[m][2][e] = active ? hnew : 0.f;
            res[m][3][e] = ig;
            res[m][4][e] = fg;
            res[m][5][e] = gg;
            res[m][6][e] = og;
        }
        gp_lds_barrier();
        publish(t + 1);
        if (wave >= 2) {                                // fp32 results: waves 2 and 3 only (their stores are off wave 0's queue)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int e = (tid - 128) + 128 * half, rb = e >> 4, cj = j0 + (e & 15);
                    if (16 * m + rb < nrows) {
                        const int b = row0 + 16 * m + rb;
                        const size_t row = (size_t)b * T + t;
                        const size_t nxt = ((size_t)b * (T + 1) + t + 1) * H + cj;
                        hstate[nxt] = res[m][0][e];
                        cstate[nxt] = res[m][1][e];
                        out[row * H + cj] = res[m][2][e];
                        float* sv = saved + row * 4 * H + cj;
                        sv[0] = res[m][3][e];
                        sv[H] = res[m][4][e];
                        sv[2 * H] = res[m][5][e];
                        sv[3 * H] = res[m][6][e];
                    }
                }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) xg[m][g] = xg1[m][g];
    }
    // beyond the group's longest sequence: states frozen, outputs zero - no matmul, no hand-off
    for (int t = gmax; t < T; ++t) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (mine[m]) {
                const int b = row0 + 16 * m + bl;
                const size_t row = (size_t)b * T + t;
                const size_t nxt = ((size_t)b * (T + 1) + t + 1) * H + j;
                hstate[nxt] = hprev[m];
                cstate[nxt] = cprev[m];
                hstate_bf[nxt] = mg_f2bf(hprev[m]);
                out[row * H + j] = 0.f;
                float* sv = saved + row * 4 * H;
                sv[j] = 0.f;
                sv[H + j] = 0.f;
                sv[2 * H + j] = 0.f;
                sv[3 * H + j] = 0.f;
            }
    }
}

// Backward.  Slot s owns d h[:, 16 s .. + 16): per step t (T-1 .. 0, then t = -1 for dh0 / dc0) it needs dgates_{t+1} of the whole
// group (R x 4H bf16, the hand-off), contracts it with its 16 rows of W_hh^T (16 x 4H bf16 = 64 VGPRs per lane at H = 512),
// applies the cell derivatives and publishes its 4 x 16 columns of dgates_t.  carry_h (the part of d h_{t-1} that does not go
// through the matmul: only where the item was frozen) and carry_c stay in registers.  Flag of a slot = gmax - t once row t is
// published.  Ring: [2 (t parity)][8 groups][H / 16 slots][4 gates][R items][16 units] bf16.
template <int MT, int KS>
__global__ __launch_bounds__(256) void lstm_bwd_persist_kernel(const float* __restrict__ grad_out, const float* __restrict__ grad_hn,
                                                               const float* __restrict__ grad_cn, const float* __restrict__ cstate,
                                                               const float* __restrict__ saved, const uint16_t* __restrict__ wt_bf, int ldt,
                                                               const int64_t* __restrict__ seq_len, int B, int T, int H, int R,
                                                               float* __restrict__ dgates, uint16_t* __restrict__ dgates_bf,
                                                               float* __restrict__ dh0, float* __restrict__ dc0, unsigned* sync,
                                                               uint16_t* ring, int force_sc1) {
    __shared__ float red[4][MT][GT * GT];
    __shared__ __attribute__((aligned(16))) uint16_t pub[4][MT * GT][GT];
    __shared__ float res[MT][4][GT * GT];
    __shared__ int s_abort, s_xcd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    const int gbase = wave * (G / 4) + 8 * q;           // = wave * H + 8 q: wave w contracts over gate w
    gbf8 fb[KS];
    unsigned rd_off[KS];
    {
        const uint16_t* wp = wt_bf + (size_t)(j0 + li) * ldt + gbase;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            fb[i] = *reinterpret_cast<const gbf8*>(wp + 32 * i);
            const int g = gbase + 32 * i, gate = g / H, col = g - gate * H;
            rd_off[i] = (unsigned)((((group * n_slots + (col >> 4)) * 4 + gate) * R) * 32 + 16 * (q & 1));
        }
    }
    const unsigned par_bytes = (unsigned)(GP_GROUPS * n_slots * R * 128);
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, (int)(2 * par_bytes), 0x00020000);
    const unsigned wr_base = (unsigned)(((group * n_slots + slot) * R) * 128);

    const int bl = tid >> 4, jl = tid & 15;
    const int j = j0 + jl;
    float carry_h[MT], carry_c[MT];
    int len[MT];
    bool mine[MT];
    const float *p_sv[MT], *p_c[MT], *p_g[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        mine[m] = 16 * m + bl < nrows;
        const int b = row0 + (mine[m] ? 16 * m + bl : 0);
        carry_h[m] = grad_hn ? grad_hn[(size_t)b * H + j] : 0.f;
        carry_c[m] = grad_cn ? grad_cn[(size_t)b * H + j] : 0.f;
        len[m] = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
        p_sv[m] = saved + (size_t)b * T * 4 * H + j;
        p_c[m] = cstate + (size_t)b * (T + 1) * H + j;
        p_g[m] = grad_out + (size_t)b * T * H + j;
    }
    for (int t = T - 1; t >= gmax; --t) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (mine[m]) {
                const size_t row = (size_t)(row0 + 16 * m + bl) * T + t;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (dgates) dgates[row * G + g * H + j] = 0.f;
                    dgates_bf[row * G + g * H + j] = 0;
                }
            }
    }
    constexpr int PIECES = (8 * GT * MT + 63) / 64;
    bool pc_ok[PIECES];
    int pc_lds[PIECES];
    size_t pc_shadow[PIECES];
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
        const int pc = lane + 64 * k, gate = pc / (2 * R), rem = pc - gate * 2 * R, rrow = rem >> 1, half = rem & 1;
        pc_ok[k] = gate < 4 && rrow < nrows;
        pc_lds[k] = (gate * MT * GT + rrow) * GT + 8 * half;
        pc_shadow[k] = (size_t)(row0 + rrow) * T * G + gate * H + j0 + 8 * half;
    }
    // cell operands of step gmax - 1 (c_new of a step is c_prev of the step after it: carried over in a register)
    float s_g4[MT][4], c_prev[MT], c_new[MT], gout[MT];
    {
        const int t0 = gmax > 0 ? gmax - 1 : 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) s_g4[m][g] = p_sv[m][(size_t)t0 * 4 * H + g * H];
            c_prev[m] = p_c[m][(size_t)t0 * H];
            c_new[m] = p_c[m][(size_t)(t0 + 1) * H];
            gout[m] = p_g[m][(size_t)t0 * H];
        }
    }
    __syncthreads();

    for (int t = gmax - 1; t >= -1; --t) {
        const bool need_mm = t + 1 < gmax;
        u32x4 raw[MT][KS];
        if (need_mm) {
            if (wave == 0 && !gp_wait_flags(flags, n_slots, (unsigned)(gmax - t - 1), lane)) s_abort = 1;
            gp_lds_barrier();
            if (s_abort) {
                if (tid == 0) __hip_atomic_store(status, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool valid = 16 * m + li < nrows;
                const unsigned off = ((t + 1) & 1) * par_bytes + (unsigned)((valid ? 16 * m + li : 0) * 32);
#pragma unroll
                for (int i = 0; i < KS; ++i) raw[m][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off + rd_off[i], 0, 16);
            }
        }
        float s_g41[MT][4], c_prev1[MT], gout1[MT];
        {
            const int t1 = t > 0 ? t - 1 : 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int g = 0; g < 4; ++g) s_g41[m][g] = p_sv[m][(size_t)t1 * 4 * H + g * H];
                c_prev1[m] = p_c[m][(size_t)t1 * H];
                gout1[m] = p_g[m][(size_t)t1 * H];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (need_mm) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 acc4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int i = 0; i < KS; ++i)
                    acc4[i % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(raw[m][i]), fb[i], acc4[i % 4], 0, 0, 0);
                const f32x4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][m][(4 * q + r) * GT + li] = acc[r];
            }
            gp_lds_barrier();
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int e = bl * GT + jl;
            const float dh_state = need_mm ? carry_h[m] + ((red[0][m][e] + red[1][m][e]) + (red[2][m][e] + red[3][m][e])) : carry_h[m];
            float di = 0.f, df = 0.f, dg = 0.f, d_o = 0.f, ch = dh_state, cc = carry_c[m];
            if (t >= 0 && t < len[m]) {
                const float s_i = s_g4[m][0], s_f = s_g4[m][1], s_g = s_g4[m][2], s_o = s_g4[m][3];
                const float dh = dh_state + gout[m];
                const float tc = lp_tanh_fast(c_new[m]);
                const float dc = carry_c[m] + dh * s_o * (1.f - tc * tc);
                di = dc * s_g * s_i * (1.f - s_i);
                df = dc * c_prev[m] * s_f * (1.f - s_f);
                dg = dc * s_i * (1.f - s_g * s_g);
                d_o = dh * tc * s_o * (1.f - s_o);
                ch = 0.f;                 // all of dh_{t-1} comes through the matmul with the gates of this step
                cc = dc * s_f;
            }
            carry_h[m] = ch;
            carry_c[m] = cc;
            res[m][0][e] = di;
            res[m][1][e] = df;
            res[m][2][e] = dg;
            res[m][3][e] = d_o;
            pub[0][16 * m + bl][jl] = mg_f2bf(di);
            pub[1][16 * m + bl][jl] = mg_f2bf(df);
            pub[2][16 * m + bl][jl] = mg_f2bf(dg);
            pub[3][16 * m + bl][jl] = mg_f2bf(d_o);
        }
        if (t < 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (mine[m]) {
                    dh0[(size_t)(row0 + 16 * m + bl) * H + j] = carry_h[m];
                    dc0[(size_t)(row0 + 16 * m + bl) * H + j] = carry_c[m];
                }
            break;
        }
        gp_lds_barrier();
        if (wave <= 1) {
#pragma unroll
            for (int k = 0; k < PIECES; ++k) {
                if (pc_ok[k]) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(&pub[0][0][0] + pc_lds[k]);
                    if (wave == 0) {
                        const unsigned off = (t & 1) * par_bytes + wr_base + (unsigned)((lane + 64 * k) * 16);
                        if (one_xcd)
                            __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 0);
                        else
                            __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 16);
                    } else {
                        *reinterpret_cast<u32x4*>(dgates_bf + pc_shadow[k] + (size_t)t * G) = v;
                    }
                }
            }
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) gp_store_flag(flags + slot, (unsigned)(gmax - t), one_xcd);
            }
        }
        if (wave >= 2 && dgates) {                      // optional fp32 copy of the gate gradients
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int e = (tid - 128) + 128 * half, rb = e >> 4, cj = j0 + (e & 15);
                    if (16 * m + rb < nrows) {
                        float* dgp = dgates + ((size_t)(row0 + 16 * m + rb) * T + t) * G + cj;
                        dgp[0] = res[m][0][e];
                        dgp[H] = res[m][1][e];
                        dgp[2 * H] = res[m][2][e];
                        dgp[3 * H] = res[m][3][e];
                    }
                }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) s_g4[m][g] = s_g41[m][g];
            c_new[m] = c_prev[m];
            c_prev[m] = c_prev1[m];
            gout[m] = gout1[m];
        }
    }
}

// =====================================================================================================================
// The whole LSTM STACK forward in one launch: a wavefront over (layer, time).  Layer l's step t needs its own h_{t-1} and the
// layer below's h_t, so L stacked layers take T + L - 1 dependent steps instead of L T when every layer has its own
// workgroups: workgroup = (layer, group, slot), G groups of R = ceil(B / G) items with L G H / 16 <= 512 workgroups (two per
// CU, <= 256 VGPRs each).  Layers above the first compute their input projection inside the step (their W_ih slice is
// resident next to the W_hh slice: 128 VGPRs per lane at H = 512) from tiles the layer below publishes.
// A (layer, group) fills half an XCD (32 of its 64 workgroup slots), so neighbouring layers live on different XCDs: every
// layer therefore publishes its tile twice -
//   ring A  2 epochs, for its OWN next step: the single-layer protocol (plain stores + flag A when the group shares an XCD)
//   ring X  4 epochs, for the layer ABOVE: always write-through (sc1) stores + sc1 flag X, written by wave 1, off wave 0's
//           publish -> poll path; the extra fabric latency is hidden by the pipeline as long as the ring is deep enough
// and three conditions gate step t of layer l (one poll: lanes 0-31 own A flags, 32-63 the lower layer's X flags, a second
// load for the upper layer's X flags):
//   own   A flags >= t + 1         h_t of this layer is published (epoch e = state h_e; flag = e + 1)
//   lower X flags >= t + 2         h_{t+1} of the layer below (its output of step t) is published
//   upper X flags >= t - 2         the layer above has finished its step t - 4: the epoch t - 3 that this step's ring-X
//                                  publish (epoch t + 1, same slot) overwrites is no longer read
// Layer 0 takes xproj = x W_ih^T + b_ih from memory (one GEMM before the launch), as the single-layer kernel does.
// =====================================================================================================================
MG_STAMP_DECL(g_stamps_lps);

struct LstmPStack {
    mg_lstm_pstack_layer l[MG_LSTM_MAX_LAYERS];
};

// workspace of the stack kernel: [512 words unused | status word + 3 pad (same place as in the single-layer kernels)]
// [flags A: up to 64 (layer, group) ids x 32 words] [flags X: 64 x 32] [XCC ids: 64 x 32] [rings A: L x 2 epochs] [rings X: L x 4]
#define LPS_MAX_IDS 64                                       // L G <= 512 / (H / 16) and H >= 128
#define LPS_XDEPTH 4
#define LPS_FLAGA_WORD (GP_FLAG_WORDS + 4)
#define LPS_FLAGX_WORD (LPS_FLAGA_WORD + LPS_MAX_IDS * GP_SLOTS)
#define LPS_XCC_WORD (LPS_FLAGX_WORD + LPS_MAX_IDS * GP_SLOTS)
#define LPS_RING_OFFSET ((size_t)(LPS_XCC_WORD + LPS_MAX_IDS * GP_SLOTS) * sizeof(unsigned))

// wave 0: the three conditions of a step.  The neighbour layers' X flags are written through (sc1), so every poll of them is a
// fabric round trip; they are normally far ahead, so each lane keeps the last value it saw (seen_lo / seen_up) and they are
// polled again only when that no longer suffices.  Own A flags: the single-layer poll loop.
__device__ __forceinline__ bool lps_wait(gu32* own, unsigned need_own, gu32* lower, int need_lower, unsigned& seen_lo, gu32* upper,
                                         int need_upper, unsigned& seen_up, int n_slots, int lane) {
    unsigned spins = 0;
    if (lower) {
        while (!__all(lane >= n_slots || (int)seen_lo >= need_lower)) {
            if (lane < n_slots) seen_lo = __hip_atomic_load(lower + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (++spins > GP_SPIN_LIMIT) return false;
        }
    }
    if (upper) {
        while (!__all(lane >= n_slots || (int)seen_up >= need_upper)) {
            if (lane < n_slots) seen_up = __hip_atomic_load(upper + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (++spins > GP_SPIN_LIMIT) return false;
        }
    }
    for (;; ++spins) {
        const unsigned f = lane < n_slots ? __hip_atomic_load(own + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : need_own;
        if (__all(f >= need_own)) return true;
        if (spins > GP_SPIN_LIMIT) return false;
        __builtin_amdgcn_s_sleep(1);
    }
}

// UT = hidden-unit tiles of 16 per workgroup.  UT = 1 (default): two workgroups per CU (256 VGPRs each), each taking in the whole state of
// its group and of the group below - 128 KB per CU and step.  UT = 2 (MG_TUNE_LSTM_BWD_STACK bit 1): ONE workgroup per CU with up to 512
// VGPRs owns 32 units - both unit tiles' weight slices resident (256 registers), every tile read once per CU and multiplied against
// both: half the intake, same bits - and 8 % slower on the shipped stack (see mg_lstm_pstack_fwd_bf16).
template <int MT, int KS, int UT>
__global__ __launch_bounds__(256, UT == 1 ? 2 : 1) void lstm_stack_fwd_persist_kernel(LstmPStack a, const int64_t* __restrict__ seq_len, int B, int T,
                                                                                      int H, int L, int G, int R, unsigned* sync, uint16_t* rings,
                                                                                      int force_sc1) {
    __shared__ float red[4][4][UT][MT][GT * GT];
    __shared__ __attribute__((aligned(16))) uint16_t hb[UT][MT][GT][GT];
    __shared__ __attribute__((aligned(16))) float res[UT][MT][6][GT * GT];   // c, out, i, f, g, o of the step on their way to waves 2 and 3
    __shared__ int s_abort, s_xcd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int n_ids = L * G;
    const int id = blockIdx.x % n_ids, slot = blockIdx.x / n_ids;
    const int layer = id / G, group = id - layer * G;
    const int n16 = H / GT, n_slots = n16 / UT;
    const int row0 = group * R;
    const int nrows = min(R, B - row0);
    if (slot >= n_slots || nrows <= 0) return;
    const mg_lstm_pstack_layer& P = a.l[layer];
    const int j0 = slot * GT * UT;
    gu32* flags_a = (gu32*)sync + LPS_FLAGA_WORD + id * GP_SLOTS;
    gu32* flags_x = (gu32*)sync + LPS_FLAGX_WORD + id * GP_SLOTS;
    gu32* flags_lo = layer > 0 ? flags_x - G * GP_SLOTS : (gu32*)nullptr;
    gu32* flags_up = layer + 1 < L ? flags_x + G * GP_SLOTS : (gu32*)nullptr;
    gu32* status = (gu32*)sync + GP_FLAG_WORDS;
    if (tid == 0) s_abort = 0;
    const int one_xcd = (force_sc1 & 1) ? 0 : gp_group_on_one_xcd((gu32*)sync + LPS_XCC_WORD + id * GP_SLOTS, slot, n_slots, tid, &s_xcd);
    if (one_xcd < 0) {
        if (tid == 0) __hip_atomic_store(status, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    int gmax = 0;
    for (int r = 0; r < nrows; ++r) {
        const int64_t n = seq_len ? seq_len[row0 + r] : (int64_t)T;
        gmax = max(gmax, (int)(n < T ? n : T));
    }
    const int kbase = wave * (H / 4) + 8 * q;
    // fwi: the W_ih slice of a layer above the first.  Layer 0 has no such slice and keeps its xproj values of the current
    // step in the same registers (XG below) - the kernel sits at the 256-VGPR limit of two workgroups per CU.
    gbf8 fwh[UT][4][KS];
    u32x4 fwi[UT][4][KS];
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint16_t* wp = P.w_hh_bf + ((size_t)g * H + j0 + GT * u + li) * P.ldwh + kbase;
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                fwh[u][g][i] = *reinterpret_cast<const gbf8*>(wp + 32 * i);
                fwi[u][g][i] = u32x4{0u, 0u, 0u, 0u};
            }
            if (layer > 0) {
                const uint16_t* wi = P.w_ih_bf + ((size_t)g * H + j0 + GT * u + li) * P.ldwi + kbase;
#pragma unroll
                for (int i = 0; i < KS; ++i) fwi[u][g][i] = *reinterpret_cast<const u32x4*>(wi + 32 * i);
            }
        }
#define XG(u, m, g) fwi[u][(m) / KS][(m) % KS][g]              /* layer 0: xproj of this step, unit tile u, item tile m, gate g */
    // rings: A [layer][2 epochs][G groups][H / 16 slots][R items][16 units] bf16, then X [layer][4 epochs][...]
    const unsigned par_bytes = (unsigned)(G * n16 * R * 32);
    const unsigned x_base = (unsigned)L * 2 * par_bytes;
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)rings, 0, (int)((2 + LPS_XDEPTH) * par_bytes * L), 0x00020000);
    const unsigned ring_own = (unsigned)layer * 2 * par_bytes;
    const unsigned ring_x = x_base + (unsigned)layer * LPS_XDEPTH * par_bytes;
    const unsigned ring_lo = x_base + (unsigned)(layer > 0 ? layer - 1 : 0) * LPS_XDEPTH * par_bytes;
    const unsigned rd_base = (unsigned)(((group * n16 + (kbase >> 4)) * R) * 32 + 16 * (q & 1));
    const unsigned rd_kstep = (unsigned)(2 * R * 32);
    const unsigned wr_base = (unsigned)(((group * n16 + slot * UT) * R) * 32);       // unit tile u: + u R 32

    const int bl = tid >> 4, jl = tid & 15;
    const int j = j0 + jl;                            // the thread's unit of tile 0; tile u: + 16 u
    float bsum[UT][4];
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) bsum[u][g] = P.b_hh[g * H + j + GT * u] + (layer > 0 ? P.b_ih[g * H + j + GT * u] : 0.f);
    float hprev[UT][MT], cprev[UT][MT];
    int len[MT];
    bool mine[MT];
    const float* xp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        mine[m] = 16 * m + bl < nrows;
        const int b = row0 + (mine[m] ? 16 * m + bl : 0);
        len[m] = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
        xp[m] = (layer == 0 ? P.xproj : P.b_hh) + (layer == 0 ? (size_t)b * T * 4 * H + j : 0);
#pragma unroll
        for (int u = 0; u < UT; ++u) {
            hprev[u][m] = P.hstate[((size_t)b * (T + 1)) * H + j + GT * u];
            cprev[u][m] = P.cstate[((size_t)b * (T + 1)) * H + j + GT * u];
            hb[u][m][bl][jl] = mg_f2bf(hprev[u][m]);
            if (layer == 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) XG(u, m, g) = __float_as_uint(xp[m][g * H + GT * u]);
            }
        }
    }
    __syncthreads();

    // wave 0: ring A + flag A (this layer's own recurrence); wave 1: ring X + flag X (for the layer above and as this layer's
    // progress report to the layer below) and the bf16 shadow of the state
    auto publish = [&](int e) {
        if (wave <= 1) {
            if (lane < 2 * 16 * MT) {
                const int rrow = lane >> 1, half = lane & 1;
                if (rrow < nrows) {
#pragma unroll
                    for (int u = 0; u < UT; ++u) {
                        const u32x4 v = *reinterpret_cast<const u32x4*>(&hb[u][rrow >> 4][rrow & 15][8 * half]);
                        const unsigned tile = wr_base + (unsigned)(u * R * 32 + rrow * 32 + half * 16);
                        if (wave == 0) {
                            const unsigned off = ring_own + (e & 1) * par_bytes + tile;
                            if (one_xcd)
                                __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 0);
                            else
                                __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 16);
                        } else {
                            if (layer + 1 < L)
                                __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, ring_x + (e & (LPS_XDEPTH - 1)) * par_bytes + tile, 0, 16);
                            if (e > 0)
                                *reinterpret_cast<u32x4*>(P.hstate_bf + ((size_t)(row0 + rrow) * (T + 1) + e) * H + j0 + GT * u + 8 * half) = v;
                        }
                    }
                }
            }
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) gp_store_flag(flags_a + slot, (unsigned)(e + 1), one_xcd);
            }
        }
    };
    // wave 1 raises flag X for an epoch only once its write-through stores are done; it does not wait for them at the publish
    // (it would hold up the workgroup's next barrier by a fabric round trip) but in the next step, behind the MFMAs
    auto raise_x = [&](int e) {
        if (wave == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(flags_x + slot, (unsigned)(e + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    unsigned seen_lo = 0, seen_up = 0;
    publish(0);
    raise_x(0);
#ifdef MG_STAMPS
    unsigned long long ta = 0, tb = 0, ts0 = 0, ts1 = 0, tr0 = 0, tr1 = 0, sum_poll = 0, sum_load = 0, sum_mm = 0, sum_cell = 0, sum_pub = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif

    for (int t = 0; t < gmax; ++t) {
        float xg1[UT][MT][4];                        // layer 0: the next step's xproj values, in flight until the end of the step
        MG_STAMP(ta);
        if (wave == 0 && !lps_wait(flags_a, (unsigned)(t + 1), flags_lo, t + 2, seen_lo, flags_up, t - (LPS_XDEPTH - 2), seen_up, n_slots, lane))
            s_abort = 1;
        gp_lds_barrier();
        if (s_abort) {
            if (tid == 0) __hip_atomic_store(status, 6u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_poll, tb, ta);
        // hand-off tiles: this layer's h_t (ring A) and the lower layer's h_{t+1} (its ring X).  The second M tile's loads are
        // issued between the first tile's MFMAs, each into the register its k-step has just freed (two tiles in flight at once
        // do not fit in 256 VGPRs next to the 128 of the two weight slices); their latency hides behind the first tile's MFMAs.
        auto tile_off = [&](int m) {
            const bool valid = 16 * m + li < nrows;
            return rd_base + (unsigned)((valid ? 16 * m + li : 0) * 32);
        };
        const unsigned own0 = ring_own + (t & 1) * par_bytes, low0 = ring_lo + ((t + 1) & (LPS_XDEPTH - 1)) * par_bytes;
        u32x4 raw[KS], rawx[KS];
        {
            const unsigned ro = tile_off(0);
#pragma unroll
            for (int i = 0; i < KS; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, own0 + ro + i * rd_kstep, 0, 16);
            if (layer > 0) {
#pragma unroll
                for (int i = 0; i < KS; ++i) rawx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, low0 + ro + i * rd_kstep, 0, 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef MG_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_load, ta, tb);
#endif
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const unsigned rn = m + 1 < MT ? tile_off(m + 1) : 0u;
            f32x4 acc[UT][4];
#pragma unroll
            for (int u = 0; u < UT; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[u][g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const gbf8 av = as_bf8(raw[i]);
#pragma unroll
                for (int u = 0; u < UT; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[u][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, fwh[u][g][i], acc[u][g], 0, 0, 0);
                if (m + 1 < MT) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, own0 + rn + i * rd_kstep, 0, 16);
            }
            if (layer > 0) {
#pragma unroll
                for (int i = 0; i < KS; ++i) {
                    const gbf8 av = as_bf8(rawx[i]);
#pragma unroll
                    for (int u = 0; u < UT; ++u)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            acc[u][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, as_bf8(fwi[u][g][i]), acc[u][g], 0, 0, 0);
                    if (m + 1 < MT) rawx[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, low0 + rn + i * rd_kstep, 0, 16);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = (4 * q + r) * GT + li;
#pragma unroll
                for (int u = 0; u < UT; ++u)
#pragma unroll
                    for (int g = 0; g < 4; ++g) red[wave][g][u][m][e] = acc[u][g][r];
            }
        }
        if (t > 0) raise_x(t);                        // epoch t's write-through stores (issued a step ago) have long landed
        if (layer == 0) {
            // the next step's xproj values (first touch: HBM latency): requested only now, behind every hand-off load of this
            // step - memory returns in order, and the second tile's loads must not wait for these
            const int t1 = t + 1 < T ? t + 1 : t;
#pragma unroll
            for (int u = 0; u < UT; ++u)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) xg1[u][m][g] = xp[m][(size_t)t1 * 4 * H + g * H + GT * u];
        }
        gp_lds_barrier();
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_mm, tb, ta);
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int e = bl * GT + jl;
                float pre[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    pre[g] = (layer == 0 ? __uint_as_float(XG(u, m, g)) : 0.f) +
                             (((red[0][g][u][m][e] + red[1][g][u][m][e]) + (red[2][g][u][m][e] + red[3][g][u][m][e])) + bsum[u][g]);
                const float ig = mg_sigmoid_fast(pre[0]), fg = mg_sigmoid_fast(pre[1]), gg = lp_tanh_fast(pre[2]), og = mg_sigmoid_fast(pre[3]);
                const float cnew = fg * cprev[u][m] + ig * gg;
                const float hnew = og * lp_tanh_fast(cnew);
                const bool active = t < len[m];
                hprev[u][m] = active ? hnew : hprev[u][m];
                cprev[u][m] = active ? cnew : cprev[u][m];
                hb[u][m][bl][jl] = mg_f2bf(hprev[u][m]);
                res[u][m][0][e] = cprev[u][m];
                res[u][m][1][e] = active ? hnew : 0.f;
                res[u][m][2][e] = ig;
                res[u][m][3][e] = fg;
                res[u][m][4][e] = gg;
                res[u][m][5][e] = og;
            }
        gp_lds_barrier();
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_cell, ta, tb);
        publish(t + 1);
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_pub, tb, ta);
        if (wave >= 2) {
            // fp32 results needed after the launch: c_t and the gate values of every layer (BPTT), the outputs of the top layer
            // only (the layers in between are read through the rings; their fp32 states are not kept - h_n is written at the end).
            // 16-byte stores: job = (array, item tile), 64 lanes per job = 16 items x 4 column quads; wave 2 and wave 3 split the jobs.
            const int n_arrays = layer + 1 == L ? 6 : 5;              // array 1 = out: top layer only
            const int rb = lane >> 2, c4 = 4 * (lane & 3);
            for (int job = wave - 2; job < n_arrays * MT * UT; job += 2) {
                const int u = job % UT, jm = job / UT, m = jm % MT, k0 = jm / MT, k = (k0 >= 1 && n_arrays == 5) ? k0 + 1 : k0;
                if (16 * m + rb < nrows) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&res[u][m][k][rb * GT + c4]);
                    const int b = row0 + 16 * m + rb;
                    float* dst = k == 0   ? P.cstate + ((size_t)b * (T + 1) + t + 1) * H
                                 : k == 1 ? P.out + ((size_t)b * T + t) * H
                                          : P.saved + ((size_t)b * T + t) * 4 * H + (size_t)(k - 2) * H;
                    *reinterpret_cast<f32x4*>(dst + j0 + GT * u + c4) = v;
                }
            }
        }
        if (layer == 0) {
#pragma unroll
            for (int u = 0; u < UT; ++u)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) XG(u, m, g) = __float_as_uint(xg1[u][m][g]);
        }
    }
#undef XG
#ifdef MG_STAMPS
    MG_STAMP(ts1);
    MG_STAMP_REAL(tr1);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 2, tr0);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 3, tr1);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 4, sum_poll);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 5, sum_load);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 6, sum_mm);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 7, sum_cell);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 8, sum_pub);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 9, (unsigned long long)gmax);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 10, (unsigned long long)(layer * 1000 + one_xcd));
#endif
    if (gmax > 0) raise_x(gmax);
    for (int t = gmax; t < T; ++t) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (mine[m]) {
                const int b = row0 + 16 * m + bl;
                const size_t row = (size_t)b * T + t;
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    const int ju = j + GT * u;
                    const size_t nxt = ((size_t)b * (T + 1) + t + 1) * H + ju;
                    P.cstate[nxt] = cprev[u][m];
                    P.hstate_bf[nxt] = mg_f2bf(hprev[u][m]);
                    if (layer + 1 == L) P.out[row * H + ju] = 0.f;
                    float* sv = P.saved + row * 4 * H;
                    sv[ju] = 0.f;
                    sv[H + ju] = 0.f;
                    sv[2 * H + ju] = 0.f;
                    sv[3 * H + ju] = 0.f;
                }
            }
    }
    // h_n: the only fp32 state row kept (slot T of hstate)
#pragma unroll
    for (int m = 0; m < MT; ++m)
        if (mine[m]) {
#pragma unroll
            for (int u = 0; u < UT; ++u) P.hstate[((size_t)(row0 + 16 * m + bl) * (T + 1) + T) * H + j + GT * u] = hprev[u][m];
        }
}

// =====================================================================================================================
// The whole LSTM STACK backward in one launch: the forward's wavefront run down in time and down through the layers.  Layer l's
// step t needs its own gate gradients of step t + 1 (through W_hh, as in lstm_bwd_persist_kernel) and, instead of a grad_out row
// from memory, the gate gradients of the layer ABOVE at step t through that layer's W_ih: d out^l_t = dgates^{l+1}_t W_ih^{l+1}.
// So L layers take T + L dependent steps instead of L (T + 1) + (L - 1) input-gradient GEMMs between the launches.
// Workgroup = (layer, group, slot); a slot owns 16 UT hidden units (UT = 2: one 256-thread workgroup per CU with up to 512
// VGPRs, its rows of W_hh^T and of the upper W_ih^T resident in 256 of them; UT = 1: two workgroups per CU as in the forward).
// Per step a workgroup takes in two hand-off tiles of R x 4H bf16 (its own and the upper layer's gate gradients): 256 KB at
// R = 32, H = 512 - four times the forward's - and that intake (66-73 GB/s per CU from L2), not the latency chain, sets the step
// time; UT = 2 halves it per CU against UT = 1 (one workgroup per CU reads each tile once for 32 units).
// Rings (bf16, [G][H / 16 unit tiles][4 gates][R items][16 units] per epoch):
//   ring A  2 epochs (t parity), this layer's own next step; plain stores + flag A when the (layer, group) shares an XCD
//   ring X  4 epochs (t mod 4), for the layer BELOW: always write-through (sc1), written by wave 1, flag X raised a step later
// Conditions of step t (wave 0, one poll each; the neighbours' flags are cached per lane as in the forward):
//   own   A flags >= gmax - t - 1      dgates_{t+1} of this layer is published (flag = steps published = gmax - t after step t)
//   upper X flags >= gmax - t          the layer above has published its step t
//   lower X flags >= gmax - t - 4      the layer below has finished step t + 4, whose ring-X epoch this step's publish overwrites
// (the lowest layer stores no ring X and raises its X flags as a progress report only).
// =====================================================================================================================
// VW-wide fp32 vectors of the backward stack kernel's cell phase (VW = 1, 2 or 4 consecutive hidden units): one load / store each
template <int VW>
struct lp_vec {
    float v[VW] = {};
};
template <int VW>
__device__ __forceinline__ lp_vec<VW> lp_ld(const float* p) {
    lp_vec<VW> r;
    if constexpr (VW == 4) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p);
        r.v[0] = x[0], r.v[1] = x[1], r.v[2] = x[2], r.v[3] = x[3];
    } else if constexpr (VW == 2) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t x = *reinterpret_cast<const f32x2_t*>(p);
        r.v[0] = x[0], r.v[1] = x[1];
    } else {
        r.v[0] = *p;
    }
    return r;
}
template <int VW>
__device__ __forceinline__ void lp_st(float* p, const lp_vec<VW>& a) {
    if constexpr (VW == 4) {
        *reinterpret_cast<f32x4*>(p) = f32x4{a.v[0], a.v[1], a.v[2], a.v[3]};
    } else if constexpr (VW == 2) {
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        *reinterpret_cast<f32x2_t*>(p) = f32x2_t{a.v[0], a.v[1]};
    } else {
        *p = a.v[0];
    }
}
template <int VW>
__device__ __forceinline__ void lp_st_bf(uint16_t* p, const lp_vec<VW>& a) {
    if constexpr (VW == 4) {
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
        *reinterpret_cast<u32x2_t*>(p) = u32x2_t{(unsigned)mg_f2bf(a.v[0]) | ((unsigned)mg_f2bf(a.v[1]) << 16),
                                                 (unsigned)mg_f2bf(a.v[2]) | ((unsigned)mg_f2bf(a.v[3]) << 16)};
    } else if constexpr (VW == 2) {
        *reinterpret_cast<unsigned*>(p) = (unsigned)mg_f2bf(a.v[0]) | ((unsigned)mg_f2bf(a.v[1]) << 16);
    } else {
        *p = mg_f2bf(a.v[0]);
    }
}

// The cell derivatives of one ACTIVE step with contraction pinned off: the UT = 1 and UT = 2 instantiations must produce the same
// bits (tests compare them), and left to itself hipcc fuses these products and sums differently from one instantiation to the next.
struct lp_cell_grad {
    float di, df, dg, d_o, cc;
};
__device__ __forceinline__ lp_cell_grad lp_cell_bwd_fast(float dh_state, float dc_state, float g_in, float s_i, float s_f, float s_g, float s_o,
                                                         float c_prev, float c_new) {
#pragma clang fp contract(off)
    lp_cell_grad r;
    const float dh = dh_state + g_in;
    const float tc = 2.f * mg_sigmoid_fast(2.f * c_new) - 1.f;
    const float dc = dc_state + dh * s_o * (1.f - tc * tc);
    r.di = dc * s_g * s_i * (1.f - s_i);
    r.df = dc * c_prev * s_f * (1.f - s_f);
    r.dg = dc * s_i * (1.f - s_g * s_g);
    r.d_o = dh * tc * s_o * (1.f - s_o);
    r.cc = dc * s_f;
    return r;
}

struct LstmPStackBwd {
    mg_lstm_pstack_bwd_layer l[MG_LSTM_MAX_LAYERS];
};

template <int MT, int KS, int UT>
__global__ __launch_bounds__(256, UT == 1 ? 2 : 1) void lstm_stack_bwd_persist_kernel(LstmPStackBwd a, const int64_t* __restrict__ seq_len, int B,
                                                                                      int T, int H, int L, int G, int R, unsigned* sync,
                                                                                      uint16_t* rings, int force_sc1) {
    __shared__ float red[4][2][UT][MT][GT * GT];
    __shared__ __attribute__((aligned(16))) uint16_t pub[UT][4][MT * GT][GT];
    __shared__ __attribute__((aligned(16))) float res[UT][MT][4][GT * GT];
    __shared__ int s_abort, s_xcd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int n_ids = L * G;
    const int id = blockIdx.x % n_ids, slot = blockIdx.x / n_ids;
    const int layer = id / G, group = id - layer * G;
    const int n16 = H / GT, n_slots = n16 / UT;
    const int row0 = group * R;
    const int nrows = min(R, B - row0);
    if (slot >= n_slots || nrows <= 0) return;
    const mg_lstm_pstack_bwd_layer& P = a.l[layer];
    const bool top = layer + 1 == L;
    const int j0 = slot * GT * UT;
    const int G4 = 4 * H;
    gu32* flags_a = (gu32*)sync + LPS_FLAGA_WORD + id * GP_SLOTS;
    gu32* flags_x = (gu32*)sync + LPS_FLAGX_WORD + id * GP_SLOTS;
    gu32* flags_up = !top ? flags_x + G * GP_SLOTS : (gu32*)nullptr;
    gu32* flags_lo = layer > 0 ? flags_x - G * GP_SLOTS : (gu32*)nullptr;
    gu32* status = (gu32*)sync + GP_FLAG_WORDS;
    if (tid == 0) s_abort = 0;
    const int one_xcd = (force_sc1 & 1) ? 0 : gp_group_on_one_xcd((gu32*)sync + LPS_XCC_WORD + id * GP_SLOTS, slot, n_slots, tid, &s_xcd);
    if (one_xcd < 0) {
        if (tid == 0) __hip_atomic_store(status, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    int gmax = 0;
    for (int r = 0; r < nrows; ++r) {
        const int64_t n = seq_len ? seq_len[row0 + r] : (int64_t)T;
        gmax = max(gmax, (int)(n < T ? n : T));
    }
    // wave w contracts over gate w: columns [w H, (w + 1) H) of the gate-gradient rows, 32 per MFMA, lane (li, q) holds 8 of them
    const int gbase = wave * H + 8 * q;
    gbf8 fhh[UT][KS], fih[UT][KS];
#pragma unroll
    for (int u = 0; u < UT; ++u) {
        const uint16_t* wp = P.w_hh_t_bf + (size_t)(j0 + GT * u + li) * P.ldt + gbase;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            fhh[u][i] = *reinterpret_cast<const gbf8*>(wp + 32 * i);
            fih[u][i] = fhh[u][i];
        }
        if (!top) {
            const uint16_t* wq = P.w_ih_up_t_bf + (size_t)(j0 + GT * u + li) * P.ldt_up + gbase;
#pragma unroll
            for (int i = 0; i < KS; ++i) fih[u][i] = *reinterpret_cast<const gbf8*>(wq + 32 * i);
        }
    }
    const unsigned par_bytes = (unsigned)(G * n16 * R * 128);
    const unsigned x_base = (unsigned)L * 2 * par_bytes;
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)rings, 0, (int)((2 + LPS_XDEPTH) * par_bytes * L), 0x00020000);
    const unsigned ring_own = (unsigned)layer * 2 * par_bytes;
    const unsigned ring_x = x_base + (unsigned)layer * LPS_XDEPTH * par_bytes;
    const unsigned ring_up = x_base + (unsigned)(top ? layer : layer + 1) * LPS_XDEPTH * par_bytes;
    const unsigned rd_base = (unsigned)((((group * n16 + (q >> 1)) * 4 + wave) * R) * 32 + 16 * (q & 1));
    const unsigned rd_kstep = (unsigned)(2 * 4 * R * 32);
    const unsigned wr_base = (unsigned)(((group * n16 + slot * UT) * R) * 128);

    // Cell phase: a thread owns VW = MT UT consecutive hidden units of ONE item row of ONE (item tile, unit tile) pair, so that every
    // LDS and memory access of the phase is one VW-wide vector (VW = 4: b128 reads of the partial sums, 8-byte bf16 publishes).
    constexpr int VW = MT * UT;                       // 1, 2 or 4
    constexpr int CPR = GT / VW;                      // threads per row of a 16-unit tile
    const int ctile = tid / (256 / VW), cwithin = tid - ctile * (256 / VW);
    const int cm = ctile / UT, cu = ctile - cm * UT;  // item tile, unit tile
    const int crow = cwithin / CPR, cc0 = (cwithin - crow * CPR) * VW;
    const int ce = crow * GT + cc0;                   // element offset inside a 16 x 16 tile
    const bool mine = 16 * cm + crow < nrows;
    const int cb = row0 + (mine ? 16 * cm + crow : 0);
    const int cj = j0 + GT * cu + cc0;                // first hidden unit of the thread
    const int len = seq_len ? (int)min((int64_t)T, seq_len[cb]) : T;
    const float* p_sv = P.saved + (size_t)cb * T * 4 * H + cj;
    const float* p_c = P.cstate + (size_t)cb * (T + 1) * H + cj;
    const float* p_g = (top && P.grad_out) ? P.grad_out + (size_t)cb * T * H + cj : (const float*)nullptr;
    lp_vec<VW> carry_h = P.grad_hn ? lp_ld<VW>(P.grad_hn + (size_t)cb * H + cj) : lp_vec<VW>{};
    lp_vec<VW> carry_c = P.grad_cn ? lp_ld<VW>(P.grad_cn + (size_t)cb * H + cj) : lp_vec<VW>{};
    if (mine)
        for (int t = T - 1; t >= gmax; --t) {
            const size_t row = (size_t)cb * T + t;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (P.dgates) lp_st<VW>(P.dgates + row * G4 + g * H + cj, lp_vec<VW>{});
                lp_st_bf<VW>(P.dgates_bf + row * G4 + g * H + cj, lp_vec<VW>{});
            }
        }
    // cell operands of step gmax - 1 (c_new of a step is c_prev of the step after it: carried over in registers)
    lp_vec<VW> s_g4[4], c_prev, c_new, gout;
    {
        const int t0 = gmax > 0 ? gmax - 1 : 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) s_g4[g] = lp_ld<VW>(p_sv + (size_t)t0 * 4 * H + g * H);
        c_prev = lp_ld<VW>(p_c + (size_t)t0 * H);
        c_new = lp_ld<VW>(p_c + (size_t)(t0 + 1) * H);
        gout = p_g ? lp_ld<VW>(p_g + (size_t)t0 * H) : lp_vec<VW>{};
    }
    __syncthreads();

    // 16-byte pieces of the slot's published tile ([u][gate][R][2 halves]): piece lane + 64 k of every step, decomposed once
    constexpr int PIECES = 2 * UT * MT;
    const int n_pieces = UT * 8 * R;
    int pc_lds[PIECES];
    unsigned pc_sh[PIECES];
    unsigned pc_ok = 0;
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
        const int pc = lane + 64 * k;
        const int u = pc / (8 * R), rem = pc - u * 8 * R, gate = rem / (2 * R), rem2 = rem - gate * 2 * R, rrow = rem2 >> 1, half = rem2 & 1;
        if (pc < n_pieces && rrow < nrows) pc_ok |= 1u << k;
        pc_lds[k] = ((min(u, UT - 1) * 4 + gate) * MT * GT + rrow) * GT + 8 * half;
        pc_sh[k] = (unsigned)(((size_t)(row0 + rrow) * T) * G4 + gate * H + j0 + GT * u + 8 * half);
    }
    auto raise_x = [&](int steps) {                   // wave 1: the write-through stores of the step before have landed by now
        if (wave == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(flags_x + slot, (unsigned)steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    unsigned seen_up = 0, seen_lo = 0;
#ifdef MG_STAMPS
    unsigned long long ta = 0, tb = 0, ts0 = 0, ts1 = 0, tr0 = 0, tr1 = 0, sum_poll = 0, sum_load = 0, sum_mm = 0, sum_cell = 0, sum_pub = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif

    for (int t = gmax - 1; t >= -1; --t) {
        const bool need_mm = t + 1 < gmax;
        const bool need_x = !top && t >= 0;
        // One product of the step over the MT item tiles: p = 0 this layer's gate gradients of step t + 1 x W_hh^T, p = 1 the upper
        // layer's of step t x its W_ih^T.  One 16-row tile is in flight in raw[]; the next tile's loads go into each register as the
        // MFMAs free it.
        const unsigned own0 = ring_own + ((t + 1) & 1) * par_bytes, up0 = ring_up + (unsigned)(t & (LPS_XDEPTH - 1)) * par_bytes;
        auto run_jobs = [&](auto pc) {
            constexpr int p = decltype(pc)::value;
            auto job_off = [&](int m) {
                const bool valid = 16 * m + li < nrows;
                return (p == 0 ? own0 : up0) + rd_base + (unsigned)((valid ? 16 * m + li : 0) * 32);
            };
            u32x4 raw[KS];
            {
                const unsigned o0 = job_off(0);
#pragma unroll
                for (int i = 0; i < KS; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, o0 + i * rd_kstep, 0, 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const unsigned nxt = job_off(m + 1 < MT ? m + 1 : m);
                f32x4 acc[UT][2];
#pragma unroll
                for (int u = 0; u < UT; ++u) acc[u][0] = acc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < KS; ++i) {
                    const gbf8 av = as_bf8(raw[i]);
#pragma unroll
                    for (int u = 0; u < UT; ++u)
                        acc[u][i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, p == 0 ? fhh[u][i] : fih[u][i], acc[u][i & 1], 0, 0, 0);
                    if (m + 1 < MT) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, nxt + i * rd_kstep, 0, 16);
                }
#pragma unroll
                for (int u = 0; u < UT; ++u) {
                    const f32x4 s = acc[u][0] + acc[u][1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave][p][u][m][(4 * q + r) * GT + li] = s[r];
                }
            }
        };
        // Phase A: the product with the upper layer's gate gradients.  It does not depend on this layer's own step t + 1 and the layer
        // above is normally steps ahead: it runs while the other slots' publishes of step t + 1 are still on their way.
        MG_STAMP(ta);
        if (need_x) {
            if (wave == 0 && !lps_wait(flags_a, 0u, flags_up, gmax - t, seen_up, (gu32*)nullptr, 0, seen_lo, n_slots, lane)) s_abort = 1;
            gp_lds_barrier();
            if (s_abort) {
                if (tid == 0) __hip_atomic_store(status, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            run_jobs(std::integral_constant<int, 1>{});
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_load, tb, ta);
        // the next step's cell operands (first touch: HBM latency): requested behind phase A's loads, back before phase B waits on its own
        lp_vec<VW> s_g41[4], c_prev1, gout1;
        {
            const int t1 = t > 0 ? t - 1 : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) s_g41[g] = lp_ld<VW>(p_sv + (size_t)t1 * 4 * H + g * H);
            c_prev1 = lp_ld<VW>(p_c + (size_t)t1 * H);
            gout1 = p_g ? lp_ld<VW>(p_g + (size_t)t1 * H) : lp_vec<VW>{};
        }
        __builtin_amdgcn_sched_barrier(0);
        // Phase B: this layer's own recurrence - the chain that sets the step time
        MG_STAMP(ta);
        if (wave == 0 && !lps_wait(flags_a, need_mm ? (unsigned)(gmax - t - 1) : 0u, (gu32*)nullptr, 0, seen_up,
                                   (layer > 0 && t >= 0) ? flags_lo : (gu32*)nullptr, gmax - t - LPS_XDEPTH, seen_lo, n_slots, lane))
            s_abort = 1;
        gp_lds_barrier();
        if (s_abort) {
            if (tid == 0) __hip_atomic_store(status, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_poll, tb, ta);
        MG_STAMP(ta);
        if (need_mm) run_jobs(std::integral_constant<int, 0>{});
        if (need_mm && layer > 0) raise_x(gmax - t - 1);          // step t + 1's ring-X stores were issued a step ago
        gp_lds_barrier();
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_mm, tb, ta);
        {
            lp_vec<VW> part[2][4];
            if (need_mm) {
#pragma unroll
                for (int w = 0; w < 4; ++w) part[0][w] = lp_ld<VW>(&red[w][0][cu][cm][ce]);
            }
            if (need_x) {
#pragma unroll
                for (int w = 0; w < 4; ++w) part[1][w] = lp_ld<VW>(&red[w][1][cu][cm][ce]);
            }
            lp_vec<VW> vdi, vdf, vdg, vdo;
            const bool active = t >= 0 && t < len;
#pragma unroll
            for (int v = 0; v < VW; ++v) {
                const float dh_state = need_mm ? carry_h.v[v] + ((part[0][0].v[v] + part[0][1].v[v]) + (part[0][2].v[v] + part[0][3].v[v])) : carry_h.v[v];
                const float g_in = need_x ? ((part[1][0].v[v] + part[1][1].v[v]) + (part[1][2].v[v] + part[1][3].v[v])) : gout.v[v];
                float di = 0.f, df = 0.f, dg = 0.f, d_o = 0.f, ch = dh_state, cc = carry_c.v[v];
                if (active) {
                    const lp_cell_grad cg = lp_cell_bwd_fast(dh_state, carry_c.v[v], g_in, s_g4[0].v[v], s_g4[1].v[v], s_g4[2].v[v], s_g4[3].v[v],
                                                             c_prev.v[v], c_new.v[v]);
                    di = cg.di;
                    df = cg.df;
                    dg = cg.dg;
                    d_o = cg.d_o;
                    ch = 0.f;                 // all of dh_{t-1} comes through the matmul with the gates of this step
                    cc = cg.cc;
                }
                carry_h.v[v] = ch;
                carry_c.v[v] = cc;
                vdi.v[v] = di;
                vdf.v[v] = df;
                vdg.v[v] = dg;
                vdo.v[v] = d_o;
            }
            if (P.dgates) {
                lp_st<VW>(&res[cu][cm][0][ce], vdi);
                lp_st<VW>(&res[cu][cm][1][ce], vdf);
                lp_st<VW>(&res[cu][cm][2][ce], vdg);
                lp_st<VW>(&res[cu][cm][3][ce], vdo);
            }
            lp_st_bf<VW>(&pub[cu][0][16 * cm + crow][cc0], vdi);
            lp_st_bf<VW>(&pub[cu][1][16 * cm + crow][cc0], vdf);
            lp_st_bf<VW>(&pub[cu][2][16 * cm + crow][cc0], vdg);
            lp_st_bf<VW>(&pub[cu][3][16 * cm + crow][cc0], vdo);
        }
        if (t < 0) {
            if (mine) {
                lp_st<VW>(P.dh0 + (size_t)cb * H + cj, carry_h);
                lp_st<VW>(P.dc0 + (size_t)cb * H + cj, carry_c);
            }
            break;
        }
        gp_lds_barrier();
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_cell, ta, tb);
        // wave 0: ring A + flag A; wave 1: ring X (layers above the lowest); wave 2: the bf16 shadow the weight-gradient GEMMs read;
        // wave 3: the optional fp32 copy
        if (wave <= 2) {
#pragma unroll
            for (int k = 0; k < PIECES; ++k) {
                const int pc = lane + 64 * k;
                if ((pc_ok >> k) & 1u) {
                    const u32x4 v = *reinterpret_cast<const u32x4*>(&pub[0][0][0][0] + pc_lds[k]);
                    if (wave == 0) {
                        const unsigned off = ring_own + (unsigned)(t & 1) * par_bytes + wr_base + (unsigned)pc * 16;
                        if (one_xcd)
                            __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 0);
                        else
                            __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 16);
                    } else if (wave == 1) {
                        if (layer > 0)
                            __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, ring_x + (unsigned)(t & (LPS_XDEPTH - 1)) * par_bytes + wr_base + (unsigned)pc * 16, 0, 16);
                    } else {
                        *reinterpret_cast<u32x4*>(P.dgates_bf + (size_t)pc_sh[k] + (size_t)t * G4) = v;
                    }
                }
            }
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) gp_store_flag(flags_a + slot, (unsigned)(gmax - t), one_xcd);
            }
            if (wave == 1 && layer == 0 && lane == 0)         // progress report only
                __hip_atomic_store(flags_x + slot, (unsigned)(gmax - t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            MG_STAMP(tb);
            MG_STAMP_ADD(sum_pub, tb, ta);
        } else if (P.dgates) {
            for (int job = lane; job < UT * MT * 4 * 64; job += 64) {       // 16-byte stores: (u, m, gate, 16 rows x 4 column quads)
                const int e4 = job & 63, k = job >> 6, gate = k & 3, m = (k >> 2) % MT, u = k / (4 * MT);
                const int rb = e4 >> 2, c4 = 4 * (e4 & 3);
                if (16 * m + rb < nrows)
                    *reinterpret_cast<f32x4*>(P.dgates + ((size_t)(row0 + 16 * m + rb) * T + t) * G4 + gate * H + j0 + GT * u + c4) =
                        *reinterpret_cast<const f32x4*>(&res[u][m][gate][rb * GT + c4]);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) s_g4[g] = s_g41[g];
        c_new = c_prev;
        c_prev = c_prev1;
        gout = gout1;
    }
#ifdef MG_STAMPS
    MG_STAMP(ts1);
    MG_STAMP_REAL(tr1);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 2, tr0);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 3, tr1);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 4, sum_poll);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 5, sum_load);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 6, sum_mm);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 7, sum_cell);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 8, sum_pub);
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 9, (unsigned long long)(gmax + 1));
    MG_STAMP_STORE(g_stamps_lps, blockIdx.x, wave, lane, 10, (unsigned long long)(layer * 1000 + one_xcd));
#endif
    if (gmax > 0 && layer > 0) raise_x(gmax);          // the last step's ring-X stores (nobody below waits for more)
}

extern "C" {

int mg_lstm_persist_supported(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0) return 0;
    if (H % 128 != 0 || H > 128 * GP_KSTEPS) return 0;
    if (mg_ceil_div(B, GP_GROUPS) > 32) return 0;
    return gp_device_holds((long)GP_GROUPS * (H / GT));
}

int mg_lstm_fwd_persist_bf16(const float* xproj, const uint16_t* w_hh_bf, int ldw, const float* b_hh, const int64_t* seq_len, int B, int T,
                             int H, float* hstate, float* cstate, uint16_t* hstate_bf, float* out, float* saved, void* workspace,
                             size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(xproj && w_hh_bf && b_hh && hstate && cstate && hstate_bf && out && saved && B > 0 && T > 0 && H > 0,
                 "mg_lstm_fwd_persist_bf16: bad arguments (B=%d T=%d H=%d)", B, T, H);
    MG_CHECK_ARG(mg_lstm_persist_supported(B, T, H) && ldw >= H && ldw % 8 == 0,
                 "mg_lstm_fwd_persist_bf16: unsupported shape (B=%d T=%d H=%d ldw=%d): needs H %% 128 == 0, H <= 512, B <= 256", B, T, H, ldw);
    MG_CHECK_ARG((((uintptr_t)w_hh_bf | (uintptr_t)hstate_bf | (uintptr_t)workspace) % 16) == 0,
                 "mg_lstm_fwd_persist_bf16: bf16 buffers and workspace must be 16-byte aligned");
    if (!workspace || workspace_bytes < mg_gru_persist_workspace_bytes(B, H)) {
        mg_set_error("mg_lstm_fwd_persist_bf16: workspace of %zu bytes needed, got %zu", mg_gru_persist_workspace_bytes(B, H), workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)GP_FLAG_WORDS * sizeof(unsigned), st) != hipSuccess) {
        mg_set_error("mg_lstm_fwd_persist_bf16: memset failed");
        return MG_ELAUNCH;
    }
    const int R = (int)mg_ceil_div(B, GP_GROUPS);
    const unsigned grid = (unsigned)(GP_GROUPS * (H / GT));
#define LP_FWD(MT, KS)                                                                                                                          \
    hipLaunchKernelGGL((lstm_fwd_persist_kernel<MT, KS>), dim3(grid), dim3(256), 0, st, xproj, w_hh_bf, ldw, b_hh, seq_len, B, T, H, R, hstate, \
                       cstate, hstate_bf, out, saved, (unsigned*)workspace, (uint16_t*)((char*)workspace + GP_RING_OFFSET),                   \
                       g_mg_tuning[MG_TUNE_GRU_HANDOFF])
#define LP_FWD_KS(MT)                 \
    switch (H / 128) {                \
        case 1: LP_FWD(MT, 1); break; \
        case 2: LP_FWD(MT, 2); break; \
        case 3: LP_FWD(MT, 3); break; \
        default: LP_FWD(MT, 4); break; \
    }
    if (R <= 16) {
        LP_FWD_KS(1)
    } else {
        LP_FWD_KS(2)
    }
    MG_CHECK_LAUNCH("mg_lstm_fwd_persist_bf16");
    return MG_OK;
}

int mg_lstm_bwd_persist_bf16(const float* grad_out, const float* grad_hn, const float* grad_cn, const float* cstate, const float* saved,
                             const uint16_t* w_hh_t_bf, int ldt, const int64_t* seq_len, int B, int T, int H, float* dgates,
                             uint16_t* dgates_bf, float* dh0, float* dc0, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(grad_out && cstate && saved && w_hh_t_bf && dgates_bf && dh0 && dc0 && B > 0 && T > 0 && H > 0,
                 "mg_lstm_bwd_persist_bf16: bad arguments (B=%d T=%d H=%d)", B, T, H);
    MG_CHECK_ARG(mg_lstm_persist_supported(B, T, H) && ldt >= 4 * H && ldt % 8 == 0,
                 "mg_lstm_bwd_persist_bf16: unsupported shape (B=%d T=%d H=%d ldt=%d): needs H %% 128 == 0, H <= 512, B <= 256", B, T, H, ldt);
    MG_CHECK_ARG((((uintptr_t)w_hh_t_bf | (uintptr_t)dgates_bf | (uintptr_t)workspace) % 16) == 0,
                 "mg_lstm_bwd_persist_bf16: bf16 buffers and workspace must be 16-byte aligned");
    if (!workspace || workspace_bytes < mg_gru_persist_workspace_bytes(B, H)) {
        mg_set_error("mg_lstm_bwd_persist_bf16: workspace of %zu bytes needed, got %zu", mg_gru_persist_workspace_bytes(B, H), workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)GP_FLAG_WORDS * sizeof(unsigned), st) != hipSuccess) {
        mg_set_error("mg_lstm_bwd_persist_bf16: memset failed");
        return MG_ELAUNCH;
    }
    const int R = (int)mg_ceil_div(B, GP_GROUPS);
    const unsigned grid = (unsigned)(GP_GROUPS * (H / GT));
#define LP_BWD(MT, KS)                                                                                                                        \
    hipLaunchKernelGGL((lstm_bwd_persist_kernel<MT, KS>), dim3(grid), dim3(256), 0, st, grad_out, grad_hn, grad_cn, cstate, saved, w_hh_t_bf, ldt, \
                       seq_len, B, T, H, R, dgates, dgates_bf, dh0, dc0, (unsigned*)workspace,                                              \
                       (uint16_t*)((char*)workspace + GP_RING_OFFSET), g_mg_tuning[MG_TUNE_GRU_HANDOFF])
#define LP_BWD_KS(MT)                   \
    switch (H / 128) {                  \
        case 1: LP_BWD(MT, 4); break;   \
        case 2: LP_BWD(MT, 8); break;   \
        case 3: LP_BWD(MT, 12); break;  \
        default: LP_BWD(MT, 16); break; \
    }
    if (R <= 16) {
        LP_BWD_KS(1)
    } else {
        LP_BWD_KS(2)
    }
    MG_CHECK_LAUNCH("mg_lstm_bwd_persist_bf16");
    return MG_OK;
}

static int lps_groups(int B, int H, int L) {
    // the largest G in {8, 4, 2, 1} with L G H / 16 <= 512 workgroups (two per CU) and at most 32 items per group; 0 = none
    for (int G = 8; G >= 1; G >>= 1) {
        if ((long)L * G * (H / GT) > 512) continue;
        if (mg_ceil_div(B, G) > 32) return 0;
        return G;
    }
    return 0;
}

int mg_lstm_pstack_supported(int B, int T, int H, int L) {
    if (B <= 0 || T <= 0 || H <= 0 || L < 2 || L > MG_LSTM_MAX_LAYERS) return 0;
    if (H % 128 != 0 || H > 128 * GP_KSTEPS) return 0;
    const int G = lps_groups(B, H, L);
    return G > 0 && gp_device_holds((long)L * G * (H / GT));
}

size_t mg_lstm_pstack_workspace_bytes(int B, int H, int L) {
    const int G = lps_groups(B, H, L);
    if (G <= 0) return 0;
    return LPS_RING_OFFSET + (size_t)L * (2 + LPS_XDEPTH) * G * mg_ceil_div(B, G) * H * 2;
}

int mg_lstm_pstack_fwd_bf16(const mg_lstm_pstack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                            size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(layers && mg_lstm_pstack_supported(B, T, H, L), "mg_lstm_pstack_fwd_bf16: unsupported shape (B=%d T=%d H=%d L=%d)", B, T, H, L);
    LstmPStack a;
    for (int l = 0; l < L; ++l) {
        a.l[l] = layers[l];
        const mg_lstm_pstack_layer& p = layers[l];
        MG_CHECK_ARG(p.w_hh_bf && p.b_hh && p.hstate && p.cstate && p.hstate_bf && p.out && p.saved && p.ldwh >= H && p.ldwh % 8 == 0,
                     "mg_lstm_pstack_fwd_bf16: layer %d: bad arguments", l);
        MG_CHECK_ARG(l == 0 ? p.xproj != nullptr : (p.w_ih_bf && p.b_ih && p.ldwi >= H && p.ldwi % 8 == 0),
                     "mg_lstm_pstack_fwd_bf16: layer %d: %s", l, l == 0 ? "xproj missing" : "w_ih_bf / b_ih missing");
        MG_CHECK_ARG((((uintptr_t)p.w_hh_bf | (uintptr_t)p.w_ih_bf | (uintptr_t)p.hstate_bf) % 16) == 0,
                     "mg_lstm_pstack_fwd_bf16: layer %d: bf16 buffers must be 16-byte aligned", l);
    }
    if (!workspace || workspace_bytes < mg_lstm_pstack_workspace_bytes(B, H, L) || ((uintptr_t)workspace % 16) != 0) {
        mg_set_error("mg_lstm_pstack_fwd_bf16: 16-byte aligned workspace of %zu bytes needed, got %zu", mg_lstm_pstack_workspace_bytes(B, H, L),
                     workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync((unsigned*)workspace + LPS_FLAGA_WORD, 0, (size_t)3 * LPS_MAX_IDS * GP_SLOTS * sizeof(unsigned), st) != hipSuccess) {
        mg_set_error("mg_lstm_pstack_fwd_bf16: memset failed");
        return MG_ELAUNCH;
    }
    const int G = lps_groups(B, H, L);
    const int R = (int)mg_ceil_div(B, G);
    // 16 hidden units per workgroup, two workgroups per CU.  The 32-unit form (one workgroup per CU, each hand-off tile read once per
    // CU: what pays in the backward, whose tiles are four times larger) is built and bit-equal (MG_TUNE_LSTM_BWD_STACK bit 1, R > 16)
    // but MEASURED slower here: the shipped 8 x LSTM-512 step 17.37-17.42 against 16.07-16.16 ms (round 4, same box, alternating) -
    // one workgroup's MFMA and cell phases double and nothing runs on the CU while it waits, which the halved intake does not buy back
    const int UT = (R > 16 && (H / GT) % 2 == 0 && (g_mg_tuning[MG_TUNE_LSTM_BWD_STACK] & 2)) ? 2 : 1;
    const unsigned grid = (unsigned)(L * G * (H / (GT * UT)));
    uint16_t* rings = (uint16_t*)((char*)workspace + LPS_RING_OFFSET);
    const int force = g_mg_tuning[MG_TUNE_GRU_HANDOFF];
#define LPS_FWD(MT, KS, UT_)                                                                                                             \
    hipLaunchKernelGGL((lstm_stack_fwd_persist_kernel<MT, KS, UT_>), dim3(grid), dim3(256), 0, st, a, seq_len, B, T, H, L, G, R, (unsigned*)workspace, \
                       rings, force)
#define LPS_FWD_KS(MT, UT_)                 \
    switch (H / 128) {                 \
        case 1: LPS_FWD(MT, 1, UT_); break; \
        case 2: LPS_FWD(MT, 2, UT_); break; \
        case 3: LPS_FWD(MT, 3, UT_); break; \
        default: LPS_FWD(MT, 4, UT_); break; \
    }
    if (R <= 16) {
        LPS_FWD_KS(1, 1)
    } else if (UT == 2) {
        LPS_FWD_KS(2, 2)
    } else {
        LPS_FWD_KS(2, 1)
    }
    MG_CHECK_LAUNCH("mg_lstm_pstack_fwd_bf16");
    return MG_OK;
}


static int lpsb_plan(int B, int H, int L, int* G_out, int* UT_out) {
    // units per slot: 32 (UT = 2, one workgroup per CU) unless MG_TUNE_LSTM_BWD_STACK = 1 asks for 16 (two per CU); the largest G in
    // {8, 4, 2, 1} whose L G H / (16 UT) workgroups are all resident, with at most 32 items per group and L G <= 64 flag rows
    for (int UT = (g_mg_tuning[MG_TUNE_LSTM_BWD_STACK] & 1) ? 1 : 2; UT >= 1; --UT)
        for (int G = 8; G >= 1; G >>= 1) {
            const long wgs = (long)L * G * (H / (GT * UT));
            if (wgs > (UT == 2 ? 256 : 512) || L * G > LPS_MAX_IDS || !gp_device_holds(UT == 2 ? 2 * wgs : wgs)) continue;
            if (mg_ceil_div(B, G) > 32) break;
            *G_out = G;
            *UT_out = UT;
            return 1;
        }
    return 0;
}

int mg_lstm_pstack_bwd_supported(int B, int T, int H, int L) {
    if (B <= 0 || T <= 0 || H <= 0 || L < 2 || L > MG_LSTM_MAX_LAYERS) return 0;
    if (H % 128 != 0 || H > 128 * GP_KSTEPS) return 0;
    if ((long)B * T * 4 * H >= 2147483647L) return 0;
    int G, UT;
    return lpsb_plan(B, H, L, &G, &UT);
}

size_t mg_lstm_pstack_bwd_workspace_bytes(int B, int H, int L) {
    int G, UT;
    if (!lpsb_plan(B, H, L, &G, &UT)) return 0;
    return LPS_RING_OFFSET + (size_t)L * (2 + LPS_XDEPTH) * G * mg_ceil_div(B, G) * H * 8;
}

int mg_lstm_pstack_bwd_bf16(const mg_lstm_pstack_bwd_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                            size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(layers && mg_lstm_pstack_bwd_supported(B, T, H, L), "mg_lstm_pstack_bwd_bf16: unsupported shape (B=%d T=%d H=%d L=%d)", B, T, H, L);
    LstmPStackBwd a;
    for (int l = 0; l < L; ++l) {
        a.l[l] = layers[l];
        const mg_lstm_pstack_bwd_layer& p = layers[l];
        MG_CHECK_ARG(p.cstate && p.saved && p.w_hh_t_bf && p.dgates_bf && p.dh0 && p.dc0 && p.ldt >= 4 * H && p.ldt % 8 == 0,
                     "mg_lstm_pstack_bwd_bf16: layer %d: bad arguments", l);
        MG_CHECK_ARG(l + 1 == L || (p.w_ih_up_t_bf && p.ldt_up >= 4 * H && p.ldt_up % 8 == 0),
                     "mg_lstm_pstack_bwd_bf16: layer %d: the transposed W_ih of the layer above is missing", l);
        MG_CHECK_ARG((((uintptr_t)p.w_hh_t_bf | (uintptr_t)p.w_ih_up_t_bf | (uintptr_t)p.dgates_bf | (uintptr_t)p.dgates) % 16) == 0,
                     "mg_lstm_pstack_bwd_bf16: layer %d: weight and gate-gradient buffers must be 16-byte aligned", l);
    }
    if (!workspace || workspace_bytes < mg_lstm_pstack_bwd_workspace_bytes(B, H, L) || ((uintptr_t)workspace % 16) != 0) {
        mg_set_error("mg_lstm_pstack_bwd_bf16: 16-byte aligned workspace of %zu bytes needed, got %zu", mg_lstm_pstack_bwd_workspace_bytes(B, H, L),
                     workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync((unsigned*)workspace + LPS_FLAGA_WORD, 0, (size_t)3 * LPS_MAX_IDS * GP_SLOTS * sizeof(unsigned), st) != hipSuccess) {
        mg_set_error("mg_lstm_pstack_bwd_bf16: memset failed");
        return MG_ELAUNCH;
    }
    int G = 0, UT = 0;
    lpsb_plan(B, H, L, &G, &UT);
    const int R = (int)mg_ceil_div(B, G);
    const unsigned grid = (unsigned)(L * G * (H / (GT * UT)));
    uint16_t* rings = (uint16_t*)((char*)workspace + LPS_RING_OFFSET);
    const int force = g_mg_tuning[MG_TUNE_GRU_HANDOFF];
#define LPS_BWD(MT, KS, UT_)                                                                                                               \
    hipLaunchKernelGGL((lstm_stack_bwd_persist_kernel<MT, KS, UT_>), dim3(grid), dim3(256), 0, st, a, seq_len, B, T, H, L, G, R, (unsigned*)workspace, \
                       rings, force)
#define LPS_BWD_KS(MT, UT_)                    \
    switch (H / 128) {                         \
        case 1: LPS_BWD(MT, 4, UT_); break;    \
        case 2: LPS_BWD(MT, 8, UT_); break;    \
        case 3: LPS_BWD(MT, 12, UT_); break;   \
        default: LPS_BWD(MT, 16, UT_); break;  \
    }
    if (UT == 2) {
        if (R <= 16) {
            LPS_BWD_KS(1, 2)
        } else {
            LPS_BWD_KS(2, 2)
        }
    } else {
        if (R <= 16) {
            LPS_BWD_KS(1, 1)
        } else {
            LPS_BWD_KS(2, 1)
        }
    }
    MG_CHECK_LAUNCH("mg_lstm_pstack_bwd_bf16");
    return MG_OK;
}

}  // extern "C"

#ifdef MG_STAMPS
extern "C" int mg_diag_read_stamps_lps(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_lps), bytes < sizeof(g_stamps_lps) ? bytes : sizeof(g_stamps_lps), 0, hipMemcpyDeviceToHost);
}
#endif
