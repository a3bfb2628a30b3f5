// K3p - the GRU recurrence as ONE launch (throughput mode, bf16 matmul operands): W_hh stays in registers for all T steps and
// the workgroups hand the new state to each other through memory instead of ending the kernel after every step.
// Reference semantics: morgana/utils.py:345-393 (RecurrentCuDNNWrapper around torch.nn.GRU), same arithmetic as
// gru_fwd_step_bf16_kernel / gru_bwd_step_bf16_kernel (gru.hip), which stay as the path for shapes this file does not cover.
//
// Decomposition.  Batch items never interact inside the recurrence, so the batch is cut into 8 independent GROUPS of
// R = ceil(B / 8) items; a group is served by H / 16 workgroups (SLOTS), slot s owning hidden units [16 s, 16 s + 16) of all
// three gates: 48 rows of W_hh (bf16, 48 KB at H = 512) = 48 VGPRs per lane, loaded once.  Group = blockIdx % 8, so that the
// workgroups of a group usually share an XCD (blocks are dealt round-robin); that is a speed affinity only - the protocol below is
// placement independent.  Per step a workgroup reads the group's whole state h_{t-1} (R x H bf16), runs 3 H / 128 MFMAs per wave,
// sums its 4 waves' partials through LDS, applies the cell and publishes its 16 columns of h_t.
//
// Hand-off (cdna_hip_programming.md §6 Guideline 16, form R1 with sc1 loads in place of the acquire): every byte of the bf16
// state that another workgroup reads in this launch is written by a 16-byte sc1 (write-through) store of wave 0, drained with
// s_waitcnt vmcnt(0), then ONE lane stores the slot's flag (epoch = t + 1, sc1).  The consumer's wave 0 polls the group's flags
// (one 4-byte sc1 load per lane, one lane per slot), joins a workgroup barrier, and every load of the state is a
// buffer_load_dwordx4 sc1 to registers.  Flags are zeroed by a memset node ahead of every launch; every spin is bounded and a
// time-out sets the sticky status word behind the flags (mg_gru_persist_status) and makes the workgroup return.
// fp32 copies of the state, the outputs and the saved gate values are plain stores (nobody reads them before the kernel ends;
// a thread carries its own h_{t-1} element in a register).
//
// Steps at or beyond the longest sequence of a group need no matmul and no hand-off: the state is frozen and the outputs are
// zero, so the group finishes them without synchronising (ragged batches: BASELINE config C5).
#include "common.h"
#include "gru_cell.h"

#include "persist_common.h"

MG_STAMP_DECL(g_stamps_gp);
MG_STAMP_DECL(g_stamps_gpb);

template <int MT, int KS>
__global__ __launch_bounds__(384) void gru_fwd_persist_kernel(const float* __restrict__ xproj, const uint16_t* __restrict__ w_bf, int ldw,
                                                              const float* __restrict__ b_hh, const int64_t* __restrict__ seq_len,
                                                              int B, int T, int H, int R, float* __restrict__ hstate,
                                                              uint16_t* __restrict__ hstate_bf, float* __restrict__ out,
                                                              float* __restrict__ saved, unsigned* sync, uint16_t* ring, int force_sc1,
                                                              const int32_t* __restrict__ xrows, uint16_t* __restrict__ out_bf) {
    __shared__ float red[4][3][MT][GT * GT];
    __shared__ __attribute__((aligned(16))) uint16_t hb[MT][GT][GT];
    __shared__ float res[MT][6][GT * GT];          // a step's fp32 results on their way to waves 2 and 3, which store them
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

    // longest sequence of the group
    int gmax = 0;
    for (int r = 0; r < nrows; ++r) {
        const int64_t n = seq_len ? seq_len[row0 + r] : (int64_t)T;
        gmax = max(gmax, (int)(n < T ? n : T));
    }

    // W_hh fragments of this slot, for the whole launch
    const int kbase = (wave & 3) * (H / 4) + 8 * q;   // (the storing waves 4, 5 run the prologue as waves 0, 1: same loads, same LDS values)
    gbf8 fr[KS], fz[KS], fn[KS];
    {
        const uint16_t* wr = w_bf + (size_t)(j0 + li) * ldw;
        const uint16_t* wz = wr + (size_t)H * ldw;
        const uint16_t* wn = wz + (size_t)H * ldw;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            fr[i] = *reinterpret_cast<const gbf8*>(wr + kbase + 32 * i);
            fz[i] = *reinterpret_cast<const gbf8*>(wz + kbase + 32 * i);
            fn[i] = *reinterpret_cast<const gbf8*>(wn + kbase + 32 * i);
        }
    }
    // Hand-off ring: [2 (epoch parity)][8 groups][H / 16 slots][R items][16 units] bf16.  A slot's tile is contiguous (R x 32 bytes:
    // whole 128-byte lines written by one store instruction), and the two parities are reused all launch long, so the lines stay
    // resident in L2 - a state written to fresh addresses every step (the [B, T+1, H] shadow) costs an L2 line allocation per
    // step and was read back 2.6x slower (1290 vs 490 cycles for the 4 loads of a lane).
    const unsigned par_bytes = (unsigned)(GP_GROUPS * n_slots * R * 32);
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, (int)(2 * par_bytes), 0x00020000);
    // reader: lane (li, q) of wave w, k-step i wants units kbase + 32 i .. + 7 = slot kbase / 16 + 2 i, half q & 1, of item li
    const unsigned rd_base = (unsigned)(((group * n_slots + (kbase >> 4)) * R) * 32 + 16 * (q & 1));
    const unsigned rd_kstep = (unsigned)(2 * R * 32);
    const unsigned wr_base = (unsigned)(((group * n_slots + slot) * R) * 32);

    // cell role: thread (bl, jl) owns element (row0 + 16 m + bl, j0 + jl) in every step
    const int bl = (tid & 255) >> 4, jl = tid & 15;
    const int j = j0 + jl;
    const float bhr = b_hh[j], bhz = b_hh[H + j], bhn = b_hh[2 * H + j];
    float hprev[MT], xr[MT], xz[MT], xn[MT];
    int len[MT];
    bool mine[MT];
    const float* xp[MT];
    const int32_t* xi[MT];
    int xrow1[MT];                                 // table row of step t + 1 (xrows only)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        mine[m] = 16 * m + bl < nrows;
        const int b = row0 + (mine[m] ? 16 * m + bl : 0);
        hprev[m] = hstate[((size_t)b * (T + 1)) * H + j];
        len[m] = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
        hb[m][bl][jl] = mg_f2bf(hprev[m]);
        // xrows: the input projections are a TABLE (one row per phone, upsample_to_repetitions' row map picks frame (b, t)'s row) - the
        // recurrence reads the table through the map instead of a [B, T, 3H] copy of its rows written and read back once (393 MB at C4).
        // The index of step t + 2 is requested during step t, the projections of step t + 1 through the index fetched a step before.
        xi[m] = xrows ? xrows + (size_t)b * T : nullptr;
        xp[m] = xrows ? xproj + j : xproj + (size_t)b * T * 3 * H + j;
        const float* x0 = xrows ? xp[m] + (size_t)xi[m][0] * 3 * H : xp[m];
        xrow1[m] = xrows ? xi[m][T > 1 ? 1 : 0] : 0;
        xr[m] = x0[0];                           // input projections of step 0; step t + 1's are requested during step t
        xz[m] = x0[H];
        xn[m] = x0[2 * H];
    }
    __syncthreads();

    // wave 0 publishes the tile in hb as epoch e (state h_e) into ring parity e & 1, then raises the slot's flag to e + 1.
    // Everything a step leaves for AFTER the launch - the bf16 shadow of the state, the fp32 state / output / saved gate values - is
    // stored by waves 4 and 5, which do nothing else: those stores go to lines nobody has touched, their acknowledgements take
    // about a step, and a wave's loads and stores complete in order (one vmcnt) - in a wave that also fetches the next state tile
    // every such fetch waits for the old stores first (the small GRU stack gained 10 % from the same split, profiles/r4_gru_small_stack.txt;
    // here, where waves 2 and 3 stored before: C4 3.396 -> 3.378, C5 6.435 -> 6.406 ms.  The same split of the BACKWARD kernel - wave 1's
    // bf16 shadows to a storing wave - measured +1.1 %: not adopted).
    auto publish = [&](int e) {
        if (wave == 0 && lane < 2 * 16 * MT) {
            const int rrow = lane >> 1, half = lane & 1;
            if (rrow < nrows) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(&hb[rrow >> 4][rrow & 15][8 * half]);
                const unsigned off = (e & 1) * par_bytes + wr_base + (unsigned)(rrow * 32 + half * 16);
                if (one_xcd)
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 0);      // stays in the group's L2
                else
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs_ring, off, 0, 16);     // sc1: written through
            }
        }
        if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) gp_store_flag(flags + slot, (unsigned)(e + 1), one_xcd);
        }
    };
    publish(0);
    if (wave >= 4) {
        // the storing waves' whole launch: the three barriers of every step (leaving with the others on a time-out), then the step's
        // values out of hb / res - both rewritten only behind the next step's second barrier
        const int st = tid - 256;                        // 0 .. 127
        for (int t = 0; t < gmax; ++t) {
            gp_lds_barrier();
            if (s_abort) return;
            gp_lds_barrier();
            gp_lds_barrier();
            if (wave == 4 && lane < 2 * 16 * MT) {
                const int rrow = lane >> 1, half = lane & 1;
                if (rrow < nrows)
                    *reinterpret_cast<u32x4*>(hstate_bf + ((size_t)(row0 + rrow) * (T + 1) + t + 1) * H + j0 + 8 * half) =
                        *reinterpret_cast<const u32x4*>(&hb[rrow >> 4][rrow & 15][8 * half]);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int e = st + 128 * half, rb = e >> 4, cj = j0 + (e & 15);
                    if (16 * m + rb < nrows) {
                        const int b = row0 + 16 * m + rb;
                        const size_t row = (size_t)b * T + t;
                        hstate[((size_t)b * (T + 1) + t + 1) * H + cj] = res[m][0][e];
                        out[row * H + cj] = res[m][1][e];
                        // bf16 copy of the output row (zero past the item's length, as `out`): the operand of the Linear layer behind
                        // the wrapper, which otherwise costs a cast pass over [B, T, H] (C4 37 us, C5 98 us)
                        if (out_bf) out_bf[row * H + cj] = mg_f2bf(res[m][1][e]);
                        float* sv = saved + row * 4 * H + cj;
                        sv[0] = res[m][2][e];
                        sv[H] = res[m][3][e];
                        sv[2 * H] = res[m][4][e];
                        sv[3 * H] = res[m][5][e];
                    }
                }
        }
        return;
    }

#ifdef MG_STAMPS
    unsigned long long ta = 0, tb = 0, ts0 = 0, ts1 = 0, tr0 = 0, tr1 = 0, sum_poll = 0, sum_load = 0, sum_mm = 0, sum_cell = 0, sum_pub = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    for (int t = 0; t < gmax; ++t) {
        MG_STAMP(ta);
        if (wave == 0 && !gp_wait_flags(flags, n_slots, (unsigned)(t + 1), lane)) s_abort = 1;
        gp_lds_barrier();
        if (s_abort) {
            if (tid == 0) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_poll, tb, ta);
        // h_t of the group: bf16, sc1 loads only; then the next step's input projections (first touch: HBM latency, hidden behind
        // this step - vector memory returns in order, so they must not be queued AHEAD of the state loads or the flag poll)
        u32x4 raw[MT][KS];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool valid = 16 * m + li < nrows;
            const unsigned off = (t & 1) * par_bytes + rd_base + (unsigned)((valid ? 16 * m + li : 0) * 32);
#pragma unroll
            for (int i = 0; i < KS; ++i) raw[m][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off + i * rd_kstep, 0, 16);
        }
        float xr1[MT], xz1[MT], xn1[MT];
        int xrow2[MT];
        const int t1 = t + 1 < T ? t + 1 : t;
        const int t2 = t + 2 < T ? t + 2 : T - 1;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float* x1 = xrows ? xp[m] + (size_t)xrow1[m] * 3 * H : xp[m] + (size_t)t1 * 3 * H;
            xr1[m] = x1[0];
            xz1[m] = x1[H];
            xn1[m] = x1[2 * H];
            xrow2[m] = xrows ? xi[m][t2] : 0;
        }
        __builtin_amdgcn_sched_barrier(0);          // every load in flight before the first MFMA waits (one round trip)
#ifdef MG_STAMPS
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_load, ta, tb);
#endif
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 acc_r = {0.f, 0.f, 0.f, 0.f}, acc_z = acc_r, acc_n = acc_r;
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const gbf8 a = as_bf8(raw[m][i]);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, fr[i], acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, fz[i], acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, fn[i], acc_n, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = (4 * q + r) * GT + li;
                red[wave][0][m][e] = acc_r[r];
                red[wave][1][m][e] = acc_z[r];
                red[wave][2][m][e] = acc_n[r];
            }
        }
        gp_lds_barrier();
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_mm, tb, ta);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int e = bl * GT + jl;
            const float hr = mg_gru_sum4(red[0][0][m][e], red[1][0][m][e], red[2][0][m][e], red[3][0][m][e], bhr);
            const float hz = mg_gru_sum4(red[0][1][m][e], red[1][1][m][e], red[2][1][m][e], red[3][1][m][e], bhz);
            const float hn = mg_gru_sum4(red[0][2][m][e], red[1][2][m][e], red[2][2][m][e], red[3][2][m][e], bhn);
            const mg_gru_cell_out c = mg_gru_cell(xr[m], xz[m], xn[m], hr, hz, hn, hprev[m]);
            hprev[m] = t < len[m] ? c.hnew : hprev[m];
            hb[m][bl][jl] = mg_f2bf(hprev[m]);
            res[m][0][e] = hprev[m];
            res[m][1][e] = t < len[m] ? c.hnew : 0.f;
            res[m][2][e] = c.r;
            res[m][3][e] = c.z;
            res[m][4][e] = c.n;
            res[m][5][e] = hn;
        }
        gp_lds_barrier();
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_cell, ta, tb);
        publish(t + 1);                             // first: the other workgroups wait for exactly this
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_pub, tb, ta);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            xr[m] = xr1[m];
            xz[m] = xz1[m];
            xn[m] = xn1[m];
            xrow1[m] = xrow2[m];
        }
    }
#ifdef MG_STAMPS
    MG_STAMP(ts1);
    MG_STAMP_REAL(tr1);
    MG_STAMP_STORE(g_stamps_gp, blockIdx.x, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_gp, blockIdx.x, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_gp, blockIdx.x, wave, lane, 2, tr0);
    MG_STAMP_STORE(g_stamps_gp, blockIdx.x, wave, lane, 3, tr1);
    MG_STAMP_STORE(g_stamps_gp, blockIdx.x, wave, lane, 4, sum_poll);
    MG_STAMP_STORE(g_stamps_gp, blockIdx.x, wave, lane, 5, sum_load);
    MG_STAMP_STORE(g_stamps_gp, blockIdx.x, wave, lane, 6, sum_mm);
    MG_STAMP_STORE(g_stamps_gp, blockIdx.x, wave, lane, 7, sum_cell);
    MG_STAMP_STORE(g_stamps_gp, blockIdx.x, wave, lane, 8, sum_pub);
    MG_STAMP_STORE(g_stamps_gp, blockIdx.x, wave, lane, 9, (unsigned long long)gmax);
#endif
    // beyond the group's longest sequence: state frozen, outputs zero - no matmul, no hand-off
    for (int t = gmax; t < T; ++t) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (mine[m]) {
                const int b = row0 + 16 * m + bl;
                const size_t row = (size_t)b * T + t;
                const size_t nxt = ((size_t)b * (T + 1) + t + 1) * H + j;
                hstate[nxt] = hprev[m];
                hstate_bf[nxt] = mg_f2bf(hprev[m]);
                out[row * H + j] = 0.f;
                if (out_bf) out_bf[row * H + j] = (uint16_t)0;
                float* sv = saved + row * 4 * H;
                sv[j] = 0.f;
                sv[H + j] = 0.f;
                sv[2 * H + j] = 0.f;
                sv[3 * H + j] = 0.f;
            }
    }
}

// Backward recurrence in one launch.  Slot s of a group owns d h[:, 16 s .. 16 s + 16): per step t (T-1 down to 0, then the
// t = -1 pass that yields dh0) it needs dhproj_{t+1} of the whole group (R x 3H bf16: the hand-off, written by all slots in
// the step before), contracts it with its 16 rows of W_hh^T (16 x 3H bf16 = 48 VGPRs per lane at H = 512, resident), applies
// the gate derivatives and publishes its 3 x 16 columns of dhproj_t.  The carry (d loss / d h_{t-1} through the z path) stays
// in a register of the thread that owns the element.  Flag of a slot = gmax - t once row t is published (gmax = the group's
// longest sequence): rows at or beyond gmax are all-zero, written without synchronising before the dependent steps start.
// Ring: [2 (t parity)][8 groups][H / 16 slots][R items][3 gates][16 units] bf16, a slot's tile contiguous (R x 96 bytes).
template <int MT, int KS>
__global__ __launch_bounds__(256) void gru_bwd_persist_kernel(const float* __restrict__ grad_out, const float* __restrict__ grad_hn,
                                                              const float* __restrict__ hstate, const float* __restrict__ saved,
                                                              const uint16_t* __restrict__ wt_bf, int ldt,
                                                              const int64_t* __restrict__ seq_len, int B, int T, int H, int R,
                                                              float* __restrict__ dxproj, float* __restrict__ dhproj,
                                                              uint16_t* __restrict__ dhproj_bf, uint16_t* __restrict__ dxproj_bf, float* __restrict__ dh0, unsigned* sync,
                                                              uint16_t* ring, int force_sc1) {
    __shared__ float red[4][MT][GT * GT];
    __shared__ __attribute__((aligned(16))) uint16_t pub[4][MT * GT][GT];      // dr, dz, dn r (the hand-off) and dn (dxproj shadow only)
    __shared__ float res[MT][4][GT * GT];          // dr, dz, dn, dn r of the step for waves 2 and 3, which store them
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
    const int G = 3 * H;
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

    // W_hh^T fragments: lane (li, q) holds row j0 + li, gate rows gbase + 32 i .. + 7
    const int gbase = wave * (G / 4) + 8 * q;
    gbf8 fb[KS];
    unsigned rd_off[KS];                               // ring offset of k-step i (item 0, parity 0)
    {
        const uint16_t* wp = wt_bf + (size_t)(j0 + li) * ldt + gbase;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            fb[i] = *reinterpret_cast<const gbf8*>(wp + 32 * i);
            const int g = gbase + 32 * i, gate = g / H, col = g - gate * H;
            rd_off[i] = (unsigned)((((group * n_slots + (col >> 4)) * 3 + gate) * R) * 32 + 16 * (q & 1));
        }
    }
    const unsigned par_bytes = (unsigned)(GP_GROUPS * n_slots * R * 96);
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, (int)(2 * par_bytes), 0x00020000);
    const unsigned wr_base = (unsigned)(((group * n_slots + slot) * R) * 96);

    const int bl = tid >> 4, jl = tid & 15;
    const int j = j0 + jl;
    float carry[MT];
    int len[MT];
    bool mine[MT];
    const float *p_sv[MT], *p_h[MT], *p_g[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        mine[m] = 16 * m + bl < nrows;
        const int b = row0 + (mine[m] ? 16 * m + bl : 0);
        carry[m] = grad_hn ? grad_hn[(size_t)b * H + j] : 0.f;
        len[m] = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
        p_sv[m] = saved + (size_t)b * T * 4 * H + j;
        p_h[m] = hstate + (size_t)b * (T + 1) * H + j;
        p_g[m] = grad_out + (size_t)b * T * H + j;
    }
    // rows beyond the group's longest sequence: zero gradients, carry untouched, nothing to wait for
    for (int t = T - 1; t >= gmax; --t) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            if (mine[m]) {
                const size_t row = (size_t)(row0 + 16 * m + bl) * T + t;
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    if (dxproj) dxproj[row * G + g * H + j] = 0.f;
                    if (dhproj) dhproj[row * G + g * H + j] = 0.f;
                    dhproj_bf[row * G + g * H + j] = 0;
                    if (dxproj_bf) dxproj_bf[row * G + g * H + j] = 0;
                }
            }
    }
    // the slot's tile as 16-byte pieces: piece p = (gate, item, half) in ring order; lane handles pieces lane + 64 k
    constexpr int PIECES = (6 * GT * MT + 63) / 64;
    bool pc_ok[PIECES], pc_gate2[PIECES];
    int pc_lds[PIECES];
    size_t pc_shadow[PIECES];
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
        const int pc = lane + 64 * k, gate = pc / (2 * R), rem = pc - gate * 2 * R, rrow = rem >> 1, half = rem & 1;
        pc_ok[k] = gate < 3 && rrow < nrows;
        pc_gate2[k] = gate == 2;
        pc_lds[k] = (gate * MT * GT + rrow) * GT + 8 * half;
        pc_shadow[k] = (size_t)(row0 + rrow) * T * G + gate * H + j0 + 8 * half;
    }
    // cell operands of step gmax - 1; step t - 1's are requested during step t
    float s_r[MT], s_z[MT], s_n[MT], s_hn[MT], hprev[MT], gout[MT];
    {
        const int t0 = gmax > 0 ? gmax - 1 : 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float* sv = p_sv[m] + (size_t)t0 * 4 * H;
            s_r[m] = sv[0];
            s_z[m] = sv[H];
            s_n[m] = sv[2 * H];
            s_hn[m] = sv[3 * H];
            hprev[m] = p_h[m][(size_t)t0 * H];
            gout[m] = p_g[m][(size_t)t0 * H];
        }
    }
    __syncthreads();

#ifdef MG_STAMPS
    unsigned long long ta = 0, tb = 0, ts0 = 0, ts1 = 0, tr0 = 0, tr1 = 0, sum_poll = 0, sum_load = 0, sum_mm = 0, sum_cell = 0, sum_pub = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    for (int t = gmax - 1; t >= -1; --t) {
        const bool need_mm = t + 1 < gmax;               // row t + 1 holds gradients of a live step
        u32x4 raw[MT][KS];
        MG_STAMP(ta);
        if (need_mm) {
            if (wave == 0 && !gp_wait_flags(flags, n_slots, (unsigned)(gmax - t - 1), lane)) s_abort = 1;
            gp_lds_barrier();
            if (s_abort) {
                if (tid == 0) __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            MG_STAMP(tb);
            MG_STAMP_ADD(sum_poll, tb, ta);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const bool valid = 16 * m + li < nrows;
                const unsigned off = ((t + 1) & 1) * par_bytes + (unsigned)((valid ? 16 * m + li : 0) * 32);
#pragma unroll
                for (int i = 0; i < KS; ++i) raw[m][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off + rd_off[i], 0, 16);
            }
        }
#ifdef MG_STAMPS
        if (!need_mm) tb = ta;
#endif
        // the next step's cell operands (first touch: HBM latency), queued BEHIND the hand-off loads
        float s_r1[MT], s_z1[MT], s_n1[MT], s_hn1[MT], hprev1[MT], gout1[MT];
        {
            const int t1 = t > 0 ? t - 1 : 0;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float* sv = p_sv[m] + (size_t)t1 * 4 * H;
                s_r1[m] = sv[0];
                s_z1[m] = sv[H];
                s_n1[m] = sv[2 * H];
                s_hn1[m] = sv[3 * H];
                hprev1[m] = p_h[m][(size_t)t1 * H];
                gout1[m] = p_g[m][(size_t)t1 * H];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef MG_STAMPS
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_load, ta, tb);
#endif
        if (need_mm) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x4 acc3[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // KS is a multiple of 3
#pragma unroll
                for (int i = 0; i < KS; ++i)
                    acc3[i % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(raw[m][i]), fb[i], acc3[i % 3], 0, 0, 0);
                const f32x4 acc = (acc3[0] + acc3[1]) + acc3[2];
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][m][(4 * q + r) * GT + li] = acc[r];
            }
            gp_lds_barrier();
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_mm, tb, ta);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int e = bl * GT + jl;
            const float dstate = need_mm ? mg_gru_dstate(carry[m], red[0][m][e], red[1][m][e], red[2][m][e], red[3][m][e]) : carry[m];
            float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, c = dstate;
            if (t >= 0 && t < len[m]) {
                const mg_gru_cell_grad g = mg_gru_cell_bwd(dstate, gout[m], s_r[m], s_z[m], s_n[m], s_hn[m], hprev[m]);
                dr = g.dr; dz = g.dz; dn = g.dn; dnr = g.dnr; c = g.carry;
            }
            carry[m] = c;
            res[m][0][e] = dr;
            res[m][1][e] = dz;
            res[m][2][e] = dn;
            res[m][3][e] = dnr;
            pub[0][16 * m + bl][jl] = mg_f2bf(dr);
            pub[1][16 * m + bl][jl] = mg_f2bf(dz);
            pub[2][16 * m + bl][jl] = mg_f2bf(dnr);
            pub[3][16 * m + bl][jl] = mg_f2bf(dn);
        }
        if (t < 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                if (mine[m]) dh0[(size_t)(row0 + 16 * m + bl) * H + j] = carry[m];
            break;
        }
        gp_lds_barrier();
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_cell, ta, tb);
        if (wave <= 1) {
            // dhproj_t[:, gate, 16 s .. 16 s + 16): per gate R items x 32 bytes, the slot's 3 R x 32 bytes contiguous in the ring;
            // wave 0 publishes, wave 1 writes the bf16 shadow for the weight-gradient GEMM
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
                        *reinterpret_cast<u32x4*>(dhproj_bf + pc_shadow[k] + (size_t)t * G) = v;
                        if (dxproj_bf) {                 // (dr, dz, dn): the n plane differs from the hand-off's dn r
                            const u32x4 vx = pc_gate2[k] ? *reinterpret_cast<const u32x4*>(&pub[0][0][0] + pc_lds[k] + MT * GT * GT) : v;
                            *reinterpret_cast<u32x4*>(dxproj_bf + pc_shadow[k] + (size_t)t * G) = vx;
                        }
                    }
                }
            }
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) gp_store_flag(flags + slot, (unsigned)(gmax - t), one_xcd);
            }
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_pub, tb, ta);
        if (wave >= 2 && dxproj) {                      // fp32 results (optional): waves 2 and 3 only (see the forward kernel)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int e = (tid - 128) + 128 * half, rb = e >> 4, cj = j0 + (e & 15);
                    if (16 * m + rb < nrows) {
                        const size_t row = (size_t)(row0 + 16 * m + rb) * T + t;
                        float* dx = dxproj + row * G + cj;
                        float* dhp = dhproj + row * G + cj;
                        const float dr = res[m][0][e], dz = res[m][1][e];
                        dx[0] = dr;  dx[H] = dz;  dx[2 * H] = res[m][2][e];
                        dhp[0] = dr; dhp[H] = dz; dhp[2 * H] = res[m][3][e];
                    }
                }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            s_r[m] = s_r1[m]; s_z[m] = s_z1[m]; s_n[m] = s_n1[m]; s_hn[m] = s_hn1[m];
            hprev[m] = hprev1[m]; gout[m] = gout1[m];
        }
    }
#ifdef MG_STAMPS
    MG_STAMP(ts1);
    MG_STAMP_REAL(tr1);
    MG_STAMP_STORE(g_stamps_gpb, blockIdx.x, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_gpb, blockIdx.x, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_gpb, blockIdx.x, wave, lane, 2, tr0);
    MG_STAMP_STORE(g_stamps_gpb, blockIdx.x, wave, lane, 3, tr1);
    MG_STAMP_STORE(g_stamps_gpb, blockIdx.x, wave, lane, 4, sum_poll);
    MG_STAMP_STORE(g_stamps_gpb, blockIdx.x, wave, lane, 5, sum_load);
    MG_STAMP_STORE(g_stamps_gpb, blockIdx.x, wave, lane, 6, sum_mm);
    MG_STAMP_STORE(g_stamps_gpb, blockIdx.x, wave, lane, 7, sum_cell);
    MG_STAMP_STORE(g_stamps_gpb, blockIdx.x, wave, lane, 8, sum_pub);
    MG_STAMP_STORE(g_stamps_gpb, blockIdx.x, wave, lane, 9, (unsigned long long)gmax);
#endif
}

// The backward recurrence with WIDE slots: 16 groups of R = ceil(B / 16) items (two per XCD) x H / 32 slots of 32 hidden units - the
// same H / 16 workgroups per XCD, the same grid and tickets as the kernel above.  Why: a step's hand-off intake is what a CU can
// request, one 128-byte line per ~8 cycles (profiles/r4_notes_falsified_kernel_ideas.txt; DESIGN R4.12), and a slot needs the gate
// gradients of ALL 3 H gate columns of its group's items: 8 items x 3 H bf16 = 192 lines = ~1 500 of the step's ~3 200 cycles.  Half
// the items per group is half the lines; the slot then owns twice the units (2 x 12 resident W_hh^T fragments per lane = 96 VGPRs,
// 24 instead of 12 MFMAs per wave and step: +190 cycles) so that the groups still fill one CU per slot.  The M dimension of the MFMA
// holds 4 valid items of 16 either way.  Products, their order and the sums are those of the kernel above: results EQUAL (tested).
// Ring: [2 (t parity)][16 groups][H / 32 slots][3 gates][R items][32 units] bf16 - a k-step of 32 gate columns is one slot's row
// of one gate: per instruction R x 64 contiguous bytes.  Flags / XCC table: the XCD group's 32 words, 16 per virtual group.
template <int KS>
__global__ __launch_bounds__(256) void gru_bwd_persist_wide_kernel(const float* __restrict__ grad_out, const float* __restrict__ grad_hn,
                                                                   const float* __restrict__ hstate, const float* __restrict__ saved,
                                                                   const uint16_t* __restrict__ wt_bf, int ldt,
                                                                   const int64_t* __restrict__ seq_len, int B, int T, int H, int R,
                                                                   float* __restrict__ dxproj, float* __restrict__ dhproj,
                                                                   uint16_t* __restrict__ dhproj_bf, uint16_t* __restrict__ dxproj_bf,
                                                                   float* __restrict__ dh0, unsigned* sync, uint16_t* ring, int force_sc1) {
    constexpr int UW = 32;                         // hidden units per slot
    constexpr int RM = 8;                          // items per group at most (B <= 128)
    __shared__ float red[4][2][GT * GT];           // [k quarter = wave][unit tile][item x unit]
    __shared__ __attribute__((aligned(16))) uint16_t pub[4][RM][UW];      // dr, dz, dn r (the hand-off) and dn (dxproj shadow only)
    __shared__ float res[4][RM][UW];               // dr, dz, dn, dn r of the step for waves 2 and 3, which store them
    __shared__ int s_abort, s_xcd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int n_slots = H / GT;                    // tickets per XCD group: as many workgroups as the narrow kernel's
    const int n_vslots = H / UW;
    const int claim = gp_claim_slot((gu32*)sync + GP_TICKET_OFFSET, n_slots, tid, &s_xcd, force_sc1 & 2);
    if (claim < 0) return;
    const int xg = claim / GP_SLOTS, ticket = claim % GP_SLOTS;
    if (ticket >= n_slots) return;
    const int vsub = ticket / n_vslots, slot = ticket - vsub * n_vslots;       // n_slots = 2 n_vslots: vsub is 0 or 1
    const int group = 2 * xg + vsub;               // virtual group 0 .. 15
    const int row0 = group * R;
    const int nrows = min(R, B - row0);
    if (nrows <= 0) return;                        // (a whole virtual group: its 16 workgroups leave together)
    const int j0 = slot * UW;
    const int G = 3 * H;
    gu32* flags = (gu32*)sync + xg * GP_SLOTS + vsub * n_vslots;
    gu32* status = (gu32*)sync + GP_FLAG_WORDS;
    if (tid == 0) s_abort = 0;
    const int one_xcd = (force_sc1 & 1) ? 0
                                        : gp_group_on_one_xcd((gu32*)sync + GP_GROUPS * GP_SLOTS + xg * GP_SLOTS + vsub * n_vslots, slot,
                                                              n_vslots, tid, &s_xcd);
    if (one_xcd < 0) {
        if (tid == 0) __hip_atomic_store(status, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }

    int gmax = 0;
    for (int r = 0; r < nrows; ++r) {
        const int64_t n = seq_len ? seq_len[row0 + r] : (int64_t)T;
        gmax = max(gmax, (int)(n < T ? n : T));
    }

    // W_hh^T fragments: lane (li, q) holds rows j0 + 16 u + li, gate columns gk + 32 i + 8 q .. + 7 (gk = the wave's quarter of 3 H)
    const int gk = wave * (G / 4);
    gbf8 fb[2][KS];
    unsigned rd_off[KS];                               // ring offset of k-step i (item 0, parity 0)
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        const int g = gk + 32 * i, gate = g / H, col = g - gate * H;       // 32 | H and 32 | G / 4: a k-step is one slot's row of one gate
#pragma unroll
        for (int u = 0; u < 2; ++u)
            fb[u][i] = *reinterpret_cast<const gbf8*>(wt_bf + (size_t)(j0 + 16 * u + li) * ldt + g + 8 * q);
        rd_off[i] = (unsigned)((((group * n_vslots + (col >> 5)) * 3 + gate) * R) * 64 + 16 * q);
    }
    const unsigned par_bytes = (unsigned)(2 * GP_GROUPS * n_vslots * R * 192);
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, (int)(2 * par_bytes), 0x00020000);
    const unsigned wr_base = (unsigned)(((group * n_vslots + slot) * 3 * R) * 64);

    const int bl = tid >> 5, jl = tid & 31;            // cell element: item bl (8 rows of threads), unit j0 + jl
    const int j = j0 + jl;
    const int eu = jl >> 4, ee = bl * GT + (jl & 15);  // ... = element ee of unit tile eu in `red`
    const bool mine = bl < nrows;
    const int b_own = row0 + (mine ? bl : 0);
    float carry = grad_hn ? grad_hn[(size_t)b_own * H + j] : 0.f;
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b_own]) : T;
    const float* p_sv = saved + (size_t)b_own * T * 4 * H + j;
    const float* p_h = hstate + (size_t)b_own * (T + 1) * H + j;
    const float* p_g = grad_out + (size_t)b_own * T * H + j;
    // rows beyond the group's longest sequence: zero gradients, carry untouched, nothing to wait for
    for (int t = T - 1; t >= gmax; --t) {
        if (mine) {
            const size_t row = (size_t)b_own * T + t;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                if (dxproj) dxproj[row * G + g * H + j] = 0.f;
                if (dhproj) dhproj[row * G + g * H + j] = 0.f;
                dhproj_bf[row * G + g * H + j] = 0;
                if (dxproj_bf) dxproj_bf[row * G + g * H + j] = 0;
            }
        }
    }
    // the slot's tile as 16-byte pieces in ring order: piece p = ((gate R + item) 4 + quarter); lane handles pieces lane, lane + 64
    constexpr int PIECES = (3 * RM * 4 + 63) / 64;
    bool pc_ok[PIECES], pc_gate2[PIECES];
    int pc_lds[PIECES];
    size_t pc_shadow[PIECES];
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
        const int pc = lane + 64 * k, gate = pc / (4 * R), rem = pc - gate * 4 * R, rrow = rem >> 2, quarter = rem & 3;
        pc_ok[k] = gate < 3 && rrow < nrows;
        pc_gate2[k] = gate == 2;
        pc_lds[k] = ((gate < 3 ? gate : 0) * RM + rrow) * UW + 8 * quarter;
        pc_shadow[k] = (size_t)(row0 + rrow) * T * G + (size_t)(gate < 3 ? gate : 0) * H + j0 + 8 * quarter;
    }
    // cell operands of step gmax - 1; step t - 1's are requested during step t
    float s_r, s_z, s_n, s_hn, hprev, gout;
    {
        const int t0 = gmax > 0 ? gmax - 1 : 0;
        const float* sv = p_sv + (size_t)t0 * 4 * H;
        s_r = sv[0];
        s_z = sv[H];
        s_n = sv[2 * H];
        s_hn = sv[3 * H];
        hprev = p_h[(size_t)t0 * H];
        gout = p_g[(size_t)t0 * H];
    }
    __syncthreads();

    for (int t = gmax - 1; t >= -1; --t) {
        const bool need_mm = t + 1 < gmax;               // row t + 1 holds gradients of a live step
        u32x4 raw[KS];
        if (need_mm) {
            if (wave == 0 && !gp_wait_flags(flags, n_vslots, (unsigned)(gmax - t - 1), lane)) s_abort = 1;
            gp_lds_barrier();
            if (s_abort) {
                if (tid == 0) __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            const unsigned off = ((t + 1) & 1) * par_bytes + (unsigned)((li < nrows ? li : 0) * 64);
#pragma unroll
            for (int i = 0; i < KS; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off + rd_off[i], 0, 16);
        }
        // the next step's cell operands (first touch: HBM latency), queued BEHIND the hand-off loads
        float s_r1, s_z1, s_n1, s_hn1, hprev1, gout1;
        {
            const int t1 = t > 0 ? t - 1 : 0;
            const float* sv = p_sv + (size_t)t1 * 4 * H;
            s_r1 = sv[0];
            s_z1 = sv[H];
            s_n1 = sv[2 * H];
            s_hn1 = sv[3 * H];
            hprev1 = p_h[(size_t)t1 * H];
            gout1 = p_g[(size_t)t1 * H];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (need_mm) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 acc3[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // KS is a multiple of 3
#pragma unroll
                for (int i = 0; i < KS; ++i)
                    acc3[i % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf8(raw[i]), fb[u][i], acc3[i % 3], 0, 0, 0);
                const f32x4 acc = (acc3[0] + acc3[1]) + acc3[2];
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave][u][(4 * q + r) * GT + li] = acc[r];
            }
            gp_lds_barrier();
        }
        {
            const float dstate = need_mm ? mg_gru_dstate(carry, red[0][eu][ee], red[1][eu][ee], red[2][eu][ee], red[3][eu][ee]) : carry;
            float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, c = dstate;
            if (t >= 0 && t < len) {
                const mg_gru_cell_grad g = mg_gru_cell_bwd(dstate, gout, s_r, s_z, s_n, s_hn, hprev);
                dr = g.dr; dz = g.dz; dn = g.dn; dnr = g.dnr; c = g.carry;
            }
            carry = c;
            res[0][bl][jl] = dr;
            res[1][bl][jl] = dz;
            res[2][bl][jl] = dn;
            res[3][bl][jl] = dnr;
            pub[0][bl][jl] = mg_f2bf(dr);
            pub[1][bl][jl] = mg_f2bf(dz);
            pub[2][bl][jl] = mg_f2bf(dnr);
            pub[3][bl][jl] = mg_f2bf(dn);
        }
        if (t < 0) {
            if (mine) dh0[(size_t)b_own * H + j] = carry;
            break;
        }
        gp_lds_barrier();
        if (wave <= 1) {
            // wave 0 publishes the slot's 3 R x 64 contiguous bytes of the ring, wave 1 writes the bf16 shadow for the weight-gradient GEMM
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
                        *reinterpret_cast<u32x4*>(dhproj_bf + pc_shadow[k] + (size_t)t * G) = v;
                        if (dxproj_bf) {                 // (dr, dz, dn): the n plane differs from the hand-off's dn r
                            const u32x4 vx = pc_gate2[k] ? *reinterpret_cast<const u32x4*>(&pub[0][0][0] + pc_lds[k] + RM * UW) : v;
                            *reinterpret_cast<u32x4*>(dxproj_bf + pc_shadow[k] + (size_t)t * G) = vx;
                        }
                    }
                }
            }
            if (wave == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) gp_store_flag(flags + slot, (unsigned)(gmax - t), one_xcd);
            }
        }
        if (wave >= 2 && dxproj) {                      // fp32 results (optional): waves 2 and 3 only (see the forward kernel)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int e = (tid - 128) + 128 * half, rb = e >> 5, cj = e & 31;
                if (rb < nrows) {
                    const size_t row = (size_t)(row0 + rb) * T + t;
                    float* dx = dxproj + row * G + j0 + cj;
                    float* dhp = dhproj + row * G + j0 + cj;
                    const float dr = res[0][rb][cj], dz = res[1][rb][cj];
                    dx[0] = dr;  dx[H] = dz;  dx[2 * H] = res[2][rb][cj];
                    dhp[0] = dr; dhp[H] = dz; dhp[2 * H] = res[3][rb][cj];
                }
            }
        }
        s_r = s_r1; s_z = s_z1; s_n = s_n1; s_hn = s_hn1;
        hprev = hprev1; gout = gout1;
    }
}

// =====================================================================================================================
// fp32 parity mode, persistent: the same group / slot scheme with exact-fp32 operands - fp32 hand-off tiles (64 bytes per
// item and slot), v_mfma_f32_16x16x4_f32 with the launch-per-step kernels' block order (wave w owns the 16-deep blocks
// w, w + 4, ... of the contraction; products and sums in the same sequence) and the same cell code (gru_cell.h), so the results
// are bit-identical to gru_fwd_step_kernel / gru_bwd_step_kernel - which is how the hand-off is tested.  One M tile: groups
// of at most 16 items (B <= 128).  The exact-fp32 MFMA rate (96 MFMAs of 32 cycles per wave and step) bounds the step at
// about 1.3 us, against 8.8 / 9.5 us for a launch per step.
// =====================================================================================================================
template <int KS>          // KS = H / 64 contraction blocks per wave
__global__ __launch_bounds__(256) void gru_fwd_persist_f32_kernel(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                                  const float* __restrict__ b_hh, const int64_t* __restrict__ seq_len,
                                                                  int B, int T, int H, int R, float* __restrict__ hstate,
                                                                  float* __restrict__ out, float* __restrict__ saved, unsigned* sync,
                                                                  float* ring, int force_sc1) {
    __shared__ float red[4][3][GT * GT];
    __shared__ __attribute__((aligned(16))) float hb[GT][GT];
    __shared__ float res[5][GT * GT];              // out, r, z, n, hn of the step for waves 2 and 3 (h itself goes through hb)
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
    // W_hh fragments: block i of this wave = columns 16 (wave + 4 i) + 4 q .. + 3 of rows j0 + li of the three gates
    f32x4 fr[KS], fz[KS], fn[KS];
    {
        const float* wr = w_hh + (size_t)(j0 + li) * H + 4 * q;
        const float* wz = wr + (size_t)H * H;
        const float* wn = wz + (size_t)H * H;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const int k0 = 16 * (wave + 4 * i);
            fr[i] = *reinterpret_cast<const f32x4*>(wr + k0);
            fz[i] = *reinterpret_cast<const f32x4*>(wz + k0);
            fn[i] = *reinterpret_cast<const f32x4*>(wn + k0);
        }
    }
    // ring: [2 (epoch parity)][8 groups][H / 16 slots][R items][16 units] f32: a contraction block IS a slot's tile
    const unsigned par_bytes = (unsigned)(GP_GROUPS * n_slots * R * 64);
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, (int)(2 * par_bytes), 0x00020000);
    const unsigned rd_base = (unsigned)(((group * n_slots + wave) * R) * 64 + 16 * q);
    const unsigned rd_blk = (unsigned)(4 * R * 64);
    const unsigned wr_base = (unsigned)(((group * n_slots + slot) * R) * 64);

    const int bl = tid >> 4, jl = tid & 15;
    const int j = j0 + jl;
    const float bhr = b_hh[j], bhz = b_hh[H + j], bhn = b_hh[2 * H + j];
    const bool mine = bl < nrows;
    const int b = row0 + (mine ? bl : 0);
    float hprev = hstate[((size_t)b * (T + 1)) * H + j];
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
    hb[bl][jl] = hprev;
    const float* xp = xproj + (size_t)b * T * 3 * H + j;
    float xr = xp[0], xz = xp[H], xn = xp[2 * H];
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
            if (tid == 0) __hip_atomic_store(status, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        u32x4 raw[KS];
        {
            const unsigned off = (t & 1) * par_bytes + rd_base + (unsigned)((li < nrows ? li : 0) * 64);
#pragma unroll
            for (int i = 0; i < KS; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off + i * rd_blk, 0, 16);
        }
        const int t1 = t + 1 < T ? t + 1 : t;
        const float xr1 = xp[(size_t)t1 * 3 * H], xz1 = xp[(size_t)t1 * 3 * H + H], xn1 = xp[(size_t)t1 * 3 * H + 2 * H];
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc_r = {0.f, 0.f, 0.f, 0.f}, acc_z = acc_r, acc_n = acc_r;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            union { u32x4 u; f32x4 f; } a;
            a.u = raw[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[e], fr[i][e], acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[e], fz[i][e], acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[e], fn[i][e], acc_n, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = (4 * q + r) * GT + li;
            red[wave][0][e] = acc_r[r];
            red[wave][1][e] = acc_z[r];
            red[wave][2][e] = acc_n[r];
        }
        gp_lds_barrier();
        {
            const int e = bl * GT + jl;
            const float hr = mg_gru_sum4(red[0][0][e], red[1][0][e], red[2][0][e], red[3][0][e], bhr);
            const float hz = mg_gru_sum4(red[0][1][e], red[1][1][e], red[2][1][e], red[3][1][e], bhz);
            const float hn = mg_gru_sum4(red[0][2][e], red[1][2][e], red[2][2][e], red[3][2][e], bhn);
            const mg_gru_cell_out c = mg_gru_cell_exact(xr, xz, xn, hr, hz, hn, hprev);
            const bool active = t < len;
            hprev = active ? c.hnew : hprev;
            hb[bl][jl] = hprev;
            res[0][e] = active ? c.hnew : 0.f;
            res[1][e] = c.r;
            res[2][e] = c.z;
            res[3][e] = c.n;
            res[4][e] = hn;
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
                    hstate[((size_t)bb * (T + 1) + t + 1) * H + cj] = hb[rb][e & 15];
                    out[row * H + cj] = res[0][e];
                    float* sv = saved + row * 4 * H + cj;
                    sv[0] = res[1][e];
                    sv[H] = res[2][e];
                    sv[2 * H] = res[3][e];
                    sv[3 * H] = res[4][e];
                }
            }
        }
        xr = xr1;
        xz = xz1;
        xn = xn1;
    }
    // beyond the group's longest sequence: state frozen, outputs zero, gate values (never read by the backward there) zero
    for (int t = gmax; t < T; ++t) {
        if (mine) {
            const size_t row = (size_t)b * T + t;
            hstate[((size_t)b * (T + 1) + t + 1) * H + j] = hprev;
            out[row * H + j] = 0.f;
            float* sv = saved + row * 4 * H;
            sv[j] = 0.f;
            sv[H + j] = 0.f;
            sv[2 * H + j] = 0.f;
            sv[3 * H + j] = 0.f;
        }
    }
}

template <int KS>          // KS = 3 H / 64 contraction blocks per wave
__global__ __launch_bounds__(256) void gru_bwd_persist_f32_kernel(const float* __restrict__ grad_out, const float* __restrict__ grad_hn,
                                                                  const float* __restrict__ hstate, const float* __restrict__ saved,
                                                                  const float* __restrict__ w_hh, const int64_t* __restrict__ seq_len,
                                                                  int B, int T, int H, int R, float* __restrict__ dxproj,
                                                                  float* __restrict__ dhproj, float* __restrict__ dh0, unsigned* sync,
                                                                  float* ring, int force_sc1) {
    __shared__ float red[4][GT * GT];
    __shared__ __attribute__((aligned(16))) float pub[3][GT][GT];   // dr, dz, dn r tiles on their way to the ring
    __shared__ float res[GT * GT];                                   // dn (dxproj's third gate) for waves 2 and 3
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
    const int G = 3 * H;
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
    // block i of this wave = gate rows 16 (wave + 4 i) + 4 q + e: W_hh[that row][j0 + li] as the B operand; its A tile sits in the
    // ring at (slot = block % n_slots, gate = block / n_slots)
    float fb[KS][4];
    unsigned rd_off[KS];
#pragma unroll
    for (int i = 0; i < KS; ++i) {
        const int blk = wave + 4 * i, gate = blk / n_slots, sl = blk - gate * n_slots;
#pragma unroll
        for (int e = 0; e < 4; ++e) fb[i][e] = w_hh[(size_t)(16 * blk + 4 * q + e) * H + j0 + li];
        rd_off[i] = (unsigned)((((group * n_slots + sl) * 3 + gate) * R) * 64 + 16 * q);
    }
    const unsigned par_bytes = (unsigned)(GP_GROUPS * n_slots * R * 192);
    const auto rs_ring = __builtin_amdgcn_make_buffer_rsrc((void*)ring, 0, (int)(2 * par_bytes), 0x00020000);
    const unsigned wr_base = (unsigned)(((group * n_slots + slot) * R) * 192);

    const int bl = tid >> 4, jl = tid & 15;
    const int j = j0 + jl;
    const bool mine = bl < nrows;
    const int b = row0 + (mine ? bl : 0);
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
    float carry = (mine && grad_hn) ? grad_hn[(size_t)b * H + j] : 0.f;
    const float* p_sv = saved + (size_t)b * T * 4 * H + j;
    const float* p_h = hstate + (size_t)b * (T + 1) * H + j;
    const float* p_g = grad_out + (size_t)b * T * H + j;
    for (int t = T - 1; t >= gmax; --t) {
        if (mine) {
            const size_t row = (size_t)b * T + t;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                dxproj[row * G + g * H + j] = 0.f;
                dhproj[row * G + g * H + j] = 0.f;
            }
        }
    }
    const int t0 = gmax > 0 ? gmax - 1 : 0;
    float s_r = p_sv[(size_t)t0 * 4 * H], s_z = p_sv[(size_t)t0 * 4 * H + H], s_n = p_sv[(size_t)t0 * 4 * H + 2 * H],
          s_hn = p_sv[(size_t)t0 * 4 * H + 3 * H], hprev = p_h[(size_t)t0 * H], gout = p_g[(size_t)t0 * H];
    __syncthreads();

    for (int t = gmax - 1; t >= -1; --t) {
        const bool need_mm = t + 1 < gmax;
        u32x4 raw[KS];
        if (need_mm) {
            if (wave == 0 && !gp_wait_flags(flags, n_slots, (unsigned)(gmax - t - 1), lane)) s_abort = 1;
            gp_lds_barrier();
            if (s_abort) {
                if (tid == 0) __hip_atomic_store(status, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            const unsigned off = ((t + 1) & 1) * par_bytes + (unsigned)((li < nrows ? li : 0) * 64);
#pragma unroll
            for (int i = 0; i < KS; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ring, off + rd_off[i], 0, 16);
        }
        const int t1 = t > 0 ? t - 1 : 0;
        const float s_r1 = p_sv[(size_t)t1 * 4 * H], s_z1 = p_sv[(size_t)t1 * 4 * H + H], s_n1 = p_sv[(size_t)t1 * 4 * H + 2 * H],
                    s_hn1 = p_sv[(size_t)t1 * 4 * H + 3 * H], hprev1 = p_h[(size_t)t1 * H], gout1 = p_g[(size_t)t1 * H];
        __builtin_amdgcn_sched_barrier(0);
        if (need_mm) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                union { u32x4 u; f32x4 f; } a;
                a.u = raw[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[e], fb[i][e], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][(4 * q + r) * GT + li] = acc[r];
            gp_lds_barrier();
        }
        {
            const int e = bl * GT + jl;
            // t + 1 == T: the per-step kernel adds a zero accumulator here; carry + 0 is carry (also for -0: results compare equal)
            const float dstate = need_mm ? mg_gru_dstate(carry, red[0][e], red[1][e], red[2][e], red[3][e]) : carry;
            float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, c = dstate;
            if (t >= 0 && t < len) {
                const mg_gru_cell_grad g = mg_gru_cell_bwd(dstate, gout, s_r, s_z, s_n, s_hn, hprev);
                dr = g.dr; dz = g.dz; dn = g.dn; dnr = g.dnr; c = g.carry;
            }
            carry = c;
            pub[0][bl][jl] = dr;
            pub[1][bl][jl] = dz;
            pub[2][bl][jl] = dnr;
            res[e] = dn;
        }
        if (t < 0) {
            if (mine) dh0[(size_t)b * H + j] = carry;
            break;
        }
        gp_lds_barrier();
        if (wave == 0) {
            // the slot's tile: 3 gates x R items x 64 bytes, contiguous in the ring; piece p = (gate, item, quarter)
            for (int pc = lane; pc < 12 * R; pc += 64) {
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
                    const size_t row = (size_t)(row0 + rb) * T + t;
                    float* dx = dxproj + row * G + j0 + cc;
                    float* dhp = dhproj + row * G + j0 + cc;
                    const float dr = pub[0][rb][cc], dz = pub[1][rb][cc];
                    dx[0] = dr;  dx[H] = dz;  dx[2 * H] = res[e];
                    dhp[0] = dr; dhp[H] = dz; dhp[2 * H] = pub[2][rb][cc];
                }
            }
        }
        s_r = s_r1; s_z = s_z1; s_n = s_n1; s_hn = s_hn1; hprev = hprev1; gout = gout1;
    }
}

extern "C" {

// workspace: [step flags 8 x 32 words | XCC ids 8 x 32 words | status word + 3 pad] [hand-off ring: 2 parities x 8 groups x
// R items x 16 H bytes]. The widest user is the fp32 LSTM backward (4 H fp32 per item); the fp32 GRU backward needs 3 H fp32, the
// bf16 backwards 3 H / 4 H bf16, the forward kernels H units.
size_t mg_gru_persist_workspace_bytes(int B, int H) {
    if (B <= 0 || H <= 0) return GP_RING_OFFSET;
    return GP_RING_OFFSET + (size_t)2 * GP_GROUPS * mg_ceil_div(B, GP_GROUPS) * 16 * H;
}

int mg_gru_persist_supported(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0) return 0;
    if (H % 128 != 0 || H > 128 * GP_KSTEPS) return 0;
    if (mg_ceil_div(B, GP_GROUPS) > 32) return 0;
    if ((size_t)B * (T + 1) * (size_t)H * 3 * 2 >= ((size_t)1 << 31)) return 0;     // 32-bit buffer offsets (backward: 3 H wide)
    return gp_device_holds((long)GP_GROUPS * (H / GT));
}

int mg_gru_persist_status(void* workspace, void* stream) {
    unsigned st = 0;
    if (hipMemcpyAsync(&st, (const unsigned*)workspace + GP_FLAG_WORDS, sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream) !=
            hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        mg_set_error("mg_gru_persist_status: could not read the status word");
        return MG_ELAUNCH;
    }
    if (st != 0) {
        (void)hipMemsetAsync((unsigned*)workspace + GP_FLAG_WORDS, 0, 16, (hipStream_t)stream);     // sticky until reported
        mg_set_error("persistent GRU kernel timed out waiting for another workgroup (status %u): results are invalid", st);
        return MG_ELAUNCH;
    }
    return MG_OK;
}

int mg_gru_fwd_persist_bf16(const float* xproj, const uint16_t* w_hh_bf, int ldw, const float* b_hh, const int64_t* seq_len, int B, int T,
                            int H, float* hstate, uint16_t* hstate_bf, float* out, float* saved, void* workspace,
                            size_t workspace_bytes, void* stream) {
    return mg_gru_fwd_persist_rows_bf16(xproj, nullptr, 0, w_hh_bf, ldw, b_hh, seq_len, B, T, H, hstate, hstate_bf, out, saved, workspace,
                                        workspace_bytes, stream);
}

// The same with the input projections read through a row map: frame (b, t) takes row xrows[b * T + t] of the table xproj
// [n_rows, 3H] (every entry in [0, n_rows): the caller's map, e.g. upsample_to_repetitions' with the padding frames on a zero row).
// xrows == NULL: xproj is [B, T, 3H] itself.
int mg_gru_fwd_persist_rows_bf16(const float* xproj, const int32_t* xrows, int64_t n_rows, const uint16_t* w_hh_bf, int ldw, const float* b_hh,
                                 const int64_t* seq_len, int B, int T, int H, float* hstate, uint16_t* hstate_bf, float* out, float* saved,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    return mg_gru_fwd_persist_out_bf16(xproj, xrows, n_rows, w_hh_bf, ldw, b_hh, seq_len, B, T, H, hstate, hstate_bf, out, nullptr, saved,
                                       workspace, workspace_bytes, stream);
}

// The same with an optional bf16 copy of `out` ([B, T, H], zero past each item's length) written by the recurrence itself.
int mg_gru_fwd_persist_out_bf16(const float* xproj, const int32_t* xrows, int64_t n_rows, const uint16_t* w_hh_bf, int ldw, const float* b_hh,
                                const int64_t* seq_len, int B, int T, int H, float* hstate, uint16_t* hstate_bf, float* out, uint16_t* out_bf,
                                float* saved, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(xproj && w_hh_bf && b_hh && hstate && hstate_bf && out && saved && B > 0 && T > 0 && H > 0 && (!xrows || n_rows > 0),
                 "mg_gru_fwd_persist_bf16: bad arguments (B=%d T=%d H=%d)", B, T, H);
    MG_CHECK_ARG(mg_gru_persist_supported(B, T, H) && ldw >= H && ldw % 8 == 0,
                 "mg_gru_fwd_persist_bf16: unsupported shape (B=%d T=%d H=%d ldw=%d): needs H %% 128 == 0, H <= 512, B <= 256", B, T, H, ldw);
    MG_CHECK_ARG((((uintptr_t)w_hh_bf | (uintptr_t)hstate_bf | (uintptr_t)workspace) % 16) == 0,
                 "mg_gru_fwd_persist_bf16: bf16 buffers and workspace must be 16-byte aligned");
    if (!workspace || workspace_bytes < mg_gru_persist_workspace_bytes(B, H)) {
        mg_set_error("mg_gru_fwd_persist_bf16: workspace of %zu bytes needed, got %zu", mg_gru_persist_workspace_bytes(B, H), workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)GP_FLAG_WORDS * sizeof(unsigned), st) != hipSuccess) {
        mg_set_error("mg_gru_fwd_persist_bf16: memset failed");
        return MG_ELAUNCH;
    }
    const int R = (int)mg_ceil_div(B, GP_GROUPS);
    const unsigned grid = (unsigned)(GP_GROUPS * (H / GT));
#define GP_FWD(MT, KS)                                                                                                                  \
    hipLaunchKernelGGL((gru_fwd_persist_kernel<MT, KS>), dim3(grid), dim3(384), 0, st, xproj, w_hh_bf, ldw, b_hh, seq_len, B, T, H, R, hstate, \
                       hstate_bf, out, saved, (unsigned*)workspace, (uint16_t*)((char*)workspace + GP_RING_OFFSET), g_mg_tuning[MG_TUNE_GRU_HANDOFF], xrows, out_bf)
#define GP_FWD_KS(MT)            \
    switch (H / 128) {           \
        case 1: GP_FWD(MT, 1); break; \
        case 2: GP_FWD(MT, 2); break; \
        case 3: GP_FWD(MT, 3); break; \
        default: GP_FWD(MT, 4); break; \
    }
    if (R <= 16) {
        GP_FWD_KS(1)
    } else {
        GP_FWD_KS(2)
    }
    MG_CHECK_LAUNCH("mg_gru_fwd_persist_bf16");
    return MG_OK;
}

int mg_gru_bwd_persist_bf16(const float* grad_out, const float* grad_hn, const float* hstate, const float* saved, const uint16_t* w_hh_t_bf,
                            int ldt, const int64_t* seq_len, int B, int T, int H, float* dxproj, float* dhproj, uint16_t* dhproj_bf,
                            uint16_t* dxproj_bf, float* dh0, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(grad_out && hstate && saved && w_hh_t_bf && dhproj_bf && dh0 && B > 0 && T > 0 && H > 0 && (!dxproj == !dhproj) &&
                     (dxproj || dxproj_bf),
                 "mg_gru_bwd_persist_bf16: bad arguments (B=%d T=%d H=%d)", B, T, H);
    MG_CHECK_ARG(mg_gru_persist_supported(B, T, H) && ldt >= 3 * H && ldt % 8 == 0,
                 "mg_gru_bwd_persist_bf16: unsupported shape (B=%d T=%d H=%d ldt=%d): needs H %% 128 == 0, H <= 512, B <= 256", B, T, H, ldt);
    MG_CHECK_ARG((((uintptr_t)w_hh_t_bf | (uintptr_t)dhproj_bf | (uintptr_t)dxproj_bf | (uintptr_t)workspace) % 16) == 0,
                 "mg_gru_bwd_persist_bf16: bf16 buffers and workspace must be 16-byte aligned");
    if (!workspace || workspace_bytes < mg_gru_persist_workspace_bytes(B, H)) {
        mg_set_error("mg_gru_bwd_persist_bf16: workspace of %zu bytes needed, got %zu", mg_gru_persist_workspace_bytes(B, H), workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)GP_FLAG_WORDS * sizeof(unsigned), st) != hipSuccess) {
        mg_set_error("mg_gru_bwd_persist_bf16: memset failed");
        return MG_ELAUNCH;
    }
    const int R = (int)mg_ceil_div(B, GP_GROUPS);
    const unsigned grid = (unsigned)(GP_GROUPS * (H / GT));
#define GP_BWD(MT, KS)                                                                                                                      \
    hipLaunchKernelGGL((gru_bwd_persist_kernel<MT, KS>), dim3(grid), dim3(256), 0, st, grad_out, grad_hn, hstate, saved, w_hh_t_bf, ldt, seq_len, \
                       B, T, H, R, dxproj, dhproj, dhproj_bf, dxproj_bf, dh0, (unsigned*)workspace, (uint16_t*)((char*)workspace + GP_RING_OFFSET),           \
                       g_mg_tuning[MG_TUNE_GRU_HANDOFF])
#define GP_BWD_KS(MT)                  \
    switch (H / 128) {                 \
        case 1: GP_BWD(MT, 3); break;  \
        case 2: GP_BWD(MT, 6); break;  \
        case 3: GP_BWD(MT, 9); break;  \
        default: GP_BWD(MT, 12); break; \
    }
    // Wide slots (gru_bwd_persist_wide_kernel): 16 groups of at most 8 items, two per XCD - half the hand-off lines per step and slot.
    // MG_TUNE_GRU_HANDOFF bit 2 (4): the narrow kernel for A/B.
    const int Rw = (int)mg_ceil_div(B, 2 * GP_GROUPS);
    if (Rw <= 8 && !(g_mg_tuning[MG_TUNE_GRU_HANDOFF] & 4)) {
#define GP_BWDW(KS)                                                                                                                          \
    hipLaunchKernelGGL((gru_bwd_persist_wide_kernel<KS>), dim3(grid), dim3(256), 0, st, grad_out, grad_hn, hstate, saved, w_hh_t_bf, ldt, seq_len, \
                       B, T, H, Rw, dxproj, dhproj, dhproj_bf, dxproj_bf, dh0, (unsigned*)workspace, (uint16_t*)((char*)workspace + GP_RING_OFFSET),  \
                       g_mg_tuning[MG_TUNE_GRU_HANDOFF] & 3)
        switch (H / 128) {
            case 1: GP_BWDW(3); break;
            case 2: GP_BWDW(6); break;
            case 3: GP_BWDW(9); break;
            default: GP_BWDW(12); break;
        }
#undef GP_BWDW
    } else if (R <= 16) {
        GP_BWD_KS(1)
    } else {
        GP_BWD_KS(2)
    }
    MG_CHECK_LAUNCH("mg_gru_bwd_persist_bf16");
    return MG_OK;
}

int mg_gru_persist_f32_supported(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0 || g_mg_tuning[MG_TUNE_PERSISTENT] == 1) return 0;
    if (H % 64 != 0 || H > 512 || H < 256) return 0;                       // H = 64 / 128: gru_small.hip
    if (mg_ceil_div(B, GP_GROUPS) > 16) return 0;                           // one 16-row MFMA tile per group
    return gp_device_holds((long)GP_GROUPS * (H / GT));
}

int mg_gru_fwd_persist_f32(const float* xproj, const float* w_hh, const float* b_hh, const int64_t* seq_len, int B, int T, int H,
                           float* hstate, float* out, float* saved, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(xproj && w_hh && b_hh && hstate && out && saved, "mg_gru_fwd_persist_f32: null argument");
    MG_CHECK_ARG(mg_gru_persist_f32_supported(B, T, H), "mg_gru_fwd_persist_f32: unsupported shape (B=%d T=%d H=%d)", B, T, H);
    MG_CHECK_ARG((((uintptr_t)w_hh | (uintptr_t)workspace) % 16) == 0, "mg_gru_fwd_persist_f32: w_hh and workspace must be 16-byte aligned");
    if (!workspace || workspace_bytes < mg_gru_persist_workspace_bytes(B, H)) {
        mg_set_error("mg_gru_fwd_persist_f32: workspace of %zu bytes needed, got %zu", mg_gru_persist_workspace_bytes(B, H), workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)GP_FLAG_WORDS * sizeof(unsigned), st) != hipSuccess) {
        mg_set_error("mg_gru_fwd_persist_f32: memset failed");
        return MG_ELAUNCH;
    }
    const int R = (int)mg_ceil_div(B, GP_GROUPS);
    const unsigned grid = (unsigned)(GP_GROUPS * (H / GT));
    float* ring = (float*)((char*)workspace + GP_RING_OFFSET);
    const int force = g_mg_tuning[MG_TUNE_GRU_HANDOFF];
#define GPF_FWD(KS) hipLaunchKernelGGL((gru_fwd_persist_f32_kernel<KS>), dim3(grid), dim3(256), 0, st, xproj, w_hh, b_hh, seq_len, B, T, H, R, hstate, out, saved, (unsigned*)workspace, ring, force)
    switch (H / 64) {
        case 4: GPF_FWD(4); break;
        case 5: GPF_FWD(5); break;
        case 6: GPF_FWD(6); break;
        case 7: GPF_FWD(7); break;
        default: GPF_FWD(8); break;
    }
    MG_CHECK_LAUNCH("mg_gru_fwd_persist_f32");
    return MG_OK;
}

int mg_gru_bwd_persist_f32(const float* grad_out, const float* grad_hn, const float* hstate, const float* saved, const float* w_hh,
                           const int64_t* seq_len, int B, int T, int H, float* dxproj, float* dhproj, float* dh0, void* workspace,
                           size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(grad_out && hstate && saved && w_hh && dxproj && dhproj && dh0, "mg_gru_bwd_persist_f32: null argument");
    MG_CHECK_ARG(mg_gru_persist_f32_supported(B, T, H), "mg_gru_bwd_persist_f32: unsupported shape (B=%d T=%d H=%d)", B, T, H);
    MG_CHECK_ARG(((uintptr_t)workspace % 16) == 0, "mg_gru_bwd_persist_f32: workspace must be 16-byte aligned");
    if (!workspace || workspace_bytes < mg_gru_persist_workspace_bytes(B, H)) {
        mg_set_error("mg_gru_bwd_persist_f32: workspace of %zu bytes needed, got %zu", mg_gru_persist_workspace_bytes(B, H), workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)GP_FLAG_WORDS * sizeof(unsigned), st) != hipSuccess) {
        mg_set_error("mg_gru_bwd_persist_f32: memset failed");
        return MG_ELAUNCH;
    }
    const int R = (int)mg_ceil_div(B, GP_GROUPS);
    const unsigned grid = (unsigned)(GP_GROUPS * (H / GT));
    float* ring = (float*)((char*)workspace + GP_RING_OFFSET);
    const int force = g_mg_tuning[MG_TUNE_GRU_HANDOFF];
#define GPF_BWD(KS) hipLaunchKernelGGL((gru_bwd_persist_f32_kernel<KS>), dim3(grid), dim3(256), 0, st, grad_out, grad_hn, hstate, saved, w_hh, seq_len, B, T, H, R, dxproj, dhproj, dh0, (unsigned*)workspace, ring, force)
    switch (H / 64) {
        case 4: GPF_BWD(12); break;
        case 5: GPF_BWD(15); break;
        case 6: GPF_BWD(18); break;
        case 7: GPF_BWD(21); break;
        default: GPF_BWD(24); break;
    }
    MG_CHECK_LAUNCH("mg_gru_bwd_persist_f32");
    return MG_OK;
}

}  // extern "C"

#ifdef MG_STAMPS
extern "C" int mg_diag_read_stamps_gpb(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_gpb), bytes < sizeof(g_stamps_gpb) ? bytes : sizeof(g_stamps_gpb), 0, hipMemcpyDeviceToHost);
}
extern "C" int mg_diag_read_stamps_gp(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_gp), bytes < sizeof(g_stamps_gp) ? bytes : sizeof(g_stamps_gp), 0, hipMemcpyDeviceToHost);
}
#endif
