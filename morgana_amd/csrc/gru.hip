// K3 - GRU recurrence behind RecurrentCuDNNWrapper (reference: morgana/utils.py:345-393 around torch.nn.GRU,
// used at models/f0_test_model.py:32-39; gate order r, z, n).
//
// Split of the work (all matmul-shaped work on MFMA):
//   * input projection  xproj = x W_ih^T + b_ih   one big GEMM over all B*T frames (mg_linear_fwd_*), done by the caller
//   * recurrence        T dependent steps of [B,H] x [H,3H]: one launch per step (this file).  Each workgroup owns a
//                       16 (batch) x 16 (hidden unit) tile, its 4 waves split the contraction, fragments are loaded
//                       straight from L2 as 16-byte lanes (h_{t-1} and W_hh are L2 resident: 128 KB + 3 MB at H=512),
//                       v_mfma_f32_16x16x4_f32 accumulates the three gate pre-activations, a 12 KB LDS exchange sums
//                       the 4 partials and the cell non-linearity is applied in the same kernel.
//   * BPTT              one launch per step: dstate_t = carry + dhproj_{t+1} W_hh (MFMA), then the gate derivatives
//   * dW_ih, dW_hh, dx  big GEMMs over all frames after the loop (mg_linear_wgrad_* / mg_linear_dgrad_*), caller side
//
// The reference sorts by length and packs; packing only restricts item b to its first seq_len[b] steps, so here a
// per-item length mask freezes the state (h_n = state at the last valid step) and zeroes the padded outputs.
//
// 16x16x4 fragment trick: lane l = 16*q + i holds, for a 16-deep contraction block, the 4 consecutive values
// k = 4q .. 4q+3 of row i (one 16-byte load); MFMA number e consumes element e, i.e. k = 4q + e.  A and B use the
// same (q, e) <-> k map, so the block's 16 products are each taken exactly once.
#include "common.h"
#include "gru_cell.h"

#define GT 16  // tile edge (batch x hidden units)

__device__ __forceinline__ f32x4 ld4_guard(const float* p, int valid, bool vec) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid >= 4 && vec) {
        v = *reinterpret_cast<const f32x4*>(p);
    } else {
        if (valid > 0) v.x = p[0];
        if (valid > 1) v.y = p[1];
        if (valid > 2) v.z = p[2];
        if (valid > 3) v.w = p[3];
    }
    return v;
}

// One forward step t.  hstate [B, T+1, H]: slot t holds h_{t-1} (slot 0 = h0), slot t+1 receives h_t.
// NB > 0: the wave owns exactly NB contraction blocks (H == 64 NB): all of their fragments are requested up front
// (4 NB 16-byte loads per lane in flight), so the L2 round trip is paid once per step instead of once per block.
template <int NB>
__global__ __launch_bounds__(256) void gru_fwd_step_kernel(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                           const float* __restrict__ b_hh, const int64_t* __restrict__ seq_len,
                                                           int B, int T, int H, int t, float* __restrict__ hstate,
                                                           float* __restrict__ out, float* __restrict__ saved, int vec) {
    __shared__ float red[4][3][GT * GT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * GT, b0 = blockIdx.y * GT;
    const int brow = b0 + li;   // A-operand row (batch item) of this lane
    const int jrow = j0 + li;   // B-operand column (hidden unit) of this lane
    const float* hp = hstate + ((size_t)(brow < B ? brow : 0) * (T + 1) + t) * H;
    const float* wr = w_hh + (size_t)(jrow < H ? jrow : 0) * H;
    const float* wz = wr + (size_t)H * H;
    const float* wn = wz + (size_t)H * H;

    f32x4 acc_r = {0.f, 0.f, 0.f, 0.f}, acc_z = acc_r, acc_n = acc_r;
    // Software pipeline over the 16-deep contraction blocks this wave owns (wave, wave + 4, ...): the loads of block i + 1
    // are in flight while block i feeds the matrix pipe (h and W_hh come from L2; a dependent load is ~1 us otherwise).
    auto load_blk = [&](int k0, f32x4& a, f32x4& br, f32x4& bz, f32x4& bn) {
        const int k = k0 + 4 * q;
        const int valid = H - k;
        a = (brow < B) ? ld4_guard(hp + k, valid, vec) : f32x4{0.f, 0.f, 0.f, 0.f};
        br = bz = bn = f32x4{0.f, 0.f, 0.f, 0.f};
        if (jrow < H) {
            br = ld4_guard(wr + k, valid, vec);
            bz = ld4_guard(wz + k, valid, vec);
            bn = ld4_guard(wn + k, valid, vec);
        }
    };
    // Operands of the cell update (this thread's batch item / hidden unit), requested before the contraction so that
    // their HBM latency (xproj is streamed, never cached) hides under it.
    const int bl = tid >> 4, jl = tid & 15;
    const int b = b0 + bl, j = j0 + jl;
    const bool mine = b < B && j < H;
    const size_t row = (size_t)(mine ? b : 0) * T + t;
    const float* xp = xproj + row * 3 * H;
    const int jj = mine ? j : 0;
    const float xr = xp[jj], xz = xp[H + jj], xn = xp[2 * H + jj];
    const float hprev = hstate[((size_t)(mine ? b : 0) * (T + 1) + t) * H + jj];
    const float bhr = b_hh[jj], bhz = b_hh[H + jj], bhn = b_hh[2 * H + jj];
    const bool active = seq_len ? ((int64_t)t < seq_len[mine ? b : 0]) : true;

    if (NB > 0) {
        f32x4 fa[NB > 0 ? NB : 1], fr[NB > 0 ? NB : 1], fz[NB > 0 ? NB : 1], fn[NB > 0 ? NB : 1];
#pragma unroll
        for (int i = 0; i < NB; ++i) load_blk(wave * 16 + 64 * i, fa[i], fr[i], fz[i], fn[i]);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fr[i][e], acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fz[i][e], acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fn[i][e], acc_n, 0, 0, 0);
            }
    } else {
        f32x4 a0, r0, z0, n0, a1, r1, z1, n1;
        int k0 = wave * 16;
        if (k0 < H) load_blk(k0, a0, r0, z0, n0);
        for (; k0 < H; k0 += 64) {
            const bool more = k0 + 64 < H;
            if (more) load_blk(k0 + 64, a1, r1, z1, n1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], r0[e], acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], z0[e], acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], n0[e], acc_n, 0, 0, 0);
            }
            if (more) { a0 = a1; r0 = r1; z0 = z1; n0 = n1; }
        }
    }
    // C/D layout 16x16: col = lane&15 (hidden unit), row = 4*(lane>>4) + reg (batch item).
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = (4 * q + r) * GT + li;
        red[wave][0][e] = acc_r[r];
        red[wave][1][e] = acc_z[r];
        red[wave][2][e] = acc_n[r];
    }
    __syncthreads();
    if (mine) {
        const int e = bl * GT + jl;
        const float hr = mg_gru_sum4(red[0][0][e], red[1][0][e], red[2][0][e], red[3][0][e], bhr);
        const float hz = mg_gru_sum4(red[0][1][e], red[1][1][e], red[2][1][e], red[3][1][e], bhz);
        const float hn = mg_gru_sum4(red[0][2][e], red[1][2][e], red[2][2][e], red[3][2][e], bhn);
        const mg_gru_cell_out c = mg_gru_cell_exact(xr, xz, xn, hr, hz, hn, hprev);
        const float r = c.r, z = c.z, n = c.n, hnew = c.hnew;
        hstate[((size_t)b * (T + 1) + t + 1) * H + j] = active ? hnew : hprev;
        out[row * H + j] = active ? hnew : 0.f;
        float* sv = saved + row * 4 * H;
        sv[j] = r;
        sv[H + j] = z;
        sv[2 * H + j] = n;
        sv[3 * H + j] = hn;
    }
}

// One backward step.  t in [0, T): dstate_t = carry + dhproj[:, t+1, :] W_hh (skipped at t == T-1), then gate
// derivatives of step t; carry <- dh_t * z_t (or dstate_t for finished items).  t == -1: only the matmul, result
// (the gradient of h0) goes to dh0.
template <int NB>
__global__ __launch_bounds__(256) void gru_bwd_step_kernel(const float* __restrict__ grad_out, const float* __restrict__ hstate,
                                                           const float* __restrict__ saved, const float* __restrict__ w_hh,
                                                           const int64_t* __restrict__ seq_len, int B, int T, int H, int t,
                                                           float* __restrict__ dxproj, float* __restrict__ dhproj,
                                                           float* __restrict__ carry, float* __restrict__ dh0, int vec) {
    __shared__ float red[4][GT * GT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * GT, b0 = blockIdx.y * GT;
    const int G = 3 * H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // cell-update operands of this thread, requested early (see the forward kernel)
    const int bl = tid >> 4, jl = tid & 15;
    const int b = b0 + bl, j = j0 + jl;
    const bool mine = b < B && j < H && t >= 0;
    const size_t row = (size_t)(mine ? b : 0) * T + (t >= 0 ? t : 0);
    const int jj = mine ? j : 0;
    const float* sv = saved + row * 4 * H;
    const float s_r = sv[jj], s_z = sv[H + jj], s_n = sv[2 * H + jj], s_hn = sv[3 * H + jj];
    const float hprev = hstate[((size_t)(mine ? b : 0) * (T + 1) + (t >= 0 ? t : 0)) * H + jj];
    const float gout = grad_out[row * H + jj];
    const float cin = carry[(size_t)((b < B) ? b : 0) * H + ((j < H) ? j : 0)];
    if (t + 1 < T) {
        const int brow = b0 + li;
        const int jcol = j0 + li;
        const float* dp = dhproj + ((size_t)(brow < B ? brow : 0) * T + (t + 1)) * G;
        auto load_blk = [&](int g0, f32x4& a, float (&bv)[4]) {
            const int g = g0 + 4 * q;
            a = (brow < B) ? ld4_guard(dp + g, G - g, vec) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (jcol < H && g + e < G) ? w_hh[(size_t)(g + e) * H + jcol] : 0.f;
        };
        if (NB > 0) {
            f32x4 fa[NB > 0 ? NB : 1];
            float fb[NB > 0 ? NB : 1][4];
#pragma unroll
            for (int i = 0; i < NB; ++i) load_blk(wave * 16 + 64 * i, fa[i], fb[i]);
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[i][e], acc, 0, 0, 0);
        } else {
            f32x4 a0, a1;
            float b0v[4], b1v[4];
            int g0 = wave * 16;
            if (g0 < G) load_blk(g0, a0, b0v);
            for (; g0 < G; g0 += 64) {
                const bool more = g0 + 64 < G;
                if (more) load_blk(g0 + 64, a1, b1v);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], b0v[e], acc, 0, 0, 0);
                if (more) {
                    a0 = a1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) b0v[e] = b1v[e];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * q + r) * GT + li] = acc[r];
    __syncthreads();
    if (b < B && j < H) {
        const int e = bl * GT + jl;
        const float dstate = mg_gru_dstate(cin, red[0][e], red[1][e], red[2][e], red[3][e]);
        if (t < 0) {
            dh0[(size_t)b * H + j] = dstate;
            return;
        }
        const bool active = seq_len ? ((int64_t)t < seq_len[b]) : true;
        float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, c = dstate;
        if (active) {
            const mg_gru_cell_grad g = mg_gru_cell_bwd(dstate, gout, s_r, s_z, s_n, s_hn, hprev);
            dr = g.dr; dz = g.dz; dn = g.dn; dnr = g.dnr; c = g.carry;
        }
        float* dx = dxproj + row * G;
        float* dhp = dhproj + row * G;
        dx[j] = dr;  dx[H + j] = dz;  dx[2 * H + j] = dn;
        dhp[j] = dr; dhp[H + j] = dz; dhp[2 * H + j] = dnr;
        carry[(size_t)b * H + j] = c;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// bf16-operand recurrence (throughput mode).  Same tiles, same 4-way K split, same cell arithmetic in fp32; only the two
// matmul operands are bf16: a bf16 shadow of the state (written by each step next to the fp32 one) against a bf16 copy of
// W_hh, on v_mfma_f32_16x16x32_bf16 (lane l holds row l & 15, k = 8 (l >> 4) .. + 7: one 16-byte load per fragment).
// Per step a workgroup pulls half the bytes of the fp32 form through L2 and spends 12 instead of 96 MFMAs per wave.
// Needs H % 128 == 0 (each wave owns H / 4 contraction columns in 32-deep steps).
// ---------------------------------------------------------------------------------------------------------------------
typedef __bf16 gbf8 __attribute__((ext_vector_type(8)));
#define GB_MAXSTEPS 8                              // H / 128 MFMA k-steps per wave: H <= 1024

__global__ __launch_bounds__(256) void gru_fwd_step_bf16_kernel(const float* __restrict__ xproj, const uint16_t* __restrict__ w_bf, int ldw,
                                                                const float* __restrict__ b_hh, const int64_t* __restrict__ seq_len,
                                                                int B, int T, int H, int t, float* __restrict__ hstate,
                                                                uint16_t* __restrict__ hstate_bf, float* __restrict__ out,
                                                                float* __restrict__ saved) {
    __shared__ float red[4][3][GT * GT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * GT, b0 = blockIdx.y * GT;
    const int brow = b0 + li, jrow = j0 + li;
    const uint16_t* hp = hstate_bf + ((size_t)(brow < B ? brow : 0) * (T + 1) + t) * H;
    const uint16_t* wr = w_bf + (size_t)(jrow < H ? jrow : 0) * ldw;
    const uint16_t* wz = wr + (size_t)H * ldw;
    const uint16_t* wn = wz + (size_t)H * ldw;
    const int n_steps = H / 128;
    const int kbase = wave * (H / 4) + 8 * q;

    const int bl = tid >> 4, jl = tid & 15;
    const int b = b0 + bl, j = j0 + jl;
    const bool mine = b < B && j < H;
    const size_t row = (size_t)(mine ? b : 0) * T + t;
    const float* xp = xproj + row * 3 * H;
    const int jj = mine ? j : 0;
    const float xr = xp[jj], xz = xp[H + jj], xn = xp[2 * H + jj];
    const float hprev = hstate[((size_t)(mine ? b : 0) * (T + 1) + t) * H + jj];
    const float bhr = b_hh[jj], bhz = b_hh[H + jj], bhn = b_hh[2 * H + jj];
    const bool active = seq_len ? ((int64_t)t < seq_len[mine ? b : 0]) : true;

    gbf8 fa[GB_MAXSTEPS], fr[GB_MAXSTEPS], fz[GB_MAXSTEPS], fn[GB_MAXSTEPS];
#pragma unroll
    for (int i = 0; i < GB_MAXSTEPS; ++i)
        if (i < n_steps) {
            const int k = kbase + 32 * i;
            fa[i] = *reinterpret_cast<const gbf8*>(hp + k);
            fr[i] = *reinterpret_cast<const gbf8*>(wr + k);
            fz[i] = *reinterpret_cast<const gbf8*>(wz + k);
            fn[i] = *reinterpret_cast<const gbf8*>(wn + k);
        }
    f32x4 acc_r = {0.f, 0.f, 0.f, 0.f}, acc_z = acc_r, acc_n = acc_r;
#pragma unroll
    for (int i = 0; i < GB_MAXSTEPS; ++i)
        if (i < n_steps) {
            acc_r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fr[i], acc_r, 0, 0, 0);
            acc_z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fz[i], acc_z, 0, 0, 0);
            acc_n = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fn[i], acc_n, 0, 0, 0);
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = (4 * q + r) * GT + li;
        red[wave][0][e] = acc_r[r];
        red[wave][1][e] = acc_z[r];
        red[wave][2][e] = acc_n[r];
    }
    __syncthreads();
    if (mine) {
        const int e = bl * GT + jl;
        const float hr = mg_gru_sum4(red[0][0][e], red[1][0][e], red[2][0][e], red[3][0][e], bhr);
        const float hz = mg_gru_sum4(red[0][1][e], red[1][1][e], red[2][1][e], red[3][1][e], bhz);
        const float hn = mg_gru_sum4(red[0][2][e], red[1][2][e], red[2][2][e], red[3][2][e], bhn);
        const mg_gru_cell_out c = mg_gru_cell(xr, xz, xn, hr, hz, hn, hprev);
        const float r = c.r, z = c.z, n = c.n, hnew = c.hnew;
        const float hnext = active ? hnew : hprev;
        const size_t nxt = ((size_t)b * (T + 1) + t + 1) * H + j;
        hstate[nxt] = hnext;
        hstate_bf[nxt] = mg_f2bf(hnext);
        out[row * H + j] = active ? hnew : 0.f;
        float* sv = saved + row * 4 * H;
        sv[j] = r;
        sv[H + j] = z;
        sv[2 * H + j] = n;
        sv[3 * H + j] = hn;
    }
}

// Backward step with bf16 matmul operands: dhproj_bf [B, T, 3H] is the bf16 shadow of dhproj (written here), wt_bf [H, ldt] =
// W_hh^T, so both fragments are 16-byte row loads; the contraction runs over the 3H gate rows (each wave 3H / 4 of them).
// Needs (3 H) % 128 == 0.
#define GBB_MAXSTEPS 24                            // 3 H / 128: H <= 1024

__global__ __launch_bounds__(256) void gru_bwd_step_bf16_kernel(const float* __restrict__ grad_out, const float* __restrict__ hstate,
                                                                const float* __restrict__ saved, const uint16_t* __restrict__ wt_bf, int ldt,
                                                                const int64_t* __restrict__ seq_len, int B, int T, int H, int t,
                                                                float* __restrict__ dxproj, float* __restrict__ dhproj,
                                                                uint16_t* __restrict__ dhproj_bf, float* __restrict__ carry,
                                                                float* __restrict__ dh0) {
    __shared__ float red[4][GT * GT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * GT, b0 = blockIdx.y * GT;
    const int G = 3 * H;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int bl = tid >> 4, jl = tid & 15;
    const int b = b0 + bl, j = j0 + jl;
    const bool mine = b < B && j < H && t >= 0;
    const size_t row = (size_t)(mine ? b : 0) * T + (t >= 0 ? t : 0);
    const int jj = mine ? j : 0;
    const float* sv = saved + row * 4 * H;
    const float s_r = sv[jj], s_z = sv[H + jj], s_n = sv[2 * H + jj], s_hn = sv[3 * H + jj];
    const float hprev = hstate[((size_t)(mine ? b : 0) * (T + 1) + (t >= 0 ? t : 0)) * H + jj];
    const float gout = grad_out[row * H + jj];
    const float cin = carry[(size_t)((b < B) ? b : 0) * H + ((j < H) ? j : 0)];
    if (t + 1 < T) {
        const int brow = b0 + li, jcol = j0 + li;
        const uint16_t* dp = dhproj_bf + ((size_t)(brow < B ? brow : 0) * T + (t + 1)) * G;
        const uint16_t* wp = wt_bf + (size_t)(jcol < H ? jcol : 0) * ldt;
        const int n_steps = G / 128;
        const int gbase = wave * (G / 4) + 8 * q;
        gbf8 fa[GBB_MAXSTEPS], fb[GBB_MAXSTEPS];
#pragma unroll
        for (int i = 0; i < GBB_MAXSTEPS; ++i)
            if (i < n_steps) {
                fa[i] = *reinterpret_cast<const gbf8*>(dp + gbase + 32 * i);
                fb[i] = *reinterpret_cast<const gbf8*>(wp + gbase + 32 * i);
            }
        // three independent accumulation chains (k-steps i = c mod 3): a dependent v_mfma_f32_16x16x32_bf16 chain of 3 H / 128
        // links would cost its full latency per link; same order in gru_persist.hip
        f32x4 acc3[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int i = 0; i < GBB_MAXSTEPS; ++i)
            if (i < n_steps) acc3[i % 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[i], acc3[i % 3], 0, 0, 0);
        acc = (acc3[0] + acc3[1]) + acc3[2];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(4 * q + r) * GT + li] = acc[r];
    __syncthreads();
    if (b < B && j < H) {
        const int e = bl * GT + jl;
        const float dstate = mg_gru_dstate(cin, red[0][e], red[1][e], red[2][e], red[3][e]);
        if (t < 0) {
            dh0[(size_t)b * H + j] = dstate;
            return;
        }
        const bool active = seq_len ? ((int64_t)t < seq_len[b]) : true;
        float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, c = dstate;
        if (active) {
            const mg_gru_cell_grad g = mg_gru_cell_bwd(dstate, gout, s_r, s_z, s_n, s_hn, hprev);
            dr = g.dr; dz = g.dz; dn = g.dn; dnr = g.dnr; c = g.carry;
        }
        float* dx = dxproj + row * G;
        float* dhp = dhproj + row * G;
        uint16_t* dhb = dhproj_bf + row * G;
        dx[j] = dr;  dx[H + j] = dz;  dx[2 * H + j] = dn;
        dhp[j] = dr; dhp[H + j] = dz; dhp[2 * H + j] = dnr;
        dhb[j] = mg_f2bf(dr); dhb[H + j] = mg_f2bf(dz); dhb[2 * H + j] = mg_f2bf(dnr);
        carry[(size_t)b * H + j] = c;
    }
}

__global__ __launch_bounds__(256) void gru_init_carry_kernel(const float* __restrict__ grad_hn, float* __restrict__ carry, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) carry[i] = grad_hn ? grad_hn[i] : 0.f;
}

extern "C" {

int mg_gru_fwd_f32(const float* xproj, const float* w_hh, const float* b_hh, const int64_t* seq_len, int B, int T, int H,
                   float* hstate, float* out, float* saved, void* stream) {
    MG_CHECK_ARG(xproj && w_hh && b_hh && hstate && out && saved && B > 0 && T > 0 && H > 0, "mg_gru_fwd_f32: bad arguments (B=%d T=%d H=%d)", B, T, H);
    if (mg_gru_small_supported(H) && ((uintptr_t)w_hh % 16) == 0)         // one launch, workgroup-local recurrence (gru_small.hip)
        return mg_gru_fwd_small_f32(xproj, w_hh, b_hh, seq_len, B, T, H, hstate, out, saved, stream);
    const int vec = (H % 4 == 0) && (((uintptr_t)hstate | (uintptr_t)w_hh) % 16 == 0);
    dim3 grid((unsigned)mg_ceil_div(H, GT), (unsigned)mg_ceil_div(B, GT));
    for (int t = 0; t < T; ++t) {
        if (H == 512 && vec)
            hipLaunchKernelGGL(gru_fwd_step_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, xproj, w_hh, b_hh, seq_len, B, T, H, t, hstate, out, saved, vec);
        else
            hipLaunchKernelGGL(gru_fwd_step_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, xproj, w_hh, b_hh, seq_len, B, T, H, t, hstate, out, saved, vec);
    }
    MG_CHECK_LAUNCH("mg_gru_fwd_f32");
    return MG_OK;
}

size_t mg_gru_bwd_workspace_bytes(int B, int H) { return mg_align_up((size_t)B * H * sizeof(float), 256); }

int mg_gru_bwd_f32(const float* grad_out, const float* grad_hn, const float* hstate, const float* saved, const float* w_hh,
                   const int64_t* seq_len, int B, int T, int H, float* dxproj, float* dhproj, float* dh0, void* workspace,
                   size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(grad_out && hstate && saved && w_hh && dxproj && dhproj && dh0 && B > 0 && T > 0 && H > 0,
                 "mg_gru_bwd_f32: bad arguments (B=%d T=%d H=%d)", B, T, H);
    if (!workspace || workspace_bytes < mg_gru_bwd_workspace_bytes(B, H)) {
        mg_set_error("mg_gru_bwd_f32: workspace of %zu bytes needed, got %zu", mg_gru_bwd_workspace_bytes(B, H), workspace_bytes);
        return MG_EWORKSPACE;
    }
    if (mg_gru_small_supported(H)) return mg_gru_bwd_small_f32(grad_out, grad_hn, hstate, saved, w_hh, seq_len, B, T, H, dxproj, dhproj, dh0, stream);
    float* carry = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int vec = ((3 * H) % 4 == 0) && (((uintptr_t)dhproj) % 16 == 0);
    int64_t n = (int64_t)B * H;
    hipLaunchKernelGGL(gru_init_carry_kernel, dim3((unsigned)mg_ceil_div(n, 256)), dim3(256), 0, st, grad_hn, carry, n);
    dim3 grid((unsigned)mg_ceil_div(H, GT), (unsigned)mg_ceil_div(B, GT));
    for (int t = T - 1; t >= -1; --t) {
        if (H == 512 && vec)
            hipLaunchKernelGGL(gru_bwd_step_kernel<24>, grid, dim3(256), 0, st, grad_out, hstate, saved, w_hh, seq_len, B, T, H, t, dxproj, dhproj, carry, dh0, vec);
        else
            hipLaunchKernelGGL(gru_bwd_step_kernel<0>, grid, dim3(256), 0, st, grad_out, hstate, saved, w_hh, seq_len, B, T, H, t, dxproj, dhproj, carry, dh0, vec);
    }
    MG_CHECK_LAUNCH("mg_gru_bwd_f32");
    return MG_OK;
}

int mg_gru_fwd_bf16(const float* xproj, const uint16_t* w_hh_bf, int ldw, const float* b_hh, const int64_t* seq_len, int B, int T, int H,
                    float* hstate, uint16_t* hstate_bf, float* out, float* saved, void* stream) {
    MG_CHECK_ARG(xproj && w_hh_bf && b_hh && hstate && hstate_bf && out && saved && B > 0 && T > 0 && H > 0,
                 "mg_gru_fwd_bf16: bad arguments (B=%d T=%d H=%d)", B, T, H);
    MG_CHECK_ARG(H % 128 == 0 && H / 128 <= GB_MAXSTEPS && ldw >= H && ldw % 8 == 0, "mg_gru_fwd_bf16: needs H %% 128 == 0, H <= %d (H=%d ldw=%d)",
                 128 * GB_MAXSTEPS, H, ldw);
    MG_CHECK_ARG((((uintptr_t)w_hh_bf | (uintptr_t)hstate_bf) % 16) == 0, "mg_gru_fwd_bf16: bf16 buffers must be 16-byte aligned");
    dim3 grid((unsigned)mg_ceil_div(H, GT), (unsigned)mg_ceil_div(B, GT));
    for (int t = 0; t < T; ++t)
        hipLaunchKernelGGL(gru_fwd_step_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, xproj, w_hh_bf, ldw, b_hh, seq_len, B, T, H, t,
                           hstate, hstate_bf, out, saved);
    MG_CHECK_LAUNCH("mg_gru_fwd_bf16");
    return MG_OK;
}

int mg_gru_bwd_bf16(const float* grad_out, const float* grad_hn, const float* hstate, const float* saved, const uint16_t* w_hh_t_bf,
                    int ldt, const int64_t* seq_len, int B, int T, int H, float* dxproj, float* dhproj, uint16_t* dhproj_bf, float* dh0,
                    void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(grad_out && hstate && saved && w_hh_t_bf && dxproj && dhproj && dhproj_bf && dh0 && B > 0 && T > 0 && H > 0,
                 "mg_gru_bwd_bf16: bad arguments (B=%d T=%d H=%d)", B, T, H);
    MG_CHECK_ARG((3 * H) % 128 == 0 && 3 * H / 128 <= GBB_MAXSTEPS && ldt >= 3 * H && ldt % 8 == 0,
                 "mg_gru_bwd_bf16: needs 3 H %% 128 == 0, H <= %d (H=%d ldt=%d)", 128 * GBB_MAXSTEPS / 3, H, ldt);
    MG_CHECK_ARG((((uintptr_t)w_hh_t_bf | (uintptr_t)dhproj_bf) % 16) == 0, "mg_gru_bwd_bf16: bf16 buffers must be 16-byte aligned");
    if (!workspace || workspace_bytes < mg_gru_bwd_workspace_bytes(B, H)) {
        mg_set_error("mg_gru_bwd_bf16: workspace of %zu bytes needed, got %zu", mg_gru_bwd_workspace_bytes(B, H), workspace_bytes);
        return MG_EWORKSPACE;
    }
    float* carry = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    int64_t n = (int64_t)B * H;
    hipLaunchKernelGGL(gru_init_carry_kernel, dim3((unsigned)mg_ceil_div(n, 256)), dim3(256), 0, st, grad_hn, carry, n);
    dim3 grid((unsigned)mg_ceil_div(H, GT), (unsigned)mg_ceil_div(B, GT));
    for (int t = T - 1; t >= -1; --t)
        hipLaunchKernelGGL(gru_bwd_step_bf16_kernel, grid, dim3(256), 0, st, grad_out, hstate, saved, w_hh_t_bf, ldt, seq_len, B, T, H, t, dxproj,
                           dhproj, dhproj_bf, carry, dh0);
    MG_CHECK_LAUNCH("mg_gru_bwd_bf16");
    return MG_OK;
}

}  // extern "C"
