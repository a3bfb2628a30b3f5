// LSTM recurrence behind RecurrentCuDNNWrapper (reference: morgana/utils.py:345-393 around torch.nn.LSTM, the cell of the
// reference's shipped acoustic model, models/RNN_SPSS.py:36-37; gate order i, f, g, o; hidden = (h, c), utils.py:374-389).
//
// Same decomposition as gru.hip: the input projection xproj = x W_ih^T + b_ih is one big GEMM over all frames (caller),
// the recurrence is one launch per time step: 16 (batch) x 16 (hidden unit) tiles, the contraction over h_{t-1} split
// across the 4 waves on v_mfma_f32_16x16x4_f32 with the fragments requested straight from L2, a 16 KB LDS exchange, and
// the cell update in the same kernel.  BPTT: dgates_t from (dh_t, dc_t), dh_{t-1} = dgates_t W_hh on the matrix pipe;
// dW_ih, dW_hh, db and dx are big GEMMs over all frames after the loop (both biases see the same dgates).
// Items past their length keep (h, c) frozen and emit zeros, as the packed sequence of the reference does.
#include "common.h"
#include "lstm_cell.h"

#define LT 16

__device__ __forceinline__ f32x4 lstm_ld4(const float* p, int valid, bool vec) {
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

// hstate / cstate [B, T+1, H]: slot t holds the state before step t (slot 0 = initial), slot t+1 receives the new state.
// xproj holds the times x_t0 .. x_t0 + x_T - 1 only ([B, x_T, 4H]): the whole sequence for a single layer (x_T = T, x_t0 = 0),
// one chunk of it for the upper layers of a skewed stack (mg_lstm_stack_fwd_f32).
// NB > 0: the wave's contraction blocks are requested NB at a time, PARTS times (H == 64 NB PARTS); NB == 0: streamed.
template <int NB, int PARTS = 1>
__device__ __forceinline__ void lstm_fwd_step_body(const float* __restrict__ xproj, int x_T, int x_t0, const float* __restrict__ w_hh,
                                                   const float* __restrict__ b_hh, const int64_t* __restrict__ seq_len,
                                                   int B, int T, int H, int t, float* __restrict__ hstate,
                                                   float* __restrict__ cstate, float* __restrict__ out,
                                                   float* __restrict__ saved, int vec) {
    __shared__ float red[4][4][LT * LT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * LT, b0 = blockIdx.y * LT;
    const int brow = b0 + li, jrow = j0 + li;
    const float* hp = hstate + ((size_t)(brow < B ? brow : 0) * (T + 1) + t) * H;
    const float* wg[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) wg[g] = w_hh + ((size_t)g * H + (jrow < H ? jrow : 0)) * H;

    const int bl = tid >> 4, jl = tid & 15;
    const int b = b0 + bl, j = j0 + jl;
    const bool mine = b < B && j < H;
    const int bb = mine ? b : 0, jj = mine ? j : 0;
    const size_t row = (size_t)bb * T + t;
    const float* xp = xproj + ((size_t)bb * x_T + (t - x_t0)) * 4 * H;
    float xg[4], bh[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        xg[g] = xp[g * H + jj];
        bh[g] = b_hh[g * H + jj];
    }
    const float hprev = hstate[((size_t)bb * (T + 1) + t) * H + jj];
    const float cprev = cstate[((size_t)bb * (T + 1) + t) * H + jj];
    const bool active = seq_len ? ((int64_t)t < seq_len[bb]) : true;

    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load_blk = [&](int k0, f32x4& a, f32x4 (&w)[4]) {
        const int k = k0 + 4 * q;
        const int valid = H - k;
        a = (brow < B) ? lstm_ld4(hp + k, valid, vec) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int g = 0; g < 4; ++g) w[g] = (jrow < H) ? lstm_ld4(wg[g] + k, valid, vec) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    if (NB > 0) {
#pragma unroll
        for (int part = 0; part < PARTS; ++part) {
            f32x4 fa[NB > 0 ? NB : 1], fw[NB > 0 ? NB : 1][4];
#pragma unroll
            for (int i = 0; i < NB; ++i) load_blk(wave * 16 + 64 * (part * NB + i), fa[i], fw[i]);
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fw[i][g][e], acc[g], 0, 0, 0);
        }
    } else {
        f32x4 a0, a1, w0[4], w1[4];
        int k0 = wave * 16;
        if (k0 < H) load_blk(k0, a0, w0);
        for (; k0 < H; k0 += 64) {
            const bool more = k0 + 64 < H;
            if (more) load_blk(k0 + 64, a1, w1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], w0[g][e], acc[g], 0, 0, 0);
            if (more) {
                a0 = a1;
#pragma unroll
                for (int g = 0; g < 4; ++g) w0[g] = w1[g];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = (4 * q + r) * LT + li;
#pragma unroll
        for (int g = 0; g < 4; ++g) red[wave][g][e] = acc[g][r];
    }
    __syncthreads();
    if (mine) {
        const int e = bl * LT + jl;
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[g] = mg_lstm_pre(xg[g], red[0][g][e], red[1][g][e], red[2][g][e], red[3][g][e], bh[g]);
        const mg_lstm_cell_out cell = mg_lstm_cell_exact(pre[0], pre[1], pre[2], pre[3], cprev);     // lstm_cell.h: contraction pinned
        const float ig = cell.i, fg = cell.f, gg = cell.g, og = cell.o, cnew = cell.c, hnew = cell.h;
        const size_t nxt = ((size_t)b * (T + 1) + t + 1) * H + j;
        hstate[nxt] = active ? hnew : hprev;
        cstate[nxt] = active ? cnew : cprev;
        out[row * H + j] = active ? hnew : 0.f;
        float* sv = saved + row * 4 * H;
        sv[j] = ig;
        sv[H + j] = fg;
        sv[2 * H + j] = gg;
        sv[3 * H + j] = og;
    }
}

template <int NB>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                            const float* __restrict__ b_hh, const int64_t* __restrict__ seq_len,
                                                            int B, int T, int H, int t, float* __restrict__ hstate,
                                                            float* __restrict__ cstate, float* __restrict__ out,
                                                            float* __restrict__ saved, int vec) {
    lstm_fwd_step_body<NB>(xproj, T, 0, w_hh, b_hh, seq_len, B, T, H, t, hstate, cstate, out, saved, vec);
}

// Skewed stack (mg_lstm_stack_fwd_f32): blockIdx.z is the layer, every layer at its own time step (-1: nothing to do).
struct LstmFwdMulti {
    mg_lstm_fwd_layer l[MG_LSTM_MAX_LAYERS];
    int t[MG_LSTM_MAX_LAYERS];
};

template <int NB, int PARTS>
__global__ __launch_bounds__(256) void lstm_fwd_multi_kernel(LstmFwdMulti a, const int64_t* __restrict__ seq_len, int B, int T, int H,
                                                             int vec) {
    const int layer = blockIdx.z;
    const int t = a.t[layer];
    if (t < 0) return;
    const mg_lstm_fwd_layer& p = a.l[layer];
    lstm_fwd_step_body<NB, PARTS>(p.xproj, p.x_T, p.x_t0, p.w_hh, p.b_hh, seq_len, B, T, H, t, p.hstate, p.cstate, p.out, p.saved, vec);
}


// Backward step t (t = -1: only the matmul; writes dh0 / dc0).  carry_h / carry_c [B,H] hold the elementwise part of the
// gradient of the state before step t+1; the matmul part dgates[:, t+1, :] W_hh is added here.
// grad_out holds the times g_t0 .. g_t0 + g_T - 1 only ([B, g_T, H]); NULL means a zero gradient.
template <int NB, int PARTS = 2>
__device__ __forceinline__ void lstm_bwd_step_body(const float* __restrict__ grad_out, int g_T, int g_t0, const float* __restrict__ cstate,
                                                   const float* __restrict__ saved, const float* __restrict__ w_hh,
                                                   const int64_t* __restrict__ seq_len, int B, int T, int H, int t,
                                                   float* __restrict__ dgates, float* __restrict__ carry_h,
                                                   float* __restrict__ carry_c, float* __restrict__ dh0,
                                                   float* __restrict__ dc0, int vec) {
    __shared__ float red[4][LT * LT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int j0 = blockIdx.x * LT, b0 = blockIdx.y * LT;
    const int G = 4 * H;
    const int bl = tid >> 4, jl = tid & 15;
    const int b = b0 + bl, j = j0 + jl;
    const bool in_range = b < B && j < H;
    const bool mine = in_range && t >= 0;
    const int bb = in_range ? b : 0, jj = in_range ? j : 0, tt = t >= 0 ? t : 0;
    const size_t row = (size_t)bb * T + tt;
    const float* sv = saved + row * 4 * H;
    const float s_i = sv[jj], s_f = sv[H + jj], s_g = sv[2 * H + jj], s_o = sv[3 * H + jj];
    const float c_prev = cstate[((size_t)bb * (T + 1) + tt) * H + jj];
    const float c_new = cstate[((size_t)bb * (T + 1) + tt + 1) * H + jj];
    const float gout = (grad_out && t >= 0) ? grad_out[((size_t)bb * g_T + (t - g_t0)) * H + jj] : 0.f;
    const float ch_in = carry_h[(size_t)bb * H + jj];
    const float cc_in = carry_c[(size_t)bb * H + jj];

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (t + 1 < T) {
        const int brow = b0 + li, jcol = j0 + li;
        const float* dp = dgates + ((size_t)(brow < B ? brow : 0) * T + (t + 1)) * G;
        auto load_blk = [&](int g0, f32x4& a, float (&bv)[4]) {
            const int g = g0 + 4 * q;
            a = (brow < B) ? lstm_ld4(dp + g, G - g, vec) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[e] = (jcol < H && g + e < G) ? w_hh[(size_t)(g + e) * H + jcol] : 0.f;
        };
        if (NB > 0) {
            // PARTS rounds of NB blocks each (4 H == 64 NB PARTS): 8 NB fragment registers in flight at a time
#pragma unroll
            for (int half = 0; half < PARTS; ++half) {
                f32x4 fa[NB > 0 ? NB : 1];
                float fb[NB > 0 ? NB : 1][4];
#pragma unroll
                for (int i = 0; i < NB; ++i) load_blk(wave * 16 + 64 * (half * NB + i), fa[i], fb[i]);
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[i][e], acc, 0, 0, 0);
            }
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
    for (int r = 0; r < 4; ++r) red[wave][(4 * q + r) * LT + li] = acc[r];
    __syncthreads();
    if (!in_range) return;
    const int e = bl * LT + jl;
    const float dh_state = mg_lstm_dstate(ch_in, red[0][e], red[1][e], red[2][e], red[3][e]);
    const float dc_state = cc_in;
    if (t < 0) {
        dh0[(size_t)b * H + j] = dh_state;
        dc0[(size_t)b * H + j] = dc_state;
        return;
    }
    const bool active = seq_len ? ((int64_t)t < seq_len[b]) : true;
    float di = 0.f, df = 0.f, dg = 0.f, d_o = 0.f, ch = dh_state, cc = dc_state;
    if (mine && active) {
        const mg_lstm_cell_grad cg = mg_lstm_cell_bwd(dh_state, dc_state, gout, s_i, s_f, s_g, s_o, c_prev, c_new);   // lstm_cell.h
        di = cg.di;
        df = cg.df;
        dg = cg.dg;
        d_o = cg.d_o;
        ch = 0.f;                 // all of dh_{t-1} comes through the matmul with the gates of this step
        cc = cg.cc;
    }
    float* dgp = dgates + row * G;
    dgp[j] = di;
    dgp[H + j] = df;
    dgp[2 * H + j] = dg;
    dgp[3 * H + j] = d_o;
    carry_h[(size_t)b * H + j] = ch;
    carry_c[(size_t)b * H + j] = cc;
}

template <int NB>
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const float* __restrict__ grad_out, const float* __restrict__ cstate,
                                                            const float* __restrict__ saved, const float* __restrict__ w_hh,
                                                            const int64_t* __restrict__ seq_len, int B, int T, int H, int t,
                                                            float* __restrict__ dgates, float* __restrict__ carry_h,
                                                            float* __restrict__ carry_c, float* __restrict__ dh0,
                                                            float* __restrict__ dc0, int vec) {
    lstm_bwd_step_body<NB>(grad_out, T, 0, cstate, saved, w_hh, seq_len, B, T, H, t, dgates, carry_h, carry_c, dh0, dc0, vec);
}

struct LstmBwdMulti {
    mg_lstm_bwd_layer l[MG_LSTM_MAX_LAYERS];
    int t[MG_LSTM_MAX_LAYERS];                  // -2: nothing to do (t = -1 is the final matmul-only step)
};

template <int NB, int PARTS>
__global__ __launch_bounds__(256) void lstm_bwd_multi_kernel(LstmBwdMulti a, const int64_t* __restrict__ seq_len, int B, int T, int H,
                                                             int vec) {
    const int layer = blockIdx.z;
    const int t = a.t[layer];
    if (t < -1) return;
    const mg_lstm_bwd_layer& p = a.l[layer];
    lstm_bwd_step_body<NB, PARTS>(p.grad_out, p.g_T, p.g_t0, p.cstate, p.saved, p.w_hh, seq_len, B, T, H, t, p.dgates, p.carry_h, p.carry_c,
                           p.dh0, p.dc0, vec);
}

__global__ __launch_bounds__(256) void lstm_init_carry_kernel(const float* __restrict__ grad_hn, const float* __restrict__ grad_cn,
                                                              float* __restrict__ carry_h, float* __restrict__ carry_c, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        carry_h[i] = grad_hn ? grad_hn[i] : 0.f;
        carry_c[i] = grad_cn ? grad_cn[i] : 0.f;
    }
}

extern "C" {

int mg_lstm_fwd_f32(const float* xproj, const float* w_hh, const float* b_hh, const int64_t* seq_len, int B, int T, int H,
                    float* hstate, float* cstate, float* out, float* saved, void* stream) {
    MG_CHECK_ARG(xproj && w_hh && b_hh && hstate && cstate && out && saved && B > 0 && T > 0 && H > 0,
                 "mg_lstm_fwd_f32: bad arguments (B=%d T=%d H=%d)", B, T, H);
    const int vec = (H % 4 == 0) && (((uintptr_t)hstate | (uintptr_t)w_hh) % 16 == 0);
    dim3 grid((unsigned)mg_ceil_div(H, LT), (unsigned)mg_ceil_div(B, LT));
    for (int t = 0; t < T; ++t) {
        if (H == 512 && vec)
            hipLaunchKernelGGL(lstm_fwd_step_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, xproj, w_hh, b_hh, seq_len, B, T, H, t, hstate, cstate, out, saved, vec);
        else
            hipLaunchKernelGGL(lstm_fwd_step_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, xproj, w_hh, b_hh, seq_len, B, T, H, t, hstate, cstate, out, saved, vec);
    }
    MG_CHECK_LAUNCH("mg_lstm_fwd_f32");
    return MG_OK;
}

size_t mg_lstm_bwd_workspace_bytes(int B, int H) { return mg_align_up((size_t)2 * B * H * sizeof(float), 256); }

int mg_lstm_bwd_f32(const float* grad_out, const float* grad_hn, const float* grad_cn, const float* cstate, const float* saved,
                    const float* w_hh, const int64_t* seq_len, int B, int T, int H, float* dgates, float* dh0, float* dc0,
                    void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(grad_out && cstate && saved && w_hh && dgates && dh0 && dc0 && B > 0 && T > 0 && H > 0,
                 "mg_lstm_bwd_f32: bad arguments (B=%d T=%d H=%d)", B, T, H);
    if (!workspace || workspace_bytes < mg_lstm_bwd_workspace_bytes(B, H)) {
        mg_set_error("mg_lstm_bwd_f32: workspace of %zu bytes needed, got %zu", mg_lstm_bwd_workspace_bytes(B, H), workspace_bytes);
        return MG_EWORKSPACE;
    }
    float* carry_h = (float*)workspace;
    float* carry_c = carry_h + (size_t)B * H;
    hipStream_t st = (hipStream_t)stream;
    const int vec = ((4 * H) % 4 == 0) && (((uintptr_t)dgates) % 16 == 0);
    const int64_t n = (int64_t)B * H;
    hipLaunchKernelGGL(lstm_init_carry_kernel, dim3((unsigned)mg_ceil_div(n, 256)), dim3(256), 0, st, grad_hn, grad_cn, carry_h, carry_c, n);
    dim3 grid((unsigned)mg_ceil_div(H, LT), (unsigned)mg_ceil_div(B, LT));
    for (int t = T - 1; t >= -1; --t) {
        if (H == 512 && vec)
            hipLaunchKernelGGL(lstm_bwd_step_kernel<16>, grid, dim3(256), 0, st, grad_out, cstate, saved, w_hh, seq_len, B, T, H, t, dgates, carry_h, carry_c, dh0, dc0, vec);
        else
            hipLaunchKernelGGL(lstm_bwd_step_kernel<0>, grid, dim3(256), 0, st, grad_out, cstate, saved, w_hh, seq_len, B, T, H, t, dgates, carry_h, carry_c, dh0, dc0, vec);
    }
    MG_CHECK_LAUNCH("mg_lstm_bwd_f32");
    return MG_OK;
}

// Skewed stack of LSTM layers (the 8 x LSTM-512 of models/RNN_SPSS.py:36-37).  Run layer after layer, the T dependent steps of
// every layer queue up behind each other: L T launches of ~10 us with 128 of the chip's 256 CUs half busy.  Here layer l runs
// `lag` steps behind layer l - 1 and ONE launch per step serves all layers (blockIdx.z = layer): T + (L - 1) lag launches.
// With all layers in one launch a step takes 25 us forward / 34 us backward for 8 x LSTM-512 at batch 64 (one layer alone: 10 /
// 11 us): 128 MB of W_hh and h fragments cross L2 -> L1 per step, i.e. the launch is bound by ~5 TB/s of L2 delivery.  A form
// with a whole contraction per wave (64 batch x 16 hidden per workgroup, cell update on the accumulators, no LDS) was tried
// and is SLOWER (32 us): its four waves each pull their own copy of the 128 KB W_hh slice through a 32 KB L1; with the slice
// staged once per workgroup through LDS (64- or 128-column chunks, double buffered) it takes the same 25 us as the tiles - also
// with a single layer active, i.e. the time is one workgroup's dependent chain: 512 exact-fp32 MFMAs per wave (8 us), the
// chunk fetches and the cell update's scattered operand loads behind them.
// Every `lag` steps the caller computes the next chunk of input projections of the upper layers (a GEMM over the lag
// outputs the layer below has just finished) - forward - and of output gradients of the lower layers - backward.
// Step s of the forward pass: layer l is at time t = s - l lag.  Step u of the backward pass: layer l is at time
// t = T_pad - 1 - (u - (L - 1 - l) lag), T_pad = T rounded up to a multiple of lag; t == -1 finishes dh0 / dc0.
int mg_lstm_stack_fwd_f32(const mg_lstm_fwd_layer* layers, int n_layers, const int64_t* seq_len, int B, int T, int H, int lag,
                          int s_begin, int s_end, void* stream) {
    MG_CHECK_ARG(layers && n_layers >= 1 && n_layers <= MG_LSTM_MAX_LAYERS && B > 0 && T > 0 && H > 0 && lag > 0 && s_begin >= 0,
                 "mg_lstm_stack_fwd_f32: bad arguments (layers=%d B=%d T=%d H=%d lag=%d)", n_layers, B, T, H, lag);
    LstmFwdMulti a;
    int vec = (H % 4 == 0);
    for (int l = 0; l < n_layers; ++l) {
        const mg_lstm_fwd_layer& p = layers[l];
        MG_CHECK_ARG(p.xproj && p.w_hh && p.b_hh && p.hstate && p.cstate && p.out && p.saved && p.x_T > 0,
                     "mg_lstm_stack_fwd_f32: layer %d has a null buffer", l);
        a.l[l] = p;
        vec = vec && (((uintptr_t)p.hstate | (uintptr_t)p.w_hh) % 16 == 0);
    }
    dim3 grid((unsigned)mg_ceil_div(H, LT), (unsigned)mg_ceil_div(B, LT), (unsigned)n_layers);
    for (int s = s_begin; s < s_end; ++s) {
        bool any = false;
        for (int l = 0; l < n_layers; ++l) {
            const int t = s - l * lag;
            a.t[l] = (t >= 0 && t < T) ? t : -1;
            any = any || a.t[l] >= 0;
            if (a.t[l] >= 0 && (t < a.l[l].x_t0 || t >= a.l[l].x_t0 + a.l[l].x_T)) {
                mg_set_error("mg_lstm_stack_fwd_f32: layer %d has no input projection for time %d (has %d..%d)", l, t, a.l[l].x_t0,
                             a.l[l].x_t0 + a.l[l].x_T - 1);
                return MG_EINVAL;
            }
        }
        if (!any) continue;
        if (H == 512 && vec)
            hipLaunchKernelGGL((lstm_fwd_multi_kernel<4, 2>), grid, dim3(256), 0, (hipStream_t)stream, a, seq_len, B, T, H, vec);   // 4 blocks in flight x 2: under 128 VGPRs, so that the workgroups of all layers are resident together
        else
            hipLaunchKernelGGL((lstm_fwd_multi_kernel<0, 1>), grid, dim3(256), 0, (hipStream_t)stream, a, seq_len, B, T, H, vec);
    }
    MG_CHECK_LAUNCH("mg_lstm_stack_fwd_f32");
    return MG_OK;
}

int mg_lstm_stack_bwd_f32(const mg_lstm_bwd_layer* layers, int n_layers, const int64_t* seq_len, int B, int T, int H, int lag,
                          int u_begin, int u_end, void* stream) {
    MG_CHECK_ARG(layers && n_layers >= 1 && n_layers <= MG_LSTM_MAX_LAYERS && B > 0 && T > 0 && H > 0 && lag > 0 && u_begin >= 0,
                 "mg_lstm_stack_bwd_f32: bad arguments (layers=%d B=%d T=%d H=%d lag=%d)", n_layers, B, T, H, lag);
    LstmBwdMulti a;
    int vec = 1;
    for (int l = 0; l < n_layers; ++l) {
        const mg_lstm_bwd_layer& p = layers[l];
        MG_CHECK_ARG(p.cstate && p.saved && p.w_hh && p.dgates && p.carry_h && p.carry_c && p.dh0 && p.dc0,
                     "mg_lstm_stack_bwd_f32: layer %d has a null buffer", l);
        a.l[l] = p;
        vec = vec && (((uintptr_t)p.dgates) % 16 == 0);
    }
    const int t_pad = (int)(mg_ceil_div(T, lag) * lag);
    dim3 grid((unsigned)mg_ceil_div(H, LT), (unsigned)mg_ceil_div(B, LT), (unsigned)n_layers);
    for (int u = u_begin; u < u_end; ++u) {
        bool any = false;
        for (int l = 0; l < n_layers; ++l) {
            const int ul = u - (n_layers - 1 - l) * lag;
            const int t = t_pad - 1 - ul;
            a.t[l] = (ul >= 0 && t >= -1 && t < T) ? t : -2;
            any = any || a.t[l] >= -1;
            if (a.t[l] >= 0 && a.l[l].grad_out && (t < a.l[l].g_t0 || t >= a.l[l].g_t0 + a.l[l].g_T)) {
                mg_set_error("mg_lstm_stack_bwd_f32: layer %d has no output gradient for time %d (has %d..%d)", l, t, a.l[l].g_t0,
                             a.l[l].g_t0 + a.l[l].g_T - 1);
                return MG_EINVAL;
            }
        }
        if (!any) continue;
        if (H == 512 && vec)
            hipLaunchKernelGGL((lstm_bwd_multi_kernel<8, 4>), grid, dim3(256), 0, (hipStream_t)stream, a, seq_len, B, T, H, vec);
        else
            hipLaunchKernelGGL((lstm_bwd_multi_kernel<0, 2>), grid, dim3(256), 0, (hipStream_t)stream, a, seq_len, B, T, H, vec);
    }
    MG_CHECK_LAUNCH("mg_lstm_stack_bwd_f32");
    return MG_OK;
}

}  // extern "C"
