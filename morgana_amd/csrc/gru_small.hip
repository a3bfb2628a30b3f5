// K3s - GRU recurrence for SMALL hidden sizes (H = 64 or 128: the three GRU-64 layers of the shipped F0 model,
// models/f0_test_model.py:32-39), exact fp32, ONE launch per direction and NO hand-off between workgroups.
// At these sizes the whole of W_hh fits in the registers of one workgroup (H = 64: 48 KB fp32 = 48 VGPRs per lane), so a
// workgroup owns R = 256 / H items outright: their state lives in LDS, a step is 3H/16 N-tiles of v_mfma_f32_16x16x4_f32 over
// the full contraction (no K split, no partial sums), an LDS round trip and the cell - about 1 us against 8.8 us for the
// launch-per-step kernel, and nothing to synchronise but the workgroup's own barriers.  ceil(B / R) workgroups.
// Same arithmetic as gru_fwd_step_kernel / gru_bwd_step_kernel (gru.hip: expf / tanhf cell, fp32 products), summed over the
// contraction in one chain per output instead of four partial chains: equal to fp32 rounding of the sum order (tests: 1e-5).
// mg_gru_fwd_f32 / mg_gru_bwd_f32 route here for H in {64, 128} (MG_TUNE key 3 = 1: the per-step kernels; = 2: H = 64 on the
// 16-row tiles instead of the 4 x 4 blocks below).
// Fragment trick as in gru.hip: lane l = 16 q + i holds for a 16-deep contraction block the 4 consecutive values k = 4q .. 4q+3
// of row i (one 16-byte read); MFMA number e consumes element e - A and B use the same (q, e) <-> k map.
#include "common.h"

#define GS_ROWS 16                   // MFMA tile rows; a workgroup owns R <= 16 items, the other rows are zero

template <int H>
__global__ __launch_bounds__(256) void gru_fwd_small_kernel(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                            const float* __restrict__ b_hh, const int64_t* __restrict__ seq_len, int B,
                                                            int T, int R, float* __restrict__ hstate, float* __restrict__ out,
                                                            float* __restrict__ saved) {
    constexpr int G = 3 * H, KB = H / 16, NT = G / 16 / 4;          // k-blocks of 16, N-tiles per wave
    constexpr int LDH = H + 4, LDG = G + 4;                          // LDS row strides (floats): +4 keeps 16-byte alignment, shifts banks
    __shared__ __attribute__((aligned(16))) float hs[GS_ROWS][LDH];  // h_{t-1} of the workgroup's items (rows >= nrows stay zero)
    __shared__ __attribute__((aligned(16))) float gl[GS_ROWS][LDG];  // W_hh h + nothing else: the three gates' recurrent pre-activations
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * R;
    const int nrows = min(R, B - row0);
    // W_hh fragments of this wave's N-tiles (tile = wave + 4 n), resident for the whole launch
    f32x4 fw[NT][KB];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const float* wp = w_hh + (size_t)((wave + 4 * n) * 16 + li) * H + 4 * q;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) fw[n][kb] = *reinterpret_cast<const f32x4*>(wp + 16 * kb);
    }
    for (int e = tid; e < GS_ROWS * LDH; e += 256) (&hs[0][0])[e] = 0.f;
    __syncthreads();
    // cell role: thread e owns element (item e / H, unit e % H); R H <= 256
    const int er = tid / H, ej = tid - er * H;
    const bool mine = er < nrows;
    const int b = row0 + (mine ? er : 0);
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
    const float bhr = b_hh[ej], bhz = b_hh[H + ej], bhn = b_hh[2 * H + ej];
    float hprev = mine ? hstate[((size_t)b * (T + 1)) * H + ej] : 0.f;
    if (mine) hs[er][ej] = hprev;
    const float* xp = xproj + (size_t)b * T * G + ej;
    float xr = xp[0], xz = xp[H], xn = xp[2 * H];
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int t1 = t + 1 < T ? t + 1 : t;
        const float xr1 = xp[(size_t)t1 * G], xz1 = xp[(size_t)t1 * G + H], xn1 = xp[(size_t)t1 * G + 2 * H];   // next step's, in flight
        f32x4 fa[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) fa[kb] = *reinterpret_cast<const f32x4*>(&hs[li][16 * kb + 4 * q]);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[kb][e], fw[n][kb][e], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) gl[4 * q + r][(wave + 4 * n) * 16 + li] = acc[r];
        }
        __syncthreads();
        if (mine) {
            const float hr = gl[er][ej] + bhr, hz = gl[er][H + ej] + bhz, hn = gl[er][2 * H + ej] + bhn;
            const float r = mg_sigmoid(xr + hr);
            const float z = mg_sigmoid(xz + hz);
            const float n = tanhf(xn + r * hn);
            const float hnew = (1.f - z) * n + z * hprev;
            const bool active = t < len;
            hprev = active ? hnew : hprev;
            hs[er][ej] = hprev;
            const size_t row = (size_t)b * T + t;
            hstate[((size_t)b * (T + 1) + t + 1) * H + ej] = hprev;
            out[row * H + ej] = active ? hnew : 0.f;
            float* sv = saved + row * 4 * H;
            sv[ej] = r;
            sv[H + ej] = z;
            sv[2 * H + ej] = n;
            sv[3 * H + ej] = hn;
        }
        xr = xr1;
        xz = xz1;
        xn = xn1;
        __syncthreads();
    }
}

// Backward.  dl holds dhproj_{t+1} = (dr, dz, dn r) of the workgroup's items; dstate_t = carry + dl W_hh: N = H (one or two
// 16-wide tiles per wave), contraction over the 3H gate rows in 4 independent accumulation chains.
template <int H>
__global__ __launch_bounds__(256) void gru_bwd_small_kernel(const float* __restrict__ grad_out, const float* __restrict__ grad_hn,
                                                            const float* __restrict__ hstate, const float* __restrict__ saved,
                                                            const float* __restrict__ w_hh, const int64_t* __restrict__ seq_len, int B,
                                                            int T, int R, float* __restrict__ dxproj, float* __restrict__ dhproj,
                                                            float* __restrict__ dh0) {
    constexpr int G = 3 * H, KB = G / 16, NT = H / 16 / 4;
    constexpr int LDG = G + 4, LDH = H + 4;
    __shared__ __attribute__((aligned(16))) float dl[GS_ROWS][LDG];
    __shared__ __attribute__((aligned(16))) float ds[GS_ROWS][LDH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * R;
    const int nrows = min(R, B - row0);
    // W_hh as the B operand of dl W_hh: lane (li = output unit of the tile, q) holds W_hh[16 kb + 4 q + e][unit]
    float fw[NT][KB][4];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int unit = (wave + 4 * n) * 16 + li;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) fw[n][kb][e] = w_hh[(size_t)(16 * kb + 4 * q + e) * H + unit];
    }
    for (int e = tid; e < GS_ROWS * LDG; e += 256) (&dl[0][0])[e] = 0.f;
    const int er = tid / H, ej = tid - er * H;
    const bool mine = er < nrows;
    const int b = row0 + (mine ? er : 0);
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
    float carry = (mine && grad_hn) ? grad_hn[(size_t)b * H + ej] : 0.f;
    const float* p_sv = saved + (size_t)b * T * 4 * H + ej;
    const float* p_h = hstate + (size_t)b * (T + 1) * H + ej;
    const float* p_g = grad_out + (size_t)b * T * H + ej;
    float s_r = p_sv[(size_t)(T - 1) * 4 * H], s_z = p_sv[(size_t)(T - 1) * 4 * H + H], s_n = p_sv[(size_t)(T - 1) * 4 * H + 2 * H],
          s_hn = p_sv[(size_t)(T - 1) * 4 * H + 3 * H], hprev = p_h[(size_t)(T - 1) * H], gout = p_g[(size_t)(T - 1) * H];
    __syncthreads();

    for (int t = T - 1; t >= -1; --t) {
        const int t1 = t > 0 ? t - 1 : 0;
        const float s_r1 = p_sv[(size_t)t1 * 4 * H], s_z1 = p_sv[(size_t)t1 * 4 * H + H], s_n1 = p_sv[(size_t)t1 * 4 * H + 2 * H],
                    s_hn1 = p_sv[(size_t)t1 * 4 * H + 3 * H], hprev1 = p_h[(size_t)t1 * H], gout1 = p_g[(size_t)t1 * H];
        if (t + 1 < T) {
            f32x4 fa[KB];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) fa[kb] = *reinterpret_cast<const f32x4*>(&dl[li][16 * kb + 4 * q]);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 acc4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc4[kb % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[kb][e], fw[n][kb][e], acc4[kb % 4], 0, 0, 0);
                const f32x4 acc = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[4 * q + r][(wave + 4 * n) * 16 + li] = acc[r];
            }
        }
        __syncthreads();
        if (mine) {
            const float dstate = carry + (t + 1 < T ? ds[er][ej] : 0.f);
            if (t < 0) {
                dh0[(size_t)b * H + ej] = dstate;
            } else {
                float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, c = dstate;
                if (t < len) {
                    const float dh = dstate + gout;
                    dn = dh * (1.f - s_z) * (1.f - s_n * s_n);
                    dz = dh * (hprev - s_n) * s_z * (1.f - s_z);
                    dr = dn * s_hn * s_r * (1.f - s_r);
                    dnr = dn * s_r;
                    c = dh * s_z;
                }
                carry = c;
                dl[er][ej] = dr;
                dl[er][H + ej] = dz;
                dl[er][2 * H + ej] = dnr;
                const size_t row = (size_t)b * T + t;
                float* dx = dxproj + row * G;
                float* dhp = dhproj + row * G;
                dx[ej] = dr;  dx[H + ej] = dz;  dx[2 * H + ej] = dn;
                dhp[ej] = dr; dhp[H + ej] = dz; dhp[2 * H + ej] = dnr;
            }
        }
        s_r = s_r1; s_z = s_z1; s_n = s_n1; s_hn = s_hn1; hprev = hprev1; gout = gout1;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// H = 64 (four items per workgroup) on v_mfma_f32_4x4x1_16B_f32: 16 independent 4 x 4 outer-product blocks per instruction, so
// the 4 MFMA rows are exactly the workgroup's 4 items and nothing is padding (the 16-row tiles above spend 3/4 of their
// products on zero rows).  Same exact-fp32 rate per product, a quarter of the instructions: a step's matrix work drops from
// 1 536 to 512 cycles.  Layout (scripts/probe/mfma4x4_probe.hip, measured on gfx950): block = lane / 4; A lane 4b + i = row i of
// block b; B lane 4b + j = column j; acc[i] in lane 4b + j = D_b[i][j].  With A = h[item lane & 3][k] in every block and
// B = W[n0 + lane][k], lane L of the wave ends up with gate column n0 + L of all 4 items in acc[0..3].
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_fwd_small64_kernel(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                              const float* __restrict__ b_hh, const int64_t* __restrict__ seq_len, int B,
                                                              int T, float* __restrict__ hstate, float* __restrict__ out,
                                                              float* __restrict__ saved) {
    constexpr int H = 64, G = 192, R = 4, LDH = H + 4, LDG = G + 4;
    __shared__ __attribute__((aligned(16))) float hs[R][LDH];       // h_{t-1}; rows of missing items stay zero
    __shared__ __attribute__((aligned(16))) float gl[R][LDG];       // recurrent pre-activations of the three gates
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * R;
    const int nrows = min(R, B - row0);
    // wave g < 3 owns gate g: lane L holds row g H + L of W_hh (64 floats) for the whole launch
    f32x4 fw[H / 4];
    if (wave < 3) {
        const float* wp = w_hh + (size_t)(wave * H + lane) * H;
#pragma unroll
        for (int k4 = 0; k4 < H / 4; ++k4) fw[k4] = *reinterpret_cast<const f32x4*>(wp + 4 * k4);
    }
    for (int e = tid; e < R * LDH; e += 256) (&hs[0][0])[e] = 0.f;
    __syncthreads();
    const int er = tid >> 6, ej = tid & 63;                          // cell role: item er, unit ej
    const bool mine = er < nrows;
    const int b = row0 + (mine ? er : 0);
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
    const float bhr = b_hh[ej], bhz = b_hh[H + ej], bhn = b_hh[2 * H + ej];
    float hprev = mine ? hstate[((size_t)b * (T + 1)) * H + ej] : 0.f;
    if (mine) hs[er][ej] = hprev;
    const float* xp = xproj + (size_t)b * T * G + ej;
    float xr = xp[0], xz = xp[H], xn = xp[2 * H];
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        const int t1 = t + 1 < T ? t + 1 : t;
        const float xr1 = xp[(size_t)t1 * G], xz1 = xp[(size_t)t1 * G + H], xn1 = xp[(size_t)t1 * G + 2 * H];
        if (wave < 3) {
            f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int k4 = 0; k4 < H / 4; ++k4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&hs[lane & 3][4 * k4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], fw[k4][e], acc[e], 0, 0, 0);
            }
            const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
            for (int i = 0; i < R; ++i) gl[i][wave * H + lane] = sum[i];
        }
        __syncthreads();
        if (mine) {
            const float hr = gl[er][ej] + bhr, hz = gl[er][H + ej] + bhz, hn = gl[er][2 * H + ej] + bhn;
            const float r = mg_sigmoid(xr + hr);
            const float z = mg_sigmoid(xz + hz);
            const float n = tanhf(xn + r * hn);
            const float hnew = (1.f - z) * n + z * hprev;
            const bool active = t < len;
            hprev = active ? hnew : hprev;
            hs[er][ej] = hprev;
            const size_t row = (size_t)b * T + t;
            hstate[((size_t)b * (T + 1) + t + 1) * H + ej] = hprev;
            out[row * H + ej] = active ? hnew : 0.f;
            float* sv = saved + row * 4 * H;
            sv[ej] = r;
            sv[H + ej] = z;
            sv[2 * H + ej] = n;
            sv[3 * H + ej] = hn;
        }
        xr = xr1;
        xz = xz1;
        xn = xn1;
        __syncthreads();
    }
}

// Backward, H = 64: dstate = carry + dl W_hh with the 192 gate rows split over the 4 waves (48 each); lane L holds column L of
// its 48 rows of W_hh; the four partial results meet in LDS.
__global__ __launch_bounds__(256) void gru_bwd_small64_kernel(const float* __restrict__ grad_out, const float* __restrict__ grad_hn,
                                                              const float* __restrict__ hstate, const float* __restrict__ saved,
                                                              const float* __restrict__ w_hh, const int64_t* __restrict__ seq_len, int B,
                                                              int T, float* __restrict__ dxproj, float* __restrict__ dhproj,
                                                              float* __restrict__ dh0) {
    constexpr int H = 64, G = 192, R = 4, LDG = G + 4, KW = G / 4;  // KW = 48 gate rows per wave
    __shared__ __attribute__((aligned(16))) float dl[R][LDG];
    __shared__ float part[4][R][H];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * R;
    const int nrows = min(R, B - row0);
    float fw[KW];
#pragma unroll
    for (int k = 0; k < KW; ++k) fw[k] = w_hh[(size_t)(wave * KW + k) * H + lane];
    for (int e = tid; e < R * LDG; e += 256) (&dl[0][0])[e] = 0.f;
    const int er = tid >> 6, ej = tid & 63;
    const bool mine = er < nrows;
    const int b = row0 + (mine ? er : 0);
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
    float carry = (mine && grad_hn) ? grad_hn[(size_t)b * H + ej] : 0.f;
    const float* p_sv = saved + (size_t)b * T * 4 * H + ej;
    const float* p_h = hstate + (size_t)b * (T + 1) * H + ej;
    const float* p_g = grad_out + (size_t)b * T * H + ej;
    float s_r = p_sv[(size_t)(T - 1) * 4 * H], s_z = p_sv[(size_t)(T - 1) * 4 * H + H], s_n = p_sv[(size_t)(T - 1) * 4 * H + 2 * H],
          s_hn = p_sv[(size_t)(T - 1) * 4 * H + 3 * H], hprev = p_h[(size_t)(T - 1) * H], gout = p_g[(size_t)(T - 1) * H];
    __syncthreads();

    for (int t = T - 1; t >= -1; --t) {
        const int t1 = t > 0 ? t - 1 : 0;
        const float s_r1 = p_sv[(size_t)t1 * 4 * H], s_z1 = p_sv[(size_t)t1 * 4 * H + H], s_n1 = p_sv[(size_t)t1 * 4 * H + 2 * H],
                    s_hn1 = p_sv[(size_t)t1 * 4 * H + 3 * H], hprev1 = p_h[(size_t)t1 * H], gout1 = p_g[(size_t)t1 * H];
        if (t + 1 < T) {
            f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int k4 = 0; k4 < KW / 4; ++k4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(&dl[lane & 3][wave * KW + 4 * k4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], fw[4 * k4 + e], acc[e], 0, 0, 0);
            }
            const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
            for (int i = 0; i < R; ++i) part[wave][i][lane] = sum[i];
        }
        __syncthreads();
        if (mine) {
            const float dstate = carry + (t + 1 < T ? ((part[0][er][ej] + part[1][er][ej]) + (part[2][er][ej] + part[3][er][ej])) : 0.f);
            if (t < 0) {
                dh0[(size_t)b * H + ej] = dstate;
            } else {
                float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, c = dstate;
                if (t < len) {
                    const float dh = dstate + gout;
                    dn = dh * (1.f - s_z) * (1.f - s_n * s_n);
                    dz = dh * (hprev - s_n) * s_z * (1.f - s_z);
                    dr = dn * s_hn * s_r * (1.f - s_r);
                    dnr = dn * s_r;
                    c = dh * s_z;
                }
                carry = c;
                dl[er][ej] = dr;
                dl[er][H + ej] = dz;
                dl[er][2 * H + ej] = dnr;
                const size_t row = (size_t)b * T + t;
                float* dx = dxproj + row * G;
                float* dhp = dhproj + row * G;
                dx[ej] = dr;  dx[H + ej] = dz;  dx[2 * H + ej] = dn;
                dhp[ej] = dr; dhp[H + ej] = dz; dhp[2 * H + ej] = dnr;
            }
        }
        s_r = s_r1; s_z = s_z1; s_n = s_n1; s_hn = s_hn1; hprev = hprev1; gout = gout1;
        __syncthreads();
    }
}

extern "C" {

int mg_gru_small_supported(int H) { return (H == 64 || H == 128) && g_mg_tuning[MG_TUNE_PERSISTENT] != 1; }

int mg_gru_fwd_small_f32(const float* xproj, const float* w_hh, const float* b_hh, const int64_t* seq_len, int B, int T, int H, float* hstate,
                         float* out, float* saved, void* stream) {
    MG_CHECK_ARG(xproj && w_hh && b_hh && hstate && out && saved && B > 0 && T > 0 && (H == 64 || H == 128),
                 "mg_gru_fwd_small_f32: bad arguments (B=%d T=%d H=%d; H must be 64 or 128)", B, T, H);
    MG_CHECK_ARG(((uintptr_t)w_hh % 16) == 0, "mg_gru_fwd_small_f32: w_hh must be 16-byte aligned");
    const int R = 256 / H;
    const unsigned grid = (unsigned)mg_ceil_div(B, R);
    if (H == 64 && g_mg_tuning[MG_TUNE_PERSISTENT] != 2)
        hipLaunchKernelGGL(gru_fwd_small64_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, xproj, w_hh, b_hh, seq_len, B, T, hstate, out, saved);
    else if (H == 64)
        hipLaunchKernelGGL(gru_fwd_small_kernel<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, xproj, w_hh, b_hh, seq_len, B, T, R, hstate, out, saved);
    else
        hipLaunchKernelGGL(gru_fwd_small_kernel<128>, dim3(grid), dim3(256), 0, (hipStream_t)stream, xproj, w_hh, b_hh, seq_len, B, T, R, hstate, out, saved);
    MG_CHECK_LAUNCH("mg_gru_fwd_small_f32");
    return MG_OK;
}

int mg_gru_bwd_small_f32(const float* grad_out, const float* grad_hn, const float* hstate, const float* saved, const float* w_hh,
                         const int64_t* seq_len, int B, int T, int H, float* dxproj, float* dhproj, float* dh0, void* stream) {
    MG_CHECK_ARG(grad_out && hstate && saved && w_hh && dxproj && dhproj && dh0 && B > 0 && T > 0 && (H == 64 || H == 128),
                 "mg_gru_bwd_small_f32: bad arguments (B=%d T=%d H=%d; H must be 64 or 128)", B, T, H);
    const int R = 256 / H;
    const unsigned grid = (unsigned)mg_ceil_div(B, R);
    if (H == 64 && g_mg_tuning[MG_TUNE_PERSISTENT] != 2)
        hipLaunchKernelGGL(gru_bwd_small64_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_out, grad_hn, hstate, saved, w_hh, seq_len, B, T,
                           dxproj, dhproj, dh0);
    else if (H == 64)
        hipLaunchKernelGGL(gru_bwd_small_kernel<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_out, grad_hn, hstate, saved, w_hh, seq_len, B, T, R,
                           dxproj, dhproj, dh0);
    else
        hipLaunchKernelGGL(gru_bwd_small_kernel<128>, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_out, grad_hn, hstate, saved, w_hh, seq_len, B, T, R,
                           dxproj, dhproj, dh0);
    MG_CHECK_LAUNCH("mg_gru_bwd_small_f32");
    return MG_OK;
}

}  // extern "C"
