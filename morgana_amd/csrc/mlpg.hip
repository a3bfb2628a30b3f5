// K8 - maximum-likelihood parameter generation on the device (reference: morgana/viz/synthesis.py:79-178 `MLPG`, called from
// predict() of the shipped models - models/f0_test_model.py:86-89, models/RNN_SPSS.py:107-118 - i.e. once per training step,
// where the reference copies the predicted delta streams to the host, solves B x D banded systems with `bandmat` in float64 and
// copies the trajectories back to feed the LF0 / MCD metrics of loss()).
//
// Per (utterance b, feature dimension d) one symmetric positive-definite banded system of N = len_b + 2 padding unknowns:
//     P = sum_w W_w^T diag(tau_w) W_w,   rhs = sum_w W_w^T (mu_w tau_w),   x = P^-1 rhs           (synthesis.py:39-77, :164-168)
// with W_w the N x N Toeplitz matrix of window w cut off at the edges (W[s, t] = c_w[l_w + t - s]), mu / tau the padded
// (first / last frame repeated `padding` times, :113-120) means and precisions of stream column w D + d.  As in the reference
// mu / var and 1 / var are formed in float32 (numpy arithmetic on the float32 arrays the model produced, :161-165) and
// everything after that is float64 (:64-65).
//
// Two kernels:
//   mlpg_band_kernel   one thread per (frame, system): the band row P[t, t-m], m = 0..HB, and rhs[t], straight into the
//                      workspace planes [plane][frame][system] (fully parallel, coalesced over systems; HBM bound).
//   mlpg_solve_kernel  one thread per system: banded LDL^T row by row (the rows of a system form a dependent chain of
//                      HB + 1 fused multiply-adds and one reciprocal each), forward substitution in the same sweep, L and
//                      z = D^-1 L^-1 rhs written over the band planes; then the backward sweep and the float32 store of
//                      frames [padding, N - padding).  The chain is latency bound (one fp64 reciprocal + ~12 FMA per row, 2 N
//                      steps per system): rows are fetched 8 at a time, a block ahead of the arithmetic, and the reciprocal is
//                      v_rcp_f64 + two Newton steps instead of the IEEE division sequence.
// Summation order differs from bandmat's (which adds the windows band by band), so results agree to float64 rounding, not bit
// for bit: tests hold the float32 trajectories to 1e-6 relative against the float64 CPU restatement.
#include "common.h"

#define MLPG_MAX_WINDOWS 4
#define MLPG_MAX_COEFF 5

struct MlpgWindows {
    int n;
    int l[MLPG_MAX_WINDOWS], u[MLPG_MAX_WINDOWS];
    double c[MLPG_MAX_WINDOWS][MLPG_MAX_COEFF];
};

template <int HB>
__global__ __launch_bounds__(256) void mlpg_band_kernel(const float* __restrict__ means, const float* __restrict__ variances,
                                                        int var_per_frame, const int64_t* __restrict__ seq_len, int B, int T, int D,
                                                        MlpgWindows win, int padding, double* __restrict__ planes) {
    const int S = B * D, n_max = T + 2 * padding;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)n_max * S) return;
    const int t = (int)(gid / S), sys = (int)(gid - (int64_t)t * S);
    const int b = sys / D, d = sys - b * D;
    const int len = (int)(seq_len ? (seq_len[b] < T ? seq_len[b] : (int64_t)T) : (int64_t)T);
    if (len <= 0) return;
    const int n = len + 2 * padding;
    if (t >= n) return;
    const int width = win.n * D;
    double band[HB + 1], rhs = 0.0;
#pragma unroll
    for (int m = 0; m <= HB; ++m) band[m] = 0.0;
    for (int w = 0; w < win.n; ++w) {
        const int l = win.l[w], u = win.u[w];
        for (int k = -l; k <= u; ++k) {                       // row s = t - k of W_w has a coefficient in column t
            const int s = t - k;
            if (s < 0 || s >= n) continue;
            int f = s - padding;
            f = f < 0 ? 0 : (f >= len ? len - 1 : f);
            const size_t at = ((size_t)b * T + f) * width + (size_t)w * D + d;
            const float var = var_per_frame ? variances[at] : variances[w * D + d];
            const float mu_tau = means[at] / var, tau = 1.0f / var;           // float32, as numpy does on the model's arrays
            const double ck = win.c[w][l + k];
            rhs += ck * (double)mu_tau;
#pragma unroll
            for (int m = 0; m <= HB; ++m) {
                const int k2 = l + k - m;
                if (t - m >= 0 && k2 >= 0 && k2 <= l + u) band[m] += (double)tau * ck * win.c[w][k2];
            }
        }
    }
#pragma unroll
    for (int m = 0; m <= HB; ++m) planes[((size_t)m * n_max + t) * S + sys] = band[m];
    planes[((size_t)(HB + 1) * n_max + t) * S + sys] = rhs;
}

#define MLPG_ROWS 8      // rows fetched ahead per block: their loads fly while the previous block's chain is worked off

// 1 / d for d > 0 (pivots of an SPD band): v_rcp_f64 + two Newton steps, ~1 ulp - a third of the IEEE division sequence, and
// it sits on the row-to-row dependency chain
__device__ __forceinline__ double mlpg_rcp(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(y, e, y);
}

template <int HB, typename OutT>
__global__ __launch_bounds__(64) void mlpg_solve_kernel(const int64_t* __restrict__ seq_len, int B, int T, int D, int padding,
                                                       double* __restrict__ planes, OutT* __restrict__ out) {
    const int S = B * D, n_max = T + 2 * padding;
    const int sys = blockIdx.x * 64 + threadIdx.x;
    if (sys >= S) return;
    const int b = sys / D, d = sys - b * D;
    const int len = (int)(seq_len ? (seq_len[b] < T ? seq_len[b] : (int64_t)T) : (int64_t)T);
    if (len <= 0) return;
    const int n = len + 2 * padding;
    const size_t plane = (size_t)n_max * S;
    double* p = planes + sys;

    // forward: row i of L (unit lower band), d_i, y_i; L[i, i-1-k] goes to plane k + 1, z_i = y_i / d_i to plane HB + 1
    double lprev[HB][HB];        // lprev[r][k]: L[i-1-r, i-1-r-(k+1)]
    double yprev[HB], dinv[HB];
#pragma unroll
    for (int r = 0; r < HB; ++r) {
        yprev[r] = 0.0; dinv[r] = 0.0;
#pragma unroll
        for (int k = 0; k < HB; ++k) lprev[r][k] = 0.0;
    }
    // one row of the factorisation from its band row a[0..HB] and right-hand side a[HB + 1]
    auto fwd_row = [&](const double (&a)[HB + 2], int i) __attribute__((always_inline)) {
        double lw[HB], ud[HB];       // lw[k] = L[i, i-1-k], ud[k] = lw[k] * d_{i-1-k}
#pragma unroll
        for (int m = HB - 1; m >= 0; --m) {           // columns i-HB .. i-1 in increasing order
            double v = a[m + 1];
#pragma unroll
            for (int k = m + 1; k < HB; ++k) v -= ud[k] * lprev[m][k - m - 1];
            ud[m] = v;
            lw[m] = v * dinv[m];                      // dinv = 0 for rows before the first: those entries are 0 anyway
        }
        double di = a[0], yi = a[HB + 1];
#pragma unroll
        for (int k = 0; k < HB; ++k) {
            di -= lw[k] * ud[k];
            yi -= lw[k] * yprev[k];
        }
        const double inv = mlpg_rcp(di);
#pragma unroll
        for (int k = 0; k < HB; ++k) p[(k + 1) * plane + (size_t)i * S] = lw[k];
        p[(HB + 1) * plane + (size_t)i * S] = yi * inv;
#pragma unroll
        for (int q = HB - 1; q > 0; --q) {
            yprev[q] = yprev[q - 1]; dinv[q] = dinv[q - 1];
#pragma unroll
            for (int k = 0; k < HB; ++k) lprev[q][k] = lprev[q - 1][k];
        }
        yprev[0] = yi; dinv[0] = inv;
#pragma unroll
        for (int k = 0; k < HB; ++k) lprev[0][k] = lw[k];
    };
    // blocks of MLPG_ROWS rows; every load is unconditional (row index clamped) so that the block ahead is in flight while this
    // one's chain runs - with per-row branches the compiler drains the memory queue (vmcnt(0)) before the first row
    double cur[MLPG_ROWS][HB + 2], nxt[MLPG_ROWS][HB + 2];
#pragma unroll
    for (int r = 0; r < MLPG_ROWS; ++r)
#pragma unroll
        for (int m = 0; m < HB + 2; ++m) cur[r][m] = p[m * plane + (size_t)min(r, n - 1) * S];
    int i0 = 0;
    for (; i0 + MLPG_ROWS <= n; i0 += MLPG_ROWS) {
#pragma unroll
        for (int r = 0; r < MLPG_ROWS; ++r)
#pragma unroll
            for (int m = 0; m < HB + 2; ++m) nxt[r][m] = p[m * plane + (size_t)min(i0 + MLPG_ROWS + r, n - 1) * S];
#pragma unroll
        for (int r = 0; r < MLPG_ROWS; ++r) fwd_row(cur[r], i0 + r);
#pragma unroll
        for (int r = 0; r < MLPG_ROWS; ++r)
#pragma unroll
            for (int m = 0; m < HB + 2; ++m) cur[r][m] = nxt[r][m];
    }
#pragma unroll
    for (int r = 0; r < MLPG_ROWS - 1; ++r)
        if (i0 + r < n) fwd_row(cur[r], i0 + r);

    // backward: x_i = z_i - sum_k L[i+1+k, i] x_{i+1+k}; per row z_i (plane HB + 1, row i) and L[i+1+k, i] (plane k + 1, row i+1+k,
    // zero past the last row: those x are zero too, so the clamped load is harmless)
    double xn[HB];
#pragma unroll
    for (int k = 0; k < HB; ++k) xn[k] = 0.0;
    OutT* o = out + (size_t)b * T * D + d;
    auto bwd_row = [&](const double (&a)[HB + 1], int i) __attribute__((always_inline)) {
        double x = a[HB];
#pragma unroll
        for (int k = 0; k < HB; ++k) x -= a[k] * xn[k];
        if (i >= padding && i < n - padding) o[(size_t)(i - padding) * D] = (OutT)x;
#pragma unroll
        for (int k = HB - 1; k > 0; --k) xn[k] = xn[k - 1];
        xn[0] = x;
    };
    auto bwd_load = [&](double (&a)[HB + 1], int i) __attribute__((always_inline)) {
        i = max(i, 0);
        a[HB] = p[(HB + 1) * plane + (size_t)i * S];
#pragma unroll
        for (int k = 0; k < HB; ++k) a[k] = p[(k + 1) * plane + (size_t)min(i + 1 + k, n - 1) * S];
    };
    double bc[MLPG_ROWS][HB + 1], bn[MLPG_ROWS][HB + 1];
#pragma unroll
    for (int r = 0; r < MLPG_ROWS; ++r) bwd_load(bc[r], n - 1 - r);
    i0 = n - 1;
    for (; i0 - MLPG_ROWS >= -1; i0 -= MLPG_ROWS) {
#pragma unroll
        for (int r = 0; r < MLPG_ROWS; ++r) bwd_load(bn[r], i0 - MLPG_ROWS - r);
#pragma unroll
        for (int r = 0; r < MLPG_ROWS; ++r) bwd_row(bc[r], i0 - r);
#pragma unroll
        for (int r = 0; r < MLPG_ROWS; ++r)
#pragma unroll
            for (int k = 0; k <= HB; ++k) bc[r][k] = bn[r][k];
    }
#pragma unroll
    for (int r = 0; r < MLPG_ROWS - 1; ++r)
        if (i0 - r >= 0) bwd_row(bc[r], i0 - r);
}

static int mlpg_half_bandwidth(int n_windows, const int* win_l, const int* win_u) {
    int hb = 0;
    for (int w = 0; w < n_windows; ++w) hb = win_l[w] + win_u[w] > hb ? win_l[w] + win_u[w] : hb;
    return hb <= 2 ? 2 : 4;
}

extern "C" {

// planes: (HB + 2) x (T + 2 padding) x (B D) doubles
size_t mg_mlpg_workspace_bytes(int B, int T, int D, int padding, int n_windows, const int* win_l, const int* win_u) {
    if (B <= 0 || T <= 0 || D <= 0 || padding < 0 || n_windows <= 0 || !win_l || !win_u) return 0;
    const int hb = mlpg_half_bandwidth(n_windows, win_l, win_u);
    return (size_t)(hb + 2) * (size_t)(T + 2 * padding) * (size_t)B * D * sizeof(double);
}

int mg_mlpg_f32(const float* means, const float* variances, int var_per_frame, const int64_t* seq_len, int B, int T, int D,
                int n_windows, const int* win_l, const int* win_u, const double* win_coeff, int padding, void* out, int out_f64,
                void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(means && variances && out && win_l && win_u && win_coeff, "mg_mlpg_f32: null argument");
    MG_CHECK_ARG(B > 0 && T > 0 && D > 0 && padding >= 0, "mg_mlpg_f32: bad shape (B=%d T=%d D=%d padding=%d)", B, T, D, padding);
    MG_CHECK_ARG(n_windows > 0 && n_windows <= MLPG_MAX_WINDOWS, "mg_mlpg_f32: 1..%d windows supported, got %d", MLPG_MAX_WINDOWS,
                 n_windows);
    MlpgWindows win = {};
    win.n = n_windows;
    for (int w = 0; w < n_windows; ++w) {
        MG_CHECK_ARG(win_l[w] >= 0 && win_u[w] >= 0 && win_l[w] + win_u[w] + 1 <= MLPG_MAX_COEFF,
                     "mg_mlpg_f32: window %d (l=%d, u=%d) is wider than %d coefficients", w, win_l[w], win_u[w], MLPG_MAX_COEFF);
        win.l[w] = win_l[w];
        win.u[w] = win_u[w];
        for (int k = 0; k <= win_l[w] + win_u[w]; ++k) win.c[w][k] = win_coeff[w * MLPG_MAX_COEFF + k];
    }
    const size_t need = mg_mlpg_workspace_bytes(B, T, D, padding, n_windows, win_l, win_u);
    MG_CHECK_ARG((size_t)(T + 2 * padding) * (size_t)B * D < ((size_t)1 << 31), "mg_mlpg_f32: too many unknowns for 32-bit frame x system ids");
    if (!workspace || workspace_bytes < need) {
        mg_set_error("mg_mlpg_f32: workspace of %zu bytes needed, got %zu", need, workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t out_bytes = (size_t)B * T * D * (out_f64 ? sizeof(double) : sizeof(float));
    if (hipMemsetAsync(out, 0, out_bytes, st) != hipSuccess) {           // frames past seq_len stay zero (synthesis.py:152, :170)
        mg_set_error("mg_mlpg_f32: memset failed");
        return MG_ELAUNCH;
    }
    const int hb = mlpg_half_bandwidth(n_windows, win_l, win_u);
    const int64_t cells = (int64_t)(T + 2 * padding) * B * D;
    const unsigned grid_a = (unsigned)mg_ceil_div(cells, 256), grid_b = (unsigned)mg_ceil_div((int64_t)B * D, 64);
    double* planes = (double*)workspace;
#define MLPG_RUN(HB)                                                                                                                  \
    do {                                                                                                                              \
        hipLaunchKernelGGL((mlpg_band_kernel<HB>), dim3(grid_a), dim3(256), 0, st, means, variances, var_per_frame, seq_len, B, T, D,  \
                           win, padding, planes);                                                                                     \
        if (out_f64)                                                                                                                  \
            hipLaunchKernelGGL((mlpg_solve_kernel<HB, double>), dim3(grid_b), dim3(64), 0, st, seq_len, B, T, D, padding, planes,      \
                               (double*)out);                                                                                         \
        else                                                                                                                          \
            hipLaunchKernelGGL((mlpg_solve_kernel<HB, float>), dim3(grid_b), dim3(64), 0, st, seq_len, B, T, D, padding, planes,       \
                               (float*)out);                                                                                          \
    } while (0)
    if (hb == 2)
        MLPG_RUN(2);
    else
        MLPG_RUN(4);
    MG_CHECK_LAUNCH("mg_mlpg_f32");
    return MG_OK;
}

}  // extern "C"
