// K3w - a STACK of small GRU layers (H = 64: the three RecurrentCuDNNWrapper(nn.GRU(., 64)) of the shipped F0 model,
// models/f0_test_model.py:31-37) as ONE launch per direction, exact fp32: the kernels of gru_small.hip run as a wavefront over
// (layer, time).  A workgroup still owns R = 4 items outright - W_hh in registers, state in LDS, nothing to synchronise inside a
// layer - and there is one such workgroup per (layer, item block): layer l works on step t while layer l - 1 is a few steps ahead,
// so L layers take T + a few dependent steps instead of L T, and the input-projection GEMMs of the upper layers (forward) and the
// input-gradient GEMMs between the layers (backward) go: a workgroup that owns its items outright holds every gate column of
// them, so x_t W_ih^T (forward) and dxproj_t W_ih (backward) are local products with W_ih resident next to W_hh.
// Hand-off between the layers of an item block: the value itself is the flag.  The buffer a layer hands down or up ([B, T, H]
// fp32: `out` of a lower layer, `dxin` of an upper one) is filled with a sentinel (all bits set: a NaN no arithmetic produces)
// by a memset ahead of the launch; the producer stores its H values of a step write-through (sc1), each consumer thread loads
// ITS element (sc1) two steps before it needs it and re-loads while it still reads the sentinel.  No flag word, no fence, no
// drain of the producer's stores; a dword store is atomic, and every (b, t) element is written exactly once (zeros on padded steps).
// Every spin is bounded; a time-out sets the sticky status word (mg_gru_persist_status) and the workgroup stops polling.
// Arithmetic of a layer: gru_small.hip's (expf / tanhf cell, fp32 products on v_mfma_f32_4x4x1); layer 0's input projection
// comes from memory as there, the upper layers' from the in-kernel product (one chain per output, bias added behind).
#include "common.h"

#include "persist_common.h"

#define GSS_SENTINEL 0xFFFFFFFFu
// A consuming layer starts only once its producer is GSS_LAG steps into its own pass: layers of equal step time keep whatever distance
// they start with, and started together the consumer's look-ahead loads (two steps early) would find the sentinel and poll every
// step.  (Measured neutral at the shipped model's shape - 3.45 ms with and without - the polls were not what the step waits for.)
// What the forward step spends (timing probes of round 4, private builds with parts removed, T = 1000, ms of the launch's ~1.25):
// products section 0.64 (LDS reads, 16 MFMAs, LDS writes - a dependent chain on the critical path; independent accumulators: equal),
// hand-off between the layers 0.29, the step's seven result stores 0.17, next step's xproj loads 0.00.
#define GSS_LAG 6

MG_STAMP_DECL(g_stamps_gss);

struct GssLayers {
    mg_gru_stack_layer l[MG_GRU_STACK_MAX_LAYERS];
};

__device__ __forceinline__ float gss_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void gss_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the consumer's side of the hand-off: `v` was loaded a while ago; poll while it is still the sentinel
__device__ __forceinline__ float gss_take(float v, const float* p, bool& dead, gu32* status, unsigned code) {
    if (__float_as_uint(v) != GSS_SENTINEL || dead) return v;
    for (unsigned spins = 0; spins < GP_SPIN_LIMIT; ++spins) {
        v = gss_load(p);
        if (__float_as_uint(v) != GSS_SENTINEL) return v;
        __builtin_amdgcn_s_sleep(1);
    }
    dead = true;
    __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0.f;
}

// FAST (throughput mode, mg_gru_stack_fwd_small_fast_f32): the cell's sigmoids and tanh on v_exp_f32 / v_rcp_f32 as in the bf16-mode
// GRU-512 recurrence (gru_cell.h) instead of expf / tanhf - the library calls are ~150 instructions of the step's dependent chain - and
// the step's products on v_mfma_f32_16x16x32_bf16 with bf16 operands (W_hh / W_ih fragments, the state and the lower layer's output
// rounded as that recurrence rounds them), fp32 accumulation: the exact-fp32 form issues 64 (layer 0) / 128 (above) v_mfma_f32_4x4x1
// per wave and step - stamps (scripts/stamps_gru_small.py): 1,135 / 2,027 of the step's 2,299 / 3,296 cycles - the bf16 form 8 / 16
// MFMAs of 16 rows (4 items + 12 zero rows).  The state itself, the cell and everything stored stay fp32.
typedef __bf16 gss_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int gss_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ gss_bf8 gss_as_bf8(gss_u32x4 v) {
    union { gss_u32x4 u; gss_bf8 b; } c;
    c.u = v;
    return c.b;
}
// 8 consecutive fp32 values -> 8 bf16 (a weight fragment, converted once per launch)
__device__ __forceinline__ gss_bf8 gss_cvt8(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    gss_u32x4 u;
    u[0] = (unsigned)mg_f2bf(a[0]) | ((unsigned)mg_f2bf(a[1]) << 16);
    u[1] = (unsigned)mg_f2bf(a[2]) | ((unsigned)mg_f2bf(a[3]) << 16);
    u[2] = (unsigned)mg_f2bf(b[0]) | ((unsigned)mg_f2bf(b[1]) << 16);
    u[3] = (unsigned)mg_f2bf(b[2]) | ((unsigned)mg_f2bf(b[3]) << 16);
    return gss_as_bf8(u);
}
template <bool FAST>
__global__ __launch_bounds__(512) void gru_stack_fwd_small64_kernel(GssLayers a, const int64_t* __restrict__ seq_len, int B, int T, int L,
                                                                    int nblk, unsigned* sync) {
    constexpr int H = 64, G = 192, R = 4, LDH = H + 4, LDG = G + 4;
    __shared__ __attribute__((aligned(16))) float hs[R][LDH];       // h_{t-1}; rows of missing items stay zero
    __shared__ __attribute__((aligned(16))) float xs[2][R][LDH];    // upper layers: the lower layer's output of step t (parity t & 1)
    __shared__ __attribute__((aligned(16))) float gl[R][LDG];       // recurrent pre-activations of the three gates
    __shared__ __attribute__((aligned(16))) float gx[R][LDG];       // upper layers: input pre-activations
    // Waves 4 .. 7 only STORE: a step leaves seven values per element for after the launch (state, output, the saved gates) on lines
    // nobody has touched - their acknowledgements take about a step - and a wave's loads and stores complete in order (one vmcnt): in
    // the waves that also load (the next step's input rows, the lower layer's outputs) every such wait was a wait for old stores too
    // (probe: 0.17 ms of the launch's 1.25).  The computing waves 0 .. 3 leave the values in `stg`; wave 4 + i stores item i's.
    __shared__ float stg[7][R][H];
    constexpr int LDB = H + 8;                                       // FAST: bf16 copies of hs / xs, the MFMA A operands
    __shared__ __attribute__((aligned(16))) uint16_t hb[R][LDB];
    __shared__ __attribute__((aligned(16))) uint16_t xsb[2][R][LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int layer = blockIdx.x / nblk, blk = blockIdx.x - layer * nblk;
    const mg_gru_stack_layer& P = a.l[layer];
    const bool upper = layer > 0, hands_up = layer + 1 < L;
    gu32* status = (gu32*)sync + GP_FLAG_WORDS;
    const int row0 = blk * R;
    const int nrows = min(R, B - row0);
    // wave g < 3 owns gate g: lane L holds row g H + L of W_hh (and of W_ih above the first layer) for the whole launch
    f32x4 fw[FAST ? 1 : H / 4], fwi[FAST ? 1 : H / 4];
    // FAST: the 192 gate columns are 12 tiles of 16, three per wave (all four waves); tile nt of this wave, k-step ks: row
    // 16 (3 wave + nt) + li of W, columns 32 ks + 8 q .. + 7
    gss_bf8 fwb[3][2], fwib[3][2];
    if (FAST && wave < 4) {
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fwb[nt][ks] = gss_cvt8(P.w_hh + (size_t)(16 * (3 * wave + nt) + li) * H + 32 * ks + 8 * q);
                fwib[nt][ks] = upper ? gss_cvt8(P.w_ih + (size_t)(16 * (3 * wave + nt) + li) * H + 32 * ks + 8 * q) : fwb[nt][ks];
            }
    }
    if (!FAST && wave < 3) {
        const float* wp = P.w_hh + (size_t)(wave * H + lane) * H;
#pragma unroll
        for (int k4 = 0; k4 < H / 4; ++k4) {
            fw[k4] = *reinterpret_cast<const f32x4*>(wp + 4 * k4);
            fwi[k4] = fw[k4];
        }
        if (upper) {
            const float* wi = P.w_ih + (size_t)(wave * H + lane) * H;
#pragma unroll
            for (int k4 = 0; k4 < H / 4; ++k4) fwi[k4] = *reinterpret_cast<const f32x4*>(wi + 4 * k4);
        }
    }
    for (int e = tid; e < R * LDH; e += 512) {
        (&hs[0][0])[e] = 0.f;
        (&xs[0][0][0])[e] = 0.f;
        (&xs[1][0][0])[e] = 0.f;
    }
    for (int e = tid; e < R * LDB; e += 512) {
        (&hb[0][0])[e] = 0;
        (&xsb[0][0][0])[e] = 0;
        (&xsb[1][0][0])[e] = 0;
    }
    __syncthreads();
    const int er = tid >> 6, ej = tid & 63;                          // cell role: item er, unit ej
    const bool mine = er < nrows;
    const int b = row0 + (mine ? er : 0);
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
    const float bhr = P.b_hh[ej], bhz = P.b_hh[H + ej], bhn = P.b_hh[2 * H + ej];
    const float bir = upper ? P.b_ih[ej] : 0.f, biz = upper ? P.b_ih[H + ej] : 0.f, bin = upper ? P.b_ih[2 * H + ej] : 0.f;
    float hprev = mine ? P.hstate[((size_t)b * (T + 1)) * H + ej] : 0.f;
    if (mine) {
        hs[er][ej] = hprev;
        hb[er][ej] = mg_f2bf(hprev);
    }
    const float* xp = (upper ? P.b_hh : P.xproj + (size_t)b * T * G) + (upper ? 0 : ej);        // layer 0: projected input rows
    const float* xin = upper ? a.l[layer - 1].out + (size_t)b * T * H + ej : P.b_hh;             // above: the lower layer's outputs
    float xr = 0.f, xz = 0.f, xn = 0.f;
    float xa = 0.f, xb = 0.f;                                         // above: x of steps t + 1 and t + 2, requested two steps ahead
    bool dead = false;
    if (!upper) {
        xr = xp[0], xz = xp[H], xn = xp[2 * H];
    } else if (mine) {
        const int t_lag = T - 1 < GSS_LAG ? T - 1 : GSS_LAG;
        (void)gss_take(gss_load(xin + (size_t)t_lag * H), xin + (size_t)t_lag * H, dead, status, 8u);     // the start lag (see GSS_LAG)
        float x0 = gss_load(xin);
        xa = T > 1 ? gss_load(xin + H) : 0.f;
        xb = T > 2 ? gss_load(xin + 2 * H) : 0.f;
        xs[0][er][ej] = gss_take(x0, xin, dead, status, 8u);
        xsb[0][er][ej] = mg_f2bf(xs[0][er][ej]);
    }
    __syncthreads();

    if (wave >= 4) {
        // the storing waves' whole launch: the two barriers of every step, then item wave - 4's values of the step out of `stg` (read
        // before the next barrier: the cell phase behind it overwrites them).  They never wait for memory.
        const int si = wave - 4, sb = row0 + (si < nrows ? si : 0);
        for (int t = 0; t < T; ++t) {
            gp_lds_barrier();
            gp_lds_barrier();
            if (si < nrows) {
                const float v_h = stg[0][si][lane], v_o = stg[1][si][lane], v_r = stg[2][si][lane], v_z = stg[3][si][lane],
                            v_n = stg[4][si][lane], v_hn = stg[5][si][lane];
                const size_t row = (size_t)sb * T + t;
                P.hstate[((size_t)sb * (T + 1) + t + 1) * H + lane] = v_h;
                if (hands_up)
                    gss_store(P.out + row * H + lane, v_o);
                else
                    P.out[row * H + lane] = v_o;
                float* sv = P.saved + row * 4 * H;
                sv[lane] = v_r;
                sv[H + lane] = v_z;
                sv[2 * H + lane] = v_n;
                sv[3 * H + lane] = v_hn;
            }
        }
        return;
    }
    // one step; x_use holds x_{t+1} (consumed at the end of the step, then reloaded with x_{t+3})
#ifdef MG_STAMPS
    unsigned long long ta = 0, tb = 0, ts0 = 0, ts1 = 0, tr0 = 0, tr1 = 0, sum_mm = 0, sum_b1 = 0, sum_cell = 0, sum_take = 0, sum_b2 = 0;
    MG_STAMP(ts0);
    MG_STAMP_REAL(tr0);
#endif
    auto step = [&](int t, float& x_use) {
        const int t1 = t + 1 < T ? t + 1 : t;
        float xr1 = 0.f, xz1 = 0.f, xn1 = 0.f;
        MG_STAMP(ta);
        if (!upper) xr1 = xp[(size_t)t1 * G], xz1 = xp[(size_t)t1 * G + H], xn1 = xp[(size_t)t1 * G + 2 * H];
        if (FAST && wave < 4) {
            // rows = the 4 items (A rows 4 .. 15 zero), 16 columns of the wave's gate per tile, K = 64 in two MFMAs
            const gss_u32x4 zero = {0u, 0u, 0u, 0u};
            gss_bf8 ah[2], ax[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ah[ks] = gss_as_bf8(li < R ? *reinterpret_cast<const gss_u32x4*>(&hb[li & 3][32 * ks + 8 * q]) : zero);
                ax[ks] = gss_as_bf8((upper && li < R) ? *reinterpret_cast<const gss_u32x4*>(&xsb[t & 1][li & 3][32 * ks + 8 * q]) : zero);
            }
            // W as the A operand, the items as the columns: a lane (item li, q) then holds four CONSECUTIVE gate columns 4 q + r of its
            // item - one 16-byte LDS write per tile instead of four 4-byte ones from the lanes of one row
            f32x4 acc[3], acx[3];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) acc[nt] = acx[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwb[nt][ks], ah[ks], acc[nt], 0, 0, 0);
                if (upper) {
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) acx[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwib[nt][ks], ax[ks], acx[nt], 0, 0, 0);
                }
            }
            if (li < R) {
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    *reinterpret_cast<f32x4*>(&gl[li][16 * (3 * wave + nt) + 4 * q]) = acc[nt];
                    if (upper) *reinterpret_cast<f32x4*>(&gx[li][16 * (3 * wave + nt) + 4 * q]) = acx[nt];
                }
            }
        }
        if (!FAST && wave < 3) {
            f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int k4 = 0; k4 < H / 4; ++k4) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(&hs[lane & 3][4 * k4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], fw[k4][e], acc[e], 0, 0, 0);
            }
            const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
            for (int i = 0; i < R; ++i) gl[i][wave * H + lane] = sum[i];
            if (upper) {
                f32x4 acx[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int k4 = 0; k4 < H / 4; ++k4) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(&xs[t & 1][lane & 3][4 * k4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acx[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], fwi[k4][e], acx[e], 0, 0, 0);
                }
                const f32x4 sx = (acx[0] + acx[1]) + (acx[2] + acx[3]);
#pragma unroll
                for (int i = 0; i < R; ++i) gx[i][wave * H + lane] = sx[i];
            }
        }
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_mm, tb, ta);
        gp_lds_barrier();
        MG_STAMP(ta);
        MG_STAMP_ADD(sum_b1, ta, tb);
        if (mine) {
            if (upper) xr = gx[er][ej] + bir, xz = gx[er][H + ej] + biz, xn = gx[er][2 * H + ej] + bin;
            const float hr = gl[er][ej] + bhr, hz = gl[er][H + ej] + bhz, hn = gl[er][2 * H + ej] + bhn;
            const float r = FAST ? mg_sigmoid_fast(xr + hr) : mg_sigmoid(xr + hr);
            const float z = FAST ? mg_sigmoid_fast(xz + hz) : mg_sigmoid(xz + hz);
            const float n = FAST ? 2.f * mg_sigmoid_fast(2.f * (xn + r * hn)) - 1.f : tanhf(xn + r * hn);
            const float hnew = (1.f - z) * n + z * hprev;
            const bool active = t < len;
            hprev = active ? hnew : hprev;
            hs[er][ej] = hprev;
            if (FAST) hb[er][ej] = mg_f2bf(hprev);
            stg[0][er][ej] = hprev;
            stg[1][er][ej] = active ? hnew : 0.f;
            stg[2][er][ej] = r;
            stg[3][er][ej] = z;
            stg[4][er][ej] = n;
            stg[5][er][ej] = hn;
            MG_STAMP(tb);
            MG_STAMP_ADD(sum_cell, tb, ta);
            if (upper && t + 1 < T) {
                const float xv = gss_take(x_use, xin + (size_t)(t + 1) * H, dead, status, 8u);
                xs[(t + 1) & 1][er][ej] = xv;
                if (FAST) xsb[(t + 1) & 1][er][ej] = mg_f2bf(xv);
                if (t + 3 < T) x_use = gss_load(xin + (size_t)(t + 3) * H);
            }
            MG_STAMP(ta);
            MG_STAMP_ADD(sum_take, ta, tb);
        }
        xr = xr1;
        xz = xz1;
        xn = xn1;
        gp_lds_barrier();
        MG_STAMP(tb);
        MG_STAMP_ADD(sum_b2, tb, ta);
    };
    int t = 0;
    for (; t + 1 < T; t += 2) {
        step(t, xa);
        step(t + 1, xb);
    }
    if (t < T) step(t, xa);
#ifdef MG_STAMPS
    MG_STAMP(ts1);
    MG_STAMP_REAL(tr1);
    MG_STAMP_STORE(g_stamps_gss, blockIdx.x, wave, lane, 0, ts0);
    MG_STAMP_STORE(g_stamps_gss, blockIdx.x, wave, lane, 1, ts1);
    MG_STAMP_STORE(g_stamps_gss, blockIdx.x, wave, lane, 2, tr0);
    MG_STAMP_STORE(g_stamps_gss, blockIdx.x, wave, lane, 3, tr1);
    MG_STAMP_STORE(g_stamps_gss, blockIdx.x, wave, lane, 4, sum_mm);
    MG_STAMP_STORE(g_stamps_gss, blockIdx.x, wave, lane, 5, sum_b1);
    MG_STAMP_STORE(g_stamps_gss, blockIdx.x, wave, lane, 6, sum_cell);
    MG_STAMP_STORE(g_stamps_gss, blockIdx.x, wave, lane, 7, sum_take);
    MG_STAMP_STORE(g_stamps_gss, blockIdx.x, wave, lane, 8, sum_b2);
    MG_STAMP_STORE(g_stamps_gss, blockIdx.x, wave, lane, 9, (unsigned long long)layer);
#endif
}

// Backward.  dl = dhproj_{t+1} = (dr, dz, dn r) feeds dstate_t = carry + dl W_hh as in gru_bwd_small64_kernel; above the first layer
// dlx = dxproj_{t+1} = (dr, dz, dn) also goes through the layer's own W_ih in the same phase: d x_{t+1} = dlx W_ih is what the layer
// below adds to its state gradient in place of a grad_out row, and is handed down through `dxin` (sentinel protocol above).
// FAST (throughput mode, mg_gru_stack_bwd_small_fast_f32): the two products of a step on v_mfma_f32_16x16x32_bf16 - gate gradients and
// W rounded to bf16 as the GRU-512 backward rounds them, fp32 accumulation; wave w owns output columns 16 w .. 16 w + 15 over the whole
// contraction (6 MFMAs per product instead of 48 v_mfma_f32_4x4x1 and a four-way partial sum) - W^T as the A operand, the items as
// columns, so a lane writes its item's four consecutive columns as one 16-byte LDS store (see the forward kernel).
template <bool FAST>
__global__ __launch_bounds__(512) void gru_stack_bwd_small64_kernel(GssLayers a, const int64_t* __restrict__ seq_len, int B, int T, int L,
                                                                    int nblk, unsigned* sync) {
    constexpr int H = 64, G = 192, R = 4, LDG = G + 4, KW = G / 4;  // KW = 48 gate rows per wave
    __shared__ __attribute__((aligned(16))) float dl[R][LDG];
    __shared__ __attribute__((aligned(16))) float dlx[R][LDG];
    __shared__ __attribute__((aligned(16))) float part[4][R][H];
    __shared__ __attribute__((aligned(16))) float partx[4][R][H];
    constexpr int LDGB = G + 8;
    __shared__ __attribute__((aligned(16))) uint16_t dlb[R][LDGB];      // FAST: bf16 copies of dl / dlx, the MFMA B operands
    __shared__ __attribute__((aligned(16))) uint16_t dlxb[R][LDGB];
    __shared__ float stg[5][R][H];                   // dr, dz, dn, dn r, d x_{t+1}: what waves 4 .. 7 store (see the forward kernel)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const int layer = blockIdx.x / nblk, blk = blockIdx.x - layer * nblk;
    const mg_gru_stack_layer& P = a.l[layer];
    const bool upper = layer > 0, top = layer + 1 == L;
    gu32* status = (gu32*)sync + GP_FLAG_WORDS;
    const int row0 = blk * R;
    const int nrows = min(R, B - row0);
    float fw[FAST ? 1 : KW], fwi[FAST ? 1 : KW];
    // FAST: A operand = W^T: row = output column 16 wave + li, k-step ks: gate rows 32 ks + 8 q .. + 7 (a strided gather, once)
    gss_bf8 fwb[6], fwib[6];
    if (FAST && wave < 4) {
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            gss_u32x4 u, ui;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const size_t g0 = (size_t)(32 * ks + 8 * q + 2 * j) * H + 16 * wave + li;
                u[j] = (unsigned)mg_f2bf(P.w_hh[g0]) | ((unsigned)mg_f2bf(P.w_hh[g0 + H]) << 16);
                ui[j] = upper ? ((unsigned)mg_f2bf(P.w_ih[g0]) | ((unsigned)mg_f2bf(P.w_ih[g0 + H]) << 16)) : 0u;
            }
            fwb[ks] = gss_as_bf8(u);
            fwib[ks] = gss_as_bf8(ui);
        }
    } else if (wave < 4) {
#pragma unroll
        for (int k = 0; k < (FAST ? 1 : KW); ++k) {
            fw[k] = P.w_hh[(size_t)(wave * KW + k) * H + lane];
            fwi[k] = upper ? P.w_ih[(size_t)(wave * KW + k) * H + lane] : 0.f;
        }
    }
    for (int e = tid; e < R * LDG; e += 512) {
        (&dl[0][0])[e] = 0.f;
        (&dlx[0][0])[e] = 0.f;
    }
    for (int e = tid; e < R * LDGB; e += 512) {
        (&dlb[0][0])[e] = 0;
        (&dlxb[0][0])[e] = 0;
    }
    for (int e = tid; e < 4 * R * H; e += 512) {          // FAST writes part[0] / partx[0] only: the other three stay zero
        (&part[0][0][0])[e] = 0.f;
        (&partx[0][0][0])[e] = 0.f;
    }
    const int er = tid >> 6, ej = tid & 63;
    const bool mine = er < nrows;
    const int b = row0 + (mine ? er : 0);
    const int len = seq_len ? (int)min((int64_t)T, seq_len[b]) : T;
    float carry = (mine && P.grad_hn) ? P.grad_hn[(size_t)b * H + ej] : 0.f;
    const float* p_sv = P.saved + (size_t)b * T * 4 * H + ej;
    const float* p_h = P.hstate + (size_t)b * (T + 1) * H + ej;
    const float* p_g = (top ? P.grad_out : P.dxin) + (size_t)b * T * H + ej;       // below the top: what the layer above hands down
    float* p_dx = upper ? a.l[layer - 1].dxin + (size_t)b * T * H + ej : (float*)nullptr;
    float s_r = p_sv[(size_t)(T - 1) * 4 * H], s_z = p_sv[(size_t)(T - 1) * 4 * H + H], s_n = p_sv[(size_t)(T - 1) * 4 * H + 2 * H],
          s_hn = p_sv[(size_t)(T - 1) * 4 * H + 3 * H], hprev = p_h[(size_t)(T - 1) * H];
    // gradient rows of steps T - 1 and T - 2, requested ahead (below the top: polled when taken)
    float ga = 0.f, gb = 0.f;
    bool dead = false;
    if (mine) {
        if (!top) {                                  // the start lag (see GSS_LAG): the layer above is GSS_LAG steps down its pass
            const int t_lag = T - 1 - GSS_LAG > 0 ? T - 1 - GSS_LAG : 0;
            (void)gss_take(gss_load(p_g + (size_t)t_lag * H), p_g + (size_t)t_lag * H, dead, status, 9u);
        }
        ga = top ? p_g[(size_t)(T - 1) * H] : gss_load(p_g + (size_t)(T - 1) * H);
        if (T > 1) gb = top ? p_g[(size_t)(T - 2) * H] : gss_load(p_g + (size_t)(T - 2) * H);
    }
    __syncthreads();

    if (wave >= 4) {
        // the storing waves' whole launch (see the forward kernel): steps T - 1 .. -1, two barriers each, then item wave - 4's values
        const int si = wave - 4, sb = row0 + (si < nrows ? si : 0);
        float* sdx = upper ? a.l[layer - 1].dxin + (size_t)sb * T * H + lane : (float*)nullptr;
        for (int t = T - 1; t >= -1; --t) {
            gp_lds_barrier();
            gp_lds_barrier();
            if (si < nrows) {
                if (upper && t + 1 < T) gss_store(sdx + (size_t)(t + 1) * H, stg[4][si][lane]);
                if (t >= 0) {
                    const float v_r = stg[0][si][lane], v_z = stg[1][si][lane], v_n = stg[2][si][lane], v_nr = stg[3][si][lane];
                    const size_t row = (size_t)sb * T + t;
                    float* dx = P.dxproj + row * G;
                    float* dhp = P.dhproj + row * G;
                    dx[lane] = v_r;  dx[H + lane] = v_z;  dx[2 * H + lane] = v_n;
                    dhp[lane] = v_r; dhp[H + lane] = v_z; dhp[2 * H + lane] = v_nr;
                }
            }
        }
        return;
    }
    // one step; g_use holds grad row t (taken in the cell phase, then reloaded with row t - 2)
    auto step = [&](int t, float& g_use) {
        const int t1 = t > 0 ? t - 1 : 0;
        const float s_r1 = p_sv[(size_t)t1 * 4 * H], s_z1 = p_sv[(size_t)t1 * 4 * H + H], s_n1 = p_sv[(size_t)t1 * 4 * H + 2 * H],
                    s_hn1 = p_sv[(size_t)t1 * 4 * H + 3 * H], hprev1 = p_h[(size_t)t1 * H];
        if (FAST && t + 1 < T) {
            const gss_u32x4 zero = {0u, 0u, 0u, 0u};
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc, acx = acc, acx2 = acc;
#pragma unroll
            for (int ks = 0; ks < 6; ks += 2) {
                const gss_bf8 b0 = gss_as_bf8(li < R ? *reinterpret_cast<const gss_u32x4*>(&dlb[li & 3][32 * ks + 8 * q]) : zero);
                const gss_bf8 b1 = gss_as_bf8(li < R ? *reinterpret_cast<const gss_u32x4*>(&dlb[li & 3][32 * (ks + 1) + 8 * q]) : zero);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwb[ks], b0, acc, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwb[ks + 1], b1, acc2, 0, 0, 0);
                if (upper) {
                    const gss_bf8 x0 = gss_as_bf8(li < R ? *reinterpret_cast<const gss_u32x4*>(&dlxb[li & 3][32 * ks + 8 * q]) : zero);
                    const gss_bf8 x1 = gss_as_bf8(li < R ? *reinterpret_cast<const gss_u32x4*>(&dlxb[li & 3][32 * (ks + 1) + 8 * q]) : zero);
                    acx = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwib[ks], x0, acx, 0, 0, 0);
                    acx2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fwib[ks + 1], x1, acx2, 0, 0, 0);
                }
            }
            if (li < R) {                            // lane (item li, q): columns 16 wave + 4 q .. + 3
                *reinterpret_cast<f32x4*>(&part[0][li][16 * wave + 4 * q]) = acc + acc2;
                if (upper) *reinterpret_cast<f32x4*>(&partx[0][li][16 * wave + 4 * q]) = acx + acx2;
            }
        }
        if (!FAST && t + 1 < T) {
            f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int k4 = 0; k4 < KW / 4; ++k4) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(&dl[lane & 3][wave * KW + 4 * k4]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], fw[4 * k4 + e], acc[e], 0, 0, 0);
            }
            const f32x4 sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
            for (int i = 0; i < R; ++i) part[wave][i][lane] = sum[i];
            if (upper) {
                f32x4 acx[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int k4 = 0; k4 < KW / 4; ++k4) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(&dlx[lane & 3][wave * KW + 4 * k4]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acx[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[e], fwi[4 * k4 + e], acx[e], 0, 0, 0);
                }
                const f32x4 sx = (acx[0] + acx[1]) + (acx[2] + acx[3]);
#pragma unroll
                for (int i = 0; i < R; ++i) partx[wave][i][lane] = sx[i];
            }
        }
        gp_lds_barrier();
        if (mine) {
            if (upper && t + 1 < T)           // d x_{t+1}: the row the layer below takes as its grad_out of step t + 1
                stg[4][er][ej] = (partx[0][er][ej] + partx[1][er][ej]) + (partx[2][er][ej] + partx[3][er][ej]);
            const float dstate = carry + (t + 1 < T ? ((part[0][er][ej] + part[1][er][ej]) + (part[2][er][ej] + part[3][er][ej])) : 0.f);
            if (t < 0) {
                P.dh0[(size_t)b * H + ej] = dstate;
            } else {
                const float gout = top ? g_use : gss_take(g_use, p_g + (size_t)t * H, dead, status, 9u);
                if (t >= 2) g_use = top ? p_g[(size_t)(t - 2) * H] : gss_load(p_g + (size_t)(t - 2) * H);
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
                if (upper) {
                    dlx[er][ej] = dr;
                    dlx[er][H + ej] = dz;
                    dlx[er][2 * H + ej] = dn;
                }
                if (FAST) {
                    dlb[er][ej] = mg_f2bf(dr);
                    dlb[er][H + ej] = mg_f2bf(dz);
                    dlb[er][2 * H + ej] = mg_f2bf(dnr);
                    if (upper) {
                        dlxb[er][ej] = mg_f2bf(dr);
                        dlxb[er][H + ej] = mg_f2bf(dz);
                        dlxb[er][2 * H + ej] = mg_f2bf(dn);
                    }
                }
                stg[0][er][ej] = dr;
                stg[1][er][ej] = dz;
                stg[2][er][ej] = dn;
                stg[3][er][ej] = dnr;
            }
        }
        s_r = s_r1; s_z = s_z1; s_n = s_n1; s_hn = s_hn1; hprev = hprev1;
        gp_lds_barrier();
    };
    int t = T - 1;
    for (; t >= 1; t -= 2) {
        step(t, ga);
        step(t - 1, gb);
    }
    // t is 0 (T odd: one real step left, then the state-gradient step) or -1 (T even)
    if (t == 0) {
        step(0, ga);
        step(-1, gb);
    } else {
        step(-1, ga);
    }
}

extern "C" {

int mg_gru_stack_small_supported(int B, int T, int H, int L) {
    if (B <= 0 || T <= 0 || H != 64 || L < 2 || L > MG_GRU_STACK_MAX_LAYERS || g_mg_tuning[MG_TUNE_PERSISTENT] == 1) return 0;
    // every (layer, item block) workgroup must be resident at once: one per CU is plenty at these sizes
    return gp_device_holds(2L * L * mg_ceil_div(B, 4));
}

size_t mg_gru_stack_small_workspace_bytes(void) { return (size_t)GP_SYNC_WORDS * sizeof(unsigned); }

static int gss_check(const char* who, const mg_gru_stack_layer* layers, int L, int B, int T, int H, void* workspace, size_t workspace_bytes) {
    MG_CHECK_ARG(layers && mg_gru_stack_small_supported(B, T, H, L), "%s: unsupported shape (B=%d T=%d H=%d L=%d): H = 64, 2..%d layers", who, B, T,
                 H, L, MG_GRU_STACK_MAX_LAYERS);
    if (!workspace || workspace_bytes < mg_gru_stack_small_workspace_bytes()) {
        mg_set_error("%s: workspace of %zu bytes needed, got %zu", who, mg_gru_stack_small_workspace_bytes(), workspace_bytes);
        return MG_EWORKSPACE;
    }
    return MG_OK;
}

static int gss_fwd(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                   size_t workspace_bytes, void* stream, bool fast);

int mg_gru_stack_fwd_small_f32(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                               size_t workspace_bytes, void* stream) {
    return gss_fwd(layers, L, seq_len, B, T, H, workspace, workspace_bytes, stream, false);
}

// The same wavefront with the cell's sigmoid / tanh on the hardware's exp and reciprocal (1 ulp each): the throughput-mode ("bf16"
// precision) form; matrix products stay exact fp32.  The backward launch is the same for both (it works from the saved gates).
int mg_gru_stack_fwd_small_fast_f32(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return gss_fwd(layers, L, seq_len, B, T, H, workspace, workspace_bytes, stream, true);
}

static int gss_fwd(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                   size_t workspace_bytes, void* stream, bool fast) {
    const int rc = gss_check("mg_gru_stack_fwd_small_f32", layers, L, B, T, H, workspace, workspace_bytes);
    if (rc != MG_OK) return rc;
    GssLayers a;
    hipStream_t st = (hipStream_t)stream;
    for (int l = 0; l < L; ++l) {
        a.l[l] = layers[l];
        const mg_gru_stack_layer& p = layers[l];
        MG_CHECK_ARG(p.w_hh && p.b_hh && p.hstate && p.out && p.saved && (l == 0 ? p.xproj != nullptr : (p.w_ih && p.b_ih)),
                     "mg_gru_stack_fwd_small_f32: layer %d: bad arguments", l);
        MG_CHECK_ARG((((uintptr_t)p.w_hh | (uintptr_t)p.w_ih) % 16) == 0, "mg_gru_stack_fwd_small_f32: layer %d: weights must be 16-byte aligned", l);
        // the hand-off buffer of every layer but the top one starts as all-sentinel
        if (l + 1 < L && hipMemsetAsync(p.out, 0xFF, (size_t)B * T * H * sizeof(float), st) != hipSuccess) {
            mg_set_error("mg_gru_stack_fwd_small_f32: memset failed");
            return MG_ELAUNCH;
        }
    }
    const int nblk = (int)mg_ceil_div(B, 4);
    if (fast)
        hipLaunchKernelGGL(gru_stack_fwd_small64_kernel<true>, dim3((unsigned)(L * nblk)), dim3(512), 0, st, a, seq_len, B, T, L, nblk, (unsigned*)workspace);
    else
        hipLaunchKernelGGL(gru_stack_fwd_small64_kernel<false>, dim3((unsigned)(L * nblk)), dim3(512), 0, st, a, seq_len, B, T, L, nblk, (unsigned*)workspace);
    MG_CHECK_LAUNCH("mg_gru_stack_fwd_small_f32");
    return MG_OK;
}

static int gss_bwd(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                   size_t workspace_bytes, void* stream, bool fast);

int mg_gru_stack_bwd_small_f32(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                               size_t workspace_bytes, void* stream) {
    return gss_bwd(layers, L, seq_len, B, T, H, workspace, workspace_bytes, stream, false);
}

// The throughput-mode ("bf16" precision) backward: the same wavefront with the step's products on bf16 MFMAs (operands rounded to
// bf16, fp32 accumulation), as mg_gru_stack_fwd_small_fast_f32's.
int mg_gru_stack_bwd_small_fast_f32(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    return gss_bwd(layers, L, seq_len, B, T, H, workspace, workspace_bytes, stream, true);
}

static int gss_bwd(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                   size_t workspace_bytes, void* stream, bool fast) {
    const int rc = gss_check("mg_gru_stack_bwd_small_f32", layers, L, B, T, H, workspace, workspace_bytes);
    if (rc != MG_OK) return rc;
    GssLayers a;
    hipStream_t st = (hipStream_t)stream;
    for (int l = 0; l < L; ++l) {
        a.l[l] = layers[l];
        const mg_gru_stack_layer& p = layers[l];
        MG_CHECK_ARG(p.w_hh && p.hstate && p.saved && p.dxproj && p.dhproj && p.dh0 && (l == 0 || p.w_ih) &&
                         (l + 1 == L ? p.grad_out != nullptr : p.dxin != nullptr),
                     "mg_gru_stack_bwd_small_f32: layer %d: bad arguments", l);
        if (l + 1 < L && hipMemsetAsync(p.dxin, 0xFF, (size_t)B * T * H * sizeof(float), st) != hipSuccess) {
            mg_set_error("mg_gru_stack_bwd_small_f32: memset failed");
            return MG_ELAUNCH;
        }
    }
    const int nblk = (int)mg_ceil_div(B, 4);
    if (fast)
        hipLaunchKernelGGL(gru_stack_bwd_small64_kernel<true>, dim3((unsigned)(L * nblk)), dim3(512), 0, st, a, seq_len, B, T, L, nblk, (unsigned*)workspace);
    else
        hipLaunchKernelGGL(gru_stack_bwd_small64_kernel<false>, dim3((unsigned)(L * nblk)), dim3(512), 0, st, a, seq_len, B, T, L, nblk, (unsigned*)workspace);
    MG_CHECK_LAUNCH("mg_gru_stack_bwd_small_f32");
    return MG_OK;
}

}  // extern "C"

#ifdef MG_STAMPS
extern "C" int mg_diag_read_stamps_gss(void* dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps_gss), bytes < sizeof(g_stamps_gss) ? bytes : sizeof(g_stamps_gss), 0, hipMemcpyDeviceToHost);
}
#endif
