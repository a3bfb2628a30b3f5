// The README F0Model's tail in the EXACT-fp32 modes, on table rows that each stand for a group of frames: from the 128-wide layer's
// pre-activations Z2 to the loss and back in ONE launch -
//     h2 = sigmoid(Z2);  h3 = sigmoid(h2 W3^T + b3);  p = h3 W4^T + b4;  loss = sum_m weight[m] (p[m] - ybar[m])^2
//     dp = 2 weight (p - ybar);  dz3 = dp W4 * h3 (1 - h3);  dW4, db4, dW3 = dz3^T h2, db3;  dZ2 = (dz3 W3) * h2 (1 - h2)
// Reference: nn.Linear(128, 32) -> nn.Sigmoid -> nn.Linear(32, 1) of README.rst:65-73 run by SequentialWithRecurrent.forward
// (morgana/utils.py:401-418), losses.mse (morgana/losses.py:29-51) in its per-phone form (csrc/phone_rate.hip: weight / ybar from
// mg_phone_front; the loss's constant term is added by mg_expand_column_loss_f32) and their autograd backward.  The fp32 and bf16x3
// modes ran these two layers as ten launches of the generic exact-fp32 kernels (two GEMMs forward; two weight gradients with two
// reduces each and two dgrads backward: ~90 us of the 0.53 ms step for 0.3 GFLOP) - this is their counterpart of mg_f0_tail_rows_bf16.
//
// Products: v_mfma_f32_16x16x4_f32 (exact fp32 fma chains).  A wave owns 16-row tiles; both layouts of W3 a wave multiplies with stay in
// its registers for the whole launch (64 + 64), as do its 32 x 128 partial sums of dW3 (64).  Fragment conventions (as gru.hip): A operand
// lane (li, q) = A[row li][k = q], B operand = B[k = q][col li], C/D register r = C[row 4 q + r][col li]; a lane feeds four MFMAs from
// one 16-byte load (MFMA e takes the k's {4 q + e}: any bijection serves as long as A and B agree).
//   phase A  z3 (16 x 32)  = h2 (A layout, straight from Z2) x W3^T          64 MFMAs
//   phase B  cell: h3, p (DPP sum over the 16 lanes of a row), dp, dz3 - all in C layout
//   phase C  dh2 (16 x 128) = dz3 (A layout: through a 2 KB LDS tile) x W3      64 MFMAs;  dZ2 = dh2 * h2 (1 - h2), h2 in C layout from LDS
//   phase D  dW3 (32 x 128) += dz3^T x h2: contraction over the tile's rows - dz3's C layout IS the A operand, h2's C layout the B operand   64 MFMAs
// Deterministic: fixed tile -> wave assignment, waves added in wave order, one slab per workgroup, the library's ordered slab reduce.
#include "common.h"
#include "slab_reduce.h"

#define TF_N3 32
#define TF_K3 128
#define TF_SLAB 4164                        // dW3 4096 | db3 32 | dW4 32 | db4 | loss | 2 pad
#define TF_MAX_BLOCKS 256                   // one workgroup per CU (296 registers per lane): a second round of workgroups costs a whole prologue + reduce

__device__ __forceinline__ f32x4 tf_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// X3 (the fused step of precision mode 'bf16x3', gemm_bf16_big.hip "pair planes"): dZ2 leaves as a [hi | lo] pair of bf16 planes
// (row stride lddz, planes lddz / 2 apart) - the operand the layer's dgrad and weight gradient multiply - and the slab starts with
// the column sums of the fp32 dZ2, the 128-wide layer's bias gradient: [db2 128 | dW3 4096 | db3 32 | dW4 32 | db4 | loss | 2 pad].
#define TF_SLAB_X3 (TF_SLAB + TF_K3)
template <int X3>
__global__ __launch_bounds__(256) void f0_tail_rows_f32_kernel(const float* __restrict__ Z2, int ldz, const float* __restrict__ W3,
                                                               const float* __restrict__ b3, const float* __restrict__ W4,
                                                               const float* __restrict__ b4, const float* __restrict__ ybar,
                                                               const float* __restrict__ weight, const int64_t* __restrict__ seq_len,
                                                               int B, int T, int64_t M, float* __restrict__ pred,
                                                               void* __restrict__ dZ2v, int lddz, float* __restrict__ slab, int z_parts) {
    float* dZ2 = reinterpret_cast<float*>(dZ2v);
    uint16_t* dZ2p = reinterpret_cast<uint16_t*>(dZ2v);
    __shared__ float acc_c[X3 ? 4 : 1][TF_K3];
    float dbz[8];                               // X3: this lane's share of the column sums of dZ2 (column kt 16 + li)
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) dbz[kt] = 0.f;
    __shared__ __attribute__((aligned(16))) float th[4][16][TF_K3 + 4];       // per wave: the tile's h2 [row][k]
    __shared__ __attribute__((aligned(16))) float t3[4][16][TF_N3 + 4];       // per wave: the tile's dz3 [row][j]
    __shared__ float acc_w[TF_N3 * TF_K3];
    __shared__ float acc_s[4][68];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, q = lane >> 4;

    f32x4 w3a[2][8], w3b[8][2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) w3a[jt][kb] = *reinterpret_cast<const f32x4*>(W3 + (size_t)(jt * 16 + li) * TF_K3 + 16 * kb + 4 * q);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int e = 0; e < 4; ++e) w3b[kt][jb][e] = W3[(size_t)(16 * jb + 4 * q + e) * TF_K3 + kt * 16 + li];
    const float b3v[2] = {b3[li], b3[16 + li]}, w4v[2] = {W4[li], W4[16 + li]};
    const float b4v = b4[0];
    f32x4 dw3acc[2][8];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) dw3acc[jt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float db3p[2] = {0.f, 0.f}, dw4p[2] = {0.f, 0.f}, db4p = 0.f, lossp = 0.f;

    const int64_t tiles = (M + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t row0 = tile * 16;
        // ---- phase A: h2 in A layout straight from Z2, z3 = h2 W3^T -------------------------------------------------------------
        const int64_t ra = row0 + li;
        const bool va = ra < M;
        f32x4 h2a[8];
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (va) z = *reinterpret_cast<const f32x4*>(Z2 + (size_t)ra * ldz + 16 * kb + 4 * q);
            if (X3 && z_parts == 3 && va) {         // the pre-activations as three partial sums (mg_linear_fwd_x3_f32 parts == 3)
                const f32x4 z1 = *reinterpret_cast<const f32x4*>(Z2 + ((size_t)M + ra) * ldz + 16 * kb + 4 * q);
                const f32x4 z2 = *reinterpret_cast<const f32x4*>(Z2 + (2 * (size_t)M + ra) * ldz + 16 * kb + 4 * q);
                z = (z + z1) + z2;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) h2a[kb][e] = va ? mg_sigmoid(z[e]) : 0.f;
            *reinterpret_cast<f32x4*>(&th[wave][li][16 * kb + 4 * q]) = h2a[kb];
        }
        f32x4 z3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) z3[jt] = tf_mfma(h2a[kb][e], w3a[jt][kb][e], z3[jt]);
        // ---- phase B: the cell, C layout (row 4 q + r, unit jt 16 + li) ---------------------------------------------------------------
        const int64_t rc = row0 + 4 * q;
        float yb[4], wt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool v = rc + r < M;
            yb[r] = v ? ybar[rc + r] : 0.f;
            if (weight) {
                wt[r] = v ? weight[rc + r] : 0.f;
            } else {
                // rows are frames (b, t): the masked MSE's own weights [t < n_b] / (n_b B) (morgana/losses.py:29-51); an utterance
                // without frames gives NaN, as the reference's mean over no frames does
                wt[r] = 0.f;
                if (v) {
                    const int64_t bi = (rc + r) / T;
                    const int ti = (int)((rc + r) - bi * T);
                    const int64_t nb = seq_len ? (seq_len[bi] < T ? seq_len[bi] : (int64_t)T) : (int64_t)T;
                    wt[r] = nb == 0 ? (ti == 0 ? __builtin_nanf("") : 0.f) : (ti < nb ? 1.f / ((float)nb * (float)B) : 0.f);
                }
            }
        }
        f32x4 h3[2], dz3[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h3[jt][r] = mg_sigmoid(z3[jt][r] + b3v[jt]);
        float dp[4];
        f32x4 pv;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float p = mg_row16_sum(w4v[0] * h3[0][r] + w4v[1] * h3[1][r]) + b4v;
            pv[r] = p;
            const float w = wt[r];
            const float d = w > 0.f ? p - yb[r] : 0.f;
            dp[r] = 2.f * w * d;
            if (li == 0) {
                lossp += (w * d) * d;
                db4p += dp[r];
            }
        }
        if (li == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rc + r < M) pred[rc + r] = pv[r];
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = h3[jt][r];
                dz3[jt][r] = dp[r] * w4v[jt] * s * (1.f - s);
                db3p[jt] += dz3[jt][r];
                dw4p[jt] += dp[r] * s;
                t3[wave][4 * q + r][jt * 16 + li] = dz3[jt][r];
            }
        // the tiles in LDS belong to this wave alone: its own LDS operations complete in order
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- phase C: dh2 = dz3 W3, dZ2 = dh2 h2 (1 - h2);  phase D: dW3 += dz3^T h2 -----------------------------------------------------
        f32x4 dz3a[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) dz3a[jb] = *reinterpret_cast<const f32x4*>(&t3[wave][li][16 * jb + 4 * q]);
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            f32x4 dh = {0.f, 0.f, 0.f, 0.f}, hc;
#pragma unroll
            for (int r = 0; r < 4; ++r) hc[r] = th[wave][4 * q + r][kt * 16 + li];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                for (int e = 0; e < 4; ++e) dh = tf_mfma(dz3a[jb][e], w3b[kt][jb][e], dh);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rc + r < M) {
                    const float x = dh[r] * hc[r] * (1.f - hc[r]);
                    if (X3) {
                        // (lo from the ROUNDED product: left to the compiler the subtraction contracts with the product's last multiply)
#pragma clang fp contract(off)
                        const uint16_t hi = mg_f2bf(x);
                        uint16_t* dst = dZ2p + (size_t)(rc + r) * lddz + kt * 16 + li;
                        dst[0] = hi;
                        dst[lddz >> 1] = mg_f2bf(x - mg_bf2f(hi));
                        dbz[kt] += x;
                    } else {
                        dZ2[(size_t)(rc + r) * lddz + kt * 16 + li] = x;
                    }
                }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int e = 0; e < 4; ++e) dw3acc[jt][kt] = tf_mfma(dz3[jt][e], hc[e], dw3acc[jt][kt]);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the next tile overwrites th / t3
    }

    // ---- the workgroup's slab: waves added in wave order -----------------------------------------------------------------------------------
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int kt = 0; kt < 8; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (jt * 16 + 4 * q + r) * TF_K3 + kt * 16 + li;
                        acc_w[idx] = (wv == 0 ? 0.f : acc_w[idx]) + dw3acc[jt][kt][r];
                    }
        }
        __syncthreads();
    }
    // the small sums: over the four lane groups of a wave (xor 16, xor 32), then over the waves in wave order
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        float a = db3p[jt], b = dw4p[jt];
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (q == 0) {
            acc_s[wave][jt * 16 + li] = a;
            acc_s[wave][32 + jt * 16 + li] = b;
        }
    }
    {
        float a = db4p, b = lossp;                  // non-zero in the lanes li == 0 only
        a += __shfl_xor(a, 16, 64);
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 16, 64);
        b += __shfl_xor(b, 32, 64);
        if (lane == 0) {
            acc_s[wave][64] = a;
            acc_s[wave][65] = b;
        }
    }
    if (X3) {
        // the column sums of dZ2: over a wave's four row groups (xor 16, xor 32), then over the waves in wave order
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
            float a = dbz[kt];
            a += __shfl_xor(a, 16, 64);
            a += __shfl_xor(a, 32, 64);
            if (q == 0) acc_c[X3 ? wave : 0][kt * 16 + li] = a;
        }
    }
    __syncthreads();
    float* out = slab + (size_t)blockIdx.x * (X3 ? TF_SLAB_X3 : TF_SLAB);
    if (X3) {
        if (tid < TF_K3) out[tid] = ((acc_c[0][tid] + acc_c[X3 ? 1 : 0][tid]) + acc_c[X3 ? 2 : 0][tid]) + acc_c[X3 ? 3 : 0][tid];
        out += TF_K3;
    }
    for (int i = tid; i < TF_N3 * TF_K3; i += 256) out[i] = acc_w[i];
    if (tid < 68) out[TF_N3 * TF_K3 + tid] = tid < 66 ? ((acc_s[0][tid] + acc_s[1][tid]) + acc_s[2][tid]) + acc_s[3][tid] : 0.f;
}

extern "C" {

size_t mg_f0_tail_rows_f32_workspace_bytes(int64_t M) {
    if (M <= 0) return 256;
    int64_t blocks = mg_ceil_div(mg_ceil_div(M, 16), 4);
    if (blocks > TF_MAX_BLOCKS) blocks = TF_MAX_BLOCKS;
    return mg_align_up((size_t)blocks * TF_SLAB * sizeof(float), 256);
}

int mg_f0_tail_rows_f32(const float* Z2, int ldz, const float* W3, const float* b3, const float* W4, const float* b4, const float* ybar,
                        const float* weight, const int64_t* seq_len, int B, int T, int64_t M, float* pred, float* dZ2, int lddz,
                        float* grads_out, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(Z2 && W3 && b3 && W4 && b4 && ybar && pred && dZ2 && grads_out && M > 0,
                 "mg_f0_tail_rows_f32: bad arguments (M=%lld)", (long long)M);
    MG_CHECK_ARG(weight || (B > 0 && T > 0 && (int64_t)B * T == M),
                 "mg_f0_tail_rows_f32: without per-row weights the rows are the B x T frames (B=%d T=%d M=%lld)", B, T, (long long)M);
    MG_CHECK_ARG(ldz >= TF_K3 && ldz % 4 == 0 && lddz >= TF_K3 && ((uintptr_t)Z2 % 16) == 0 && ((uintptr_t)W3 % 16) == 0 &&
                     ((uintptr_t)grads_out % 16) == 0,
                 "mg_f0_tail_rows_f32: ldz=%d (multiple of 4, >= 128) lddz=%d (>= 128); Z2, W3 and grads_out 16-byte aligned", ldz, lddz);
    if (!workspace || workspace_bytes < mg_f0_tail_rows_f32_workspace_bytes(M) || ((uintptr_t)workspace % 16) != 0) {
        mg_set_error("mg_f0_tail_rows_f32: 16-byte aligned workspace of %zu bytes needed, got %zu", mg_f0_tail_rows_f32_workspace_bytes(M),
                     workspace_bytes);
        return MG_EWORKSPACE;
    }
    int64_t blocks = mg_ceil_div(mg_ceil_div(M, 16), 4);
    if (blocks > TF_MAX_BLOCKS) blocks = TF_MAX_BLOCKS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(f0_tail_rows_f32_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, Z2, ldz, W3, b3, W4, b4, ybar, weight, seq_len, B, T, M, pred,
                       (void*)dZ2, lddz, (float*)workspace, 1);
    MG_CHECK_LAUNCH("mg_f0_tail_rows_f32");
    mg_launch_slab_reduce((const float*)workspace, TF_SLAB, TF_SLAB, (int)blocks, grads_out, 0, st);
    MG_CHECK_LAUNCH("mg_f0_tail_rows_f32/reduce");
    return MG_OK;
}

size_t mg_f0_tail_rows_x3_workspace_bytes(int64_t M) {
    if (M <= 0) return 256;
    int64_t blocks = mg_ceil_div(mg_ceil_div(M, 16), 4);
    if (blocks > TF_MAX_BLOCKS) blocks = TF_MAX_BLOCKS;
    return mg_align_up((size_t)blocks * TF_SLAB_X3 * sizeof(float), 256);
}

// mg_f0_tail_rows_f32 for the fused 'bf16x3' step: dZ2 as a [hi | lo] pair of bf16 planes [M, lddz] (planes lddz / 2 apart), slabs of
// MG_F0_TAIL_X3_SLAB floats [db2 | dW3 | db3 | dW4 | db4 | loss | pad] left in `workspace` (*n_slabs of them, *stride floats apart) for
// the caller's ordered reduce (mg_slab_reduce_f32) or the update kernel's plan; grads_out != NULL: reduced here (MG_F0_TAIL_X3_SLAB floats).
int mg_f0_tail_rows_x3(const float* Z2, int ldz, int z_parts, const float* W3, const float* b3, const float* W4, const float* b4,
                       const float* ybar, const float* weight, int64_t M, float* pred, uint16_t* dZ2, int lddz, float* grads_out,
                       void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream) {
    MG_CHECK_ARG(Z2 && W3 && b3 && W4 && b4 && ybar && weight && pred && dZ2 && n_slabs && stride && M > 0 && (z_parts == 1 || z_parts == 3),
                 "mg_f0_tail_rows_x3: bad arguments (M=%lld z_parts=%d)", (long long)M, z_parts);
    MG_CHECK_ARG(ldz >= TF_K3 && ldz % 4 == 0 && lddz >= 2 * TF_K3 && lddz % 16 == 0 && ((uintptr_t)Z2 % 16) == 0 && ((uintptr_t)W3 % 16) == 0 &&
                     ((uintptr_t)dZ2 % 16) == 0 && (!grads_out || ((uintptr_t)grads_out % 16) == 0),
                 "mg_f0_tail_rows_x3: ldz=%d (multiple of 4, >= 128) lddz=%d (two planes of >= 128, multiple of 16); Z2, W3, dZ2, grads_out 16-byte aligned",
                 ldz, lddz);
    static_assert(TF_SLAB_X3 == MG_F0_TAIL_X3_SLAB, "header and kernel disagree");
    if (!workspace || workspace_bytes < mg_f0_tail_rows_x3_workspace_bytes(M) || ((uintptr_t)workspace % 16) != 0) {
        mg_set_error("mg_f0_tail_rows_x3: 16-byte aligned workspace of %zu bytes needed, got %zu", mg_f0_tail_rows_x3_workspace_bytes(M), workspace_bytes);
        return MG_EWORKSPACE;
    }
    int64_t blocks = mg_ceil_div(mg_ceil_div(M, 16), 4);
    if (blocks > TF_MAX_BLOCKS) blocks = TF_MAX_BLOCKS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(f0_tail_rows_f32_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, Z2, ldz, W3, b3, W4, b4, ybar, weight,
                       (const int64_t*)nullptr, 0, 0, M, pred, (void*)dZ2, lddz, (float*)workspace, z_parts);
    MG_CHECK_LAUNCH("mg_f0_tail_rows_x3");
    *n_slabs = (int)blocks;
    *stride = TF_SLAB_X3;
    if (grads_out) {
        mg_launch_slab_reduce((const float*)workspace, TF_SLAB_X3, TF_SLAB_X3, (int)blocks, grads_out, 0, st);
        MG_CHECK_LAUNCH("mg_f0_tail_rows_x3/reduce");
    }
    return MG_OK;
}

}  // extern "C"
