// Fused tail of the README F0Model (README.rst:65-73 layers 3-4 + morgana/losses.py:29-51), bf16 throughput mode:
//
//     H3 = sigmoid(H2 W3^T + b3)    pred = H3 w4 + b4    L = masked MSE(pred, target, seq_len)
//     dpred = dL/dpred   dZ3 = (dpred w4) * H3 (1 - H3)   dZ2 = (dZ3 W3) * H2 (1 - H2)
//     dW3 = dZ3^T H2, db3, dW4 = dpred^T H3, db4
//
// in ONE pass over H2 (M x 128 bf16).  The unfused path ran ~25 launches for this (two thin GEMMs, the loss, its
// gradient, casts, two wgrads + reduces, two dgrads), each at the ~5 us launch floor: ~200 us of a 1.05 ms step for 5 %
// of the arithmetic.  Here every wave owns 32-row tiles; the three matrix products run on v_mfma_f32_32x32x16_bf16:
//   Z3^T  = W3 . H2^T          weights as the A operand, so the lane holds one frame and its registers hold hidden units:
//                               bias, sigmoid, the 32-wide dot with w4 and all of the loss stay in registers
//   dH2^T = W3^T . dZ3^T       the dZ3^T accumulator registers ARE the B operand (no lane movement); W3^T fragments are
//                               pre-permuted to the accumulator's row order (cdna_hip_programming.md section 3)
//   dW3  += dZ3^T . H2         contraction over frames: both tiles are read back from LDS with ds_read_b64_tr_b16
// Partial sums are reduced lane -> wave -> workgroup in a fixed order and finished by an ordered slab reduce
// (deterministic, no atomics).
#include "common.h"
#include "slab_reduce.h"

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfv4 __attribute__((ext_vector_type(4)));
typedef __bf16 bfv2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define TAIL_K 128
#define TAIL_N 32
#define TAIL_SLAB (TAIL_N * TAIL_K + TAIL_N + TAIL_N + 2)   // dW3 | db3 | dW4 | db4 | loss
#define W3B_PITCH 272                                         // bytes per W3 row in LDS (256 + 16: conflict-free b128 rows)

// LDS map (bytes): [0, 8704) W3 rows bf16 (pitch 272) | [8704, 16896) permuted W3^T fragments | per wave: 8 KB H2 tile
// + 2 KB dZ3 tile.  After the loop the per-wave area is reused for the workgroup reduction.
#define TAIL_W3B 0
#define TAIL_W3P (TAIL_N * W3B_PITCH)
#define TAIL_WAVE0 (TAIL_W3P + 4 * 2 * 64 * 16)
#define TAIL_WAVE_BYTES (32 * 256 + 32 * 64)
#define TAIL_LDS (TAIL_WAVE0 + 4 * TAIL_WAVE_BYTES)

__device__ __forceinline__ bfv8 tail_tr_frag(const unsigned char* tile, int pitch, int col0, int lane, int ks, bool swz) {
    const int i = lane & 15, g = lane >> 4;
    const int q = i >> 2, p = i & 3;
    const int mrow = ks * 16 + 8 * (g >> 1) + q;
    const int col = col0 + 16 * (g & 1) + 4 * p;
    const int cpos = swz ? ((col >> 3) ^ ((mrow & 3) << 2)) : (col >> 3);
    const unsigned char* addr = tile + mrow * pitch + ((cpos << 4) | ((col & 7) << 1));
    const bfv4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(addr));
    const bfv4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bfv4*)(addr + 4 * pitch));
    return bfv8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256, 2) void f0_tail_kernel(const uint16_t* __restrict__ H2, int ldh, const float* __restrict__ W3,
                                                         const float* __restrict__ b3, const float* __restrict__ W4,
                                                         const float* __restrict__ b4, const float* __restrict__ target,
                                                         const int64_t* __restrict__ seq_len, int64_t M, int B, int T,
                                                         float grad_scale, float* __restrict__ pred, uint16_t* __restrict__ dZ2,
                                                         float* __restrict__ slab, const float* __restrict__ row_weight) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[TAIL_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mi = lane & 31, lh = lane >> 5;

    // ---- one-time: W3 as bf16 rows, and the permuted W3^T fragments of the dH2 product --------------------------------
    for (int e = tid; e < TAIL_N * TAIL_K; e += 256) {
        const int n = e / TAIL_K, k = e % TAIL_K;
        *reinterpret_cast<uint16_t*>(smem + TAIL_W3B + n * W3B_PITCH + k * 2) = mg_f2bf(W3[e]);
    }
    for (int e = tid; e < 4 * 2 * 64 * 8; e += 256) {        // [kt][s][lane][j]
        const int j = e & 7, l = (e >> 3) & 63, s = (e >> 9) & 1, kt = e >> 10;
        const int r = l & 31, h = l >> 5;
        const int n = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
        *reinterpret_cast<uint16_t*>(smem + TAIL_W3P + e * 2) = mg_f2bf(W3[n * TAIL_K + 32 * kt + r]);
    }
    __syncthreads();

    unsigned char* Ht = smem + TAIL_WAVE0 + wave * TAIL_WAVE_BYTES;   // [32 m][256 B], chunk c at c ^ ((m & 3) << 2)
    unsigned char* Dz = Ht + 32 * 256;                               // [32 m][64 B]

    // register r of a C^T tile <-> hidden unit / column offset 8 (r >> 2) + 4 lh + (r & 3)
    float b3v[16], w4v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = 8 * (r >> 2) + 4 * lh + (r & 3);
        b3v[r] = b3[n];
        w4v[r] = W4[n];
    }
    const float b4v = b4[0];

    f32x16 acc_w3[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_w3[kt][r] = 0.f;
    float dw4p[16], db3p[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dw4p[r] = db3p[r] = 0.f;
    float db4p = 0.f, lossp = 0.f;

    const int64_t n_tiles = (M + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t m = tile * 32 + mi;
        const bool live = m < M;

        // (1) this lane's frame row as B fragments: k = 16 ks + 8 lh + (0..7)
        u32x4 hB[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            hB[ks] = live ? *reinterpret_cast<const u32x4*>(H2 + (size_t)m * ldh + 16 * ks + 8 * lh) : u32x4{0u, 0u, 0u, 0u};

        // (2) Z3^T = W3 . H2^T
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bfv8 a = *reinterpret_cast<const bfv8*>(smem + TAIL_W3B + mi * W3B_PITCH + (2 * ks + lh) * 16);
            z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bfv8, hB[ks]), z, 0, 0, 0);
        }
        // (3) keep the tile in LDS for the transposed reads and the H2 (1 - H2) factors
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            *reinterpret_cast<u32x4*>(Ht + mi * 256 + (((2 * ks + lh) ^ ((mi & 3) << 2)) << 4)) = hB[ks];

        // (4) sigmoid, prediction
        float h3[16];
        float ph = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            h3[r] = mg_sigmoid_fast(z[r] + b3v[r]);
            ph += h3[r] * w4v[r];
        }
        const float p = ph + __shfl_xor(ph, 32, 64) + b4v;

        // (5) masked MSE of this frame
        float dpred = 0.f;
        if (live && row_weight) {
            // phone-rate rows (mg_f0_tail_rows_bf16): the row stands for the frames of one phone; w = sum of their loss weights,
            // target = their weighted mean (mg_phone_target_stats), so w (p - target)^2 has the gradient of the frames' total
            const float e = p - target[m];
            const float w = row_weight[m];
            dpred = (e * w) * (2.f * grad_scale);
            if (lh == 0) {
                pred[m] = p;
                lossp += e * e * w;
                db4p += dpred;
            }
        } else if (live) {
            const int b = (int)(m / T);
            const int t = (int)(m - (int64_t)b * T);
            int64_t nb = seq_len ? seq_len[b] : (int64_t)T;
            if (nb > T) nb = T;
            if (nb < 0) nb = 0;
            const float maskf = (int64_t)t < nb ? 1.f : 0.f;
            const float e = p - target[m];
            const float inv = 1.f / ((float)nb * (float)B);          // nb == 0 -> inf: 0 * inf = NaN, as the reference
            dpred = (e * maskf) * (2.f * grad_scale * inv);
            if (lh == 0) {
                pred[m] = p;
                lossp += (e * e * maskf) * inv;
                db4p += dpred;
            }
        }

        // (6) backward of layer 4 and of the layer-3 sigmoid
        float dz3[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dz3[r] = dpred * w4v[r] * h3[r] * (1.f - h3[r]);
            dw4p[r] += dpred * h3[r];
            db3p[r] += dz3[r];
        }
        // (7) dZ3 as bf16: registers 8 s .. 8 s + 7 are the B fragment of k-step s; the row-major copy goes to LDS
        u32x4 dzf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            unsigned int w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[q] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)dz3[8 * s + 2 * q], (__bf16)dz3[8 * s + 2 * q + 1]});
            dzf[s] = u32x4{w[0], w[1], w[2], w[3]};
            // registers 8s..8s+3 -> columns 16 s + 4 lh + (0..3); registers 8s+4..8s+7 -> columns 16 s + 8 + 4 lh + (0..3)
            *reinterpret_cast<u32x2*>(Dz + mi * 64 + (16 * s + 4 * lh) * 2) = u32x2{w[0], w[1]};
            *reinterpret_cast<u32x2*>(Dz + mi * 64 + (16 * s + 8 + 4 * lh) * 2) = u32x2{w[2], w[3]};
        }

        // (8) dH2^T = W3^T . dZ3^T, then dZ2 = dH2 * H2 (1 - H2), written as 16-byte row pieces
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bfv8 a = *reinterpret_cast<const bfv8*>(smem + TAIL_W3P + ((kt * 2 + s) * 64 + lane) * 16);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bfv8, dzf[s]), d, 0, 0, 0);
            }
            unsigned int pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = 4 * kt + g;                                  // 16-byte chunk of columns 32 kt + 8 g ..
                const bfv4 hv = *reinterpret_cast<const bfv4*>(Ht + mi * 256 + ((c ^ ((mi & 3) << 2)) << 4) + 8 * lh);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = (float)hv[e];
                    v[e] = d[4 * g + e] * h * (1.f - h);
                }
                pk[g][0] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[0], (__bf16)v[1]});
                pk[g][1] = __builtin_bit_cast(unsigned int, bfv2{(__bf16)v[2], (__bf16)v[3]});
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto r0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                if (live) *reinterpret_cast<u32x4*>(dZ2 + (size_t)m * ldh + 32 * kt + 8 * g + 8 * lh) = u32x4{r0[0], r1[0], r0[1], r1[1]};
            }
        }

        // (9) dW3 += dZ3^T . H2 (contraction over the 32 frames of the tile)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bfv8 a = tail_tr_frag(Dz, 64, 0, lane, s, false);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const bfv8 bq = tail_tr_frag(Ht, 256, 32 * kt, lane, s, true);
                acc_w3[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc_w3[kt], 0, 0, 0);
            }
        }
    }

    // ---- reduction: lanes (frames) -> wave -> workgroup, fixed order -----------------------------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            dw4p[r] += __shfl_xor(dw4p[r], off, 64);
            db3p[r] += __shfl_xor(db3p[r], off, 64);
        }
    }
    db4p = mg_wave_sum(db4p);
    lossp = mg_wave_sum(lossp);

    __syncthreads();                                   // every wave is done with its tiles: reuse the per-wave area
    float* red = reinterpret_cast<float*>(smem + TAIL_WAVE0);      // [TAIL_SLAB] floats (16.6 KB <= 40 KB)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = (r & 3) + 8 * (r >> 2) + 4 * lh;         // plain C layout: row = hidden unit n
                    const int k = 32 * kt + mi;                            //                 col = input feature k
                    float* dst = red + n * TAIL_K + k;
                    *dst = (w == 0 ? 0.f : *dst) + acc_w3[kt][r];
                }
            if (mi == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 8 * (r >> 2) + 4 * lh + (r & 3);
                    float* d3 = red + TAIL_N * TAIL_K + n;
                    float* d4 = red + TAIL_N * TAIL_K + TAIL_N + n;
                    *d3 = (w == 0 ? 0.f : *d3) + db3p[r];
                    *d4 = (w == 0 ? 0.f : *d4) + dw4p[r];
                }
            }
            if (lane == 0) {
                float* s4 = red + TAIL_N * TAIL_K + 2 * TAIL_N;
                s4[0] = (w == 0 ? 0.f : s4[0]) + db4p;
                s4[1] = (w == 0 ? 0.f : s4[1]) + lossp;
            }
        }
        __syncthreads();
    }
    float* out = slab + (size_t)blockIdx.x * TAIL_SLAB;
    for (int e = tid; e < TAIL_SLAB; e += 256) out[e] = red[e];
}

static int tail_blocks(int64_t M) {
    int64_t tiles = mg_ceil_div(M, 32);
    int64_t blocks = mg_ceil_div(tiles, 4);
    if (blocks > 512) blocks = 512;
    return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" {

size_t mg_f0_tail_workspace_bytes(int64_t M) { return mg_align_up((size_t)tail_blocks(M) * TAIL_SLAB * sizeof(float), 256); }

static int f0_tail_launch(const char* name, const uint16_t* H2, int ldh, int K3, const float* W3, const float* b3, const float* W4,
                          const float* b4, const float* target, const int64_t* seq_len, const float* row_weight, int64_t M, int B, int T,
                          float grad_scale, float* pred, float* loss, uint16_t* dZ2, float* grads, int accumulate, void* workspace,
                          size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(H2 && W3 && b3 && W4 && b4 && target && pred && loss && dZ2 && grads && M > 0, "%s: bad arguments (M=%lld)", name, (long long)M);
    MG_CHECK_ARG(K3 == TAIL_K && ldh >= TAIL_K && ldh % 8 == 0, "%s: needs a 128-wide hidden layer (K3=%d ldh=%d)", name, K3, ldh);
    MG_CHECK_ARG((((uintptr_t)H2 | (uintptr_t)dZ2) % 16) == 0, "%s: H2 / dZ2 must be 16-byte aligned", name);
    if (!workspace || workspace_bytes < mg_f0_tail_workspace_bytes(M)) {
        mg_set_error("%s: workspace of %zu bytes needed, got %zu", name, mg_f0_tail_workspace_bytes(M), workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int blocks = tail_blocks(M);
    float* slab = (float*)workspace;
    hipLaunchKernelGGL(f0_tail_kernel, dim3(blocks), dim3(256), 0, st, H2, ldh, W3, b3, W4, b4, target, seq_len, M, B, T, grad_scale,
                       pred, dZ2, slab, row_weight);
    MG_CHECK_LAUNCH(name);
    // grads = [dW3 (32*128) | db3 (32) | dW4 (32) | db4 (1)]; the loss is the last slab entry
    const int n_grads = TAIL_SLAB - 1;
    if (loss == grads + n_grads && !accumulate) {
        mg_launch_slab_reduce(slab, TAIL_SLAB, TAIL_SLAB, blocks, grads, 0, st);      // loss stored right behind the gradients
    } else {
        mg_launch_slab_reduce(slab, n_grads, TAIL_SLAB, blocks, grads, accumulate, st);
        mg_launch_slab_reduce(slab + (TAIL_SLAB - 1), 1, TAIL_SLAB, blocks, loss, 0, st);
    }
    MG_CHECK_LAUNCH(name);
    return MG_OK;
}

int mg_f0_tail_bf16(const uint16_t* H2, int ldh, int K3, const float* W3, const float* b3, const float* W4, const float* b4,
                    const float* target, const int64_t* seq_len, int B, int T, float grad_scale, float* pred, float* loss,
                    uint16_t* dZ2, float* grads, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(B > 0 && T > 0, "mg_f0_tail_bf16: bad arguments (B=%d T=%d)", B, T);
    return f0_tail_launch("mg_f0_tail_bf16", H2, ldh, K3, W3, b3, W4, b4, target, seq_len, nullptr, (int64_t)B * T, B, T, grad_scale, pred,
                          loss, dZ2, grads, accumulate, workspace, workspace_bytes, stream);
}

// The same tail on M rows that each stand for a GROUP of frames with one shared input row (the phone-rate step, csrc/phone_rate.hip):
// loss = sum_m row_weight[m] (pred[m] - target[m])^2 and its backward; target / row_weight f32 [M] from mg_phone_target_stats.
int mg_f0_tail_rows_bf16(const uint16_t* H2, int ldh, int K3, const float* W3, const float* b3, const float* W4, const float* b4,
                         const float* target, const float* row_weight, int64_t M, float grad_scale, float* pred, float* loss,
                         uint16_t* dZ2, float* grads, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(row_weight, "mg_f0_tail_rows_bf16: row_weight is null");
    return f0_tail_launch("mg_f0_tail_rows_bf16", H2, ldh, K3, W3, b3, W4, b4, target, nullptr, row_weight, M, 1, (int)(M < 2147483647LL ? M : 1),
                          grad_scale, pred, loss, dZ2, grads, accumulate, workspace, workspace_bytes, stream);
}

}  // extern "C"
