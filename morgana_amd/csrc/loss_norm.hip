// K4 - seq_len-masked MSE forward+backward (reference: morgana/losses.py:29-51, mask from morgana/utils.py:115-144)
// K5 - mvn / minmax normalisers (reference: morgana/data.py:533-538, 579-590)
// sequence_mask (reference: morgana/utils.py:115-144)
//
// All HBM-bound elementwise / reduction kernels.  The reference runs ~8 eager kernels for the loss (mse_loss, mask
// build + H2D arange, mul, 2x sum, div, mean) plus their autograd mirrors; here one pass reads pred+target once and
// writes the gradient, a second tiny pass finishes the per-utterance normalisation in a fixed order (deterministic).
// Algorithmic bytes for K4: B*T*D*(4+4) read + B*T*D*4 written.
#include "common.h"

#define MSE_CHUNK 8192  // elements of one utterance handled by one workgroup (256 threads x 32)

__global__ __launch_bounds__(256) void sequence_mask_kernel(const int64_t* __restrict__ seq_len, int B, int max_len,
                                                            void* __restrict__ mask, int elem_size, int as_float) {
    const int64_t n = (int64_t)B * max_len;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / max_len);
        const int t = (int)(i - (int64_t)b * max_len);
        const bool on = (int64_t)t < seq_len[b];
        if (elem_size == 1) ((uint8_t*)mask)[i] = on ? 1 : 0;
        else if (elem_size == 4) {
            if (as_float) ((float*)mask)[i] = on ? 1.f : 0.f; else ((int32_t*)mask)[i] = on ? 1 : 0;
        } else {
            if (as_float) ((double*)mask)[i] = on ? 1.0 : 0.0; else ((int64_t*)mask)[i] = on ? 1 : 0;
        }
    }
}

// Stage 1: grid (chunks, B).  partial[b*chunks + chunk] = sum over the chunk of m*(p-y)^2; grad written if non-null.
template <bool VEC4>
__global__ __launch_bounds__(256) void masked_mse_stage1(const float* __restrict__ pred, const float* __restrict__ target,
                                                         const int64_t* __restrict__ seq_len, int T, int D, float grad_scale,
                                                         int B, float* __restrict__ grad, float* __restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int chunk = blockIdx.x;
    const int64_t row_elems = (int64_t)T * D;
    int64_t n_b = seq_len ? seq_len[b] : (int64_t)T;
    if (n_b > T) n_b = T;
    if (n_b < 0) n_b = 0;
    const int64_t valid_elems = n_b * D;
    // 2 * grad_scale / (n_b * B * D); n_b == 0 gives inf so that 0 * inf = NaN on every element of that utterance.
    const float coef = (2.0f * grad_scale) / ((float)n_b * (float)((int64_t)B * D));
    const size_t base = (size_t)b * row_elems;
    const int64_t lo = (int64_t)chunk * MSE_CHUNK;
    const int64_t hi = min(lo + (int64_t)MSE_CHUNK, row_elems);
    float acc = 0.f;
    if (VEC4) {
        for (int64_t e = lo + (int64_t)threadIdx.x * 4; e < hi; e += 1024) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(pred + base + e);
            const f32x4 y = *reinterpret_cast<const f32x4*>(target + base + e);
            f32x4 g;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = p[j] - y[j];
                const float m = (e + j) < valid_elems ? 1.f : 0.f;
                acc += d * d * m;
                g[j] = (d * m) * coef;
            }
            if (grad) *reinterpret_cast<f32x4*>(grad + base + e) = g;
        }
    } else {
        for (int64_t e = lo + threadIdx.x; e < hi; e += 256) {
            const float d = pred[base + e] - target[base + e];
            const float m = e < valid_elems ? 1.f : 0.f;
            acc += d * d * m;
            if (grad) grad[base + e] = (d * m) * coef;
        }
    }
    acc = mg_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Stage 2: one workgroup.  loss = (1/(B*D)) * sum_b ( sum_chunks partial[b,:] / n_b ), fixed summation order.
__global__ __launch_bounds__(256) void masked_mse_stage2(const float* __restrict__ partial, const int64_t* __restrict__ seq_len,
                                                         int B, int T, int D, int chunks, float* __restrict__ loss) {
    __shared__ float red[256];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        float s = 0.f;
        for (int c = 0; c < chunks; ++c) s += partial[(size_t)b * chunks + c];
        int64_t n_b = seq_len ? seq_len[b] : (int64_t)T;
        if (n_b > T) n_b = T;
        if (n_b < 0) n_b = 0;
        acc += s / (float)n_b;   // 0 / 0 = NaN when an utterance has no valid frame (reference behaviour)
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = red[0] / (float)((int64_t)B * D);
}

__global__ __launch_bounds__(256) void normalise_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                        const float* __restrict__ p0, const float* __restrict__ p1,
                                                        int64_t n, int D, int kind) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int d = (int)(i % D);
    const int step = (int)(stride % D);
    for (; i < n; i += stride) {
        const float a = p0[d], b = p1[d];
        const float v = x[i];
        float r;
        if (kind == MG_NORM_MVN) {
            r = (v - a) / (b + 1e-8f);
        } else if (kind == MG_DENORM_MVN) {
            r = v * b + a;
        } else {
            float scale = b - a;
            if (fabsf(scale) <= 1e-8f) scale = 1.f;
            r = kind == MG_NORM_MINMAX ? (v - a) / scale : v * scale + a;
        }
        out[i] = r;
        d += step;
        if (d >= D) d -= D;
    }
}

extern "C" {

int mg_sequence_mask(const int64_t* seq_len, int B, int max_len, void* mask, int elem_size, int as_float, void* stream) {
    MG_CHECK_ARG(seq_len && mask && B > 0 && max_len >= 0, "mg_sequence_mask: bad arguments (B=%d max_len=%d)", B, max_len);
    MG_CHECK_ARG(elem_size == 1 || elem_size == 4 || elem_size == 8, "mg_sequence_mask: elem_size %d not in {1,4,8}", elem_size);
    if (max_len == 0) return MG_OK;
    int64_t blocks = mg_ceil_div((int64_t)B * max_len, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(sequence_mask_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, seq_len, B, max_len, mask, elem_size, as_float);
    MG_CHECK_LAUNCH("mg_sequence_mask");
    return MG_OK;
}

size_t mg_masked_mse_workspace_bytes(int B, int T, int D) {
    const int64_t chunks = mg_ceil_div((int64_t)T * D, MSE_CHUNK);
    return mg_align_up((size_t)B * (size_t)(chunks < 1 ? 1 : chunks) * sizeof(float), 256);
}

int mg_masked_mse_f32(const float* pred, const float* target, const int64_t* seq_len, int B, int T, int D,
                      float grad_scale, float* loss, float* grad, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(pred && target && loss && B > 0 && T > 0 && D > 0, "mg_masked_mse_f32: bad arguments (B=%d T=%d D=%d)", B, T, D);
    MG_CHECK_ARG(B <= 65535, "mg_masked_mse_f32: B=%d exceeds 65535", B);
    if (!workspace || workspace_bytes < mg_masked_mse_workspace_bytes(B, T, D)) {
        mg_set_error("mg_masked_mse_f32: workspace of %zu bytes needed, got %zu", mg_masked_mse_workspace_bytes(B, T, D), workspace_bytes);
        return MG_EWORKSPACE;
    }
    const int chunks = (int)mg_ceil_div((int64_t)T * D, MSE_CHUNK);
    const bool vec = (((int64_t)T * D) % 4 == 0) &&
                     ((((uintptr_t)pred | (uintptr_t)target | (uintptr_t)(grad ? grad : pred)) % 16) == 0);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    if (vec)
        hipLaunchKernelGGL(masked_mse_stage1<true>, dim3(chunks, B), dim3(256), 0, st, pred, target, seq_len, T, D, grad_scale, B, grad, partial);
    else
        hipLaunchKernelGGL(masked_mse_stage1<false>, dim3(chunks, B), dim3(256), 0, st, pred, target, seq_len, T, D, grad_scale, B, grad, partial);
    MG_CHECK_LAUNCH("mg_masked_mse_f32/stage1");
    hipLaunchKernelGGL(masked_mse_stage2, dim3(1), dim3(256), 0, st, partial, seq_len, B, T, D, chunks, loss);
    MG_CHECK_LAUNCH("mg_masked_mse_f32/stage2");
    return MG_OK;
}

int mg_normalise_f32(const float* x, float* out, const float* p0, const float* p1, int64_t n_rows, int D, int kind, void* stream) {
    MG_CHECK_ARG(x && out && p0 && p1 && n_rows >= 0 && D > 0, "mg_normalise_f32: bad arguments (rows=%lld D=%d)", (long long)n_rows, D);
    MG_CHECK_ARG(kind >= MG_NORM_MVN && kind <= MG_DENORM_MINMAX, "mg_normalise_f32: unknown kind %d", kind);
    const int64_t n = n_rows * D;
    if (n == 0) return MG_OK;
    int64_t blocks = mg_ceil_div(n, 256 * 4);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(normalise_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, out, p0, p1, n, D, kind);
    MG_CHECK_LAUNCH("mg_normalise_f32");
    return MG_OK;
}

}  // extern "C"
