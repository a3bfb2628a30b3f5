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
// KIND 0: squared error (losses.py:49-51).  KIND 1: binary cross entropy with torch's log clamp at -100 (losses.py:54-56).
template <int KIND>
__device__ __forceinline__ void seq_loss_elem(float p, float y, float& l, float& g) {
    if (KIND == 0) {
        const float d = p - y;
        l = d * d;
        g = 2.f * d;
    } else {
        const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
        l = -(y * lp + (1.f - y) * lq);
        // torch's binary_cross_entropy backward: (p - y) / max(p (1 - p), 1e-12), not the derivative of the clamped logs
        g = (p - y) / fmaxf((1.f - p) * p, 1e-12f);
    }
}

template <bool VEC4, int KIND>
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
    // grad_scale / (n_b * B * D); n_b == 0 gives inf so that 0 * inf = NaN on every element of that utterance.
    const float coef = grad_scale / ((float)n_b * (float)((int64_t)B * D));
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
                float l, dl;
                seq_loss_elem<KIND>(p[j], y[j], l, dl);
                const float m = (e + j) < valid_elems ? 1.f : 0.f;
                acc += l * m;
                g[j] = (dl * m) * coef;
            }
            if (grad) *reinterpret_cast<f32x4*>(grad + base + e) = g;
        }
    } else {
        for (int64_t e = lo + threadIdx.x; e < hi; e += 256) {
            float l, dl;
            seq_loss_elem<KIND>(pred[base + e], target[base + e], l, dl);
            const float m = e < valid_elems ? 1.f : 0.f;
            acc += l * m;
            if (grad) grad[base + e] = (dl * m) * coef;
        }
    }
    acc = mg_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}

// Stage 2: one workgroup.  loss = (1/(B*D)) * sum_b ( sum_chunks partial[b,:] / n_b ), fixed summation order.
__global__ __launch_bounds__(256) void masked_mse_stage2(const float* __restrict__ partial, const int64_t* __restrict__ seq_len,
                                                         int B, int T, float final_div, int chunks, float* __restrict__ loss) {
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
    if (threadIdx.x == 0) loss[0] = red[0] / final_div;
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


// Multi-stream loss of the LSTM acoustic model (reference: models/RNN_SPSS.py:120-139): the prediction's columns are
// split into streams (lf0 deltas, vuv, mcep deltas, bap deltas), each stream has its own target tensor and is scored
// with losses.mse or - after torch.sigmoid - losses.bce; the stream losses are averaged.  One pass over pred.
struct stream_args {
    mg_stream_desc s[MG_STREAMS_MAX];
    int n;
};

__global__ __launch_bounds__(256) void stream_loss_stage1(const float* __restrict__ pred, stream_args sa,
                                                          const int64_t* __restrict__ seq_len, int T, int D, float grad_scale,
                                                          int B, float* __restrict__ grad, float* __restrict__ prob,
                                                          float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char col_stream[];  // D entries: stream of a column or 0xff
    __shared__ float red[4];
    __shared__ float col_w[MG_STREAMS_MAX];
    for (int c = threadIdx.x; c < D; c += 256) {
        unsigned char which = 0xff;
        for (int k = 0; k < sa.n; ++k)
            if (c >= sa.s[k].col0 && c < sa.s[k].col0 + sa.s[k].width) which = (unsigned char)k;
        col_stream[c] = which;
    }
    // weight of one element of stream k in the total: 1 / (B * width_k * n_streams)   (mean over (b,d), mean over streams)
    if (threadIdx.x < MG_STREAMS_MAX)
        col_w[threadIdx.x] = threadIdx.x < sa.n ? 1.f / ((float)B * (float)sa.s[threadIdx.x].width * (float)sa.n) : 0.f;
    __syncthreads();
    const int b = blockIdx.y;
    const int64_t row_elems = (int64_t)T * D;
    int64_t n_b = seq_len ? seq_len[b] : (int64_t)T;
    if (n_b > T) n_b = T;
    if (n_b < 0) n_b = 0;
    const float inv_nb = grad_scale / (float)n_b;   // inf for an empty utterance: 0 * inf = NaN, as the reference's 0 / 0
    const size_t base = (size_t)b * row_elems;
    const int64_t lo = (int64_t)blockIdx.x * MSE_CHUNK;
    const int64_t hi = min(lo + (int64_t)MSE_CHUNK, row_elems);
    int t = (int)((lo + threadIdx.x) / D);
    int c = (int)((lo + threadIdx.x) - (int64_t)t * D);
    const int dt = 256 / D, dc = 256 % D;
    float acc = 0.f;
    for (int64_t e = lo + threadIdx.x; e < hi; e += 256) {
        const int k = col_stream[c];
        float g = 0.f;
        if (k != 0xff) {
            const mg_stream_desc& sd = sa.s[k];
            const size_t ti = ((size_t)b * T + t) * sd.ldt + (c - sd.col0);
            const float y = sd.target[ti];
            const float x = pred[base + e];
            float l, dl;
            if (sd.kind == MG_LOSS_MSE) {
                seq_loss_elem<0>(x, y, l, dl);
            } else {
                const float p = 1.f / (1.f + expf(-x));        // torch.sigmoid, models/RNN_SPSS.py:93
                seq_loss_elem<1>(p, y, l, dl);
                dl *= (1.f - p) * p;                           // sigmoid backward
                if (prob) prob[((size_t)b * T + t) * sd.width + (c - sd.col0)] = p;
            }
            const float m = t < n_b ? 1.f : 0.f;
            acc += l * m * col_w[k];
            g = (dl * m) * (inv_nb * col_w[k]);
        }
        if (grad) grad[base + e] = g;
        t += dt;
        c += dc;
        if (c >= D) { c -= D; ++t; }
    }
    acc = mg_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}


// Device-side collate (reference: FilesDataset.__getitem__ normalising on the host, data.py:119-127, then collate_fn's zero
// padding, :159-224, then ToDeviceWrapper, :648-663): the utterances of a batch arrive packed back to back ([sum of
// lengths, D], one H2D copy) with their start offsets; one pass writes the zero-padded raw feature and / or its normalised
// twin.  Pad frames are zero in both (the reference pads the already normalised feature with zeros).  grid (chunks, B).
__global__ __launch_bounds__(256) void pad_normalise_kernel(const float* __restrict__ packed, const int64_t* __restrict__ offsets,
                                                            int T, int D, const float* __restrict__ p0, const float* __restrict__ p1,
                                                            int kind, float* __restrict__ raw_out, float* __restrict__ norm_out) {
    const int b = blockIdx.y;
    const int64_t lo = offsets[b];
    int64_t len = offsets[b + 1] - lo;
    if (len > T) len = T;
    const int64_t row_elems = (int64_t)T * D, valid = len * D;
    const float* src = packed + lo * D;
    const size_t base = (size_t)b * row_elems;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < row_elems; e += (int64_t)gridDim.x * 256) {
        float v = 0.f, r = 0.f;
        if (e < valid) {
            v = src[e];
            if (norm_out) {
                const int d = (int)(e % D);
                const float a = p0[d], c = p1[d];
                if (kind == MG_NORM_MVN) {
                    r = (v - a) / (c + 1e-8f);
                } else {
                    float scale = c - a;
                    if (fabsf(scale) <= 1e-8f) scale = 1.f;
                    r = (v - a) / scale;
                }
            }
        }
        if (raw_out) raw_out[base + e] = v;
        if (norm_out) norm_out[base + e] = r;
    }
}

// The same pass with the loader-side half of bf16 mode: next to the fp32 outputs it writes the bf16 OPERAND TABLE of the normalised
// feature - [B*T + extra_rows, ldb] bf16, row b*T + t = the normalised frame (zero for pad frames), columns D .. ldb-1 zero (the
// large-tile kernels' K padding), extra_rows zero rows behind (what pad frames gather in the phone-rate step).  One read of the
// packed feature; the stand-alone cast (mg_cast_pad_bf16) re-read the 49 MB fp32 feature and wrote the table in every training
// step of a loop whose loader did not provide it.  grid (chunks, B + 1): block row B zeroes the extra rows.
__global__ __launch_bounds__(256) void pad_normalise_bf16_kernel(const float* __restrict__ packed, const int64_t* __restrict__ offsets,
                                                                 int B, int T, int D, const float* __restrict__ p0,
                                                                 const float* __restrict__ p1, int kind, float* __restrict__ raw_out,
                                                                 float* __restrict__ norm_out, uint16_t* __restrict__ bf_out, int ldb,
                                                                 int extra_rows) {
    const int b = blockIdx.y;
    if (b == B) {                                        // the zero rows behind the table
        uint16_t* dst = bf_out + (size_t)B * T * ldb;
        const int64_t n = (int64_t)extra_rows * ldb;
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) dst[e] = 0;
        return;
    }
    const int64_t lo = offsets[b];
    int64_t len = offsets[b + 1] - lo;
    if (len > T) len = T;
    const float* src = packed + lo * D;
    const size_t base = (size_t)b * T * D, bf_base = (size_t)b * T * ldb;
    const int64_t n = (int64_t)T * ldb;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const int64_t t = e / ldb;
        const int d = (int)(e - t * ldb);
        float v = 0.f, r = 0.f;
        if (d < D && t < len) {
            v = src[t * D + d];
            const float a = p0[d], c = p1[d];
            if (kind == MG_NORM_MVN) {
                r = (v - a) / (c + 1e-8f);
            } else {
                float scale = c - a;
                if (fabsf(scale) <= 1e-8f) scale = 1.f;
                r = (v - a) / scale;
            }
        }
        if (d < D) {
            if (raw_out) raw_out[base + t * D + d] = v;
            if (norm_out) norm_out[base + t * D + d] = r;
        }
        bf_out[bf_base + e] = mg_f2bf(r);
    }
}

static size_t masked_ws_bytes(int B, int T, int D) {
    const int64_t chunks = mg_ceil_div((int64_t)T * D, MSE_CHUNK);
    return mg_align_up((size_t)B * (size_t)(chunks < 1 ? 1 : chunks) * sizeof(float), 256);
}

static int masked_loss(int kind, const float* pred, const float* target, const int64_t* seq_len, int B, int T, int D,
                       float grad_scale, float* loss, float* grad, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(pred && target && loss && B > 0 && T > 0 && D > 0, "mg_masked_mse_f32: bad arguments (B=%d T=%d D=%d)", B, T, D);
    MG_CHECK_ARG(B <= 65535, "mg_masked_mse_f32: B=%d exceeds 65535", B);
    if (!workspace || workspace_bytes < masked_ws_bytes(B, T, D)) {
        mg_set_error("mg_masked_mse_f32: workspace of %zu bytes needed, got %zu", masked_ws_bytes(B, T, D), workspace_bytes);
        return MG_EWORKSPACE;
    }
    const int chunks = (int)mg_ceil_div((int64_t)T * D, MSE_CHUNK);
    const bool vec = (((int64_t)T * D) % 4 == 0) &&
                     ((((uintptr_t)pred | (uintptr_t)target | (uintptr_t)(grad ? grad : pred)) % 16) == 0);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
#define LAUNCH_S1(V_, K_) hipLaunchKernelGGL((masked_mse_stage1<V_, K_>), dim3(chunks, B), dim3(256), 0, st, pred, target, seq_len, T, D, grad_scale, B, grad, partial)
    if (kind == 0) {
        if (vec) LAUNCH_S1(true, 0); else LAUNCH_S1(false, 0);
    } else {
        if (vec) LAUNCH_S1(true, 1); else LAUNCH_S1(false, 1);
    }
#undef LAUNCH_S1
    MG_CHECK_LAUNCH("mg_masked_mse_f32/stage1");
    hipLaunchKernelGGL(masked_mse_stage2, dim3(1), dim3(256), 0, st, partial, seq_len, B, T, (float)((int64_t)B * D), chunks, loss);
    MG_CHECK_LAUNCH("mg_masked_mse_f32/stage2");
    return MG_OK;
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

size_t mg_masked_mse_workspace_bytes(int B, int T, int D) { return masked_ws_bytes(B, T, D); }

int mg_masked_mse_f32(const float* pred, const float* target, const int64_t* seq_len, int B, int T, int D,
                      float grad_scale, float* loss, float* grad, void* workspace, size_t workspace_bytes, void* stream) {
    return masked_loss(0, pred, target, seq_len, B, T, D, grad_scale, loss, grad, workspace, workspace_bytes, stream);
}

int mg_masked_bce_f32(const float* pred, const float* target, const int64_t* seq_len, int B, int T, int D,
                      float grad_scale, float* loss, float* grad, void* workspace, size_t workspace_bytes, void* stream) {
    return masked_loss(1, pred, target, seq_len, B, T, D, grad_scale, loss, grad, workspace, workspace_bytes, stream);
}

size_t mg_stream_loss_workspace_bytes(int B, int T, int D) { return masked_ws_bytes(B, T, D); }

int mg_stream_loss_f32(const float* pred, const mg_stream_desc* streams, int n_streams, const int64_t* seq_len, int B, int T, int D,
                       float grad_scale, float* loss, float* grad, float* prob, void* workspace, size_t workspace_bytes,
                       void* stream) {
    MG_CHECK_ARG(pred && streams && loss && B > 0 && T > 0 && D > 0, "mg_stream_loss_f32: bad arguments (B=%d T=%d D=%d)", B, T, D);
    MG_CHECK_ARG(n_streams >= 1 && n_streams <= MG_STREAMS_MAX, "mg_stream_loss_f32: n_streams=%d not in 1..%d", n_streams, MG_STREAMS_MAX);
    MG_CHECK_ARG(B <= 65535 && D <= 16384, "mg_stream_loss_f32: B=%d or D=%d too large", B, D);
    stream_args sa;
    sa.n = n_streams;
    int n_bce = 0;
    for (int k = 0; k < n_streams; ++k) {
        const mg_stream_desc& d = streams[k];
        MG_CHECK_ARG(d.target && d.width > 0 && d.col0 >= 0 && d.col0 + d.width <= D && d.ldt >= d.width,
                     "mg_stream_loss_f32: stream %d (col0=%d width=%d ldt=%d) does not fit D=%d", k, d.col0, d.width, d.ldt, D);
        MG_CHECK_ARG(d.kind == MG_LOSS_MSE || d.kind == MG_LOSS_SIGMOID_BCE, "mg_stream_loss_f32: stream %d has unknown kind %d", k, d.kind);
        for (int j = 0; j < k; ++j)
            MG_CHECK_ARG(d.col0 >= streams[j].col0 + streams[j].width || streams[j].col0 >= d.col0 + d.width,
                         "mg_stream_loss_f32: streams %d and %d overlap", j, k);
        n_bce += d.kind == MG_LOSS_SIGMOID_BCE;
        sa.s[k] = d;
    }
    MG_CHECK_ARG(!prob || n_bce == 1, "mg_stream_loss_f32: prob output needs exactly one sigmoid-BCE stream, got %d", n_bce);
    if (!workspace || workspace_bytes < masked_ws_bytes(B, T, D)) {
        mg_set_error("mg_stream_loss_f32: workspace of %zu bytes needed, got %zu", masked_ws_bytes(B, T, D), workspace_bytes);
        return MG_EWORKSPACE;
    }
    const int chunks = (int)mg_ceil_div((int64_t)T * D, MSE_CHUNK);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    hipLaunchKernelGGL(stream_loss_stage1, dim3(chunks, B), dim3(256), (size_t)mg_align_up((size_t)D, 16), st, pred, sa, seq_len, T, D,
                       grad_scale, B, grad, prob, partial);
    MG_CHECK_LAUNCH("mg_stream_loss_f32/stage1");
    hipLaunchKernelGGL(masked_mse_stage2, dim3(1), dim3(256), 0, st, partial, seq_len, B, T, 1.f, chunks, loss);
    MG_CHECK_LAUNCH("mg_stream_loss_f32/stage2");
    return MG_OK;
}

int mg_pad_normalise_f32(const float* packed, const int64_t* offsets, int B, int T, int D, const float* p0, const float* p1,
                         int kind, float* raw_out, float* norm_out, void* stream) {
    MG_CHECK_ARG(packed && offsets && B > 0 && T >= 0 && D > 0, "mg_pad_normalise_f32: bad arguments (B=%d T=%d D=%d)", B, T, D);
    MG_CHECK_ARG(raw_out || norm_out, "mg_pad_normalise_f32: no output requested");
    MG_CHECK_ARG(!norm_out || (p0 && p1 && (kind == MG_NORM_MVN || kind == MG_NORM_MINMAX)),
                 "mg_pad_normalise_f32: a normalised output needs parameters and kind MG_NORM_MVN or MG_NORM_MINMAX (kind=%d)", kind);
    MG_CHECK_ARG(B <= 65535, "mg_pad_normalise_f32: B=%d exceeds 65535", B);
    if (T == 0) return MG_OK;
    int64_t chunks = mg_ceil_div((int64_t)T * D, 256 * 8);
    if (chunks > 1024) chunks = 1024;
    hipLaunchKernelGGL(pad_normalise_kernel, dim3((unsigned)chunks, B), dim3(256), 0, (hipStream_t)stream, packed, offsets, T, D, p0, p1, kind,
                       raw_out, norm_out);
    MG_CHECK_LAUNCH("mg_pad_normalise_f32");
    return MG_OK;
}

int mg_pad_normalise_bf16_f32(const float* packed, const int64_t* offsets, int B, int T, int D, const float* p0, const float* p1, int kind,
                              float* raw_out, float* norm_out, uint16_t* table_bf16, int ldb, int extra_rows, void* stream) {
    MG_CHECK_ARG(packed && offsets && table_bf16 && B > 0 && T >= 0 && D > 0, "mg_pad_normalise_bf16_f32: bad arguments (B=%d T=%d D=%d)", B, T,
                 D);
    MG_CHECK_ARG(p0 && p1 && (kind == MG_NORM_MVN || kind == MG_NORM_MINMAX),
                 "mg_pad_normalise_bf16_f32: needs normaliser parameters and kind MG_NORM_MVN or MG_NORM_MINMAX (kind=%d)", kind);
    MG_CHECK_ARG(ldb >= D && extra_rows >= 0, "mg_pad_normalise_bf16_f32: ldb=%d must cover D=%d, extra_rows=%d must not be negative", ldb, D,
                 extra_rows);
    MG_CHECK_ARG(B < 65535, "mg_pad_normalise_bf16_f32: B=%d exceeds 65534", B);
    if (T == 0 && extra_rows == 0) return MG_OK;
    int64_t chunks = mg_ceil_div((int64_t)T * ldb, 256 * 8);
    if (chunks > 1024) chunks = 1024;
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(pad_normalise_bf16_kernel, dim3((unsigned)chunks, B + (extra_rows > 0 ? 1 : 0)), dim3(256), 0, (hipStream_t)stream,
                       packed, offsets, B, T, D, p0, p1, kind, raw_out, norm_out, table_bf16, ldb, extra_rows);
    MG_CHECK_LAUNCH("mg_pad_normalise_bf16_f32");
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
