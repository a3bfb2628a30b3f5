// K7 - streaming metrics on the device (reference: morgana/metrics.py:359-695 - Mean, RMSE, MAE, Distortion, F0Distortion,
// LF0Distortion, MelCepDistortion as registered by models/RNN_SPSS.py:44-48 and accumulated inside the model's loss(),
// models/RNN_SPSS.py:120-129, i.e. once per training step).  The reference reduces with torch ops and pulls the frame count to
// the host with .item() in every call (metrics.py:394, :610); here one call = two small launches that ADD (sum, count) to a
// device accumulator of two doubles, and nothing is read back until result() is asked for.
// HBM bound: one pass over target / pred (8 B per element), deterministic (fixed-order two-stage reduction, no float atomics).
//
// Semantics kept from the reference, per kind (x = element value, m = frame mask t < seq_len[b], v = voiced flag):
//   MEAN           sum += sum x m             count += frames if seq_len else elements     (metrics.py:383-394: the masked count
//   SQDIFF         x = (t - p)^2                                                            is in FRAMES, the unmasked one in
//   ABSDIFF        x = |t - p|                                                              ELEMENTS - kept as it is)
//   ROOT_SQ        per frame x = sqrt(sum_d (t - p)^2): count in frames either way         (Distortion, :657-665)
//   SQDIFF_VOICED  x = (t - p)^2 v m          count += sum v m                              (F0Distortion, :597-609)
//   SQDIFF_VOICED_EXP  the same on exp(t), exp(p)                                           (LF0Distortion, :630-634)
// Columns [col0, col0 + width) of the D-wide rows take part (MelCepDistortion drops c0: col0 = 1, :690-694).
#include "common.h"

#define MK_MEAN 0
#define MK_SQDIFF 1
#define MK_ABSDIFF 2
#define MK_ROOT_SQ 3
#define MK_SQDIFF_VOICED 4
#define MK_SQDIFF_VOICED_EXP 5
#define MK_BLOCKS 1024

__global__ __launch_bounds__(256) void metric_stage1(int kind, const float* __restrict__ target, const float* __restrict__ pred,
                                                     const float* __restrict__ voiced, const int64_t* __restrict__ seq_len, int B, int T,
                                                     int D, int col0, int width, double* __restrict__ partial) {
    __shared__ double red_s[256], red_c[256];
    const int64_t frames = (int64_t)B * T;
    double s = 0.0, c = 0.0;
    for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < frames; f += (int64_t)gridDim.x * 256) {
        const int b = (int)(f / T), t = (int)(f - (int64_t)b * T);
        const bool live = seq_len ? ((int64_t)t < seq_len[b]) : true;
        if (!live) continue;
        const float* tp = target + f * D + col0;
        const float* pp = pred ? pred + f * D + col0 : nullptr;
        float v = 1.f;
        if (kind >= MK_SQDIFF_VOICED) {
            v = voiced[f];
            c += (double)v;
        } else if (seq_len || kind == MK_ROOT_SQ) {
            c += 1.0;
        } else {
            c += (double)width;
        }
        float acc = 0.f;
        for (int d = 0; d < width; ++d) {
            const float a = tp[d];
            if (kind == MK_MEAN) {
                acc += a;
            } else {
                float diff = a - pp[d];
                if (kind == MK_SQDIFF_VOICED_EXP) diff = expf(a) - expf(pp[d]);
                acc += kind == MK_ABSDIFF ? fabsf(diff) : diff * diff;
            }
        }
        if (kind == MK_ROOT_SQ) acc = sqrtf(acc);
        s += (double)(acc * v);
    }
    red_s[threadIdx.x] = s;
    red_c[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            red_s[threadIdx.x] += red_s[threadIdx.x + off];
            red_c[threadIdx.x] += red_c[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = red_s[0];
        partial[2 * blockIdx.x + 1] = red_c[0];
    }
}

__global__ __launch_bounds__(256) void metric_stage2(const double* __restrict__ partial, int n_blocks, double* __restrict__ accum) {
    __shared__ double red_s[256], red_c[256];
    double s = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < n_blocks; i += 256) {
        s += partial[2 * i];
        c += partial[2 * i + 1];
    }
    red_s[threadIdx.x] = s;
    red_c[threadIdx.x] = c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            red_s[threadIdx.x] += red_s[threadIdx.x + off];
            red_c[threadIdx.x] += red_c[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        accum[0] += red_s[0];
        accum[1] += red_c[0];
    }
}

extern "C" {

size_t mg_metric_workspace_bytes(void) { return (size_t)MK_BLOCKS * 2 * sizeof(double); }

int mg_metric_accumulate_f32(int kind, const float* target, const float* pred, const float* voiced, const int64_t* seq_len, int B, int T,
                             int D, int col0, int width, double* accum, void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(kind >= MK_MEAN && kind <= MK_SQDIFF_VOICED_EXP, "mg_metric_accumulate_f32: unknown kind %d", kind);
    MG_CHECK_ARG(target && accum && B > 0 && T > 0 && D > 0 && col0 >= 0 && width > 0 && col0 + width <= D,
                 "mg_metric_accumulate_f32: bad arguments (B=%d T=%d D=%d col0=%d width=%d)", B, T, D, col0, width);
    MG_CHECK_ARG(kind == MK_MEAN || pred, "mg_metric_accumulate_f32: kind %d needs predictions", kind);
    MG_CHECK_ARG(kind < MK_SQDIFF_VOICED || voiced, "mg_metric_accumulate_f32: kind %d needs the voiced flags", kind);
    if (!workspace || workspace_bytes < mg_metric_workspace_bytes()) {
        mg_set_error("mg_metric_accumulate_f32: workspace of %zu bytes needed, got %zu", mg_metric_workspace_bytes(), workspace_bytes);
        return MG_EWORKSPACE;
    }
    int64_t blocks = mg_ceil_div((int64_t)B * T, 256);
    if (blocks > MK_BLOCKS) blocks = MK_BLOCKS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(metric_stage1, dim3((unsigned)blocks), dim3(256), 0, st, kind, target, pred, voiced, seq_len, B, T, D, col0, width,
                       (double*)workspace);
    hipLaunchKernelGGL(metric_stage2, dim3(1), dim3(256), 0, st, (const double*)workspace, (int)blocks, accum);
    MG_CHECK_LAUNCH("mg_metric_accumulate_f32");
    return MG_OK;
}

}  // extern "C"
