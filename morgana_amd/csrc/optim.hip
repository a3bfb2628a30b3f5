// Optimiser-side elementwise kernels: fused flat Adam (torch.optim.Adam at experiment_builder.py:516 / :474),
// EMA (morgana/utils.py:443-456), fp32<->bf16 casts for the bf16 GEMM operands, stand-alone sigmoid.
// All HBM-bound; one pass over flat buffers (the reference's foreach Adam is ~10 passes over 8 tensors).
#include "common.h"
#include "expand_reduce.h"

#include <math.h>

// One element of the Adam update, shared by every kernel below with the floating-point contraction PINNED: left to the compiler, which
// of the two products of  v * beta2 + (1 - beta2) * g * g  is fused into an fma depends on the surrounding code, and the three kernels
// would round differently (measured: mg_adam_step_plan_f32 against mg_adam_step_dev_f32 differed in the last bit).
struct mg_adam_out {
    float p, m, v;
};
__device__ __forceinline__ mg_adam_out mg_adam_update(float p, float g, float mi, float vi, float beta1, float beta2, float eps,
                                                       float weight_decay, float step_size, float bc2_sqrt, float grad_scale) {
#pragma clang fp contract(off)
    g = g * grad_scale;
    if (weight_decay != 0.f) g = __fmaf_rn(weight_decay, p, g);  // L2 penalty added to the gradient
    mi = __fmaf_rn(g - mi, 1.f - beta1, mi);                      // exp_avg.lerp_(grad, 1 - beta1)
    vi = __fmaf_rn((1.f - beta2) * g, g, vi * beta2);             // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p = p - step_size * (mi / denom);                             // param.addcdiv_(exp_avg, denom, value=-step_size)
    return mg_adam_out{p, mi, vi};
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float beta1,
                                                   float beta2, float eps, float weight_decay, float step_size,
                                                   float bc2_sqrt, float grad_scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const mg_adam_out o = mg_adam_update(param[i], grad[i], m[i], v[i], beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, grad_scale);
        param[i] = o.p;
        m[i] = o.m;
        v[i] = o.v;
    }
}

// The same update with (step_size, bc2_sqrt) read from device memory: a captured launch (hipGraph replay of the whole training step,
// morgana_amd/graphs.py) must not bake the step-dependent scalars into its arguments.
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                       float* __restrict__ m, float* __restrict__ v, int64_t n, float beta1,
                                                       float beta2, float eps, float weight_decay,
                                                       const float* __restrict__ scalars, float grad_scale) {
    const float step_size = scalars[0], bc2_sqrt = scalars[1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const mg_adam_out o = mg_adam_update(param[i], grad[i], m[i], v[i], beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, grad_scale);
        param[i] = o.p;
        m[i] = o.m;
        v[i] = o.v;
    }
}

// The update as the last node of a step (include/morgana_hip.h: mg_adam_step_plan_f32): split-M slabs of the weight-gradient GEMMs are
// summed here in the slab reduce's own order (slab_reduce.h: 16 interleaved partitions, each ascending, then added in ascending order
// onto the accumulator - so the result is bitwise what reduce launch + mg_adam_step_dev_f32 produce), the bf16 operands of the next
// step's GEMMs are refreshed from the updated weights, and the gradient is zeroed behind the read.
// Work split: a workgroup owns 64 consecutive elements; thread (p, e) = (tid >> 4, tid & 15) sums slabs p, p + 16, ... for elements
// 4e .. 4e+3 of them (16-byte loads where the range allows), the 16 partial sums of an element meet in LDS and one thread per element
// adds them in ascending p - exactly mg_slab_reduce4_kernel's arithmetic - and then runs the update for that element.
__global__ __launch_bounds__(256) void adam_plan_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, float beta1, float beta2, float eps,
                                                        float weight_decay, const float* __restrict__ scalars, float grad_scale,
                                                        mg_adam_plan plan, int wide_chunks) {
    __shared__ float part[16][68];
    // the forward's deferred tail (mg_adam_tail): the first blocks repeat the prediction / form the loss before their share of the update
    if (plan.tail.frames > 0 || plan.tail.n > 0) {
        const mg_adam_tail& t = plan.tail;
        const int64_t by_frames = (t.frames + 2047) / 2048;
        int riders = (int)(by_frames > 1 ? by_frames : 1);
        if (riders > (int)gridDim.x) riders = (int)gridDim.x;
        if ((int)blockIdx.x < riders) {
            const int64_t chunks = (t.n + 15) / 16;
            const ExpandReduceArgs xr{t.table, t.rows, t.frames, t.out, t.partial, t.n_partial, t.slab, t.n, t.stride, t.n_slabs, t.dst,
                                      t.n > 0 ? (t.n - 1) / 16 : chunks};
            mg_expand_reduce_rider<256>(xr, (int)blockIdx.x, riders, reinterpret_cast<unsigned char*>(part));
        }
    }
    const float step_size = scalars[0], bc2_sqrt = scalars[1];
    // one element: the update, the gradient zeroed behind the read, the bf16 operand copies refreshed
    auto update = [&](int64_t i, float g) {
        if (plan.clear_grad) grad[i] = 0.f;
        const mg_adam_out o = mg_adam_update(param[i], g, m[i], v[i], beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, grad_scale);
        const float w = o.p;
        param[i] = w;
        m[i] = o.m;
        v[i] = o.v;
        for (int k = 0; k < plan.n_shadows; ++k) {
            const mg_adam_shadow sh = plan.shadows[k];
            const int64_t j = i - sh.offset;
            if (j < 0 || j >= (int64_t)sh.rows * sh.cols) continue;
            const unsigned r = (unsigned)j / (unsigned)sh.cols, cc = (unsigned)j - r * (unsigned)sh.cols;     // rows * cols < 2^31 (checked)
            const uint16_t b = mg_f2bf(w);
            if (sh.dst) sh.dst[(size_t)r * sh.ldd + cc] = b;
            if (sh.dst_t) sh.dst_t[(size_t)cc * sh.ldt + r] = b;
            if (sh.pair) {                             // [hi | lo] pair planes (precision 'bf16x3'): the lo plane half a row further
                const uint16_t lo = mg_f2bf(w - mg_bf2f(b));
                if (sh.dst) sh.dst[(size_t)r * sh.ldd + (sh.ldd >> 1) + cc] = lo;
                if (sh.dst_t) sh.dst_t[(size_t)cc * sh.ldt + (sh.ldt >> 1) + r] = lo;
            }
        }
    };
    const int e = threadIdx.x & 15, p = threadIdx.x >> 4;
    // 64 consecutive elements with slab sources to sum (all threads: barriers inside)
    auto slab_chunk = [&](int64_t base) {
        const int64_t i0 = base + 4 * e;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        bool any = false;
        for (int k = 0; k < plan.n_slab_srcs; ++k) {
            const mg_adam_slab_src src = plan.slabs[k];
            const int64_t j0 = i0 - src.begin;
            if (j0 + 3 < 0 || j0 >= src.count) continue;
            any = true;
            if (j0 >= 0 && j0 + 3 < src.count && ((j0 | src.stride) & 3) == 0 && ((uintptr_t)src.slab & 15) == 0) {
                for (int s = p; s < src.n_slabs; s += 16) acc += *reinterpret_cast<const f32x4*>(src.slab + (size_t)s * src.stride + j0);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int64_t j = j0 + c;
                    if (j < 0 || j >= src.count) continue;
                    float t = 0.f;
                    for (int s = p; s < src.n_slabs; s += 16) t += src.slab[(size_t)s * src.stride + j];
                    acc[c] += t;
                }
            }
        }
        // wave-uniform enough: a workgroup either lies in slab ranges or does not (ranges are long); the barrier is taken by all
        const bool block_any = __syncthreads_or(any ? 1 : 0) != 0;
        if (block_any) {
            *reinterpret_cast<f32x4*>(&part[p][4 * e]) = acc;
            __syncthreads();
        }
        // The update itself: ONE element per thread of the first wave (it was four elements on each of 16 threads, with a 64-bit
        // division per element and operand copy: 27 us at the phone-rate step for 85 MB of slabs).  Same sums in the same order.
        if (threadIdx.x < 64) {
            const int64_t i = base + threadIdx.x;
            if (i < n) {
                float g = grad[i];
                if (block_any) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) g += part[q][threadIdx.x];
                }
                update(i, g);
            }
        }
        if (block_any) __syncthreads();
    };
    if (wide_chunks) {
        // Few or no slabs to sum (the recurrent models' steps: their large weight-gradient launches reduce into `grad` themselves): a
        // workgroup takes 256 consecutive elements - one per thread, no LDS and no barrier, where no slab source reaches into them
        // (it was one wave of the workgroup per 64 elements: 274 us for the LSTM model's 17.5 M parameters, now 142), as four 64-element
        // chunks of the summing form where one does.  The same arithmetic per element either way.
        const int64_t n_chunks = (n + 255) / 256;
        for (int64_t c = n_chunks - 1 - (int64_t)blockIdx.x; c >= 0; c -= (int64_t)gridDim.x) {
            const int64_t sbase = c * 256;
            bool hit = false;
            for (int k = 0; k < plan.n_slab_srcs; ++k)
                if (sbase < plan.slabs[k].begin + plan.slabs[k].count && sbase + 256 > plan.slabs[k].begin) hit = true;
            if (!hit) {
                const int64_t i = sbase + threadIdx.x;
                if (i < n) update(i, grad[i]);
                continue;
            }
            for (int sub = 3; sub >= 0; --sub)
                if (sbase + 64 * sub < n) slab_chunk(sbase + 64 * sub);
        }
        return;
    }
    // Blocks walk the flat buffer from its END: the last parameters are the fused tail's (their source has 168-256 slabs, five times
    // the loads of the others), and the blocks that are dispatched first should be the ones that take longest.
    const int64_t n_chunks = (n + 63) / 64;
    for (int64_t c = n_chunks - 1 - (int64_t)blockIdx.x; c >= 0; c -= (int64_t)gridDim.x) slab_chunk(c * 64);
}

__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ shadow, const float* __restrict__ param, int64_t n, float one_minus_decay) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float s = shadow[i];
        shadow[i] = s - one_minus_decay * (s - param[i]);
    }
}

__global__ __launch_bounds__(256) void cast_pad_bf16_kernel(const float* __restrict__ src, int lds, uint16_t* __restrict__ dst,
                                                            int ldd, int64_t rows, int cols) {
    const int64_t n = rows * ldd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ldd;
        const int c = (int)(i - r * ldd);
        dst[i] = c < cols ? mg_f2bf(src[r * lds + c]) : (uint16_t)0;
    }
}

// Eight columns per thread: two 16-byte loads, one 16-byte store (needs lds % 4 == 0, ldd % 8 == 0, 16-byte aligned buffers).
// The element-wise form above spends a 64-bit division per element and moves 2 bytes per lane: 2.9 TB/s on the 49 MB label
// table of config C2 against ~2x that here.
__global__ __launch_bounds__(256) void cast_pad_bf16_x8_kernel(const float* __restrict__ src, int lds, uint16_t* __restrict__ dst,
                                                               int ldd, int64_t rows, int cols) {
    const int chunks = ldd >> 3;
    const int64_t n = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int c0 = (int)(i - r * chunks) << 3;
        float v[8];
        const float* sp = src + r * lds + c0;
        if (c0 + 8 <= cols) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
            v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (c0 + j < cols) ? sp[j] : 0.f;
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)mg_f2bf(v[j]);
        *reinterpret_cast<bf16x8*>(dst + r * ldd + c0) = o;
    }
}

// dst[c, r] = src[r, c]; 32x32 LDS tile transpose.
__global__ __launch_bounds__(256) void cast_transpose_bf16_kernel(const float* __restrict__ src, int lds, uint16_t* __restrict__ dst,
                                                                  int ldd, int rows, int cols) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? src[(size_t)r * lds + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;   // output row c, output column r
        if (c < cols && r < ldd) dst[(size_t)c * ldd + r] = r < rows ? mg_f2bf(tile[tx][j]) : (uint16_t)0;
    }
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const uint16_t* __restrict__ src, int lds, float* __restrict__ dst,
                                                            int ldd, int64_t rows, int cols) {
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ldd + c] = mg_bf2f(src[r * lds + c]);
    }
}

__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) y[i] = mg_sigmoid(x[i]);
}

__global__ __launch_bounds__(256) void sigmoid_grad_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           float* __restrict__ dx, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float s = y[i];
        dx[i] = dy[i] * s * (1.f - s);
    }
}

// All weight matrices of a layer stack in ONE launch: descriptor d gives an fp32 [rows, cols] matrix and asks for its
// bf16 copy [rows, ldd] (zero padded) and / or its bf16 transpose [cols, ldt] (zero padded).  blockIdx.y = descriptor.
struct CastBatch {
    int count;
    mg_cast_desc d[MG_CAST_MAX];
};

__global__ __launch_bounds__(256) void cast_params_kernel(CastBatch batch) {
    __shared__ float tile[32][33];
    const mg_cast_desc d = batch.d[blockIdx.y];
    if (d.dst) {
        const int64_t n = (int64_t)d.rows * d.ldd;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const int r = (int)(i / d.ldd), c = (int)(i - (int64_t)r * d.ldd);
            d.dst[i] = c < d.cols ? mg_f2bf(d.src[(size_t)r * d.cols + c]) : (uint16_t)0;
        }
    }
    if (d.dst_t) {
        const int tiles_c = (d.cols + 31) / 32, tiles_r = (d.ldt + 31) / 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
            const int c0 = (t % tiles_c) * 32, r0 = (t / tiles_c) * 32;
            __syncthreads();
            for (int j = ty; j < 32; j += 8) {
                const int r = r0 + j, c = c0 + tx;
                tile[j][tx] = (r < d.rows && c < d.cols) ? d.src[(size_t)r * d.cols + c] : 0.f;
            }
            __syncthreads();
            for (int j = ty; j < 32; j += 8) {
                const int c = c0 + j, r = r0 + tx;
                if (c < d.cols && r < d.ldt) d.dst_t[(size_t)c * d.ldt + r] = r < d.rows ? mg_f2bf(tile[tx][j]) : (uint16_t)0;
            }
        }
    }
}

struct CopyBatch {
    mg_copy_desc d[MG_COPY_MAX];
};

// blockIdx.y = descriptor; 16 bytes per thread and trip where both ends are 16-byte aligned, bytes otherwise
__global__ __launch_bounds__(256) void copy_many_kernel(CopyBatch batch) {
    const mg_copy_desc d = batch.d[blockIdx.y];
    const unsigned char* src = reinterpret_cast<const unsigned char*>(d.src);
    unsigned char* dst = reinterpret_cast<unsigned char*>(d.dst);
    const int64_t stride = (int64_t)gridDim.x * 256;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const int64_t n16 = d.bytes >> 4;
        typedef unsigned int cm_u32x4 __attribute__((ext_vector_type(4)));
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
            reinterpret_cast<cm_u32x4*>(dst)[i] = reinterpret_cast<const cm_u32x4*>(src)[i];
        for (int64_t i = (n16 << 4) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.bytes; i += stride) dst[i] = src[i];
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < d.bytes; i += stride) dst[i] = src[i];
    }
}

static int flat_grid(int64_t n) {
    int64_t blocks = mg_ceil_div(n, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" {

int mg_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                     float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
    MG_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, "mg_adam_step_f32: bad arguments (n=%lld step=%lld)",
                 (long long)n, (long long)step);
    if (n == 0) return MG_OK;
    // Host-side scalars in double, as torch computes them in Python floats.
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(adam_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, grad_scale);
    MG_CHECK_LAUNCH("mg_adam_step_f32");
    return MG_OK;
}

__global__ void store_pair_kernel(float* __restrict__ dst, float a, float b) {
    dst[0] = a;
    dst[1] = b;
}

// dst[0..1] = (a, b) in stream order.  The values travel as kernel arguments, so the host may run any number of steps ahead of the
// device (an async copy from one reused pinned buffer would be read when the copy executes, not when it was issued).
int mg_store_pair_f32(float* dst, float a, float b, void* stream) {
    MG_CHECK_ARG(dst, "mg_store_pair_f32: null destination");
    hipLaunchKernelGGL(store_pair_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst, a, b);
    MG_CHECK_LAUNCH("mg_store_pair_f32");
    return MG_OK;
}

struct StorePairs {
    float v[2 * MG_STORE_PAIRS_MAX];
};
__global__ void store_pairs_kernel(float* __restrict__ dst, StorePairs values, int n) {
    if ((int)threadIdx.x < 2 * n) dst[threadIdx.x] = values.v[threadIdx.x];
}

// dst[0 .. 2 n) = the n pairs of the HOST array `values`, in stream order, carried as kernel arguments like mg_store_pair_f32: the
// scalars of the n steps one graph replay performs (morgana_amd/graphs.py, steps_per_replay), staged by ONE launch.
int mg_store_pairs_f32(float* dst, const float* values, int n_pairs, void* stream) {
    MG_CHECK_ARG(dst && values && n_pairs >= 1 && n_pairs <= MG_STORE_PAIRS_MAX, "mg_store_pairs_f32: bad arguments (n_pairs=%d, at most %d)", n_pairs,
                 MG_STORE_PAIRS_MAX);
    StorePairs sp;
    for (int i = 0; i < 2 * n_pairs; ++i) sp.v[i] = values[i];
    for (int i = 2 * n_pairs; i < 2 * MG_STORE_PAIRS_MAX; ++i) sp.v[i] = 0.f;
    hipLaunchKernelGGL(store_pairs_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, dst, sp, n_pairs);
    MG_CHECK_LAUNCH("mg_store_pairs_f32");
    return MG_OK;
}

// (step_size, bc2_sqrt) of step `step` exactly as mg_adam_step_f32 forms them (host doubles), for mg_adam_step_dev_f32
void mg_adam_scalars(float lr, float beta1, float beta2, int64_t step, float* out2) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    out2[0] = (float)((double)lr / bc1);
    out2[1] = (float)sqrt(bc2);
}

int mg_adam_step_dev_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2,
                         float eps, float weight_decay, const float* scalars, float grad_scale, void* stream) {
    MG_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && scalars && n >= 0, "mg_adam_step_dev_f32: bad arguments (n=%lld)", (long long)n);
    if (n == 0) return MG_OK;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       beta1, beta2, eps, weight_decay, scalars, grad_scale);
    MG_CHECK_LAUNCH("mg_adam_step_dev_f32");
    return MG_OK;
}

int mg_adam_step_plan_f32(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2, float eps,
                          float weight_decay, const float* scalars, float grad_scale, const mg_adam_plan* plan, void* stream) {
    MG_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && scalars && plan && n >= 0, "mg_adam_step_plan_f32: bad arguments (n=%lld)", (long long)n);
    MG_CHECK_ARG(plan->n_slab_srcs >= 0 && plan->n_slab_srcs <= MG_ADAM_MAX_SLABS && plan->n_shadows >= 0 && plan->n_shadows <= MG_ADAM_MAX_SHADOWS,
                 "mg_adam_step_plan_f32: %d slab sources / %d shadows exceed %d / %d", plan->n_slab_srcs, plan->n_shadows, MG_ADAM_MAX_SLABS,
                 MG_ADAM_MAX_SHADOWS);
    for (int k = 0; k < plan->n_slab_srcs; ++k) {
        const mg_adam_slab_src& src = plan->slabs[k];
        MG_CHECK_ARG(src.slab && src.begin >= 0 && src.count > 0 && src.begin + src.count <= n && src.n_slabs >= 1 && src.stride >= src.count,
                     "mg_adam_step_plan_f32: slab source %d (begin %lld count %lld of %lld, %d slabs, stride %lld)", k, (long long)src.begin,
                     (long long)src.count, (long long)n, src.n_slabs, (long long)src.stride);
    }
    for (int k = 0; k < plan->n_shadows; ++k) {
        const mg_adam_shadow& sh = plan->shadows[k];
        MG_CHECK_ARG(sh.offset >= 0 && sh.rows > 0 && sh.cols > 0 && (int64_t)sh.rows * sh.cols < 2147483647LL &&
                         sh.offset + (int64_t)sh.rows * sh.cols <= n && (sh.dst || sh.dst_t),
                     "mg_adam_step_plan_f32: shadow %d does not lie inside the flat buffer", k);
        MG_CHECK_ARG((!sh.dst || sh.ldd >= sh.cols) && (!sh.dst_t || sh.ldt >= sh.rows), "mg_adam_step_plan_f32: shadow %d: ldd %d / ldt %d too small", k,
                     sh.ldd, sh.ldt);
        MG_CHECK_ARG(!sh.pair || ((!sh.dst || (sh.ldd % 2 == 0 && sh.ldd / 2 >= sh.cols)) && (!sh.dst_t || (sh.ldt % 2 == 0 && sh.ldt / 2 >= sh.rows))),
                     "mg_adam_step_plan_f32: shadow %d: pair planes need even ldd %d / ldt %d with half of it covering the matrix", k, sh.ldd, sh.ldt);
    }
    {
        const mg_adam_tail& t = plan->tail;
        MG_CHECK_ARG(t.frames >= 0 && t.n >= 0 && (t.frames == 0 || (t.table && t.rows && t.out)) &&
                         (t.n == 0 || (t.slab && t.dst && t.stride >= t.n && t.n_slabs >= 1 && (t.n_partial == 0 || t.partial))),
                     "mg_adam_step_plan_f32: bad deferred tail (frames %lld, n %lld)", (long long)t.frames, (long long)t.n);
        MG_CHECK_ARG((t.frames == 0 && t.n == 0) || n > 0, "mg_adam_step_plan_f32: a deferred tail needs a launch (n = 0)");
    }
    if (n == 0) return MG_OK;
    // 256-element chunks when slab sources cover less than half of the buffer (adam_plan_kernel: wide_chunks)
    int64_t covered = 0;
    for (int k = 0; k < plan->n_slab_srcs; ++k) covered += plan->slabs[k].count;
    const int wide = 2 * covered < n ? 1 : 0;
    hipLaunchKernelGGL(adam_plan_kernel, dim3((unsigned)mg_ceil_div(n, wide ? 256 : 64)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                       exp_avg_sq, n, beta1, beta2, eps, weight_decay, scalars, grad_scale, *plan, wide);
    MG_CHECK_LAUNCH("mg_adam_step_plan_f32");
    return MG_OK;
}

int mg_ema_update_f32(float* shadow, const float* param, int64_t n, float decay, void* stream) {
    MG_CHECK_ARG(shadow && param && n >= 0, "mg_ema_update_f32: bad arguments");
    if (n == 0) return MG_OK;
    hipLaunchKernelGGL(ema_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, shadow, param, n, 1.0f - decay);
    MG_CHECK_LAUNCH("mg_ema_update_f32");
    return MG_OK;
}

int mg_cast_pad_bf16(const float* src, int lds, uint16_t* dst, int ldd, int64_t rows, int cols, void* stream) {
    MG_CHECK_ARG(src && dst && rows >= 0 && cols > 0 && lds >= cols && ldd >= cols, "mg_cast_pad_bf16: bad arguments (rows=%lld cols=%d lds=%d ldd=%d)",
                 (long long)rows, cols, lds, ldd);
    if (rows == 0) return MG_OK;
    if (lds % 4 == 0 && ldd % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) % 16) == 0)
        hipLaunchKernelGGL(cast_pad_bf16_x8_kernel, dim3(flat_grid(rows * (ldd / 8))), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, rows, cols);
    else
        hipLaunchKernelGGL(cast_pad_bf16_kernel, dim3(flat_grid(rows * ldd)), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, rows, cols);
    MG_CHECK_LAUNCH("mg_cast_pad_bf16");
    return MG_OK;
}

int mg_cast_transpose_bf16(const float* src, int lds, uint16_t* dst, int ldd, int rows, int cols, void* stream) {
    MG_CHECK_ARG(src && dst && rows > 0 && cols > 0 && lds >= cols && ldd >= rows, "mg_cast_transpose_bf16: bad arguments (rows=%d cols=%d lds=%d ldd=%d)",
                 rows, cols, lds, ldd);
    dim3 grid((unsigned)mg_ceil_div(cols, 32), (unsigned)mg_ceil_div(ldd, 32));
    hipLaunchKernelGGL(cast_transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, rows, cols);
    MG_CHECK_LAUNCH("mg_cast_transpose_bf16");
    return MG_OK;
}

int mg_cast_params_bf16(const mg_cast_desc* descs, int count, void* stream) {
    MG_CHECK_ARG(descs && count > 0 && count <= MG_CAST_MAX, "mg_cast_params_bf16: count %d not in 1..%d", count, MG_CAST_MAX);
    CastBatch batch;
    batch.count = count;
    for (int i = 0; i < count; ++i) {
        const mg_cast_desc& d = descs[i];
        MG_CHECK_ARG(d.src && d.rows > 0 && d.cols > 0 && (d.dst || d.dst_t), "mg_cast_params_bf16: bad descriptor %d", i);
        MG_CHECK_ARG(!d.dst || d.ldd >= d.cols, "mg_cast_params_bf16: descriptor %d: ldd %d < cols %d", i, d.ldd, d.cols);
        MG_CHECK_ARG(!d.dst_t || d.ldt >= d.rows, "mg_cast_params_bf16: descriptor %d: ldt %d < rows %d", i, d.ldt, d.rows);
        batch.d[i] = d;
    }
    hipLaunchKernelGGL(cast_params_kernel, dim3(256, count), dim3(256), 0, (hipStream_t)stream, batch);
    MG_CHECK_LAUNCH("mg_cast_params_bf16");
    return MG_OK;
}

int mg_copy_many(const mg_copy_desc* descs, int count, void* stream) {
    MG_CHECK_ARG(descs && count > 0 && count <= MG_COPY_MAX, "mg_copy_many: count %d not in 1..%d", count, MG_COPY_MAX);
    CopyBatch batch;
    int64_t most = 0;
    for (int i = 0; i < count; ++i) {
        MG_CHECK_ARG(descs[i].bytes >= 0 && (descs[i].bytes == 0 || (descs[i].src && descs[i].dst)), "mg_copy_many: bad descriptor %d", i);
        batch.d[i] = descs[i];
        if (descs[i].bytes > most) most = descs[i].bytes;
    }
    if (most == 0) return MG_OK;
    int64_t blocks = mg_ceil_div(most, 256 * 16 * 4);          // ~4 trips of 16 bytes per thread for the largest one
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)blocks, count), dim3(256), 0, (hipStream_t)stream, batch);
    MG_CHECK_LAUNCH("mg_copy_many");
    return MG_OK;
}

int mg_cast_bf16_f32(const uint16_t* src, int lds, float* dst, int ldd, int64_t rows, int cols, void* stream) {
    MG_CHECK_ARG(src && dst && rows >= 0 && cols > 0 && lds >= cols && ldd >= cols, "mg_cast_bf16_f32: bad arguments");
    if (rows == 0) return MG_OK;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(flat_grid(rows * cols)), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, rows, cols);
    MG_CHECK_LAUNCH("mg_cast_bf16_f32");
    return MG_OK;
}

int mg_sigmoid_f32(const float* x, float* y, int64_t n, void* stream) {
    MG_CHECK_ARG(x && y && n >= 0, "mg_sigmoid_f32: bad arguments");
    if (n == 0) return MG_OK;
    hipLaunchKernelGGL(sigmoid_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    MG_CHECK_LAUNCH("mg_sigmoid_f32");
    return MG_OK;
}

int mg_sigmoid_grad_f32(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    MG_CHECK_ARG(dy && y && dx && n >= 0, "mg_sigmoid_grad_f32: bad arguments");
    if (n == 0) return MG_OK;
    hipLaunchKernelGGL(sigmoid_grad_kernel, dim3(flat_grid(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    MG_CHECK_LAUNCH("mg_sigmoid_grad_f32");
    return MG_OK;
}

}  // extern "C"
