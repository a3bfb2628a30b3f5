// Ordered reduction of split-M partial results ("slabs"): dst[i] (+)= sum_s slab[s * stride + i], i < n.
// 16 consecutive elements x 16 slab partitions per workgroup: partition p adds slabs p, p + 16, ... in ascending order,
// then the 16 partials are added in ascending p.  The order is fixed, so results are bitwise reproducible (no float
// atomics anywhere in the weight-gradient path).  Shared by the fp32, bf16, fused-backward and fused-tail kernels.
#pragma once

#include "common.h"

static __global__ __launch_bounds__(256) void mg_slab_reduce_kernel(const float* __restrict__ slab, int64_t n, int64_t stride,
                                                                    int S, float* __restrict__ dst, int accumulate) {
    __shared__ float part[16][17];
    const int e = threadIdx.x & 15, p = threadIdx.x >> 4;
    for (int64_t base = (int64_t)blockIdx.x * 16; base < n; base += (int64_t)gridDim.x * 16) {
        const int64_t i = base + e;
        float v = 0.f;
        if (i < n) {
#pragma unroll 4
            for (int s = p; s < S; s += 16) v += slab[(size_t)s * stride + i];
        }
        part[p][e] = v;
        __syncthreads();
        if (p == 0 && i < n) {
            float t = accumulate ? dst[i] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += part[q][e];
            dst[i] = t;
        }
        __syncthreads();
    }
}

// Same sums in the same order, four consecutive elements per thread (16-byte loads: a 16-lane partition reads 256 contiguous
// bytes of a slab instead of 64).  Needs n, stride multiples of 4 and 16-byte aligned buffers.
static __global__ __launch_bounds__(256) void mg_slab_reduce4_kernel(const float* __restrict__ slab, int64_t n, int64_t stride,
                                                                     int S, float* __restrict__ dst, int accumulate) {
    __shared__ f32x4 part[16][17];
    const int e = threadIdx.x & 15, p = threadIdx.x >> 4;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < n; base += (int64_t)gridDim.x * 64) {
        const int64_t i = base + 4 * e;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < n) {
#pragma unroll 4
            for (int s = p; s < S; s += 16) v += *reinterpret_cast<const f32x4*>(slab + (size_t)s * stride + i);
        }
        part[p][e] = v;
        __syncthreads();
        if (p == 0 && i < n) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
            if (accumulate) t = *reinterpret_cast<const f32x4*>(dst + i);
#pragma unroll
            for (int q = 0; q < 16; ++q) t += part[q][e];
            *reinterpret_cast<f32x4*>(dst + i) = t;
        }
        __syncthreads();
    }
}

static inline void mg_launch_slab_reduce(const float* slab, int64_t n, int64_t stride, int S, float* dst, int accumulate,
                                         hipStream_t st) {
#ifdef MG_EXPERIMENTS
    if (g_mg_tuning[MG_TUNE_SKIP_REDUCE]) return;     // lab builds only: time the producing kernel alone (results then invalid)
#endif
    if (n % 4 == 0 && stride % 4 == 0 && n >= 4096 && (((uintptr_t)slab | (uintptr_t)dst) % 16) == 0) {
        int64_t blocks4 = mg_ceil_div(n, 64);
        if (blocks4 > 32768) blocks4 = 32768;
        hipLaunchKernelGGL(mg_slab_reduce4_kernel, dim3((unsigned)blocks4), dim3(256), 0, st, slab, n, stride, S, dst, accumulate);
        return;
    }
    int64_t blocks = mg_ceil_div(n, 16);
    if (blocks > 32768) blocks = 32768;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mg_slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, slab, n, stride, S, dst, accumulate);
}
