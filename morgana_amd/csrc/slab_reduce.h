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

static inline void mg_launch_slab_reduce(const float* slab, int64_t n, int64_t stride, int S, float* dst, int accumulate,
                                         hipStream_t st) {
    int64_t blocks = mg_ceil_div(n, 16);
    if (blocks > 32768) blocks = 32768;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mg_slab_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, slab, n, stride, S, dst, accumulate);
}
