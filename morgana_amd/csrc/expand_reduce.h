// The last two small jobs of the phone-rate step's forward - the per-phone prediction repeated to frames and the ordered sum of the
// fused tail's slabs (dW3 | db3 | dW4 | db4 | loss) - as device functions, so that they can run either as a launch of their own
// (expand_reduce_kernel, phone_rate.hip) or as RIDER blocks at the end of a later launch's grid (wgrad_dgrad_pair_kernel,
// gemm_bf16_big.hip).  One arithmetic, whoever runs it: the same bits.
#pragma once
#include "common.h"

struct ExpandReduceArgs {
    const float* table;       // one-column table (the prediction per phone row)
    const int32_t* rows;      // frame -> table row, >= 0
    int64_t M;                // frames
    float* out;               // [M]
    const float* partial;     // per-block partial sums of the loss's constant term (mg_phone_target_stats), n_partial of them
    int n_partial;
    const float* slab;        // S slabs, `stride` floats apart, n floats used of each
    int64_t n, stride;
    int S;
    float* dst;               // [n]; dst[n - 1] (the loss) also receives the constant term
    int64_t first_chunk;      // riders only: chunks of 16 elements below this one are left out (the update kernel sums the slabs itself)
};

// Elements [base, base + 16) of the ordered slab sum, by the 256 threads with tid < 256 (`active`; a larger block's other threads
// must call too: the barriers are the block's).  mg_slab_reduce_kernel's arithmetic: 16 interleaved partitions, each ascending, then
// ascending over the partitions.  part: float[16][17], red: float[256] (LDS).  base is block-uniform.
__device__ __forceinline__ void mg_tail_chunk_reduce(const ExpandReduceArgs& a, int64_t base, int tid, bool active, float (*part)[17], float* red) {
    if (base >= a.n) return;                           // block-uniform
    const int e = tid & 15, p = (tid >> 4) & 15;
    const int64_t i = base + e;
    float v = 0.f;
    if (active && i < a.n) {
        // the partition's slabs in ascending order, their loads eight at a time (168 slabs at C2: 10-11 per partition - two round
        // trips instead of three)
        int s = p;
        for (; s + 112 < a.S; s += 128) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = a.slab[(size_t)(s + 16 * q) * a.stride + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) v += t[q];
        }
#pragma unroll 4
        for (; s < a.S; s += 16) v += a.slab[(size_t)s * a.stride + i];
    }
    if (active) part[p][e] = v;
    const bool owns_loss = base <= a.n - 1 && a.n - 1 < base + 16;        // block-uniform
    float c = 0.f;
    if (owns_loss && active)
        for (int k = tid; k < a.n_partial; k += 256) c += a.partial[k];
    if (active) red[tid] = c;
    __syncthreads();
    if (owns_loss) {
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
    }
    if (active && p == 0 && i < a.n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][e];
        if (i == a.n - 1) t += red[0];
        a.dst[i] = t;
    }
}

// Rider block `rid` of `riders` (NT threads each, NT >= 256): frames rid NT + tid, + riders NT, ... four at a time (index loads, then
// table loads, then stores: three round trips per batch), then the chunks rid, rid + riders, ... of the slab sum.
template <int NT>
__device__ __forceinline__ void mg_expand_reduce_rider(const ExpandReduceArgs& a, int rid, int riders, unsigned char* smem) {
    float (*part)[17] = reinterpret_cast<float (*)[17]>(smem);
    float* red = reinterpret_cast<float*>(smem + 16 * 17 * sizeof(float));
    const int tid = threadIdx.x;
    const int64_t span = (int64_t)riders * NT;
    for (int64_t f0 = (int64_t)rid * NT + tid; f0 < a.M; f0 += 4 * span) {
        int r[4];
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (f0 + j * span < a.M) ? a.rows[f0 + j * span] : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = a.table[r[j]];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (f0 + j * span < a.M) a.out[f0 + j * span] = v[j];
    }
    const int64_t chunks = (a.n + 15) / 16;
    for (int64_t c = a.first_chunk + rid; c < chunks; c += riders) {
        mg_tail_chunk_reduce(a, c * 16, tid, tid < 256, part, red);
        __syncthreads();                               // part / red are reused by the next chunk
    }
}
