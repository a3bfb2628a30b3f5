// Active dropout on the HIP path: nn.Dropout(p) of the shipped models (models/RNN_SPSS.py:19,34,40, models/f0_test_model.py:22,31-43)
// in training mode.  y = x * keep / (1 - p) with keep ~ Bernoulli(1 - p) per element, as torch.nn.functional.dropout defines it
// (inverted dropout); identity in eval mode and for p == 0 (handled by the caller: no launch).
//
// The mask is never stored: it is a pure function of (seed, site, step counter, element index) through Philox4x32-10 (Salmon et al.,
// "Parallel random numbers: as easy as 1, 2, 3", SC'11 - the generator torch's own CUDA dropout uses; bit parity with torch's mask
// stream is not attainable and not attempted: torch's offsets depend on its launch geometry), so the backward pass regenerates it
// from the same four numbers.  counter words: (index / 4 low, index / 4 high, step counter low, site ^ step counter high); key =
// seed.  Four elements share one Philox block.  The step counter lives in device memory: a captured HIP graph then draws a new mask
// on every replay (mg_dropout_advance copies the value a call used into a per-call word for its backward, and increments).
#include "common.h"

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

struct u32q { unsigned x, y, z, w; };

__host__ __device__ static inline u32q philox4x32_10(u32q c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)PHILOX_M0 * c.x, p1 = (unsigned long long)PHILOX_M1 * c.z;
        const u32q n = {(unsigned)(p1 >> 32) ^ c.y ^ k0, (unsigned)p1, (unsigned)(p0 >> 32) ^ c.w ^ k1, (unsigned)p0};
        c = n;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    return c;
}

__device__ __forceinline__ unsigned drop_threshold(float p) {
    const double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
}

// one thread per 4 consecutive elements (one Philox block)
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, float p, float scale,
                                                      unsigned seed_lo, unsigned seed_hi, unsigned site,
                                                      const unsigned long long* __restrict__ counter) {
    const unsigned long long ctr = counter ? counter[0] : 0ull;
    const unsigned thr = drop_threshold(p);
    const int64_t blocks4 = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < blocks4; q += (int64_t)gridDim.x * 256) {
        const u32q r = philox4x32_10(u32q{(unsigned)q, (unsigned)((unsigned long long)q >> 32), (unsigned)ctr, site ^ (unsigned)(ctr >> 32)},
                                     seed_lo, seed_hi);
        const unsigned u[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t i = 4 * q + e;
            if (i < n) {
                float v;
                if constexpr (sizeof(T) == 2) v = mg_bf2f(x[i]);
                else v = x[i];
                v = u[e] >= thr ? v * scale : 0.f;
                if constexpr (sizeof(T) == 2) y[i] = mg_f2bf(v);
                else y[i] = v;
            }
        }
    }
}

__global__ void dropout_advance_kernel(unsigned long long* state, unsigned long long* used) {
    used[0] = state[0];
    state[0] += 1ull;
}

static int drop_grid(int64_t n) {
    int64_t blocks = mg_ceil_div(mg_ceil_div(n, 4), 256);
    if (blocks > 8192) blocks = 8192;
    return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" {

int mg_dropout(const void* x, void* y, int64_t n, int bf16, float p, uint64_t seed, uint32_t site, const uint64_t* counter, void* stream) {
    MG_CHECK_ARG(x && y && n >= 0 && p >= 0.f && p < 1.f, "mg_dropout: bad arguments (n=%lld p=%g; p must be in [0, 1))", (long long)n, (double)p);
    if (n == 0) return MG_OK;
    const float scale = 1.f / (1.f - p);
    const unsigned lo = (unsigned)seed, hi = (unsigned)(seed >> 32);
    if (bf16)
        hipLaunchKernelGGL(dropout_kernel<uint16_t>, dim3(drop_grid(n)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, n, p,
                           scale, lo, hi, site, (const unsigned long long*)counter);
    else
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(drop_grid(n)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, n, p, scale, lo,
                           hi, site, (const unsigned long long*)counter);
    MG_CHECK_LAUNCH("mg_dropout");
    return MG_OK;
}

int mg_dropout_advance(uint64_t* state, uint64_t* used, void* stream) {
    MG_CHECK_ARG(state && used, "mg_dropout_advance: null argument");
    hipLaunchKernelGGL(dropout_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)state, (unsigned long long*)used);
    MG_CHECK_LAUNCH("mg_dropout_advance");
    return MG_OK;
}

// Host restatement entry for the tests (no device involved): the four words of one Philox4x32-10 block.
void mg_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]) {
    const u32q r = philox4x32_10(u32q{counter[0], counter[1], counter[2], counter[3]}, key[0], key[1]);
    out[0] = r.x, out[1] = r.y, out[2] = r.z, out[3] = r.w;
}

}  // extern "C"
