// Calibration only (never part of the product): how does the bf16 MFMA rate the chip sustains on random operands depend on the NUMBER
// of CUs that multiply?  Under a chip-level power limit (MI355X_MICROARCH.md, DVFS give-back) fewer multiplying CUs could hold a higher
// clock, so a launch that gives some CUs to latency-bound work (a fused tail beside a GEMM) would lose less matrix rate than the CUs it
// gives away.  Each workgroup declares 160 KB of LDS (one per CU) and runs a register-operand 16x16x32 loop, one or two waves per SIMD;
// the grid is n workgroups = n CUs.  Optionally the other CUs stream HBM meanwhile (a second kernel on a second stream).
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/cu_partition.hip -o /tmp/cu_partition && /tmp/cu_partition
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_mfma(const uint16_t* __restrict__ src, float* __restrict__ out, int trips) {
    extern __shared__ unsigned char dyn[];
    const int tid = blockIdx.x * THREADS + threadIdx.x;
    bfv8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const bfv8*>(src + ((size_t)(tid & 4095) * 8 + i) * 8);
        b[i] = *reinterpret_cast<const bfv8*>(src + ((size_t)(tid & 4095) * 8 + 4 + i) * 8);
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[4 * i + j], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += acc[i];
    if (trips < 0) dyn[threadIdx.x] = 1;                  // keeps the LDS declaration alive
    out[tid] = s[0] + s[1] + s[2] + s[3];
}

// A stream kernel for the CUs the MFMA grid leaves free: 160 KB of LDS per workgroup as well, float4 copy of its slice, `passes` times.
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ src, float4* __restrict__ dst, size_t n_per_wg, int passes) {
    extern __shared__ unsigned char dyn[];
    const float4* s = src + (size_t)blockIdx.x * n_per_wg;
    float4* d = dst + (size_t)blockIdx.x * n_per_wg;
    for (int p = 0; p < passes; ++p)
        for (size_t i = threadIdx.x; i < n_per_wg; i += 256) d[i] = s[i];
    if (passes < 0) dyn[threadIdx.x] = 1;
}

int main() {
    const int LDS = 160 * 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mfma<256>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mfma<512>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stream), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    std::vector<uint16_t> h(4096 * 64);
    srand(1);
    for (auto& v : h) {                                    // random bf16 in about [-1, 1)
        const float f = (float)rand() / RAND_MAX * 2.f - 1.f;
        uint32_t u;
        memcpy(&u, &f, 4);
        v = (uint16_t)(u >> 16);
    }
    uint16_t* src;
    float* out;
    CHECK(hipMalloc(&src, h.size() * 2));
    CHECK(hipMalloc(&out, 256 * 512 * 4));
    CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const size_t per_wg = (size_t)512 * 1024;               // float4 per streaming workgroup: 8 MB
    const size_t stream_bytes = per_wg * 16 * 192;
    float4 *sa, *sb;
    CHECK(hipMalloc(&sa, stream_bytes));
    CHECK(hipMalloc(&sb, stream_bytes));
    CHECK(hipMemset(sa, 1, stream_bytes));
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1));
    CHECK(hipStreamCreate(&s2));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));

    const int trips = 40000;                               // 16 MFMAs x 16,384 FLOP per trip and wave
    // warm the chip: ~1 s of full-grid work
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k_mfma<256>, dim3(256), dim3(256), LDS, s1, src, out, trips);
    CHECK(hipStreamSynchronize(s1));

    printf("%-8s %-6s %-8s %10s %12s %14s\n", "threads", "CUs", "stream", "ms", "PFLOP/s", "per-CU GF/s");
    for (int threads : {256, 512}) {
        for (int with_stream = 0; with_stream < 2; ++with_stream) {
            for (int n : {256, 240, 224, 208, 192, 160, 128, 64}) {
                if (with_stream && n == 256) continue;
                const int free_cus = 256 - n;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    if (with_stream) {
                        hipLaunchKernelGGL(k_stream, dim3(free_cus), dim3(256), LDS, s2, sa, sb, per_wg, 64);
                    }
                    CHECK(hipEventRecord(e0, s1));
                    for (int i = 0; i < 4; ++i) {
                        if (threads == 256)
                            hipLaunchKernelGGL(k_mfma<256>, dim3(n), dim3(256), LDS, s1, src, out, trips);
                        else
                            hipLaunchKernelGGL(k_mfma<512>, dim3(n), dim3(512), LDS, s1, src, out, trips / 2);
                    }
                    CHECK(hipEventRecord(e1, s1));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipStreamSynchronize(s2));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                const double waves = (double)n * (threads / 64);
                const double flop = 4.0 * waves * (threads == 256 ? trips : trips / 2) * 16.0 * 16384.0;
                printf("%-8d %-6d %-8s %10.3f %12.3f %14.1f\n", threads, n, with_stream ? "yes" : "no", best, flop / best / 1e12,
                       flop / best / 1e6 / n);
            }
        }
    }
    return 0;
}
