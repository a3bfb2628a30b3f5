// Calibration only (never part of the product): what bf16 MFMA rate does THIS MI355X sustain when the matrix pipe is the only thing
// that works - register operands, no LDS, no memory traffic inside the loop?  The 2.5 PFLOP/s roof of bench.py's fractions is the
// guide's dense peak at 2.4 GHz; under load the chip lowers its clock (MI355X_MICROARCH.md, DVFS give-back), more so with real
// (random) operands than with zeros.  This prints the rate per instruction shape, operand content, waves per SIMD and launch length,
// so that a kernel's "fraction of peak" can also be read as a fraction of what the silicon holds.
//   hipcc --offload-arch=gfx950 -O3 scripts/probe/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef __bf16 bfv8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

// 16 independent 16x16 accumulators, 4 + 4 operand fragments: 16 MFMAs (16x16x32: 16,384 FLOP each) per trip
__global__ __launch_bounds__(256) void k16(const uint16_t* __restrict__ src, float* __restrict__ out, int trips) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
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
    out[tid] = s[0] + s[1] + s[2] + s[3];
}

// 4 independent 32x32 accumulators, 2 + 2 fragments: 4 MFMAs (32x32x16: 32,768 FLOP each) per trip
__global__ __launch_bounds__(256) void k32(const uint16_t* __restrict__ src, float* __restrict__ out, int trips) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    bfv8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const bfv8*>(src + ((size_t)(tid & 4095) * 8 + i) * 8);
        b[i] = *reinterpret_cast<const bfv8*>(src + ((size_t)(tid & 4095) * 8 + 4 + i) * 8);
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[2 * i + j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
}

static uint16_t bf16_of(float x) {
    uint32_t u;
    memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int main() {
    const int n_src = 4096 * 8 * 8;
    std::vector<uint16_t> host(n_src);
    uint16_t *zeros, *randoms;
    float* out;
    CHECK(hipMalloc(&zeros, n_src * 2));
    CHECK(hipMalloc(&randoms, n_src * 2));
    CHECK(hipMalloc(&out, 1024 * 256 * 4));
    CHECK(hipMemset(zeros, 0, n_src * 2));
    srand(1);
    for (int i = 0; i < n_src; ++i) {                       // roughly normal values of unit scale: what a sigmoid layer's operands look like
        float s = 0.f;
        for (int j = 0; j < 12; ++j) s += (float)rand() / (float)RAND_MAX;
        host[i] = bf16_of((s - 6.f) * 0.5f);
    }
    CHECK(hipMemcpy(randoms, host.data(), n_src * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("bf16 MFMA rate with register operands, 256 CUs (dense peak of the guide: 2,500 TFLOP/s at 2.4 GHz)\n");
    printf("%-12s %-8s %-10s %-10s %10s %12s\n", "shape", "operands", "waves/SIMD", "launch ms", "TFLOP/s", "of 2.5 PF/s");
    for (int shape = 0; shape < 2; ++shape)
        for (int content = 0; content < 2; ++content)
            for (int wps = 1; wps <= 2; ++wps)
                for (int len = 0; len < 3; ++len) {
                    const int trips = len == 0 ? 2000 : len == 1 ? 12000 : 120000;     // about 0.1 / 0.6 / 6 ms
                    const int blocks = 256 * wps;
                    const uint16_t* src = content ? randoms : zeros;
                    float best = 1e30f;
                    for (int rep = 0; rep < (len == 2 ? 3 : 8); ++rep) {
                        CHECK(hipEventRecord(e0, 0));
                        if (shape == 0)
                            hipLaunchKernelGGL(k16, dim3(blocks), dim3(256), 0, 0, src, out, trips);
                        else
                            hipLaunchKernelGGL(k32, dim3(blocks), dim3(256), 0, 0, src, out, trips * 2);
                        CHECK(hipEventRecord(e1, 0));
                        CHECK(hipEventSynchronize(e1));
                        float ms;
                        CHECK(hipEventElapsedTime(&ms, e0, e1));
                        if (rep > 0 && ms < best) best = ms;
                    }
                    // per trip and wave: k16 16 x 16,384 FLOP; k32 (2 x trips) x 4 x 32,768 FLOP - the same 262,144 FLOP per unit of `trips`
                    const double flop = (double)blocks * 4.0 * trips * 262144.0;
                    const double tf = flop / (best * 1e-3) / 1e12;
                    printf("%-12s %-8s %-10d %-10.3f %10.1f %11.1f%%\n", shape == 0 ? "16x16x32" : "32x32x16", content ? "random" : "zeros", wps, best, tf,
                           100.0 * tf / 2500.0);
                    fflush(stdout);
                }
    return 0;
}
