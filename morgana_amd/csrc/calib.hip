// Calibration entry (a measurement, not part of any training step): the bf16 MFMA rate THIS chip sustains when its matrix pipe is the
// only thing that works - register operands (caller-provided, so random data is the caller's choice), no LDS, no memory traffic
// inside the loop, two waves per SIMD on every CU.  bench.py times it with HIP events and reports the frame-rate step's TFLOP/s as a
// fraction of this rate next to the fraction of the 2.5 PFLOP/s dense peak (MI355X_MICROARCH.md: under load the chip lowers its
// clock, more so on random operands than on zeros - "DVFS give-back"; profiles/r3_calib_mfma_rate.txt, r4_calib_cu_partition.txt).
#include "common.h"

typedef __bf16 cbfv8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void calib_mfma_kernel(const uint16_t* __restrict__ src, float* __restrict__ sink, int trips) {
    const int tid = blockIdx.x * 512 + threadIdx.x;
    cbfv8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const cbfv8*>(src + ((size_t)(tid & 4095) * 8 + i) * 8);
        b[i] = *reinterpret_cast<const cbfv8*>(src + ((size_t)(tid & 4095) * 8 + 4 + i) * 8);
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
    sink[tid] = s[0] + s[1] + s[2] + s[3];
}

extern "C" {

int mg_calib_mfma_bf16(const uint16_t* operands, float* sink, int n_workgroups, int trips, double* flop, void* stream) {
    MG_CHECK_ARG(operands && sink && n_workgroups > 0 && n_workgroups <= 65536 && trips > 0, "mg_calib_mfma_bf16: bad arguments");
    hipLaunchKernelGGL(calib_mfma_kernel, dim3((unsigned)n_workgroups), dim3(512), 0, (hipStream_t)stream, operands, sink, trips);
    MG_CHECK_LAUNCH("mg_calib_mfma_bf16");
    if (flop) *flop = (double)n_workgroups * 8.0 * (double)trips * 16.0 * 16384.0;
    return MG_OK;
}

}  // extern "C"
