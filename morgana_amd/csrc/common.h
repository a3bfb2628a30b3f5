// Shared host/device helpers for libmorgana_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/morgana_hip.h"

#define MG_WAVE 64

void mg_set_error(const char* fmt, ...);

#define MG_CHECK_ARG(cond, ...)          \
    do {                                 \
        if (!(cond)) {                   \
            mg_set_error(__VA_ARGS__);   \
            return MG_EINVAL;            \
        }                                \
    } while (0)

// Call right after a kernel launch: turns a launch-time error into MG_ELAUNCH.
#define MG_CHECK_LAUNCH(name)                                                          \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            mg_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));        \
            return MG_ELAUNCH;                                                         \
        }                                                                              \
    } while (0)

static inline int64_t mg_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t mg_align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

#ifdef __HIPCC__
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

// float -> bf16 bits, round to nearest even; the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs.
__device__ __forceinline__ uint16_t mg_f2bf(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float mg_bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

__device__ __forceinline__ float mg_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float mg_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
#endif
