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
extern int g_mg_tuning[];
// Lab builds (make lab / diag, -DMG_EXPERIMENTS) only: key 1 != 0 makes the weight-gradient entry points skip their slab-reduce launch
// (results then invalid) so that a script can time the producing kernel alone.  The product library refuses the key.
#define MG_TUNE_SKIP_REDUCE 1

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
// bf16-mode sigmoid: one v_exp_f32 and one v_rcp_f32 (1 ulp each, far below the bf16 rounding of the result).  1/(1+e) through
// `/` or __frcp_rn is the IEEE division sequence (div_scale, rcp, 6 fma, div_fmas, div_fixup) and exp2f adds denormal
// handling: ~20 VALU instructions per value, which made the layer-1 forward epilogue a third of its kernel.
__device__ __forceinline__ float mg_sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// The 32x32x16 bf16 MFMA of the large GEMM kernels behind one name, so that a PROBE build (make probe16 -> libmorgana_hip_probe16.so,
// -DMG_PROBE16, results garbage, never loaded by the package) can issue two v_mfma_f32_16x16x32_bf16 in its place on the same
// registers: same matrix cycles, same operand traffic, a different clock under load (MI355X_MICROARCH.md, DVFS give-back item 7).
typedef __bf16 mg_bfv8 __attribute__((ext_vector_type(8)));
#ifdef MG_PROBE16
#define MG_MFMA_PER_TILE 2
__device__ __forceinline__ f32x16 mg_mfma_32x32x16(mg_bfv8 a, mg_bfv8 b, f32x16 c) {
    f32x4 lo = {c[0], c[1], c[2], c[3]}, hi = {c[8], c[9], c[10], c[11]};
    lo = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, hi, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        c[r] = lo[r];
        c[8 + r] = hi[r];
    }
    asm volatile("" : "+v"(c));                        // the other registers stay opaque: the epilogues keep all their work
    return c;
}
#else
#define MG_MFMA_PER_TILE 1
__device__ __forceinline__ f32x16 mg_mfma_32x32x16(mg_bfv8 a, mg_bfv8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#endif

// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes (from its own address `src`) land at LDS byte (lds_wave_base + 16 l); no VGPR staging,
// completion counted by vmcnt.  Inline asm on purpose: hipcc keeps no record of the pending LDS write (its own waits stay
// conservative - a wait for a tracked load also covers every older DMA); the caller drains vmcnt before the barrier in front of the
// first read.  M0 is saved and restored inside the statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void mg_glds16(const void* src, unsigned char* lds_wave_base) {
    const unsigned lds_off = (unsigned)(unsigned long long)((__attribute__((address_space(3))) unsigned char*)lds_wave_base);
    const unsigned lds_uni = __builtin_amdgcn_readfirstlane(lds_off);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_uni)
                 : "memory");
}

// Sum over each row of 16 lanes, in every lane of the row, on the DPP path (row_ror 8, 4, 2, 1: four VALU instructions, where four
// __shfl_xor go through the LDS crossbar one after the other).  Bit for bit the xor butterfly 8, 4, 2, 1: at every level a lane's
// partner belongs to the same class of lanes either way, and a + b == b + a.
template <int CTRL>
__device__ __forceinline__ float mg_dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float mg_row16_sum(float v) {
    v += mg_dpp_f32<0x128>(v);
    v += mg_dpp_f32<0x124>(v);
    v += mg_dpp_f32<0x122>(v);
    v += mg_dpp_f32<0x121>(v);
    return v;
}

__device__ __forceinline__ float mg_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
#endif


// ---------------------------------------------------------------------------------------------------------------------
// In-kernel stamps (cdna_hip_programming.md section 7).  Compiled only into the DIAGNOSTIC library
// (make diag -> ../libmorgana_hip_diag.so, -DMG_STAMPS); the product library contains none of this.
// Each stamping kernel owns a file-local buffer: [block][wave 0 or 4][MG_STAMP_SLOTS] shader-clock values.
// ---------------------------------------------------------------------------------------------------------------------
#define MG_STAMP_SLOTS 16
#define MG_STAMP_BLOCKS 4096
#ifdef MG_STAMPS
#define MG_STAMP_DECL(name) __device__ unsigned long long name[MG_STAMP_BLOCKS * 2 * MG_STAMP_SLOTS]
#define MG_STAMP(var)                                                                         \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#define MG_STAMP_REAL(var)                                                                    \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");       \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
#define MG_STAMP_ADD(sum, t1, t0) ((sum) += (t1) - (t0))
#define MG_STAMP_STORE(buf, block, wave, lane, slot, value)                                   \
    do {                                                                                      \
        if ((block) < MG_STAMP_BLOCKS && ((wave) & 3) == 0 && (lane) == 0)                    \
            buf[((size_t)(block) * 2 + ((wave) >> 2)) * MG_STAMP_SLOTS + (slot)] = (value);   \
    } while (0)
#else
#define MG_STAMP_DECL(name)
#define MG_STAMP(var) do { } while (0)
#define MG_STAMP_REAL(var) do { } while (0)
#define MG_STAMP_ADD(sum, t1, t0) do { } while (0)
#define MG_STAMP_STORE(buf, block, wave, lane, slot, value) do { } while (0)
#endif
