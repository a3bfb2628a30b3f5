// Phone-rate first layer.  The model input of the README F0Model and of the RNN_SPSS layout is
// upsample_to_repetitions(normalised_lab, dur) (morgana/utils.py:175-228): every phone row repeated dur[b, p] times, then
// Linear(600, 512) + Sigmoid over all B*T frame rows (README.rst:65-73, morgana/utils.py:401-418).  A Linear commutes with
// repeating rows:  gather(X) W^T = gather(X W^T).  So the 600-wide product is done ONCE PER PHONE (B*P rows, 12.5x fewer than
// frames at the synthetic 12.5 frames per phone) and the frame-rate activation is a gather of that table:
//     forward   Z = X_phone W^T (fp32, existing GEMM)            H[f] = act(Z[row(f)] + b)             (expand_rows_kernel)
//     backward  dZ_phone[r] = sum over the frames f of phone r of dZ[f]   (segment_sum_kernel)         dW = dZ_phone^T X_phone
// Per frame row the fp32 dot product, the bias add and the sigmoid are the same operations in the same order as in the
// frame-rate GEMM epilogue, so H is unchanged; dW sums the same bf16 (fp32 in parity mode) frame gradients, grouped by phone.
// Both kernels are HBM bound: expand writes M x N activations (reads of Z hit L2: a phone's row is used by consecutive
// frames), segment_sum reads M x N gradients once.  Frames with row -1 (padding past an utterance's end, the reference's zero
// row, utils.py:206-214) take Z = 0 in expand; in segment_sum their gradients go to `extra` rows behind the table's R rows
// (the input rows there are zero, so dW ignores them and the bias gradient still sums every frame).
#include "common.h"

typedef uint32_t pr_u32x4 __attribute__((ext_vector_type(4)));

// seg_start[r] / seg_end[r]: the run of frames whose row is r (rows of one phone are consecutive frames); untouched (0, 0)
// for phones without frames.
__global__ __launch_bounds__(256) void segment_bounds_kernel(const int32_t* __restrict__ rows, int64_t M, int R,
                                                             int32_t* __restrict__ seg_start, int32_t* __restrict__ seg_end,
                                                             int32_t* __restrict__ rows_mapped, int pad_row) {
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= M) return;
    const int r = rows[f];
    if (rows_mapped) rows_mapped[f] = r < 0 ? pad_row : r;
    if (r < 0 || r >= R) return;
    const int prev = f > 0 ? rows[f - 1] : -2, next = f + 1 < M ? rows[f + 1] : -2;
    if (prev != r) seg_start[r] = (int32_t)f;
    if (next != r) seg_end[r] = (int32_t)(f + 1);
}

template <typename OutT> struct PrStore;
template <> struct PrStore<uint16_t> {
    static __device__ __forceinline__ void put(uint16_t* dst, const float (&v)[8]) {
        pr_u32x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = (uint32_t)mg_f2bf(v[2 * e]) | ((uint32_t)mg_f2bf(v[2 * e + 1]) << 16);
        *reinterpret_cast<pr_u32x4*>(dst) = pk;
    }
};
template <> struct PrStore<float> {
    static __device__ __forceinline__ void put(float* dst, const float (&v)[8]) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
};

// H[f, c] = act(Z[rows[f], c] + bias[c]).  A thread owns one 8-column chunk (its bias values stay in registers) and walks frame
// rows with a grid stride, four rows per trip so that four Z-row loads are in flight; threadIdx.y picks the row inside a block.
// N % 8 == 0; columns N..ldh-1 are zeroed.
#define EXPAND_ROWS_PER_BLOCK 4
#define EXPAND_UNROLL 4
template <typename OutT, bool FAST>
__global__ __launch_bounds__(256) void expand_rows_kernel(const float* __restrict__ Z, int ldz, const int32_t* __restrict__ rows,
                                                          int64_t M, const float* __restrict__ bias, int N, int act,
                                                          OutT* __restrict__ H, int ldh) {
    const int c = (blockIdx.y * 64 + threadIdx.x) * 8;
    if (c >= ldh) return;
    const bool live = c < N;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (live && bias) ? bias[c + e] : 0.f;
    const int64_t stride = (int64_t)gridDim.x * EXPAND_ROWS_PER_BLOCK;
    for (int64_t f0 = (int64_t)blockIdx.x * EXPAND_ROWS_PER_BLOCK + threadIdx.y; f0 < M; f0 += stride * EXPAND_UNROLL) {
        f32x4 lo[EXPAND_UNROLL], hi[EXPAND_UNROLL];
#pragma unroll
        for (int u = 0; u < EXPAND_UNROLL; ++u) {
            const int64_t f = f0 + u * stride;
            const int r = (live && f < M) ? rows[f] : -1;
            if (r >= 0) {
                lo[u] = *reinterpret_cast<const f32x4*>(Z + (size_t)r * ldz + c);
                hi[u] = *reinterpret_cast<const f32x4*>(Z + (size_t)r * ldz + c + 4);
            } else {
                lo[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                hi[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < EXPAND_UNROLL; ++u) {
            const int64_t f = f0 + u * stride;
            if (f >= M) break;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = 0.f;
                if (live) {
                    x = (e < 4 ? lo[u][e & 3] : hi[u][e & 3]) + bv[e];
                    if (act == MG_ACT_SIGMOID) x = FAST ? mg_sigmoid_fast(x) : mg_sigmoid(x);
                }
                v[e] = x;
            }
            PrStore<OutT>::put(H + (size_t)f * ldh + c, v);
        }
    }
}

template <typename T> struct PrLoad;
template <> struct PrLoad<uint16_t> {
    static __device__ __forceinline__ void add(const uint16_t* src, float (&acc)[8]) {
        const pr_u32x4 pk = *reinterpret_cast<const pr_u32x4*>(src);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[2 * e] += __uint_as_float(pk[e] << 16);
            acc[2 * e + 1] += __uint_as_float(pk[e] & 0xffff0000u);
        }
    }
};
template <> struct PrLoad<float> {
    static __device__ __forceinline__ void add(const float* src, float (&acc)[8]) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += lo[e]; acc[4 + e] += hi[e]; }
    }
};

// out[r, c] = sum of G[f, c] over the frames of row r (fp32 accumulation in frame order), r < R; out[R + j, c] = sum over the
// frames of the j-th of `extra` equal shares of the frame axis whose row is -1.  blockIdx.x = table row, threads = 8-column chunks (loop if ldo / 8 > blockDim).
template <typename T>
__global__ __launch_bounds__(64) void segment_sum_kernel(const T* __restrict__ G, int ldg, const int32_t* __restrict__ rows,
                                                        int64_t M, const int32_t* __restrict__ seg_start,
                                                        const int32_t* __restrict__ seg_end, int R, int extra, int N,
                                                        T* __restrict__ out, int ldo) {
    const int r = blockIdx.x;
    const int c = (blockIdx.y * 64 + threadIdx.x) * 8;       // this thread's 8 columns; all 64 lanes stay for the ballots below
    const bool live = c < N;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (r < R) {
        if (live) {
            const int f0 = seg_start[r], f1 = seg_end[r];
            const T* src = G + (size_t)f0 * ldg + c;
            int f = f0;
            for (; f + 4 <= f1; f += 4) {          // four independent loads in flight, added in frame order
                float t0[8], t1[8], t2[8], t3[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) t0[e] = t1[e] = t2[e] = t3[e] = 0.f;
                PrLoad<T>::add(src, t0);
                PrLoad<T>::add(src + (size_t)ldg, t1);
                PrLoad<T>::add(src + (size_t)2 * ldg, t2);
                PrLoad<T>::add(src + (size_t)3 * ldg, t3);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = (((acc[e] + t0[e]) + t1[e]) + t2[e]) + t3[e];
                src += (size_t)4 * ldg;
            }
            for (; f < f1; ++f, src += ldg) PrLoad<T>::add(src, acc);
        }
    } else {
        // padding frames of this block's share [j C, (j + 1) C) of the frame axis: 64 row ids per look, then only the frames that
        // are padding
        const int64_t chunk = (M + extra - 1) / extra, lo = (int64_t)(r - R) * chunk, hi = lo + chunk < M ? lo + chunk : M;
        for (int64_t base = lo; base < hi; base += 64) {
            const int64_t mine = base + threadIdx.x;
            unsigned long long pads = __ballot(mine < hi && (rows[mine] < 0 || rows[mine] >= R));      // -1, or mapped to a pad row
            while (pads) {
                const int bit = __builtin_ctzll(pads);
                pads &= pads - 1;
                if (live) PrLoad<T>::add(G + (size_t)(base + bit) * ldg + c, acc);
            }
        }
    }
    if (c < ldo) PrStore<T>::put(out + (size_t)r * ldo + c, acc);
}

extern "C" {

int mg_segment_bounds(const int32_t* rows, int64_t M, int R, int32_t* seg_start, int32_t* seg_end, int32_t* rows_mapped, int pad_row,
                      void* stream) {
    MG_CHECK_ARG(rows && seg_start && seg_end && M > 0 && R > 0 && M < 2147483647LL, "mg_segment_bounds: bad arguments (M=%lld R=%d)",
                 (long long)M, R);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(seg_start, 0, (size_t)R * sizeof(int32_t), st) != hipSuccess ||
        hipMemsetAsync(seg_end, 0, (size_t)R * sizeof(int32_t), st) != hipSuccess) {
        mg_set_error("mg_segment_bounds: memset failed");
        return MG_ELAUNCH;
    }
    hipLaunchKernelGGL(segment_bounds_kernel, dim3((unsigned)mg_ceil_div(M, 256)), dim3(256), 0, st, rows, M, R, seg_start, seg_end,
                       rows_mapped, pad_row);
    MG_CHECK_LAUNCH("mg_segment_bounds");
    return MG_OK;
}

int mg_expand_rows(const float* Z, int ldz, const int32_t* rows, int64_t M, const float* bias, int N, int act, void* H, int ldh,
                   int h_bf16, void* stream) {
    MG_CHECK_ARG(Z && rows && H && M > 0 && N > 0, "mg_expand_rows: bad arguments (M=%lld N=%d)", (long long)M, N);
    MG_CHECK_ARG(N % 8 == 0 && ldz % 4 == 0 && ldz >= N && ldh % 8 == 0 && ldh >= N, "mg_expand_rows: N=%d ldz=%d ldh=%d (N, ldh multiples of 8)", N,
                 ldz, ldh);
    MG_CHECK_ARG(((uintptr_t)Z % 16) == 0 && ((uintptr_t)H % 16) == 0, "mg_expand_rows: buffers must be 16-byte aligned");
    MG_CHECK_ARG(act == MG_ACT_NONE || act == MG_ACT_SIGMOID, "mg_expand_rows: unknown activation %d", act);
    const unsigned col_blocks = (unsigned)mg_ceil_div(ldh / 8, 64);
    int64_t row_blocks = mg_ceil_div(M, EXPAND_ROWS_PER_BLOCK * EXPAND_UNROLL);
    if (row_blocks > 16384) row_blocks = 16384;
    const dim3 grid((unsigned)row_blocks, col_blocks), block(64, EXPAND_ROWS_PER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    if (h_bf16)
        hipLaunchKernelGGL((expand_rows_kernel<uint16_t, true>), grid, block, 0, st, Z, ldz, rows, M, bias, N, act, (uint16_t*)H, ldh);
    else
        hipLaunchKernelGGL((expand_rows_kernel<float, false>), grid, block, 0, st, Z, ldz, rows, M, bias, N, act, (float*)H, ldh);
    MG_CHECK_LAUNCH("mg_expand_rows");
    return MG_OK;
}

int mg_segment_sum(const void* G, int ldg, int g_bf16, const int32_t* rows, int64_t M, const int32_t* seg_start,
                   const int32_t* seg_end, int R, int extra, int N, void* out, int ldo, void* stream) {
    MG_CHECK_ARG(G && rows && seg_start && seg_end && out && M > 0 && R > 0 && extra >= 0 && N > 0,
                 "mg_segment_sum: bad arguments (M=%lld R=%d extra=%d N=%d)", (long long)M, R, extra, N);
    MG_CHECK_ARG(N % 8 == 0 && ldg % 8 == 0 && ldg >= N && ldo % 8 == 0 && ldo >= N, "mg_segment_sum: N=%d ldg=%d ldo=%d must be multiples of 8", N,
                 ldg, ldo);
    MG_CHECK_ARG(((uintptr_t)G % 16) == 0 && ((uintptr_t)out % 16) == 0, "mg_segment_sum: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)(R + extra), (unsigned)mg_ceil_div(ldo / 8, 64));
    if (g_bf16)
        hipLaunchKernelGGL((segment_sum_kernel<uint16_t>), grid, dim3(64), 0, st, (const uint16_t*)G, ldg, rows, M, seg_start, seg_end, R, extra, N,
                           (uint16_t*)out, ldo);
    else
        hipLaunchKernelGGL((segment_sum_kernel<float>), grid, dim3(64), 0, st, (const float*)G, ldg, rows, M, seg_start, seg_end, R, extra, N,
                           (float*)out, ldo);
    MG_CHECK_LAUNCH("mg_segment_sum");
    return MG_OK;
}

}  // extern "C"
