// Operand splits of precision mode 'bf16x3' (split-bf16: three bf16 MFMA products per fp32 product, fp32 accumulate).
//
// The reference multiplies in fp32 end to end (morgana/experiment_builder.py:262-263, morgana/data.py:127); bf16 mode rounds both
// operands of every product to 8 significant bits and reads 5.8e-4 on the reference's 20-step loss curve, fp32 mode (exact-fp32 MFMA,
// 1/16 of the bf16 matrix rate) 1.2e-7.  Between them:  x = hi + lo  with  hi = bf16(x), lo = bf16(x - hi)  carries 16 significant
// bits, and
//     x w  ~=  hi_x hi_w + hi_x lo_w + lo_x hi_w                         (the dropped lo_x lo_w term is 2^-16 relative)
// where every one of the three products is exact in the fp32 accumulator of a bf16 MFMA.  A sum of three GEMMs over the same
// contraction index is ONE GEMM over a contraction index three times as long:
//     [hi_x | hi_x | lo_x] [hi_w | lo_w | hi_w]^T
// so the tile programs of bf16 mode run unchanged on operands written by this file - `order` 0 is the left layout (activation
// side), `order` 1 the right one (weight side); each of the three planes is padded with zeros to `ldp` columns (a multiple of 8,
// 64 for the wide tile programs), which both sides share, so padding multiplies padding.  With `transpose` the planes hold the
// split of src^T (the dgrad operand W^T without a transposed fp32 copy).  Weight gradients contract over the ROWS: they take the
// planes one pair at a time, three accumulating launches (ops.linear_wgrad_x3) - from `order` 2, two SEPARATE planes [hi ; lo], each
// [rows, ldp] with its own zero padding (the weight-gradient tile programs read a row up to its leading dimension: a plane cut out of
// a three-plane row would hand them its neighbours as padding).  `order` 3 / 4: THREE row-stacked planes [hi ; hi ; lo] / [hi ; lo ; hi] -
// the three products as ONE weight-gradient launch over a three times longer row axis, [hi ; hi ; lo]^T [hi ; lo ; hi]; the bias
// gradient (the column sums of the UNSPLIT fp32 values, exact) then comes out of the split pass itself: `colsum` slabs, one per
// workgroup, summed by the library's ordered slab reduce.  The same one-launch weight gradient needs NO row-stacked planes at all:
// a three-plane buffer [rows, 3 ldp] read as [3 rows, ldp] IS a row-interleaved stack, so the gradient split [hi | lo | hi] (order 1,
// the layout its dgrad takes against W^T in order 0) against the forward's own activation split [hi | hi | lo] (order 0) pairs
// (hi, hi), (lo, hi), (hi, lo) - the three products - and the backward splits only the gradient, once (functional.LinearStackFn).
#include "common.h"

#define MG_SPLIT3_MAX_ 16

struct Split3Batch {
    int count;
    mg_split3_desc d[MG_SPLIT3_MAX_];
};

__device__ __forceinline__ void split_pair(float x, uint16_t& hi, uint16_t& lo) {
    hi = mg_f2bf(x);
    lo = mg_f2bf(x - mg_bf2f(hi));
}

typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));

// blockIdx.y = descriptor.  Plain layout: one thread per 8 consecutive columns of one row (two 16-byte loads where the row allows,
// three 16-byte stores); transposed layout: 32 x 32 tiles through LDS.
__global__ __launch_bounds__(256) void split3_kernel(Split3Batch batch) {
    __shared__ float tile[32][33];
    const mg_split3_desc d = batch.d[blockIdx.y];
    const int second = (d.order == 0 || d.order == 3) ? 0 : 1;           // what the middle plane holds: 0 = hi again, 1 = lo
    auto pack = [](const uint16_t* v) {
        return su32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                      (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
    };
    // one row's 8-column chunk: load (+ the fused sigmoid gradient); false where the row has no such chunk
    auto load_chunk = [&](int64_t r, int c0, bool vec, float (&x)[8]) {
        const float* src = d.src + (size_t)r * d.lds + c0;
        if (vec && c0 + 8 <= d.cols) {
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w, x[4] = b.x, x[5] = b.y, x[6] = b.z, x[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = c0 + e < d.cols ? src[e] : 0.f;
        }
        if (d.sig) {                                   // the sigmoid gradient, fused: x <- x s (1 - s)  (the product order of sigmoid_grad_kernel)
            const float* sg = d.sig + (size_t)r * d.ldsig + c0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (c0 + e < d.cols) {
                    const float s = sg[e];
                    x[e] = x[e] * s * (1.f - s);
                }
        }
    };
    if (!d.transpose && d.colsum) {
        // row-stacked planes WITH the column sums of the values: workgroup b < colsum_blocks owns a fixed range of rows, thread =
        // (row lane, 8-column chunk) so that a thread's sums stay in one chunk; lanes added in lane order through LDS
        if ((int)blockIdx.x >= d.colsum_blocks) return;
        const int chunks = d.ldp >> 3, lanes = 256 / chunks;               // host: 256 % chunks == 0
        const int ch = threadIdx.x % chunks, ln = threadIdx.x / chunks, c0 = ch * 8;
        const int64_t per = (d.rows + d.colsum_blocks - 1) / d.colsum_blocks, r_lo = (int64_t)blockIdx.x * per,
                      r_hi = r_lo + per < d.rows ? r_lo + per : d.rows;
        const bool vec = (d.lds & 3) == 0 && ((size_t)d.src & 15) == 0;
        // row-stacked planes (orders 2-4) lie plane_rows x ldp apart, the planes of a three-plane row (orders 0 / 1) ldp apart
        const bool stacked = d.order >= 2;
        const size_t plane = stacked ? (size_t)(d.plane_rows > 0 ? d.plane_rows : d.rows) * d.ldp : (size_t)d.ldp;
        const size_t row_stride = stacked ? (size_t)d.ldp : 3 * (size_t)d.ldp;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int64_t r = r_lo + ln; r < r_hi; r += lanes) {
            float x[8];
            load_chunk(r, c0, vec, x);
            uint16_t hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                split_pair(x[e], hi[e], lo[e]);
                acc[e] += x[e];
            }
            const su32x4 ph = pack(hi), pl = pack(lo);
            uint16_t* dst = d.dst + (size_t)r * row_stride + c0;
            *reinterpret_cast<su32x4*>(dst) = ph;
            if (d.order == 2) {
                *reinterpret_cast<su32x4*>(dst + plane) = pl;
            } else {
                *reinterpret_cast<su32x4*>(dst + plane) = second ? pl : ph;
                *reinterpret_cast<su32x4*>(dst + 2 * plane) = second ? ph : pl;
            }
        }
        __shared__ float sums[256 * 8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sums[(ln * chunks + ch) * 8 + e] = acc[e];
        __syncthreads();
        if (ln == 0) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = 0.f;
            for (int l = 0; l < lanes; ++l)
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] += sums[(l * chunks + ch) * 8 + e];
            float* out = d.colsum + (size_t)blockIdx.x * d.ldp + c0;
#pragma unroll
            for (int e = 0; e < 8; ++e) out[e] = t[e];
        }
        return;
    }
    if (!d.transpose) {
        const int chunks = d.ldp >> 3;
        const int64_t n = d.rows * (int64_t)chunks;
        const bool vec = (d.lds & 3) == 0 && ((size_t)d.src & 15) == 0;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
            const int64_t r = i / chunks;
            const int c0 = (int)(i - r * chunks) * 8;
            float x[8];
            load_chunk(r, c0, vec, x);
            uint16_t hi[8], lo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) split_pair(x[e], hi[e], lo[e]);
            const su32x4 ph = pack(hi), pl = pack(lo);
            if (d.order == 5) {                        // pair planes [hi | lo] in one row of 2 ldp columns
                uint16_t* dst = d.dst + (size_t)r * (2 * (size_t)d.ldp) + c0;
                *reinterpret_cast<su32x4*>(dst) = ph;
                *reinterpret_cast<su32x4*>(dst + d.ldp) = pl;
            } else if (d.order == 2) {
                uint16_t* dst = d.dst + (size_t)r * d.ldp + c0;
                *reinterpret_cast<su32x4*>(dst) = ph;
                *reinterpret_cast<su32x4*>(dst + (size_t)(d.plane_rows > 0 ? d.plane_rows : d.rows) * d.ldp) = pl;
            } else if (d.order >= 3) {
                const size_t plane = (size_t)(d.plane_rows > 0 ? d.plane_rows : d.rows) * d.ldp;
                uint16_t* dst = d.dst + (size_t)r * d.ldp + c0;
                *reinterpret_cast<su32x4*>(dst) = ph;
                *reinterpret_cast<su32x4*>(dst + plane) = second ? pl : ph;
                *reinterpret_cast<su32x4*>(dst + 2 * plane) = second ? ph : pl;
            } else {
                uint16_t* dst = d.dst + (size_t)r * (3 * (size_t)d.ldp) + c0;
                *reinterpret_cast<su32x4*>(dst) = ph;
                *reinterpret_cast<su32x4*>(dst + d.ldp) = second ? pl : ph;
                *reinterpret_cast<su32x4*>(dst + 2 * (size_t)d.ldp) = second ? ph : pl;
            }
        }
    } else {
        // dst [cols, 3 ldp]: row c of dst = column c of src; plane columns r < rows hold values, rows <= r < ldp zeros
        const int tiles_c = (d.cols + 31) / 32, tiles_r = (d.ldp + 31) / 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
            const int c0 = (t % tiles_c) * 32, r0 = (t / tiles_c) * 32;
            __syncthreads();
            for (int j = ty; j < 32; j += 8) {
                const int64_t r = r0 + j;
                const int c = c0 + tx;
                tile[j][tx] = (r < d.rows && c < d.cols) ? d.src[(size_t)r * d.lds + c] : 0.f;
            }
            __syncthreads();
            for (int j = ty; j < 32; j += 8) {
                const int c = c0 + j, r = r0 + tx;
                if (c < d.cols && r < d.ldp) {
                    uint16_t hi, lo;
                    split_pair(tile[tx][j], hi, lo);
                    if (d.order == 5) {
                        uint16_t* dst = d.dst + (size_t)c * (2 * (size_t)d.ldp) + r;
                        dst[0] = hi;
                        dst[d.ldp] = lo;
                        continue;
                    }
                    uint16_t* dst = d.dst + (size_t)c * (3 * (size_t)d.ldp) + r;
                    dst[0] = hi;
                    dst[d.ldp] = second ? lo : hi;
                    dst[2 * (size_t)d.ldp] = second ? hi : lo;
                }
            }
        }
    }
}

extern "C" {

int mg_split3_bf16(const mg_split3_desc* descs, int count, void* stream) {
    MG_CHECK_ARG(descs && count > 0 && count <= MG_SPLIT3_MAX, "mg_split3_bf16: count %d not in 1..%d", count, MG_SPLIT3_MAX);
    static_assert(MG_SPLIT3_MAX == MG_SPLIT3_MAX_, "header and kernel disagree");
    Split3Batch batch;
    batch.count = count;
    int64_t most = 0;
    for (int i = 0; i < count; ++i) {
        const mg_split3_desc& d = descs[i];
        MG_CHECK_ARG(d.src && d.dst && d.rows > 0 && d.cols > 0 && d.lds >= d.cols, "mg_split3_bf16: bad descriptor %d", i);
        MG_CHECK_ARG(!d.sig || (!d.transpose && d.ldsig >= d.cols), "mg_split3_bf16: descriptor %d: a fused sigmoid gradient goes with the plain layouts and needs ldsig >= cols", i);
        MG_CHECK_ARG(d.order == 0 || d.order == 1 || d.order == 5 || (d.order >= 2 && d.order <= 4 && !d.transpose),
                     "mg_split3_bf16: descriptor %d: order %d is none of 0 (hi|hi|lo), 1 (hi|lo|hi), 2 (separate planes hi ; lo), 3 (hi ; hi ; lo), 4 (hi ; lo ; hi) (2-4 not transposed), 5 (pair planes hi|lo)", i, d.order);
        MG_CHECK_ARG(d.order != 5 || (!d.colsum && !d.sig), "mg_split3_bf16: descriptor %d: pair planes take neither column sums nor a fused sigmoid gradient", i);
        MG_CHECK_ARG(d.plane_rows == 0 || (d.order >= 2 && d.order <= 4 && d.plane_rows >= d.rows), "mg_split3_bf16: descriptor %d: plane_rows %lld goes with order 2 and must cover the %lld rows", i, (long long)d.plane_rows, (long long)d.rows);
        MG_CHECK_ARG(!d.colsum || (!d.transpose && d.colsum_blocks >= 1 && d.colsum_blocks <= 4096 && d.ldp <= 2048 && 256 % (d.ldp / 8) == 0),
                     "mg_split3_bf16: descriptor %d: column sums go with the plain layouts, 1..4096 blocks and plane widths of 8, 16, ... 2048 columns that divide 2048 (ldp %d)", i, d.ldp);
        MG_CHECK_ARG(d.ldp % 8 == 0 && ((size_t)d.dst & 15) == 0, "mg_split3_bf16: descriptor %d: ldp %d must be a multiple of 8 and dst 16-byte aligned", i, d.ldp);
        if (d.transpose)
            MG_CHECK_ARG(d.ldp >= d.rows && d.rows < 2147483647LL, "mg_split3_bf16: descriptor %d: transposed planes of %d columns cannot hold %lld rows", i, d.ldp, (long long)d.rows);
        else
            MG_CHECK_ARG(d.ldp >= d.cols, "mg_split3_bf16: descriptor %d: planes of %d columns cannot hold %d columns", i, d.ldp, d.cols);
        batch.d[i] = d;
        const int64_t work = d.transpose ? (int64_t)((d.cols + 31) / 32) * ((d.ldp + 31) / 32) : mg_ceil_div(d.rows * (int64_t)(d.ldp / 8), 256);
        if (work > most) most = work;
        if (d.colsum && d.colsum_blocks > most) most = d.colsum_blocks;
    }
    int64_t blocks = most;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)blocks, count), dim3(256), 0, (hipStream_t)stream, batch);
    MG_CHECK_LAUNCH("mg_split3_bf16");
    return MG_OK;
}

}  // extern "C"
