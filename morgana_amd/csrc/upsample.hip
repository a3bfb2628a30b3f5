// K1 - upsample_to_repetitions as a segmented repeat (reference: morgana/utils.py:175-228).
//
// The reference builds the frame->phone map on the host (two D2H syncs, a Python loop of np.repeat over the batch,
// one H2D) and then runs an advanced-index gather.  Here the map is built on the device: one workgroup per
// utterance scans the durations in LDS and every frame binary-searches its phone (coalesced int stores); the gather
// moves whole rows with 16-byte lanes.  Both are HBM-bound: algorithmic bytes = B*P*F*4 read (L2/MALL resident,
// each phone row is re-read dur times) + B*T*F*s written.
#include "common.h"

#define MG_MAX_PHONES 12288

// Inclusive scan of dur[b, 0:P] into cum[0:P] (LDS, int32).  256 threads; scratch holds 256 ints.
__device__ __forceinline__ void block_scan_durations(const int64_t* __restrict__ dur_row, int P, int* cum, int* scratch) {
    const int tid = threadIdx.x;
    const int per = (P + 255) / 256;
    const int lo = tid * per;
    const int hi = min(lo + per, P);
    int local = 0;
    for (int p = lo; p < hi; ++p) {
        long long d = dur_row[p];
        local += d > 0 ? (int)d : 0;
    }
    scratch[tid] = local;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 256 partials.
    for (int off = 1; off < 256; off <<= 1) {
        int v = tid >= off ? scratch[tid - off] : 0;
        __syncthreads();
        scratch[tid] += v;
        __syncthreads();
    }
    int run = tid > 0 ? scratch[tid - 1] : 0;
    for (int p = lo; p < hi; ++p) {
        long long d = dur_row[p];
        run += d > 0 ? (int)d : 0;
        cum[p] = run;
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void upsample_lengths_kernel(const int64_t* __restrict__ dur, int P,
                                                               int64_t* __restrict__ n_frames,
                                                               unsigned long long* __restrict__ tmax) {
    __shared__ long long red[4];
    const int b = blockIdx.x;
    long long s = 0;
    for (int p = threadIdx.x; p < P; p += 256) s += dur[(size_t)b * P + p];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long total = red[0] + red[1] + red[2] + red[3];
        if (n_frames) n_frames[b] = total;
        if (tmax && total > 0) atomicMax(tmax, (unsigned long long)total);
    }
}

__global__ __launch_bounds__(256) void upsample_index_kernel(const int64_t* __restrict__ dur, int P, int t_cap,
                                                             int64_t* __restrict__ idx64, int32_t* __restrict__ rows32,
                                                             int32_t* __restrict__ rows_mapped, int pad_row,
                                                             int32_t* __restrict__ seg_start, int32_t* __restrict__ seg_end) {
    extern __shared__ __attribute__((aligned(16))) int smem_i[];
    int* scratch = smem_i;
    int* cum = smem_i + 256;
    const int b = blockIdx.x;
    block_scan_durations(dur + (size_t)b * P, P, cum, scratch);
    const int total = P > 0 ? cum[P - 1] : 0;
    for (int t = threadIdx.x; t < t_cap; t += 256) {
        int phone = -1;
        if (t < total) {
            // first p with cum[p] > t  (phones of duration 0 are skipped, as np.repeat does)
            int lo = 0, hi = P - 1;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (cum[mid] > t) hi = mid; else lo = mid + 1;
            }
            phone = lo;
        }
        const size_t o = (size_t)b * t_cap + t;
        if (idx64) idx64[o] = phone;
        if (rows32) rows32[o] = phone < 0 ? -1 : b * P + phone;
        if (rows_mapped) rows_mapped[o] = phone < 0 ? pad_row : b * P + phone;
    }
    if (seg_start) {
        // the frame run of each phone row, as mg_segment_bounds finds it from the map: (0, 0) for a phone without frames
        for (int p = threadIdx.x; p < P; p += 256) {
            const int s = min(p ? cum[p - 1] : 0, t_cap), e = min(cum[p], t_cap);
            seg_start[(size_t)b * P + p] = e > s ? b * t_cap + s : 0;
            seg_end[(size_t)b * P + p] = e > s ? b * t_cap + e : 0;
        }
    }
}


// Segment index maps (reference: split_to_segments / get_segment_ends, morgana/utils.py:231-330).  One workgroup per
// sequence; seg_lens [S] is scanned in LDS exactly as the durations of K1 are.
//   split[b, s, j] = b * T + start_s + j  for j < len_s (and start_s + j < T), else -1     (the reference's segment_idxs)
//   ends[b, s]     = b * T + cumsum_s - 1 for len_s > 0 (and cumsum_s <= T), else -1      (cumsum * mask - 1, :325-328)
__global__ __launch_bounds__(256) void segment_index_kernel(const int64_t* __restrict__ seg_lens, int S, int T, int L,
                                                            int32_t* __restrict__ split, int32_t* __restrict__ ends) {
    extern __shared__ __attribute__((aligned(16))) int smem_i[];
    int* scratch = smem_i;
    int* cum = smem_i + 256;
    const int b = blockIdx.x;
    block_scan_durations(seg_lens + (size_t)b * S, S, cum, scratch);
    if (ends) {
        for (int s = threadIdx.x; s < S; s += 256) {
            const int len = cum[s] - (s ? cum[s - 1] : 0);
            ends[(size_t)b * S + s] = (len > 0 && cum[s] <= T) ? b * T + cum[s] - 1 : -1;
        }
    }
    if (split) {
        const int64_t n = (int64_t)S * L;
        for (int64_t e = threadIdx.x; e < n; e += 256) {
            const int s = (int)(e / L), j = (int)(e - (int64_t)s * L);
            const int start = s ? cum[s - 1] : 0;
            const int len = cum[s] - start;
            split[(size_t)b * n + e] = (j < len && start + j < T) ? b * T + start + j : -1;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Packed frames.  A ragged batch (B, T, .) zero padded to its longest utterance (collate_fn, morgana/data.py:183-193) holds
// sum_b min(seq_len[b], T) valid frame rows; the reference runs its Linear layers on all B*T rows (morgana/utils.py:401-418 applies
// nn.Linear to the padded tensor) and masks the loss (losses.py:37-39).  The row-wise layers here run on the valid rows plus ONE
// representative padding row (a zero input row: what every padded frame holds behind a recurrent wrapper, utils.py:383):
//   rows    [total + 1] : dense row b*T + t of packed row i, in (b, t) order; rows[total] = -1 (the zero row)
//   inverse [B*T]       : packed row of dense row (b, t); total for padded frames
//   offsets [B + 1]     : first packed row of utterance b; offsets[B] = total
// frame_layout_scan_kernel: one workgroup scans the clamped lengths; frame_layout_fill_kernel: one thread per dense row.
// `total` is the caller's (host-side) sum of the lengths: rows and inverse are sized by it, so entries at or beyond it are never written.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frame_layout_scan_kernel(const int64_t* __restrict__ seq_len, int B, int T, int32_t* __restrict__ offsets) {
    __shared__ int scratch[256];
    const int tid = threadIdx.x;
    const int per = (B + 255) / 256;
    const int lo = tid * per, hi = min(lo + per, B);
    int local = 0;
    for (int b = lo; b < hi; ++b) {
        const long long n = seq_len[b];
        local += n < 0 ? 0 : (n > T ? T : (int)n);
    }
    scratch[tid] = local;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = tid >= off ? scratch[tid - off] : 0;
        __syncthreads();
        scratch[tid] += v;
        __syncthreads();
    }
    int run = tid > 0 ? scratch[tid - 1] : 0;
    for (int b = lo; b < hi; ++b) {
        offsets[b] = run;
        const long long n = seq_len[b];
        run += n < 0 ? 0 : (n > T ? T : (int)n);
    }
    if (tid == 255) offsets[B] = scratch[255];
}

__global__ __launch_bounds__(256) void frame_layout_fill_kernel(const int64_t* __restrict__ seq_len, const int32_t* __restrict__ offsets, int B,
                                                                int T, int total, int32_t* __restrict__ rows, int32_t* __restrict__ inverse) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m == 0) rows[total] = -1;
    // a host total LARGER than the device lengths add up to (taken before T clipped them): the surplus packed rows name no frame -
    // they gather the zero row like the representative padding row instead of keeping whatever the allocation held (total <= B T)
    if (m >= offsets[B] && m < total) rows[m] = -1;
    if (m >= (int64_t)B * T) return;
    const int b = (int)(m / T), t = (int)(m - (int64_t)b * T);
    const long long n = seq_len[b];
    int idx = total;
    if (t < n) {
        const int i = offsets[b] + t;
        if (i < total) {                 // a host total smaller than the device lengths: the surplus frames count as padding
            idx = i;
            rows[i] = (int32_t)m;
        }
    }
    inverse[m] = idx;
}

// out[d] = sum over padded frames (t >= seq_len[b]) of g[b, t, d]: the gradient that reaches the ONE representative padding row of the
// packed form from all the dense rows it stands for.  Two fixed-order stages (deterministic): partial[b][chunk][d] over 256-frame
// chunks (chunks without padded frames write zeros without reading), then one column sum over the B * chunks partials.
#define PAD_CHUNK 256
__global__ __launch_bounds__(256) void pad_rows_colsum_stage1_kernel(const float* __restrict__ g, const int64_t* __restrict__ seq_len, int T, int D,
                                                                     int n_chunks, float* __restrict__ partial) {
    extern __shared__ float red_cs[];                     // [4][D]
    const int b = blockIdx.y, c = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long n = seq_len[b];
    const int t_lo = max(c * PAD_CHUNK, n < 0 ? 0 : (n > T ? T : (int)n)), t_hi = min(c * PAD_CHUNK + PAD_CHUNK, T);
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        float s = 0.f;
        if (d < D)
            for (int t = t_lo + wave; t < t_hi; t += 4) s += g[((size_t)b * T + t) * D + d];
        if (d < D) red_cs[wave * D + d] = s;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256)
        partial[((size_t)b * n_chunks + c) * D + d] = (red_cs[d] + red_cs[D + d]) + (red_cs[2 * D + d] + red_cs[3 * D + d]);
}

__global__ __launch_bounds__(1024) void pad_rows_colsum_stage2_kernel(const float* __restrict__ partial, int n_partials, int D, float* __restrict__ out) {
    __shared__ float red2[16][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int d = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (d < D) {
#pragma unroll 8
        for (int i = wave; i < n_partials; i += 16) s += partial[(size_t)i * D + d];
    }
    red2[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && d < D) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red2[w][lane];
        out[d] = t;
    }
}

// dst[rows[m], :] = src[m, :] for rows[m] >= 0 (the targets are distinct: the adjoint of a gather whose rows are unique).
__global__ __launch_bounds__(256) void scatter_rows_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                                               float* __restrict__ dst, int64_t M, int F) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t m = wave; m < M; m += n_waves) {
        const int r = rows[m];
        if (r < 0) continue;
        const float* s = src + (size_t)m * F;
        float* o = dst + (size_t)r * F;
        for (int c = lane; c < F; c += 64) o[c] = s[c];
    }
}

// One wave per output row, 16 bytes per lane.
template <bool VEC4>
__global__ __launch_bounds__(256) void gather_rows_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                                              float* __restrict__ out, int64_t M, int F) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t m = wave; m < M; m += n_waves) {
        const int r = rows ? rows[m] : (int)m;
        if (VEC4) {
            const int f4 = F >> 2;
            const f32x4* s = reinterpret_cast<const f32x4*>(src + (size_t)(r < 0 ? 0 : r) * F);
            f32x4* o = reinterpret_cast<f32x4*>(out + (size_t)m * F);
            for (int c = lane; c < f4; c += 64) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r >= 0) v = s[c];
                o[c] = v;
            }
        } else {
            const float* s = src + (size_t)(r < 0 ? 0 : r) * F;
            float* o = out + (size_t)m * F;
            for (int c = lane; c < F; c += 64) o[c] = r >= 0 ? s[c] : 0.f;
        }
    }
}

// bf16 output, 8 elements (16 bytes) per lane, zero fill up to ldo.
__global__ __launch_bounds__(256) void gather_rows_bf16_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                                               uint16_t* __restrict__ out, int64_t M, int F, int ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int chunks = ldo >> 3;
    const bool vec = (F & 3) == 0;
    for (int64_t m = wave; m < M; m += n_waves) {
        const int r = rows ? rows[m] : (int)m;
        const float* s = src + (size_t)(r < 0 ? 0 : r) * F;
        uint16_t* o = out + (size_t)m * ldo;
        for (int c = lane; c < chunks; c += 64) {
            const int k = c << 3;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            if (r >= 0) {
                if (vec && k + 8 <= F) {
                    f32x4 a = *reinterpret_cast<const f32x4*>(s + k);
                    f32x4 c4 = *reinterpret_cast<const f32x4*>(s + k + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                    v[4] = c4.x; v[5] = c4.y; v[6] = c4.z; v[7] = c4.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (k + j < F) v[j] = s[k + j];
                }
            }
            bf16x8 p;
#pragma unroll
            for (int j = 0; j < 8; ++j) p[j] = (short)mg_f2bf(v[j]);
            *reinterpret_cast<bf16x8*>(o + k) = p;
        }
    }
}


// Gather + concat: out[m, 0:F] = src[rows[m], :] (0 for rows[m] < 0), out[m, F:F+C] = extra[m, 0:C], out[m, F+C:ldo] = 0.
// The layer-1 input of the shipped models: upsampled labels next to the frame-level counters
// (models/RNN_SPSS.py:76-81, models/f0_test_model.py:78-79: upsample_to_repetitions + torch.cat).  One wave per row;
// OUT_BF16 writes 8 converted elements (16 bytes) per lane, the f32 form writes one float per lane and step.
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void gather_concat_kernel(const float* __restrict__ src, const int32_t* __restrict__ rows,
                                                            const float* __restrict__ extra, void* __restrict__ out_, int64_t M,
                                                            int F, int C, int ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const bool vec = (F & 3) == 0;
    for (int64_t m = wave; m < M; m += n_waves) {
        const int r = rows ? rows[m] : (int)m;
        const float* s = src + (size_t)(r < 0 ? 0 : r) * F;
        const float* x = extra + (size_t)m * C;
        if (OUT_BF16) {
            uint16_t* o = (uint16_t*)out_ + (size_t)m * ldo;
            for (int c = lane; c < (ldo >> 3); c += 64) {
                const int k = c << 3;
                float v[8];
                if (r >= 0 && vec && k + 8 <= F) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(s + k);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(s + k + 4);
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
                    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int e = k + j;
                        v[j] = e < F ? (r >= 0 ? s[e] : 0.f) : (e < F + C ? x[e - F] : 0.f);
                    }
                }
                bf16x8 p;
#pragma unroll
                for (int j = 0; j < 8; ++j) p[j] = (short)mg_f2bf(v[j]);
                *reinterpret_cast<bf16x8*>(o + k) = p;
            }
        } else {
            float* o = (float*)out_ + (size_t)m * ldo;
            for (int e = lane; e < ldo; e += 64) o[e] = e < F ? (r >= 0 ? s[e] : 0.f) : (e < F + C ? x[e - F] : 0.f);
        }
    }
}

// grad_src[b,p,:] = sum_{t in phone p} grad_out[b,t,:].  grid (B, phone_chunks); one wave per phone.
__global__ __launch_bounds__(256) void upsample_backward_kernel(const float* __restrict__ grad_out, const int64_t* __restrict__ dur,
                                                                float* __restrict__ grad_src, int P, int T, int F) {
    extern __shared__ __attribute__((aligned(16))) int smem_i[];
    int* scratch = smem_i;
    int* cum = smem_i + 256;
    const int b = blockIdx.x;
    block_scan_durations(dur + (size_t)b * P, P, cum, scratch);
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.y * 4;
    for (int p = wave; p < P; p += n_waves) {
        const int start = min(p > 0 ? cum[p - 1] : 0, T);
        const int end = min(cum[p], T);
        const float* g = grad_out + (size_t)b * T * F;
        float* o = grad_src + ((size_t)b * P + p) * F;
        for (int c = lane; c < F; c += 64) {
            float acc = 0.f;
            for (int t = start; t < end; ++t) acc += g[(size_t)t * F + c];
            o[c] = acc;
        }
    }
}

extern "C" {

int mg_upsample_lengths(const int64_t* dur, int B, int P, int64_t* n_frames, int64_t* tmax, void* stream) {
    MG_CHECK_ARG(dur && B > 0 && P > 0, "mg_upsample_lengths: need dur, B > 0, P > 0 (B=%d P=%d)", B, P);
    hipStream_t st = (hipStream_t)stream;
    if (tmax) {
        if (hipMemsetAsync(tmax, 0, sizeof(int64_t), st) != hipSuccess) {
            mg_set_error("mg_upsample_lengths: memset failed");
            return MG_ELAUNCH;
        }
    }
    hipLaunchKernelGGL(upsample_lengths_kernel, dim3(B), dim3(256), 0, st, dur, P, n_frames, (unsigned long long*)tmax);
    MG_CHECK_LAUNCH("mg_upsample_lengths");
    return MG_OK;
}

int mg_upsample_index(const int64_t* dur, int B, int P, int t_cap, int64_t* idx64, int32_t* rows32, void* stream) {
    MG_CHECK_ARG(dur && B > 0 && P > 0 && t_cap >= 0, "mg_upsample_index: bad shape B=%d P=%d t_cap=%d", B, P, t_cap);
    MG_CHECK_ARG(P <= MG_MAX_PHONES, "mg_upsample_index: P=%d exceeds %d phones per utterance", P, MG_MAX_PHONES);
    MG_CHECK_ARG((int64_t)B * P < 2147483647LL, "mg_upsample_index: B*P overflows int32 row ids");
    if (t_cap == 0) return MG_OK;
    const size_t lds = (size_t)(256 + P) * sizeof(int);
    hipLaunchKernelGGL(upsample_index_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, dur, P, t_cap, idx64, rows32, nullptr, 0,
                       nullptr, nullptr);
    MG_CHECK_LAUNCH("mg_upsample_index");
    return MG_OK;
}

// mg_upsample_index + what the phone-rate step needs from mg_segment_bounds, in the same launch: rows_mapped (-1 -> pad_row) and the
// frame run [seg_start, seg_end) of every phone row (frame ids b * t_cap + t).
int mg_upsample_index_maps(const int64_t* dur, int B, int P, int t_cap, int32_t* rows32, int32_t* rows_mapped, int pad_row,
                           int32_t* seg_start, int32_t* seg_end, void* stream) {
    MG_CHECK_ARG(dur && rows32 && rows_mapped && seg_start && seg_end && B > 0 && P > 0 && t_cap > 0,
                 "mg_upsample_index_maps: bad arguments (B=%d P=%d t_cap=%d)", B, P, t_cap);
    MG_CHECK_ARG(P <= MG_MAX_PHONES, "mg_upsample_index_maps: P=%d exceeds %d phones per utterance", P, MG_MAX_PHONES);
    MG_CHECK_ARG((int64_t)B * P < 2147483647LL && (int64_t)B * t_cap < 2147483647LL, "mg_upsample_index_maps: int32 ids overflow");
    const size_t lds = (size_t)(256 + P) * sizeof(int);
    hipLaunchKernelGGL(upsample_index_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, dur, P, t_cap, nullptr, rows32, rows_mapped,
                       pad_row, seg_start, seg_end);
    MG_CHECK_LAUNCH("mg_upsample_index_maps");
    return MG_OK;
}

static int gather_grid(int64_t M) {
    int64_t blocks = mg_ceil_div(M, 4);
    if (blocks > 8192) blocks = 8192;
    return (int)(blocks < 1 ? 1 : blocks);
}

int mg_segment_index(const int64_t* seg_lens, int B, int S, int T, int L, int32_t* split, int32_t* ends, void* stream) {
    MG_CHECK_ARG(seg_lens && B > 0 && S > 0 && T >= 0 && L >= 0, "mg_segment_index: bad arguments (B=%d S=%d T=%d L=%d)", B, S, T, L);
    MG_CHECK_ARG(S <= MG_MAX_PHONES, "mg_segment_index: S=%d exceeds %d segments per sequence", S, MG_MAX_PHONES);
    MG_CHECK_ARG((int64_t)B * T < 2147483647LL, "mg_segment_index: B*T overflows int32 row ids");
    if (!split && !ends) return MG_OK;
    const size_t lds = (size_t)(256 + S) * sizeof(int);
    hipLaunchKernelGGL(segment_index_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, seg_lens, S, T, L, L > 0 ? split : nullptr, ends);
    MG_CHECK_LAUNCH("mg_segment_index");
    return MG_OK;
}

int mg_frame_layout(const int64_t* seq_len, int B, int T, int64_t total, int32_t* offsets, int32_t* rows, int32_t* inverse,
                    void* stream) {
    MG_CHECK_ARG(seq_len && offsets && rows && inverse && B > 0 && T > 0 && total >= 0, "mg_frame_layout: bad arguments (B=%d T=%d total=%lld)", B, T,
                 (long long)total);
    MG_CHECK_ARG((int64_t)B * T < 2147483647LL && total <= (int64_t)B * T, "mg_frame_layout: B*T=%lld overflows int32 row ids or total=%lld exceeds it",
                 (long long)B * T, (long long)total);
    hipLaunchKernelGGL(frame_layout_scan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, seq_len, B, T, offsets);
    hipLaunchKernelGGL(frame_layout_fill_kernel, dim3((unsigned)mg_ceil_div((int64_t)B * T, 256)), dim3(256), 0, (hipStream_t)stream, seq_len,
                       offsets, B, T, (int)total, rows, inverse);
    MG_CHECK_LAUNCH("mg_frame_layout");
    return MG_OK;
}

size_t mg_pad_rows_colsum_workspace_bytes(int B, int T, int D) {
    if (B <= 0 || T <= 0 || D <= 0) return 0;
    return (size_t)B * (size_t)mg_ceil_div(T, PAD_CHUNK) * (size_t)D * sizeof(float);
}

int mg_pad_rows_colsum_f32(const float* g, const int64_t* seq_len, int B, int T, int D, float* out, void* workspace, size_t workspace_bytes,
                           void* stream) {
    MG_CHECK_ARG(g && seq_len && out && workspace && B > 0 && T > 0 && D > 0 && D <= 4096, "mg_pad_rows_colsum_f32: bad arguments (B=%d T=%d D=%d)", B, T, D);
    MG_CHECK_ARG(workspace_bytes >= mg_pad_rows_colsum_workspace_bytes(B, T, D), "mg_pad_rows_colsum_f32: workspace too small");
    MG_CHECK_ARG(B <= 65535, "mg_pad_rows_colsum_f32: B=%d exceeds the grid's y extent", B);
    const int n_chunks = (int)mg_ceil_div(T, PAD_CHUNK);
    hipLaunchKernelGGL(pad_rows_colsum_stage1_kernel, dim3(n_chunks, B), dim3(256), (size_t)4 * D * sizeof(float), (hipStream_t)stream, g, seq_len, T,
                       D, n_chunks, (float*)workspace);
    hipLaunchKernelGGL(pad_rows_colsum_stage2_kernel, dim3((unsigned)mg_ceil_div(D, 64)), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace,
                       B * n_chunks, D, out);
    MG_CHECK_LAUNCH("mg_pad_rows_colsum_f32");
    return MG_OK;
}

int mg_scatter_rows_f32(const float* src, const int32_t* rows, float* dst, int64_t M, int F, void* stream) {
    MG_CHECK_ARG(src && rows && dst && M >= 0 && F > 0, "mg_scatter_rows_f32: bad arguments (M=%lld F=%d)", (long long)M, F);
    if (M == 0) return MG_OK;
    hipLaunchKernelGGL(scatter_rows_f32_kernel, dim3(gather_grid(M)), dim3(256), 0, (hipStream_t)stream, src, rows, dst, M, F);
    MG_CHECK_LAUNCH("mg_scatter_rows_f32");
    return MG_OK;
}

int mg_gather_rows_f32(const float* src, const int32_t* rows, float* out, int64_t M, int F, void* stream) {
    MG_CHECK_ARG(src && out && M >= 0 && F > 0, "mg_gather_rows_f32: bad arguments (M=%lld F=%d)", (long long)M, F);
    if (M == 0) return MG_OK;
    const bool vec = (F % 4 == 0) && (((uintptr_t)src | (uintptr_t)out) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(gather_rows_f32_kernel<true>, dim3(gather_grid(M)), dim3(256), 0, (hipStream_t)stream, src, rows, out, M, F);
    else
        hipLaunchKernelGGL(gather_rows_f32_kernel<false>, dim3(gather_grid(M)), dim3(256), 0, (hipStream_t)stream, src, rows, out, M, F);
    MG_CHECK_LAUNCH("mg_gather_rows_f32");
    return MG_OK;
}

int mg_gather_rows_bf16(const float* src, const int32_t* rows, uint16_t* out, int64_t M, int F, int ldo, void* stream) {
    MG_CHECK_ARG(src && out && M >= 0 && F > 0, "mg_gather_rows_bf16: bad arguments (M=%lld F=%d)", (long long)M, F);
    MG_CHECK_ARG(ldo >= F && ldo % 8 == 0, "mg_gather_rows_bf16: ldo=%d must be >= F=%d and a multiple of 8", ldo, F);
    MG_CHECK_ARG(((uintptr_t)out % 16 == 0) && ((uintptr_t)src % 16 == 0), "mg_gather_rows_bf16: buffers must be 16-byte aligned");
    if (M == 0) return MG_OK;
    hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(gather_grid(M)), dim3(256), 0, (hipStream_t)stream, src, rows, out, M, F, ldo);
    MG_CHECK_LAUNCH("mg_gather_rows_bf16");
    return MG_OK;
}

int mg_gather_concat_f32(const float* src, const int32_t* rows, const float* extra, float* out, int64_t M, int F, int C,
                         int ldo, void* stream) {
    MG_CHECK_ARG(src && extra && out && M >= 0 && F > 0 && C > 0, "mg_gather_concat_f32: bad arguments (M=%lld F=%d C=%d)", (long long)M, F, C);
    MG_CHECK_ARG(ldo >= F + C, "mg_gather_concat_f32: ldo=%d must be >= F+C=%d", ldo, F + C);
    if (M == 0) return MG_OK;
    hipLaunchKernelGGL(gather_concat_kernel<false>, dim3(gather_grid(M)), dim3(256), 0, (hipStream_t)stream, src, rows, extra, (void*)out, M, F, C, ldo);
    MG_CHECK_LAUNCH("mg_gather_concat_f32");
    return MG_OK;
}

int mg_gather_concat_bf16(const float* src, const int32_t* rows, const float* extra, uint16_t* out, int64_t M, int F, int C,
                          int ldo, void* stream) {
    MG_CHECK_ARG(src && extra && out && M >= 0 && F > 0 && C > 0, "mg_gather_concat_bf16: bad arguments (M=%lld F=%d C=%d)", (long long)M, F, C);
    MG_CHECK_ARG(ldo >= F + C && ldo % 8 == 0, "mg_gather_concat_bf16: ldo=%d must be >= F+C=%d and a multiple of 8", ldo, F + C);
    MG_CHECK_ARG(((uintptr_t)out % 16 == 0) && ((uintptr_t)src % 16 == 0), "mg_gather_concat_bf16: buffers must be 16-byte aligned");
    if (M == 0) return MG_OK;
    hipLaunchKernelGGL(gather_concat_kernel<true>, dim3(gather_grid(M)), dim3(256), 0, (hipStream_t)stream, src, rows, extra, (void*)out, M, F, C, ldo);
    MG_CHECK_LAUNCH("mg_gather_concat_bf16");
    return MG_OK;
}

int mg_upsample_backward_f32(const float* grad_out, const int64_t* dur, float* grad_src, int B, int P, int T, int F,
                             void* stream) {
    MG_CHECK_ARG(grad_out && dur && grad_src && B > 0 && P > 0 && T >= 0 && F > 0,
                 "mg_upsample_backward_f32: bad arguments (B=%d P=%d T=%d F=%d)", B, P, T, F);
    MG_CHECK_ARG(P <= MG_MAX_PHONES, "mg_upsample_backward_f32: P=%d exceeds %d", P, MG_MAX_PHONES);
    int chunks = (int)mg_ceil_div(P, 8);
    if (chunks > 16) chunks = 16;
    if (chunks < 1) chunks = 1;
    const size_t lds = (size_t)(256 + P) * sizeof(int);
    hipLaunchKernelGGL(upsample_backward_kernel, dim3(B, chunks), dim3(256), lds, (hipStream_t)stream, grad_out, dur, grad_src, P, T, F);
    MG_CHECK_LAUNCH("mg_upsample_backward_f32");
    return MG_OK;
}

}  // extern "C"
