// Phone-rate first layer.  The model input of the README F0Model and of the RNN_SPSS layout is
// upsample_to_repetitions(normalised_lab, dur) (morgana/utils.py:175-228): every phone row repeated dur[b, p] times, then
// Linear(600, 512) + Sigmoid over all B*T frame rows (README.rst:65-73, morgana/utils.py:401-418).  A Linear commutes with
// repeating rows:  gather(X) W^T = gather(X W^T).  So the 600-wide product is done ONCE PER PHONE (B*P rows, 12.5x fewer than
// frames at the synthetic 12.5 frames per phone) and the frame-rate activation is a gather of that table:
//     forward   H_table = sigmoid(X_phone W^T + b) with the existing GEMM on B*P (+ padding) rows; the NEXT layer's GEMM gathers
//               rows of the table through the row map (segment_bounds_kernel writes it with -1 -> the table's zero-input row)
//     backward  S[r] = sum over the frames f of phone r of dZ_next[f]  (segment_sum_kernel); sigma'(H) is constant over a phone's
//               frames, so dW_next = S^T H_table, dZ = (S W_next) * H_table (1 - H_table), dW = dZ^T X_phone all run on table rows
// Per frame row the fp32 dot product, the bias add, the sigmoid and the bf16 rounding are the same operations in the same
// order as in the frame-rate GEMM epilogue, so the activations are unchanged.  segment_sum is HBM bound (reads the M x N
// gradient once).  Frames with row -1 (padding past an utterance's end, the reference's zero row, utils.py:206-214) use the
// table rows behind the R phone rows (zero inputs: sigmoid(b)); their gradients are summed into those `extra` rows, so dW
// ignores them (their input is zero) and the bias gradients still see every frame.
#include "common.h"
#include "phone_front.h"
#include "expand_reduce.h"

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

// Per table row: the frames it stands for, as the masked MSE sees them (morgana/losses.py:29-51: frame weight w_f = [t < n_b] / (n_b B)).
// weight[r] = sum_f w_f, ybar[r] = sum_f w_f y_f / weight[r]; then  sum_f w_f (p - y_f)^2 = weight[r] (p - ybar[r])^2 + c_r  for
// any prediction p shared by the row's frames, c_r = sum_f w_f (y_f - ybar[r])^2 (two passes over the row's frames).  The c_r are
// summed per block into `partial`.  Blocks [0, phone_blocks): 16 lanes per phone row (its ~12 consecutive frames).  Blocks behind
// them: one WAVE per extra row, which takes the padding frames of its share of the frame axis, 64 frames per look.

__global__ __launch_bounds__(256) void phone_target_stats_kernel(const float* __restrict__ target, const int32_t* __restrict__ rows,
                                                                 int64_t M, const int32_t* __restrict__ seg_start,
                                                                 const int32_t* __restrict__ seg_end,
                                                                 const int64_t* __restrict__ seq_len, int B, int T, int R, int extra,
                                                                 int phone_blocks, float* __restrict__ ybar,
                                                                 float* __restrict__ weight, float* __restrict__ partial) {
    __shared__ float red[256];
    float c = 0.f;
    if ((int)blockIdx.x < phone_blocks) {
        // 16 lanes per phone row: its frames lie in one utterance, so the weight of a live frame is one value per row
        const int r = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
        float w_sum = 0.f, wy = 0.f, inv = 0.f;
        int lo = 0, hi = 0, t0 = 0, nb = 0;
        if (r < R) {
            lo = seg_start[r];
            hi = seg_end[r];
            const int b = lo / T;
            t0 = lo - b * T;
            int64_t n = seq_len ? seq_len[b] : (int64_t)T;
            n = n > T ? T : (n < 0 ? 0 : n);
            nb = (int)n;
            inv = 1.f / ((float)nb * (float)B);                  // n_b == 0 -> inf; 0 * inf = NaN below, as the reference
            for (int f = lo + sub; f < hi; f += 16) {
                const float w = (t0 + (f - lo) < nb ? 1.f : 0.f) * inv;
                w_sum += w;
                wy += w * target[f];
            }
        }
#pragma unroll
        for (int d = 8; d > 0; d >>= 1) {
            w_sum += __shfl_xor(w_sum, d, 64);
            wy += __shfl_xor(wy, d, 64);
        }
        const float mean = w_sum > 0.f ? wy / w_sum : 0.f;
        for (int f = lo + sub; f < hi; f += 16) {
            const float d = target[f] - mean;
            c += ((t0 + (f - lo) < nb ? 1.f : 0.f) * inv) * d * d;
        }
        if (r < R && sub == 0) {
            ybar[r] = mean;
            weight[r] = w_sum;
        }
    } else {
        const int lane = threadIdx.x & 63;
        const int j = ((int)blockIdx.x - phone_blocks) * 4 + (threadIdx.x >> 6);        // extra row of this wave
        if (j < extra) {
            const int64_t chunk = (M + extra - 1) / extra, lo = (int64_t)j * chunk, hi = lo + chunk < M ? lo + chunk : M;
            float w_sum = 0.f, wy = 0.f;
            for (int64_t f = lo + lane; f < hi; f += 64) {
                const int rf = rows[f];
                if (rf < 0 || rf >= R) {
                    const float w = pf_frame_weight(f, seq_len, B, T);
                    w_sum += w;
                    wy += w * target[f];
                }
            }
            w_sum = mg_wave_sum(w_sum);
            wy = mg_wave_sum(wy);
            const float mean = w_sum > 0.f ? wy / w_sum : 0.f;
            for (int64_t f = lo + lane; f < hi; f += 64) {
                const int rf = rows[f];
                if (rf < 0 || rf >= R) {
                    const float d = target[f] - mean;
                    c += pf_frame_weight(f, seq_len, B, T) * d * d;
                }
            }
            if (lane == 0) {
                ybar[R + j] = mean;
                weight[R + j] = w_sum;
            }
        }
    }
    red[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// out[f] = table[rows[f]] for a one-column table (the repeated prediction); rows >= 0.  With `partial` the LAST block also adds the
// per-block partial sums of the loss's constant term to `loss` in place (a fixed-order sum by one block: deterministic) - the
// kernel runs after the tail wrote the loss, and this saves a launch node of its own.
__global__ __launch_bounds__(256) void expand_column_kernel(const float* __restrict__ table, const int32_t* __restrict__ rows, int64_t M,
                                                            float* __restrict__ out, const float* __restrict__ partial, int n_partial,
                                                            float* __restrict__ loss) {
    __shared__ float red[256];
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f < M) out[f] = table[rows[f]];
    if (partial && blockIdx.x == gridDim.x - 1) {
        float v = 0.f;
        for (int i = threadIdx.x; i < n_partial; i += 256) v += partial[i];
        red[threadIdx.x] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss[0] += red[0];
    }
}

// expand_column_kernel and the ordered slab reduce of the fused tail in ONE launch (two nodes at the ~5 us launch floor become one):
// block b repeats the prediction for its 256 frames and, while b < ceil(n / 16), reduces elements [16 b, 16 b + 16) of the S slabs with
// mg_slab_reduce_kernel's arithmetic (16 interleaved partitions, each ascending, then ascending over the partitions: the same bits).
// The block that owns the LAST element (the loss, stored behind the gradients) adds the sum of the per-block partial sums of the
// loss's constant term to it, as the last block of expand_column_kernel did after the reduce launch.
__global__ __launch_bounds__(256) void expand_reduce_kernel(ExpandReduceArgs a) {
    __shared__ float part[16][17];
    __shared__ float red[256];
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f < a.M) a.out[f] = a.table[a.rows[f]];
    mg_tail_chunk_reduce(a, (int64_t)blockIdx.x * 16, threadIdx.x, true, part, red);       // expand_reduce.h: the riders' arithmetic
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int n, float* __restrict__ out,
                                                           int accumulate) {
    __shared__ float red[256];
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + red[0] : red[0];
}

// The frame map and the statistics in one launch: one job per utterance / per four extra rows (phone_front.h)
__global__ __launch_bounds__(256) void phone_front_kernel(PhoneFrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) int pf_lds[];
    phone_front_block<256>(a, blockIdx.x, gridDim.x, pf_lds);
}

extern "C" {

int mg_segment_bounds(const int32_t* rows, int64_t M, int R, int32_t* seg_start, int32_t* seg_end, int32_t* rows_mapped, int pad_row,
                      void* stream) {
    MG_CHECK_ARG(rows && seg_start && seg_end && M > 0 && R > 0 && M < 2147483647LL, "mg_segment_bounds: bad arguments (M=%lld R=%d)",
                 (long long)M, R);
    hipStream_t st = (hipStream_t)stream;
    const bool joined = seg_end == seg_start + R;                   // one (2, R) buffer: one memset node
    if (hipMemsetAsync(seg_start, 0, (size_t)R * sizeof(int32_t) * (joined ? 2 : 1), st) != hipSuccess ||
        (!joined && hipMemsetAsync(seg_end, 0, (size_t)R * sizeof(int32_t), st) != hipSuccess)) {
        mg_set_error("mg_segment_bounds: memset failed");
        return MG_ELAUNCH;
    }
    hipLaunchKernelGGL(segment_bounds_kernel, dim3((unsigned)mg_ceil_div(M, 256)), dim3(256), 0, st, rows, M, R, seg_start, seg_end,
                       rows_mapped, pad_row);
    MG_CHECK_LAUNCH("mg_segment_bounds");
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

size_t mg_phone_target_stats_workspace_bytes(int R, int extra) {
    return (size_t)(mg_ceil_div(R, 16) + mg_ceil_div(extra, 4)) * sizeof(float);
}

int mg_phone_target_stats(const float* target, const int32_t* rows, int64_t M, const int32_t* seg_start, const int32_t* seg_end,
                          const int64_t* seq_len, int B, int T, int R, int extra, float* ybar, float* weight, float* loss_const,
                          void* workspace, size_t workspace_bytes, void* stream) {
    MG_CHECK_ARG(target && rows && seg_start && seg_end && ybar && weight && B > 0 && T > 0 && R > 0 && extra >= 0 &&
                     M == (int64_t)B * T,
                 "mg_phone_target_stats: bad arguments (M=%lld B=%d T=%d R=%d extra=%d)", (long long)M, B, T, R, extra);
    if (!workspace || workspace_bytes < mg_phone_target_stats_workspace_bytes(R, extra)) {
        mg_set_error("mg_phone_target_stats: workspace of %zu bytes needed, got %zu", mg_phone_target_stats_workspace_bytes(R, extra),
                     workspace_bytes);
        return MG_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int phone_blocks = (int)mg_ceil_div(R, 16), blocks = phone_blocks + (int)mg_ceil_div(extra, 4);
    hipLaunchKernelGGL(phone_target_stats_kernel, dim3(blocks), dim3(256), 0, st, target, rows, M, seg_start, seg_end, seq_len, B, T, R, extra,
                       phone_blocks, ybar, weight, (float*)workspace);
    if (loss_const) hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, blocks, loss_const, 0);
    MG_CHECK_LAUNCH("mg_phone_target_stats");
    return MG_OK;
}

// mg_upsample_index_maps + mg_phone_target_stats as one launch (phone_front.h).  The same rows32 / rows_mapped / seg_start / seg_end /
// ybar / weight bit for bit; the workspace holds the constant's partial sums in another grouping (one per utterance instead of one per
// 16 phone rows: the consumers sum all of its ceil(R / 16) + ceil(extra / 4) slots either way).
int mg_phone_front_check(const int64_t* dur, int B, int P, int T, const float* target, int extra, const int32_t* rows32,
                         const int32_t* rows_mapped, const int32_t* seg_start, const int32_t* seg_end, const float* ybar, const float* weight,
                         const void* workspace, size_t workspace_bytes, const char* who) {
    if (!(dur && target && rows32 && rows_mapped && seg_start && seg_end && ybar && weight && B > 0 && P > 0 && T > 0 && extra >= 0)) {
        mg_set_error("%s: bad arguments (B=%d P=%d T=%d extra=%d)", who, B, P, T, extra);
        return MG_EINVAL;
    }
    if (P > 12288 || (int64_t)B * P >= 2147483647LL || (int64_t)B * T >= 2147483647LL) {
        mg_set_error("%s: P=%d exceeds 12288 phones per utterance or int32 ids overflow", who, P);
        return MG_EINVAL;
    }
    if ((int64_t)B > mg_ceil_div((int64_t)B * P, 16)) {
        mg_set_error("%s: B=%d utterances of P=%d phones leave no partial-sum slot per utterance (use the two launches)", who, B, P);
        return MG_EINVAL;
    }
    if (phone_front_lds_ints(B, P, T, extra) > 16000) {
        mg_set_error("%s: extra=%d rows over %lld frames span too many utterances per job (use the two launches)", who, extra, (long long)B * T);
        return MG_EINVAL;
    }
    if (!workspace || workspace_bytes < mg_phone_target_stats_workspace_bytes(B * P, extra)) {
        mg_set_error("%s: workspace of %zu bytes needed, got %zu", who, mg_phone_target_stats_workspace_bytes(B * P, extra), workspace_bytes);
        return MG_EWORKSPACE;
    }
    return MG_OK;
}

int mg_phone_front(const int64_t* dur, int B, int P, int T, const float* target, const int64_t* seq_len, int extra, int32_t* rows32,
                   int32_t* rows_mapped, int pad_row, int32_t* seg_start, int32_t* seg_end, float* ybar, float* weight, void* workspace,
                   size_t workspace_bytes, void* stream) {
    const int rc = mg_phone_front_check(dur, B, P, T, target, extra, rows32, rows_mapped, seg_start, seg_end, ybar, weight, workspace,
                                        workspace_bytes, "mg_phone_front");
    if (rc != MG_OK) return rc;
    PhoneFrontArgs a{dur, target, seq_len, B, P, T, extra, rows32, rows_mapped, pad_row, seg_start, seg_end, ybar, weight, (float*)workspace, 0, 0};
    const int64_t ints = phone_front_lds_ints(B, P, T, extra);
    a.lds_ints = (int)ints;
    hipLaunchKernelGGL(phone_front_kernel, dim3((unsigned)phone_front_jobs(B, extra)), dim3(256), (size_t)ints * sizeof(int), (hipStream_t)stream,
                       a);
    MG_CHECK_LAUNCH("mg_phone_front");
    return MG_OK;
}

// loss += the constant term of mg_phone_target_stats (its per-block partial sums still in `workspace`), after the tail wrote loss
int mg_phone_loss_const_add(const void* workspace, int R, int extra, float* loss, void* stream) {
    MG_CHECK_ARG(workspace && loss && R > 0 && extra >= 0, "mg_phone_loss_const_add: bad arguments");
    const int blocks = (int)(mg_ceil_div(R, 16) + mg_ceil_div(extra, 4));
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, blocks, loss, 1);
    MG_CHECK_LAUNCH("mg_phone_loss_const_add");
    return MG_OK;
}

int mg_expand_column_f32(const float* table, const int32_t* rows, int64_t M, float* out, void* stream) {
    MG_CHECK_ARG(table && rows && out && M > 0, "mg_expand_column_f32: bad arguments (M=%lld)", (long long)M);
    hipLaunchKernelGGL(expand_column_kernel, dim3((unsigned)mg_ceil_div(M, 256)), dim3(256), 0, (hipStream_t)stream, table, rows, M, out,
                       nullptr, 0, nullptr);
    MG_CHECK_LAUNCH("mg_expand_column_f32");
    return MG_OK;
}

// mg_expand_column_f32 and mg_phone_loss_const_add in one launch (the phone-rate step's last two nodes of the forward)
int mg_expand_column_loss_f32(const float* table, const int32_t* rows, int64_t M, float* out, const void* stats_workspace, int R,
                              int extra, float* loss, void* stream) {
    MG_CHECK_ARG(table && rows && out && stats_workspace && loss && M > 0 && R > 0 && extra >= 0,
                 "mg_expand_column_loss_f32: bad arguments (M=%lld)", (long long)M);
    const int n_partial = (int)(mg_ceil_div(R, 16) + mg_ceil_div(extra, 4));
    hipLaunchKernelGGL(expand_column_kernel, dim3((unsigned)mg_ceil_div(M, 256)), dim3(256), 0, (hipStream_t)stream, table, rows, M, out,
                       (const float*)stats_workspace, n_partial, loss);
    MG_CHECK_LAUNCH("mg_expand_column_loss_f32");
    return MG_OK;
}

// mg_expand_column_loss_f32 and the tail's slab reduce in one launch: dst[0 .. n) = ordered sum of the S slabs (mg_f0_l2tail_rows_slabs_bf16
// leaves them), dst[n - 1] (the loss) + the loss's constant term, out = the repeated prediction.
int mg_expand_column_reduce_f32(const float* table, const int32_t* rows, int64_t M, float* out, const void* stats_workspace, int R,
                                int extra, const float* slab, int64_t n, int64_t stride, int S, float* dst, void* stream) {
    MG_CHECK_ARG(table && rows && out && stats_workspace && slab && dst && M > 0 && R > 0 && extra >= 0 && n > 0 && stride >= n && S > 0,
                 "mg_expand_column_reduce_f32: bad arguments (M=%lld n=%lld S=%d)", (long long)M, (long long)n, S);
    const int n_partial = (int)(mg_ceil_div(R, 16) + mg_ceil_div(extra, 4));
    int64_t blocks = mg_ceil_div(M, 256);
    if (mg_ceil_div(n, 16) > blocks) blocks = mg_ceil_div(n, 16);
    const ExpandReduceArgs xa{table, rows, M, out, (const float*)stats_workspace, n_partial, slab, n, stride, S, dst, 0};
    hipLaunchKernelGGL(expand_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, xa);
    MG_CHECK_LAUNCH("mg_expand_column_reduce_f32");
    return MG_OK;
}

}  // extern "C"
