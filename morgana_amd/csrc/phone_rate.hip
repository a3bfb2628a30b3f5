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

// mg_segment_sum of a bf16 gradient and, in the same pass over G, the weight gradient of C per-frame input features (the frame
// counters behind the repeated phone rows, models/RNN_SPSS.py:76-81):  slab[b][c][n] = sum over workgroup b's frames of
// G[f, n] feat[f, c];  feat_wgrad_reduce_kernel sums the slabs in index order into dW[n, col0 + c].  A wave takes a contiguous
// range of table rows (their frames are consecutive), 8 columns per lane; the C x 8 products per frame ride on the row that the
// sum loads anyway.  Deterministic: fixed row ranges, the waves of a workgroup added in wave order, the slabs in slab order.
#define SSF_WAVES 8
#define SSF_BLOCKS 256
__global__ __launch_bounds__(64 * SSF_WAVES) void segment_sum_feat_kernel(const uint16_t* __restrict__ G, int ldg, const int32_t* __restrict__ rows,
                                                                          int64_t M, const int32_t* __restrict__ seg_start,
                                                                          const int32_t* __restrict__ seg_end, int R, int extra, int N,
                                                                          uint16_t* __restrict__ out, int ldo, const float* __restrict__ feat,
                                                                          int C, float* __restrict__ slab) {
    __shared__ float red[16 * 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = (blockIdx.y * 64 + lane) * 8;
    const bool live = c < N;
    float cacc[16][8];
#pragma unroll
    for (int cc = 0; cc < 16; ++cc)
#pragma unroll
        for (int e = 0; e < 8; ++e) cacc[cc][e] = 0.f;
    const int n_rows = R + extra;
    const int per = (n_rows + (int)gridDim.x * SSF_WAVES - 1) / ((int)gridDim.x * SSF_WAVES);
    const int r_lo = ((int)blockIdx.x * SSF_WAVES + wave) * per, r_hi = r_lo + per < n_rows ? r_lo + per : n_rows;

    // one frame: its 8 gradient columns into the row's sum and, times each feature, into the feature gradients
    auto frame = [&](int64_t f, float (&acc)[8]) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = 0.f;
        const float fv = lane < C ? feat[f * C + lane] : 0.f;
        if (live) PrLoad<uint16_t>::add(G + (size_t)f * ldg + c, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += t[e];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fv), cc));
#pragma unroll
            for (int e = 0; e < 8; ++e) cacc[cc][e] = fmaf(x, t[e], cacc[cc][e]);
        }
    };

    for (int r = r_lo; r < r_hi; ++r) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (r < R) {
            const int f0 = seg_start[r], f1 = seg_end[r];
#pragma unroll 4
            for (int f = f0; f < f1; ++f) frame(f, acc);
        } else {
            // padding frames of this row's share of the frame axis, as segment_sum_kernel finds them
            const int64_t chunk = (M + extra - 1) / extra, lo = (int64_t)(r - R) * chunk, hi = lo + chunk < M ? lo + chunk : M;
            for (int64_t base = lo; base < hi; base += 64) {
                const int64_t mine = base + lane;
                unsigned long long pads = __ballot(mine < hi && (rows[mine] < 0 || rows[mine] >= R));
                while (pads) {
                    const int bit = __builtin_ctzll(pads);
                    pads &= pads - 1;
                    frame(base + bit, acc);
                }
            }
        }
        if (c < ldo) PrStore<uint16_t>::put(out + (size_t)r * ldo + c, acc);
    }
    // the workgroup's waves, added in wave order
    for (int wv = 0; wv < SSF_WAVES; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) {
                if (cc < C) {
                    float* dst = red + cc * 512 + lane * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) dst[e] = (wv == 0 ? 0.f : dst[e]) + cacc[cc][e];
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < C * 512; i += 64 * SSF_WAVES) {
        const int cc = i >> 9, col = blockIdx.y * 512 + (i & 511);
        if (col < ldo) slab[((size_t)blockIdx.x * C + cc) * ldo + col] = red[i];
    }
}

// dW[n, col0 + c] (+)= sum_b slab[b][c][n], b in index order.  blockIdx.x = 64 columns n, blockIdx.y = c; thread (q, n): the slabs
// b = q mod 4, partial sums added in q order.
__global__ __launch_bounds__(256) void feat_wgrad_reduce_kernel(const float* __restrict__ slab, int n_slabs, int C, int ldo, int N,
                                                                float* __restrict__ dW, int ldw, int col0, int accumulate) {
    __shared__ float part[4][64];
    const int q = threadIdx.x >> 6, nl = threadIdx.x & 63, n = blockIdx.x * 64 + nl, c = blockIdx.y;
    float s = 0.f;
    if (n < N)
        for (int b = q; b < n_slabs; b += 4) s += slab[((size_t)b * C + c) * ldo + n];
    part[q][nl] = s;
    __syncthreads();
    if (q == 0 && n < N) {
        const float total = ((part[0][nl] + part[1][nl]) + part[2][nl]) + part[3][nl];
        float* dst = dW + (size_t)n * ldw + col0 + c;
        *dst = accumulate ? *dst + total : total;
    }
}

// The first Linear of a model whose input is cat(upsample_to_repetitions(lab, durations), counters) (models/RNN_SPSS.py:76-81,
// models/f0_test_model.py:78-79): W = [W_lab | W_cnt], so z[f] = (lab W_lab^T)[phone(f)] + counters[f] W_cnt^T + b.  The first
// product runs once per phone (mg_linear_fwd_bf16 on the table, f32 output, no bias); this kernel adds the per-frame part:
//   Y[f, n] = act(P[rows[f], n] + sum_c feat[f, c] W[n, col0 + c] + bias[n])     (bf16 or f32, the padding columns of Y zero)
// A wave takes chunks of 16 CONSECUTIVE frames, COLS columns per lane (blockIdx.y = chunk of 64 COLS columns): a phone's frames
// follow each other, so the wave fetches a row of P only where the row id changes (~ once per 11 frames) and keeps it in
// registers; the lane's C x COLS counter weights come through LDS once per workgroup (W's rows are 4 ldw bytes apart: read per
// lane they cost a cache line each, per wave) and stay in registers.  Write bound: 4 N (f32) or 2 N bytes per frame out.
#define PCL_CHUNK 16
template <int COLS, typename OutT> struct PclStore;
template <> struct PclStore<8, uint16_t> { static __device__ __forceinline__ void put(uint16_t* d, const float (&v)[8]) { PrStore<uint16_t>::put(d, v); } };
template <> struct PclStore<8, float> { static __device__ __forceinline__ void put(float* d, const float (&v)[8]) { PrStore<float>::put(d, v); } };
template <> struct PclStore<4, uint16_t> {
    static __device__ __forceinline__ void put(uint16_t* d, const float (&v)[4]) {
        uint2 pk;
        pk.x = (uint32_t)mg_f2bf(v[0]) | ((uint32_t)mg_f2bf(v[1]) << 16);
        pk.y = (uint32_t)mg_f2bf(v[2]) | ((uint32_t)mg_f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(d) = pk;
    }
};
template <> struct PclStore<4, float> {
    static __device__ __forceinline__ void put(float* d, const float (&v)[4]) { *reinterpret_cast<f32x4*>(d) = f32x4{v[0], v[1], v[2], v[3]}; }
};

template <int ACT, typename OutT, int COLS>
__global__ __launch_bounds__(256) void phone_concat_layer_kernel(const float* __restrict__ P, int ldp, const int32_t* __restrict__ rows,
                                                                 int64_t M, const float* __restrict__ feat, int C,
                                                                 const float* __restrict__ W, int ldw, int col0,
                                                                 const float* __restrict__ bias, int N, OutT* __restrict__ Y, int ldy) {
    __shared__ float wl[64 * COLS * 16];                     // [this block's 64 COLS columns][C]
    const int tid = threadIdx.x, lane = tid & 63;
    const int cbase = blockIdx.y * 64 * COLS;
    for (int i = tid; i < 64 * COLS * C; i += 256) {
        const int n = cbase + i / C;
        wl[i] = n < N ? W[(size_t)n * ldw + col0 + i % C] : 0.f;
    }
    __syncthreads();
    const int c0 = cbase + lane * COLS;
    const bool live = c0 < ldy;                             // this lane writes Y[f, c0 .. c0 + COLS); columns >= N get zeros
    const bool has_p = c0 < N;                              // ldp >= N rounded up to 8 (and COLS divides 8): the lane's columns exist in P
    float wc[16][COLS], bv[COLS];
#pragma unroll
    for (int e = 0; e < COLS; ++e) {
        bv[e] = (c0 + e < N && bias) ? bias[c0 + e] : 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) wc[c][e] = c < C ? wl[(lane * COLS + e) * C + c] : 0.f;
    }
    const int64_t chunks = (M + PCL_CHUNK - 1) / PCL_CHUNK;
    for (int64_t ch = (int64_t)blockIdx.x * 4 + (tid >> 6); ch < chunks; ch += (int64_t)gridDim.x * 4) {
        const int64_t f0 = ch * PCL_CHUNK;
        const int rv = (lane < PCL_CHUNK && f0 + lane < M) ? rows[f0 + lane] : -1;
        int prev = -1;
        float pz[COLS];
#pragma unroll
        for (int e = 0; e < COLS; ++e) pz[e] = bv[e];
#pragma unroll
        for (int j = 0; j < PCL_CHUNK; j += 4) {
            float fv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) fv[u] = (f0 + j + u < M && lane < C) ? feat[(f0 + j + u) * C + lane] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t f = f0 + j + u;
                if (f < M) {                                 // wave-uniform
                    const int row = __builtin_amdgcn_readlane(rv, j + u);
                    if (row != prev) {                       // wave-uniform: the phone changed
                        prev = row;
#pragma unroll
                        for (int e = 0; e < COLS; ++e) pz[e] = bv[e];
                        if (has_p) {
                            const float* src = P + (size_t)row * ldp + c0;
#pragma unroll
                            for (int q = 0; q < COLS / 4; ++q) {
                                const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * q);
#pragma unroll
                                for (int e = 0; e < 4; ++e) pz[4 * q + e] += v[e];
                            }
                        }
                    }
                    float z[COLS];
#pragma unroll
                    for (int e = 0; e < COLS; ++e) z[e] = pz[e];
                    // the counters after the table part, in column order: the order the frame-rate GEMM's k loop meets them
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(fv[u]), c));
#pragma unroll
                        for (int e = 0; e < COLS; ++e) z[e] = fmaf(x, wc[c][e], z[e]);
                    }
                    if (live) {
#pragma unroll
                        for (int e = 0; e < COLS; ++e) {
                            if (ACT == MG_ACT_SIGMOID) z[e] = mg_sigmoid_fast(z[e]);
                            if (c0 + e >= N) z[e] = 0.f;
                        }
                        PclStore<COLS, OutT>::put(Y + (size_t)f * ldy + c0, z);
                    }
                }
            }
        }
    }
}

// The masked MSE (morgana/losses.py:29-51) of a ONE-column prediction that is constant over each table row's frames, from the per-row
// statistics of mg_phone_target_stats / mg_phone_front:  sum_f w_f (p_r - y_f)^2 = weight[r] (p_r - ybar[r])^2 + c_r.  This kernel:
// loss = sum_r weight[r] (pred[r] - ybar[r])^2 (the c_r are added by mg_phone_loss_const_add / mg_expand_column_loss_f32) and
// dpred[r] = 2 weight[r] (pred[r] - ybar[r]).  ONE workgroup, fixed order: thread t sums rows t, t + 1024, ... in double, tree in LDS.
__global__ __launch_bounds__(1024) void phone_mse_rows_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ ybar,
                                                              const float* __restrict__ weight, int n, float* __restrict__ loss,
                                                              float* __restrict__ dpred) {
    __shared__ double part[1024];
    double acc = 0.0;
    for (int r = threadIdx.x; r < n; r += 1024) {
        const float w = weight[r];
        const float d = w > 0.f ? pred[(size_t)r * ldp] - ybar[r] : 0.f;
        dpred[r] = 2.f * w * d;
        acc += (double)(w * d) * (double)d;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)part[0];
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

size_t mg_segment_sum_feat_workspace_bytes(int C, int ldo) {
    return C > 0 && ldo > 0 ? (size_t)SSF_BLOCKS * (size_t)C * (size_t)ldo * sizeof(float) : 0;
}

int mg_segment_sum_feat_bf16(const uint16_t* G, int ldg, const int32_t* rows, int64_t M, const int32_t* seg_start, const int32_t* seg_end,
                             int R, int extra, int N, uint16_t* out, int ldo, const float* feat, int C, void* slabs, size_t slabs_bytes,
                             void* stream) {
    MG_CHECK_ARG(G && rows && seg_start && seg_end && out && feat && slabs && M > 0 && R > 0 && extra >= 0 && N > 0,
                 "mg_segment_sum_feat_bf16: bad arguments (M=%lld R=%d extra=%d N=%d)", (long long)M, R, extra, N);
    MG_CHECK_ARG(N % 8 == 0 && ldg % 8 == 0 && ldg >= N && ldo % 8 == 0 && ldo >= N, "mg_segment_sum_feat_bf16: N=%d ldg=%d ldo=%d must be multiples of 8",
                 N, ldg, ldo);
    MG_CHECK_ARG(C >= 1 && C <= 16, "mg_segment_sum_feat_bf16: C=%d (1..16)", C);
    MG_CHECK_ARG(slabs_bytes >= mg_segment_sum_feat_workspace_bytes(C, ldo), "mg_segment_sum_feat_bf16: slabs too small (%zu bytes)", slabs_bytes);
    MG_CHECK_ARG(((uintptr_t)G % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)slabs % 16) == 0,
                 "mg_segment_sum_feat_bf16: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(SSF_BLOCKS, (unsigned)mg_ceil_div(ldo / 8, 64));
    hipLaunchKernelGGL(segment_sum_feat_kernel, grid, dim3(64 * SSF_WAVES), 0, st, G, ldg, rows, M, seg_start, seg_end, R, extra, N, out, ldo, feat, C,
                       (float*)slabs);
    MG_CHECK_LAUNCH("mg_segment_sum_feat_bf16");
    return MG_OK;
}

int mg_feat_wgrad_reduce(const void* slabs, int C, int ldo, int N, float* dW, int ldw, int col0, int accumulate, void* stream) {
    MG_CHECK_ARG(slabs && dW && C >= 1 && C <= 16 && N > 0 && ldo >= N && col0 >= 0 && ldw >= col0 + C,
                 "mg_feat_wgrad_reduce: bad arguments (C=%d N=%d ldo=%d ldw=%d col0=%d)", C, N, ldo, ldw, col0);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(feat_wgrad_reduce_kernel, dim3((unsigned)mg_ceil_div(N, 64), (unsigned)C), dim3(256), 0, st, (const float*)slabs, SSF_BLOCKS, C,
                       ldo, N, dW, ldw, col0, accumulate);
    MG_CHECK_LAUNCH("mg_feat_wgrad_reduce");
    return MG_OK;
}

int mg_phone_concat_layer_bf16(const float* P, int ldp, const int32_t* rows, int64_t M, const float* feat, int C, const float* W, int ldw,
                               int col0, const float* bias, int N, int act, void* Y, int ldy, int y_f32, void* stream) {
    MG_CHECK_ARG(P && rows && feat && W && Y && M > 0 && N > 0, "mg_phone_concat_layer_bf16: bad arguments (M=%lld N=%d)", (long long)M, N);
    MG_CHECK_ARG(C >= 1 && C <= 16 && col0 >= 0 && ldw >= col0 + C, "mg_phone_concat_layer_bf16: C=%d (1..16) col0=%d ldw=%d", C, col0, ldw);
    MG_CHECK_ARG(ldp % 8 == 0 && ldp >= ((N + 7) / 8) * 8 && ldy % 8 == 0 && ldy >= N,
                 "mg_phone_concat_layer_bf16: N=%d ldp=%d ldy=%d (multiples of 8, ldp >= N rounded up to 8)", N, ldp, ldy);
    MG_CHECK_ARG(act == MG_ACT_NONE || act == MG_ACT_SIGMOID, "mg_phone_concat_layer_bf16: act=%d", act);
    MG_CHECK_ARG(((uintptr_t)P % 16) == 0 && ((uintptr_t)Y % 16) == 0, "mg_phone_concat_layer_bf16: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int cols = ldy <= 256 ? 4 : 8;                                // columns per lane: 64 lanes cover 256 / 512 columns per pass
    const int64_t want = mg_ceil_div(mg_ceil_div(M, PCL_CHUNK), 4 * 4);  // ~4 chunks of 16 frames per wave
    const dim3 grid((unsigned)(want < 1024 ? (want < 1 ? 1 : want) : 1024), (unsigned)mg_ceil_div(ldy, 64 * cols));
#define LAUNCH_PCL(ACT_, T_, COLS_) hipLaunchKernelGGL((phone_concat_layer_kernel<ACT_, T_, COLS_>), grid, dim3(256), 0, st, P, ldp, rows, M, feat, C, W, ldw, col0, bias, N, (T_*)Y, ldy)
#define LAUNCH_PCL_T(ACT_, T_) do { if (cols == 4) LAUNCH_PCL(ACT_, T_, 4); else LAUNCH_PCL(ACT_, T_, 8); } while (0)
    if (act == MG_ACT_SIGMOID) {
        if (y_f32) LAUNCH_PCL_T(MG_ACT_SIGMOID, float); else LAUNCH_PCL_T(MG_ACT_SIGMOID, uint16_t);
    } else {
        if (y_f32) LAUNCH_PCL_T(MG_ACT_NONE, float); else LAUNCH_PCL_T(MG_ACT_NONE, uint16_t);
    }
#undef LAUNCH_PCL_T
#undef LAUNCH_PCL
    MG_CHECK_LAUNCH("mg_phone_concat_layer_bf16");
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

int mg_phone_mse_rows_f32(const float* pred, int ldp, const float* ybar, const float* weight, int n_rows, float* loss, float* dpred,
                          void* stream) {
    MG_CHECK_ARG(pred && ybar && weight && loss && dpred && n_rows > 0 && ldp >= 1, "mg_phone_mse_rows_f32: bad arguments (n_rows=%d ldp=%d)",
                 n_rows, ldp);
    hipLaunchKernelGGL(phone_mse_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred, ldp, ybar, weight, n_rows, loss, dpred);
    MG_CHECK_LAUNCH("mg_phone_mse_rows_f32");
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
