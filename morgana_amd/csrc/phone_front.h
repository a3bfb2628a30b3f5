// The front of the phone-rate step as ONE tile program: upsample_to_repetitions' frame map (K1, csrc/upsample.hip; reference
// morgana/utils.py:175-228) and the per-phone statistics of the masked MSE (phone_target_stats_kernel, csrc/phone_rate.hip; reference
// morgana/losses.py:29-51) per utterance.  The statistics of a phone read only the frame run of that phone, which the scan of its own
// utterance's durations gives - no other workgroup's output - so the two launches are one job per utterance here, and because a job is a
// device function on a caller-provided LDS block it can also ride in the grid of an unrelated launch (the first layer's GEMM reads
// neither: phone_front_gemm_kernel, gemm_bf16_big.hip).
//
// Jobs [0, B): utterance b - rows32 / rows_mapped / seg_start / seg_end exactly as upsample_index_kernel writes them, ybar / weight of
//   its phone rows exactly as phone_target_stats_kernel computes them (16 lanes per row, the same order of sums), and the utterance's
//   share of the loss's constant term (summed per block over its jobs: one partial sum in the slot of the block's first job).
// Jobs [B, B + ceil(extra / 4)): four extra rows (one wave each) - the padding frames of their chunk of the frame axis, found from the
//   durations' totals (t >= total_b) instead of from the map the launch is still writing.
// partial[]: the layout of mg_phone_target_stats' workspace, ceil(R / 16) + ceil(extra / 4) floats whose sum is the constant: slot b for
//   utterance b, zeros up to ceil(R / 16), then one slot per extra job (needs B <= ceil(R / 16): at least ~16 phones per utterance).
// 256 or 512 threads per job; LDS: (threads + P + T + 1) ints for an utterance, (threads + utterances spanned by four chunks) for an extra job.
#pragma once
#include "common.h"

struct PhoneFrontArgs {
    const int64_t* dur;        // [B, P]
    const float* target;       // [B, T] (frame b * T + t)
    const int64_t* seq_len;    // [B] or null
    int B, P, T, extra;
    int32_t* rows32;           // [B, T]: b * P + phone or -1
    int32_t* rows_mapped;      // [B, T]: -1 -> pad_row
    int pad_row;
    int32_t* seg_start;        // [B * P]
    int32_t* seg_end;
    float* ybar;               // [B * P + extra]
    float* weight;
    float* partial;            // [ceil(B * P / 16) + ceil(extra / 4)]
    int lds_ints;              // ints of LDS the caller provides (host-checked against the two needs above)
    int probe;                 // timing probes (results garbage): bit 0 = no utterance compute, bit 1 = no extra compute, bit 2 = no extra jobs, bit 3 = (rider) no GEMM
};

__host__ __device__ inline int phone_front_jobs(int B, int extra) { return B + (extra + 3) / 4; }

__device__ __forceinline__ float pf_frame_weight(int64_t f, const int64_t* __restrict__ seq_len, int B, int T) {
    const int b = (int)(f / T);
    const int t = (int)(f - (int64_t)b * T);
    int64_t nb = seq_len ? seq_len[b] : (int64_t)T;
    if (nb > T) nb = T;
    if (nb < 0) nb = 0;
    const float maskf = (int64_t)t < nb ? 1.f : 0.f;
    return maskf * (1.f / ((float)nb * (float)B));          // n_b == 0 -> 0 * inf = NaN, as the reference
}

// A block's work is a list of jobs (first, first + stride, ...: one job as a launch of its own, several as a rider).  They run in
// batches of as many as the LDS holds: the global reads of a whole batch are issued first (durations, targets, seq_len -> LDS), then
// the jobs are worked off from LDS - one memory latency per batch instead of two or three per job, which is what a rider beside a
// GEMM that saturates the memory system pays for.  Every one of the block's NT threads (256 or 512) makes the same calls.
//
// LDS: [0, 8) the scan's wave totals, [16, 32) two sets of wave sums of the constant, from NT on the batch's regions.
// Region of an utterance: cum[P] | tgt[T] | n_b.  Region of an extra job: the totals of the utterances its four chunks span.

// Stage the batch's utterances first + k stride, k < n (regions ureg ints apart from `reg0`): one flat index space over all of them,
// four independent loads in flight per thread before the first LDS store waits for one.
template <int NT>
__device__ __forceinline__ void pf_stage_utterances(const PhoneFrontArgs& a, int first, int stride, int n, int* __restrict__ reg0, int ureg) {
    const int tid = threadIdx.x, P = a.P, T = a.T;
    const int per = P + T;                               // staged words per utterance besides n_b
    const int words = n * per;
    for (int base = 0; base < words; base += 4 * NT) {
        int v[4];
        int where[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * NT + tid;
            where[u] = -1;
            if (i < words) {
                const int k = i / per, o = i - k * per, b = first + k * stride;
                where[u] = k * ureg + o;
                if (o < P) {
                    const long long d = a.dur[(size_t)b * P + o];
                    v[u] = d > 0 ? (int)d : 0;
                } else {
                    v[u] = __builtin_bit_cast(int, a.target[(size_t)b * T + (o - P)]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (where[u] >= 0) reg0[where[u]] = v[u];
    }
    for (int k = tid; k < n; k += NT) {
        const int b = first + k * stride;
        int64_t nb = a.seq_len ? a.seq_len[b] : (int64_t)T;
        reg0[k * ureg + per] = (int)(nb > T ? T : (nb < 0 ? 0 : nb));
    }
}

// after a barrier behind the staging; returns this thread's share of the constant
template <int NT>
__device__ __forceinline__ float pf_compute_utterance(const PhoneFrontArgs& a, int b, int* __restrict__ lds, int* __restrict__ reg) {
    const int tid = threadIdx.x;
    const int B = a.B, P = a.P, T = a.T;
    int* scratch = lds;
    int* cum = reg;
    const float* tgt = reinterpret_cast<const float*>(reg + P);
    float c = 0.f;
    {   // inclusive scan of cum[0:P] in place (the sums of block_scan_durations, upsample.hip): a scan inside each wave, the waves'
        // totals through LDS - two barriers, where a Hillis-Steele scan over the block takes eighteen
        const int per = (P + NT - 1) / NT;
        const int lo = min(tid * per, P), hi = min(lo + per, P);
        int local = 0;
        for (int p = lo; p < hi; ++p) local += cum[p];
        int incl = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if ((tid & 63) >= off) incl += v;
        }
        if ((tid & 63) == 63) scratch[tid >> 6] = incl;
        __syncthreads();
        int run = incl - local;
        for (int w = 0; w < (tid >> 6); ++w) run += scratch[w];
        for (int p = lo; p < hi; ++p) {
            run += cum[p];
            cum[p] = run;
        }
        __syncthreads();
    }
    // the frame map: a phone's 16 lanes write its frames' rows in the statistics loop below (no search per frame: as a rider the job is
    // a chain of dependent instructions on two waves per SIMD, and seven dependent LDS reads per frame were a third of it); here the
    // padding frames behind the utterance's last phone
    const int total = P > 0 ? min(cum[P - 1], T) : 0;
    for (int t = total + tid; t < T; t += NT) {
        const size_t o = (size_t)b * T + t;
        a.rows32[o] = -1;
        a.rows_mapped[o] = a.pad_row;
    }
    const int nb = reg[P + T];
    const float inv = 1.f / ((float)nb * (float)B);      // n_b == 0 -> inf; 0 * inf = NaN below, as the reference
    const int sub = tid & 15;
    for (int p0 = 0; p0 < P; p0 += NT / 16) {
        const int p = p0 + (tid >> 4);
        float w_sum = 0.f, wy = 0.f;
        int s = 0, e = 0;                            // the phone's frames [s, e) of this utterance; (0, 0): none
        if (p < P) {
            s = min(p ? cum[p - 1] : 0, T);
            e = min(cum[p], T);
            if (e <= s) s = e = 0;
            if (sub == 0) {
                a.seg_start[(size_t)b * P + p] = e > s ? b * T + s : 0;
                a.seg_end[(size_t)b * P + p] = e > s ? b * T + e : 0;
            }
            for (int t = s + sub; t < e; t += 16) {
                const float w = (t < nb ? 1.f : 0.f) * inv;
                w_sum += w;
                wy += w * tgt[t];
                const size_t o = (size_t)b * T + t;
                a.rows32[o] = b * P + p;                 // phones of duration 0 own no frame, as np.repeat skips them
                a.rows_mapped[o] = b * P + p;
            }
        }
        w_sum = mg_row16_sum(w_sum);
        wy = mg_row16_sum(wy);
        const float mean = w_sum > 0.f ? wy / w_sum : 0.f;
        for (int t = s + sub; t < e; t += 16) {
            const float d = tgt[t] - mean;
            c += ((t < nb ? 1.f : 0.f) * inv) * d * d;
        }
        if (p < P && sub == 0) {
            a.ybar[(size_t)b * P + p] = mean;
            a.weight[(size_t)b * P + p] = w_sum;
        }
    }
    const int phone_blocks = (B * P + 15) / 16;
    for (int i = B + b + tid * B; i < phone_blocks; i += NT * B) a.partial[i] = 0.f;     // the slots no utterance owns
    return c;
}

// the utterances the four chunks of extra job `xj` span: [b_lo, b_hi] (frame ids and M + 4 chunk fit 32 bits unsigned)
__device__ __forceinline__ void pf_extra_span(const PhoneFrontArgs& a, int xj, unsigned chunk, int& b_lo, int& b_hi) {
    const unsigned M = (unsigned)a.B * (unsigned)a.T, T = (unsigned)a.T;
    const unsigned f_lo = (unsigned)xj * 4u * chunk, f_hi = min(M, f_lo + 4u * chunk);
    b_lo = f_lo < M ? (int)(f_lo / T) : 0;
    b_hi = f_hi > f_lo ? (int)((f_hi - 1u) / T) : b_lo - 1;
}

__device__ __forceinline__ unsigned pf_chunk(const PhoneFrontArgs& a) {
    return (unsigned)(((int64_t)a.B * a.T + a.extra - 1) / a.extra);
}

// Totals of the utterances the extra jobs first + k stride (k < n, job ids counted from the first extra job) span, regions xreg ints
// apart: one wave per (job, utterance) pair, the pairs dealt round robin so that their loads are in flight together.
template <int NT>
__device__ __forceinline__ void pf_stage_extras(const PhoneFrontArgs& a, int first, int stride, int n, int* __restrict__ reg0, int xreg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, P = a.P;
    const unsigned chunk = pf_chunk(a);
    for (int pair = wave; pair < n * xreg; pair += NT / 64) {
        const int k = pair / xreg, i = pair - k * xreg;
        int b_lo, b_hi;
        pf_extra_span(a, first + k * stride, chunk, b_lo, b_hi);
        const int b = b_lo + i;
        if (b > b_hi) continue;
        int s = 0;
        for (int p = lane; p < P; p += 64) {
            const long long d = a.dur[(size_t)b * P + p];
            s += d > 0 ? (int)d : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) reg0[k * xreg + i] = s;
    }
}

// One extra row j (a whole wave): the padding frames of its chunk [lo, hi) - per utterance b the frames from b T + total_b on,
// visited directly, no test per frame.  Lane l takes the frames f = lo + l (mod 64) in ascending order, as phone_target_stats_kernel
// does, so the sums come out bit for bit.  Returns the lane's share of the constant.
__device__ __forceinline__ float pf_extra_row(const PhoneFrontArgs& a, int j, int lo, int hi, const int* __restrict__ tot, int b_lo) {
    const int lane = threadIdx.x & 63;
    const int B = a.B, T = a.T, R = a.B * a.P;
    const int b0 = lo / T, b1 = (hi - 1) / T;
    float w_sum = 0.f, wy = 0.f, c = 0.f;
    for (int b = b0; b <= b1; ++b) {
        const int p_lo = max(lo, b * T + min(tot[b - b_lo], T)), p_hi = min(hi, (b + 1) * T);
        if (p_lo >= p_hi) continue;
        for (int f = lo + ((p_lo - lo) & ~63) + lane; f < p_hi; f += 64) {
            if (f >= p_lo) {
                const float w = pf_frame_weight(f, a.seq_len, B, T);
                w_sum += w;
                wy += w * a.target[f];
            }
        }
    }
    w_sum = mg_wave_sum(w_sum);
    wy = mg_wave_sum(wy);
    const float mean = w_sum > 0.f ? wy / w_sum : 0.f;
    for (int b = b0; b <= b1; ++b) {
        const int p_lo = max(lo, b * T + min(tot[b - b_lo], T)), p_hi = min(hi, (b + 1) * T);
        if (p_lo >= p_hi) continue;
        for (int f = lo + ((p_lo - lo) & ~63) + lane; f < p_hi; f += 64) {
            if (f >= p_lo) {
                const float d = a.target[f] - mean;
                c += pf_frame_weight(f, a.seq_len, B, T) * d * d;
            }
        }
    }
    if (lane == 0) {
        a.ybar[R + j] = mean;
        a.weight[R + j] = w_sum;
    }
    return c;
}

// The extra rows of the jobs first + k stride (k < n; job ids counted from the first extra job), their totals staged xreg ints apart
// (after a barrier behind the staging).  One LANE per row first: does its chunk hold padding frames at all?  (A batch without padding
// ends here: zeros for its rows, one pass of integer arithmetic for all of them.)  Then every wave works off the rows its own lanes
// found padding in, one after the other (pf_extra_row) - no LDS, no barrier.
template <int NT>
__device__ __forceinline__ float pf_compute_extras(const PhoneFrontArgs& a, int first, int stride, int n, const int* __restrict__ reg0,
                                                   int xreg) {
    const int tid = threadIdx.x;
    const int T = a.T, R = a.B * a.P;
    const unsigned M = (unsigned)a.B * (unsigned)T, chunk = pf_chunk(a);
    float c = 0.f;
    for (int r0 = 0; r0 < 4 * n; r0 += NT) {
        const int r = r0 + tid, k = r >> 2;
        const int j = (first + k * stride) * 4 + (r & 3);
        bool any = false;
        int lo = 0, hi = 0, b_lo = 0, b_hi = -1;
        if (r < 4 * n && j < a.extra) {
            pf_extra_span(a, first + k * stride, chunk, b_lo, b_hi);
            const unsigned lo_u = min((unsigned)j * chunk, M);
            lo = (int)lo_u;
            hi = (int)min(lo_u + chunk, M);
            if (lo < hi) {
                const int* tot = reg0 + k * xreg;
                for (int b = lo / T; b * T < hi; ++b) any = any || max(lo, b * T + min(tot[b - b_lo], T)) < min(hi, (b + 1) * T);
            }
            if (!any) {
                a.ybar[R + j] = 0.f;
                a.weight[R + j] = 0.f;
            }
        }
        unsigned long long todo = __ballot(any);
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int kk = __shfl(k, src, 64);
            c += pf_extra_row(a, __shfl(j, src, 64), __shfl(lo, src, 64), __shfl(hi, src, 64), reg0 + kk * xreg, __shfl(b_lo, src, 64));
        }
    }
    return c;
}

// The block's share of the constant: wave sums, then eight values by one thread, into the slot of `job`
template <int NT>
__device__ __forceinline__ void pf_store_partial(const PhoneFrontArgs& a, int job, float c, int* __restrict__ lds) {
    const int tid = threadIdx.x;
    float* red = reinterpret_cast<float*>(lds + 16);
    c = mg_wave_sum(c);
    if ((tid & 63) == 0) red[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
        float tot_c = 0.f;
        for (int w = 0; w < NT / 64; ++w) tot_c += red[w];
        a.partial[job < a.B ? job : (a.B * a.P + 15) / 16 + (job - a.B)] = tot_c;
    }
}

template <int NT>
__device__ __forceinline__ void phone_front_block(const PhoneFrontArgs& a, int first_job, int stride, int* __restrict__ lds) {
    const int tid = threadIdx.x;
    const int B = a.B, jobs = phone_front_jobs(a.B, a.extra);
    if (first_job >= jobs) return;
    int room = a.lds_ints - NT;
    // The extra jobs' staging (a few totals each) goes ahead of everything when all of it fits beside an utterance's region: it then
    // shares the first batch's trip to memory.
    const int ureg = a.P + a.T + 1;
    int xreg = 0, n_x = 0, first_x = first_job;
    if (first_x < B) first_x += (B - first_x + stride - 1) / stride * stride;
    if (a.extra > 0 && !(a.probe & 4)) {
        const int M = B * a.T, chunk = (int)(((int64_t)M + a.extra - 1) / a.extra);
        xreg = (int)(((int64_t)4 * chunk + a.T - 1) / a.T) + 2;
        n_x = first_x < jobs ? (jobs - first_x + stride - 1) / stride : 0;
    }
    const bool x_ahead = n_x > 0 && (int64_t)n_x * xreg <= room - ureg;
    int* xbase = lds + NT + room - n_x * xreg;
    if (x_ahead) {
        room -= n_x * xreg;
        pf_stage_extras<NT>(a, first_x - B, stride, n_x, xbase, xreg);
    }
    float c = 0.f;                                       // this thread's share of the constant over all of the block's jobs
    // utterances
    const int ubatch = max(1, room / ureg);
    int job = first_job;
    while (job < B) {
        const int n = min(ubatch, (B - job + stride - 1) / stride);
        pf_stage_utterances<NT>(a, job, stride, n, lds + NT, ureg);
        __syncthreads();
        for (int k = 0; k < n; ++k, job += stride)
            if (!(a.probe & 1)) c += pf_compute_utterance<NT>(a, job, lds, lds + NT + k * ureg);
        __syncthreads();                                 // the regions are staged again
    }
    // extra rows
    if (x_ahead) {
        if (first_job >= B) __syncthreads();             // no utterance batch stood between the staging and here
        if (!(a.probe & 2)) c += pf_compute_extras<NT>(a, first_x - B, stride, n_x, xbase, xreg);
    } else if (n_x > 0) {
        const int xbatch = max(1, room / xreg);
        for (int done = 0; done < n_x;) {
            const int n = min(xbatch, n_x - done);
            pf_stage_extras<NT>(a, first_x + done * stride - B, stride, n, lds + NT, xreg);
            __syncthreads();
            if (!(a.probe & 2)) c += pf_compute_extras<NT>(a, first_x + done * stride - B, stride, n, lds + NT, xreg);
            __syncthreads();
            done += n;
        }
    }
    // one partial sum per block, in the slot of its first job; its other jobs' slots hold zero
    pf_store_partial<NT>(a, first_job, c, lds);
    const int phone_blocks = (B * a.P + 15) / 16;
    for (int k = 1 + tid; first_job + (int64_t)k * stride < jobs; k += NT) {
        const int jb = first_job + k * stride;
        a.partial[jb < B ? jb : phone_blocks + (jb - B)] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// WAVE jobs: the same jobs, each worked off by ONE wave with no workgroup barrier - the form the riders of a GEMM grid take.  A block
// job spreads an utterance over 256-512 threads that mostly wait for each other (two barriers in the scan, one per batch) and costs
// its block ~2.3 us; a rider block is a whole CU because the grid's LDS size is the GEMM's, so 32 rider blocks took 37 us for C2's 512
// jobs.  As wave jobs the eight waves of a rider block run eight jobs at once (~4 us each): 256 at a time on 32 CUs.
// Outputs: rows32 / rows_mapped / seg_start / seg_end / ybar / weight bit for bit the block jobs' (the phone loop is the same code at
// four phones per pass, 16 lanes per phone); partial[]: one sum per job in the job's own slot (same total, another order of sums).
// LDS of a wave: max(P + T + 1, span + 2) ints (phone_front_wave_ints).
__device__ __forceinline__ void pf_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }      // LDS of ONE wave: in order

__device__ __forceinline__ void pf_wave_utterance(const PhoneFrontArgs& a, int b, int* __restrict__ reg) {
    const int lane = threadIdx.x & 63;
    const int B = a.B, P = a.P, T = a.T;
    int* cum = reg;
    float* tgt = reinterpret_cast<float*>(reg + P);
    for (int p = lane; p < P; p += 64) {
        const long long d = a.dur[(size_t)b * P + p];
        cum[p] = d > 0 ? (int)d : 0;
    }
    for (int t = lane; t < T; t += 64) tgt[t] = a.target[(size_t)b * T + t];
    int64_t nbl = a.seq_len ? a.seq_len[b] : (int64_t)T;
    const int nb = (int)(nbl > T ? T : (nbl < 0 ? 0 : nbl));
    pf_wave_fence();
    int carry = 0;                                       // inclusive scan of the durations, 64 at a time
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int v = p0 + lane < P ? cum[p0 + lane] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        if (p0 + lane < P) cum[p0 + lane] = carry + incl;
        carry += __shfl(incl, 63, 64);
    }
    pf_wave_fence();
    float c = 0.f;
    const int total = P > 0 ? min(cum[P - 1], T) : 0;
    for (int t = total + lane; t < T; t += 64) {
        const size_t o = (size_t)b * T + t;
        a.rows32[o] = -1;
        a.rows_mapped[o] = a.pad_row;
    }
    const float inv = 1.f / ((float)nb * (float)B);      // n_b == 0 -> inf; 0 * inf = NaN below, as the reference
    const int sub = lane & 15;
    for (int p0 = 0; p0 < P; p0 += 4) {
        const int p = p0 + (lane >> 4);
        float w_sum = 0.f, wy = 0.f;
        int s = 0, e = 0;
        if (p < P) {
            s = min(p ? cum[p - 1] : 0, T);
            e = min(cum[p], T);
            if (e <= s) s = e = 0;
            if (sub == 0) {
                a.seg_start[(size_t)b * P + p] = e > s ? b * T + s : 0;
                a.seg_end[(size_t)b * P + p] = e > s ? b * T + e : 0;
            }
            for (int t = s + sub; t < e; t += 16) {
                const float w = (t < nb ? 1.f : 0.f) * inv;
                w_sum += w;
                wy += w * tgt[t];
                const size_t o = (size_t)b * T + t;
                a.rows32[o] = b * P + p;
                a.rows_mapped[o] = b * P + p;
            }
        }
        w_sum = mg_row16_sum(w_sum);
        wy = mg_row16_sum(wy);
        const float mean = w_sum > 0.f ? wy / w_sum : 0.f;
        for (int t = s + sub; t < e; t += 16) {
            const float d = tgt[t] - mean;
            c += ((t < nb ? 1.f : 0.f) * inv) * d * d;
        }
        if (p < P && sub == 0) {
            a.ybar[(size_t)b * P + p] = mean;
            a.weight[(size_t)b * P + p] = w_sum;
        }
    }
    const int phone_blocks = (B * P + 15) / 16;
    for (int i = B + b + lane * B; i < phone_blocks; i += 64 * B) a.partial[i] = 0.f;      // the slots no utterance owns
    c = mg_wave_sum(c);
    if (lane == 0) a.partial[b] = c;
    pf_wave_fence();                                     // the region is staged again by the wave's next job
}

__device__ __forceinline__ void pf_wave_extras(const PhoneFrontArgs& a, int xj, int* __restrict__ reg) {
    const int lane = threadIdx.x & 63;
    const int T = a.T, P = a.P, R = a.B * a.P;
    const unsigned M = (unsigned)a.B * (unsigned)T, chunk = pf_chunk(a);
    int b_lo, b_hi;
    pf_extra_span(a, xj, chunk, b_lo, b_hi);
    for (int b = b_lo; b <= b_hi; ++b) {                 // totals of the utterances the four chunks span (pf_stage_extras' sums)
        int s = 0;
        for (int p = lane; p < P; p += 64) {
            const long long d = a.dur[(size_t)b * P + p];
            s += d > 0 ? (int)d : 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        if (lane == 0) reg[b - b_lo] = s;
    }
    pf_wave_fence();
    float c = 0.f;
    for (int r = 0; r < 4; ++r) {
        const int j = 4 * xj + r;
        if (j >= a.extra) break;
        const unsigned lo_u = min((unsigned)j * chunk, M);
        const int lo = (int)lo_u, hi = (int)min(lo_u + chunk, M);
        bool any = false;                                // wave-uniform: does the chunk hold padding frames at all?
        if (lo < hi)
            for (int b = lo / T; b * T < hi; ++b) any = any || max(lo, b * T + min(reg[b - b_lo], T)) < min(hi, (b + 1) * T);
        if (any) {
            c += pf_extra_row(a, j, lo, hi, reg, b_lo);
        } else if (lane == 0) {
            a.ybar[R + j] = 0.f;
            a.weight[R + j] = 0.f;
        }
    }
    c = mg_wave_sum(c);
    if (lane == 0) a.partial[(R + 15) / 16 + xj] = c;
    pf_wave_fence();
}

// Every wave of the caller: jobs first_wave, first_wave + n_waves, ... (first_wave = this wave's index among all rider waves), on the
// wave's own LDS region `reg`.
__device__ __forceinline__ void phone_front_wave_jobs(const PhoneFrontArgs& a, int first_wave, int n_waves, int* __restrict__ reg) {
    const int jobs = phone_front_jobs(a.B, a.extra);
    for (int job = first_wave; job < jobs; job += n_waves) {
        if (job < a.B) {
            if (!(a.probe & 1)) pf_wave_utterance(a, job, reg);
        } else if (!(a.probe & 2)) {
            pf_wave_extras(a, job - a.B, reg);
        }
    }
}

// ints of LDS ONE WAVE needs for its jobs
static inline int64_t phone_front_wave_ints(int B, int P, int T, int extra) {
    int64_t need = (int64_t)P + T + 1;
    if (extra > 0) {
        const int64_t M = (int64_t)B * T, chunk = (M + extra - 1) / extra;
        const int64_t span = (4 * chunk + T - 1) / T + 2;
        if (span > need) need = span;
    }
    return (need + 3) / 4 * 4;
}

// ints of LDS one job needs (more holds a batch), for either block size (host side: the launchers check it against what their kernel provides)
static inline int64_t phone_front_lds_ints(int B, int P, int T, int extra) {
    int64_t need = 512 + (int64_t)P + T + 1;
    if (extra > 0) {
        const int64_t M = (int64_t)B * T, chunk = (M + extra - 1) / extra;
        const int64_t span = (4 * chunk + T - 1) / T + 2;
        if (512 + span > need) need = 512 + span;
    }
    return need;
}
