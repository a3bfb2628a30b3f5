/*
 * morgana_hip.h - C ABI of libmorgana_hip.so: the MI355X (gfx950) kernels behind morgana's acoustic-model
 * training hot path (ExperimentBuilder.train_epoch -> BaseSPSS.forward -> backward -> Adam).
 *
 * The reference (ZackHodari/morgana) is pure Python and has no FFI; every entry point below replaces the
 * PyTorch-eager op sequence of the cited reference lines.  The reference-side binding a maintainer would add is
 * the ctypes stub shown in INTEGRATION.md (morgana_amd/_lib.py is that stub, used by this repo's own host layer).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.  All pointers are DEVICE pointers unless a
 *     parameter is documented as host.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - Every function is asynchronous on `stream`, never allocates or frees memory and never synchronises the
 *     device: the caller owns outputs and workspaces (sizes from the *_workspace_bytes helpers).
 *   - Return value: 0 on success, negative MG_E* on error; mg_last_error() gives a thread-local message.
 *   - Row-major everywhere.  "bf16" buffers hold raw 16-bit bfloat16 values (uint16_t).
 *   - Gather convention ("rows"): an int32 array, one entry per output row, holding the source-table row or -1
 *     for the all-zero pad row (reference: the zero `padder` frame appended at utils.py:206-207).
 */
#ifndef MORGANA_HIP_H
#define MORGANA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_OK 0
#define MG_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define MG_ELAUNCH (-2)  /* HIP reported a launch error */
#define MG_EWORKSPACE (-3) /* workspace too small */

#define MG_ACT_NONE 0
#define MG_ACT_SIGMOID 1
/* OR-ed into `act` of mg_linear_fwd_bf16: `rows` is made of RUNS of equal consecutive indices (the frame map of
 * upsample_to_repetitions: every phone row repeated `dur` times).  A performance hint only - results do not depend on it: the
 * gathered operand is then staged once per distinct row (csrc/gemm_nt_runs.hip); rows without runs cost up to 4 passes there. */
#define MG_ACT_ROWS_RUNS 0x100

/* ------------------------------------------------------------------------------------------------------------------
 * Library
 * ---------------------------------------------------------------------------------------------------------------- */
const char* mg_last_error(void);
/* Scheduling choices: which of several kernels / split plans that compute THE SAME RESULTS an entry point launches.  For A/B runs
 * inside one process and for the equality tests that hold the forms against each other.  PROCESS-GLOBAL state (not per call, not
 * thread-local): set it between launches, not concurrently with them; everything else in this ABI is stateless apart from the
 * thread-local error text.  Unknown keys and values are refused (MG_EINVAL).  Measured-slower experiments and timing probes are
 * not in this library at all: they are compiled into the lab builds (make -C morgana_amd/csrc lab / diag) only. */
#define MG_TUNING_KEYS 8
#define MG_TUNE_FORM 0          /* large bf16 GEMM kernels: 0 = default; 3 = 32-deep instead of 64-deep stages of the 128-wide NT tile;
                                 * 6 = per-tile instead of persistent NT kernel; 14 = frame-staged instead of run-staged layer-1 forward,
                                 * 16 = its fragments read one k-step ahead;
                                 * fused backward: 12 = tiles two steps ahead (larger ring), 13 = 32-frame steps, 7 = single-buffered */
#define MG_TUNE_GRU_HANDOFF 2   /* persistent GRU / LSTM kernels: 0 = groups found on one XCD hand the state over through that XCD's L2,
                                 * bit 0 (1) = always write-through (sc1) stores, the placement-independent form; bit 1 (2) = group
                                 * membership from the block index (block % 8) instead of a ticket of the XCD a workgroup runs on;
                                 * bit 2 (4) = the bf16 GRU backward with 8 groups x 16-unit slots (default: 16 groups x 32-unit slots, B <= 128) */
#define MG_TUNE_PERSISTENT 3    /* recurrences: 0 = persistent kernels where the shape has them, 1 = one launch per time step,
                                 * 2 = H = 64 on the 16-row tile instead of the 4x4x1 blocks */
#define MG_TUNE_WGRAD_SPLITS 4  /* wide weight-gradient kernel: != 0 overrides the planned number of split-M slabs (a multiple of 8);
                                 * 1000 + S / 2000 + S override the one-n-tile / the 4+-n-tile plans only */
#define MG_TUNE_WGRAD_ORDER 5   /* wide weight-gradient kernel, block order: 0 = planned, 1 = n tile fastest, 2 = the n tiles of a split
                                 * on one XCD, 3 = the 128 x 128 kernel instead of the wide one */
#define MG_TUNE_LSTM_BWD_STACK 6 /* LSTM stack wavefronts, hidden units per slot: 0 = backward 32 where they fit (one workgroup per CU),
                                 * forward 16 (two per CU); bit 0 (1) = backward 16; bit 1 (2) = forward 32 (same bits, measured slower) */
#define MG_TUNE_AB 7            /* shared-grid launches as their separate launches, half-width tiles off: 65 = mg_linear_wgrad_dgrad_bf16 as
                                 * two launches, 66 = mg_phone_front_linear_fwd_bf16 as two, 86 = mg_f0_l2tail_x3 with four waves per workgroup (default: eight), 87 / 88 = 64-deep LDS stages in the phone-rate forward GEMM (pair planes,
                                 * three passes / bf16; measured equal to slower, round 5), 89 = the pair-plane forward GEMM (mg_phone_front_linear_fwd_x3) as three passes
                                 * over the plane (default: all four planes' k-tile per stage), 90 = the pair-plane weight gradients
                                 * (mg_linear_wgrad_slabs_x3) as three walks over their rows (default: one walk, all four planes per
                                 * stage - the same products in another order), 91 = 128 x 512 tiles for the 512-wide weight
                                 * gradient where the square 256 x 256 tile is the default (N a multiple of 256), 92 = 128 x 640 tiles for the 640-wide weight
                                 * gradient at phone-rate rows, 93 = 128 x 512 tiles for the 512-wide one, 94 = mg_phone_front_linear_fwd_bf16 with the
                                 * front's jobs as block jobs and 256-row tiles (round 2's form), 95 = wave jobs but 256-row tiles,
                                 * 96 = mg_f0_l2tail_*_bf16 walks H1 from its first tile at every size (default: from the last one when
                                 * H1 is larger than the L2s hold; per-workgroup sums then add in another order);
                                 * the wide NT GEMM's tile for few-tile shapes: 97 = 256-wide whenever N allows, 99 = 128-wide always,
                                 * 98 = under 32,768 rows the 128 x 128 kernel (default: 128-wide when 256-wide tiles would be < 128) */
int mg_set_tuning(int key, int value);
int mg_version(void);           /* ABI version, bumped on incompatible change */
const char* mg_build_arch(void); /* "gfx950" */

/* ------------------------------------------------------------------------------------------------------------------
 * K1  upsample_to_repetitions            reference: morgana/utils.py:175-228
 * ---------------------------------------------------------------------------------------------------------------- */
/* n_frames[b] = sum_p dur[b,p]; *tmax = max_b n_frames[b]  (utils.py:198-199).  dur int64 [B,P]. */
int mg_upsample_lengths(const int64_t* dur, int B, int P, int64_t* n_frames, int64_t* tmax, void* stream);

/* Frame->phone map (utils.py:214-220, the per-utterance np.repeat loop).  Any of the outputs may be NULL.
 *   idx64 [B,t_cap]  phone index or -1   (the reference's `repeated_idxs`, bit exact)
 *   rows32[B,t_cap]  b*P + phone or -1   (flat source row, the gather convention of this library)
 * Frames t >= n_frames[b] get -1.  Requires t_cap >= 0 and P <= 16384.  If t_cap < n_frames[b] the row is cropped. */
int mg_upsample_index(const int64_t* dur, int B, int P, int t_cap, int64_t* idx64, int32_t* rows32, void* stream);
/* The same launch also writing the phone-rate maps (csrc/phone_rate.hip): rows_mapped = rows32 with -1 replaced by pad_row, and
 * seg_start / seg_end int32 [B*P] = the run of frame ids (b * t_cap + t) of every phone row, (0, 0) for a phone without frames. */
int mg_upsample_index_maps(const int64_t* dur, int B, int P, int t_cap, int32_t* rows32, int32_t* rows_mapped, int pad_row,
                           int32_t* seg_start, int32_t* seg_end, void* stream);

/* out[m,:] = rows[m] < 0 ? 0 : src[rows[m],:]   (utils.py:226).  src [R,F] f32, out [M,F] f32. */
int mg_gather_rows_f32(const float* src, const int32_t* rows, float* out, int64_t M, int F, void* stream);
/* Same gather, output converted to bf16 with leading dimension ldo >= F; columns F..ldo-1 are zero filled. */
int mg_gather_rows_bf16(const float* src, const int32_t* rows, uint16_t* out, int64_t M, int F, int ldo, void* stream);

/* Segment index maps (reference: split_to_segments / get_segment_ends, morgana/utils.py:231-330; the same scan-and-index
 * pattern as mg_upsample_index).  seg_lens int64 [B,S]; T = length of the sequence feature; L = longest segment.
 *   split [B,S,L] : flat source row b*T + start_s + j of position j of segment s, -1 where j >= len_s (the zero padder)
 *   ends  [B,S]   : flat source row b*T + cumsum_s - 1 of the last frame of segment s, -1 for empty segments
 * Either output may be NULL.  Feed them to mg_gather_rows_f32; rows that would fall beyond T come out as -1.
 * mg_scatter_rows_f32 is the adjoint of such a gather (distinct targets): dst[rows[m],:] = src[m,:] for rows[m] >= 0;
 * dst must be zero-filled by the caller. */
int mg_segment_index(const int64_t* seg_lens, int B, int S, int T, int L, int32_t* split, int32_t* ends, void* stream);
int mg_scatter_rows_f32(const float* src, const int32_t* rows, float* dst, int64_t M, int F, void* stream);

/* Packed frames for ragged batches.  The reference pads a batch to its longest utterance (collate_fn, morgana/data.py:183-193), runs
 * every nn.Linear on all B*T rows (morgana/utils.py:401-418) and masks the loss (losses.py:37-39); behind a recurrent wrapper the
 * padded frames are zero rows (utils.py:383).  mg_frame_layout builds the maps that let the row-wise layers run on the
 * total = sum_b min(seq_len[b], T) valid rows plus ONE representative zero row:
 *   offsets int32 [B+1]     first packed row of utterance b (offsets[B] = total)
 *   rows    int32 [total+1] dense row b*T + t of packed row i, utterance by utterance; rows[total] = -1
 *   inverse int32 [B*T]     packed row of dense row (b, t); `total` for padded frames
 * `total` is supplied by the caller (the host knows the lengths it collated); frames beyond it count as padding.
 * pack = mg_gather_rows_*(x, rows), unpack = mg_gather_rows_f32(packed, inverse); the adjoint of the unpack is a gather by `rows` for
 * the valid rows and, for the representative row, the sum over all padded dense rows: mg_pad_rows_colsum_f32 (g [B,T,D] -> out [D],
 * fixed summation order; workspace mg_pad_rows_colsum_workspace_bytes(B,T,D)). */
int mg_frame_layout(const int64_t* seq_len, int B, int T, int64_t total, int32_t* offsets, int32_t* rows, int32_t* inverse,
                    void* stream);
size_t mg_pad_rows_colsum_workspace_bytes(int B, int T, int D);
int mg_pad_rows_colsum_f32(const float* g, const int64_t* seq_len, int B, int T, int D, float* out, void* workspace, size_t workspace_bytes,
                           void* stream);

/* Gather fused with the frame-level concat the shipped models do right after it (models/RNN_SPSS.py:76-81,
 * models/f0_test_model.py:78-79: upsample_to_repetitions, then torch.cat with `normalised_counters`):
 *   out[m, 0:F] = src[rows[m], :] (0 where rows[m] < 0), out[m, F:F+C] = extra[m, 0:C], out[m, F+C:ldo] = 0.
 * extra [M,C] f32.  The bf16 form needs ldo % 8 == 0 and produces the padded layer-1 operand directly. */
int mg_gather_concat_f32(const float* src, const int32_t* rows, const float* extra, float* out, int64_t M, int F, int C,
                         int ldo, void* stream);
int mg_gather_concat_bf16(const float* src, const int32_t* rows, const float* extra, uint16_t* out, int64_t M, int F, int C,
                          int ldo, void* stream);

/* Adjoint of the gather (autograd of utils.py:226): grad_src[b,p,:] = sum of grad_out[b,t,:] over the frames of
 * phone p (frames of a phone are contiguous).  grad_out [B,T,F], dur int64 [B,P], grad_src [B,P,F]. Deterministic. */
int mg_upsample_backward_f32(const float* grad_out, const int64_t* dur, float* grad_src, int B, int P, int T, int F,
                             void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * sequence_mask                           reference: morgana/utils.py:115-144
 * ---------------------------------------------------------------------------------------------------------------- */
/* mask[b,t] = t < seq_len[b].  elem_size selects the output type: 1 = uint8, 4 = float32, 8 = int64. */
int mg_sequence_mask(const int64_t* seq_len, int B, int max_len, void* mask, int elem_size, int as_float,
                     void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K4  seq_len-masked MSE, forward + backward   reference: morgana/losses.py:29-51
 * ---------------------------------------------------------------------------------------------------------------- */
/* loss = mean_{b,d}( sum_t m[b,t] (p-y)^2 / n_b ), n_b = min(seq_len[b], T) (T when seq_len == NULL);
 * grad = 2 (p-y) m / (n_b B D) * grad_scale, written for every element (zeros on padding).  n_b == 0 gives NaN,
 * as the reference does.  pred/target [B,T,D] f32; loss: one float; grad may be NULL.
 * workspace: mg_masked_mse_workspace_bytes(B,T,D) bytes.  Deterministic two-stage reduction (no atomics). */
size_t mg_masked_mse_workspace_bytes(int B, int T, int D);
int mg_masked_mse_f32(const float* pred, const float* target, const int64_t* seq_len, int B, int T, int D,
                      float grad_scale, float* loss, float* grad, void* workspace, size_t workspace_bytes,
                      void* stream);

/* Same contract with binary cross entropy as the per-element loss (reference: losses.bce, morgana/losses.py:54-56, used for
 * the voicing stream at models/RNN_SPSS.py:137): -(y log p + (1-y) log(1-p)), logs clamped at -100 as torch does. */
int mg_masked_bce_f32(const float* pred, const float* target, const int64_t* seq_len, int B, int T, int D,
                      float grad_scale, float* loss, float* grad, void* workspace, size_t workspace_bytes,
                      void* stream);

/* Multi-stream loss (reference: LSTMAcousticModel.loss, models/RNN_SPSS.py:120-139, with the torch.sigmoid of :93):
 * pred [B,T,D] f32 holds the streams side by side (torch.split at :87-88); stream k covers columns col0..col0+width-1
 * and is scored against its own target [B,T,width] (row stride ldt) with losses.mse (MG_LOSS_MSE) or with
 * losses.bce(sigmoid(pred)) (MG_LOSS_SIGMOID_BCE).  loss = mean over streams of the per-stream sequence loss
 * (the `loss / 4.` of :139).  grad [B,T,D] (may be NULL) = grad_scale * d loss / d pred, zero in columns no stream
 * covers; prob [B,T,width] (may be NULL) receives sigmoid(pred) of the one BCE stream (`pred_vuv`, :93).
 * `streams` is a HOST array, n_streams <= MG_STREAMS_MAX. */
#define MG_STREAMS_MAX 8
#define MG_LOSS_MSE 0
#define MG_LOSS_SIGMOID_BCE 1
typedef struct {
    const float* target;
    int ldt;
    int col0;
    int width;
    int kind;
} mg_stream_desc;
size_t mg_stream_loss_workspace_bytes(int B, int T, int D);
int mg_stream_loss_f32(const float* pred, const mg_stream_desc* streams, int n_streams, const int64_t* seq_len, int B, int T,
                       int D, float grad_scale, float* loss, float* grad, float* prob, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Streaming metrics (reference: morgana/metrics.py:359-695; accumulated inside the shipped model's loss() every step,
 * models/RNN_SPSS.py:120-129): accum[0] += sum, accum[1] += count on the device - no host read-back (the reference calls .item()
 * per accumulate).  kind: MG_METRIC_MEAN (target only), _SQDIFF (RMSE / MelCepDistortion), _ABSDIFF (MAE), _ROOT_SQ (Distortion:
 * per-frame root of the summed squares), _SQDIFF_VOICED (F0Distortion), _SQDIFF_VOICED_EXP (LF0Distortion: on exp of both).
 * target / pred [B,T,D] f32, columns [col0, col0 + width) take part; voiced [B,T] f32 0/1 (voiced kinds); seq_len may be NULL.
 * The masked count is in frames, the unmasked one in elements, as in the reference (metrics.py:383-394). */
#define MG_METRIC_MEAN 0
#define MG_METRIC_SQDIFF 1
#define MG_METRIC_ABSDIFF 2
#define MG_METRIC_ROOT_SQ 3
#define MG_METRIC_SQDIFF_VOICED 4
#define MG_METRIC_SQDIFF_VOICED_EXP 5
size_t mg_metric_workspace_bytes(void);
int mg_metric_accumulate_f32(int kind, const float* target, const float* pred, const float* voiced, const int64_t* seq_len, int B,
                             int T, int D, int col0, int width, double* accum, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K8  maximum-likelihood parameter generation    reference: morgana/viz/synthesis.py:79-178 (MLPG), :8-36 (_build_win_mats),
 *     :39-77 (_build_poe); called per training step from predict() - models/f0_test_model.py:86-89, models/RNN_SPSS.py:107-118
 * ----------------------------------------------------------------------------------------------------------------
 * means f32 [B,T,W*D] (stream column w*D + d = window w of dimension d), variances f32 [W*D] (var_per_frame = 0, the global
 * variance both call sites pass) or [B,T,W*D] (var_per_frame = 1); seq_len int64 [B] or NULL; windows as parallel arrays:
 * win_l / win_u [W] ints (left / right extent, l + u <= 4), win_coeff [W][5] doubles (first l + u + 1 used).  Each utterance is
 * cut to its length and its first / last frame repeated `padding` times (:113-120, :156-157).  Per (b, d) the banded SPD system
 * sum_w W_w^T diag(1/var_w) W_w x = sum_w W_w^T (mean_w/var_w) is solved in float64 (LDL^T on the band); out [B,T,D] (f32, or
 * f64 if out_f64) holds the trajectory, zero past seq_len.  workspace: mg_mlpg_workspace_bytes(...) bytes. */
size_t mg_mlpg_workspace_bytes(int B, int T, int D, int padding, int n_windows, const int* win_l, const int* win_u);
int mg_mlpg_f32(const float* means, const float* variances, int var_per_frame, const int64_t* seq_len, int B, int T, int D,
                int n_windows, const int* win_l, const int* win_u, const double* win_coeff, int padding, void* out, int out_f64,
                void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K5  mvn / minmax normalisers            reference: morgana/data.py:533-538, 579-590
 * ---------------------------------------------------------------------------------------------------------------- */
#define MG_NORM_MVN 0         /* (x - mean) / (std_dev + 1e-8)       p0 = mean, p1 = std_dev */
#define MG_DENORM_MVN 1       /* x * std_dev + mean                                             */
#define MG_NORM_MINMAX 2      /* (x - mmin) / scale,  scale = mmax - mmin, |scale| <= 1e-8 -> 1  p0 = mmin, p1 = mmax */
#define MG_DENORM_MINMAX 3    /* x * scale + mmin                                               */
/* x, out: n_rows x D f32 (any leading batch dims flattened); p0, p1: D f32.  In place (out == x) is allowed. */
/* Device-side collate of one sequence feature (reference: normalise on load, data.py:119-127; zero-padding collate_fn,
 * data.py:159-224; ToDeviceWrapper, :648-663).  packed f32 [sum_b len_b, D]: the batch's utterances back to back;
 * offsets int64 [B+1]: their first rows (offsets[B] = total).  raw_out / norm_out [B,T,D] (either may be NULL): the feature
 * zero padded to T frames and its normalised twin (kind MG_NORM_MVN or MG_NORM_MINMAX), zero in the pad frames. */
int mg_pad_normalise_f32(const float* packed, const int64_t* offsets, int B, int T, int D, const float* p0, const float* p1,
                         int kind, float* raw_out, float* norm_out, void* stream);
/* The same pass with the loader-side half of bf16 mode (reference: the float32 cast on load, data.py:127 - here the bf16 operand of
 * the first Linear is data preparation too): besides raw_out / norm_out (either may be NULL) it writes table_bf16
 * [B*T + extra_rows, ldb] bf16 = the normalised feature's rows (zero in pad frames), columns D .. ldb-1 zero, extra_rows zero rows
 * behind - what mg_cast_pad_bf16 would make of norm_out, without a second pass.  ldb >= D; kind MG_NORM_MVN or MG_NORM_MINMAX. */
int mg_pad_normalise_bf16_f32(const float* packed, const int64_t* offsets, int B, int T, int D, const float* p0, const float* p1, int kind,
                              float* raw_out, float* norm_out, uint16_t* table_bf16, int ldb, int extra_rows, void* stream);
int mg_normalise_f32(const float* x, float* out, const float* p0, const float* p1, int64_t n_rows, int D, int kind,
                     void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K2  Linear (+Sigmoid) stack             reference: nn.Linear / nn.Sigmoid in README.rst:65-73 run by
 *                                         SequentialWithRecurrent.forward, morgana/utils.py:401-418
 * f32 = exact fp32 (v_mfma_f32_32x32x2_f32), the 1e-4 parity mode.
 * bf16 = bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16), the throughput mode.
 * ---------------------------------------------------------------------------------------------------------------- */
/* Y[m,:] = act( A[row(m),:] W^T + bias ).  A [*,K] (lda), rows NULL = identity else gather; W [N,K]; Y [M,N] (ldy). */
int mg_linear_fwd_f32(const float* A, int lda, const int32_t* rows, int64_t M, int K, const float* W,
                      const float* bias, int N, float* Y, int ldy, int act, void* stream);
/* dX = (dY W) [* H (1-H)].  dY [M,N]; W [N,K]; H NULL or [M,K] = the sigmoid OUTPUT feeding this layer; dX [M,K]. */
int mg_linear_dgrad_f32(const float* dY, int64_t M, int N, const float* W, int K, const float* H, float* dX,
                        void* stream);
/* dW[n,k] (+)= sum_m dY[m,n] A[row(m),k];  db[n] (+)= sum_m dY[m,n].  accumulate != 0 adds into dW/db.
 * workspace: mg_linear_wgrad_workspace_bytes(M,N,K).  Deterministic (split-M slabs + ordered reduce). */
size_t mg_linear_wgrad_workspace_bytes(int64_t M, int N, int K);
int mg_linear_wgrad_f32(const float* dY, const float* A, int lda, const int32_t* rows, int64_t M, int N, int K,
                        float* dW, float* db, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* Phone-rate first layer (csrc/phone_rate.hip): Linear commutes with upsample_to_repetitions' row repetition (reference:
 * morgana/utils.py:175-228 feeding README.rst:65-73 / morgana/utils.py:401-418), so the first layer's product runs once per
 * phone and these kernels move between phone rate and frame rate.
 *  mg_segment_bounds  seg_start / seg_end int32 [R]: the run of frames f with rows[f] == r (0, 0 if none).  rows int32 [M] as
 *                     mg_upsample_index writes them (b*P + phone, -1 = padding); the frames of a row must be consecutive.
 *                     rows_mapped (optional, int32 [M]): rows with -1 replaced by pad_row (the table row of a zero input).
 *  mg_segment_sum     out[r, :] = sum of G[f, :] over the frames of row r (fp32 accumulation, frame order), r < R; rows
 *                     R..R+extra-1 of out take the frames with rows[f] < 0 (those of the j-th of `extra` equal shares of the frame axis go to row R + j).  G, out bf16
 *                     (g_bf16) or f32. */
/*  mg_phone_target_stats  what the masked MSE (morgana/losses.py:29-51) needs per table row when every frame of a row shares one
 *                     prediction: weight[r] = sum of the frame weights [t < n_b] / (n_b B) of the row's frames, ybar[r] = their weighted
 *                     mean target, loss_const = sum_r sum_f w_f (y_f - ybar[r])^2.  Then
 *                     loss = sum_r weight[r] (pred[r] - ybar[r])^2 + loss_const  exactly, with the same gradient (mg_f0_tail_rows_bf16).
 *                     target f32 [B*T]; ybar, weight f32 [R + extra]; rows R.. take the padding frames. */
/*  mg_f0_tail_rows_f32  the README F0Model's tail (Linear(128, 32) -> Sigmoid -> Linear(32, 1), /root/reference/README.rst:65-73) with the
 *                     masked MSE in its per-phone form, forward AND backward, exact fp32 (v_mfma_f32_16x16x4_f32), ONE launch + the ordered
 *                     slab reduce - the fp32 / bf16x3 modes' counterpart of mg_f0_tail_rows_bf16 (csrc/tail_f32.hip).  Z2 f32 [M, ldz]: the
 *                     128-wide layer's PRE-activations (h2 = sigmoid(Z2) is taken here); ybar, weight f32 [M] as mg_phone_front leaves them -
 *                     or weight == NULL: the rows are the B x T frames, ybar = the targets [B * T], and the kernel forms the masked MSE's own
 *                     weights [t < n_b] / (n_b B) from seq_len (NULL = all T; an utterance without frames gives NaN, as the reference's mean).
 *                     pred f32 [M]; dZ2 f32 [M, lddz] = d loss / d Z2; grads_out f32 [4164] = dW3 [32 x 128] | db3 [32] | dW4 [32] | db4 |
 *                     loss (= sum_m weight (pred - ybar)^2, the constant term excluded) | 2 unused. */
size_t mg_f0_tail_rows_f32_workspace_bytes(int64_t M);
int mg_f0_tail_rows_f32(const float* Z2, int ldz, const float* W3, const float* b3, const float* W4, const float* b4, const float* ybar,
                        const float* weight, const int64_t* seq_len, int B, int T, int64_t M, float* pred, float* dZ2, int lddz,
                        float* grads_out, void* workspace, size_t workspace_bytes, void* stream);
/*  mg_phone_mse_rows_f32  the masked MSE of a ONE-column prediction that is constant over each table row's frames, exact-fp32 modes:
 *                     loss[0] = sum_r weight[r] (pred[r * ldp] - ybar[r])^2 over the R + extra rows of mg_phone_target_stats (its constant
 *                     term is added by mg_phone_loss_const_add / mg_expand_column_loss_f32), dpred[r] = 2 weight[r] (pred[r * ldp] - ybar[r]).
 *                     One workgroup, fixed summation order (double partial sums). */
int mg_phone_mse_rows_f32(const float* pred, int ldp, const float* ybar, const float* weight, int n_rows, float* loss, float* dpred,
                          void* stream);
/*  mg_expand_column_f32  out[f] = table[rows[f]] for a one-column f32 table (the per-phone prediction repeated to frames); rows >= 0. */
int mg_expand_column_f32(const float* table, const int32_t* rows, int64_t M, float* out, void* stream);
/* ... and mg_phone_loss_const_add in the same launch (stats_workspace as left by mg_phone_target_stats). */
int mg_expand_column_loss_f32(const float* table, const int32_t* rows, int64_t M, float* out, const void* stats_workspace, int R,
                              int extra, float* loss, void* stream);
/* ... and the ordered reduce of the fused tail's slabs in the same launch: dst[0 .. n) = sum of the S slabs (stride floats apart; the
 * arithmetic of the library's slab reduce), dst[n - 1] - the loss, stored behind the gradients - additionally receives the loss's
 * constant term.  Pairs with mg_f0_l2tail_rows_slabs_bf16. */
int mg_expand_column_reduce_f32(const float* table, const int32_t* rows, int64_t M, float* out, const void* stats_workspace, int R,
                                int extra, const float* slab, int64_t n, int64_t stride, int S, float* dst, void* stream);
size_t mg_phone_target_stats_workspace_bytes(int R, int extra);
/* loss_const may be NULL: then mg_phone_loss_const_add(workspace, R, extra, loss) adds the constant to a loss in place later on
 * (the workspace must be left untouched in between). */
int mg_phone_loss_const_add(const void* workspace, int R, int extra, float* loss, void* stream);
int mg_phone_target_stats(const float* target, const int32_t* rows, int64_t M, const int32_t* seg_start, const int32_t* seg_end,
                          const int64_t* seq_len, int B, int T, int R, int extra, float* ybar, float* weight, float* loss_const,
                          void* workspace, size_t workspace_bytes, void* stream);
/* The front of the phone-rate step in ONE launch: mg_upsample_index_maps (upsample_to_repetitions' frame map, morgana/utils.py:175-228)
 * and mg_phone_target_stats (morgana/losses.py:29-51 per phone row) - a phone's statistics read only the frame run its own utterance's
 * duration scan gives, so one workgroup per utterance does both (csrc/phone_front.h).  dur int64 [B, P]; target f32 [B*T] (T = the
 * frame axis the map is cut to); R = B*P phone rows + `extra` rows for the padding frames.  rows32 / rows_mapped int32 [B*T],
 * seg_start / seg_end int32 [R], ybar / weight f32 [R + extra]: bit for bit what the two launches write.  workspace
 * (mg_phone_target_stats_workspace_bytes(R, extra)): partial sums of the loss's constant term for mg_phone_loss_const_add and the
 * mg_expand_column_* launches, grouped per utterance (so the constant may differ from the two-launch one in its last bits).
 * MG_EINVAL where the form does not apply (fewer than ~16 phones per utterance; extra rows that each span many utterances). */
int mg_phone_front(const int64_t* dur, int B, int P, int T, const float* target, const int64_t* seq_len, int extra, int32_t* rows32,
                   int32_t* rows_mapped, int pad_row, int32_t* seg_start, int32_t* seg_end, float* ybar, float* weight, void* workspace,
                   size_t workspace_bytes, void* stream);
/* mg_phone_front and mg_linear_fwd_bf16(A, rows = NULL, ... bf16 output) of the phone table's first layer (README.rst:66-67), which
 * reads nothing the front writes: ONE grid where the GEMM is the persistent 256-wide form and leaves at least 32 CUs without a tile
 * (C2: 168 tiles) - the front's jobs run on those CUs - otherwise the two launches.  Same results either way. */
int mg_phone_front_linear_fwd_bf16(const int64_t* dur, int B, int P, int T, const float* target, const int64_t* seq_len, int extra,
                                   int32_t* rows32, int32_t* rows_mapped, int pad_row, int32_t* seg_start, int32_t* seg_end, float* ybar,
                                   float* weight, void* workspace, size_t workspace_bytes, const uint16_t* A, int lda, int64_t M, int K,
                                   const uint16_t* W, int ldw, const float* bias, int N, uint16_t* Y, int ldy, int act, void* stream);
int mg_segment_bounds(const int32_t* rows, int64_t M, int R, int32_t* seg_start, int32_t* seg_end, int32_t* rows_mapped,
                      int pad_row, void* stream);
int mg_segment_sum(const void* G, int ldg, int g_bf16, const int32_t* rows, int64_t M, const int32_t* seg_start,
                   const int32_t* seg_end, int R, int extra, int N, void* out, int ldo, void* stream);
/* mg_segment_sum (bf16) and, in the same pass over G, the weight gradient of C <= 16 per-frame input features (the frame counters behind
 * the repeated phone rows, /root/reference/models/RNN_SPSS.py:76-81): partial sums  slab[b][c][n] = sum over workgroup b's frames of
 * G[f, n] feat[f, c]  into `slabs` (mg_segment_sum_feat_workspace_bytes(C, ldo) bytes; must stay untouched until the reduce);
 * mg_feat_wgrad_reduce then writes dW[n, col0 + c] (+)= sum_b slab[b][c][n].  feat f32 [M, C].  Deterministic. */
size_t mg_segment_sum_feat_workspace_bytes(int C, int ldo);
int mg_segment_sum_feat_bf16(const uint16_t* G, int ldg, const int32_t* rows, int64_t M, const int32_t* seg_start, const int32_t* seg_end,
                             int R, int extra, int N, uint16_t* out, int ldo, const float* feat, int C, void* slabs, size_t slabs_bytes,
                             void* stream);
int mg_feat_wgrad_reduce(const void* slabs, int C, int ldo, int N, float* dW, int ldw, int col0, int accumulate, void* stream);
/* The first Linear of a model whose input is cat(upsample_to_repetitions(lab, durations), frame counters)
 * (/root/reference/models/RNN_SPSS.py:76-81, models/f0_test_model.py:78-79), with W = [W_lab | W_cnt]: the lab part runs once per phone
 * (P = table W_lab^T through mg_linear_fwd_bf16, f32 output, no bias) and this kernel finishes the layer per frame,
 *   Y[f, n] = act(P[rows[f], n] + sum_c feat[f, c] W[n, col0 + c] + bias[n]),   n < N; Y's padding columns [N, ldy) zero.
 * P f32 [table rows, ldp] (ldp >= N rounded up to 8); rows int32 [M], every entry a row of P (-1 already mapped to a zero-input
 * row); feat f32 [M, C], 1 <= C <= 16; W f32 [N, ldw] with the counters' weights in columns col0 .. col0 + C; Y [M, ldy] bf16, or
 * f32 with y_f32 != 0 (the layer's output leaves the fused run). */
int mg_phone_concat_layer_bf16(const float* P, int ldp, const int32_t* rows, int64_t M, const float* feat, int C, const float* W, int ldw,
                               int col0, const float* bias, int N, int act, void* Y, int ldy, int y_f32, void* stream);

/* bf16 variants: A, W, H, Y are bf16; K and N of the bf16 buffers are padded: lda/ldw/ldy multiples of 8 elements,
 * padding columns must be zero (mg_cast_pad_bf16 / mg_gather_rows_bf16 produce such buffers).  bias, dW, db f32. */
/* Y is bf16 [M,ldy] (y_f32 == 0) or f32 [M,ldy] (y_f32 != 0); ldy a multiple of 8; columns N..ldy-1 are written as 0. */
int mg_linear_fwd_bf16(const uint16_t* A, int lda, const int32_t* rows, int64_t M, int K, const uint16_t* W, int ldw,
                       const float* bias, int N, void* Y, int ldy, int y_f32, int act, void* stream);
/* WT = W^T as bf16 [K,N] (ldwt). */
int mg_linear_dgrad_bf16(const uint16_t* dY, int lddy, int64_t M, int N, const uint16_t* WT, int ldwt, int K,
                         const uint16_t* H, int ldh, void* dX, int lddx, int dx_f32, void* stream);
/* The same with H read from a table: frame m's sigmoid output is H[h_rows[m]] (phone-rate first layer; h_rows >= 0).  Needs the
 * wide-tile shape (M >= 2048, K % 128 == 0, lddy and ldwt multiples of 64); MG_EINVAL otherwise. */
int mg_linear_dgrad_gathered_bf16(const uint16_t* dY, int lddy, int64_t M, int N, const uint16_t* WT, int ldwt, int K,
                                  const uint16_t* H, int ldh, const int32_t* h_rows, void* dX, int lddx, int dx_f32, void* stream);
int mg_linear_wgrad_bf16(const uint16_t* dY, int lddy, const uint16_t* A, int lda, const int32_t* rows, int64_t M,
                         int N, int K, float* dW, float* db, int accumulate, void* workspace, size_t workspace_bytes,
                         void* stream);

/* mg_linear_wgrad_bf16 with both operands gathered: dW = sum_{m < M} dY[dy_rows[m]]^T A[rows[m]] (rows NULL: A[dy_rows[m]]); M counts the
 * index pairs.  What it replaces: the weight gradients autograd forms for nn.GRU / nn.LSTM on a PackedSequence
 * (morgana/utils.py:366-385: pack_padded_sequence in front of the layer, pad_packed_sequence behind it) - here the recurrences keep
 * padded (B, T) arrays and this product visits the sum_b T_b valid frames only.  Wide-tile shapes with lda == 512 (384 < K <= 512,
 * N % 128 == 0, M >= 4096, not the half-width plan): MG_EINVAL otherwise, the caller then multiplies the padded rows.  workspace:
 * mg_linear_wgrad_workspace_bytes(M, N, K). */
int mg_linear_wgrad_rows_bf16(const uint16_t* dY, int lddy, const int32_t* dy_rows, const uint16_t* A, int lda, const int32_t* rows, int64_t M,
                              int N, int K, float* dW, float* db, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* mg_linear_wgrad_bf16 without its reduce launch: the split-M partial results stay in `workspace` as *n_slabs slabs of *stride floats,
 * slab s = [N*K weight partials | N bias partials], for a consumer that sums them itself (mg_adam_step_plan_f32).  Only the wide-tile
 * plan has this form: returns MG_EINVAL for shapes mg_linear_wgrad_bf16 would run on its 128 x 128 kernels (call that instead). */
int mg_linear_wgrad_slabs_bf16(const uint16_t* dY, int lddy, const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K,
                               void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream);
/* mg_linear_wgrad_slabs_bf16 (no gather) and mg_linear_dgrad_bf16 of ONE layer whose input A bf16 [M, lda] is the output of the Sigmoid
 * below it (reference: the autograd backward of nn.Linear + nn.Sigmoid inside F0Model, README.rst (the nn.Sequential of Linear + Sigmoid layers)): the slabs of
 * dW = dY^T A, db, and dX bf16 [M, lddx] = (dY W) * A (1 - A), WT = W^T bf16 [K, ldwt].  The two products are independent; for the
 * phone-rate shape of the README model (N = 128, K = lda = lddx = 512, the tiles of both within one wave of workgroups) they run as
 * ONE grid, with the split count cut to what the dgrad tiles leave free per XCD (*n_slabs says how many); other shapes run the two
 * launches.  Needs a wide-tile weight-gradient shape (as mg_linear_wgrad_slabs_bf16: MG_EINVAL otherwise); workspace as there. */
int mg_linear_wgrad_dgrad_bf16(const uint16_t* dY, int lddy, const uint16_t* A, int lda, int64_t M, int N, int K, const uint16_t* WT, int ldwt,
                               uint16_t* dX, int lddx, void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream);
/* mg_linear_wgrad_dgrad_bf16 with mg_expand_column_reduce_f32 (K1 section) riding at the end of the same grid: the phone-rate step's
 * repeated prediction (out[f] = table[rows[f]], `frames` of them) and the ordered sum of the fused tail's slabs into tail_dst (+ the
 * loss's constant term from stats_workspace / R / extra) are read by nothing before the update, so a step captured as a HIP graph
 * lets them start on the CUs the dgrad tiles free first instead of paying a launch of their own.  Arguments as the two entry points;
 * the same results bit for bit; shapes the one-grid form does not take run the separate launches.  loss_only != 0: of the slab sum
 * only the last 16-element chunk - the one that holds the loss at tail_dst[tail_n - 1] - is formed (the caller's update kernel sums
 * the slabs of the gradients itself: mg_adam_step_plan_f32 with the tail's slabs as a source). */
int mg_linear_wgrad_dgrad_expand_bf16(const uint16_t* dY, int lddy, const uint16_t* A, int lda, int64_t M, int N, int K, const uint16_t* WT,
                                      int ldwt, uint16_t* dX, int lddx, void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride,
                                      const float* table, const int32_t* rows, int64_t frames, float* out, const void* stats_workspace, int R,
                                      int extra, const float* tail_slab, int64_t tail_n, int64_t tail_stride, int tail_S, float* tail_dst,
                                      int loss_only, void* stream);
/* dst[0 .. count) (+)= the ordered sum of n_slabs slabs, `stride` floats apart (the reduce launch of mg_linear_wgrad_bf16, bit for bit),
 * for a caller that took slabs from mg_linear_wgrad_slabs_bf16 / mg_linear_wgrad_dgrad_bf16 and needs the finished gradient before the
 * update (a data-parallel rank: its all-reduce comes first).  With db stored right behind dW (count = N*K + N) one launch does both. */
int mg_slab_reduce_f32(const float* slab, int n_slabs, int64_t stride, int64_t count, float* dst, int accumulate, void* stream);
/* Fused backward of Linear(K -> N) + Sigmoid feeding Linear(N -> N2): dW, db of the FIRST layer straight from dZ2, the
 * pre-activation gradient of the second one, without materialising dZ1 = (dZ2 W2) * H1 (1 - H1):
 *   dZ2 bf16 [M, lddz] (N2 = 128 columns); W2T = W2^T bf16 [N, ldwt]; H1 bf16 [M, ldh] (sigmoid outputs, N % 128 == 0);
 *   A bf16 [*, lda = 640] with `rows` (the layer-1 input, gathered), 512 < K <= 608.
 * dW f32 [N, K], db f32 [N] (accumulate != 0 adds).  workspace: mg_linear_bwd_fused_workspace_bytes(M, N, K). */
size_t mg_linear_bwd_fused_workspace_bytes(int64_t M, int N, int K);
int mg_linear_bwd_fused_bf16(const uint16_t* dZ2, int lddz, int N2, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh,
                             const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K, float* dW, float* db,
                             int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* The same without its reduce launch: the split-M partial results stay in `workspace` as *n_slabs slabs of *stride floats, slab s =
 * [N*K weight partials | N bias partials] (the layout of mg_linear_wgrad_slabs_bf16), for mg_adam_step_plan_f32 to sum. */
int mg_linear_bwd_fused_slabs_bf16(const uint16_t* dZ2, int lddz, int N2, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh,
                                   const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K, void* workspace,
                                   size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream);
/* The fused backward WITH the second layer's weight gradient: one launch leaves the split-M slabs of dW1 | db1 (as
 * mg_linear_bwd_fused_slabs_bf16) AND of dW2 = dZ2^T H1 | db2 = column sums of dZ2 - the stand-alone weight-gradient launch of the
 * N -> 128 layer, which re-read H1 and dZ2 from HBM, goes (reference: autograd of README.rst:65-73, the mm + sum of layer 2's
 * backward).  Needs the row map (rows != NULL).  workspace (floats): [*n_slabs x *stride1] first-layer slabs as above, then from
 * float *offset2 on [*n_slabs x *stride2] second-layer slabs, each [128 x N weight partials | 128 bias partials]; the caller sums
 * them in order (mg_slab_reduce_f32 or the update kernel's plan). */
size_t mg_linear_bwd_fused2_workspace_bytes(int64_t M, int N, int K);
int mg_linear_bwd_fused2_slabs_bf16(const uint16_t* dZ2, int lddz, int N2, const uint16_t* W2T, int ldwt, const uint16_t* H1, int ldh,
                                    const uint16_t* A, int lda, const int32_t* rows, int64_t M, int N, int K, void* workspace,
                                    size_t workspace_bytes, int* n_slabs, int64_t* stride1, int64_t* offset2, int64_t* stride2, void* stream);

/* dst[r, 0:cols] = bf16(src[r, 0:cols]), dst[r, cols:ldd] = 0.  src f32 [rows, cols] (lds). */
int mg_cast_pad_bf16(const float* src, int lds, uint16_t* dst, int ldd, int64_t rows, int cols, void* stream);
/* dst[c, 0:rows] = bf16(src[r, c]) transposed, dst [cols, ldd], zero padded. */
int mg_cast_transpose_bf16(const float* src, int lds, uint16_t* dst, int ldd, int rows, int cols, void* stream);
/* Batched weight refresh (one launch for a whole layer stack): for each descriptor, dst = bf16(src) [rows, ldd] and / or
 * dst_t = bf16(src^T) [cols, ldt], zero padded; either output may be NULL.  `descs` is a HOST array, count <= MG_CAST_MAX. */
#define MG_CAST_MAX 16
typedef struct {
    const float* src; /* device, fp32 [rows, cols], contiguous */
    int rows, cols;
    uint16_t* dst;    /* device bf16 [rows, ldd] or NULL */
    int ldd;
    uint16_t* dst_t;  /* device bf16 [cols, ldt] or NULL */
    int ldt;
} mg_cast_desc;
int mg_cast_params_bf16(const mg_cast_desc* descs, int count, void* stream);
/* Up to MG_COPY_MAX device-to-device copies as ONE launch (dst[i][0 .. bytes[i]) = src[i][...], buffers 16-byte aligned or any size
 * handled bytewise at the ends): a captured training step replayed on a NEW batch takes the batch's tensors into its static buffers
 * this way - the feature dict of /root/reference/morgana/data.py:648-663 is five tensors at BASELINE config C2, five copy launches of
 * 3-15 us each where one does (graphs.GraphedTrainStep.load).  `descs` is a HOST array. */
#define MG_COPY_MAX 16
typedef struct {
    const void* src;
    void* dst;
    int64_t bytes;
} mg_copy_desc;
int mg_copy_many(const mg_copy_desc* descs, int count, void* stream);
/* HOST half of the loader (no device work): `count` utterance arrays (srcs[i], bytes[i]) laid back to back at dst - the pinned staging
 * buffer one H2D copy then takes across PCIe - by `threads` host threads (1..64).  Replaces the np / torch concatenation in front of
 * /root/reference/morgana/data.py:159-224 (collate_fn's per-utterance pad copies; there the padding happens on the host, here on the
 * device: mg_pad_normalise_f32).  dst_bytes = capacity of dst; MG_EINVAL when the pieces do not fit. */
int mg_host_pack(const void* const* srcs, const int64_t* bytes, int count, void* dst, int64_t dst_bytes, int threads);
/* Operand splits of precision mode 'bf16x3' (split-bf16: hi = bf16(x), lo = bf16(x - hi); x w ~= hi hi + hi lo + lo hi as ONE bf16
 * GEMM over a contraction index three times as long; csrc/split3.hip).  The reference computes these products in fp32
 * (morgana/experiment_builder.py:262-263: no autocast, morgana/data.py:127: float32 features); this mode reproduces them to ~1e-5
 * at three bf16 MFMA products each.  For each descriptor: dst bf16 [rows, 3 ldp] = three planes of ldp columns (zero padded):
 * order 0 -> [hi | hi | lo] (the activation side of a product), order 1 -> [hi | lo | hi] (the weight side); transpose != 0 -> the
 * planes hold the split of src^T: dst [cols, 3 ldp] with ldp >= rows.  order 2 (not transposed) -> dst bf16 [2, rows, ldp]: the hi
 * plane, then the lo plane, each a matrix of its own (the operands of a weight gradient, which contracts over the rows: three
 * accumulating launches of mg_linear_wgrad_bf16 on plane pairs).  order 3 / 4 (not transposed) -> dst bf16 [3, rows, ldp]: three
 * row-stacked planes [hi ; hi ; lo] / [hi ; lo ; hi] - the same three products as ONE mg_linear_wgrad_bf16 launch over 3 rows rows
 * (dY = the order-3 planes, A = the order-4 planes); its bias gradient comes from `colsum`.  `descs` is a HOST array, count <= MG_SPLIT3_MAX. */
#define MG_SPLIT3_MAX 16
typedef struct {
    const float* src; /* device, fp32 [rows, cols], row stride lds */
    int64_t rows;
    int cols, lds;
    uint16_t* dst;    /* device, bf16, 16-byte aligned: [rows, 3 ldp], or [cols, 3 ldp] when transposed */
    int ldp;          /* columns per plane: multiple of 8, >= cols (>= rows when transposed) */
    int order;        /* 0: hi | hi | lo;  1: hi | lo | hi;  2: two planes [hi ; lo] of [plane_rows, ldp];  3: [hi ; hi ; lo];  4: [hi ; lo ; hi];
                       * 5: PAIR PLANES hi | lo, dst bf16 [rows, 2 ldp] ([cols, 2 ldp] with transpose): the operands of the mg_*_x3 entry points */
    int transpose;
    int64_t plane_rows; /* orders 2-4: rows of one plane in dst (>= rows; the caller owns the rows behind the split, e.g. zeros); 0 = rows */
    const float* sig; /* optional (not transposed): fp32 [rows, cols] (ldsig) sigmoid outputs s - the split is taken of src * s * (1 - s):
                       * the sigmoid gradient of autograd (README.rst:65-73's nn.Sigmoid) fused into the split of the gradient */
    int ldsig;
    float* colsum;    /* optional (not transposed): f32 [colsum_blocks, ldp] - workgroup b's partial column sums of the (sigmoid-gradient-fused)
                       * source values over its share of the rows; summed in slab order they are the bias gradient of the layer, exact */
    int colsum_blocks;
} mg_split3_desc;
int mg_split3_bf16(const mg_split3_desc* descs, int count, void* stream);

/* ---- The FUSED step of precision mode 'bf16x3' on PAIR PLANES -------------------------------------------------------------------
 * Reference arithmetic is fp32 end to end (/root/reference/morgana/experiment_builder.py:262-263, /root/reference/morgana/data.py:127);
 * x = hi + lo with hi = bf16(x), lo = bf16(x - hi), x w ~= hi hi + hi lo + lo hi, every product exact in the fp32 accumulator of a bf16
 * MFMA.  A pair-plane operand is bf16 [rows, ld] with ld = 2 ldp: columns [0, ldp) = hi, [ldp, 2 ldp) = lo (ldp a multiple of 64,
 * padding columns zero) - mg_split3_bf16 order 5 writes one, the update kernel keeps the weights' pairs current (mg_adam_shadow.pair),
 * and the entry points below write their outputs split in their epilogues, so a training step of the README F0Model at phone rate
 * (/root/reference/README.rst:65-73 behind /root/reference/morgana/utils.py:175-228) launches no split pass and no cast.  The tile programs
 * are those of bf16 mode with the contraction run three times over the plane pairs (csrc/gemm_bf16_big.hip, "pair planes").
 *
 *  mg_phone_front_linear_fwd_x3   mg_phone_front + Y = split(act(A W^T + b)): A [M, lda = 2 pa] the phone table's pair, W [N, ldw = 2 pw]
 *                     (pa, pw >= K rounded up to 64; N a multiple of 256), Y bf16 [M, ldy = 2 N] the activation's pair.  One grid where
 *                     the GEMM leaves CUs idle (as mg_phone_front_linear_fwd_bf16), else two launches.  dur == NULL: the GEMM alone.
 *  mg_linear_fwd_x3_f32  Y f32 [M, N] = act(A W^T + b) from pair-plane operands (N a multiple of 128, ldy == N).  parts == 3 (act none):
 *                     Y is [3, M, N] - the partial sums of the three products hi hi (+ b), hi lo, lo hi from three sets of workgroups
 *                     (a few-tile GEMM fills the chip with chains a third as long); the consumer adds them (mg_f0_tail_rows_x3 z_parts).
 *  mg_f0_tail_rows_x3  mg_f0_tail_rows_f32 with Z2 f32 [z_parts, M, ldz] given as z_parts (1 or 3) partial sums to add, dZ2 as a pair
 *                     [M, lddz >= 256] and the 128-wide layer's bias gradient (the column sums of
 *                     the fp32 dZ2) in front of the tail's gradients: slabs of MG_F0_TAIL_X3_SLAB floats = db2 [128] | dW3 [32 x 128] |
 *                     db3 [32] | dW4 [32] | db4 | loss | 2 unused, *n_slabs of them *stride floats apart in `workspace`
 *                     (mg_f0_tail_rows_x3_workspace_bytes(M)), reduced into grads_out [MG_F0_TAIL_X3_SLAB] when that is not NULL.
 *  mg_f0_l2tail_x3    mg_linear_fwd_x3_f32 of the 512 -> 128 layer AND mg_f0_tail_rows_x3 as ONE launch (csrc/l2tail_x3.hip): a workgroup
 *                     owns <= 96 consecutive rows, streams its rows of H1 (pair [M, 1024], whole lines, once) and W2's pair [128, 1024]
 *                     through LDS, keeps Z2 = H1 W2^T + b2 on chip and runs the exact-fp32 tail on it.  Outputs and slab layout as
 *                     mg_f0_tail_rows_x3 (workspace mg_f0_l2tail_x3_workspace_bytes(M)).
 *  mg_linear_wgrad_dgrad_x3  the 512 -> 128 layer's backward as ONE grid: dW slabs (N K floats used of each `*stride`, no bias sums) from
 *                     dY pair [M, lddy = 2 * 128] and A = H pair [M, lda = 2 * 512], and dX pair [M, lddx = 2 K] = split((dY W) * H (1 - H)),
 *                     H = hi + lo; `colsum` f32 receives *n_colsum slabs of K floats whose ordered sum is the column sum of the fp32 dX
 *                     (the bias gradient of the layer below; 2 ceil(M / 256) slabs: mg_linear_wgrad_dgrad_x3_colsum_floats(M, K) floats).
 *                     WT [K, ldwt = 2 * 128] = the pair of W^T.  N == 128, K == 512 only (MG_EINVAL otherwise).
 *  mg_linear_wgrad_slabs_x3  mg_linear_wgrad_slabs_bf16 on pairs: dY [M, lddy = 2 pn], A [M, lda = 2 pk], slabs of N K floats (+ N unused,
 *                     the bf16 layout's stride), no bias sums; the wide-tile shapes only (pk 640 or 512). */
#define MG_F0_TAIL_X3_SLAB 4292
int mg_phone_front_linear_fwd_x3(const int64_t* dur, int B, int P, int T, const float* target, const int64_t* seq_len, int extra,
                                 int32_t* rows32, int32_t* rows_mapped, int pad_row, int32_t* seg_start, int32_t* seg_end, float* ybar,
                                 float* weight, void* workspace, size_t workspace_bytes, const uint16_t* A, int lda, int64_t M, int K,
                                 const uint16_t* W, int ldw, const float* bias, int N, uint16_t* Y, int ldy, int act, void* stream);
int mg_linear_fwd_x3_f32(const uint16_t* A, int lda, int64_t M, int K, const uint16_t* W, int ldw, const float* bias, int N, float* Y,
                         int ldy, int act, int parts, void* stream);
size_t mg_f0_tail_rows_x3_workspace_bytes(int64_t M);
int mg_f0_tail_rows_x3(const float* Z2, int ldz, int z_parts, const float* W3, const float* b3, const float* W4, const float* b4,
                       const float* ybar, const float* weight, int64_t M, float* pred, uint16_t* dZ2, int lddz, float* grads_out,
                       void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream);
size_t mg_f0_l2tail_x3_workspace_bytes(int64_t M);
int mg_f0_l2tail_x3(const uint16_t* H1, int ldh, const uint16_t* W2, int ldw, const float* b2, const float* W3, const float* b3,
                    const float* W4, const float* b4, const float* ybar, const float* weight, int64_t M, float* pred, uint16_t* dZ2,
                    int lddz, float* grads_out, void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream);
size_t mg_linear_wgrad_dgrad_x3_colsum_floats(int64_t M, int K);
int mg_linear_wgrad_dgrad_x3(const uint16_t* dY, int lddy, const uint16_t* A, int lda, int64_t M, int N, int K, const uint16_t* WT, int ldwt,
                             uint16_t* dX, int lddx, void* workspace, size_t workspace_bytes, int* n_slabs, int64_t* stride, float* colsum,
                             size_t colsum_floats, int* n_colsum, void* stream);
int mg_linear_wgrad_slabs_x3(const uint16_t* dY, int lddy, const uint16_t* A, int lda, int64_t M, int N, int K, void* workspace,
                             size_t workspace_bytes, int* n_slabs, int64_t* stride, void* stream);
/* Active dropout (nn.Dropout(p) of the shipped models in training mode: /root/reference/models/RNN_SPSS.py:19,34,40,
 * /root/reference/models/f0_test_model.py:22,31-43; torch.nn.functional.dropout semantics: y = x * keep / (1 - p), keep ~ Bernoulli(1 - p)
 * per element).  The mask is a function of (seed, site, *counter, element index) through Philox4x32-10 and is never stored: the
 * backward pass calls the same entry on the gradient with the same four numbers.  x, y: n elements, fp32 (bf16 = 0) or bf16 (bf16 = 1),
 * y may be x.  counter: DEVICE uint64 (NULL = 0) - the step counter this call draws from; mg_dropout_advance(state, used) copies
 * *state to *used (the word a call's backward keeps) and increments *state, in stream order, so that a replayed HIP graph draws a new
 * mask every replay.  Bit parity with torch's mask stream is not attempted (its Philox offsets depend on its launch geometry). */
int mg_dropout(const void* x, void* y, int64_t n, int bf16, float p, uint64_t seed, uint32_t site, const uint64_t* counter, void* stream);
int mg_dropout_advance(uint64_t* state, uint64_t* used, void* stream);
/* Host-only: one Philox4x32-10 block (the generator mg_dropout draws from), for known-answer tests. */
void mg_philox4x32_10(const uint32_t counter[4], const uint32_t key[2], uint32_t out[4]);
/* CALIBRATION (a measurement entry; no training step calls it): one launch of a register-operand bf16 MFMA loop - n_workgroups x 512
 * threads (two waves per SIMD), every wave issues 16 x trips v_mfma_f32_16x16x32_bf16 on operands read once from `operands`
 * (bf16, at least 4096 x 64 values: the caller chooses the data, e.g. random) and writes one float per thread to `sink`
 * (n_workgroups x 512 floats).  *flop (host, optional) receives the launch's FLOP count; the caller times it with events.
 * Used by bench.py for `frac_of_sustained`: what the matrix pipe of THIS chip holds under its power limit. */
int mg_calib_mfma_bf16(const uint16_t* operands, float* sink, int n_workgroups, int trips, double* flop, void* stream);
/* dst f32 [rows, cols] = src bf16 [rows, cols] (lds). */
int mg_cast_bf16_f32(const uint16_t* src, int lds, float* dst, int ldd, int64_t rows, int cols, void* stream);
/* elementwise sigmoid forward / backward for a stand-alone nn.Sigmoid. */
int mg_sigmoid_f32(const float* x, float* y, int64_t n, void* stream);
int mg_sigmoid_grad_f32(const float* dy, const float* y, float* dx, int64_t n, void* stream);

/* Fused tail of a Linear/Sigmoid stack ending in ... -> 128 -> 32 -> 1 under the masked MSE (the README F0Model's layers 3-4,
 * README.rst:65-73, with morgana/losses.py:29-51), bf16 mode: forward of both layers, the loss, and the whole backward
 * through both, in one pass over H2.
 *   H2 bf16 [B*T, ldh] (the 128 sigmoid outputs of the previous layer); W3 f32 [32,128], b3 [32], W4 f32 [1,32], b4 [1];
 *   target f32 [B*T] (D = 1); seq_len int64 [B] or NULL; grad_scale multiplies dL/dpred.
 * Outputs: pred f32 [B*T]; loss f32 [1]; dZ2 bf16 [B*T, ldh] = dL/d(pre-activation of the 128-wide layer);
 *   grads f32 [32*128 + 32 + 32 + 1] = dW3 | db3 | dW4 | db4 (accumulate != 0 adds into it).  If `loss` points at the
 *   float right behind `grads` (and accumulate == 0) both are finished by a single reduce launch.
 * workspace: mg_f0_tail_workspace_bytes(B*T).  Deterministic. */
size_t mg_f0_tail_workspace_bytes(int64_t M);
int mg_f0_tail_bf16(const uint16_t* H2, int ldh, int K3, const float* W3, const float* b3, const float* W4, const float* b4,
                    const float* target, const int64_t* seq_len, int B, int T, float grad_scale, float* pred, float* loss,
                    uint16_t* dZ2, float* grads, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* The tail on M rows that each stand for a group of frames sharing one input row (phone-rate step): loss = sum_m row_weight[m]
 * (pred[m] - target[m])^2 and its whole backward; target / row_weight from mg_phone_target_stats. */
int mg_f0_tail_rows_bf16(const uint16_t* H2, int ldh, int K3, const float* W3, const float* b3, const float* W4, const float* b4,
                         const float* target, const float* row_weight, int64_t M, float grad_scale, float* pred, float* loss,
                         uint16_t* dZ2, float* grads, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* The 512 -> 128 Linear + Sigmoid in front of that tail fused INTO it (README.rst:65-73 layers 2-4 + morgana/losses.py:29-51):
 * one pass over H1, the 128-wide activation H2 never reaches memory (the backward of layer 2 needs dZ2 and H1 only).
 *   H1 bf16 [B*T, ldh1] (the 512 sigmoid outputs of the first layer); W2 bf16 [128, ldw2] (the optimiser's bf16 copy of the
 *   weight, mg_cast_params_bf16 / mg_adam_step_plan_f32), b2 f32 [128]; everything else as mg_f0_tail_bf16.
 * Outputs as mg_f0_tail_bf16, with dZ2 bf16 [B*T, lddz].  Same arithmetic per frame as mg_linear_fwd_bf16 + mg_f0_tail_bf16
 * up to the summation order inside the 512-deep dot products.  workspace: mg_f0_l2tail_workspace_bytes(B*T).  Deterministic. */
size_t mg_f0_l2tail_workspace_bytes(int64_t M);
int mg_f0_l2tail_bf16(const uint16_t* H1, int ldh1, int K2, const uint16_t* W2, int ldw2, int N2, const float* b2, const float* W3,
                      const float* b3, const float* W4, const float* b4, const float* target, const int64_t* seq_len, int B, int T,
                      float grad_scale, float* pred, float* loss, uint16_t* dZ2, int lddz, float* grads, int accumulate, void* workspace,
                      size_t workspace_bytes, void* stream);
/* mg_f0_l2tail_bf16 without its reduce launch (as mg_f0_l2tail_rows_slabs_bf16 below): the slabs stay in `workspace` for the update
 * kernel's plan - gradients as a slab source, the loss through mg_adam_tail.  For a step captured whole into a HIP graph. */
int mg_f0_l2tail_slabs_bf16(const uint16_t* H1, int ldh1, int K2, const uint16_t* W2, int ldw2, int N2, const float* b2, const float* W3,
                            const float* b3, const float* W4, const float* b4, const float* target, const int64_t* seq_len, int B, int T,
                            float grad_scale, float* pred, uint16_t* dZ2, int lddz, void* workspace, size_t workspace_bytes, int* n_slabs,
                            void* stream);
/* ... on M rows that each stand for a group of frames (phone-rate step), as mg_f0_tail_rows_bf16. */
int mg_f0_l2tail_rows_bf16(const uint16_t* H1, int ldh1, int K2, const uint16_t* W2, int ldw2, int N2, const float* b2, const float* W3,
                           const float* b3, const float* W4, const float* b4, const float* target, const float* row_weight, int64_t M,
                           float grad_scale, float* pred, float* loss, uint16_t* dZ2, int lddz, float* grads, int accumulate,
                           void* workspace, size_t workspace_bytes, void* stream);
/* mg_f0_l2tail_rows_bf16 without its reduce launch: the workgroups' sums (32*128 + 32 + 32 + 2 floats each: dW3 | db3 | dW4 | db4 | loss)
 * stay at the start of `workspace`, *n_slabs of them, mg_f0_l2tail_slab_stride() floats apart, for mg_expand_column_reduce_f32 (or the
 * update kernel's plan) to sum. */
int64_t mg_f0_l2tail_slab_stride(void);
int mg_f0_l2tail_rows_slabs_bf16(const uint16_t* H1, int ldh1, int K2, const uint16_t* W2, int ldw2, int N2, const float* b2, const float* W3,
                                 const float* b3, const float* W4, const float* b4, const float* target, const float* row_weight, int64_t M,
                                 float grad_scale, float* pred, uint16_t* dZ2, int lddz, void* workspace, size_t workspace_bytes, int* n_slabs,
                                 void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * K3  GRU through RecurrentCuDNNWrapper   reference: morgana/utils.py:345-393 + torch.nn.GRU (gates r, z, n)
 * The sort / pack / unpack of the reference only restricts item b to its first seq_len[b] steps; here that is a
 * per-item length mask, outputs beyond the length are exactly 0 and h_n is the state at the last valid step.
 * ---------------------------------------------------------------------------------------------------------------- */
/* Forward recurrence, one launch per step.
 *   xproj  [B,T,3H] = x W_ih^T + b_ih (computed with mg_linear_fwd_*); w_hh [3H,H]; b_hh [3H]; seq_len NULL or int64 [B]
 *   hstate [B,T+1,H]: slot 0 must hold h0 (or zeros) on entry; slot t+1 receives the state after step t (frozen once
 *                     t >= seq_len[b]), so h_n = hstate[:, T, :] and h_{t-1} = hstate[:, t, :] for the backward
 *   out    [B,T,H]   = h_t on valid steps, exactly 0 on padded steps (pad_packed_sequence, utils.py:383)
 *   saved  [B,T,4H]  = (r, z, n, W_hn h + b_hn) for the backward
 * CONTRACT for steps past an item's length (t >= seq_len[b]), all GRU / LSTM entry points of this header:
 *   - out is exactly 0 there and hstate repeats the last valid state (both specified, both tested);
 *   - saved[b,t,:] is UNSPECIFIED there.  The forward kernels differ in what they leave: the per-step kernels (mg_gru_fwd_f32,
 *     mg_gru_fwd_bf16, mg_lstm_fwd_f32) store the gate values the frozen state and whatever xproj holds at that position produce;
 *     the workgroup-local kernels (mg_gru_fwd_small_f32) do the same; the persistent kernels (mg_gru_fwd_persist_*,
 *     mg_lstm_fwd_persist_bf16, mg_lstm_pstack_fwd_bf16) do so up to the longest sequence of the item's group and write zeros or
 *     nothing at all beyond it.  Callers must not compare or consume these elements;
 *   - every backward entry point treats such a step as inactive: the loads of saved[b,t,:] and grad_out[b,t,:] may be issued, but
 *     their values are discarded by a select (never multiplied by a 0/1 mask), so NaN or garbage there cannot reach dxproj,
 *     dhproj, dgates, dh0 / dc0 or any shadow; the gate gradients written for such a step are exactly 0.
 *     tests/test_gpu_configs.py::test_gru_backward_never_reads_saved_past_seq_len poisons these elements with NaN in all five
 *     backward forms and requires bit-identical, finite gradients. */
int mg_gru_fwd_f32(const float* xproj, const float* w_hh, const float* b_hh, const int64_t* seq_len, int B, int T, int H,
                   float* hstate, float* out, float* saved, void* stream);
/* BPTT, one launch per step.  grad_out [B,T,H]; grad_hn NULL or [B,H] (gradient of h_n).
 * Produces dxproj [B,T,3H] (= dL/d xproj: feed to wgrad / dgrad of W_ih), dhproj [B,T,3H] (= dL/d(h W_hh^T + b_hh):
 * dW_hh = dhproj^T h_prev with h_prev rows = hstate[:, t, :], db_hh = column sums) and dh0 [B,H].
 * workspace: mg_gru_bwd_workspace_bytes(B,H). */
size_t mg_gru_bwd_workspace_bytes(int B, int H);
int mg_gru_bwd_f32(const float* grad_out, const float* grad_hn, const float* hstate, const float* saved, const float* w_hh,
                   const int64_t* seq_len, int B, int T, int H, float* dxproj, float* dhproj, float* dh0, void* workspace,
                   size_t workspace_bytes, void* stream);

/* Small hidden sizes (H = 64 or 128, e.g. the GRU-64 layers of models/f0_test_model.py): the same recurrence, exact fp32, as ONE
 * launch per direction in which a workgroup owns 256 / H items outright (W_hh in its registers, the state in its LDS, no
 * hand-off between workgroups).  Same arguments and results as mg_gru_fwd_f32 / mg_gru_bwd_f32 (no workspace); sums over the
 * contraction in a different order (equal to fp32 rounding).  mg_gru_fwd_f32 / mg_gru_bwd_f32 route here themselves when
 * mg_gru_small_supported(H) != 0. */
int mg_gru_small_supported(int H);
int mg_gru_fwd_small_f32(const float* xproj, const float* w_hh, const float* b_hh, const int64_t* seq_len, int B, int T, int H,
                         float* hstate, float* out, float* saved, void* stream);
int mg_gru_bwd_small_f32(const float* grad_out, const float* grad_hn, const float* hstate, const float* saved, const float* w_hh,
                         const int64_t* seq_len, int B, int T, int H, float* dxproj, float* dhproj, float* dh0, void* stream);

/* fp32 parity mode as ONE launch per direction for 256 <= H <= 512, H % 64 == 0, B <= 128: the group / slot scheme of the bf16
 * persistent kernels below with fp32 hand-off tiles and the exact-fp32 MFMA in the per-step kernels' block order, the same cell
 * code - results bit-identical to mg_gru_fwd_f32 / mg_gru_bwd_f32 on the live steps (gate values of steps beyond an item
 * group's longest sequence are written as zeros).  Workspace, status word and residency: as for mg_gru_fwd_persist_bf16. */
int mg_gru_persist_f32_supported(int B, int T, int H);
int mg_gru_fwd_persist_f32(const float* xproj, const float* w_hh, const float* b_hh, const int64_t* seq_len, int B, int T, int H,
                           float* hstate, float* out, float* saved, void* workspace, size_t workspace_bytes, void* stream);
int mg_gru_bwd_persist_f32(const float* grad_out, const float* grad_hn, const float* hstate, const float* saved, const float* w_hh,
                           const int64_t* seq_len, int B, int T, int H, float* dxproj, float* dhproj, float* dh0, void* workspace,
                           size_t workspace_bytes, void* stream);

/* The same recurrence with bf16 matmul operands (throughput mode; cell arithmetic, states, gate values and all outputs stay
 * f32): w_hh_bf = bf16(W_hh) [3H, ldw]; hstate_bf [B,T+1,H] is a bf16 shadow of hstate that the caller initialises at slot 0 and
 * the kernel extends; backward takes w_hh_t_bf = bf16(W_hh^T) [H, ldt >= 3H] and fills dhproj_bf [B,T,3H], the shadow of dhproj.
 * Needs H % 128 == 0 (H <= 1024). */
int mg_gru_fwd_bf16(const float* xproj, const uint16_t* w_hh_bf, int ldw, const float* b_hh, const int64_t* seq_len, int B, int T,
                    int H, float* hstate, uint16_t* hstate_bf, float* out, float* saved, void* stream);
int mg_gru_bwd_bf16(const float* grad_out, const float* grad_hn, const float* hstate, const float* saved,
                    const uint16_t* w_hh_t_bf, int ldt, const int64_t* seq_len, int B, int T, int H, float* dxproj, float* dhproj,
                    uint16_t* dhproj_bf, float* dh0, void* workspace, size_t workspace_bytes, void* stream);

/* The bf16-operand recurrence as ONE launch per direction (persistent kernel: W_hh in registers for all T steps, the state
 * handed between workgroups through write-through stores and per-slot flags; the batch is cut into 8 independent groups).
 * Same arguments and results as mg_gru_fwd_bf16 / mg_gru_bwd_bf16 plus a workspace of mg_gru_persist_workspace_bytes(B, H) holding the
 * hand-off ring, the flags and a status word.  mg_gru_persist_supported(B, T, H) != 0 says the shape is covered (H % 128 == 0, H <= 512,
 * B <= 256, buffers < 2 GiB); other shapes return MG_EINVAL - use the per-step entry points.  All workgroups of the launch
 * must be resident together (at most 256 of 256 threads): the calling stream must own the device.  The workspace (16-byte
 * aligned, zeroed ONCE by the caller after allocating it) may be reused by later launches on the same stream: each launch resets
 * the flags itself, the status word behind them is sticky.  Every wait in the kernel is bounded;
 * mg_gru_persist_status(workspace, stream) synchronises the stream, returns MG_ELAUNCH if any launch since the last call timed
 * out (its results are invalid) and clears the word. */
size_t mg_gru_persist_workspace_bytes(int B, int H);
int mg_gru_persist_supported(int B, int T, int H);
int mg_gru_persist_status(void* workspace, void* stream);
int mg_gru_fwd_persist_bf16(const float* xproj, const uint16_t* w_hh_bf, int ldw, const float* b_hh, const int64_t* seq_len, int B,
                            int T, int H, float* hstate, uint16_t* hstate_bf, float* out, float* saved, void* workspace,
                            size_t workspace_bytes, void* stream);
/* mg_gru_fwd_persist_bf16 with the input projections read through a row map: frame (b, t) takes row xrows[b * T + t] of the table xproj
 * [n_rows, 3H] (int32, every entry in [0, n_rows)) - the repetition of upsample_to_repetitions (morgana/utils.py:175-228) applied
 * inside the recurrence instead of to a [B, T, 3H] copy of the projected rows.  xrows NULL: xproj is [B, T, 3H] (the entry above). */
int mg_gru_fwd_persist_rows_bf16(const float* xproj, const int32_t* xrows, int64_t n_rows, const uint16_t* w_hh_bf, int ldw, const float* b_hh,
                                 const int64_t* seq_len, int B, int T, int H, float* hstate, uint16_t* hstate_bf, float* out, float* saved,
                                 void* workspace, size_t workspace_bytes, void* stream);
/* mg_gru_fwd_persist_rows_bf16 that also writes out_bf (optional, may be NULL): the bf16 copy of `out` [B, T, H] (zero past each item's
 * length) - the operand of the nn.Linear that follows the wrapper in models/RNN_SPSS.py:38, without a cast pass over [B, T, H]. */
int mg_gru_fwd_persist_out_bf16(const float* xproj, const int32_t* xrows, int64_t n_rows, const uint16_t* w_hh_bf, int ldw, const float* b_hh,
                                const int64_t* seq_len, int B, int T, int H, float* hstate, uint16_t* hstate_bf, float* out, uint16_t* out_bf,
                                float* saved, void* workspace, size_t workspace_bytes, void* stream);
/* backward: dxproj_bf (optional) = bf16 shadow of dxproj [B,T,3H]; dxproj and dhproj (both or neither) may be NULL when the caller
 * only needs the bf16 shadows (the weight- and input-gradient GEMMs of bf16 mode) - the fp32 arrays are then not written */
int mg_gru_bwd_persist_bf16(const float* grad_out, const float* grad_hn, const float* hstate, const float* saved,
                            const uint16_t* w_hh_t_bf, int ldt, const int64_t* seq_len, int B, int T, int H, float* dxproj,
                            float* dhproj, uint16_t* dhproj_bf, uint16_t* dxproj_bf, float* dh0, void* workspace,
                            size_t workspace_bytes, void* stream);

/* fp32 parity mode of the LSTM recurrence as ONE launch per direction for 256 <= H <= 512, H % 64 == 0, B <= 128 (csrc/lstm_persist_f32.hip):
 * W_hh resident in registers, fp32 hand-off tiles, the exact-fp32 MFMA in the per-step kernels' block order and the same cell code -
 * results bit-identical to mg_lstm_fwd_f32 / mg_lstm_bwd_f32 on the live steps (gate values of steps beyond an item group's longest
 * sequence are written as zeros).  Arguments as mg_lstm_fwd_f32 / mg_lstm_bwd_f32 (grad_out may be NULL = zero); workspace
 * (mg_gru_persist_workspace_bytes(B, H)), status word and residency requirement as for mg_gru_fwd_persist_bf16. */
int mg_lstm_persist_f32_supported(int B, int T, int H);
int mg_lstm_fwd_persist_f32(const float* xproj, const float* w_hh, const float* b_hh, const int64_t* seq_len, int B, int T, int H,
                            float* hstate, float* cstate, float* out, float* saved, void* workspace, size_t workspace_bytes,
                            void* stream);
int mg_lstm_bwd_persist_f32(const float* grad_out, const float* grad_hn, const float* grad_cn, const float* cstate, const float* saved,
                            const float* w_hh, const int64_t* seq_len, int B, int T, int H, float* dgates, float* dh0, float* dc0,
                            void* workspace, size_t workspace_bytes, void* stream);

/* The LSTM recurrence (gates i, f, g, o) in the same persistent form: bf16 matmul operands, fp32 cell, one launch per
 * direction.  w_hh_bf = bf16(W_hh) [4H, ldw]; hstate_bf [B,T+1,H] = bf16 shadow of hstate (slot 0 set by the caller, the rest
 * written here); backward takes w_hh_t_bf = bf16(W_hh^T) [H, ldt >= 4H] and fills dgates [B,T,4H] (optional: NULL = not written)
 * and its bf16 shadow dgates_bf.
 * Other arguments as mg_lstm_fwd_f32 / mg_lstm_bwd_f32.  Workspace, status word and residency requirement: as for the GRU entry
 * points (mg_gru_persist_workspace_bytes(B, H) covers both; mg_gru_persist_status reads the shared status word). */
int mg_lstm_persist_supported(int B, int T, int H);
int mg_lstm_fwd_persist_bf16(const float* xproj, const uint16_t* w_hh_bf, int ldw, const float* b_hh, const int64_t* seq_len, int B,
                             int T, int H, float* hstate, float* cstate, uint16_t* hstate_bf, float* out, float* saved,
                             void* workspace, size_t workspace_bytes, void* stream);
int mg_lstm_bwd_persist_bf16(const float* grad_out, const float* grad_hn, const float* grad_cn, const float* cstate,
                             const float* saved, const uint16_t* w_hh_t_bf, int ldt, const int64_t* seq_len, int B, int T, int H,
                             float* dgates, uint16_t* dgates_bf, float* dh0, float* dc0, void* workspace, size_t workspace_bytes,
                             void* stream);

/* A whole stack of L LSTM layers (2 <= L <= MG_LSTM_MAX_LAYERS, the 8 x nn.LSTM(512, 512) of models/RNN_SPSS.py:36-37) forward
 * in ONE launch: a wavefront over (layer, time) - T + L - 1 dependent steps instead of L T.  Layer 0 takes xproj = x W_ih^T +
 * b_ih [B,T,4H] from memory; the layers above compute their input projection inside the step from the hand-off tiles of the
 * layer below (w_ih_bf = bf16(W_ih) [4H, ldwi], input size == H).  Per layer: cstate, saved and hstate_bf complete; of hstate
 * only slot T (h_n); out only for the top layer (the other layers' out buffers are not written).
 * mg_lstm_pstack_supported(B, T, H, L): H % 128 == 0, H <= 512 and L G H / 16 <= 512 workgroups for a group count G in
 * {8, 4, 2, 1} with ceil(B / G) <= 32 (L = 8, H = 512: B <= 64).  All workgroups must be resident together, two per CU.
 * The workspace (mg_lstm_pstack_workspace_bytes, zeroed once by the caller) carries the sticky status word at the same offset
 * as the other persistent entry points (mg_gru_persist_status). */
typedef struct {
    const float* xproj;           /* layer 0 only */
    const uint16_t* w_ih_bf;      /* layers >= 1 */
    const float* b_ih;            /* layers >= 1 */
    const uint16_t* w_hh_bf;
    const float* b_hh;
    float* hstate;                /* [B,T+1,H], slot 0 = h0 set by the caller; only slot T (h_n) is written */
    float* cstate;                /* [B,T+1,H], slot 0 = c0 set by the caller */
    uint16_t* hstate_bf;          /* [B,T+1,H] bf16 shadow, slot 0 set by the caller */
    float* out;                   /* [B,T,H]; written for the top layer only */
    float* saved;                 /* [B,T,4H] gate values i, f, g, o */
    int ldwi, ldwh;
} mg_lstm_pstack_layer;
int mg_lstm_pstack_supported(int B, int T, int H, int L);
size_t mg_lstm_pstack_workspace_bytes(int B, int H, int L);
int mg_lstm_pstack_fwd_bf16(const mg_lstm_pstack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                            size_t workspace_bytes, void* stream);

/* The backward of the same stack in ONE launch (reference: autograd through the 8 chained nn.LSTM of models/RNN_SPSS.py:36-37): the
 * forward's wavefront run down in time and down through the layers.  Layer l's step t takes the gate gradients of the layer above
 * at step t through that layer's W_ih inside the step (d out^l_t = dgates^{l+1}_t W_ih^{l+1}) instead of a grad_out row from memory,
 * so the L - 1 input-gradient GEMMs between per-layer launches go as well.  Per layer: dgates_bf [B,T,4H] (bf16 gate gradients
 * i, f, g, o: the operand of the weight-gradient GEMMs and, for layer 0, of the input-gradient GEMM; exactly 0 on padded steps),
 * optionally the fp32 copy dgates, dh0 / dc0 [B,H].  Same arithmetic as L chained mg_lstm_bwd_persist_bf16 calls with
 * mg_linear_dgrad_bf16 between them, up to the summation order of the in-step product.
 * mg_lstm_pstack_bwd_supported(B, T, H, L): H % 128 == 0, H <= 512, B T 4H < 2^31 and L G H / 32 <= 256 workgroups (one per CU, a
 * slot of 32 hidden units) or L G H / 16 <= 512 (two per CU) for a group count G in {8, 4, 2, 1} with ceil(B / G) <= 32.
 * Workspace: mg_lstm_pstack_bwd_workspace_bytes, zeroed once by the caller, status word as for the other persistent launches. */
typedef struct {
    const float* grad_out;          /* top layer only: [B,T,H] or NULL (zero); ignored below the top */
    const float* grad_hn;           /* NULL or [B,H] */
    const float* grad_cn;           /* NULL or [B,H] */
    const float* cstate;            /* [B,T+1,H] of the forward */
    const float* saved;             /* [B,T,4H] gate values of the forward */
    const uint16_t* w_hh_t_bf;      /* bf16(W_hh^T) [H, ldt] */
    const uint16_t* w_ih_up_t_bf;   /* layers below the top: bf16(W_ih^T) of the layer ABOVE, [H, ldt_up] */
    float* dgates;                  /* NULL or [B,T,4H] */
    uint16_t* dgates_bf;            /* [B,T,4H] */
    float* dh0;                     /* [B,H] */
    float* dc0;                     /* [B,H] */
    int ldt, ldt_up;
} mg_lstm_pstack_bwd_layer;
int mg_lstm_pstack_bwd_supported(int B, int T, int H, int L);
size_t mg_lstm_pstack_bwd_workspace_bytes(int B, int H, int L);
int mg_lstm_pstack_bwd_bf16(const mg_lstm_pstack_bwd_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                            size_t workspace_bytes, void* stream);

/* A STACK of small GRU layers (reference: the three RecurrentCuDNNWrapper(nn.GRU(., 64)) of the shipped F0 model,
 * models/f0_test_model.py:31-37, one after the other through SequentialWithRecurrent, utils.py:396-418) as ONE launch per direction:
 * the workgroup-local kernels above run as a wavefront over (layer, time), one workgroup per (layer, block of 4 items).  Layers above
 * the first compute their input projection inside the step (w_ih [3H, H], b_ih: their input IS the lower layer's output, input size
 * == H) and, backward, the gradient they hand to the layer below (dxproj_t W_ih), so the projection and input-gradient GEMMs between
 * the layers go.  Exact fp32; per layer the same results as mg_gru_fwd_small_f32 / mg_gru_bwd_small_f32 fed with the neighbouring
 * layer's rows, up to the summation order of those in-step products.
 * Hand-off buffers: `out` of every layer below the top (forward) and `dxin` of every layer below the top (backward) are filled with a
 * sentinel by the entry point and written exactly once per element by the producing layer; after the launch they hold ordinary
 * values (`out`: the layer's outputs, zero on padded steps, as from mg_gru_fwd_f32).
 * mg_gru_stack_small_supported: H == 64, 2 <= L <= MG_GRU_STACK_MAX_LAYERS, all L ceil(B / 4) workgroups resident.
 * Workspace (mg_gru_stack_small_workspace_bytes, zeroed once by the caller): the sticky status word of mg_gru_persist_status. */
#define MG_GRU_STACK_MAX_LAYERS 4
typedef struct {
    /* forward */
    const float* xproj;     /* layer 0: [B,T,3H] = x W_ih^T + b_ih */
    const float* w_ih;      /* layers >= 1: [3H,H] (both directions) */
    const float* b_ih;      /* layers >= 1: [3H] */
    const float* w_hh;      /* [3H,H] (both directions) */
    const float* b_hh;      /* [3H] */
    float* hstate;          /* [B,T+1,H], slot 0 = h0 on entry (read by the backward) */
    float* out;             /* [B,T,H] */
    float* saved;           /* [B,T,4H] (read by the backward) */
    /* backward */
    const float* grad_out;  /* top layer: [B,T,H] */
    const float* grad_hn;   /* NULL or [B,H] */
    float* dxin;            /* layers below the top: [B,T,H], the gradient of the layer's outputs as the layer above hands it down */
    float* dxproj;          /* [B,T,3H] */
    float* dhproj;          /* [B,T,3H] */
    float* dh0;             /* [B,H] */
} mg_gru_stack_layer;
int mg_gru_stack_small_supported(int B, int T, int H, int L);
size_t mg_gru_stack_small_workspace_bytes(void);
int mg_gru_stack_fwd_small_f32(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                               size_t workspace_bytes, void* stream);
/* The throughput-mode ("bf16" precision) forms, same arguments: the cell's sigmoid / tanh on v_exp_f32 / v_rcp_f32 and the step's
 * products on v_mfma_f32_16x16x32_bf16 (W_hh / W_ih, the state and what the layers hand each other rounded to bf16 as operands, fp32
 * accumulation - as the bf16-mode GRU-512 recurrence); state, cell and everything stored stay fp32.  The _f32 entries are exact fp32. */
int mg_gru_stack_fwd_small_fast_f32(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                                    size_t workspace_bytes, void* stream);
int mg_gru_stack_bwd_small_f32(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                               size_t workspace_bytes, void* stream);
int mg_gru_stack_bwd_small_fast_f32(const mg_gru_stack_layer* layers, int L, const int64_t* seq_len, int B, int T, int H, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* LSTM through RecurrentCuDNNWrapper   reference: morgana/utils.py:345-393 + torch.nn.LSTM (gates i, f, g, o), the cell of
 * the reference's shipped acoustic model (models/RNN_SPSS.py:36-37).  Same conventions as the GRU entry points:
 *   xproj [B,T,4H] = x W_ih^T + b_ih; w_hh [4H,H]; b_hh [4H]; hstate / cstate [B,T+1,H] with slot 0 = (h0, c0) on entry;
 *   out [B,T,H] zero on padded steps; saved [B,T,4H] = activated gates (i, f, g, o).
 * Backward: dgates [B,T,4H] = dL/d(gate pre-activations) (feeds dW_ih, dW_hh, both biases and dx); dh0, dc0 [B,H];
 *   grad_hn / grad_cn NULL or [B,H]; workspace mg_lstm_bwd_workspace_bytes(B,H). */
int mg_lstm_fwd_f32(const float* xproj, const float* w_hh, const float* b_hh, const int64_t* seq_len, int B, int T, int H,
                    float* hstate, float* cstate, float* out, float* saved, void* stream);
size_t mg_lstm_bwd_workspace_bytes(int B, int H);
int mg_lstm_bwd_f32(const float* grad_out, const float* grad_hn, const float* grad_cn, const float* cstate, const float* saved,
                    const float* w_hh, const int64_t* seq_len, int B, int T, int H, float* dgates, float* dh0, float* dc0,
                    void* workspace, size_t workspace_bytes, void* stream);

/* A stack of LSTM layers run skewed in time (reference: the 8 x RecurrentCuDNNWrapper(nn.LSTM) of models/RNN_SPSS.py:36-37,
 * or one multi-layer nn.LSTM): layer l runs `lag` steps behind layer l-1 and ONE launch per step serves every layer, so the
 * stack costs T + (L-1) lag dependent launches instead of L T.  The caller interleaves, every `lag` steps, the GEMMs that turn
 * the outputs a layer has just finished into the next chunk of its upper neighbour's input projections (forward) or the
 * gate gradients of a layer into the next chunk of its lower neighbour's output gradients (backward).
 *   forward  step s: layer l processes time t = s - l*lag (0 <= t < T), reading xproj rows of times x_t0 .. x_t0+x_T-1
 *   backward step u: layer l processes time t = T_pad-1 - (u - (L-1-l)*lag), T_pad = ceil(T/lag)*lag; t = -1 finishes
 *                    dh0 / dc0; grad_out (NULL = zero) holds times g_t0 .. g_t0+g_T-1; carry_h / carry_c [B,H] must hold
 *                    grad_hn / grad_cn (or zeros) before the first step.
 * Buffers per layer as in mg_lstm_fwd_f32 / mg_lstm_bwd_f32.  `layers` is a HOST array, n_layers <= MG_LSTM_MAX_LAYERS. */
#define MG_LSTM_MAX_LAYERS 8
typedef struct {
    const float* xproj;
    int x_T, x_t0;
    const float* w_hh;
    const float* b_hh;
    float* hstate;
    float* cstate;
    float* out;
    float* saved;
} mg_lstm_fwd_layer;
typedef struct {
    const float* grad_out;
    int g_T, g_t0;
    const float* cstate;
    const float* saved;
    const float* w_hh;
    float* dgates;
    float* carry_h;
    float* carry_c;
    float* dh0;
    float* dc0;
} mg_lstm_bwd_layer;
int mg_lstm_stack_fwd_f32(const mg_lstm_fwd_layer* layers, int n_layers, const int64_t* seq_len, int B, int T, int H, int lag,
                          int s_begin, int s_end, void* stream);
int mg_lstm_stack_bwd_f32(const mg_lstm_bwd_layer* layers, int n_layers, const int64_t* seq_len, int B, int T, int H, int lag,
                          int u_begin, int u_end, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Optimiser / EMA                         reference: torch.optim.Adam at experiment_builder.py:516, :468-474;
 *                                         ExponentialMovingAverage.update_params, morgana/utils.py:443-456
 * ---------------------------------------------------------------------------------------------------------------- */
/* One Adam step over a flat fp32 parameter buffer (torch defaults: L2 weight decay added to the gradient).
 * grad is read as grad * grad_scale (1/world_size after a sum all-reduce).  step is 1-based. */
int mg_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                     void* stream);
/* The same step with its two step-dependent scalars read from DEVICE memory (scalars[0] = lr / (1 - beta1^step),
 * scalars[1] = sqrt(1 - beta2^step); mg_adam_scalars forms them on the host exactly as mg_adam_step_f32 does): the launch can be
 * captured in a hipGraph and replayed for every step with only those 8 bytes rewritten. */
void mg_adam_scalars(float lr, float beta1, float beta2, int64_t step, float* out2);
/* dst[0..1] = (a, b) in stream order, the values carried as kernel arguments (safe however far the host runs ahead). */
int mg_store_pair_f32(float* dst, float a, float b, void* stream);
/* dst[0 .. 2 n_pairs) = the pairs of the HOST array `values` (n_pairs <= MG_STORE_PAIRS_MAX), the same way: one launch stages the
 * scalars of every step a multi-step graph replay performs (slot j = dst + 2 j, handed to the j-th update launch). */
#define MG_STORE_PAIRS_MAX 32
int mg_store_pairs_f32(float* dst, const float* values, int n_pairs, void* stream);
int mg_adam_step_dev_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1,
                         float beta2, float eps, float weight_decay, const float* scalars, float grad_scale, void* stream);
/* The update as the LAST node of a training step that hands it more than a finished gradient (scalars from device memory as in
 * mg_adam_step_dev_f32).  `plan` is a HOST struct, copied into the launch:
 *   slabs[i]   : elements [begin, begin+count) of the flat gradient additionally receive sum_s slab[s*stride + (j - begin)],
 *                s < n_slabs, summed in the fixed order of the library's slab reduce (16 interleaved partitions, ascending) - the
 *                split-M partial results of a weight-gradient GEMM (mg_linear_wgrad_slabs_bf16), consumed here instead of by a
 *                reduce launch of their own; n_slabs == 1 adds a plain buffer (the fused tail's gradients)
 *   shadows[i] : the fp32 matrix [rows, cols] at flat offset `offset` is re-cast after its update into dst bf16 [rows, ldd] and / or
 *                its transpose dst_t bf16 [cols, ldt] (padding columns are never written: zero them once) - the operands of the next
 *                step's GEMMs, so no cast launch is needed
 *   clear_grad : != 0 zeroes the flat gradient behind the read (the next step needs no memset)
 * Results equal mg_adam_step_dev_f32 on the reduced gradient bit for bit. */
#define MG_ADAM_MAX_SLABS 4
#define MG_ADAM_MAX_SHADOWS 8
typedef struct {
    int64_t begin, count;
    const float* slab;
    int n_slabs;
    int64_t stride;
} mg_adam_slab_src;
typedef struct {
    int64_t offset;
    int rows, cols;
    uint16_t* dst;
    int ldd;
    uint16_t* dst_t;
    int ldt;
    int pair;       /* != 0: dst / dst_t are [hi | lo] PAIR PLANES (precision 'bf16x3', the mg_*_x3 entry points): ldd / ldt = the row stride
                     * of both planes, hi = bf16(w) at column c, lo = bf16(w - hi) at column ldd / 2 + c (ldt / 2 + r for the transpose) */
} mg_adam_shadow;
/* What a step captured whole into a HIP graph leaves of its FORWARD for the update's launch (nothing can observe the step half done
 * inside a replay): the per-phone prediction repeated to frames (out[f] = table[rows[f]], `frames` of them; 0 = none) and the loss =
 * ordered sum over n_slabs slabs of their element n - 1 (+ the constant term: n_partial partial sums, or none), stored at dst[n - 1]
 * (the 16-element chunk that holds it is formed: dst[16 ((n - 1) / 16) .. n)); n == 0: none.  Done by the launch's first blocks
 * before their share of the update; the arithmetic of mg_expand_column_reduce_f32. */
typedef struct {
    const float* table;
    const int32_t* rows;
    int64_t frames;
    float* out;
    const float* partial;
    int n_partial;
    const float* slab;
    int64_t n, stride;
    int n_slabs;
    float* dst;
} mg_adam_tail;
typedef struct {
    int n_slab_srcs;
    mg_adam_slab_src slabs[MG_ADAM_MAX_SLABS];
    int n_shadows;
    mg_adam_shadow shadows[MG_ADAM_MAX_SHADOWS];
    int clear_grad;
    mg_adam_tail tail;
} mg_adam_plan;
int mg_adam_step_plan_f32(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2, float eps,
                          float weight_decay, const float* scalars, float grad_scale, const mg_adam_plan* plan, void* stream);
/* shadow -= (1 - decay) * (shadow - param). */
int mg_ema_update_f32(float* shadow, const float* param, int64_t n, float decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MORGANA_HIP_H */
