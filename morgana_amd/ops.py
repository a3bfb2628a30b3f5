"""Tensor-level wrappers over the C ABI: validate, allocate outputs (torch is the allocator), launch on torch's current
HIP stream.  No arithmetic happens here and nothing falls back to torch ops: a missing library or a CPU tensor raises.
"""
import ctypes

import os

import torch

from . import _lib

ACT_NONE, ACT_SIGMOID = 0, 1
NORM_MVN, DENORM_MVN, NORM_MINMAX, DENORM_MINMAX = 0, 1, 2, 3

_workspaces = {}
SIDE_STREAM_IDS = set()      # raw handles of the side streams functional._Beside launches on (their scratch is their own)
_retired = []


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    # the current stream's raw handle (torch.cuda.current_stream().cuda_stream builds a Stream object first: 9 us a call, five calls
    # per eager F0Model step)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _require(t, dtype, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor, got %s' % (name, type(t)))
    if not t.is_cuda:
        raise _lib.MorganaHipError('%s is on %s: the morgana_amd ops run only on an MI355X device (no CPU fallback)'
                                   % (name, t.device))
    if t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def workspace(nbytes, device):
    """One grow-only scratch buffer per device; all kernels of a step run on one stream, so it is shared.  A buffer that is
    outgrown is kept alive (``_retired``): a captured HIP graph (morgana_amd/graphs.py) may still launch kernels that point at it."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    # ... and one more for launches that run BESIDE the step on a side stream (functional._Beside): they must not share scratch with it
    key = (index, bool(SIDE_STREAM_IDS) and device.type == 'cuda' and torch._C._cuda_getCurrentRawStream(index) in SIDE_STREAM_IDS)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


_tail_slab_bufs = {}


def _tail_slabs(nbytes, device):
    """Grow-only buffer per device for the fused tail's slabs when their sum is deferred (f0_l2tail_rows_expand(defer=True)); an
    outgrown one is retired, not dropped (captured graphs keep its address)."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    buf = _tail_slab_bufs.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _tail_slab_bufs[key] = buf
    return buf


def _slab_buffer(slab, nbytes, device):
    """A per-layer slab buffer of at least ``nbytes``: ``slab`` itself when it is large enough, else a new one - and the outgrown
    buffer is RETIRED, not dropped (as `workspace` does): a HIP graph captured at the smaller shape still launches the weight-gradient
    and update kernels with its address baked in, and torch's caching allocator would hand that memory to the next tensor."""
    if slab is not None and slab.numel() >= nbytes:
        return slab
    if slab is not None:
        _retired.append(slab)
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def pad_ld(n):
    """Leading dimension of a bf16 buffer with n columns: a multiple of 64 (the large-tile kernels' K step) once the
    row is at least 64 wide, a multiple of 8 (one 16-byte chunk) below that.  Padding columns hold zeros."""
    return (n + 63) // 64 * 64 if n >= 64 else (n + 7) // 8 * 8


pad8 = pad_ld


# ----------------------------------------------------------------------------------------------------------------- K1
def upsample_lengths(dur):
    """dur int64 (B, P) -> (n_frames int64 (B,), tmax int64 0-d), both on device (no sync)."""
    lib = _lib.load()
    dur = _require(dur, torch.int64, 'dur')
    b, p = dur.shape
    n_frames = torch.empty((b,), dtype=torch.int64, device=dur.device)
    tmax = torch.empty((), dtype=torch.int64, device=dur.device)
    _lib.check(lib.mg_upsample_lengths(_p(dur), b, p, _p(n_frames), _p(tmax), _stream()), 'mg_upsample_lengths')
    return n_frames, tmax


def upsample_index(dur, t_cap, want_idx64=False, want_rows=True):
    lib = _lib.load()
    dur = _require(dur, torch.int64, 'dur')
    b, p = dur.shape
    idx64 = torch.empty((b, t_cap), dtype=torch.int64, device=dur.device) if want_idx64 else None
    rows = torch.empty((b, t_cap), dtype=torch.int32, device=dur.device) if want_rows else None
    _lib.check(lib.mg_upsample_index(_p(dur), b, p, int(t_cap), _p(idx64), _p(rows), _stream()), 'mg_upsample_index')
    return idx64, rows


def upsample_index_maps(dur, t_cap):
    """(rows (B, T) int32 with -1 padding, rows_mapped (-1 -> B*P), seg (2, B*P) frame runs) in one launch (mg_upsample_index_maps)."""
    lib = _lib.load()
    dur = _require(dur, torch.int64, 'dur')
    b, p = dur.shape
    rows = torch.empty((2, b, t_cap), dtype=torch.int32, device=dur.device)
    seg = torch.empty((2, b * p), dtype=torch.int32, device=dur.device)
    _lib.check(lib.mg_upsample_index_maps(_p(dur), b, p, int(t_cap), _p(rows[0]), _p(rows[1]), b * p, _p(seg[0]), _p(seg[1]),
                                          _stream()), 'mg_upsample_index_maps')
    return rows[0], rows[1], seg


def gather_rows(src2d, rows, out_bf16=False, ld=None):
    lib = _lib.load()
    src2d = _require(src2d, torch.float32, 'src')
    rows = _require(rows, torch.int32, 'rows')
    m, f = rows.numel(), src2d.shape[1]
    if out_bf16:
        ldo = pad8(f) if ld is None else int(ld)
        out = torch.empty((m, ldo), dtype=torch.bfloat16, device=src2d.device)
        _lib.check(lib.mg_gather_rows_bf16(_p(src2d), _p(rows), _p(out), m, f, ldo, _stream()), 'mg_gather_rows_bf16')
    else:
        out = torch.empty((m, f), dtype=torch.float32, device=src2d.device)
        _lib.check(lib.mg_gather_rows_f32(_p(src2d), _p(rows), _p(out), m, f, _stream()), 'mg_gather_rows_f32')
    return out


def segment_index(seg_lens2d, t, max_len=None, want_split=True, want_ends=True):
    """Flat row maps of split_to_segments / get_segment_ends (mg_segment_index): (split int32 (B,S,L) or None, ends (B,S))."""
    lib = _lib.load()
    seg_lens2d = _require(seg_lens2d, torch.int64, 'segment_lens')
    b, s = seg_lens2d.shape
    length = int(max_len) if want_split else 0
    split = torch.empty((b, s, length), dtype=torch.int32, device=seg_lens2d.device) if want_split else None
    ends = torch.empty((b, s), dtype=torch.int32, device=seg_lens2d.device) if want_ends else None
    _lib.check(lib.mg_segment_index(_p(seg_lens2d), b, s, int(t), length, _p(split), _p(ends), _stream()), 'mg_segment_index')
    return split, ends


def scatter_rows(src2d, rows, n_dst_rows):
    """Adjoint of gather_rows for distinct targets: zeros (n_dst_rows, F) with dst[rows[m]] = src2d[m]."""
    lib = _lib.load()
    src2d = _require(src2d, torch.float32, 'src')
    rows = _require(rows, torch.int32, 'rows')
    dst = torch.zeros((n_dst_rows, src2d.shape[1]), dtype=torch.float32, device=src2d.device)
    _lib.check(lib.mg_scatter_rows_f32(_p(src2d), _p(rows), _p(dst), rows.numel(), src2d.shape[1], _stream()),
               'mg_scatter_rows_f32')
    return dst


def frame_layout(seq_len, t, total):
    """Packed-frame maps of a ragged (B, t, .) batch (mg_frame_layout): (offsets (B+1,), rows (total+1,), inverse (B*t,)) int32.
    ``total`` = sum of min(seq_len, t), known on the host."""
    lib = _lib.load()
    seq_len = _require(seq_len, torch.int64, 'seq_len')
    b = seq_len.numel()
    offsets = torch.empty(b + 1, dtype=torch.int32, device=seq_len.device)
    rows = torch.empty(int(total) + 1, dtype=torch.int32, device=seq_len.device)
    inverse = torch.empty(b * int(t), dtype=torch.int32, device=seq_len.device)
    _lib.check(lib.mg_frame_layout(_p(seq_len), b, int(t), int(total), _p(offsets), _p(rows), _p(inverse), _stream()), 'mg_frame_layout')
    return offsets, rows, inverse


def pad_rows_colsum(g, seq_len, out):
    """out (D,) = sum of g[b, t, :] over the padded frames t >= seq_len[b] (mg_pad_rows_colsum_f32); g (B, T, D) f32."""
    lib = _lib.load()
    g = _require(g, torch.float32, 'gradient')
    b, t, d = g.shape
    ws = torch.empty(lib.mg_pad_rows_colsum_workspace_bytes(b, t, d), dtype=torch.uint8, device=g.device)
    _lib.check(lib.mg_pad_rows_colsum_f32(_p(g), _p(seq_len), b, t, d, _p(out), _p(ws), ws.numel(), _stream()), 'mg_pad_rows_colsum_f32')
    return out


def gather_concat(src2d, rows, extra2d, out_bf16=False):
    """[src2d[rows] | extra2d] per frame (mg_gather_concat_*): f32 (M, F+C), or bf16 (M, pad_ld(F+C)) zero padded."""
    lib = _lib.load()
    src2d = _require(src2d, torch.float32, 'src')
    extra2d = _require(extra2d, torch.float32, 'extra')
    rows = _require(rows, torch.int32, 'rows')
    m, f, c = rows.numel(), src2d.shape[1], extra2d.shape[1]
    if extra2d.shape[0] != m:
        raise ValueError('gather_concat: %d frame rows but %d rows of frame-level features' % (m, extra2d.shape[0]))
    if out_bf16:
        ldo = pad_ld(f + c)
        out = torch.empty((m, ldo), dtype=torch.bfloat16, device=src2d.device)
        _lib.check(lib.mg_gather_concat_bf16(_p(src2d), _p(rows), _p(extra2d), _p(out), m, f, c, ldo, _stream()),
                   'mg_gather_concat_bf16')
    else:
        out = torch.empty((m, f + c), dtype=torch.float32, device=src2d.device)
        _lib.check(lib.mg_gather_concat_f32(_p(src2d), _p(rows), _p(extra2d), _p(out), m, f, c, f + c, _stream()),
                   'mg_gather_concat_f32')
    return out


def upsample_backward(grad_out, dur, n_phones):
    lib = _lib.load()
    grad_out = _require(grad_out, torch.float32, 'grad_out')
    dur = _require(dur, torch.int64, 'dur')
    b, t, f = grad_out.shape
    grad_src = torch.empty((b, n_phones, f), dtype=torch.float32, device=grad_out.device)
    _lib.check(lib.mg_upsample_backward_f32(_p(grad_out), _p(dur), _p(grad_src), b, n_phones, t, f, _stream()),
               'mg_upsample_backward_f32')
    return grad_src


# --------------------------------------------------------------------------------------------------- mask / K4 / K5
_MASK_TYPES = {torch.uint8: (1, 0), torch.bool: (1, 0), torch.int8: (1, 0), torch.float32: (4, 1), torch.int32: (4, 0),
               torch.int64: (8, 0), torch.float64: (8, 1)}


def sequence_mask(seq_len, max_len, dtype):
    lib = _lib.load()
    seq_len = _require(seq_len, torch.int64, 'seq_len')
    if dtype not in _MASK_TYPES:
        raise TypeError('sequence_mask: unsupported mask dtype %s' % dtype)
    elem, as_float = _MASK_TYPES[dtype]
    b = seq_len.shape[0]
    mask = torch.empty((b, max_len, 1), dtype=dtype, device=seq_len.device)
    _lib.check(lib.mg_sequence_mask(_p(seq_len), b, int(max_len), _p(mask), elem, as_float, _stream()),
               'mg_sequence_mask')
    return mask


def masked_mse(pred, target, seq_len, want_grad, grad_scale=1.0, kind='mse'):
    """Returns (loss 0-d f32 tensor, grad or None).  kind: 'mse' or 'bce'."""
    lib = _lib.load()
    pred = _require(pred, torch.float32, 'predictions')
    target = _require(target, torch.float32, 'targets')
    if pred.shape != target.shape or pred.dim() != 3:
        raise ValueError('mse: predictions %s and targets %s must both be (B, T, D)' % (tuple(pred.shape),
                                                                                     tuple(target.shape)))
    if seq_len is not None:
        seq_len = _require(seq_len, torch.int64, 'seq_len')
    b, t, d = pred.shape
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    nbytes = lib.mg_masked_mse_workspace_bytes(b, t, d)
    ws = workspace(nbytes, pred.device)
    fn = lib.mg_masked_mse_f32 if kind == 'mse' else lib.mg_masked_bce_f32
    _lib.check(fn(_p(pred), _p(target), _p(seq_len), b, t, d, float(grad_scale), _p(loss), _p(grad), _p(ws), ws.numel(),
                  _stream()), 'mg_masked_%s_f32' % kind)
    return loss, grad


def stream_loss(pred, targets, kinds, seq_len, want_grad, want_prob=False, grad_scale=1.0):
    """Multi-stream loss (mg_stream_loss_f32): pred (B, T, sum of widths), targets[k] (B, T, width_k) scored side by side
    in column order, kinds[k] in {'mse', 'sigmoid_bce'}.  Returns (loss 0-d, grad or None, prob or None)."""
    lib = _lib.load()
    pred = _require(pred, torch.float32, 'predictions')
    if pred.dim() != 3 or len(targets) != len(kinds) or not 1 <= len(targets) <= _lib.STREAMS_MAX:
        raise ValueError('stream_loss: predictions must be (B, T, D) with 1..%d streams' % _lib.STREAMS_MAX)
    b, t, d = pred.shape
    if sum(int(y.shape[-1]) for y in targets) != d:
        raise ValueError('stream_loss: stream widths %s do not add up to D=%d' % ([int(y.shape[-1]) for y in targets], d))
    if seq_len is not None:
        seq_len = _require(seq_len, torch.int64, 'seq_len')
    descs = (_lib.StreamDesc * len(targets))()
    keep, col0, prob = [], 0, None
    for k, (y, kind) in enumerate(zip(targets, kinds)):
        y = _require(y, torch.float32, 'targets[%d]' % k)
        if y.dim() != 3 or y.shape[0] != b or y.shape[1] != t:
            raise ValueError('stream_loss: targets[%d] %s does not match predictions (%d, %d, *)' % (k, tuple(y.shape), b, t))
        keep.append(y)
        w = int(y.shape[2])
        descs[k].target, descs[k].ldt, descs[k].col0, descs[k].width = y.data_ptr(), w, col0, w
        descs[k].kind = {'mse': _lib.LOSS_MSE, 'sigmoid_bce': _lib.LOSS_SIGMOID_BCE}[kind]
        if kind == 'sigmoid_bce' and want_prob:
            prob = torch.empty((b, t, w), dtype=torch.float32, device=pred.device)
        col0 += w
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    ws = workspace(lib.mg_stream_loss_workspace_bytes(b, t, d), pred.device)
    _lib.check(lib.mg_stream_loss_f32(_p(pred), ctypes.cast(descs, ctypes.c_void_p), len(targets), _p(seq_len), b, t, d,
                                      float(grad_scale), _p(loss), _p(grad), _p(prob), _p(ws), ws.numel(), _stream()),
               'mg_stream_loss_f32')
    return loss, grad, prob


def pad_normalise(packed, offsets, t, p0=None, p1=None, kind=None, want_raw=True, bf16_extra_rows=None):
    """Packed utterances (sum len, D) + offsets (B+1) -> (raw (B,t,D) or None, normalised (B,t,D) or None).
    ``bf16_extra_rows`` (an int, needs ``kind``): the same pass also writes the normalised feature's bf16 operand table
    (B*t + bf16_extra_rows, pad_ld(D)) - what cast_pad_bf16 would make of the normalised output - and it is returned as a third value."""
    lib = _lib.load()
    packed = _require(packed, torch.float32, 'packed feature')
    offsets = _require(offsets, torch.int64, 'offsets')
    b, d = offsets.numel() - 1, packed.shape[1]
    raw = torch.empty((b, t, d), dtype=torch.float32, device=packed.device) if want_raw else None
    norm = None
    if kind is not None:
        p0 = _require(p0, torch.float32, 'param0')
        p1 = _require(p1, torch.float32, 'param1')
        if p0.numel() != d or p1.numel() != d:
            raise ValueError('normaliser parameters have %d / %d entries, feature dim is %d' % (p0.numel(), p1.numel(), d))
        norm = torch.empty((b, t, d), dtype=torch.float32, device=packed.device)
    if bf16_extra_rows is not None:
        if kind is None:
            raise ValueError('pad_normalise: the bf16 table is the NORMALISED feature: it needs normaliser parameters')
        ldb = pad_ld(d)
        table = torch.empty((b * int(t) + int(bf16_extra_rows), ldb), dtype=torch.bfloat16, device=packed.device)
        _lib.check(lib.mg_pad_normalise_bf16_f32(_p(packed), _p(offsets), b, int(t), d, _p(p0), _p(p1), kind, _p(raw), _p(norm), _p(table), ldb,
                                                 int(bf16_extra_rows), _stream()), 'mg_pad_normalise_bf16_f32')
        return raw, norm, table
    _lib.check(lib.mg_pad_normalise_f32(_p(packed), _p(offsets), b, int(t), d, _p(p0), _p(p1), -1 if kind is None else kind,
                                        _p(raw), _p(norm), _stream()), 'mg_pad_normalise_f32')
    return raw, norm


def normalise(x, p0, p1, kind):
    lib = _lib.load()
    x = _require(x, torch.float32, 'feature')
    p0 = _require(p0, torch.float32, 'param0')
    p1 = _require(p1, torch.float32, 'param1')
    d = x.shape[-1]
    if p0.numel() != d or p1.numel() != d:
        raise ValueError('normaliser parameters have %d / %d entries, feature dim is %d' % (p0.numel(), p1.numel(), d))
    out = torch.empty_like(x)
    _lib.check(lib.mg_normalise_f32(_p(x), _p(out), _p(p0), _p(p1), x.numel() // d, d, kind, _stream()),
               'mg_normalise_f32')
    return out


# ----------------------------------------------------------------------------------------------------------------- K2
def linear_fwd_f32(a, rows, m, weight, bias, act):
    """a: (R, K) f32 table/input; rows: int32 (m,) or None.  Returns (m, N) f32."""
    lib = _lib.load()
    n, k = weight.shape
    y = torch.empty((m, n), dtype=torch.float32, device=weight.device)
    _lib.check(lib.mg_linear_fwd_f32(_p(a), a.shape[1], _p(rows), m, k, _p(weight), _p(bias), n, _p(y), n, act,
                                     _stream()), 'mg_linear_fwd_f32')
    return y


def linear_dgrad_f32(dy, weight, h):
    lib = _lib.load()
    n, k = weight.shape
    m = dy.shape[0]
    dx = torch.empty((m, k), dtype=torch.float32, device=dy.device)
    _lib.check(lib.mg_linear_dgrad_f32(_p(dy), m, n, _p(weight), k, _p(h), _p(dx), _stream()), 'mg_linear_dgrad_f32')
    return dx


def linear_wgrad_f32(dy, a, rows, n, k, want_bias=True):
    lib = _lib.load()
    m = dy.shape[0]
    dw = torch.empty((n, k), dtype=torch.float32, device=dy.device)
    db = torch.empty((n,), dtype=torch.float32, device=dy.device) if want_bias else None
    nbytes = lib.mg_linear_wgrad_workspace_bytes(m, n, k)
    ws = workspace(nbytes, dy.device)
    _lib.check(lib.mg_linear_wgrad_f32(_p(dy), _p(a), a.shape[1], _p(rows), m, n, k, _p(dw), _p(db), 0, _p(ws),
                                       ws.numel(), _stream()), 'mg_linear_wgrad_f32')
    return dw, db


def cast_pad_bf16(x2d, ld=None, extra_rows=0):
    """(rows, cols) f32 -> (rows + extra_rows, ld) bf16, padding columns and the extra rows zero."""
    lib = _lib.load()
    rows, cols = x2d.shape
    ld = pad8(cols) if ld is None else ld
    out = torch.empty((rows + extra_rows, ld), dtype=torch.bfloat16, device=x2d.device)
    if extra_rows:
        out[rows:].zero_()
    _lib.check(lib.mg_cast_pad_bf16(_p(x2d), x2d.shape[1], _p(out), ld, rows, cols, _stream()), 'mg_cast_pad_bf16')
    return out


def cast_transpose_bf16(x2d):
    """(rows, cols) f32 -> (cols, pad8(rows)) bf16."""
    lib = _lib.load()
    rows, cols = x2d.shape
    ld = pad8(rows)
    out = torch.empty((cols, ld), dtype=torch.bfloat16, device=x2d.device)
    _lib.check(lib.mg_cast_transpose_bf16(_p(x2d), cols, _p(out), ld, rows, cols, _stream()), 'mg_cast_transpose_bf16')
    return out


def cast_bf16_f32(x_bf16, cols):
    lib = _lib.load()
    rows, ld = x_bf16.shape
    out = torch.empty((rows, cols), dtype=torch.float32, device=x_bf16.device)
    _lib.check(lib.mg_cast_bf16_f32(_p(x_bf16), ld, _p(out), cols, rows, cols, _stream()), 'mg_cast_bf16_f32')
    return out


ACT_ROWS_RUNS = 0x100      # MG_ACT_ROWS_RUNS: `rows` is a frame map of upsample_to_repetitions (runs of equal indices); a hint only


def linear_fwd_bf16(a, rows, m, k, w_bf16, bias, n, act, out_f32=False, rows_runs=False):
    """a: (R, lda) bf16; w_bf16: (n, ldw) bf16.  Returns (m, pad8(n)) bf16 (or (m, n rounded up to 8) f32), padding columns zero.
    ``rows_runs``: the row map consists of runs of equal consecutive indices (performance hint, same results)."""
    lib = _lib.load()
    # a bf16 result is the next layer's operand (rows padded to the large tiles' k step); an fp32 one leaves the stack: padded to the
    # 16-byte chunk only, so that its consumer needs no slice-and-copy pass (N = 80: 64,000 x 128 -> x 80 was a 41 MB copy at C4)
    ldy = (n + 7) // 8 * 8 if out_f32 else pad8(n)
    y = torch.empty((m, ldy), dtype=torch.float32 if out_f32 else torch.bfloat16, device=a.device)
    if rows_runs and rows is not None:
        act = act | ACT_ROWS_RUNS
    _lib.check(lib.mg_linear_fwd_bf16(_p(a), a.shape[1], _p(rows), m, k, _p(w_bf16), w_bf16.shape[1], _p(bias), n,
                                      _p(y), ldy, 1 if out_f32 else 0, act, _stream()), 'mg_linear_fwd_bf16')
    return y


def linear_dgrad_bf16(dy, m, n, wt_bf16, k, h, out_f32=False):
    """dy (m, lddy) bf16; wt_bf16 = W^T (k, pad8(n)) bf16; h None or (m, ldh) bf16.  Returns (m, pad8(k)) bf16/f32."""
    lib = _lib.load()
    lddx = pad8(k)
    dx = torch.empty((m, lddx), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dy.device)
    _lib.check(lib.mg_linear_dgrad_bf16(_p(dy), dy.shape[1], m, n, _p(wt_bf16), wt_bf16.shape[1], k, _p(h),
                                        h.shape[1] if h is not None else 0, _p(dx), lddx, 1 if out_f32 else 0,
                                        _stream()), 'mg_linear_dgrad_bf16')
    return dx


def linear_wgrad_bf16(dy, a, rows, m, n, k, want_bias=True, out_w=None, out_b=None, accumulate=False):
    """out_w / out_b: optional preallocated fp32 destinations (e.g. slices of one gradient buffer); accumulate adds into them."""
    lib = _lib.load()
    if out_w is None and out_b is None and want_bias:
        # db right behind dW in one buffer: the wide-tile path then finishes both with ONE reduce launch (a slab is [dW | db])
        both = torch.empty((n * k + n,), dtype=torch.float32, device=dy.device)
        dw, db = both[:n * k].view(n, k), both[n * k:]
    else:
        dw = out_w if out_w is not None else torch.empty((n, k), dtype=torch.float32, device=dy.device)
        db = None
        if want_bias:
            db = out_b if out_b is not None else torch.empty((n,), dtype=torch.float32, device=dy.device)
    nbytes = lib.mg_linear_wgrad_workspace_bytes(m, n, k)
    ws = workspace(nbytes, dy.device)
    _lib.check(lib.mg_linear_wgrad_bf16(_p(dy), dy.shape[1], _p(a), a.shape[1], _p(rows), m, n, k, _p(dw), _p(db), int(bool(accumulate)),
                                        _p(ws), ws.numel(), _stream()), 'mg_linear_wgrad_bf16')
    return dw, db


def wgrad_rows_ok(m, n, k, lda, lddy):
    """Shapes mg_linear_wgrad_rows_bf16 takes (both operands gathered: the valid frames of a ragged batch out of padded arrays): the
    128 x 512 wide tiles, not their half-width plan (csrc/gemm_bf16_big.hip: wgrad_ksplit)."""
    return (m >= 4096 and n % 128 == 0 and lddy >= n and lddy % 8 == 0 and lda == 512 and 384 < k <= 512 and
            not (n == 128 and m <= 32768))


def linear_wgrad_rows_bf16(dy, dy_rows, a, rows, m, n, k, want_bias=True, out_w=None, out_b=None, accumulate=False):
    """dW = sum_{i < m} dy[dy_rows[i]]^T a[rows[i]] (rows None: a[dy_rows[i]]), db = sum_i dy[dy_rows[i]]: linear_wgrad_bf16 on the m
    index pairs only (``dy_rows`` int32, every entry a valid row of ``dy``).  out_w / out_b / accumulate as linear_wgrad_bf16."""
    lib = _lib.load()
    dy_rows = _require(dy_rows, torch.int32, 'dy_rows')
    if out_w is not None or out_b is not None:
        dw = out_w if out_w is not None else torch.empty((n, k), dtype=torch.float32, device=dy.device)
        db = (out_b if out_b is not None else torch.empty((n,), dtype=torch.float32, device=dy.device)) if want_bias else None
    elif want_bias:                                # db right behind dW: one reduce launch for both
        both = torch.empty((n * k + n,), dtype=torch.float32, device=dy.device)
        dw, db = both[:n * k].view(n, k), both[n * k:]
    else:
        dw, db = torch.empty((n, k), dtype=torch.float32, device=dy.device), None
    nbytes = lib.mg_linear_wgrad_workspace_bytes(m, n, k)
    ws = workspace(nbytes, dy.device)
    _lib.check(lib.mg_linear_wgrad_rows_bf16(_p(dy), dy.shape[1], _p(dy_rows), _p(a), a.shape[1], _p(rows), m, n, k, _p(dw), _p(db),
                                             int(bool(accumulate)), _p(ws), ws.numel(), _stream()), 'mg_linear_wgrad_rows_bf16')
    return dw, db


def wgrad_slabs_ok(m, n, k, lda, lddy):
    """Shapes whose weight gradient runs on the wide-tile kernel and can leave its split-M slabs to the optimiser
    (mg_linear_wgrad_slabs_bf16; the plan of csrc/gemm_bf16_big.hip)."""
    # m <= 32768: the plan then cuts 48-96 slabs; at frame-rate row counts it cuts 192-256, which a reduce launch of its own sums as
    # fast as the update kernel would (A/B after the update kernel's rework: 0.591 vs 0.592 ms per frame-rate step; before it 101 us
    # for 256 slabs of the 128 x 512 gradient on one thread per element)
    return (4096 <= m <= 32768 and n % 128 == 0 and lddy >= n and lddy % 8 == 0 and
            ((lda == 640 and 512 < k <= 640) or (lda == 512 and 384 < k <= 512)))


def wgrad_wide_ok(m, n, k, lda, lddy):
    """Shapes whose weight gradient runs on the wide-tile kernel at all (any row count from 4096 up): its slabs can be taken with
    linear_wgrad_slabs_bf16 and summed by ONE slab_reduce launch (dW | db together) instead of linear_wgrad_bf16's two."""
    return (m >= 4096 and n % 128 == 0 and lddy >= n and lddy % 8 == 0 and
            ((lda == 640 and 512 < k <= 640) or (lda == 512 and 384 < k <= 512)))


def linear_wgrad_slabs_bf16(dy, a, rows, m, n, k, slab=None):
    """linear_wgrad_bf16 without the reduce: returns (slab buffer, n_slabs, stride); slab s holds [n*k weight partials | n bias
    partials].  ``slab`` = a buffer to reuse (kept by the caller until the optimiser has consumed it)."""
    lib = _lib.load()
    nbytes = lib.mg_linear_wgrad_workspace_bytes(m, n, k)
    slab = _slab_buffer(slab, nbytes, dy.device)
    n_slabs, stride = ctypes.c_int(0), ctypes.c_int64(0)
    _lib.check(lib.mg_linear_wgrad_slabs_bf16(_p(dy), dy.shape[1], _p(a), a.shape[1], _p(rows), m, n, k, _p(slab), slab.numel(),
                                              ctypes.byref(n_slabs), ctypes.byref(stride), _stream()), 'mg_linear_wgrad_slabs_bf16')
    return slab, n_slabs.value, stride.value


def linear_wgrad_dgrad_bf16(dy, a, m, n, k, wt_bf16, slab=None, tail=None):
    """linear_wgrad_slabs_bf16(dy, a) and linear_dgrad_bf16(dy, wt_bf16, h=a) of a Linear(k -> n) whose input ``a`` (m, lda) is the
    output of the Sigmoid below it - one grid for both where the shapes allow (mg_linear_wgrad_dgrad_bf16), the two launches otherwise.
    Returns (slab buffer, n_slabs, stride, dx (m, pad8(k)) bf16).  Needs wgrad_slabs_ok(m, n, k, lda, lddy).
    ``tail``: the deferred end of f0_l2tail_rows_expand (its ``defer=True`` return) - the repeated prediction and the tail's slab sum
    then ride at the end of this grid (mg_linear_wgrad_dgrad_expand_bf16)."""
    lib = _lib.load()
    nbytes = lib.mg_linear_wgrad_workspace_bytes(m, n, k)
    slab = _slab_buffer(slab, nbytes, dy.device)
    lddx = pad8(k)
    dx = torch.empty((m, lddx), dtype=torch.bfloat16, device=dy.device)
    n_slabs, stride = ctypes.c_int(0), ctypes.c_int64(0)
    if tail is not None:
        _lib.check(lib.mg_linear_wgrad_dgrad_expand_bf16(
            _p(dy), dy.shape[1], _p(a), a.shape[1], m, n, k, _p(wt_bf16), wt_bf16.shape[1], _p(dx), lddx, _p(slab), slab.numel(),
            ctypes.byref(n_slabs), ctypes.byref(stride), _p(tail['pred_rows']), _p(tail['rows']), tail['rows'].numel(), _p(tail['out']),
            _p(tail['partials']), tail['n_table_rows'], tail['extra'], _p(tail['ws']), tail['n'], tail['stride'], tail['n_slabs'],
            _p(tail['grads_out']), 1 if tail.get('loss_only') else 0, _stream()), 'mg_linear_wgrad_dgrad_expand_bf16')
        return slab, n_slabs.value, stride.value, dx
    _lib.check(lib.mg_linear_wgrad_dgrad_bf16(_p(dy), dy.shape[1], _p(a), a.shape[1], m, n, k, _p(wt_bf16), wt_bf16.shape[1], _p(dx), lddx,
                                              _p(slab), slab.numel(), ctypes.byref(n_slabs), ctypes.byref(stride), _stream()),
               'mg_linear_wgrad_dgrad_bf16')
    return slab, n_slabs.value, stride.value, dx


def finish_deferred_tail(tail):
    """The deferred end of f0_l2tail_rows_expand as its own launch (mg_expand_column_reduce_f32) - for a backward pass that turns
    out not to run the launch the tail was meant to ride in; of f0_l2tail(defer=True): its reduce launch (mg_slab_reduce_f32)."""
    if tail.get('rows') is None:
        slab_reduce(tail['ws'], tail['n_slabs'], tail['stride'], tail['n'], tail['grads_out'])
        return
    _lib.check(_lib.load().mg_expand_column_reduce_f32(_p(tail['pred_rows']), _p(tail['rows']), tail['rows'].numel(), _p(tail['out']),
                                                        _p(tail['partials']), tail['n_table_rows'], tail['extra'], _p(tail['ws']), tail['n'],
                                                        tail['stride'], tail['n_slabs'], _p(tail['grads_out']), _stream()),
               'mg_expand_column_reduce_f32')


def slab_reduce(slab, n_slabs, stride, count, dst, accumulate=False):
    """dst[:count] (+)= ordered sum of the n_slabs slabs (f32, ``stride`` floats apart) in ``slab`` (a byte or float buffer): the reduce
    launch of linear_wgrad_bf16 on slabs taken from linear_wgrad_slabs_bf16 / linear_wgrad_dgrad_bf16 (mg_slab_reduce_f32)."""
    dst = _require(dst, torch.float32, 'dst')
    if dst.numel() < count:
        raise ValueError('slab_reduce: destination of %d floats for %d sums' % (dst.numel(), count))
    _lib.check(_lib.load().mg_slab_reduce_f32(_p(slab), int(n_slabs), int(stride), int(count), _p(dst), int(bool(accumulate)), _stream()),
               'mg_slab_reduce_f32')
    return dst


def can_fuse_bwd(m, n2, n_hidden, k0, lda0):
    """Shapes mg_linear_bwd_fused_bf16 handles (the README F0Model's first two layers at training batch sizes)."""
    return n2 == 128 and n_hidden % 128 == 0 and 512 < k0 <= 608 and lda0 == 640 and m >= 4096


def linear_bwd_fused_bf16(dz2, wt2, h1, a, rows, m, n_hidden, k0, out_w=None, out_b=None, accumulate=False):
    """dW, db of Linear(k0 -> n_hidden)+Sigmoid from dz2 = dL/d(pre-activation of the following Linear(n_hidden -> 128))."""
    lib = _lib.load()
    dw = out_w if out_w is not None else torch.empty((n_hidden, k0), dtype=torch.float32, device=dz2.device)
    db = out_b if out_b is not None else torch.empty((n_hidden,), dtype=torch.float32, device=dz2.device)
    nbytes = lib.mg_linear_bwd_fused_workspace_bytes(m, n_hidden, k0)
    ws = workspace(nbytes, dz2.device)
    _lib.check(lib.mg_linear_bwd_fused_bf16(_p(dz2), dz2.shape[1], 128, _p(wt2), wt2.shape[1], _p(h1), h1.shape[1], _p(a),
                                            a.shape[1], _p(rows), m, n_hidden, k0, _p(dw), _p(db), int(bool(accumulate)), _p(ws), ws.numel(),
                                            _stream()), 'mg_linear_bwd_fused_bf16')
    return dw, db


def linear_bwd_fused_slabs_bf16(dz2, wt2, h1, a, rows, m, n_hidden, k0, slab=None):
    """linear_bwd_fused_bf16 without the reduce (mg_linear_bwd_fused_slabs_bf16): returns (slab buffer, n_slabs, stride) for the
    optimiser's update kernel to sum; ``slab`` = a buffer to reuse (kept by the caller until the optimiser has consumed it)."""
    lib = _lib.load()
    nbytes = lib.mg_linear_bwd_fused_workspace_bytes(m, n_hidden, k0)
    slab = _slab_buffer(slab, nbytes, dz2.device)
    n_slabs, stride = ctypes.c_int(0), ctypes.c_int64(0)
    _lib.check(lib.mg_linear_bwd_fused_slabs_bf16(_p(dz2), dz2.shape[1], 128, _p(wt2), wt2.shape[1], _p(h1), h1.shape[1], _p(a), a.shape[1],
                                                  _p(rows), m, n_hidden, k0, _p(slab), slab.numel(), ctypes.byref(n_slabs),
                                                  ctypes.byref(stride), _stream()), 'mg_linear_bwd_fused_slabs_bf16')
    return slab, n_slabs.value, stride.value


def linear_bwd_fused2_slabs_bf16(dz2, wt2, h1, a, rows, m, n_hidden, k0, slab=None):
    """The fused backward with the second layer's weight gradient riding along (mg_linear_bwd_fused2_slabs_bf16): ONE launch leaves the
    split-M slabs of (dW1 | db1) and of (dW2 | db2) of a Linear(k0 -> n_hidden) + Sigmoid -> Linear(n_hidden -> 128) pair whose input
    is gathered through ``rows``.  Returns (slab buffer, n_slabs, (float offset, stride, count) of the first layer's slabs, the same
    for the second layer's): slab i of a layer starts at float offset + i * stride and holds ``count`` partial sums."""
    lib = _lib.load()
    if rows is None:
        raise ValueError('linear_bwd_fused2_slabs_bf16 needs the row map of the gathered input')
    nbytes = lib.mg_linear_bwd_fused2_workspace_bytes(m, n_hidden, k0)
    slab = _slab_buffer(slab, nbytes, dz2.device)
    n_slabs, stride1, off2, stride2 = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.mg_linear_bwd_fused2_slabs_bf16(_p(dz2), dz2.shape[1], 128, _p(wt2), wt2.shape[1], _p(h1), h1.shape[1], _p(a), a.shape[1],
                                                   _p(rows), m, n_hidden, k0, _p(slab), slab.numel(), ctypes.byref(n_slabs),
                                                   ctypes.byref(stride1), ctypes.byref(off2), ctypes.byref(stride2), _stream()),
               'mg_linear_bwd_fused2_slabs_bf16')
    return (slab, n_slabs.value, (0, stride1.value, n_hidden * k0 + n_hidden), (off2.value, stride2.value, 128 * n_hidden + 128))


def copy_many(pairs):
    """dst.copy_(src) for every (dst, src) pair of equal-sized contiguous device tensors, MG_COPY_MAX pairs per launch (mg_copy_many):
    the tensors of a new batch into a captured step's static buffers as one launch instead of one per tensor."""
    lib = _lib.load()
    pairs = list(pairs)
    for i in range(0, len(pairs), _lib.COPY_MAX):
        chunk = pairs[i:i + _lib.COPY_MAX]
        descs = (_lib.CopyDesc * len(chunk))()
        for j, (dst, src) in enumerate(chunk):
            if (not dst.is_cuda or not src.is_cuda or dst.device != src.device or not dst.is_contiguous() or not src.is_contiguous()
                    or dst.dtype != src.dtype or dst.numel() != src.numel()):
                raise ValueError('copy_many: pairs of contiguous tensors of one device, dtype and size are required')
            descs[j].src, descs[j].dst, descs[j].bytes = src.data_ptr(), dst.data_ptr(), src.numel() * src.element_size()
        _lib.check(lib.mg_copy_many(ctypes.cast(descs, ctypes.c_void_p), len(chunk), _stream()), 'mg_copy_many')


def cast_params_bf16(weights, want_plain=True, want_t=()):
    """One launch: bf16 copies [N, pad_ld(K)] of every fp32 weight and, for the indices in `want_t`, the transposed
    copies [K, pad_ld(N)].  Returns (plain list, transposed list with None where not requested)."""
    lib = _lib.load()
    if len(weights) > _lib.CAST_MAX:
        raise ValueError('cast_params_bf16: at most %d matrices per call' % _lib.CAST_MAX)
    descs = (_lib.CastDesc * len(weights))()
    plain, trans = [], []
    for i, w in enumerate(weights):
        w = _require(w, torch.float32, 'weight')
        n, k = w.shape
        wb = torch.empty((n, pad_ld(k)), dtype=torch.bfloat16, device=w.device) if want_plain else None
        wt = torch.empty((k, pad_ld(n)), dtype=torch.bfloat16, device=w.device) if i in want_t else None
        descs[i].src, descs[i].rows, descs[i].cols = w.data_ptr(), n, k
        descs[i].dst, descs[i].ldd = (wb.data_ptr() if wb is not None else None), (wb.shape[1] if wb is not None else 0)
        descs[i].dst_t, descs[i].ldt = (wt.data_ptr() if wt is not None else None), (wt.shape[1] if wt is not None else 0)
        plain.append(wb)
        trans.append(wt)
    _lib.check(lib.mg_cast_params_bf16(ctypes.cast(descs, ctypes.c_void_p), len(weights), _stream()),
               'mg_cast_params_bf16')
    return plain, trans


WEIGHT_SHADOWS = os.environ.get('MORGANA_WEIGHT_SHADOWS', '1') != '0'      # A/B: 0 = a cast launch per weight and step, as before


def weight_operands(weights, transposed=False):
    """bf16 operands ([N, pad_ld(K)], or the transposes [K, pad_ld(N)]) of a list of fp32 weights: the shadows that live on the
    parameters (param_shadows: kept current by the optimiser's update kernel, stale ones re-cast by ONE batched launch), or a cast
    launch each (MORGANA_WEIGHT_SHADOWS=0)."""
    weights = list(weights)
    if not WEIGHT_SHADOWS:
        return [cast_transpose_bf16(w) if transposed else cast_pad_bf16(_require(w, torch.float32, 'weight')) for w in weights]
    out = []
    for i in range(0, len(weights), _lib.CAST_MAX):      # mg_cast_params_bf16 takes at most MG_CAST_MAX descriptors per launch
        part = weights[i:i + _lib.CAST_MAX]
        plain, trans = param_shadows(part, want_t=tuple(range(len(part))) if transposed else ())
        out += trans if transposed else plain
    return out


def mark_updated(param):
    """A parameter's storage was written by something torch's version counter does not see (a HIP kernel on ``param.data``, a
    collective on ``param.data``): its bf16 operand copies (param_shadows) are stale from here on."""
    param._mg_updates = getattr(param, '_mg_updates', 0) + 1


def param_shadows(weights, want_t=()):
    """bf16 operands of a run of fp32 weight matrices: ([N, pad_ld(K)] copies, transposes [K, pad_ld(N)] for the indices in
    ``want_t``, None elsewhere).  The copies live ON the parameter (``w._mg_shadow``) and are kept current by
    ``morgana_amd.optim.Adam``'s update kernel, which re-casts every weight it has just changed (mg_adam_step_plan_f32), so a training
    step launches no cast at all; a weight changed by anything else (``load_state_dict``, another optimiser: its torch version
    counter moves; or one of our updates that did not refresh it: ``_mg_updates`` moves) is simply cast again here."""
    stale = []
    for i, w in enumerate(weights):
        sh = getattr(w, '_mg_shadow', None)
        ok = (sh is not None and sh['version'] == (w._version, getattr(w, '_mg_updates', 0)) and sh['plain'].device == w.device
              and (i not in want_t or sh['t'] is not None))
        if not ok:
            stale.append(i)
    if stale:
        descs = (_lib.CastDesc * len(stale))()
        for j, i in enumerate(stale):
            w = _require(weights[i], torch.float32, 'weight')
            n, k = w.shape
            old = getattr(w, '_mg_shadow', None)
            plain = old['plain'] if old is not None and old['plain'].device == w.device else \
                torch.zeros((n, pad_ld(k)), dtype=torch.bfloat16, device=w.device)
            trans = old['t'] if old is not None and old['t'] is not None and old['t'].device == w.device else None
            if trans is None and i in want_t:
                trans = torch.zeros((k, pad_ld(n)), dtype=torch.bfloat16, device=w.device)
            descs[j].src, descs[j].rows, descs[j].cols = w.data_ptr(), n, k
            descs[j].dst, descs[j].ldd = plain.data_ptr(), plain.shape[1]
            descs[j].dst_t, descs[j].ldt = (trans.data_ptr(), trans.shape[1]) if trans is not None else (None, 0)
            w._mg_shadow = {'plain': plain, 't': trans, 'version': (w._version, getattr(w, '_mg_updates', 0))}
        _lib.check(_lib.load().mg_cast_params_bf16(ctypes.cast(descs, ctypes.c_void_p), len(stale), _stream()), 'mg_cast_params_bf16')
    return [w._mg_shadow['plain'] for w in weights], [w._mg_shadow['t'] if i in want_t else None for i, w in enumerate(weights)]


def refresh_shadows(params):
    """Re-cast the EXISTING bf16 operand copies of ``params`` (plain, and transposed where one is allocated) from their fp32 weights:
    one batched launch per MG_CAST_MAX parameters on the current stream, no allocation, no stamp (the caller stamps).  Used by
    ``optim.Adam`` for the copies its update kernel's plan has no room for, so that EVERY copy is current when an update ends -
    inside a captured step as well, where nothing on the host can notice a stale copy later."""
    lib = _lib.load()
    for i in range(0, len(params), _lib.CAST_MAX):
        chunk = params[i:i + _lib.CAST_MAX]
        descs = (_lib.CastDesc * len(chunk))()
        for j, w in enumerate(chunk):
            sh = w._mg_shadow
            n, k = w.shape
            descs[j].src, descs[j].rows, descs[j].cols = w.data_ptr(), n, k
            descs[j].dst, descs[j].ldd = sh['plain'].data_ptr(), sh['plain'].shape[1]
            descs[j].dst_t, descs[j].ldt = (sh['t'].data_ptr(), sh['t'].shape[1]) if sh['t'] is not None else (None, 0)
        _lib.check(lib.mg_cast_params_bf16(ctypes.cast(descs, ctypes.c_void_p), len(chunk), _stream()), 'mg_cast_params_bf16')


# ------------------------------------------------------------------------------------------------------------------ active dropout
_dropout_state = {}      # device index -> int64 (1,) step counter on the device (mg_dropout_advance increments it per call)


def dropout_seed():
    """The 64-bit key of the dropout masks: torch's seed (``torch.manual_seed`` makes runs repeatable, as it does for nn.Dropout)."""
    return int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF


def dropout_state(device):
    """The device's dropout step counter (created on first use, OUTSIDE any stream capture: a counter zero-filled inside a capture would
    live in the graph's private pool and be reset by every replay - all replays would then draw the same masks, silently)."""
    device = torch.device(device)
    index = device.index if device.index is not None else torch.cuda.current_device()
    state = _dropout_state.get(index)
    if state is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError('dropout: the device step counter does not exist yet and cannot be created inside a stream capture (every '
                               'replay would reset it and repeat its masks): call morgana_amd.ops.dropout_state(device) - or run one eager '
                               'step - before the capture (graphs.GraphedTrainStep does)')
        state = _dropout_state[index] = torch.zeros(1, dtype=torch.int64, device=device)
    return state


def dropout_draw(device):
    """Reserve one draw of the device's dropout step counter: returns a (1,) int64 device tensor holding the counter value this call
    (and its backward) uses; the device counter moves on in stream order - also inside a replayed HIP graph (csrc/dropout.hip)."""
    state = dropout_state(device)
    used = torch.empty(1, dtype=torch.int64, device=device)
    _lib.check(_lib.load().mg_dropout_advance(_p(state), _p(used), _stream()), 'mg_dropout_advance')
    return used


def dropout(x, p, seed, site, used, out=None):
    """y = x * keep / (1 - p) with the mask of (seed, site, used[0], element index) - mg_dropout.  x: contiguous fp32 or bf16; ``out``
    may be x (in place).  The same call on a gradient with the same (seed, site, used) is the backward."""
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_contiguous():
        raise TypeError('dropout: a contiguous float32 or bfloat16 tensor is required')
    y = torch.empty_like(x) if out is None else out
    _lib.check(_lib.load().mg_dropout(_p(x), _p(y), x.numel(), int(x.dtype == torch.bfloat16), float(p), int(seed), int(site) & 0xFFFFFFFF,
                                      _p(used), _stream()), 'mg_dropout')
    return y


# ----------------------------------------------------------------------------------------- precision 'bf16x3' (split-bf16 operands)
CAPTURE_EPOCH = [0]      # bumped by graphs.GraphedTrainStep at the start of every capture (see x3_weight_operands)


def begin_capture_epoch():
    CAPTURE_EPOCH[0] += 1


SPLIT3_COLSUM_BLOCKS = int(os.environ.get('MORGANA_SPLIT3_COLSUM_BLOCKS', '512'))      # workgroups (= slabs) of a split that also sums its columns; C2 bf16x3 step by count: 64 0.555, 128 0.488, 256 0.458, 512 0.457, 1024 0.463, 2048 0.480 ms


def split3_colsum_ok(cols):
    """Plane widths whose 8-column chunks divide a 256-thread workgroup (mg_split3_bf16's column-sum form)."""
    return 256 % (pad_ld(cols) // 8) == 0


def split3(jobs):
    """Operand splits of precision mode 'bf16x3' (csrc/split3.hip): ``jobs`` = [(fp32 2-D tensor, order, transpose)] with order 0 =
    [hi | hi | lo] (activation side), 1 = [hi | lo | hi] (weight side), 2 = two separate planes, 3 / 4 = three row-stacked planes
    [hi ; hi ; lo] / [hi ; lo ; hi]; one batched launch per MG_SPLIT3_MAX jobs.  Returns one bf16 tensor per job: (rows, 3 ldp) -
    (cols, 3 ldp) with ``transpose`` - or (2 or 3, rows, ldp) for orders 2-4, where ldp = pad_ld(columns of a plane).  A job with a
    sixth element True (plain layouts) returns (planes, slabs): the per-workgroup column sums of the split values (``slab_reduce``)."""
    lib = _lib.load()
    outs = []
    for i in range(0, len(jobs), _lib.SPLIT3_MAX):
        chunk = jobs[i:i + _lib.SPLIT3_MAX]
        descs = (_lib.Split3Desc * len(chunk))()
        for j, job in enumerate(chunk):
            x, order, transpose = job[:3]
            extra = int(job[3]) if len(job) > 3 else 0          # zero rows appended behind the split (not with transpose)
            sig = job[4] if len(job) > 4 else None              # fp32 sigmoid outputs of x's shape: split x * s * (1 - s) instead of x
            want_colsum = bool(job[5]) if len(job) > 5 else False
            if sig is not None and (transpose or sig.dtype != torch.float32 or tuple(sig.shape) != tuple(x.shape) or sig.stride(1) != 1):
                raise ValueError('split3: the fused sigmoid gradient needs an fp32 tensor of the operand\'s shape (plain layouts)')
            if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
                raise TypeError('split3: operands must be 2-D float32 with unit column stride')
            if extra and transpose:
                raise ValueError('split3: extra zero rows go with the plain layouts')
            rows, cols = x.shape
            ldp = pad_ld(rows if transpose else cols)
            stacked = order >= 2
            out = torch.empty((2 if order == 2 else 3, rows + extra, ldp) if stacked else ((cols if transpose else rows) + extra, 3 * ldp),
                              dtype=torch.bfloat16, device=x.device)
            if extra:
                (out[:, rows:] if stacked else out[rows:]).zero_()
            descs[j].src, descs[j].rows, descs[j].cols, descs[j].lds = x.data_ptr(), rows, cols, x.stride(0)
            descs[j].dst, descs[j].ldp, descs[j].order, descs[j].transpose = out.data_ptr(), ldp, int(order), int(bool(transpose))
            descs[j].plane_rows = rows + extra if stacked else 0
            descs[j].sig, descs[j].ldsig = (sig.data_ptr(), sig.stride(0)) if sig is not None else (None, 0)
            if want_colsum:
                if transpose or 256 % (ldp // 8) != 0:
                    raise ValueError('split3: column sums go with the plain layouts and plane widths that divide 2048')
                slabs = torch.empty((SPLIT3_COLSUM_BLOCKS, ldp), dtype=torch.float32, device=x.device)
                descs[j].colsum, descs[j].colsum_blocks = slabs.data_ptr(), SPLIT3_COLSUM_BLOCKS
                outs.append((out, slabs))
            else:
                descs[j].colsum, descs[j].colsum_blocks = None, 0
                outs.append(out)
        if chunk:
            _lib.check(lib.mg_split3_bf16(ctypes.cast(descs, ctypes.c_void_p), len(chunk), _stream()), 'mg_split3_bf16')
    return outs


def x3_weight_operands(weights, want_t=()):
    """Weight-side operands of 'bf16x3' for a run of fp32 weight matrices [N, K]: ([N, 3 pad_ld(K)] splits in order 1, and for the
    indices in ``want_t`` the order-0 splits of W^T [K, 3 pad_ld(N)] (the dgrad operand, against gradients in order 1), None elsewhere).  Cached on the parameter
    under the same version stamps as the bf16 operand copies (param_shadows); every stale one is re-split by ONE batched launch."""
    jobs, slots = [], []
    # inside a stream capture a split that is current NOW says nothing about the replays: every step of a captured graph re-splits
    # the weights it reads (once per capture epoch and update count - graphs.GraphedTrainStep opens an epoch per capture)
    capturing = bool(weights) and weights[0].is_cuda and torch.cuda.is_current_stream_capturing()
    for i, w in enumerate(weights):
        st = getattr(w, '_mg_x3', None)
        stamp = (w._version, getattr(w, '_mg_updates', 0), CAPTURE_EPOCH[0] if capturing else -1)
        if st is None or st['version'] != stamp or st['plain'].device != w.device:
            st = w._mg_x3 = {'plain': None, 't': None, 'version': stamp}
        if st['plain'] is None:
            jobs.append((_require(w, torch.float32, 'weight'), 1, False))
            slots.append((st, 'plain'))
        if i in want_t and st['t'] is None:
            jobs.append((_require(w, torch.float32, 'weight'), 0, True))          # W^T [hi | hi | lo]: against gradients split [hi | lo | hi]
            slots.append((st, 't'))
    for (st, key), out in zip(slots, split3(jobs)):
        st[key] = out
    return [w._mg_x3['plain'] for w in weights], [w._mg_x3['t'] if i in want_t else None for i, w in enumerate(weights)]


def linear_fwd_x3(a3, rows, m, w3, bias, n, act):
    """fp32 (m, n) = act(gather(A, rows) W^T + bias) from split operands: a3 (R, 3 ldp) order 0, w3 (n, 3 ldp) order 1 - the bf16
    tile programs over a contraction index of 3 ldp (mg_linear_fwd_bf16, fp32 output)."""
    if a3.shape[1] != w3.shape[1]:
        raise ValueError('linear_fwd_x3: operand planes differ (%d vs %d columns)' % (a3.shape[1], w3.shape[1]))
    y = linear_fwd_bf16(a3, rows, m, a3.shape[1], w3, bias, n, act, out_f32=True)
    return y if y.shape[1] == n else y[:, :n].contiguous()


def linear_dgrad_x3(g3, m, wt3, k):
    """fp32 (m, k) = dY W from split operands: g3 (m, 3 ldp(n)) order 1 [hi | lo | hi], wt3 = split of W^T (k, 3 ldp(n)) order 0."""
    if g3.shape[1] != wt3.shape[1]:
        raise ValueError('linear_dgrad_x3: operand planes differ (%d vs %d columns)' % (g3.shape[1], wt3.shape[1]))
    dx = linear_dgrad_bf16(g3, m, g3.shape[1], wt3, k, None, out_f32=True)
    return dx if dx.shape[1] == k else dx[:, :k].contiguous()


def linear_wgrad_x3_rows(g1, colsum, a0, n, k, out_w=None, out_b=None, accumulate=False):
    """dW (n, k), db (n,) as ONE launch from the three-plane buffers the other products of the layer use anyway: g1 (m, 3 ldp(n)) = the
    gradient split [hi | lo | hi] (order 1), a0 (m, 3 ldp(k)) = the forward's activation split [hi | hi | lo] (order 0).  Read as
    (3 m, ldp) matrices they are row-interleaved stacks pairing (hi, hi), (lo, hi), (hi, lo): the three products of the weight
    gradient in one contraction over 3 m rows.  db = the ordered sum of ``colsum`` (split3's column sums of the fp32 gradient)."""
    lib = _lib.load()
    if g1.dim() != 2 or a0.dim() != 2 or g1.shape[0] != a0.shape[0] or g1.shape[1] % 3 or a0.shape[1] % 3:
        raise ValueError('linear_wgrad_x3_rows: operands must be three-plane buffers of equal row count')
    m3, ldn, ldk = 3 * g1.shape[0], g1.shape[1] // 3, a0.shape[1] // 3
    if out_w is None:
        both = torch.empty((n * k + n,), dtype=torch.float32, device=g1.device)
        dw, db = both[:n * k].view(n, k), (both[n * k:] if colsum is not None else None)
    else:
        dw, db = out_w, (out_b if colsum is not None else None)
    ws = workspace(lib.mg_linear_wgrad_workspace_bytes(m3, n, k), g1.device)
    _lib.check(lib.mg_linear_wgrad_bf16(_p(g1), ldn, _p(a0), ldk, None, m3, n, k, _p(dw), None, int(bool(accumulate)), _p(ws), ws.numel(),
                                        _stream()), 'mg_linear_wgrad_bf16')
    if db is not None:
        slab_reduce(colsum, colsum.shape[0], colsum.shape[1], n, db, accumulate=accumulate)
    return dw, db


def linear_wgrad_x3_stacked(g3, colsum, a3, n, k, out_w=None, out_b=None, accumulate=False):
    """dW (n, k), db (n,) of split operands as ONE launch: g3 (3, m, ldp(n)) = [hi ; hi ; lo] (split3 order 3), a3 (3, m, ldp(k)) =
    [hi ; lo ; hi] (order 4): [hi ; hi ; lo]^T [hi ; lo ; hi] over 3 m rows = the three products of ``linear_wgrad_x3``.  db = the
    ordered sum of ``colsum`` (the split pass's per-workgroup column sums of the fp32 gradient: exact), or None."""
    lib = _lib.load()
    if g3.shape[0] != 3 or a3.shape[0] != 3 or g3.shape[1] != a3.shape[1]:
        raise ValueError('linear_wgrad_x3_stacked: operands must be three row-stacked planes of equal row count')
    m3 = 3 * g3.shape[1]
    if out_w is None:
        both = torch.empty((n * k + n,), dtype=torch.float32, device=g3.device)
        dw, db = both[:n * k].view(n, k), (both[n * k:] if colsum is not None else None)
    else:
        dw, db = out_w, (out_b if colsum is not None else None)
    ws = workspace(lib.mg_linear_wgrad_workspace_bytes(m3, n, k), g3.device)
    _lib.check(lib.mg_linear_wgrad_bf16(_p(g3), g3.shape[2], _p(a3), a3.shape[2], None, m3, n, k, _p(dw), None, int(bool(accumulate)),
                                        _p(ws), ws.numel(), _stream()), 'mg_linear_wgrad_bf16')
    if db is not None:
        slab_reduce(colsum, colsum.shape[0], colsum.shape[1], n, db, accumulate=accumulate)
    return dw, db


def linear_wgrad_x3(g2, a2, rows, m, n, k, out_w=None, out_b=None, accumulate=False):
    """dW (n, k), db (n,) from split operands in separate planes (split3 order 2): g2 (2, m, ldp(n)), a2 (2, R, ldp(k)).  A weight
    gradient contracts over the ROWS, so the three products are three accumulating launches on plane pairs: hi^T hi (+ the bias
    sums of hi), hi^T lo, lo^T hi (+ the bias sums of lo)."""
    lib = _lib.load()
    if out_w is None:
        both = torch.empty((n * k + n,), dtype=torch.float32, device=g2.device)
        dw, db = both[:n * k].view(n, k), both[n * k:]
    else:
        dw, db = out_w, out_b
    ws = workspace(lib.mg_linear_wgrad_workspace_bytes(m, n, k), g2.device)
    plans = ((0, 0, True), (0, 1, False), (1, 0, True))                   # (dY plane, A plane, with bias sums)
    for idx, (gp, ap, with_b) in enumerate(plans):
        acc = int(bool(accumulate)) if idx == 0 else 1
        _lib.check(lib.mg_linear_wgrad_bf16(_p(g2[gp]), g2.shape[2], _p(a2[ap]), a2.shape[2], _p(rows), m, n, k, _p(dw),
                                            _p(db) if (with_b and db is not None) else None, acc, _p(ws), ws.numel(), _stream()),
                   'mg_linear_wgrad_bf16')
    return dw, db


# ------------------------------------------------------------------------------- precision 'bf16x3', the FUSED step on pair planes
# (include/morgana_hip.h, "The FUSED step of precision mode 'bf16x3' on PAIR PLANES").  A pair is a bf16 (rows, 2 ldp) buffer [hi | lo].
F0_TAIL_X3_SLAB = 4292                 # MG_F0_TAIL_X3_SLAB: db2 128 | dW3 4096 | db3 32 | dW4 32 | db4 | loss | 2 unused
F0_TAIL_X3_N = F0_TAIL_X3_SLAB - 2     # ... up to and including the loss


def split_pair(x2d, transpose=False, extra_rows=0):
    """[hi | lo] pair planes of an fp32 matrix (mg_split3_bf16 order 5): (rows + extra_rows, 2 pad_ld(cols)) bf16, the extra rows zero;
    ``transpose``: the pair of x2d^T, (cols, 2 pad_ld(rows))."""
    lib = _lib.load()
    x2d = _require(x2d, torch.float32, 'operand')
    if x2d.dim() != 2:
        raise ValueError('split_pair: a 2-D operand is required')
    rows, cols = x2d.shape
    if transpose and extra_rows:
        raise ValueError('split_pair: extra zero rows go with the plain layout')
    ldp = pad_ld(rows if transpose else cols)
    out = torch.empty(((cols if transpose else rows) + extra_rows, 2 * ldp), dtype=torch.bfloat16, device=x2d.device)
    if extra_rows:
        out[rows:].zero_()
    descs = (_lib.Split3Desc * 1)()
    d = descs[0]
    d.src, d.rows, d.cols, d.lds = x2d.data_ptr(), rows, cols, x2d.stride(0)
    d.dst, d.ldp, d.order, d.transpose, d.plane_rows = out.data_ptr(), ldp, 5, int(bool(transpose)), 0
    d.sig, d.ldsig, d.colsum, d.colsum_blocks = None, 0, None, 0
    _lib.check(lib.mg_split3_bf16(ctypes.cast(descs, ctypes.c_void_p), 1, _stream()), 'mg_split3_bf16')
    return out


def pair_shadows(weights, want_t=()):
    """Pair-plane operands of a run of fp32 weight matrices: ([N, 2 pad_ld(K)] pairs, pairs of W^T [K, 2 pad_ld(N)] for the indices in
    ``want_t``, None elsewhere).  As ``param_shadows``: the pairs live ON the parameter (``w._mg_pair``), ``morgana_amd.optim.Adam``'s
    update kernel re-splits every weight it has just changed (mg_adam_shadow.pair), so a training step launches no split; a weight
    changed by anything else is split again here."""
    plain, trans = [], []
    for i, w in enumerate(weights):
        sh = getattr(w, '_mg_pair', None)
        ok = (sh is not None and sh['version'] == (w._version, getattr(w, '_mg_updates', 0)) and sh['plain'].device == w.device
              and (i not in want_t or sh['t'] is not None))
        if not ok:
            w32 = _require(w, torch.float32, 'weight')
            n, k = w32.shape
            old = sh if sh is not None and sh['plain'].device == w.device else None
            # re-split INTO the existing buffers (a captured graph reads them by address)
            new_plain = split_pair(w32.detach())
            if old is not None:
                old['plain'].copy_(new_plain)
                new_plain = old['plain']
            new_t = old['t'] if old is not None else None
            if i in want_t or new_t is not None:
                fresh = split_pair(w32.detach(), transpose=True)
                if new_t is not None:
                    new_t.copy_(fresh)
                else:
                    new_t = fresh
            w._mg_pair = {'plain': new_plain, 't': new_t, 'version': (w._version, getattr(w, '_mg_updates', 0))}
        plain.append(w._mg_pair['plain'])
        trans.append(w._mg_pair['t'] if i in want_t else None)
    return plain, trans


def x3_step_ok(n_table_rows, m, k0, n0, n1, extra=None):
    """Shapes of the fused 'bf16x3' phone-rate step: Linear(k0 -> 512) + Sigmoid, Linear(512 -> 128) on a phone table whose row
    count takes the wide weight-gradient tiles, k0's plane 640 or 512 columns wide."""
    extra = PHONE_RATE_EXTRA if extra is None else extra
    rows = n_table_rows + extra
    ldp = pad_ld(k0)
    return (n0 == 512 and n1 == 128 and 4096 <= rows and rows < m and
            ((ldp == 640 and 512 < k0 <= 640) or (ldp == 512 and 384 < k0 <= 512)))


def phone_front_x3(front, linear):
    """``phone_front`` with the first layer on PAIR PLANES: ``front`` = (dur, target, seq_len, t, extra) or None (the GEMM alone),
    ``linear`` = (a pair (M, 2 pa), k, w pair (n, 2 pw), bias, n, act).  Returns the front's six results (when asked for) + y, the pair
    (M, 2 n) of act(a w^T + bias) (mg_phone_front_linear_fwd_x3)."""
    lib = _lib.load()
    a, k, w2, bias, n, act = linear
    m = a.shape[0]
    y = torch.empty((m, 2 * n), dtype=torch.bfloat16, device=a.device)
    if front is None:
        _lib.check(lib.mg_phone_front_linear_fwd_x3(None, 0, 0, 0, None, None, 0, None, None, 0, None, None, None, None, None, 0,
                                                    _p(a), a.shape[1], m, k, _p(w2), w2.shape[1], _p(bias), n, _p(y), 2 * n, act, _stream()),
                   'mg_phone_front_linear_fwd_x3')
        return (y,)
    dur, target, seq_len, t, extra = front
    dur = _require(dur, torch.int64, 'dur')
    target = _require(target, torch.float32, 'targets')
    b, p = dur.shape
    if target.numel() != b * t:
        raise ValueError('phone_front_x3: %d targets for %d x %d frames' % (target.numel(), b, t))
    r = b * p
    rows = torch.empty((2, b, t), dtype=torch.int32, device=dur.device)
    seg = torch.empty((2, r), dtype=torch.int32, device=dur.device)
    stats = torch.empty((2, r + extra), dtype=torch.float32, device=dur.device)
    ws = torch.empty(lib.mg_phone_target_stats_workspace_bytes(r, extra), dtype=torch.uint8, device=dur.device)
    _lib.check(lib.mg_phone_front_linear_fwd_x3(_p(dur), b, p, int(t), _p(target), _p(seq_len), extra, _p(rows[0]), _p(rows[1]), r, _p(seg[0]),
                                                _p(seg[1]), _p(stats[0]), _p(stats[1]), _p(ws), ws.numel(), _p(a), a.shape[1], m, k, _p(w2),
                                                w2.shape[1], _p(bias), n, _p(y), 2 * n, act, _stream()), 'mg_phone_front_linear_fwd_x3')
    return rows[0], rows[1], seg, stats[0], stats[1], ws, y


X3_FWD_PARTS = int(os.environ.get('MORGANA_X3_FWD_PARTS', '3'))      # A/B: 1 = the 128-wide layer's three passes in one workgroup per tile


def linear_fwd_x3_f32(a2, k, w2, bias, n, act, parts=1):
    """fp32 (M, n) = act(a w^T + bias) from pair-plane operands a2 (M, 2 pa), w2 (n, 2 pw) (mg_linear_fwd_x3_f32).  ``parts`` = 3
    (no activation): (3, M, n) partial sums of the three products, from three sets of workgroups - the consumer adds them."""
    m = a2.shape[0]
    y = torch.empty((m, n) if parts == 1 else (parts, m, n), dtype=torch.float32, device=a2.device)
    _lib.check(_lib.load().mg_linear_fwd_x3_f32(_p(a2), a2.shape[1], m, k, _p(w2), w2.shape[1], _p(bias), n, _p(y), n, act, int(parts), _stream()),
               'mg_linear_fwd_x3_f32')
    return y


def f0_tail_rows_x3(z2, w3, b3, w4, b4, ybar, weight, keep_slabs=False):
    """``f0_tail_rows_f32`` of the fused 'bf16x3' step (mg_f0_tail_rows_x3): returns (pred (rows,), dz2 PAIR (rows, 256) bf16, slab buffer,
    n_slabs, stride); the slabs [db2 | dW3 | db3 | dW4 | db4 | loss | pad] are left unreduced for the caller (``slab_reduce``,
    mg_expand_column_reduce_f32 or the update kernel's plan).  ``keep_slabs``: a buffer of their own (they outlive the next launches)."""
    lib = _lib.load()
    z2 = _require(z2, torch.float32, 'pre-activations')
    z_parts = z2.shape[0] if z2.dim() == 3 else 1              # (3, rows, 128): three partial sums to add (linear_fwd_x3_f32 parts = 3)
    m = z2.shape[-2]
    if (z_parts not in (1, 3) or z2.shape[-1] != 128 or tuple(w3.shape) != (32, 128) or tuple(w4.shape) != (1, 32) or ybar.numel() != m
            or weight.numel() != m):
        raise ValueError('f0_tail_rows_x3: needs (rows, 128) pre-activations, a 128 -> 32 -> 1 tail and one statistics row per table row')
    pred = torch.empty((m,), dtype=torch.float32, device=z2.device)
    dz2 = torch.empty((m, 256), dtype=torch.bfloat16, device=z2.device)
    ws = (_tail_slabs if keep_slabs else workspace)(lib.mg_f0_tail_rows_x3_workspace_bytes(m), z2.device)
    n_slabs, stride = ctypes.c_int(0), ctypes.c_int64(0)
    _lib.check(lib.mg_f0_tail_rows_x3(_p(z2), z2.shape[-1], z_parts, _p(_require(w3, torch.float32, 'w3')), _p(b3), _p(_require(w4, torch.float32, 'w4')),
                                      _p(b4), _p(ybar), _p(weight), m, _p(pred), _p(dz2), 256, None, _p(ws), ws.numel(),
                                      ctypes.byref(n_slabs), ctypes.byref(stride), _stream()), 'mg_f0_tail_rows_x3')
    return pred, dz2, ws, n_slabs.value, stride.value


X3_L2TAIL = os.environ.get('MORGANA_X3_L2TAIL', '1') != '0'      # A/B: 0 = the 128-wide layer and the fp32 tail as two launches


def f0_l2tail_x3(h1, w2p, b2, w3, b3, w4, b4, ybar, weight, keep_slabs=False):
    """``linear_fwd_x3_f32`` of the 512 -> 128 layer and ``f0_tail_rows_x3`` as ONE launch (mg_f0_l2tail_x3): h1 pair (rows, 1024), w2p
    pair (128, 1024).  Returns what ``f0_tail_rows_x3`` returns."""
    lib = _lib.load()
    m = h1.shape[0]
    if (h1.shape[1] != 1024 or tuple(w2p.shape) != (128, 1024) or tuple(w3.shape) != (32, 128) or tuple(w4.shape) != (1, 32)
            or ybar.numel() != m or weight.numel() != m):
        raise ValueError('f0_l2tail_x3: needs the (rows, 1024) pair of a 512-wide activation, a 512 -> 128 -> 32 -> 1 tail and one statistics row per table row')
    pred = torch.empty((m,), dtype=torch.float32, device=h1.device)
    dz2 = torch.empty((m, 256), dtype=torch.bfloat16, device=h1.device)
    ws = (_tail_slabs if keep_slabs else workspace)(lib.mg_f0_l2tail_x3_workspace_bytes(m), h1.device)
    n_slabs, stride = ctypes.c_int(0), ctypes.c_int64(0)
    _lib.check(lib.mg_f0_l2tail_x3(_p(h1), h1.shape[1], _p(w2p), w2p.shape[1], _p(b2), _p(_require(w3, torch.float32, 'w3')), _p(b3),
                                   _p(_require(w4, torch.float32, 'w4')), _p(b4), _p(ybar), _p(weight), m, _p(pred), _p(dz2), 256, None,
                                   _p(ws), ws.numel(), ctypes.byref(n_slabs), ctypes.byref(stride), _stream()), 'mg_f0_l2tail_x3')
    return pred, dz2, ws, n_slabs.value, stride.value


def linear_wgrad_dgrad_x3(dy2, a2, m, n, k, wt2, slab=None, colsum=None):
    """The 512 -> 128 layer's backward of the fused 'bf16x3' step as one grid (mg_linear_wgrad_dgrad_x3): dy2 pair (m, 2 * 128), a2 = the
    sigmoid output's pair (m, 2 * 512), wt2 = the pair of W^T (k, 2 * 128).  Returns (slab buffer, n_slabs, stride, dx pair (m, 2 k),
    colsum buffer (n_colsum, k) f32 - its ordered sum is the bias gradient of the layer below, n_colsum)."""
    lib = _lib.load()
    slab = _slab_buffer(slab, lib.mg_linear_wgrad_workspace_bytes(m, n, k), dy2.device)
    need = lib.mg_linear_wgrad_dgrad_x3_colsum_floats(m, k)
    colsum = _slab_buffer(colsum, 4 * need, dy2.device)
    dx = torch.empty((m, 2 * k), dtype=torch.bfloat16, device=dy2.device)
    n_slabs, stride, n_colsum = ctypes.c_int(0), ctypes.c_int64(0), ctypes.c_int(0)
    _lib.check(lib.mg_linear_wgrad_dgrad_x3(_p(dy2), dy2.shape[1], _p(a2), a2.shape[1], m, n, k, _p(wt2), wt2.shape[1], _p(dx), 2 * k, _p(slab),
                                            slab.numel(), ctypes.byref(n_slabs), ctypes.byref(stride), _p(colsum), colsum.numel() // 4,
                                            ctypes.byref(n_colsum), _stream()), 'mg_linear_wgrad_dgrad_x3')
    return slab, n_slabs.value, stride.value, dx, colsum, n_colsum.value


def linear_wgrad_slabs_x3(dy2, a2, m, n, k, slab=None):
    """``linear_wgrad_slabs_bf16`` on pair planes (mg_linear_wgrad_slabs_x3): slabs of n k weight partials (stride n k + n), no bias sums."""
    lib = _lib.load()
    slab = _slab_buffer(slab, lib.mg_linear_wgrad_workspace_bytes(m, n, k), dy2.device)
    n_slabs, stride = ctypes.c_int(0), ctypes.c_int64(0)
    _lib.check(lib.mg_linear_wgrad_slabs_x3(_p(dy2), dy2.shape[1], _p(a2), a2.shape[1], m, n, k, _p(slab), slab.numel(), ctypes.byref(n_slabs),
                                            ctypes.byref(stride), _stream()), 'mg_linear_wgrad_slabs_x3')
    return slab, n_slabs.value, stride.value


def f0_tail(h2, w3, b3, w4, b4, target, seq_len, b, t, grads_out, grad_scale=1.0):
    """Fused layers 3-4 + masked MSE, forward and backward (mg_f0_tail_bf16).  Returns (pred (b*t,), loss 0-d, dz2)."""
    lib = _lib.load()
    m = b * t
    pred = torch.empty((m,), dtype=torch.float32, device=h2.device)
    n_grads = 32 * 128 + 32 + 32 + 1
    # a gradient buffer with one spare float behind the tail's gradients receives the loss from the same reduce launch
    loss = grads_out[n_grads] if grads_out.numel() > n_grads else torch.empty((), dtype=torch.float32, device=h2.device)
    dz2 = torch.empty_like(h2)
    nbytes = lib.mg_f0_tail_workspace_bytes(m)
    ws = workspace(nbytes, h2.device)
    _lib.check(lib.mg_f0_tail_bf16(_p(h2), h2.shape[1], w3.shape[1], _p(w3), _p(b3), _p(w4), _p(b4), _p(target),
                                   _p(seq_len), b, t, float(grad_scale), _p(pred), _p(loss), _p(dz2), _p(grads_out), 0,
                                   _p(ws), ws.numel(), _stream()), 'mg_f0_tail_bf16')
    return pred, loss, dz2


def f0_tail_rows(h2, w3, b3, w4, b4, target_rows, row_weight, grads_out, grad_scale=1.0):
    """f0_tail on table rows that each stand for a group of frames (mg_f0_tail_rows_bf16): loss = sum_m w[m] (pred[m] - y[m])^2.
    Returns (pred (rows,), loss 0-d, dz2 (rows, ld))."""
    lib = _lib.load()
    m = h2.shape[0]
    pred = torch.empty((m,), dtype=torch.float32, device=h2.device)
    n_grads = 32 * 128 + 32 + 32 + 1
    loss = grads_out[n_grads] if grads_out.numel() > n_grads else torch.empty((), dtype=torch.float32, device=h2.device)
    dz2 = torch.empty_like(h2)
    ws = workspace(lib.mg_f0_tail_workspace_bytes(m), h2.device)
    _lib.check(lib.mg_f0_tail_rows_bf16(_p(h2), h2.shape[1], w3.shape[1], _p(w3), _p(b3), _p(w4), _p(b4), _p(target_rows), _p(row_weight),
                                        m, float(grad_scale), _p(pred), _p(loss), _p(dz2), _p(grads_out), 0, _p(ws), ws.numel(),
                                        _stream()), 'mg_f0_tail_rows_bf16')
    return pred, loss, dz2


def l2tail_ok(w2, w3, w4, act2):
    """The 512 -> 128 sigmoid layer and the 128 -> 32 -> 1 tail as ONE kernel (csrc/l2tail_bf16.hip): shapes it is built for."""
    return (tuple(w2.shape) == (128, 512) and tuple(w3.shape) == (32, 128) and tuple(w4.shape) == (1, 32) and act2 == ACT_SIGMOID
            and os.environ.get('MORGANA_L2TAIL', '1') != '0')


def f0_l2tail(h1, w2_bf, b2, w3, b3, w4, b4, target, seq_len, b, t, grads_out, grad_scale=1.0, defer=False):
    """Layer 2 (512 -> 128, sigmoid) + layers 3-4 + masked MSE, forward and backward, in one pass over H1 (mg_f0_l2tail_bf16).
    Returns (pred (b*t,), loss 0-d, dz2 (b*t, 128) bf16); the 128-wide activation is never written.
    ``defer``: the reduce launch is NOT made (mg_f0_l2tail_slabs_bf16); a fourth return value describes what is left - the sum of the
    workgroups' slabs into ``grads_out`` (gradients and, behind them, the loss) - for optim.Adam.defer_tail / finish_deferred_tail."""
    lib = _lib.load()
    m = b * t
    pred = torch.empty((m,), dtype=torch.float32, device=h1.device)
    n_grads = 32 * 128 + 32 + 32 + 1
    loss = grads_out[n_grads] if grads_out.numel() > n_grads else torch.empty((), dtype=torch.float32, device=h1.device)
    dz2 = torch.empty((m, 128), dtype=torch.bfloat16, device=h1.device)
    if defer:
        if grads_out.numel() <= n_grads:
            raise ValueError('f0_l2tail(defer=True): the gradient buffer needs one float behind the gradients for the loss')
        ws = _tail_slabs(lib.mg_f0_l2tail_workspace_bytes(m), h1.device)
        n_slabs = ctypes.c_int(0)
        _lib.check(lib.mg_f0_l2tail_slabs_bf16(_p(h1), h1.shape[1], 512, _p(w2_bf), w2_bf.shape[1], 128, _p(b2), _p(w3), _p(b3), _p(w4),
                                               _p(b4), _p(target), _p(seq_len), b, t, float(grad_scale), _p(pred), _p(dz2), 128, _p(ws),
                                               ws.numel(), ctypes.byref(n_slabs), _stream()), 'mg_f0_l2tail_slabs_bf16')
        tail = dict(rows=None, partials=None, ws=ws, n=n_grads + 1, stride=int(lib.mg_f0_l2tail_slab_stride()), n_slabs=n_slabs.value,
                    grads_out=grads_out)
        return pred, loss, dz2, tail
    ws = workspace(lib.mg_f0_l2tail_workspace_bytes(m), h1.device)
    _lib.check(lib.mg_f0_l2tail_bf16(_p(h1), h1.shape[1], 512, _p(w2_bf), w2_bf.shape[1], 128, _p(b2), _p(w3), _p(b3), _p(w4), _p(b4),
                                     _p(target), _p(seq_len), b, t, float(grad_scale), _p(pred), _p(loss), _p(dz2), 128,
                                     _p(grads_out), 0, _p(ws), ws.numel(), _stream()), 'mg_f0_l2tail_bf16')
    return pred, loss, dz2


def f0_l2tail_rows(h1, w2_bf, b2, w3, b3, w4, b4, target_rows, row_weight, grads_out, grad_scale=1.0):
    """f0_l2tail on table rows that each stand for a group of frames (mg_f0_l2tail_rows_bf16, the phone-rate step)."""
    lib = _lib.load()
    m = h1.shape[0]
    pred = torch.empty((m,), dtype=torch.float32, device=h1.device)
    n_grads = 32 * 128 + 32 + 32 + 1
    loss = grads_out[n_grads] if grads_out.numel() > n_grads else torch.empty((), dtype=torch.float32, device=h1.device)
    dz2 = torch.empty((m, 128), dtype=torch.bfloat16, device=h1.device)
    ws = workspace(lib.mg_f0_l2tail_workspace_bytes(m), h1.device)
    _lib.check(lib.mg_f0_l2tail_rows_bf16(_p(h1), h1.shape[1], 512, _p(w2_bf), w2_bf.shape[1], 128, _p(b2), _p(w3), _p(b3), _p(w4),
                                          _p(b4), _p(target_rows), _p(row_weight), m, float(grad_scale), _p(pred), _p(loss), _p(dz2),
                                          128, _p(grads_out), 0, _p(ws), ws.numel(), _stream()), 'mg_f0_l2tail_rows_bf16')
    return pred, loss, dz2


def f0_l2tail_rows_expand(h1, w2_bf, b2, w3, b3, w4, b4, target_rows, row_weight, grads_out, rows, loss_const, grad_scale=1.0, defer=False):
    """f0_l2tail_rows + expand_column with the tail's reduce folded into the expansion's launch (mg_f0_l2tail_rows_slabs_bf16 +
    mg_expand_column_reduce_f32): one node less in the phone-rate step, same sums in the same order.  ``grads_out`` must have room for
    the loss behind the tail's gradients; ``loss_const`` = (partials, n_table_rows, extra).  Returns (pred (frames,), loss, dz2).
    ``defer``: the second launch is NOT made; a fourth return value describes it, for linear_wgrad_dgrad_bf16(tail=...) /
    finish_deferred_tail - pred, loss and the tail's gradients are valid only after one of them ran (a step captured whole into a
    HIP graph: functional.DEFER_TAIL)."""
    lib = _lib.load()
    m = h1.shape[0]
    n = 32 * 128 + 32 + 32 + 2                           # dW3 | db3 | dW4 | db4 | loss
    if grads_out.numel() < n:
        raise ValueError('f0_l2tail_rows_expand: the gradient buffer needs one float behind the gradients for the loss')
    pred_rows = torch.empty((m,), dtype=torch.float32, device=h1.device)
    dz2 = torch.empty((m, 128), dtype=torch.bfloat16, device=h1.device)
    # deferred: the slabs must outlive this call (they are summed at the end of a later launch), so they do not go to the scratch
    # buffer every kernel of the step shares
    ws = (_tail_slabs if defer else workspace)(lib.mg_f0_l2tail_workspace_bytes(m), h1.device)
    n_slabs = ctypes.c_int(0)
    _lib.check(lib.mg_f0_l2tail_rows_slabs_bf16(_p(h1), h1.shape[1], 512, _p(w2_bf), w2_bf.shape[1], 128, _p(b2), _p(w3), _p(b3), _p(w4),
                                                _p(b4), _p(target_rows), _p(row_weight), m, float(grad_scale), _p(pred_rows), _p(dz2),
                                                128, _p(ws), ws.numel(), ctypes.byref(n_slabs), _stream()),
               'mg_f0_l2tail_rows_slabs_bf16')
    partials, n_table_rows, extra = loss_const
    out = torch.empty((rows.numel(),), dtype=torch.float32, device=h1.device)
    stride = int(lib.mg_f0_l2tail_slab_stride())
    if defer:
        tail = dict(pred_rows=pred_rows, rows=rows, out=out, partials=partials, n_table_rows=int(n_table_rows), extra=int(extra), ws=ws,
                    n=n, stride=stride, n_slabs=n_slabs.value, grads_out=grads_out)
        return out, grads_out[n - 1], dz2, tail
    _lib.check(lib.mg_expand_column_reduce_f32(_p(pred_rows), _p(rows), rows.numel(), _p(out), _p(partials), n_table_rows, extra, _p(ws),
                                               n, stride, n_slabs.value, _p(grads_out), _stream()), 'mg_expand_column_reduce_f32')
    return out, grads_out[n - 1], dz2


def phone_target_stats(target, rows, seg, seq_len, b, t, n_table_rows, extra):
    """(ybar (R + extra,), weight (R + extra,), partials) of the masked MSE per table row (mg_phone_target_stats); ``partials`` holds
    the per-block sums of the loss's constant term for ``phone_loss_const_add``."""
    lib = _lib.load()
    target = _require(target, torch.float32, 'targets')
    stats = torch.empty((2, n_table_rows + extra), dtype=torch.float32, device=target.device)
    ws = torch.empty(lib.mg_phone_target_stats_workspace_bytes(n_table_rows, extra), dtype=torch.uint8, device=target.device)
    _lib.check(lib.mg_phone_target_stats(_p(target), _p(rows), rows.numel(), _p(seg[0]), _p(seg[1]), _p(seq_len), b, t, n_table_rows,
                                         extra, _p(stats[0]), _p(stats[1]), None, _p(ws), ws.numel(), _stream()),
               'mg_phone_target_stats')
    return stats[0], stats[1], ws


def phone_front_ok(b, p, t, extra):
    """Shapes the one-launch front of the phone-rate step takes (mg_phone_front): a partial-sum slot per utterance (about 16 phones per
    utterance or more) and extra rows whose frame chunks span few utterances."""
    if not (b > 0 and p > 0 and t > 0 and extra > 0 and p <= 12288 and b <= -(-(b * p) // 16)):
        return False
    chunk = -(-(b * t) // extra)
    return 512 + max(p + t + 1, -(-(4 * chunk) // t) + 2) <= 16000


def phone_front(dur, target, seq_len, t, extra, linear=None):
    """upsample_index_maps + phone_target_stats in ONE launch (mg_phone_front): returns (rows (B, t), rows_mapped (B, t), seg (2, B*P),
    ybar, weight, partials).  ``linear`` = (a bf16 (M, lda), k, w_bf16, bias, n, act): additionally y = act(a w^T + bias) (bf16 (M, n)),
    appended to the result, in the SAME grid where the GEMM leaves CUs idle (mg_phone_front_linear_fwd_bf16)."""
    lib = _lib.load()
    dur = _require(dur, torch.int64, 'dur')
    target = _require(target, torch.float32, 'targets')
    b, p = dur.shape
    if target.numel() != b * t:
        raise ValueError('phone_front: %d targets for %d x %d frames' % (target.numel(), b, t))
    r = b * p
    rows = torch.empty((2, b, t), dtype=torch.int32, device=dur.device)
    seg = torch.empty((2, r), dtype=torch.int32, device=dur.device)
    stats = torch.empty((2, r + extra), dtype=torch.float32, device=dur.device)
    ws = torch.empty(lib.mg_phone_target_stats_workspace_bytes(r, extra), dtype=torch.uint8, device=dur.device)
    front = (_p(dur), b, p, int(t), _p(target), _p(seq_len), extra, _p(rows[0]), _p(rows[1]), r, _p(seg[0]), _p(seg[1]), _p(stats[0]),
             _p(stats[1]), _p(ws), ws.numel())
    if linear is None:
        _lib.check(lib.mg_phone_front(*front, _stream()), 'mg_phone_front')
        return rows[0], rows[1], seg, stats[0], stats[1], ws
    a, k, w_bf16, bias, n, act = linear
    m = a.shape[0]
    y = torch.empty((m, pad8(n)), dtype=torch.bfloat16, device=a.device)
    _lib.check(lib.mg_phone_front_linear_fwd_bf16(*front, _p(a), a.shape[1], m, k, _p(w_bf16), w_bf16.shape[1], _p(bias), n, _p(y), pad8(n), act,
                                                  _stream()), 'mg_phone_front_linear_fwd_bf16')
    return rows[0], rows[1], seg, stats[0], stats[1], ws, y


F0_TAIL_F32_GRADS = 32 * 128 + 32 + 32 + 1          # dW3 | db3 | dW4 | db4, then the loss (and two unused floats)


def f0_tail_rows_f32(z2, w3, b3, w4, b4, ybar, weight, seq_len=None, frames=None):
    """The README tail 128 -> 32 -> 1 with the per-phone masked MSE, forward and backward, exact fp32, one launch (mg_f0_tail_rows_f32).
    z2 (rows, 128) f32 pre-activations of the 128-wide layer.  Returns (pred (rows,), dz2 (rows, 128), flat (4164,) = the tail's
    parameter gradients in parameter order, the loss without its constant term at index F0_TAIL_F32_GRADS).
    ``weight`` None with ``frames`` = (B, T): the rows are the frames, ``ybar`` the targets, the weights the masked MSE's own (from seq_len)."""
    lib = _lib.load()
    z2 = _require(z2, torch.float32, 'pre-activations')
    m = z2.shape[0]
    if (z2.shape[1] != 128 or tuple(w3.shape) != (32, 128) or tuple(w4.shape) != (1, 32) or ybar.numel() != m
            or (weight.numel() != m if weight is not None else (frames is None or frames[0] * frames[1] != m))):
        raise ValueError('f0_tail_rows_f32: needs (rows, 128) pre-activations, a 128 -> 32 -> 1 tail and one statistics row per table row')
    pred = torch.empty((m,), dtype=torch.float32, device=z2.device)
    dz2 = torch.empty((m, 128), dtype=torch.float32, device=z2.device)
    flat = torch.empty((F0_TAIL_F32_GRADS + 3,), dtype=torch.float32, device=z2.device)
    ws = workspace(lib.mg_f0_tail_rows_f32_workspace_bytes(m), z2.device)
    _lib.check(lib.mg_f0_tail_rows_f32(_p(z2), z2.shape[1], _p(_require(w3, torch.float32, 'w3')), _p(b3), _p(_require(w4, torch.float32, 'w4')),
                                       _p(b4), _p(ybar), _p(weight), _p(seq_len), frames[0] if frames else 0, frames[1] if frames else 0, m,
                                       _p(pred), _p(dz2), dz2.shape[1], _p(flat), _p(ws), ws.numel(), _stream()),
               'mg_f0_tail_rows_f32')
    return pred, dz2, flat


def phone_mse_rows(pred_table, ybar, weight):
    """(loss (1,) without the constant term, dpred (rows,)) of the masked MSE of a one-column prediction per table row
    (mg_phone_mse_rows_f32); pred_table (rows, ld) f32, column 0."""
    pred_table = _require(pred_table, torch.float32, 'prediction')
    n = ybar.numel()
    if pred_table.shape[0] != n:
        raise ValueError('phone_mse_rows: %d prediction rows for %d statistics rows' % (pred_table.shape[0], n))
    out = torch.empty((n + 1,), dtype=torch.float32, device=pred_table.device)
    _lib.check(_lib.load().mg_phone_mse_rows_f32(_p(pred_table), pred_table.shape[1], _p(ybar), _p(weight), n, _p(out[n:]), _p(out), _stream()),
               'mg_phone_mse_rows_f32')
    return out[n:], out[:n]


def phone_loss_const_add(partials, n_table_rows, extra, loss):
    """loss (0-d f32, in place) += the constant term left by phone_target_stats."""
    _lib.check(_lib.load().mg_phone_loss_const_add(_p(partials), n_table_rows, extra, _p(loss), _stream()), 'mg_phone_loss_const_add')


def expand_column(table, rows, loss_const=None):
    """out[f] = table[rows[f]] for a (rows,) f32 table and an int32 map without negative entries (mg_expand_column_f32).
    ``loss_const`` = (partials, n_table_rows, extra, loss): the same launch adds phone_target_stats' constant to ``loss`` in place."""
    lib = _lib.load()
    table = _require(table, torch.float32, 'table')
    out = torch.empty((rows.numel(),), dtype=torch.float32, device=table.device)
    if loss_const is None:
        _lib.check(lib.mg_expand_column_f32(_p(table), _p(rows), rows.numel(), _p(out), _stream()), 'mg_expand_column_f32')
    else:
        partials, n_table_rows, extra, loss = loss_const
        _lib.check(lib.mg_expand_column_loss_f32(_p(table), _p(rows), rows.numel(), _p(out), _p(partials), n_table_rows, extra,
                                                 _p(loss), _stream()), 'mg_expand_column_loss_f32')
    return out


def sigmoid(x):
    lib = _lib.load()
    x = _require(x, torch.float32, 'input')
    y = torch.empty_like(x)
    _lib.check(lib.mg_sigmoid_f32(_p(x), _p(y), x.numel(), _stream()), 'mg_sigmoid_f32')
    return y


def sigmoid_grad(dy, y):
    lib = _lib.load()
    dy = _require(dy, torch.float32, 'grad')
    dx = torch.empty_like(y)
    _lib.check(lib.mg_sigmoid_grad_f32(_p(dy), _p(y), _p(dx), y.numel(), _stream()), 'mg_sigmoid_grad_f32')
    return dx


# ----------------------------------------------------------------------------------------------------------------- K3
def gru_persist_f32_ok(b, t, h):
    return PERSISTENT_RECURRENCE and bool(_lib.load().mg_gru_persist_f32_supported(b, t, h))


def gru_fwd(xproj, w_hh, b_hh, seq_len, h0, b, t, h, persistent=None):
    """xproj (b, t, 3h) f32.  Returns (out (b,t,h), hstate (b,t+1,h), saved (b,t,4h)).  persistent: the one-launch fp32 kernel
    (None = whenever the shape is covered; bit-identical to the per-step kernels on the live steps)."""
    lib = _lib.load()
    dev = xproj.device
    if persistent is None:
        persistent = gru_persist_f32_ok(b, t, h)
    hstate = torch.empty((b, t + 1, h), dtype=torch.float32, device=dev)
    if h0 is None:
        hstate[:, 0].zero_()
    else:
        hstate[:, 0].copy_(h0.reshape(b, h))
    out = torch.empty((b, t, h), dtype=torch.float32, device=dev)
    saved = torch.empty((b, t, 4 * h), dtype=torch.float32, device=dev)
    if persistent:
        ws = _persist_workspace(dev, b, h)
        _lib.check(lib.mg_gru_fwd_persist_f32(_p(xproj), _p(w_hh), _p(b_hh), _p(seq_len), b, t, h, _p(hstate), _p(out), _p(saved),
                                              _p(ws), ws.numel(), _stream()), 'mg_gru_fwd_persist_f32')
        return out, hstate, saved
    _lib.check(lib.mg_gru_fwd_f32(_p(xproj), _p(w_hh), _p(b_hh), _p(seq_len), b, t, h, _p(hstate), _p(out), _p(saved),
                                  _stream()), 'mg_gru_fwd_f32')
    return out, hstate, saved


def gru_bwd(grad_out, grad_hn, hstate, saved, w_hh, seq_len, b, t, h, persistent=None):
    lib = _lib.load()
    dev = grad_out.device
    dxproj = torch.empty((b, t, 3 * h), dtype=torch.float32, device=dev)
    dhproj = torch.empty((b, t, 3 * h), dtype=torch.float32, device=dev)
    dh0 = torch.empty((b, h), dtype=torch.float32, device=dev)
    if persistent is None:
        persistent = gru_persist_f32_ok(b, t, h)
    if persistent:
        ws = _persist_workspace(dev, b, h)
        _lib.check(lib.mg_gru_bwd_persist_f32(_p(grad_out), _p(grad_hn), _p(hstate), _p(saved), _p(w_hh), _p(seq_len), b, t, h,
                                              _p(dxproj), _p(dhproj), _p(dh0), _p(ws), ws.numel(), _stream()), 'mg_gru_bwd_persist_f32')
        return dxproj, dhproj, dh0
    nbytes = lib.mg_gru_bwd_workspace_bytes(b, h)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)   # own buffer: lives across the wgrad calls that follow
    _lib.check(lib.mg_gru_bwd_f32(_p(grad_out), _p(grad_hn), _p(hstate), _p(saved), _p(w_hh), _p(seq_len), b, t, h,
                                  _p(dxproj), _p(dhproj), _p(dh0), _p(ws), ws.numel(), _stream()), 'mg_gru_bwd_f32')
    return dxproj, dhproj, dh0


def gru_bf16_ok(h):
    """The bf16-operand recurrence needs H % 128 == 0 and H <= 1024 (include/morgana_hip.h: mg_gru_fwd_bf16)."""
    return h % 128 == 0 and h <= 1024


_PERSIST_WORKSPACES = {}          # (device, stream) -> sync block shared by the persistent launches of that stream
PERSISTENT_RECURRENCE = os.environ.get('MORGANA_PERSISTENT', '1') != '0'


def gru_persist_ok(b, t, h):
    return PERSISTENT_RECURRENCE and bool(_lib.load().mg_gru_persist_supported(b, t, h))


def _persist_workspace(dev, b, h):
    key = (dev, torch.cuda.current_stream().cuda_stream)
    need = _lib.load().mg_gru_persist_workspace_bytes(b, h)
    ws = _PERSIST_WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            check_persistent_status()                 # read the old block's status word before dropping it
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        _PERSIST_WORKSPACES[key] = ws
    return ws


def check_persistent_status():
    """Synchronise and raise if a persistent recurrent kernel launched since the last call gave up waiting for a peer workgroup
    (its results are invalid).  Called once per epoch by ExperimentBuilder, by bench.py, smoke() and the GPU tests."""
    lib = _lib.load()
    for key, ws in list(_PERSIST_WORKSPACES.items()):
        _lib.check(lib.mg_gru_persist_status(_p(ws), ctypes.c_void_p(key[1])), 'mg_gru_persist_status')


def gru_fwd_bf16(xproj, w_hh, b_hh, seq_len, h0, b, t, h, persistent=None, xrows=None, out_bf=None, w_bf=None):
    """gru_fwd with bf16 matmul operands.  Returns (out, hstate, saved, hstate_bf (b,t+1,h) bf16).
    out_bf (persistent launch only): a (b, t, h) bf16 tensor that receives the bf16 copy of ``out`` from the recurrence itself.
    persistent: one launch for all steps (None = whenever the shape is covered).
    xrows (persistent launch only): int32 (b, t) row map - ``xproj`` is then a table (rows, 3h) and frame (b, t) takes row
    ``xrows[b, t]`` of it (mg_gru_fwd_persist_rows_bf16: the repetition applied inside the recurrence).
    w_bf: the (3h, pad_ld(h)) bf16 copy of ``w_hh`` if the caller holds a current one (param_shadows); cast here otherwise."""
    lib = _lib.load()
    if persistent is None:
        persistent = gru_persist_ok(b, t, h)
    if xrows is not None and not persistent:
        raise ValueError('gru_fwd_bf16: a row map needs the persistent launch (gather the rows first)')
    if out_bf is not None and (not persistent or out_bf.dtype != torch.bfloat16 or tuple(out_bf.shape) != (b, t, h) or not out_bf.is_contiguous()):
        raise ValueError('gru_fwd_bf16: out_bf must be a contiguous (b, t, h) bfloat16 tensor and needs the persistent launch')
    dev = xproj.device
    hstate = torch.empty((b, t + 1, h), dtype=torch.float32, device=dev)
    hstate_bf = torch.empty((b, t + 1, h), dtype=torch.bfloat16, device=dev)
    if h0 is None:
        hstate[:, 0].zero_()
        hstate_bf[:, 0].zero_()
    else:
        hstate[:, 0].copy_(h0.reshape(b, h))
        hstate_bf[:, 0].copy_(h0.reshape(b, h))
    if w_bf is None or w_bf.dtype != torch.bfloat16 or w_bf.shape[0] != 3 * h or w_bf.shape[1] < h or not w_bf.is_contiguous():
        w_bf = cast_pad_bf16(w_hh)
    out = torch.empty((b, t, h), dtype=torch.float32, device=dev)
    saved = torch.empty((b, t, 4 * h), dtype=torch.float32, device=dev)
    if persistent:
        ws = _persist_workspace(dev, b, h)
        if xrows is not None:
            xrows = _require(xrows, torch.int32, 'xrows')
            if xrows.numel() != b * t or xproj.ndim != 2 or xproj.shape[1] != 3 * h:
                raise ValueError('gru_fwd_bf16: row map of %d entries / table %s for b=%d t=%d h=%d' % (xrows.numel(), tuple(xproj.shape), b, t, h))
            _lib.check(lib.mg_gru_fwd_persist_out_bf16(_p(xproj), _p(xrows), xproj.shape[0], _p(w_bf), w_bf.shape[1], _p(b_hh), _p(seq_len),
                                                       b, t, h, _p(hstate), _p(hstate_bf), _p(out), _p(out_bf), _p(saved), _p(ws), ws.numel(),
                                                       _stream()), 'mg_gru_fwd_persist_out_bf16')
            return out, hstate, saved, hstate_bf
        _lib.check(lib.mg_gru_fwd_persist_out_bf16(_p(xproj), None, 0, _p(w_bf), w_bf.shape[1], _p(b_hh), _p(seq_len), b, t, h, _p(hstate),
                                                   _p(hstate_bf), _p(out), _p(out_bf), _p(saved), _p(ws), ws.numel(), _stream()),
                   'mg_gru_fwd_persist_out_bf16')
    else:
        _lib.check(lib.mg_gru_fwd_bf16(_p(xproj), _p(w_bf), w_bf.shape[1], _p(b_hh), _p(seq_len), b, t, h, _p(hstate), _p(hstate_bf),
                                       _p(out), _p(saved), _stream()), 'mg_gru_fwd_bf16')
    return out, hstate, saved, hstate_bf


def gru_bwd_bf16(grad_out, grad_hn, hstate, saved, w_hh, seq_len, b, t, h, persistent=None, shadows_only=False, wt_bf=None):
    """gru_bwd with bf16 matmul operands.  Returns (dxproj, dhproj, dh0, dhproj_bf (b,t,3h) bf16).
    shadows_only (persistent kernel only): returns (dxproj_bf, dhproj_bf, dh0) and does not write the fp32 arrays.
    wt_bf: the (h, pad_ld(3h)) bf16 transpose of ``w_hh`` if the caller holds a current one (param_shadows); cast here otherwise."""
    lib = _lib.load()
    if persistent is None:
        persistent = gru_persist_ok(b, t, h)
    dev = grad_out.device
    if wt_bf is not None and (wt_bf.dtype != torch.bfloat16 or wt_bf.shape[0] != h or wt_bf.shape[1] < 3 * h or not wt_bf.is_contiguous()):
        wt_bf = None
    if shadows_only and persistent:
        dxproj_bf = torch.empty((b, t, 3 * h), dtype=torch.bfloat16, device=dev)
        dhproj_bf = torch.empty((b, t, 3 * h), dtype=torch.bfloat16, device=dev)
        dh0 = torch.empty((b, h), dtype=torch.float32, device=dev)
        if wt_bf is None:
            wt_bf = cast_transpose_bf16(w_hh)
        ws = _persist_workspace(dev, b, h)
        _lib.check(lib.mg_gru_bwd_persist_bf16(_p(grad_out), _p(grad_hn), _p(hstate), _p(saved), _p(wt_bf), wt_bf.shape[1], _p(seq_len),
                                               b, t, h, None, None, _p(dhproj_bf), _p(dxproj_bf), _p(dh0), _p(ws), ws.numel(),
                                               _stream()), 'mg_gru_bwd_persist_bf16')
        return dxproj_bf, dhproj_bf, dh0
    dxproj = torch.empty((b, t, 3 * h), dtype=torch.float32, device=dev)
    dhproj = torch.empty((b, t, 3 * h), dtype=torch.float32, device=dev)
    dhproj_bf = torch.empty((b, t, 3 * h), dtype=torch.bfloat16, device=dev)
    dh0 = torch.empty((b, h), dtype=torch.float32, device=dev)
    if wt_bf is None:
        wt_bf = cast_transpose_bf16(w_hh)                          # (h, 3h)
    if persistent:
        ws = _persist_workspace(dev, b, h)
        _lib.check(lib.mg_gru_bwd_persist_bf16(_p(grad_out), _p(grad_hn), _p(hstate), _p(saved), _p(wt_bf), wt_bf.shape[1], _p(seq_len),
                                               b, t, h, _p(dxproj), _p(dhproj), _p(dhproj_bf), None, _p(dh0), _p(ws), ws.numel(),
                                               _stream()), 'mg_gru_bwd_persist_bf16')
        return dxproj, dhproj, dh0, dhproj_bf
    nbytes = lib.mg_gru_bwd_workspace_bytes(b, h)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.mg_gru_bwd_bf16(_p(grad_out), _p(grad_hn), _p(hstate), _p(saved), _p(wt_bf), wt_bf.shape[1], _p(seq_len), b, t, h,
                                   _p(dxproj), _p(dhproj), _p(dhproj_bf), _p(dh0), _p(ws), ws.numel(), _stream()),
               'mg_gru_bwd_bf16')
    return dxproj, dhproj, dh0, dhproj_bf


def lstm_persist_f32_ok(b, t, h):
    """The one-launch fp32 LSTM recurrence covers this shape (include/morgana_hip.h: mg_lstm_fwd_persist_f32)."""
    return PERSISTENT_RECURRENCE and bool(_lib.load().mg_lstm_persist_f32_supported(b, t, h))


def lstm_fwd(xproj, w_hh, b_hh, seq_len, h0, c0, b, t, h, persistent=None):
    """xproj (b, t, 4h) f32.  Returns (out (b,t,h), hstate (b,t+1,h), cstate (b,t+1,h), saved (b,t,4h)).  persistent: the one-launch
    fp32 kernel (None = whenever the shape is covered; bit-identical to the per-step kernels on the live steps)."""
    lib = _lib.load()
    dev = xproj.device
    if persistent is None:
        persistent = lstm_persist_f32_ok(b, t, h)
    hstate = torch.empty((b, t + 1, h), dtype=torch.float32, device=dev)
    cstate = torch.empty((b, t + 1, h), dtype=torch.float32, device=dev)
    for state, init in ((hstate, h0), (cstate, c0)):
        if init is None:
            state[:, 0].zero_()
        else:
            state[:, 0].copy_(init.reshape(b, h))
    out = torch.empty((b, t, h), dtype=torch.float32, device=dev)
    saved = torch.empty((b, t, 4 * h), dtype=torch.float32, device=dev)
    if persistent:
        ws = _persist_workspace(dev, b, h)
        _lib.check(lib.mg_lstm_fwd_persist_f32(_p(xproj), _p(w_hh), _p(b_hh), _p(seq_len), b, t, h, _p(hstate), _p(cstate), _p(out),
                                               _p(saved), _p(ws), ws.numel(), _stream()), 'mg_lstm_fwd_persist_f32')
        return out, hstate, cstate, saved
    _lib.check(lib.mg_lstm_fwd_f32(_p(xproj), _p(w_hh), _p(b_hh), _p(seq_len), b, t, h, _p(hstate), _p(cstate), _p(out),
                                   _p(saved), _stream()), 'mg_lstm_fwd_f32')
    return out, hstate, cstate, saved


def lstm_bwd(grad_out, grad_hn, grad_cn, cstate, saved, w_hh, seq_len, b, t, h, persistent=None):
    lib = _lib.load()
    dev = grad_out.device
    dgates = torch.empty((b, t, 4 * h), dtype=torch.float32, device=dev)
    dh0 = torch.empty((b, h), dtype=torch.float32, device=dev)
    dc0 = torch.empty((b, h), dtype=torch.float32, device=dev)
    if persistent is None:
        persistent = lstm_persist_f32_ok(b, t, h)
    if persistent:
        ws = _persist_workspace(dev, b, h)
        _lib.check(lib.mg_lstm_bwd_persist_f32(_p(grad_out), _p(grad_hn), _p(grad_cn), _p(cstate), _p(saved), _p(w_hh), _p(seq_len), b, t, h,
                                               _p(dgates), _p(dh0), _p(dc0), _p(ws), ws.numel(), _stream()), 'mg_lstm_bwd_persist_f32')
        return dgates, dh0, dc0
    nbytes = lib.mg_lstm_bwd_workspace_bytes(b, h)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.mg_lstm_bwd_f32(_p(grad_out), _p(grad_hn), _p(grad_cn), _p(cstate), _p(saved), _p(w_hh), _p(seq_len), b,
                                   t, h, _p(dgates), _p(dh0), _p(dc0), _p(ws), ws.numel(), _stream()), 'mg_lstm_bwd_f32')
    return dgates, dh0, dc0


def lstm_persist_ok(b, t, h):
    """The one-launch bf16-operand LSTM recurrence covers this shape (include/morgana_hip.h: mg_lstm_fwd_persist_bf16)."""
    return PERSISTENT_RECURRENCE and bool(_lib.load().mg_lstm_persist_supported(b, t, h))


def lstm_fwd_bf16(xproj, w_hh, b_hh, seq_len, h0, c0, b, t, h):
    """lstm_fwd with bf16 matmul operands, one persistent launch.  Returns (out, hstate, cstate, saved, hstate_bf)."""
    lib = _lib.load()
    dev = xproj.device
    hstate = torch.empty((b, t + 1, h), dtype=torch.float32, device=dev)
    cstate = torch.empty((b, t + 1, h), dtype=torch.float32, device=dev)
    hstate_bf = torch.empty((b, t + 1, h), dtype=torch.bfloat16, device=dev)
    for state, init in ((hstate, h0), (cstate, c0), (hstate_bf, h0)):
        if init is None:
            state[:, 0].zero_()
        else:
            state[:, 0].copy_(init.reshape(b, h))
    w_bf = cast_pad_bf16(w_hh)
    out = torch.empty((b, t, h), dtype=torch.float32, device=dev)
    saved = torch.empty((b, t, 4 * h), dtype=torch.float32, device=dev)
    ws = _persist_workspace(dev, b, h)
    _lib.check(lib.mg_lstm_fwd_persist_bf16(_p(xproj), _p(w_bf), w_bf.shape[1], _p(b_hh), _p(seq_len), b, t, h, _p(hstate), _p(cstate),
                                            _p(hstate_bf), _p(out), _p(saved), _p(ws), ws.numel(), _stream()),
               'mg_lstm_fwd_persist_bf16')
    return out, hstate, cstate, saved, hstate_bf


def lstm_bwd_bf16(grad_out, grad_hn, grad_cn, cstate, saved, w_hh, seq_len, b, t, h, want_f32=True):
    """lstm_bwd with bf16 matmul operands, one persistent launch.  Returns (dgates, dh0, dc0, dgates_bf); want_f32=False: the
    fp32 gate gradients are not written (dgates = None) - bf16 mode's GEMMs take the shadow."""
    lib = _lib.load()
    dev = grad_out.device
    dgates = torch.empty((b, t, 4 * h), dtype=torch.float32, device=dev) if want_f32 else None
    dgates_bf = torch.empty((b, t, 4 * h), dtype=torch.bfloat16, device=dev)
    dh0 = torch.empty((b, h), dtype=torch.float32, device=dev)
    dc0 = torch.empty((b, h), dtype=torch.float32, device=dev)
    wt_bf = cast_transpose_bf16(w_hh)                              # (h, 4h)
    ws = _persist_workspace(dev, b, h)
    _lib.check(lib.mg_lstm_bwd_persist_bf16(_p(grad_out), _p(grad_hn), _p(grad_cn), _p(cstate), _p(saved), _p(wt_bf), wt_bf.shape[1],
                                            _p(seq_len), b, t, h, _p(dgates), _p(dgates_bf), _p(dh0), _p(dc0), _p(ws), ws.numel(),
                                            _stream()), 'mg_lstm_bwd_persist_bf16')
    return dgates, dh0, dc0, dgates_bf


def lstm_pstack_ok(b, t, h, n_layers):
    """The whole-stack forward wavefront (mg_lstm_pstack_fwd_bf16) covers this shape."""
    return PERSISTENT_RECURRENCE and bool(_lib.load().mg_lstm_pstack_supported(b, t, h, n_layers))


def lstm_pstack_fwd(xproj0, w_ih, w_hh, b_ih, b_hh, seq_len, h0s, c0s, b, t, h):
    """L stacked LSTM layers forward in one persistent launch.  xproj0 (b,t,4h) f32 = layer 0's input projection incl. b_ih;
    w_ih[l] (l >= 1), w_hh[l] f32 (cast to bf16 here).  Returns (out of the top layer, per-layer lists hstate, cstate, saved,
    hstate_bf); of hstate[l] only the rows 0 (h0) and T (h_n) are valid."""
    lib = _lib.load()
    dev = xproj0.device
    n_layers = len(w_hh)
    hstate, cstate, saved, hstate_bf, keep, o = [], [], [], [], [], None
    descs = (_lib.LstmPStackLayer * n_layers)()
    # the state arrays of all layers in three allocations: their initial rows are set by three launches, not three per layer
    if os.environ.get('MORGANA_LSTM_BATCH_STATES', '1') != '0':
        hs_all = torch.empty((n_layers, b, t + 1, h), dtype=torch.float32, device=dev)
        cs_all = torch.empty((n_layers, b, t + 1, h), dtype=torch.float32, device=dev)
        hb_all = torch.empty((n_layers, b, t + 1, h), dtype=torch.bfloat16, device=dev)
        for state, init in ((hs_all, h0s), (cs_all, c0s), (hb_all, h0s)):
            if init is None:
                state[:, :, 0].zero_()
            else:
                state[:, :, 0].copy_(init.reshape(n_layers, b, h))
    else:
        hs_all = [torch.empty((b, t + 1, h), dtype=torch.float32, device=dev) for _ in range(n_layers)]
        cs_all = [torch.empty((b, t + 1, h), dtype=torch.float32, device=dev) for _ in range(n_layers)]
        hb_all = [torch.empty((b, t + 1, h), dtype=torch.bfloat16, device=dev) for _ in range(n_layers)]
        for l in range(n_layers):
            for state, init in ((hs_all[l], h0s), (cs_all[l], c0s), (hb_all[l], h0s)):
                if init is None:
                    state[:, 0].zero_()
                else:
                    state[:, 0].copy_(init[l].reshape(b, h))
    # bf16 operands of W_hh (every layer) and W_ih (layers 1..): the parameters' shadows, stale ones re-cast by one batched launch
    wh_bf = weight_operands(w_hh)
    wi_bf = [None] + weight_operands(w_ih[1:])
    for l in range(n_layers):
        hs, cs, hb = hs_all[l], cs_all[l], hb_all[l]
        if l == n_layers - 1:
            o = torch.empty((b, t, h), dtype=torch.float32, device=dev)
        sv = torch.empty((b, t, 4 * h), dtype=torch.float32, device=dev)
        whb = wh_bf[l]
        d = descs[l]
        d.w_hh_bf, d.ldwh, d.b_hh = whb.data_ptr(), whb.shape[1], b_hh[l].data_ptr()
        if l == 0:
            d.xproj = xproj0.data_ptr()
        else:
            wib = wi_bf[l]
            d.w_ih_bf, d.ldwi, d.b_ih = wib.data_ptr(), wib.shape[1], b_ih[l].data_ptr()
            keep.append(wib)
        d.hstate, d.cstate, d.hstate_bf, d.saved = hs.data_ptr(), cs.data_ptr(), hb.data_ptr(), sv.data_ptr()
        d.out = o.data_ptr() if o is not None else sv.data_ptr()         # never written below the top layer
        keep.append(whb)
        hstate.append(hs); cstate.append(cs); saved.append(sv); hstate_bf.append(hb)
    key = (dev, torch.cuda.current_stream().cuda_stream, 'pstack')
    need = lib.mg_lstm_pstack_workspace_bytes(b, h, n_layers)
    ws = _PERSIST_WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            check_persistent_status()
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        _PERSIST_WORKSPACES[key] = ws
    _lib.check(lib.mg_lstm_pstack_fwd_bf16(ctypes.cast(descs, ctypes.c_void_p), n_layers, _p(seq_len), b, t, h, _p(ws), ws.numel(),
                                           _stream()), 'mg_lstm_pstack_fwd_bf16')
    return o, hstate, cstate, saved, hstate_bf


def gru_stack_small_ok(b, t, h, n_layers):
    """The small-GRU stack wavefront (mg_gru_stack_fwd_small_f32 / _bwd_small_f32) covers this shape."""
    return PERSISTENT_RECURRENCE and bool(_lib.load().mg_gru_stack_small_supported(b, t, h, n_layers))


def _gru_stack_workspace(dev):
    lib = _lib.load()
    key = (dev, torch.cuda.current_stream().cuda_stream, 'gru_stack')
    ws = _PERSIST_WORKSPACES.get(key)
    if ws is None:
        ws = torch.zeros(lib.mg_gru_stack_small_workspace_bytes(), dtype=torch.uint8, device=dev)
        _PERSIST_WORKSPACES[key] = ws
    return ws


def gru_stack_small_fwd(xproj0, w_ih, w_hh, b_ih, b_hh, seq_len, h0s, b, t, h, fast=False):
    """L stacked small GRU layers forward in one launch.  xproj0 (b,t,3h) f32 = layer 0's input projection incl. b_ih; w_ih[l], b_ih[l]
    for l >= 1 (input size == h); h0s None or (L,b,h).  Returns per-layer lists (out, hstate, saved).
    fast: the cell's sigmoid / tanh on the hardware exp / reciprocal (mg_gru_stack_fwd_small_fast_f32: throughput mode)."""
    lib = _lib.load()
    dev = xproj0.device
    n_layers = len(w_hh)
    descs = (_lib.GruStackLayer * n_layers)()
    outs, hstates, saveds = [], [], []
    for l in range(n_layers):
        hs = torch.empty((b, t + 1, h), dtype=torch.float32, device=dev)
        if h0s is None:
            hs[:, 0].zero_()
        else:
            hs[:, 0].copy_(h0s[l].reshape(b, h))
        o = torch.empty((b, t, h), dtype=torch.float32, device=dev)
        sv = torch.empty((b, t, 4 * h), dtype=torch.float32, device=dev)
        d = descs[l]
        d.w_hh, d.b_hh = w_hh[l].data_ptr(), b_hh[l].data_ptr()
        if l == 0:
            d.xproj = xproj0.data_ptr()
        else:
            d.w_ih, d.b_ih = w_ih[l].data_ptr(), b_ih[l].data_ptr()
        d.hstate, d.out, d.saved = hs.data_ptr(), o.data_ptr(), sv.data_ptr()
        outs.append(o); hstates.append(hs); saveds.append(sv)
    ws = _gru_stack_workspace(dev)
    entry = lib.mg_gru_stack_fwd_small_fast_f32 if fast else lib.mg_gru_stack_fwd_small_f32
    _lib.check(entry(ctypes.cast(descs, ctypes.c_void_p), n_layers, _p(seq_len), b, t, h, _p(ws), ws.numel(), _stream()),
               'mg_gru_stack_fwd_small_fast_f32' if fast else 'mg_gru_stack_fwd_small_f32')
    return outs, hstates, saveds


def gru_stack_small_bwd(grad_out, grad_hn, hstate, saved, w_ih, w_hh, seq_len, b, t, h, fast=False):
    """The backward of gru_stack_small_fwd in one launch.  grad_out (b,t,h) = gradient of the TOP layer's outputs; grad_hn None or a
    per-layer list of (b,h) tensors / None.  Returns per-layer lists (dxproj, dhproj) and dh0 (L,b,h)."""
    lib = _lib.load()
    dev = grad_out.device
    n_layers = len(w_hh)
    descs = (_lib.GruStackLayer * n_layers)()
    dxprojs, dhprojs, keep = [], [], []
    dh0 = torch.empty((n_layers, b, h), dtype=torch.float32, device=dev)
    for l in range(n_layers):
        d = descs[l]
        d.w_hh = w_hh[l].data_ptr()
        if l > 0:
            d.w_ih = w_ih[l].data_ptr()
        d.hstate, d.saved = hstate[l].data_ptr(), saved[l].data_ptr()
        if l + 1 == n_layers:
            d.grad_out = grad_out.data_ptr()
        else:
            dxin = torch.empty((b, t, h), dtype=torch.float32, device=dev)
            keep.append(dxin)
            d.dxin = dxin.data_ptr()
        if grad_hn is not None and grad_hn[l] is not None:
            g = _require(grad_hn[l].reshape(b, h), torch.float32, 'grad_hn')
            keep.append(g)
            d.grad_hn = g.data_ptr()
        dxp = torch.empty((b, t, 3 * h), dtype=torch.float32, device=dev)
        dhp = torch.empty((b, t, 3 * h), dtype=torch.float32, device=dev)
        d.dxproj, d.dhproj, d.dh0 = dxp.data_ptr(), dhp.data_ptr(), dh0[l].data_ptr()
        dxprojs.append(dxp); dhprojs.append(dhp)
    ws = _gru_stack_workspace(dev)
    entry = lib.mg_gru_stack_bwd_small_fast_f32 if fast else lib.mg_gru_stack_bwd_small_f32      # fast: bf16-operand products (throughput mode)
    _lib.check(entry(ctypes.cast(descs, ctypes.c_void_p), n_layers, _p(seq_len), b, t, h, _p(ws), ws.numel(), _stream()),
               'mg_gru_stack_bwd_small_fast_f32' if fast else 'mg_gru_stack_bwd_small_f32')
    return dxprojs, dhprojs, dh0


def lstm_pstack_bwd_ok(b, t, h, n_layers):
    """The whole-stack backward wavefront (mg_lstm_pstack_bwd_bf16) covers this shape."""
    return PERSISTENT_RECURRENCE and bool(_lib.load().mg_lstm_pstack_bwd_supported(b, t, h, n_layers))


def lstm_pstack_bwd(grad_out, grad_hn, grad_cn, cstate, saved, w_ih, w_hh, seq_len, b, t, h, want_f32=False):
    """L stacked LSTM layers backward in one persistent launch.  grad_out (b,t,h) f32 = gradient of the TOP layer's outputs; grad_hn /
    grad_cn None or per-layer lists of (b,h) tensors (None entries allowed); cstate[l], saved[l] from lstm_pstack_fwd; w_ih[l] (l >= 1)
    and w_hh[l] f32.  Returns per-layer lists (dgates or None, dgates_bf) and dh0, dc0 as (L,b,h) tensors."""
    lib = _lib.load()
    dev = grad_out.device
    n_layers = len(w_hh)
    descs = (_lib.LstmPStackBwdLayer * n_layers)()
    dgates, dgates_bf, keep = [], [], []
    dh0 = torch.empty((n_layers, b, h), dtype=torch.float32, device=dev)
    dc0 = torch.empty((n_layers, b, h), dtype=torch.float32, device=dev)
    wh_t = weight_operands(w_hh, transposed=True)                        # (h, 4h) each: the parameters' shadows
    wi_t = [None] + weight_operands(w_ih[1:], transposed=True)
    for l in range(n_layers):
        d = descs[l]
        wt = wh_t[l]
        keep.append(wt)
        d.w_hh_t_bf, d.ldt = wt.data_ptr(), wt.shape[1]
        if l + 1 < n_layers:
            wu = wi_t[l + 1]                                             # (h, 4h): the layer above reads this layer's outputs
            keep.append(wu)
            d.w_ih_up_t_bf, d.ldt_up = wu.data_ptr(), wu.shape[1]
        else:
            d.grad_out = grad_out.data_ptr()
        for name, src in (('grad_hn', grad_hn), ('grad_cn', grad_cn)):
            if src is not None and src[l] is not None:
                g = _require(src[l].reshape(b, h), torch.float32, name)
                keep.append(g)
                setattr(d, name, g.data_ptr())
        dg = torch.empty((b, t, 4 * h), dtype=torch.float32, device=dev) if want_f32 else None
        dgb = torch.empty((b, t, 4 * h), dtype=torch.bfloat16, device=dev)
        d.cstate, d.saved = cstate[l].data_ptr(), saved[l].data_ptr()
        d.dgates, d.dgates_bf = (dg.data_ptr() if dg is not None else None), dgb.data_ptr()
        d.dh0, d.dc0 = dh0[l].data_ptr(), dc0[l].data_ptr()
        dgates.append(dg)
        dgates_bf.append(dgb)
    key = (dev, torch.cuda.current_stream().cuda_stream, 'pstack_bwd')
    need = lib.mg_lstm_pstack_bwd_workspace_bytes(b, h, n_layers)
    ws = _PERSIST_WORKSPACES.get(key)
    if ws is None or ws.numel() < need:
        if ws is not None:
            check_persistent_status()
        ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        _PERSIST_WORKSPACES[key] = ws
    _lib.check(lib.mg_lstm_pstack_bwd_bf16(ctypes.cast(descs, ctypes.c_void_p), n_layers, _p(seq_len), b, t, h, _p(ws), ws.numel(),
                                           _stream()), 'mg_lstm_pstack_bwd_bf16')
    return dgates, dgates_bf, dh0, dc0


def lstm_stack_fwd(descs, n_layers, seq_len, b, t, h, lag, s_begin, s_end):
    lib = _lib.load()
    _lib.check(lib.mg_lstm_stack_fwd_f32(ctypes.cast(descs, ctypes.c_void_p), n_layers, _p(seq_len), b, t, h, lag, s_begin, s_end,
                                         _stream()), 'mg_lstm_stack_fwd_f32')


def lstm_stack_bwd(descs, n_layers, seq_len, b, t, h, lag, u_begin, u_end):
    lib = _lib.load()
    _lib.check(lib.mg_lstm_stack_bwd_f32(ctypes.cast(descs, ctypes.c_void_p), n_layers, _p(seq_len), b, t, h, lag, u_begin, u_end,
                                         _stream()), 'mg_lstm_stack_bwd_f32')


# ------------------------------------------------------------------------------------------------------- optimiser
def adam_step(param, grad, exp_avg, exp_avg_sq, lr, betas, eps, weight_decay, step, grad_scale=1.0):
    lib = _lib.load()
    for name, t in (('param', param), ('grad', grad), ('exp_avg', exp_avg), ('exp_avg_sq', exp_avg_sq)):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError('adam_step: %s must be a contiguous float32 device tensor' % name)
    _lib.check(lib.mg_adam_step_f32(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(lr),
                                    float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                                    float(grad_scale), _stream()), 'mg_adam_step_f32')


def adam_scalars(lr, betas, step):
    """(step_size, bc2_sqrt) of one Adam step as two float32, formed as mg_adam_step_f32 forms them."""
    out = (ctypes.c_float * 2)()
    _lib.load().mg_adam_scalars(float(lr), float(betas[0]), float(betas[1]), int(step), out)
    return out[0], out[1]


def store_pair(dst, a, b):
    """dst[0:2] = (a, b) in stream order; the values are kernel arguments, not a host buffer read later."""
    _lib.check(_lib.load().mg_store_pair_f32(_p(dst), float(a), float(b), _stream()), 'mg_store_pair_f32')


STORE_PAIRS_MAX = 32


def store_pairs(dst, pairs):
    """dst[2 j : 2 j + 2] = pairs[j] in stream order (mg_store_pairs_f32): the Adam scalars of the steps of one multi-step replay."""
    flat = (ctypes.c_float * (2 * len(pairs)))(*[float(x) for pair in pairs for x in pair])
    _lib.check(_lib.load().mg_store_pairs_f32(_p(dst), ctypes.cast(flat, ctypes.c_void_p), len(pairs), _stream()), 'mg_store_pairs_f32')


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, betas, eps, weight_decay, scalars, grad_scale=1.0):
    """adam_step with (step_size, bc2_sqrt) read from the 2-float device tensor ``scalars`` (capturable in a HIP graph)."""
    lib = _lib.load()
    _lib.check(lib.mg_adam_step_dev_f32(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(betas[0]),
                                        float(betas[1]), float(eps), float(weight_decay), _p(scalars), float(grad_scale), _stream()),
               'mg_adam_step_dev_f32')


def adam_step_plan(param, grad, exp_avg, exp_avg_sq, betas, eps, weight_decay, scalars, grad_scale=1.0, slab_srcs=(), shadows=(),
                   clear_grad=False, tail=None):
    """The update as the last node of a step (mg_adam_step_plan_f32): ``slab_srcs`` = [(begin, count, slab tensor, n_slabs, stride)]
    are summed into the gradient on the fly (split-M partial results of the weight-gradient GEMMs, in the slab reduce's order),
    ``shadows`` = [(offset, rows, cols, bf16 [rows, ldd] or None, bf16 transpose [cols, ldt] or None)] are refreshed from the
    updated weights, ``clear_grad`` zeroes the flat gradient behind the read."""
    lib = _lib.load()
    if len(slab_srcs) > _lib.ADAM_MAX_SLABS or len(shadows) > _lib.ADAM_MAX_SHADOWS:
        raise ValueError('adam_step_plan: at most %d slab sources and %d shadows' % (_lib.ADAM_MAX_SLABS, _lib.ADAM_MAX_SHADOWS))
    plan = _lib.AdamPlan()
    plan.n_slab_srcs, plan.n_shadows, plan.clear_grad = len(slab_srcs), len(shadows), int(bool(clear_grad))
    for i, (begin, count, slab, n_slabs, stride) in enumerate(slab_srcs):
        src = plan.slabs[i]
        src.begin, src.count, src.slab, src.n_slabs, src.stride = int(begin), int(count), slab.data_ptr(), int(n_slabs), int(stride)
    for i, entry in enumerate(shadows):
        offset, rows, cols, dst, dst_t = entry[:5]
        sh = plan.shadows[i]
        sh.pair = int(bool(entry[5])) if len(entry) > 5 else 0        # [hi | lo] pair planes (pair_shadows)
        sh.offset, sh.rows, sh.cols = int(offset), int(rows), int(cols)
        sh.dst, sh.ldd = (dst.data_ptr(), dst.shape[1]) if dst is not None else (None, 0)
        sh.dst_t, sh.ldt = (dst_t.data_ptr(), dst_t.shape[1]) if dst_t is not None else (None, 0)
    if tail is not None:
        # the forward's deferred tail (f0_l2tail_rows_expand(defer=True) / f0_l2tail(defer=True)): the launch's first blocks repeat the
        # prediction and form the loss
        t = plan.tail
        if tail.get('rows') is not None:
            t.table, t.rows, t.frames, t.out = tail['pred_rows'].data_ptr(), tail['rows'].data_ptr(), tail['rows'].numel(), tail['out'].data_ptr()
        if tail.get('partials') is not None:
            t.partial = tail['partials'].data_ptr()
            t.n_partial = (int(tail['n_table_rows']) + 15) // 16 + (int(tail['extra']) + 3) // 4
        t.slab, t.n, t.stride, t.n_slabs, t.dst = tail['ws'].data_ptr(), int(tail['n']), int(tail['stride']), int(tail['n_slabs']), \
            tail['grads_out'].data_ptr()
    _lib.check(lib.mg_adam_step_plan_f32(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(betas[0]),
                                         float(betas[1]), float(eps), float(weight_decay), _p(scalars), float(grad_scale),
                                         ctypes.byref(plan), _stream()), 'mg_adam_step_plan_f32')


def ema_update(shadow, param, decay):
    lib = _lib.load()
    if not (shadow.is_cuda and shadow.is_contiguous() and param.is_contiguous()):
        raise ValueError('ema_update: tensors must be contiguous device tensors')
    _lib.check(lib.mg_ema_update_f32(_p(shadow), _p(param), shadow.numel(), float(decay), _stream()),
               'mg_ema_update_f32')


# ----------------------------------------------------------------------------------------------------------- metrics
METRIC_MEAN, METRIC_SQDIFF, METRIC_ABSDIFF, METRIC_ROOT_SQ, METRIC_SQDIFF_VOICED, METRIC_SQDIFF_VOICED_EXP = range(6)


def metric_accumulate(kind, accum, target, pred=None, voiced=None, seq_len=None, col0=0, width=None):
    """accum (2,) float64 on the device: accum += (sum, count) of one batch (csrc/metrics.hip).  target / pred (B, T, D) f32."""
    lib = _lib.load()
    target = _require(target, torch.float32, 'target')
    if target.dim() != 3:
        raise ValueError('metric inputs must have shape (batch, time, features), got %s' % (tuple(target.shape),))
    b, t, d = target.shape
    width = d - col0 if width is None else width
    if pred is not None:
        pred = _require(pred, torch.float32, 'pred')
        if pred.shape != target.shape:
            raise ValueError('target %s and prediction %s differ in shape' % (tuple(target.shape), tuple(pred.shape)))
    if voiced is not None:
        voiced = voiced.reshape(b, t).to(torch.float32).contiguous()
    ws = torch.empty(lib.mg_metric_workspace_bytes(), dtype=torch.uint8, device=target.device)
    _lib.check(lib.mg_metric_accumulate_f32(kind, _p(target), _p(pred), _p(voiced), _p(seq_len), b, t, d, col0, width, _p(accum), _p(ws),
                                            ws.numel(), _stream()), 'mg_metric_accumulate_f32')


MLPG_MAX_WINDOWS, MLPG_MAX_COEFF = 4, 5


def mlpg(means, variances, windows, padding_size=0, seq_len=None, out_dtype=torch.float32):
    """Most probable trajectories of (B, T, W*D) f32 delta-stream means (csrc/mlpg.hip, morgana/viz/synthesis.py:79-178).
    variances (W*D,) global or (B, T, W*D) per frame, f32; windows [(l, u, coeffs)]; returns (B, T, D), zero past seq_len."""
    lib = _lib.load()
    means = _require(means, torch.float32, 'means')
    variances = _require(variances, torch.float32, 'variances')
    if means.dim() != 3:
        raise ValueError('means must have shape (batch, time, windows * features), got %s' % (tuple(means.shape),))
    b, t, width = means.shape
    n_win = len(windows)
    if n_win == 0 or n_win > MLPG_MAX_WINDOWS or width % n_win != 0:
        raise ValueError('%d windows for %d stream columns (1..%d windows, columns a multiple of it)' % (n_win, width, MLPG_MAX_WINDOWS))
    d = width // n_win
    if variances.dim() == 1 and variances.shape[0] == width:
        per_frame = 0
    elif variances.shape == means.shape:
        per_frame = 1
    else:
        raise ValueError('variances %s fit neither (%d,) nor %s' % (tuple(variances.shape), width, tuple(means.shape)))
    win_l = (ctypes.c_int * n_win)()
    win_u = (ctypes.c_int * n_win)()
    win_c = (ctypes.c_double * (n_win * MLPG_MAX_COEFF))()
    for w, (l, u, coeff) in enumerate(windows):
        l, u = int(l), int(u)
        if l < 0 or u < 0 or len(coeff) != l + u + 1:                 # the asserts of synthesis.py:30-31
            raise ValueError('window %d: %d coefficients for extents l=%d, u=%d' % (w, len(coeff), l, u))
        if l + u + 1 > MLPG_MAX_COEFF:
            raise ValueError('window %d is wider than %d coefficients' % (w, MLPG_MAX_COEFF))
        win_l[w], win_u[w] = l, u
        for k, c in enumerate(coeff):
            win_c[w * MLPG_MAX_COEFF + k] = float(c)
    if seq_len is not None:
        seq_len = _require(seq_len, torch.int64, 'seq_len')
    if out_dtype not in (torch.float32, torch.float64):
        raise TypeError('out_dtype must be torch.float32 or torch.float64')
    out = torch.empty((b, t, d), dtype=out_dtype, device=means.device)
    nbytes = lib.mg_mlpg_workspace_bytes(b, t, d, int(padding_size), n_win, win_l, win_u)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=means.device)
    _lib.check(lib.mg_mlpg_f32(_p(means), _p(variances), per_frame, _p(seq_len), b, t, d, n_win, win_l, win_u, win_c, int(padding_size),
                               _p(out), int(out_dtype == torch.float64), _p(ws), ws.numel(), _stream()), 'mg_mlpg_f32')
    return out


# ---------------------------------------------------------------------------------------------- phone-rate first layer
# MORGANA_PHONE_RATE=0: every product at frame rate (the reference's order of operations).  This is only the DEFAULT of a per-call
# choice: utils.upsample_to_repetitions(..., phone_rate=) records the choice on the lazy sequence it returns, the layers that consume
# the sequence read it there (models pass their own ``phone_rate`` attribute), so two models with different orders of operations
# live in one process without anybody writing a global.
PHONE_RATE = os.environ.get('MORGANA_PHONE_RATE', '1') != '0'


def phone_rate_choice(choice=None):
    """The order of operations a call asks for: ``choice`` if given, else the process default (MORGANA_PHONE_RATE)."""
    return PHONE_RATE if choice is None else bool(choice)
PHONE_RATE_EXTRA = 1024          # rows behind the table that collect the gradients of padding frames (for the bias gradient)


def segment_bounds(rows, n_table_rows, pad_row=None):
    """(2, R) int32 frame runs per table row; with ``pad_row`` also the map with -1 replaced by that row (returned second)."""
    lib = _lib.load()
    rows = _require(rows, torch.int32, 'rows')
    seg = torch.empty((2, n_table_rows), dtype=torch.int32, device=rows.device)
    mapped = torch.empty_like(rows) if pad_row is not None else None
    _lib.check(lib.mg_segment_bounds(_p(rows), rows.numel(), n_table_rows, _p(seg[0]), _p(seg[1]), _p(mapped),
                                     -1 if pad_row is None else int(pad_row), _stream()), 'mg_segment_bounds')
    return seg if pad_row is None else (seg, mapped)


def phone_rate_table_ok(n_table_rows, m, n0, n1, act, enabled=None):
    """The table form of the phone-rate first layer (bf16 mode): sigmoid(X_phone W0^T + b0) is kept per phone and the second
    layer's GEMMs gather its rows, so the frame-rate activation never exists.  Needs the wide-tile kernels' shapes."""
    return (phone_rate_choice(enabled) and act == ACT_SIGMOID and n_table_rows < m and m >= 4096 and n_table_rows >= 2048
            and n0 % 128 == 0 and 384 < n0 <= 512 and pad8(n1) % 64 == 0)


def phone_rate_gru_ok(n_table_rows, m, width, enabled=None):
    """Linear / Sigmoid layers between an upsample and a GRU wrapper on the phone rows (utils.PhoneTable): worth it when there are
    several frames per phone; the table's width must suit mg_segment_sum (multiple of 8)."""
    return phone_rate_choice(enabled) and width % 8 == 0 and 2 * (n_table_rows + PHONE_RATE_EXTRA) <= m


def linear_dgrad_gathered_bf16(dy, m, n, wt_bf16, k, h_table, h_rows):
    """linear_dgrad_bf16 with the sigmoid outputs read from the per-phone table: dx[f] = (dy[f] W) h (1 - h), h = h_table[h_rows[f]]."""
    lib = _lib.load()
    dx = torch.empty((m, pad8(k)), dtype=torch.bfloat16, device=dy.device)
    _lib.check(lib.mg_linear_dgrad_gathered_bf16(_p(dy), dy.shape[1], m, n, _p(wt_bf16), wt_bf16.shape[1], k, _p(h_table),
                                                 h_table.shape[1], _p(h_rows), _p(dx), dx.shape[1], 0, _stream()),
               'mg_linear_dgrad_gathered_bf16')
    return dx


def phone_concat_layer(table_bf16, k_lab, rows_mapped, feat, w, w_bf16, bias, n, act, out_f32=False):
    """First Linear of ``cat(upsampled lab, frame counters)`` with the lab part at phone rate: ``act(P[rows] + feat W_cnt^T + b)``
    with P = table W_lab^T (one small GEMM, f32).  table_bf16 (R + extra, ld) bf16; feat (M, C) f32; w (n, k_lab + C) f32 and
    its bf16 copy w_bf16 (n, ldw).  Returns (M, pad8(n)) bf16 - the operand of the next layer - or (M, n rounded up to 8) f32."""
    lib = _lib.load()
    feat = _require(feat, torch.float32, 'frame features')
    w = _require(w, torch.float32, 'weight')
    m, c = feat.shape
    part = linear_fwd_bf16(table_bf16, None, table_bf16.shape[0], k_lab, w_bf16, None, n, ACT_NONE, out_f32=True)
    y = torch.empty((m, (n + 7) // 8 * 8 if out_f32 else pad8(n)), dtype=torch.float32 if out_f32 else torch.bfloat16, device=feat.device)
    _lib.check(lib.mg_phone_concat_layer_bf16(_p(part), part.shape[1], _p(rows_mapped), m, _p(feat), c, _p(w), w.shape[1], k_lab, _p(bias), n,
                                              act, _p(y), y.shape[1], int(bool(out_f32)), _stream()), 'mg_phone_concat_layer_bf16')
    return y


def segment_sum_feat(g, rows, seg, n_table_rows, n, feat, extra=PHONE_RATE_EXTRA):
    """``segment_sum`` of a bf16 gradient plus, from the same pass, the slabs of the frame features' weight gradient (g^T feat):
    returns (sums (R + extra, ld) bf16, slabs) - ``feat_wgrad_reduce`` adds the slabs into the features' columns of dW."""
    lib = _lib.load()
    g = _require(g, torch.bfloat16, 'gradient')
    feat = _require(feat, torch.float32, 'frame features')
    c = feat.shape[1]
    out = torch.empty((n_table_rows + extra, g.shape[1]), dtype=g.dtype, device=g.device)
    nbytes = lib.mg_segment_sum_feat_workspace_bytes(c, out.shape[1])
    slabs = torch.empty((nbytes // 4,), dtype=torch.float32, device=g.device)
    _lib.check(lib.mg_segment_sum_feat_bf16(_p(g), g.shape[1], _p(rows), rows.numel(), _p(seg[0]), _p(seg[1]), n_table_rows, extra, n, _p(out),
                                            out.shape[1], _p(feat), c, _p(slabs), nbytes, _stream()), 'mg_segment_sum_feat_bf16')
    return out, slabs


def feat_wgrad_reduce(slabs, c, ldo, n, dw, col0, accumulate=True):
    """dw[:, col0 : col0 + c] (+)= the sum of segment_sum_feat's slabs; dw (n, ldw) f32 contiguous."""
    lib = _lib.load()
    _lib.check(lib.mg_feat_wgrad_reduce(_p(slabs), c, ldo, n, _p(dw), dw.shape[1], col0, int(bool(accumulate)), _stream()), 'mg_feat_wgrad_reduce')


def segment_sum(g, rows, seg, n_table_rows, n, extra=PHONE_RATE_EXTRA):
    """(R + extra, ld) sums of the frame-rate rows of g per table row; the extra rows take the frames with row -1."""
    lib = _lib.load()
    bf16 = g.dtype == torch.bfloat16
    g = _require(g, torch.bfloat16 if bf16 else torch.float32, 'gradient')
    out = torch.empty((n_table_rows + extra, g.shape[1]), dtype=g.dtype, device=g.device)
    _lib.check(lib.mg_segment_sum(_p(g), g.shape[1], int(bf16), _p(rows), rows.numel(), _p(seg[0]), _p(seg[1]), n_table_rows, extra, n,
                                  _p(out), out.shape[1], _stream()), 'mg_segment_sum')
    return out
