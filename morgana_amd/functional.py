"""torch.autograd glue over the HIP kernels: every forward and backward below is a call into libmorgana_hip.so.

Three nodes cover the hot path of the reference's train step (experiment_builder.py:468-474):
  UpsampleFn      utils.upsample_to_repetitions (utils.py:175-228) and its adjoint (per-phone segment sum)
  LinearStackFn   a run of nn.Linear(+nn.Sigmoid) modules (README.rst:65-73), optionally fed by the fused gather
  GRUFn           RecurrentCuDNNWrapper(nn.GRU) with seq_len (utils.py:345-393)
  MaskedMSEFn     losses.mse (losses.py:29-51): loss and d loss / d prediction come out of one kernel pass

Precision: 'fp32' = exact-fp32 MFMA (parity mode, 1e-4 vs the reference), 'bf16' = bf16 operands / fp32 accumulate
(throughput mode; master weights, biases, loss and optimiser state stay fp32), 'bf16x3' = split-bf16 operands (hi + lo, three
bf16 MFMA products per fp32 product, fp32 accumulate and fp32 activations: parity grade - 1e-4 vs the reference - at the bf16
matrix rate; row-wise Linear / Sigmoid layers only, recurrent cells run their exact-fp32 form in this mode).
"""
import os

import torch

from . import ops

_PRECISION = 'fp32'


def set_precision(precision):
    """Global default for modules that do not pin their own precision: 'fp32', 'bf16' or 'bf16x3'."""
    global _PRECISION
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of %s, got %r" % (', '.join(repr(p) for p in PRECISIONS), precision))
    _PRECISION = precision


PRECISIONS = ('fp32', 'bf16', 'bf16x3')


def recurrent_precision(precision):
    """The precision a recurrent cell runs in when its container asks for ``precision``: 'bf16x3' is a mode of the row-wise layers
    (their operands are split); a recurrence under it takes its exact-fp32 form."""
    return 'fp32' if precision == 'bf16x3' else precision


def get_precision():
    return _PRECISION


_ONES = {}

# Called (no arguments) from inside a backward pass at the moment every parameter gradient EXCEPT those of the stack's first Linear
# layer has reached the optimiser's flat buffer - the data-parallel step starts exchanging that part while the first layer's weight
# gradient (81 % of the bytes of the README F0Model, and the last kernel of the backward) is still being computed.
_EARLY_GRADS_HOOK = None


def set_early_grads_hook(fn):
    """Install (or with None remove) the callback described above; returns the previous one.  The callback receives the firing
    stack's parameter list (weight, bias, weight, ...): the step checks that this stack IS the model before it trusts the promise."""
    global _EARLY_GRADS_HOOK
    prev, _EARLY_GRADS_HOOK = _EARLY_GRADS_HOOK, fn
    return prev


def backward(loss):
    """``loss.backward()`` (experiment_builder.py:473) with the implicit gradient of one taken from a per-device cache:
    autograd otherwise allocates and fills a fresh ``ones_like(loss)`` every step - a launch of its own at the ~5 us floor."""
    global _DIRECT_BACKWARD
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    _DIRECT_BACKWARD += 1          # the row-wise and recurrent layers may add their weight gradients straight into .grad (_direct_params)
    try:
        loss.backward(gradient=one if loss.dim() == 0 else None)
    finally:
        _DIRECT_BACKWARD -= 1
        _join_side()               # weight gradients that ran beside the backward pass are complete before anything that follows


# ---- weight gradients beside the backward pass ------------------------------------------------------------------------------------
# In a backward pass the gradient chain (dgrad, the recurrence) is the critical path; a layer's WEIGHT gradient feeds nothing but the
# update at the very end.  Inside ``backward`` above - where the parameter gradients go straight into the optimiser's flat buffer and
# nobody reads them before the update - the row-wise layers outside the fused F0 stack and the GRU wrapper issue their weight-gradient
# GEMMs on a side stream: behind the event "their operands are ready", beside the rest of the pass (C4: the post-GRU layers' weight
# gradients run beside the backward recurrence's first steps, the 100 GFLOP recurrent weight gradient beside the input-side chain
# segment sum -> input weight gradient -> dgrad -> first layer).  ``backward`` joins the stream before it returns.  Scratch is per
# stream (ops.workspace), operands stay referenced until the join; inside a stream capture fork and join become graph edges.
# MEASURED (round 4, same-box A/B, profiles/r4_notes_side_streams.txt) and therefore OFF by default: level 1 (the recurrent weight
# gradient beside the input-side chain) changes nothing - C4 3.383-3.393 against 3.391-3.392 ms, C5 6.598-6.615 against 6.606-6.620: each
# of these GEMMs fills every CU with a 512-thread workgroup, so the "two streams" take turns - and level 2 (the post-GRU layers'
# weight gradients too, which then run when the backward recurrence is launched) costs C4 0.7 ms (4.10-4.12 ms): the persistent
# launch finds CUs taken, its workgroups land unevenly over the XCDs and the groups fall to the write-through hand-off.
SIDE_STREAMS = int(os.environ.get('MORGANA_SIDE_STREAMS', '0'))      # 0 = off, 1 = the recurrent weight gradient, 2 = row-wise stacks too
_side_streams = {}
_side_pending = []      # (main stream, side stream, tensors the side work reads) since the last join


def _side_ok(device, level=1):
    return (SIDE_STREAMS >= level and _DIRECT_BACKWARD > 0 and _EARLY_GRADS_HOOK is None and device.type == 'cuda')


class _Beside(object):
    """``with _Beside(device, *tensors):`` - the launches inside go to the device's side stream, ordered behind everything the current
    stream holds so far; ``tensors`` (what those launches read) stay alive until ``_join_side``."""

    def __init__(self, device, *keep):
        self.device, self.keep = device, keep

    def __enter__(self):
        main = torch.cuda.current_stream(self.device)
        side = _side_streams.get(self.device.index)
        if side is None:
            side = _side_streams[self.device.index] = torch.cuda.Stream(self.device)
            ops.SIDE_STREAM_IDS.add(side.cuda_stream)
        side.wait_stream(main)
        _side_pending.append((main, side, self.keep))
        self._ctx = torch.cuda.stream(side)
        self._ctx.__enter__()
        return side

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)


def _join_side():
    joined = set()
    for main, side, _ in _side_pending:
        if (main.cuda_stream, side.cuda_stream) not in joined:
            joined.add((main.cuda_stream, side.cuda_stream))
            main.wait_stream(side)
    del _side_pending[:]


_DIRECT_BACKWARD = 0
# Set by graphs.GraphedTrainStep while it captures a whole training step: the fused stack then leaves the last small jobs of its forward
# (phone rate: the repeated prediction and the fused tail's slab sum; frame rate: the tail's slab sum) to a later launch of the same
# step - the update launch's first blocks on one rank, rider blocks behind the backward's first grid on a data-parallel rank
# (LinearStackMSEFn).  A replay cannot observe the step half done; the eager loop keeps the separate launches.
DEFER_TAIL = False


def set_defer_tail(enabled):
    global DEFER_TAIL
    prev, DEFER_TAIL = DEFER_TAIL, bool(enabled) and os.environ.get('MORGANA_DEFER_TAIL', '1') != '0'
    return prev


def _is_unit_grad(g):
    """Is ``g`` the cached gradient of one that ``backward`` above hands to autograd (the loss IS the root: multiplying by it is a no-op)?"""
    return g is not None and any(g.data_ptr() == one.data_ptr() for one in _ONES.values())
DIRECT_GRADS = os.environ.get('MORGANA_DIRECT_GRADS', '1') != '0'          # A/B switches of the two launch savers below
WEIGHT_SHADOWS = ops.WEIGHT_SHADOWS


def _w_plain(weights):
    """bf16 [N, pad_ld(K)] operands of a list of fp32 weights: the shadows that live on the parameters and that the optimiser's update
    kernel keeps current (ops.param_shadows: no cast launch per layer and step), or a cast each (MORGANA_WEIGHT_SHADOWS=0)."""
    if WEIGHT_SHADOWS:
        return ops.param_shadows(list(weights))[0]
    return [ops.cast_pad_bf16(ops._require(w, torch.float32, 'weight')) for w in weights]


def _w_t(w):
    """bf16 W^T [K, pad_ld(N)] (the dgrad operand) of one fp32 weight, from its shadow where there is one."""
    return ops.param_shadows([w], want_t=(0,))[1][0] if WEIGHT_SHADOWS else ops.cast_transpose_bf16(w)


def _direct_params(*params):
    """May a layer's backward add its parameter gradients straight into ``.grad`` and hand autograd ``None`` for them?  Only inside
    ``backward`` above (``torch.autograd.grad`` and double backward must get tensors), and only when every parameter's ``.grad`` is
    a live fp32 view of ``morgana_amd.optim.Adam``'s flat gradient, still asks for a gradient (a parameter frozen after the optimiser
    was built must receive none) and carries no hook (tensor hooks and post-accumulate-grad hooks fire from autograd's own
    accumulation, which this path bypasses).  What it saves: autograd's AccumulateGrad adds the returned tensor into the existing
    ``.grad`` with one elementwise launch PER PARAMETER (ten of them in an RNN_SPSS step)."""
    return (DIRECT_GRADS and _DIRECT_BACKWARD > 0 and not torch.is_grad_enabled() and
            all(p is not None and getattr(p, '_mg_direct_grad', False) and p.requires_grad and p.grad is not None and
                p.grad.is_contiguous() and p.grad.dtype == torch.float32 and not _has_hooks(p) for p in params))


def _x3_layer(w):
    """Does a Linear layer of this weight run on split-bf16 operands in 'bf16x3' mode?  The wide layers do (that is where the
    products are); a narrow one - fewer than 16,384 weights: the README stack's 128 -> 32 and 32 -> 1 - runs the exact-fp32 tile
    programs of fp32 mode instead: its cost is nothing, and its output is a cancelling sum of a few O(0.1) terms, where the 2^-17
    of a split operand is a visible fraction of the result (C2 prediction vs the oracle: 1.6e-4 split, 2e-5 exact)."""
    return w.shape[0] * w.shape[1] >= 16384


def _has_hooks(p):
    return bool(getattr(p, '_backward_hooks', None)) or bool(getattr(p, '_post_accumulate_grad_hooks', None))


def _linear_grads_direct(w_param, b_param, g, a_in, rows, m, n, k):
    """dW | db of one bf16 Linear into the optimiser (slabs deferred to its update kernel or reduced into .grad: _wgrad_into)."""
    opt = getattr(w_param, '_mg_optimizer', None)
    mode = 'defer' if (opt is not None and opt.defers_slabs() and _grads_adjacent(w_param, b_param)) else 'direct'
    _wgrad_into(mode, opt, w_param, b_param, g, a_in, rows, m, n, k)


class UpsampleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sequence_feature, dur2d, t_cap):
        x = ops._require(sequence_feature, torch.float32, 'sequence_feature')
        b, p, f = x.shape
        _, rows = ops.upsample_index(dur2d, t_cap)
        out = ops.gather_rows(x.view(b * p, f), rows.view(-1)).view(b, t_cap, f)
        ctx.save_for_backward(dur2d)
        ctx.n_phones = p
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (dur2d,) = ctx.saved_tensors
        return ops.upsample_backward(grad_out.contiguous(), dur2d, ctx.n_phones), None, None


class DropoutFn(torch.autograd.Function):
    """Active ``nn.Dropout(p)`` (training mode) as a HIP kernel: y = x * keep / (1 - p); the mask is a function of (torch's seed, site,
    the device's step counter, element index) - csrc/dropout.hip - and the backward regenerates it.  forward(ctx, x, p, site)."""

    @staticmethod
    def forward(ctx, x, p, site):
        x = x.contiguous()
        seed, used = ops.dropout_seed(), ops.dropout_draw(x.device)
        ctx.args = (float(p), seed, int(site))
        ctx.save_for_backward(used)
        return ops.dropout(x, p, seed, site, used)

    @staticmethod
    def backward(ctx, grad):
        (used,) = ctx.saved_tensors
        p, seed, site = ctx.args
        return ops.dropout(grad.contiguous(), p, seed, site, used), None, None


class _RunDropout(object):
    """The dropout masks of one LinearStackFn node: ``drops[i]`` = probability behind layer i (0 = none), one draw of the device's
    step counter for the node, site = site0 + i.  ``apply`` is forward and backward alike (the mask multiplies either)."""

    def __init__(self, spec, device):
        self.drops, self.site0 = spec
        self.seed, self.used = ops.dropout_seed(), ops.dropout_draw(device)

    def active(self, i):
        return 0 <= i < len(self.drops) and self.drops[i] > 0

    def apply(self, x, i, inplace=False):
        if not self.active(i):
            return x
        x = x if x.is_contiguous() else x.contiguous()
        return ops.dropout(x, self.drops[i], self.seed, self.site0 + i, self.used, out=x if inplace else None)


class LinearStackFn(torch.autograd.Function):
    """y = L_n(...sigma(L_1(x))...) over rows of a 2-D input (or of a gathered table).

    forward(ctx, spec, x2d, rows, *params); spec = (acts, precision); params = w0, b0, w1, b1, ... (bias may be None).
    With ``rows`` the input row m is ``x2d[rows[m]]`` (zero row for -1): the frame-rate tensor is never materialised.
    When such a gathered input needs a gradient (packed frames, ``utils.FrameLayout``: the rows are distinct) the input gradient is
    scattered back to the rows of ``x2d``; rows no index points at get zeros.
    """

    @staticmethod
    def forward(ctx, spec, x2d, rows, *params):
        acts, precision = spec[:2]
        extra = spec[2] if len(spec) > 2 else 0       # zero rows appended behind x2d (phone-rate tables: what padding frames gather)
        rows_runs = bool(spec[3]) if len(spec) > 3 else False      # `rows` is an upsample frame map (runs of equal indices): a hint
        # active dropout behind layers of the run (utils._Run.drop_spec): masks drawn here, regenerated in backward; the activation
        # saved for the backward is the one BEFORE the mask (the sigmoid gradient needs it), the masked copy feeds the next layer
        drop = _RunDropout(spec[4], x2d.device) if len(spec) > 4 and spec[4] is not None else None
        ctx.drop = drop
        # a bf16 copy of x2d somebody already made (the GRU recurrence writes one of its output: GRUFn out_bf): the first layer's operand
        shadow = spec[5] if len(spec) > 5 else None
        # `front` = (rows, seg, frame features): the input is cat(x2d[rows], frame features) - the table of phone rows repeated by
        # upsample_to_repetitions with per-frame counters behind it (models/RNN_SPSS.py:76-81).  bf16 mode: the table's part of the first
        # layer runs once per phone, ops.phone_concat_layer adds the counters' part per frame; rows: -1 mapped to the first extra row
        front = spec[6] if len(spec) > 6 else None
        n_layers = len(acts)
        weights = [params[2 * i] for i in range(n_layers)]
        biases = [params[2 * i + 1] for i in range(n_layers)]
        # a bf16 input is an already padded layer-1 operand (ops.gather_concat); anything else must be fp32
        pre_cast = precision == 'bf16' and x2d.dtype == torch.bfloat16
        x2d = ops._require(x2d, torch.bfloat16 if pre_cast else torch.float32, 'input')
        if extra and (pre_cast or rows is not None or ctx.needs_input_grad[1]):
            raise ValueError('LinearStackFn: extra zero rows go with a plain fp32 input that needs no gradient')
        m = rows.numel() if rows is not None else x2d.shape[0] + extra
        if front is not None:
            if precision != 'bf16' or pre_cast or rows is not None or ctx.needs_input_grad[1] or not extra:
                raise ValueError('LinearStackFn: a phone table with frame features is a bf16-mode input that needs no gradient')
            m = front[0].numel()
        ctx.spec, ctx.m = spec, m
        ctx.n_src = x2d.shape[0]
        gathered_grad = rows is not None and ctx.needs_input_grad[1]
        rows_k = rows                                  # the row map the kernels' loaders apply (None once the input is packed)
        ctx.has_bias = [b is not None for b in biases]
        ctx.dims = [(w.shape[0], w.shape[1]) for w in weights]
        k_in = weights[0].shape[1] - (front[2].shape[1] if front is not None else 0)
        if x2d.shape[1] != (ops.pad_ld(k_in) if pre_cast else k_in):
            raise ValueError('Linear expects %d input features, got %d' % (k_in, x2d.shape[1]))
        hidden = []
        if precision == 'fp32':
            if extra:
                x2d = torch.cat((x2d, x2d.new_zeros((extra, x2d.shape[1]))))
            a, r = x2d, rows
            for i in range(n_layers):
                w = ops._require(weights[i], torch.float32, 'weight')
                a = ops.linear_fwd_f32(a, r, m, w, biases[i], acts[i])
                r = None
                hidden.append(a)
                if drop is not None:
                    a = drop.apply(a, i)
            out = a
            ctx.save_for_backward(x2d, rows_k, rows if gathered_grad else None, *weights, *hidden)
        elif precision == 'bf16x3':
            # split-bf16: fp32 activations as in fp32 mode, every product as ONE bf16 GEMM over split operands (csrc/split3.hip)
            split = [_x3_layer(w) for w in weights]
            if extra and not split[0]:
                x2d, extra = torch.cat((x2d, x2d.new_zeros((extra, x2d.shape[1])))), 0
            a, r = x2d, rows
            w3s = ops.x3_weight_operands([w for w, s in zip(weights, split) if s])[0]
            kept = [None] * n_layers              # the activation splits: the backward's weight gradients multiply with them again
            for i in range(n_layers):
                if split[i]:
                    # the zero rows behind a phone table (what padding frames gather) are rows of the split operand, never of an fp32 copy
                    a3 = ops.split3([(a, 0, False, extra if i == 0 else 0)])[0]
                    kept[i] = a3
                    a = ops.linear_fwd_x3(a3, r, m, w3s.pop(0), biases[i], weights[i].shape[0], acts[i])
                else:
                    a = ops.linear_fwd_f32(a, r, m, ops._require(weights[i], torch.float32, 'weight'), biases[i], acts[i])
                r = None
                hidden.append(a)
                if drop is not None:
                    a = drop.apply(a, i)
            out = a
            ctx.save_for_backward(x2d, rows_k, rows if gathered_grad else None, *weights, *hidden)
            ctx.param_refs = (list(weights), list(biases))
            ctx.x3_extra = extra
            ctx.x3_kept = kept
        else:
            if gathered_grad:
                if pre_cast:
                    raise ValueError('LinearStackFn: a gathered input that needs a gradient must be fp32')
                # pack and cast in one pass: the bf16 operand holds exactly the gathered rows
                a, rows_k = ops.gather_rows(x2d, rows, out_bf16=True, ld=ops.pad_ld(k_in)), None
            elif (shadow is not None and not pre_cast and not extra and shadow.dtype == torch.bfloat16 and shadow.is_contiguous()
                  and tuple(shadow.shape) == (x2d.shape[0], ops.pad_ld(k_in)) and k_in == ops.pad_ld(k_in)):
                a = shadow
            else:
                # (a phone table in front of frame features is padded to the layer's full input width: its weight gradient then
                # spans all of dW, with zeros in the features' columns - ops.feat_wgrad_reduce fills those)
                a = x2d if pre_cast else ops.cast_pad_bf16(x2d, ld=ops.pad_ld(weights[0].shape[1]) if front is not None else None,
                                                           extra_rows=extra)
            a0, r = a, rows_k
            # bf16 operands of the weights: copies that live on the parameters, refreshed by the optimiser's update kernel
            # (ops.param_shadows) - no cast launch per layer and step
            w_bfs = _w_plain(weights)
            for i in range(n_layers):
                n, k = weights[i].shape
                last = i == n_layers - 1
                if i == 0 and front is not None:
                    a = ops.phone_concat_layer(a0, k_in, front[0], front[2], weights[0], w_bfs[0], biases[0], n, acts[0], out_f32=last)
                else:
                    a = ops.linear_fwd_bf16(a, r, m, k, w_bfs[i], biases[i], n, acts[i], out_f32=last, rows_runs=rows_runs)
                r = None
                hidden.append(a)
                if drop is not None and not last:
                    a = drop.apply(a, i)                 # over the padded bf16 buffer: zero padding stays zero
            n_last = weights[-1].shape[0]
            out = hidden[-1]
            if out.shape[1] != n_last:
                out = out[:, :n_last].contiguous()
            unmasked = out                               # what a trailing sigmoid's gradient reads in backward
            if drop is not None:
                out = drop.apply(out, n_layers - 1)
            ctx.save_for_backward(a0, rows_k, rows if gathered_grad else None, *weights, *hidden[:-1], unmasked)
            ctx.param_refs = (list(weights), list(biases))        # the Parameter objects (their .grad views, their shadows)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        acts, precision = ctx.spec[:2]
        n_layers = len(acts)
        saved = ctx.saved_tensors
        x_in, rows, scatter_to = saved[0], saved[1], saved[2]
        weights = saved[3:3 + n_layers]
        hidden = saved[3 + n_layers:]
        m = ctx.m
        grads = [None] * (2 * n_layers)
        need_x = ctx.needs_input_grad[1]
        drop = getattr(ctx, 'drop', None)

        def masked_input(i):
            # the operand layer i multiplied in the forward pass: the saved activation of layer i - 1 under its dropout mask (regenerated)
            return hidden[i - 1] if drop is None else drop.apply(hidden[i - 1], i - 1)

        def unmask(g_, i):
            # a gradient with respect to a MASKED activation -> with respect to the activation: the same mask (it commutes with the
            # sigmoid gradient the dgrad kernels fuse, which reads the unmasked activation)
            return g_ if drop is None else drop.apply(g_, i, inplace=True)

        g = grad_out.contiguous()
        if drop is not None and drop.active(n_layers - 1):
            g = drop.apply(g, n_layers - 1)
        if acts[-1] == ops.ACT_SIGMOID:
            g = ops.sigmoid_grad(g, hidden[-1])
        grad_x = None
        if precision == 'fp32':
            for i in range(n_layers - 1, -1, -1):
                n, k = ctx.dims[i]
                a_in, r = (x_in, rows) if i == 0 else (masked_input(i), None)
                dw, db = ops.linear_wgrad_f32(g, a_in, r, n, k, want_bias=ctx.has_bias[i])
                grads[2 * i], grads[2 * i + 1] = dw, db
                if i > 0:
                    h = hidden[i - 1] if acts[i - 1] == ops.ACT_SIGMOID else None
                    g = unmask(ops.linear_dgrad_f32(g, weights[i], h), i - 1)
                elif need_x:
                    grad_x = ops.linear_dgrad_f32(g, weights[0], None)
        elif precision == 'bf16x3':
            w_params, b_params = ctx.param_refs
            direct = all(ctx.has_bias) and _direct_params(*w_params, *b_params)
            split = [_x3_layer(w) for w in w_params]
            dgrad_of = [i for i in range(n_layers) if split[i] and (i > 0 or need_x)]
            wt3_list = ops.x3_weight_operands([w_params[i] for i in dgrad_of], want_t=tuple(range(len(dgrad_of))))[1]
            wt3s = dict(zip(dgrad_of, wt3_list))
            sig = None        # sigmoid outputs whose gradient factor s (1 - s) g still lacks: fused into the next split of g (one pass less)
            for i in range(n_layers - 1, -1, -1):
                n, k = ctx.dims[i]
                a_in, r = (x_in, rows) if i == 0 else (masked_input(i), None)
                if not split[i]:                       # a narrow layer: exact fp32 products (see _x3_layer)
                    grads[2 * i], grads[2 * i + 1] = ops.linear_wgrad_f32(g, a_in, r, n, k, want_bias=ctx.has_bias[i])
                    if i > 0:
                        g = unmask(ops.linear_dgrad_f32(g, weights[i], hidden[i - 1] if acts[i - 1] == ops.ACT_SIGMOID else None), i - 1)
                    elif need_x:
                        grad_x = ops.linear_dgrad_f32(g, weights[0], None)
                    continue
                need_g3 = i > 0 or need_x
                kept = ctx.x3_kept[i]
                if r is None and kept is not None and ops.split3_colsum_ok(n):
                    # no row map: the gradient is split ONCE, [hi | lo | hi] - the layout its dgrad takes against W^T [hi | hi | lo] - and
                    # the weight gradient is one launch of that buffer against the FORWARD's split of the layer's input, both read as
                    # row-interleaved (3 m, ldp) stacks (ops.linear_wgrad_x3_rows); the bias gradient = the exact column sums of the
                    # fp32 gradient, taken by the split pass.  (Was: gradient and input split again into row-stacked planes.)
                    g1, colsum = ops.split3([(g, 1, False, 0, sig, True)])[0]
                    sig = None
                    g3 = g1
                    if direct:
                        ops.linear_wgrad_x3_rows(g1, colsum, kept, n, k, out_w=w_params[i].grad, out_b=b_params[i].grad, accumulate=True)
                    else:
                        dw, db = ops.linear_wgrad_x3_rows(g1, colsum if ctx.has_bias[i] else None, kept, n, k)
                        grads[2 * i], grads[2 * i + 1] = dw, db
                else:
                    parts = ops.split3([(g, 2, False, 0, sig), (a_in, 2, False, ctx.x3_extra if i == 0 else 0)] +
                                       ([(g, 1, False, 0, sig)] if need_g3 else []))
                    sig = None
                    g2, a2, g3 = parts[0], parts[1], (parts[2] if need_g3 else None)
                    if direct:
                        ops.linear_wgrad_x3(g2, a2, r, m, n, k, out_w=w_params[i].grad, out_b=b_params[i].grad, accumulate=True)
                    else:
                        dw, db = ops.linear_wgrad_x3(g2, a2, r, m, n, k)
                        grads[2 * i], grads[2 * i + 1] = dw, (db if ctx.has_bias[i] else None)
                if i > 0:
                    g = unmask(ops.linear_dgrad_x3(g3, m, wt3s[i], k), i - 1)
                    if acts[i - 1] == ops.ACT_SIGMOID:
                        if split[i - 1]:
                            sig = hidden[i - 1]             # layer i - 1 splits g next: the sigmoid gradient rides in that pass
                        else:
                            g = ops.sigmoid_grad(g, hidden[i - 1])
                elif need_x:
                    grad_x = ops.linear_dgrad_x3(g3, m, wt3s[0], k)
        else:
            g = ops.cast_pad_bf16(g)
            w_params, b_params = ctx.param_refs
            direct = all(ctx.has_bias) and _direct_params(*w_params, *b_params)
            beside = direct and _side_ok(g.device, level=2)
            front = ctx.spec[6] if len(ctx.spec) > 6 else None
            extra0 = ctx.spec[2] if len(ctx.spec) > 2 else 0
            for i in range(n_layers - 1, -1, -1):
                n, k = ctx.dims[i]
                a_in, r = (x_in, rows) if i == 0 else (masked_input(i), None)
                if i == 0 and front is not None:
                    # dW = [dW_lab | dW_cnt]: the table's part from the per-phone sums of g (every frame of a phone multiplied the same
                    # table row; the table's columns behind the labels are zero), the counters' part from the same pass over g
                    # (slabs, reduced into their columns of dW afterwards); db = the column sums of the sums
                    rows_f, seg_f, feat = front
                    k_lab, n_tab = k - feat.shape[1], x_in.shape[0]
                    sums, slabs = ops.segment_sum_feat(g, rows_f, seg_f, n_tab - extra0, g.shape[1], feat, extra=extra0)
                    if direct:
                        ops.linear_wgrad_bf16(sums, x_in, None, n_tab, n, k, out_w=w_params[0].grad, out_b=b_params[0].grad, accumulate=True)
                        ops.feat_wgrad_reduce(slabs, feat.shape[1], sums.shape[1], n, w_params[0].grad, k_lab)
                    else:
                        dw, db = ops.linear_wgrad_bf16(sums, x_in, None, n_tab, n, k, want_bias=ctx.has_bias[0])
                        ops.feat_wgrad_reduce(slabs, feat.shape[1], sums.shape[1], n, dw, k_lab)
                        grads[0], grads[1] = dw, db
                    continue
                if beside and (i > 0 or need_x):
                    # the weight gradient feeds only the update: beside the dgrad chain (the last layer of the pass has no chain left
                    # on this node, its weight gradient stays in line)
                    with _Beside(g.device, g, a_in, r):
                        _linear_grads_direct(w_params[i], b_params[i], g, a_in, r, m, n, k)
                elif direct:
                    _linear_grads_direct(w_params[i], b_params[i], g, a_in, r, m, n, k)
                else:
                    dw, db = ops.linear_wgrad_bf16(g, a_in, r, m, n, k, want_bias=ctx.has_bias[i])
                    grads[2 * i], grads[2 * i + 1] = dw, db
                if i > 0:
                    wt = _w_t(w_params[i])
                    h = hidden[i - 1] if acts[i - 1] == ops.ACT_SIGMOID else None
                    g = unmask(ops.linear_dgrad_bf16(g, m, n, wt, k, h), i - 1)
                elif need_x:
                    wt = _w_t(w_params[0])
                    grad_x = ops.linear_dgrad_bf16(g, m, n, wt, k, None, out_f32=True)
                    if grad_x.shape[1] != k:
                        grad_x = grad_x[:, :k].contiguous()
        if need_x and scatter_to is not None:
            # the input rows were gathered (distinct rows): their gradients go back where they came from, all other rows get zeros
            grad_x = ops.scatter_rows(grad_x, scatter_to, ctx.n_src)
        return (None, grad_x, None) + tuple(grads)


class RepeatTableRowsFn(torch.autograd.Function):
    """``upsample_to_repetitions`` (morgana/utils.py:175-228) applied AFTER a row-wise stack instead of before it: out[f] =
    table[rows[f]] for a table of phone rows + extra rows (rows: -1 already mapped to the first extra row).  Backward sums each
    phone's frames (``mg_segment_sum``; padding frames into the extra rows), so the stack's whole backward runs on table rows."""

    @staticmethod
    def forward(ctx, table, rows, seg, n_phone_rows):
        table = ops._require(table, torch.float32, 'table')
        ctx.save_for_backward(rows, seg)
        ctx.n_phone_rows, ctx.extra, ctx.width = n_phone_rows, table.shape[0] - n_phone_rows, table.shape[1]
        return ops.gather_rows(table, rows)

    @staticmethod
    def backward(ctx, grad_out):
        rows, seg = ctx.saved_tensors
        g = grad_out.contiguous()
        width = ctx.width
        if width % 8 != 0:                       # mg_segment_sum works on 8-column chunks: pad narrow outputs (the 1-wide F0 stream)
            g = torch.nn.functional.pad(g, (0, 8 - width % 8))
        sums = ops.segment_sum(g, rows, seg, ctx.n_phone_rows, g.shape[1], extra=ctx.extra)
        return (sums[:, :width].contiguous() if sums.shape[1] != width else sums), None, None, None


class PhoneMSEFn(torch.autograd.Function):
    """``losses.mse(upsample_to_repetitions(table), target, seq_len)`` (morgana/losses.py:29-51 behind utils.py:175-228) for a ONE-column
    table of per-phone predictions, without the frame-rate tensor in the differentiable path: every frame of a phone shares the
    phone's prediction p, so  sum_f w_f (p - y_f)^2 = weight (p - ybar)^2 + c  with the phone's weighted target mean ybar and a
    constant c (ops.phone_front / phone_target_stats); loss and d loss / d table come from the table rows (``mg_phone_mse_rows_f32``)
    - no gather of the prediction in front of the loss, no segment sum of its gradient behind it.  The exact-fp32 modes' counterpart
    of the bf16 step's fused tail.  forward(ctx, table (R + extra, 1), target (B, T, 1), seq_len, holder) -> (loss, prediction
    (B, T, 1), repeated for reporting, not differentiable).  ``holder``: the ``utils.UpsampledSequence`` that owns the frame map."""

    @staticmethod
    def forward(ctx, table, target, seq_len, holder):
        table = ops._require(table, torch.float32, 'prediction table')
        target = ops._require(target, torch.float32, 'targets')
        b, t = target.shape[0], target.shape[1]
        extra = table.shape[0] - holder.source.shape[0] * holder.source.shape[1]
        rows_mapped, ybar, weight, partials, n_src = _phone_stats(holder, target, seq_len, extra)
        loss, dpred = ops.phone_mse_rows(table, ybar, weight)
        pred = ops.expand_column(table.reshape(-1) if table.shape[1] == 1 else table[:, 0].contiguous(), rows_mapped,
                                 loss_const=(partials, n_src, extra, loss))
        ctx.save_for_backward(dpred)
        ctx.cols = table.shape[1]
        pred = pred.view(b, t, 1)
        ctx.mark_non_differentiable(pred)
        return loss.reshape(()), pred

    @staticmethod
    def backward(ctx, grad_loss, grad_pred):
        (dpred,) = ctx.saved_tensors
        g = (dpred if _is_unit_grad(grad_loss) else dpred * grad_loss).unsqueeze(1)
        if ctx.cols != 1:
            g = torch.nn.functional.pad(g, (0, ctx.cols - 1))
        return g, None, None, None


def _phone_stats(holder, target, seq_len, extra):
    """(rows_mapped, ybar, weight, partials, n_src) of the per-phone masked MSE for the frame map ``holder`` owns: map and statistics
    from ONE launch where the map has not been built yet (ops.phone_front), else the statistics alone."""
    b, t = target.shape[0], target.shape[1]
    n_src = holder.source.shape[0] * holder.source.shape[1]
    if seq_len.dtype != torch.int64:
        seq_len = seq_len.long()
    if holder.pending() and holder.t_cap == t and ops.phone_front_ok(holder.source.shape[0], holder.source.shape[1], t, extra):
        rows, rows_mapped, seg, ybar, weight, partials = ops.phone_front(holder.dur, target.reshape(-1), seq_len, t, extra)
        holder.adopt(rows, (seg, rows_mapped.reshape(-1)))
    else:
        rows = holder.rows
        seg, rows_mapped = holder.phone_maps()
        ybar, weight, partials = ops.phone_target_stats(target.reshape(-1), rows.reshape(-1), seg, seq_len, b, t, n_src, extra)
    return rows_mapped.reshape(-1), ybar, weight, partials, n_src


class F0TailRowsF32Fn(torch.autograd.Function):
    """The README F0Model's tail ``Sigmoid -> Linear(128, 32) -> Sigmoid -> Linear(32, 1)`` (README.rst:65-73) TOGETHER with
    ``losses.mse`` (morgana/losses.py:29-51) on per-phone rows, exact fp32, forward and backward in one launch (``mg_f0_tail_rows_f32``):
    the ``fp32`` / ``bf16x3`` modes' counterpart of the bf16 step's fused tail.  forward(ctx, z2 (R + extra, 128) pre-activations of
    the 128-wide layer, target (B, T, 1), seq_len, holder, w3, b3, w4, b4) -> (loss, prediction (B, T, 1): repeated for reporting, not
    differentiable).  ``holder`` None: z2 holds the (B * T, 128) frame rows themselves (the reference's order of operations) and the
    kernel takes the targets and the loss's own weights per frame.  The backward hands out what the forward's launch computed."""

    @staticmethod
    def forward(ctx, z2, target, seq_len, holder, w3, b3, w4, b4):
        target = ops._require(target, torch.float32, 'targets')
        b, t = target.shape[0], target.shape[1]
        n_g = ops.F0_TAIL_F32_GRADS
        if holder is None:
            # rows = the B x T frames themselves (the reference's order of operations): targets and the masked MSE's own weights per row
            sl = seq_len if (seq_len is None or seq_len.dtype == torch.int64) else seq_len.long()
            pred_rows, dz2, flat = ops.f0_tail_rows_f32(z2, w3, b3, w4, b4, target.reshape(-1), None, seq_len=sl, frames=(b, t))
            loss = flat[n_g:n_g + 1]
            pred = pred_rows.view(b, t, 1)
        else:
            extra = z2.shape[0] - holder.source.shape[0] * holder.source.shape[1]
            rows_mapped, ybar, weight, partials, n_src = _phone_stats(holder, target, seq_len, extra)
            pred_rows, dz2, flat = ops.f0_tail_rows_f32(z2, w3, b3, w4, b4, ybar, weight)
            loss = flat[n_g:n_g + 1]
            pred = ops.expand_column(pred_rows, rows_mapped, loss_const=(partials, n_src, extra, loss)).view(b, t, 1)
        ctx.save_for_backward(dz2, flat)
        ctx.params = (w3, b3, w4, b4)
        ctx.mark_non_differentiable(pred)
        return loss.reshape(()), pred

    @staticmethod
    def backward(ctx, grad_loss, grad_pred):
        dz2, flat = ctx.saved_tensors
        unit = _is_unit_grad(grad_loss)
        n_g = ops.F0_TAIL_F32_GRADS
        grads = _deliver_param_grads(ctx.params, flat[:n_g], [0, 32 * 128, 32 * 128 + 32, 32 * 128 + 64], None if unit else grad_loss)
        return (dz2 if unit else dz2 * grad_loss, None, None, None) + tuple(grads)


class UnpackRowsFn(torch.autograd.Function):
    """Packed rows (total + 1, D) -> dense (B*T, D): dense row (b, t) takes its packed row, every padded frame takes the one
    representative row (index ``total``).  Backward: the valid rows' gradients are gathered back by ``rows`` and the representative
    row receives the sum over all padded frames (``mg_pad_rows_colsum_f32``) - exact also for losses that do not mask."""

    @staticmethod
    def forward(ctx, packed, rows, inverse, seq_len, b, t):
        packed = ops._require(packed, torch.float32, 'packed rows')
        ctx.save_for_backward(rows, seq_len)
        ctx.shape = (b, t, packed.shape[1])
        return ops.gather_rows(packed, inverse)

    @staticmethod
    def backward(ctx, grad_out):
        rows, seq_len = ctx.saved_tensors
        b, t, d = ctx.shape
        g = grad_out.contiguous()
        out = ops.gather_rows(g.view(b * t, d), rows)            # rows[total] = -1: a zero row, replaced by the padded frames' sum
        ops.pad_rows_colsum(g.view(b, t, d), seq_len, out[rows.numel() - 1])
        return out, None, None, None, None, None


def _deliver_param_grads(params, flat, offsets, grad_loss=None):
    """Hand the parameter gradients held in one contiguous buffer `flat` (parameter order) to autograd.

    Fast path: the parameters belong to morgana_amd.optim.Adam (flag `_mg_direct_grad`) and their .grad tensors are
    consecutive views of ITS flat gradient buffer in the same order - then one fused add (times grad_loss) into that
    buffer replaces one autograd accumulation kernel per parameter, and None is returned for every parameter.
    Otherwise ordinary per-parameter gradient tensors are returned.
    """
    live = [p for p in params if p is not None]
    direct = all(getattr(p, '_mg_direct_grad', False) and p.grad is not None and p.grad.is_contiguous() for p in live)
    if direct:
        base = live[0].grad.data_ptr()
        for p, off in zip(live, offsets):
            if p.grad.data_ptr() != base + 4 * off or p.grad.dtype != torch.float32:
                direct = False
                break
    if direct:
        seg = live[0].grad.reshape(-1).as_strided((flat.numel(),), (1,))
        if grad_loss is None:
            seg.add_(flat)
        else:
            seg.addcmul_(flat, grad_loss)
        return [None] * len(params)
    out, j = [], 0
    for p in params:
        if p is None:
            out.append(None)
            continue
        g = flat[offsets[j]:offsets[j] + p.numel()].view(p.shape)
        out.append(g if grad_loss is None else g * grad_loss)
        j += 1
    return out


def _deliver_early(params, flat, offsets, grad_loss):
    """With an early-gradients hook installed and the optimiser's flat buffer as target: deliver everything but the first layer's
    (weight, bias) now and fire the hook.  Returns True if it did (``_deliver_rest`` then hands over the first layer only)."""
    if _EARLY_GRADS_HOOK is None or len(params) <= 2 or any(p is None for p in params):
        return False
    tail = _deliver_param_grads(params[2:], flat[offsets[2]:], [o - offsets[2] for o in offsets[2:]], grad_loss)
    if any(t is not None for t in tail):
        raise RuntimeError('early gradient exchange needs the parameters in morgana_amd.optim.Adam\'s flat buffer')
    _EARLY_GRADS_HOOK(params)
    return True


def _deliver_rest(params, flat, offsets, grad_loss, early):
    if not early:
        return _deliver_param_grads(params, flat, offsets, grad_loss)
    head = _deliver_param_grads(params[:2], flat[:offsets[2]], offsets[:2], grad_loss)
    return list(head) + [None] * (len(params) - 2)


def _grad_mode(params, grad_loss):
    """How a backward pass hands over its parameter gradients: 'defer' (split-M slabs left for the optimiser's update kernel),
    'direct' (accumulated straight into the optimiser's flat gradient) or 'temp' (ordinary tensors for autograd).  The first two need
    the parameters' .grad to be consecutive views of morgana_amd.optim.Adam's flat buffer and the incoming loss gradient to be the
    cached unit of ``backward`` (no scaling to apply).  Returns (mode, optimiser)."""
    if any(p is None for p in params) or grad_loss is None or not any(grad_loss.data_ptr() == one.data_ptr() for one in _ONES.values()):
        return 'temp', None
    if not all(getattr(p, '_mg_direct_grad', False) and p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32
               for p in params):
        return 'temp', None
    base, off = params[0].grad.data_ptr(), 0
    for p in params:
        if p.grad.data_ptr() != base + 4 * off:
            return 'temp', None
        off += p.numel()
    opt = getattr(params[0], '_mg_optimizer', None)
    if opt is not None and opt.defers_slabs():
        return 'defer', opt
    return 'direct', opt


def _grads_adjacent(w_param, b_param):
    """Is b.grad stored right behind W.grad (the optimiser's flat layout)?  Then one slab reduce finishes both."""
    return (w_param.grad is not None and b_param.grad is not None and w_param.grad.is_contiguous() and
            b_param.grad.data_ptr() == w_param.grad.data_ptr() + 4 * w_param.numel())


def _slabs_into(mode, opt, w_param, b_param, slab, n_slabs, stride, n, k):
    """Split-M slabs of one layer's dW | db to the optimiser: left for the update kernel to sum ('defer'), or summed now into .grad
    (accumulating) by one reduce launch - a data-parallel rank needs the finished gradient for its all-reduce."""
    w_param._mg_slab_buf = slab                          # one buffer per layer, reused every step (and by every graph replay)
    if mode == 'defer':
        opt.defer_slabs(w_param, n * k + n, slab, n_slabs, stride)
    else:
        ops.slab_reduce(slab, n_slabs, stride, n * k + n, w_param.grad.reshape(-1).as_strided((n * k + n,), (1,)), accumulate=True)


def _wgrad_into(mode, opt, w_param, b_param, g, a_in, rows, m, n, k):
    """dW, db of one layer into the optimiser: as slabs where the shape has them (_slabs_into), else reduced into .grad (accumulating)
    by the weight-gradient entry point's own two reduce launches."""
    if ops.wgrad_slabs_ok(m, n, k, a_in.shape[1], g.shape[1]) and (mode == 'defer' or _grads_adjacent(w_param, b_param)):
        slab, n_slabs, stride = ops.linear_wgrad_slabs_bf16(g, a_in, rows, m, n, k, slab=getattr(w_param, '_mg_slab_buf', None))
        _slabs_into(mode, opt, w_param, b_param, slab, n_slabs, stride, n, k)
    elif ops.wgrad_wide_ok(m, n, k, a_in.shape[1], g.shape[1]) and _grads_adjacent(w_param, b_param):
        # frame-rate row counts: 192-256 slabs, which a reduce launch of its own sums as fast as the update kernel would - one launch
        # for dW | db instead of the entry point's two
        slab, n_slabs, stride = ops.linear_wgrad_slabs_bf16(g, a_in, rows, m, n, k, slab=getattr(w_param, '_mg_slab_buf', None))
        _slabs_into('direct', opt, w_param, b_param, slab, n_slabs, stride, n, k)
    else:
        ops.linear_wgrad_bf16(g, a_in, rows, m, n, k, out_w=w_param.grad, out_b=b_param.grad, accumulate=True)


class LinearStackMSEFn(torch.autograd.Function):
    """bf16 Linear/Sigmoid stack ending in ... -> 128 -> 32 -> 1 TOGETHER with the masked MSE (losses.py:29-51).

    forward(ctx, acts, x2d, rows, target (B,T,1), seq_len, *params) -> (loss, pred (B,T,1)).
    Layers up to the 128-wide one run as in LinearStackFn; the last two layers, the loss and their whole backward are ONE
    kernel (mg_f0_tail_bf16) that leaves dL/d(pre-activation of the 128-wide layer) for the remaining backward; when the 128-wide
    layer is 512 -> 128 + sigmoid it runs inside that kernel too (mg_f0_l2tail_bf16) and its output is never written.
    `pred` is returned for reporting only (non-differentiable): the loss is the one consumer of the prediction.
    """

    @staticmethod
    def forward(ctx, acts, x2d, rows, target, seq_len, *params):
        acts_in = acts
        maps = table_bf16 = pending = order = None
        if len(acts) in (2, 3, 4) and not isinstance(acts[0], int):  # (acts, maps[, table[, phone_rate]]): phone-rate maps that came with
            acts, maps = acts[0], acts[1]                            # the frame map, the loader's bf16 copy of the phone table
            table_bf16 = acts_in[2] if len(acts_in) >= 3 else None   # (data.add_bf16_table), and the caller's order of operations
            order = acts_in[3] if len(acts_in) == 4 else None        # (utils.upsample_to_repetitions(phone_rate=); None = default)
            if maps is not None and not isinstance(maps, tuple):
                # an utils.UpsampledSequence whose frame map no kernel has built yet (rows is None): the phone-rate step builds it
                # in the launch of its own front (ops.phone_front); every other path asks the sequence for it now
                pending, maps = maps, None
        n_layers = len(acts)
        weights = [params[2 * i] for i in range(n_layers)]
        biases = [params[2 * i + 1] for i in range(n_layers)]
        x2d = ops._require(x2d, torch.float32, 'input')
        target = ops._require(target, torch.float32, 'targets')
        b, t = target.shape[0], target.shape[1]
        m = b * t
        n_layers_lead = n_layers - 2
        front = (pending is not None and pending.t_cap == t and pending.dur.shape[0] == b and n_layers_lead == 2 and
                 ops.phone_rate_table_ok(x2d.shape[0], m, weights[0].shape[0], weights[1].shape[0], acts[0], order) and
                 ops.phone_front_ok(b, pending.dur.shape[1], t, ops.PHONE_RATE_EXTRA))
        if pending is not None and not front:
            rows, maps = pending.rows.reshape(-1), pending.maps
        if (m if front else rows.numel() if rows is not None else x2d.shape[0]) != m:
            raise ValueError('prediction rows (%d) and target rows (%d) differ' % (
                rows.numel() if rows is not None else x2d.shape[0], m))
        lead = n_layers - 2
        # bf16 operands of the leading layers: copies that live on the parameters and are kept current by the optimiser's update
        # kernel (ops.param_shadows) - a training step launches no weight cast
        w_bf, w_t = ops.param_shadows(weights[:lead], want_t=tuple(range(1, lead)))
        # The 512 -> 128 sigmoid layer in front of the tail runs inside the tail kernel (csrc/l2tail_bf16.hip): its 128-wide output is
        # needed by nothing but the tail, so it is never written (frame rate: 68 + 43 us and 130 MB of traffic -> one pass over H1)
        l2tail = lead >= 2 and ops.l2tail_ok(weights[lead - 1], weights[lead], weights[lead + 1], acts[lead - 1])
        # Phone-rate step (csrc/phone_rate.hip).  The stack's input is upsample_to_repetitions(lab, dur): every phone row repeated.
        # Linear and Sigmoid commute with repeating rows, and the README model has no frame-level input, so EVERY layer's output is
        # constant over a phone's frames: the layers run once per phone row (plus extra zero rows = what padding frames gather),
        # the prediction is repeated instead of the input, and the masked MSE over a phone's frames reduces exactly to
        #     sum_f w_f (p - y_f)^2 = W (p - ybar)^2 + const        (W, ybar, const from the targets alone: mg_phone_target_stats)
        # whose gradient the fused tail forms per phone row.
        n_table = x2d.shape[0]
        phone_rate = front or (rows is not None and lead == 2
                               and ops.phone_rate_table_ok(n_table, m, weights[0].shape[0], weights[1].shape[0], acts[0], order))
        ctx.phone_rate = phone_rate
        if phone_rate:
            extra = ops.PHONE_RATE_EXTRA
            if not front:
                seg, rows = maps if maps is not None else ops.segment_bounds(rows, n_table, pad_row=n_table)
            if table_bf16 is not None and tuple(table_bf16.shape) == (n_table + extra, ops.pad_ld(x2d.shape[1])):
                a0 = table_bf16                                   # cast once when the batch was loaded, not once per step
            else:
                a0 = ops.cast_pad_bf16(x2d, extra_rows=extra)
            n_rows = a0.shape[0]
            hidden, a = [], a0
            for i in range(lead - (1 if l2tail else 0)):
                n, k = weights[i].shape
                if front and i == 0:
                    # the frame map, the per-phone loss statistics and this layer's GEMM read nothing of each other: one launch
                    rows_bt, rows_mapped, seg, ybar, weight, partials, a = ops.phone_front(
                        pending.dur, target.reshape(-1), seq_len, t, extra, linear=(a, k, w_bf[i], biases[i], n, acts[i]))
                    rows = rows_mapped.reshape(-1)
                    pending.adopt(rows_bt, (seg, rows))
                else:
                    a = ops.linear_fwd_bf16(a, None, n_rows, k, w_bf[i], biases[i], n, acts[i])
                hidden.append(a)
            sizes = [p.numel() for p in params]
            offsets = [0]
            for sz in sizes[:-1]:
                offsets.append(offsets[-1] + sz)
            flat = torch.empty(sum(sizes) + 1, dtype=torch.float32, device=x2d.device)
            if not front:
                ybar, weight, partials = ops.phone_target_stats(target.reshape(-1), rows, seg, seq_len, b, t, n_table, extra)
            ctx.deferred_tail = None
            if l2tail and os.environ.get('MORGANA_EXPAND_REDUCE', '1') != '0':
                # the tail's slab reduce rides in the launch that repeats the prediction (one node less, the same sums) - and inside a
                # step captured whole into a HIP graph (DEFER_TAIL: graphs.GraphedTrainStep) both are left to a later launch of the
                # step (backward decides which: the update launch's first blocks, or riders behind its first grid): pred / loss / the
                # tail's gradients are then complete when the graph's update has run, which is all a replay can observe
                defer = DEFER_TAIL and any(ctx.needs_input_grad[5:])        # (a backward pass will come: graphs.GraphedTrainStep)
                res = ops.f0_l2tail_rows_expand(hidden[-1], w_bf[lead - 1], biases[lead - 1], weights[lead], biases[lead],
                                                weights[lead + 1], biases[lead + 1], ybar, weight, flat[offsets[2 * lead]:],
                                                rows, (partials, n_table, extra), defer=defer)
                pred, loss, dz2 = res[:3]
                ctx.deferred_tail = res[3] if defer else None
                pred = pred.view(b, t, 1)
            else:
                if l2tail:
                    pred_rows, loss, dz2 = ops.f0_l2tail_rows(hidden[-1], w_bf[lead - 1], biases[lead - 1], weights[lead], biases[lead],
                                                              weights[lead + 1], biases[lead + 1], ybar, weight, flat[offsets[2 * lead]:])
                else:
                    pred_rows, loss, dz2 = ops.f0_tail_rows(hidden[-1], weights[lead], biases[lead], weights[lead + 1], biases[lead + 1],
                                                            ybar, weight, flat[offsets[2 * lead]:])
                pred = ops.expand_column(pred_rows, rows, loss_const=(partials, n_table, extra, loss)).view(b, t, 1)
            ctx.acts, ctx.m, ctx.lead = acts, m, lead
            ctx.dims = [(w.shape[0], w.shape[1]) for w in weights]
            ctx.offsets = offsets
            ctx.params = list(params)
            ctx.save_for_backward(a0, rows, dz2, flat, *hidden[:lead - 1], *[wt for wt in w_t if wt is not None])
            ctx.mark_non_differentiable(pred)
            ctx.set_materialize_grads(False)
            return loss, pred
        if table_bf16 is not None and table_bf16.shape[0] >= x2d.shape[0] and table_bf16.shape[1] == ops.pad_ld(x2d.shape[1]):
            a = table_bf16                                        # rows beyond the phone rows are never indexed by the gather
        else:
            a = ops.cast_pad_bf16(x2d)
        a0, r = a, rows
        hidden = []
        for i in range(lead - (1 if l2tail else 0)):
            n, k = weights[i].shape
            a = ops.linear_fwd_bf16(a, r, m, k, w_bf[i], biases[i], n, acts[i], rows_runs=True)      # rows: the upsample frame map
            r = None
            hidden.append(a)
        sizes = [p.numel() for p in params]
        offsets = [0]
        for sz in sizes[:-1]:
            offsets.append(offsets[-1] + sz)
        flat = torch.empty(sum(sizes) + 1, dtype=torch.float32, device=x2d.device)    # + 1: the loss (see ops.f0_tail)
        tail_off = offsets[2 * lead]
        ctx.deferred_tail = None
        if l2tail:
            # inside a step captured whole (DEFER_TAIL) the tail's reduce launch is left out: the update kernel sums the slabs (a source
            # of its plan) and forms the loss (optim.Adam.defer_tail) - see the phone-rate branch above
            defer = DEFER_TAIL and any(ctx.needs_input_grad[5:])
            res = ops.f0_l2tail(hidden[-1], w_bf[lead - 1], biases[lead - 1], weights[lead], biases[lead], weights[lead + 1],
                                biases[lead + 1], target.reshape(-1), seq_len, b, t, flat[tail_off:], defer=defer)
            pred, loss, dz2 = res[:3]
            ctx.deferred_tail = res[3] if defer else None
        else:
            pred, loss, dz2 = ops.f0_tail(hidden[-1], weights[lead], biases[lead], weights[lead + 1], biases[lead + 1],
                                          target.reshape(-1), seq_len, b, t, flat[tail_off:])
        ctx.acts, ctx.m, ctx.lead = acts, m, lead
        ctx.dims = [(w.shape[0], w.shape[1]) for w in weights]
        ctx.offsets = offsets
        ctx.params = list(params)
        ctx.save_for_backward(a0, rows, dz2, flat, *hidden[:lead - 1], *[wt for wt in w_t if wt is not None])
        pred = pred.view(b, t, 1)
        ctx.mark_non_differentiable(pred)
        ctx.set_materialize_grads(False)
        return loss, pred

    @staticmethod
    def backward(ctx, grad_loss, grad_pred):
        saved = ctx.saved_tensors
        a0, rows, g, flat = saved[0], saved[1], saved[2], saved[3]
        lead, m = ctx.lead, ctx.m
        hidden = list(saved[4:4 + lead - 1])                  # outputs of layers 0 .. lead-2
        w_t = [None] + list(saved[4 + lead - 1:])             # transposed bf16 weights of layers 1 .. lead-1
        def grad_slots(i):
            n_, k_ = ctx.dims[i]
            return (flat[ctx.offsets[2 * i]:ctx.offsets[2 * i] + n_ * k_].view(n_, k_),
                    flat[ctx.offsets[2 * i + 1]:ctx.offsets[2 * i + 1] + n_])

        mode, opt = _grad_mode(ctx.params, grad_loss)
        tail = getattr(ctx, 'deferred_tail', None)            # forward left the repeated prediction and the tail's slab sum to us
        ctx.deferred_tail = None
        if mode != 'temp':
            # The gradients go where the optimiser reads them, without a temporary and an add in between: either straight into its flat
            # buffer (the reduce launches accumulate there) or - one rank, fused loop - as split-M slabs the update kernel sums itself.
            params = ctx.params
            tail_off = ctx.offsets[2 * lead]
            tail_count = flat.numel() - 1 - tail_off
            m_rows = hidden[0].shape[0] if ctx.phone_rate else m
            r0 = None if ctx.phone_rate else rows
            top = lead - 1                                            # the 128-wide layer: its dZ came out of the fused tail
            n, k = ctx.dims[top]
            g_below = None
            pair = (ctx.phone_rate and top > 0 and ctx.acts[top - 1] == ops.ACT_SIGMOID and
                    ops.wgrad_slabs_ok(m_rows, n, k, hidden[top - 1].shape[1], g.shape[1]) and
                    (mode == 'defer' or _grads_adjacent(params[2 * top], params[2 * top + 1])))
            tail_slabs = None
            if tail is not None and tail.get('rows') is None:         # frame-rate tail: only its reduce launch was left out
                if mode == 'defer':
                    tail_slabs = (tail['ws'].view(torch.float32), tail['n_slabs'], tail['stride'])
                    opt.defer_tail(params[2 * lead], tail)            # the update launch forms the loss
                else:
                    ops.finish_deferred_tail(tail)
                tail = None
            if tail is not None and not pair:                         # no launch to ride in: the two jobs run now, as their own launch
                ops.finish_deferred_tail(tail)
                tail = None

            def hand_over_tail_grads():
                if tail_slabs is not None:                            # the update kernel sums the tail's slabs itself
                    opt.defer_slabs(params[2 * lead], tail_count, tail_slabs[0], tail_slabs[1], tail_slabs[2])
                elif mode == 'defer':
                    opt.defer_slabs(params[2 * lead], tail_count, flat[tail_off:tail_off + tail_count], 1, tail_count)
                else:
                    params[2 * lead].grad.reshape(-1).as_strided((tail_count,), (1,)).add_(flat[tail_off:tail_off + tail_count])
            # Frame rate, Linear + Sigmoid -> Linear(. -> 128) at the bottom of the stack: the second layer's weight gradient rides in
            # the fused backward of the first (ops.linear_bwd_fused2_slabs_bf16: the kernel stages dZ2 and H1 anyway), and the
            # stand-alone launch that re-read H1 from HBM goes.  Not beside an early gradient exchange: that one promises the second
            # layer's gradient final BEFORE the fused kernel starts.
            fuse2 = (not ctx.phone_rate and top == 1 and rows is not None and ctx.acts[0] == ops.ACT_SIGMOID and _EARLY_GRADS_HOOK is None and
                     os.environ.get('MORGANA_FUSE_WGRAD2', '1') != '0' and ops.can_fuse_bwd(m, n, k, ctx.dims[0][1], a0.shape[1]) and
                     (mode == 'defer' or (_grads_adjacent(params[0], params[1]) and _grads_adjacent(params[2], params[3]))))
            if fuse2:
                n0_, k0_ = ctx.dims[0]
                slab, n_slabs, (off1, st1, cnt1), (off2, st2, cnt2) = ops.linear_bwd_fused2_slabs_bf16(
                    g, w_t[1], hidden[0], a0, rows, m, n0_, k0_, slab=getattr(params[0], '_mg_fused_slab_buf', None))
                params[0]._mg_fused_slab_buf = slab
                floats = slab.view(torch.float32)
                for first, off, st, cnt in ((params[0], off1, st1, cnt1), (params[2], off2, st2, cnt2)):
                    if mode == 'defer':                               # the update kernel sums the slabs
                        opt.defer_slabs(first, cnt, floats[off:], n_slabs, st)
                    else:
                        ops.slab_reduce(floats[off:], n_slabs, st, cnt, first.grad.reshape(-1).as_strided((cnt,), (1,)), accumulate=True)
                hand_over_tail_grads()
                return (None, None, None, None, None) + (None,) * len(params)
            if pair and tail is not None and mode == 'defer':
                # the update kernel sums the tail's slabs itself (one more source of its plan), and its launch's first blocks repeat
                # the prediction and form the loss (optim.Adam.defer_tail: mg_adam_tail) - measured against riders at the end of the
                # pair grid below, which is where these two jobs go when the gradients must be complete before the update (a
                # data-parallel rank: mode 'direct')
                tail_slabs = (tail['ws'].view(torch.float32), tail['n_slabs'], tail['stride'])
                if os.environ.get('MORGANA_TAIL_RIDERS', 'adam') == 'adam':
                    opt.defer_tail(params[2 * lead], tail)
                    tail = None
                else:
                    tail = dict(tail, loss_only=True)
            if pair:
                # this layer's weight gradient and the dgrad + sigmoid backward below it are independent and each fills part of the
                # chip: one grid for both (mg_linear_wgrad_dgrad_bf16)
                w_param = params[2 * top]
                slab, n_slabs, stride, g_below = ops.linear_wgrad_dgrad_bf16(g, hidden[top - 1], m_rows, n, k, w_t[top],
                                                                             slab=getattr(w_param, '_mg_slab_buf', None), tail=tail)
                tail = None
                _slabs_into(mode, opt, w_param, params[2 * top + 1], slab, n_slabs, stride, n, k)
            else:
                _wgrad_into(mode, opt, params[2 * top], params[2 * top + 1], g, hidden[top - 1] if top > 0 else a0,
                            None if top > 0 else r0, m_rows, n, k)
            hand_over_tail_grads()
            for i in range(top, 0, -1):
                n, k = ctx.dims[i]
                if i == 1 and _EARLY_GRADS_HOOK is not None:
                    _EARLY_GRADS_HOOK(params)                         # everything but the first layer's gradient is final
                if (not ctx.phone_rate and i == 1 and ctx.acts[0] == ops.ACT_SIGMOID and
                        ops.can_fuse_bwd(m, n, k, ctx.dims[0][1], a0.shape[1])):
                    n0_, k0_ = ctx.dims[0]
                    if mode == 'defer':                               # its split-M slabs stay for the update kernel to sum
                        slab, n_slabs, stride = ops.linear_bwd_fused_slabs_bf16(g, w_t[1], hidden[0], a0, rows, m, n0_, k0_,
                                                                                slab=getattr(params[0], '_mg_fused_slab_buf', None))
                        params[0]._mg_fused_slab_buf = slab
                        opt.defer_slabs(params[0], n0_ * k0_ + n0_, slab, n_slabs, stride)
                    else:
                        ops.linear_bwd_fused_bf16(g, w_t[1], hidden[0], a0, rows, m, n0_, k0_, out_w=params[0].grad, out_b=params[1].grad,
                                                  accumulate=True)
                    break
                if i == top and g_below is not None:
                    g = g_below
                else:
                    h = hidden[i - 1] if ctx.acts[i - 1] == ops.ACT_SIGMOID else None
                    g = ops.linear_dgrad_bf16(g, m_rows, n, w_t[i], k, h)
                n_, k_ = ctx.dims[i - 1]
                _wgrad_into(mode, opt, params[2 * (i - 1)], params[2 * (i - 1) + 1], g, hidden[i - 2] if i > 1 else a0,
                            None if i > 1 else r0, m_rows, n_, k_)
            return (None, None, None, None, None) + (None,) * len(params)

        if tail is not None:                                          # gradients as tensors for autograd: nothing to ride in
            ops.finish_deferred_tail(tail)
            tail = None
        if ctx.phone_rate:
            # g is dL/dZ_1 per TABLE row already (the tail ran on phone rows with the frames' summed loss weights), so the remaining
            # backward is the ordinary chain on R + extra rows:  dW_1 = g^T H_table,  dZ_0 = (g W_1) * H (1 - H),  dW_0 = dZ_0^T X
            (n1, k1), (n0, k0) = ctx.dims[1], ctx.dims[0]
            table = hidden[0]
            ow, ob = grad_slots(1)
            ops.linear_wgrad_bf16(g, table, None, table.shape[0], n1, k1, out_w=ow, out_b=ob)
            early = _deliver_early(ctx.params, flat[:flat.numel() - 1], ctx.offsets, grad_loss)
            dz0 = ops.linear_dgrad_bf16(g, table.shape[0], n1, w_t[1], k1, table)
            ow, ob = grad_slots(0)
            ops.linear_wgrad_bf16(dz0, a0, None, table.shape[0], n0, k0, out_w=ow, out_b=ob)
            grads = _deliver_rest(ctx.params, flat[:flat.numel() - 1], ctx.offsets, grad_loss, early)
            return (None, None, None, None, None) + tuple(grads)
        early = False
        if (lead == 2 and rows is not None and ctx.acts[0] == ops.ACT_SIGMOID and _EARLY_GRADS_HOOK is None and
                os.environ.get('MORGANA_FUSE_WGRAD2', '1') != '0' and ops.can_fuse_bwd(m, ctx.dims[1][0], ctx.dims[1][1], ctx.dims[0][1], a0.shape[1])):
            # the same one-launch backward as the optimiser-bound modes above (both layers' slabs, then the ordered reduces): whichever
            # way the gradients are handed over, they are the same bits
            n0_, k0_ = ctx.dims[0]
            slab, n_slabs, (off1, st1, cnt1), (off2, st2, cnt2) = ops.linear_bwd_fused2_slabs_bf16(g, w_t[1], hidden[0], a0, rows, m, n0_, k0_)
            floats = slab.view(torch.float32)
            ops.slab_reduce(floats[off1:], n_slabs, st1, cnt1, flat[ctx.offsets[0]:ctx.offsets[0] + cnt1])
            ops.slab_reduce(floats[off2:], n_slabs, st2, cnt2, flat[ctx.offsets[2]:ctx.offsets[2] + cnt2])
            grads = _deliver_rest(ctx.params, flat[:flat.numel() - 1], ctx.offsets, grad_loss, False)
            return (None, None, None, None, None) + tuple(grads)
        for i in range(lead - 1, -1, -1):
            n, k = ctx.dims[i]
            a_in, r = (a0, rows) if i == 0 else (hidden[i - 1], None)
            ow, ob = grad_slots(i)
            ops.linear_wgrad_bf16(g, a_in, r, m, n, k, out_w=ow, out_b=ob)
            if i == 1:
                early = _deliver_early(ctx.params, flat[:flat.numel() - 1], ctx.offsets, grad_loss)
            if i == 1 and ctx.acts[0] == ops.ACT_SIGMOID and ops.can_fuse_bwd(m, n, k, ctx.dims[0][1], a0.shape[1]):
                # layer 0's dW, db straight from this layer's dZ: dZ_0 = (dZ_1 W_1) * H_0 (1 - H_0) stays on chip
                n0_, k0_ = ctx.dims[0]
                ops.linear_bwd_fused_bf16(g, w_t[1], hidden[0], a0, rows, m, n0_, k0_,
                                          out_w=flat[ctx.offsets[0]:ctx.offsets[0] + n0_ * k0_].view(n0_, k0_),
                                          out_b=flat[ctx.offsets[1]:ctx.offsets[1] + n0_])
                break
            if i > 0:
                h = hidden[i - 1] if ctx.acts[i - 1] == ops.ACT_SIGMOID else None
                g = ops.linear_dgrad_bf16(g, m, n, w_t[i], k, h)
        grads = _deliver_rest(ctx.params, flat[:flat.numel() - 1], ctx.offsets, grad_loss, early)
        return (None, None, None, None, None) + tuple(grads)


class F0StackX3Fn(torch.autograd.Function):
    """The README F0Model's stack ``Linear(k, 512) Sigmoid Linear(512, 128) Sigmoid Linear(128, 32) Sigmoid Linear(32, 1)``
    (README.rst:65-73) on an ``upsample_to_repetitions`` input (morgana/utils.py:175-228) TOGETHER with ``losses.mse``
    (morgana/losses.py:29-51), precision mode 'bf16x3', at phone rate - the parity-grade counterpart of ``LinearStackMSEFn``'s
    phone-rate step: SIX launches inside a captured step (five kernels + the update) where the generic 'bf16x3' path takes 25.

    Every bf16 operand is a [hi | lo] PAIR of planes (ops.split_pair): the phone table's pair comes from the loader, the weights'
    pairs are kept current by the optimiser's update kernel (ops.pair_shadows), H1 / dZ2 / dZ1 leave the kernels that produce them
    already split, and the tile programs run each contraction three times over the plane pairs (csrc/gemm_bf16_big.hip).  The
    128 -> 32 -> 1 tail, the loss and their backward stay exact fp32 (csrc/tail_f32.hip); bias gradients are column sums of the
    fp32 gradients, taken by the kernels that produce them.

    forward(ctx, info, x2d (B*P, k) f32, target (B, T, 1), seq_len, w1, b1, w2, b2, w3, b3, w4, b4) -> (loss, pred (B, T, 1));
    ``info`` = (the ``utils.UpsampledSequence`` that owns the frame map, the loader's pair table or None).  ``pred`` is returned for
    reporting only (non-differentiable)."""

    @staticmethod
    def forward(ctx, info, x2d, target, seq_len, *params):
        holder, table = info
        w1, b1, w2, b2, w3, b3, w4, b4 = params
        x2d = ops._require(x2d, torch.float32, 'input')
        target = ops._require(target, torch.float32, 'targets')
        b, t = target.shape[0], target.shape[1]
        m, n_table, extra = b * t, x2d.shape[0], ops.PHONE_RATE_EXTRA
        (n1, k0), (n2, k1) = w1.shape, w2.shape
        if table is not None and tuple(table.shape) == (n_table + extra, 2 * ops.pad_ld(k0)) and table.dtype == torch.bfloat16:
            a0 = table                                         # split once when the batch was loaded, not once per step
        else:
            a0 = ops.split_pair(x2d, extra_rows=extra)
        n_rows = a0.shape[0]
        (w1p, w2p), (_, w2tp) = ops.pair_shadows([w1, w2], want_t=(1,))
        lin1 = (a0, k0, w1p, b1, n1, ops.ACT_SIGMOID)
        if holder.pending() and holder.t_cap == t and holder.dur.shape[0] == b and ops.phone_front_ok(b, holder.dur.shape[1], t, extra):
            # the frame map, the per-phone loss statistics and the first layer's GEMM read nothing of each other: one launch
            rows_bt, rows_mapped, seg, ybar, weight, partials, h1 = ops.phone_front_x3((holder.dur, target.reshape(-1), seq_len, t, extra), lin1)
            rows = rows_mapped.reshape(-1)
            holder.adopt(rows_bt, (seg, rows))
        else:
            seg, rows = holder.phone_maps()
            ybar, weight, partials = ops.phone_target_stats(target.reshape(-1), holder.rows.reshape(-1), seg, seq_len, b, t, n_table, extra)
            (h1,) = ops.phone_front_x3(None, lin1)
        if rows.numel() != m:
            raise ValueError('prediction rows (%d) and target rows (%d) differ' % (rows.numel(), m))
        # inside a step captured whole (DEFER_TAIL: graphs.GraphedTrainStep) the repeated prediction and the tail's slab sum are left to
        # the update launch's first blocks, as in LinearStackMSEFn
        defer = DEFER_TAIL and any(ctx.needs_input_grad[4:])
        if ops.X3_L2TAIL:
            # the 128-wide layer, the exact-fp32 tail, the loss and their backward in one launch: Z2 never leaves the chip
            pred_rows, dz2, ws, n_slabs, stride = ops.f0_l2tail_x3(h1, w2p, b2, w3, b3, w4, b4, ybar, weight, keep_slabs=defer)
        else:
            # (A/B: two launches - the layer's three passes as three sets of workgroups, the tail adds the partial sums)
            z2 = ops.linear_fwd_x3_f32(h1, k1, w2p, b2, n2, ops.ACT_NONE, parts=ops.X3_FWD_PARTS)
            pred_rows, dz2, ws, n_slabs, stride = ops.f0_tail_rows_x3(z2, w3, b3, w4, b4, ybar, weight, keep_slabs=defer)
        n = ops.F0_TAIL_X3_N
        flat_tail = torch.empty((ops.F0_TAIL_X3_SLAB,), dtype=torch.float32, device=x2d.device)
        pred = torch.empty((m,), dtype=torch.float32, device=x2d.device)
        tail = dict(pred_rows=pred_rows, rows=rows, out=pred, partials=partials, n_table_rows=int(n_table), extra=int(extra), ws=ws, n=n,
                    stride=stride, n_slabs=n_slabs, grads_out=flat_tail)
        if defer:
            ctx.deferred_tail = tail
        else:
            ops.finish_deferred_tail(tail)
            ctx.deferred_tail = None
        ctx.params = list(params)
        ctx.n_rows = n_rows
        ctx.save_for_backward(a0, h1, dz2, flat_tail, w2tp)
        pred = pred.view(b, t, 1)
        ctx.mark_non_differentiable(pred)
        ctx.set_materialize_grads(False)
        return flat_tail[n - 1], pred

    @staticmethod
    def backward(ctx, grad_loss, grad_pred):
        a0, h1, dz2, flat_tail, w2tp = ctx.saved_tensors
        params = ctx.params
        w1, b1, w2, b2, w3 = params[:5]
        (n1, k0), (n2, k1) = w1.shape, w2.shape
        m = ctx.n_rows
        mode, opt = _grad_mode(params, grad_loss)
        tail = getattr(ctx, 'deferred_tail', None)
        ctx.deferred_tail = None
        n_tail = ops.F0_TAIL_X3_N - 1                             # db2 | dW3 | db3 | dW4 | db4 (the loss behind them)
        tail_slabs = None
        if tail is not None:
            if mode == 'defer' and os.environ.get('MORGANA_TAIL_RIDERS', 'adam') == 'adam':
                # the update kernel sums the tail's slabs (one more source of its plan), its first blocks repeat the prediction and
                # form the loss (optim.Adam.defer_tail)
                tail_slabs = (tail['ws'].view(torch.float32), tail['n_slabs'], tail['stride'])
                opt.defer_tail(b2, tail)
            else:
                ops.finish_deferred_tail(tail)
        slab2, s2, st2, dz1, colsum, n_cs = ops.linear_wgrad_dgrad_x3(dz2, h1, m, n2, k1, w2tp, slab=getattr(w2, '_mg_slab_buf', None),
                                                                      colsum=getattr(b1, '_mg_slab_buf', None))
        w2._mg_slab_buf, b1._mg_slab_buf = slab2, colsum
        if mode == 'defer':
            if tail_slabs is not None:
                opt.defer_slabs(b2, n_tail, tail_slabs[0], tail_slabs[1], tail_slabs[2])
            else:
                opt.defer_slabs(b2, n_tail, flat_tail, 1, n_tail)
            opt.defer_slabs(w2, n2 * k1, slab2, s2, st2)
            opt.defer_slabs(b1, n1, colsum.view(torch.float32), n_cs, n1)
        elif mode == 'direct':
            b2.grad.reshape(-1).as_strided((n_tail,), (1,)).add_(flat_tail[:n_tail])
            ops.slab_reduce(slab2, s2, st2, n2 * k1, w2.grad.reshape(-1), accumulate=True)
            ops.slab_reduce(colsum, n_cs, n1, n1, b1.grad, accumulate=True)
            if _EARLY_GRADS_HOOK is not None:
                _EARLY_GRADS_HOOK(params)                         # everything but the first layer's weight gradient is final
        slab1, s1, st1 = ops.linear_wgrad_slabs_x3(dz1, a0, m, n1, k0, slab=getattr(w1, '_mg_slab_buf', None))
        w1._mg_slab_buf = slab1
        if mode == 'defer':
            opt.defer_slabs(w1, n1 * k0, slab1, s1, st1)
            return (None, None, None, None) + (None,) * len(params)
        if mode == 'direct':
            ops.slab_reduce(slab1, s1, st1, n1 * k0, w1.grad.reshape(-1), accumulate=True)
            return (None, None, None, None) + (None,) * len(params)
        # gradients as tensors for autograd
        sizes = [p.numel() for p in params]
        offsets = [0]
        for sz in sizes[:-1]:
            offsets.append(offsets[-1] + sz)
        flat = torch.empty((sum(sizes),), dtype=torch.float32, device=a0.device)
        ops.slab_reduce(slab1, s1, st1, n1 * k0, flat[offsets[0]:offsets[0] + n1 * k0])
        ops.slab_reduce(colsum, n_cs, n1, n1, flat[offsets[1]:offsets[1] + n1])
        ops.slab_reduce(slab2, s2, st2, n2 * k1, flat[offsets[2]:offsets[2] + n2 * k1])
        flat[offsets[3]:offsets[3] + n_tail].copy_(flat_tail[:n_tail])
        grads = _deliver_param_grads(params, flat, offsets, None if _is_unit_grad(grad_loss) else grad_loss)
        return (None, None, None, None) + tuple(grads)


_STATE_ROWS = {}


def state_rows(b, t, device, shift=0):
    """int32 row indices b (T+1) + t + shift, t < T, into a (B, T+1, H) state array viewed as (B (T+1), H): shift 0 = h_{t-1}
    (the recurrent operand of step t), shift 1 = h_t.  Cached per shape: five tiny launches per call otherwise."""
    key = (b, t, str(device), shift)
    rows = _STATE_ROWS.get(key)
    if rows is None:
        rows = (torch.arange(b, device=device, dtype=torch.int32)[:, None] * (t + 1) +
                torch.arange(t, device=device, dtype=torch.int32)[None, :] + shift).reshape(-1).contiguous()
        if len(_STATE_ROWS) > 64:
            _STATE_ROWS.clear()
        _STATE_ROWS[key] = rows
    return rows


# In bf16 precision the GRU recurrence runs its two per-step matmuls on bf16 operands (fp32 accumulate, fp32 cell arithmetic and
# states) when the hidden size allows it; set_recurrence_bf16(False) keeps the exact-fp32 MFMA recurrence under bf16 layers.
RECURRENCE_BF16 = os.environ.get('MORGANA_RECURRENCE', 'bf16') != 'fp32'


def set_recurrence_bf16(enabled):
    global RECURRENCE_BF16
    RECURRENCE_BF16 = bool(enabled)


def gru_shadow_ok(precision, b, t, hid):
    """Will GRUFn run the launch that can write a bf16 copy of its output (the persistent bf16 recurrence)?"""
    return precision == 'bf16' and RECURRENCE_BF16 and ops.gru_bf16_ok(hid) and ops.gru_persist_ok(b, t, hid)


class GRUFn(torch.autograd.Function):
    """One GRU layer (batch_first) restricted to seq_len[b] steps per item; returns (outputs, h_n).

    With ``rows`` (int32 (B, T), every entry >= 0) and ``seg`` the input is a TABLE ``x`` (rows, I) whose row ``rows[b, t]`` is the
    input of frame (b, t) - the phone-rate form (csrc/phone_rate.hip): the input projection runs once per table row and its rows
    are repeated (``mg_gather_rows_f32``); backward sums the gate gradients per table row (``mg_segment_sum``) before the
    weight-gradient and input-gradient GEMMs, which then run on table rows too."""

    @staticmethod
    def forward(ctx, precision, x, h0, seq_len, w_ih, w_hh, b_ih, b_hh, rows=None, seg=None, layout=None, out_bf=None):
        x = ops._require(x, torch.float32, 'inputs')
        ctx.layout = layout            # utils.FrameLayout of a ragged batch or None: backward's weight gradients skip the padded frames
        hid = w_hh.shape[1]
        if rows is not None:
            (b, t), i_dim = rows.shape, x.shape[1]
            x2, m_in = x, x.shape[0]
        else:
            b, t, i_dim = x.shape
            x2, m_in = x.view(b * t, i_dim), b * t
        if precision == 'fp32':
            xproj = ops.linear_fwd_f32(x2, None, m_in, w_ih, b_ih, ops.ACT_NONE)
            x_saved = x2
        else:
            x_saved = ops.cast_pad_bf16(x2)
            xproj = ops.linear_fwd_bf16(x_saved, None, m_in, i_dim, _w_plain([w_ih])[0], b_ih, 3 * hid,
                                        ops.ACT_NONE, out_f32=True)
            if xproj.shape[1] != 3 * hid:
                xproj = xproj[:, :3 * hid].contiguous()
        ctx.param_refs = (w_ih, w_hh, b_ih, b_hh)
        ctx.bf16_recurrence = precision == 'bf16' and RECURRENCE_BF16 and ops.gru_bf16_ok(hid)
        in_kernel = rows is not None and ctx.bf16_recurrence and ops.gru_persist_ok(b, t, hid)
        if rows is not None and not in_kernel:
            xproj = ops.gather_rows(xproj, rows.reshape(-1))          # the repetition, applied to the projected rows
        hstate_bf = None
        # out_bf (optional, from the caller): a (B, T, H) bf16 buffer the persistent recurrence fills with the bf16 copy of its output -
        # the operand of a Linear layer behind the wrapper (gru_shadow_ok says when the launch that writes it will run)
        if out_bf is not None and not gru_shadow_ok(precision, b, t, hid):
            raise ValueError('GRUFn: out_bf needs the persistent bf16 recurrence')
        # bf16 operands of W_hh (plain for this launch, transposed for the backward's): copies that live on the parameter, kept current
        # by the optimiser's update kernel (ops.param_shadows) - no cast launch per step and direction
        w_hh_bf = ops.param_shadows([w_hh], want_t=(0,))[0][0] if (ctx.bf16_recurrence and hid == ops.pad_ld(hid)) else None
        if in_kernel:
            # the persistent recurrence reads the projected table through the row map itself: no (B, T, 3H) copy of its rows
            out, hstate, saved, hstate_bf = ops.gru_fwd_bf16(xproj.contiguous(), w_hh.contiguous(), b_hh.contiguous(), seq_len, h0, b, t, hid,
                                                             xrows=rows, out_bf=out_bf, w_bf=w_hh_bf)
        elif ctx.bf16_recurrence:
            out, hstate, saved, hstate_bf = ops.gru_fwd_bf16(xproj.view(b, t, 3 * hid), w_hh.contiguous(), b_hh.contiguous(),
                                                             seq_len, h0, b, t, hid, out_bf=out_bf, w_bf=w_hh_bf)
        else:
            out, hstate, saved = ops.gru_fwd(xproj.view(b, t, 3 * hid), w_hh.contiguous(), b_hh.contiguous(), seq_len, h0,
                                             b, t, hid)
        ctx.precision = precision
        ctx.shape = (b, t, i_dim, hid)
        ctx.has_h0 = h0 is not None
        ctx.save_for_backward(x_saved, seq_len, w_ih, w_hh, hstate, saved, hstate_bf, rows, seg)
        return out, hstate[:, t].unsqueeze(0).contiguous()

    @staticmethod
    def backward(ctx, grad_out, grad_hn):
        x_saved, seq_len, w_ih, w_hh, hstate, saved, hstate_bf, rows, seg = ctx.saved_tensors
        b, t, i_dim, hid = ctx.shape
        g_out = grad_out.contiguous() if grad_out is not None else torch.zeros((b, t, hid), dtype=torch.float32,
                                                                                device=hstate.device)
        g_hn = grad_hn.reshape(b, hid).contiguous() if grad_hn is not None else None
        dhproj_bf = dxproj_bf = dxp2 = dhp2 = None
        m = b * t
        if ctx.bf16_recurrence and ops.gru_persist_ok(b, t, hid):
            # persistent recurrence: only the bf16 shadows of the gate gradients are written (the GEMMs below take bf16)
            wt_bf = ops.param_shadows([ctx.param_refs[1]], want_t=(0,))[1][0] if hid == ops.pad_ld(hid) else None
            dxproj_bf, dhproj_bf, dh0 = ops.gru_bwd_bf16(g_out, g_hn, hstate, saved, w_hh, seq_len, b, t, hid, shadows_only=True, wt_bf=wt_bf)
        elif ctx.bf16_recurrence:
            wt_bf = ops.param_shadows([ctx.param_refs[1]], want_t=(0,))[1][0] if hid == ops.pad_ld(hid) else None
            dxproj, dhproj, dh0, dhproj_bf = ops.gru_bwd_bf16(g_out, g_hn, hstate, saved, w_hh, seq_len, b, t, hid, wt_bf=wt_bf)
            dxp2, dhp2 = dxproj.view(m, 3 * hid), dhproj.view(m, 3 * hid)
        else:
            dxproj, dhproj, dh0 = ops.gru_bwd(g_out, g_hn, hstate, saved, w_hh, seq_len, b, t, hid)
            dxp2, dhp2 = dxproj.view(m, 3 * hid), dhproj.view(m, 3 * hid)
        # h_{t-1} rows of hstate (B, T+1, H): row b*(T+1) + t
        prev_rows = state_rows(b, t, hstate.device)
        hs2 = hstate.view(b * (t + 1), hid)
        need_x = ctx.needs_input_grad[1]
        dx = None
        m_in = m
        if rows is not None:
            m_in = x_saved.shape[0]                       # table rows: phone rows + the extra rows of the padding frames
            n_phone = m_in - ops.PHONE_RATE_EXTRA
        if ctx.precision == 'fp32':
            if rows is not None:
                dxp2 = ops.segment_sum(dxp2.contiguous(), rows.reshape(-1), seg, n_phone, 3 * hid)
            dw_ih, db_ih = ops.linear_wgrad_f32(dxp2, x_saved, None, 3 * hid, i_dim)
            dw_hh, db_hh = ops.linear_wgrad_f32(dhp2, hs2, prev_rows, 3 * hid, hid)
            if need_x:
                dx = ops.linear_dgrad_f32(dxp2, w_ih, None)
                dx = dx if rows is not None else dx.view(b, t, i_dim)
        else:
            # the bf16 recurrence already wrote the bf16 shadows of dhproj and of the states
            dhp_bf = dhproj_bf.view(m, 3 * hid) if dhproj_bf is not None else ops.cast_pad_bf16(dhp2)
            hs_bf = hstate_bf.view(b * (t + 1), hid) if hstate_bf is not None else ops.cast_pad_bf16(hs2)
            # Ragged batch with its layout at hand: both products visit the sum_b T_b valid frames only (the gate gradients of the
            # padded frames are zero rows; the reference's PackedSequence never holds them, utils.py:366-385)
            lay = ctx.layout if (ctx.layout is not None and (ctx.layout.b, ctx.layout.t) == (b, t)) else None
            p_ih, p_hh, pb_ih, pb_hh = ctx.param_refs
            # inside functional.backward the four gradients are added straight into the optimiser's flat gradient (no AccumulateGrad
            # launch per parameter); autograd gets None for them
            into = dict(accumulate=True) if _direct_params(p_ih, p_hh, pb_ih, pb_hh) else None
            kw_ih = dict(out_w=p_ih.grad, out_b=pb_ih.grad, **into) if into else {}
            kw_hh = dict(out_w=p_hh.grad, out_b=pb_hh.grad, **into) if into else {}

            def recurrent_wgrad():
                if lay is not None and ops.wgrad_rows_ok(lay.total, 3 * hid, hid, hs_bf.shape[1], dhp_bf.shape[1]):
                    return ops.linear_wgrad_rows_bf16(dhp_bf, lay.frame_rows(), hs_bf, lay.state_rows(), lay.total, 3 * hid, hid, **kw_hh)
                return ops.linear_wgrad_bf16(dhp_bf, hs_bf, prev_rows, m, 3 * hid, hid, **kw_hh)

            # The recurrent weight gradient (C4: 100 GFLOP, the longest launch behind the recurrence) feeds nothing but the update: it
            # goes beside the input-side chain below (segment sum -> input weight gradient -> dgrad -> the layers in front)
            recurrent_wgrad_out = None
            if into and _side_ok(hs_bf.device):
                if lay is not None:
                    lay.frame_rows(), lay.state_rows()     # built (once per layout) on the main stream, not inside the fork
                with _Beside(hs_bf.device, dhp_bf, hs_bf, prev_rows, lay):
                    recurrent_wgrad_out = recurrent_wgrad()
            dxp_bf = dxproj_bf.view(m, 3 * hid) if dxproj_bf is not None else ops.cast_pad_bf16(dxp2)
            if rows is not None:
                dxp_bf = ops.segment_sum(dxp_bf, rows.reshape(-1), seg, n_phone, 3 * hid)
            if lay is not None and rows is None and ops.wgrad_rows_ok(lay.total, 3 * hid, i_dim, x_saved.shape[1], dxp_bf.shape[1]):
                dw_ih, db_ih = ops.linear_wgrad_rows_bf16(dxp_bf, lay.frame_rows(), x_saved, None, lay.total, 3 * hid, i_dim, **kw_ih)
            else:
                dw_ih, db_ih = ops.linear_wgrad_bf16(dxp_bf, x_saved, None, m_in, 3 * hid, i_dim, **kw_ih)
            dw_hh, db_hh = recurrent_wgrad_out if recurrent_wgrad_out is not None else recurrent_wgrad()
            if into:
                dw_ih = dw_hh = db_ih = db_hh = None
            if need_x:
                dx = ops.linear_dgrad_bf16(dxp_bf, m_in, 3 * hid, _w_t(p_ih), i_dim, None,
                                           out_f32=True)
                if dx.shape[1] != i_dim:
                    dx = dx[:, :i_dim].contiguous()
                dx = dx if rows is not None else dx.view(b, t, i_dim)
        return (None, dx, dh0.view(1, b, hid) if ctx.has_h0 else None, None, dw_ih, dw_hh, db_ih, db_hh, None, None, None, None)


def lstm_persistent(precision, b, t, hid):
    """bf16 mode runs an LSTM layer as two persistent launches (csrc/lstm_persist.hip) when the shape is covered."""
    return precision == 'bf16' and RECURRENCE_BF16 and ops.lstm_persist_ok(b, t, hid)


GRU_STACK_WAVEFRONT = os.environ.get('MORGANA_GRU_STACK_WAVEFRONT', '1') != '0'


def gru_stack_small(b, t, hid, n_layers):
    """Consecutive small GRU layers run as one wavefront launch per direction (csrc/gru_small_stack.hip)."""
    return GRU_STACK_WAVEFRONT and n_layers >= 2 and ops.gru_stack_small_ok(b, t, hid, n_layers)


class GRUStackSmallFn(torch.autograd.Function):
    """L stacked single-layer GRUs with a small hidden size (the three RecurrentCuDNNWrapper(nn.GRU(., 64)) of models/f0_test_model.py:
    31-37): forward and backward are ONE launch each, a wavefront over (layer, time) of the workgroup-local small-GRU kernels
    (mg_gru_stack_fwd_small_f32 / _bwd_small_f32).  The upper layers' input projections and the gradients handed down between the
    layers are products inside the step (exact fp32 in both precision modes); layer 0's input projection and every weight gradient
    are GEMMs in the mode's precision, as in ``GRUFn``.

    forward(ctx, precision, x (B,T,I), seq_len, h0s ((L,B,H) or None), *params) with params = w_ih, w_hh, b_ih, b_hh per layer;
    returns (outputs of the top layer (B,T,H), h_n (L,B,H))."""

    @staticmethod
    def forward(ctx, precision, x, seq_len, h0s, *params):
        n_layers = len(params) // 4
        w_ih = [params[4 * l].contiguous() for l in range(n_layers)]
        w_hh = [params[4 * l + 1].contiguous() for l in range(n_layers)]
        b_ih = [params[4 * l + 2].contiguous() for l in range(n_layers)]
        b_hh = [params[4 * l + 3].contiguous() for l in range(n_layers)]
        x = ops._require(x, torch.float32, 'inputs')
        b, t, i_dim = x.shape
        hid = w_hh[0].shape[1]
        x2 = x.view(b * t, i_dim)
        if precision == 'fp32':
            xproj0 = ops.linear_fwd_f32(x2, None, b * t, w_ih[0], b_ih[0], ops.ACT_NONE)
            x_saved = x2
        else:
            x_saved = ops.cast_pad_bf16(x2)
            xproj0 = ops.linear_fwd_bf16(x_saved, None, b * t, i_dim, ops.cast_pad_bf16(w_ih[0]), b_ih[0], 3 * hid, ops.ACT_NONE,
                                         out_f32=True)
            if xproj0.shape[1] != 3 * hid:
                xproj0 = xproj0[:, :3 * hid].contiguous()
        # throughput mode: the cell's transcendentals on v_exp / v_rcp and the step's products on bf16 MFMAs, as the bf16-mode GRU-512 recurrence
        outs, hstates, saveds = ops.gru_stack_small_fwd(xproj0.view(b, t, 3 * hid), w_ih, w_hh, b_ih, b_hh, seq_len, h0s, b, t, hid,
                                                        fast=precision == 'bf16' and RECURRENCE_BF16)
        ctx.precision = precision
        ctx.shape = (b, t, i_dim, hid, n_layers)
        ctx.has_h0 = h0s is not None
        ctx.save_for_backward(x_saved, seq_len, *w_ih, *w_hh, *hstates, *saveds, *outs[:-1])
        hn = torch.stack([hstates[l][:, t] for l in range(n_layers)], dim=0)
        return outs[-1], hn

    @staticmethod
    def backward(ctx, grad_out, grad_hn):
        b, t, i_dim, hid, n_layers = ctx.shape
        sv = ctx.saved_tensors
        x_saved, seq_len = sv[0], sv[1]
        pos = 2
        w_ih = sv[pos:pos + n_layers]; pos += n_layers
        w_hh = sv[pos:pos + n_layers]; pos += n_layers
        hstate = sv[pos:pos + n_layers]; pos += n_layers
        saved = sv[pos:pos + n_layers]; pos += n_layers
        lower_out = sv[pos:pos + n_layers - 1]
        dev = x_saved.device
        m = b * t
        g_out = grad_out.contiguous() if grad_out is not None else torch.zeros((b, t, hid), dtype=torch.float32, device=dev)
        g_hn = [grad_hn[l] for l in range(n_layers)] if grad_hn is not None else None
        dxprojs, dhprojs, dh0 = ops.gru_stack_small_bwd(g_out, g_hn, hstate, saved, w_ih, w_hh, seq_len, b, t, hid,
                                                        fast=ctx.precision == 'bf16' and RECURRENCE_BF16)
        prev_rows = state_rows(b, t, dev)
        grads = [None] * (4 * n_layers)
        dx = None
        for l in range(n_layers):
            dxp2, dhp2 = dxprojs[l].view(m, 3 * hid), dhprojs[l].view(m, 3 * hid)
            hs2 = hstate[l].view(b * (t + 1), hid)
            k_in = i_dim if l == 0 else hid
            if ctx.precision == 'fp32':
                x_in = x_saved if l == 0 else lower_out[l - 1].view(m, hid)
                dw_ih, db_ih = ops.linear_wgrad_f32(dxp2, x_in, None, 3 * hid, k_in)
                dw_hh, db_hh = ops.linear_wgrad_f32(dhp2, hs2, prev_rows, 3 * hid, hid)
                if l == 0 and ctx.needs_input_grad[1]:
                    dx = ops.linear_dgrad_f32(dxp2, w_ih[0], None).view(b, t, i_dim)
            else:
                dxp_bf, dhp_bf = ops.cast_pad_bf16(dxp2), ops.cast_pad_bf16(dhp2)
                x_in = x_saved if l == 0 else ops.cast_pad_bf16(lower_out[l - 1].view(m, hid))
                dw_ih, db_ih = ops.linear_wgrad_bf16(dxp_bf, x_in, None, m, 3 * hid, k_in)
                dw_hh, db_hh = ops.linear_wgrad_bf16(dhp_bf, ops.cast_pad_bf16(hs2), prev_rows, m, 3 * hid, hid)
                if l == 0 and ctx.needs_input_grad[1]:
                    dx = ops.linear_dgrad_bf16(dxp_bf, m, 3 * hid, ops.cast_transpose_bf16(w_ih[0]), i_dim, None, out_f32=True)
                    if dx.shape[1] != i_dim:
                        dx = dx[:, :i_dim].contiguous()
                    dx = dx.view(b, t, i_dim)
            grads[4 * l:4 * l + 4] = [dw_ih, dw_hh, db_ih, db_hh]
        return (None, dx, None, dh0 if ctx.has_h0 else None, *grads)


def lstm_layerwise(precision, b, t, hid):
    """A stack of LSTM layers runs layer by layer (two persistent launches each) rather than as the time-skewed stack of per-step
    launches: bf16 mode with the persistent bf16-operand recurrence, or fp32 parity mode with the persistent fp32 one
    (csrc/lstm_persist_f32.hip: bit-identical to the per-step kernels, without the launch per time step)."""
    return lstm_persistent(precision, b, t, hid) or (precision == 'fp32' and ops.lstm_persist_f32_ok(b, t, hid))


class LSTMFn(torch.autograd.Function):
    """One LSTM layer (batch_first) restricted to seq_len[b] steps per item; returns (outputs, h_n, c_n)."""

    @staticmethod
    def forward(ctx, precision, x, h0, c0, seq_len, w_ih, w_hh, b_ih, b_hh):
        x = ops._require(x, torch.float32, 'inputs')
        b, t, i_dim = x.shape
        hid = w_hh.shape[1]
        x2 = x.view(b * t, i_dim)
        if precision == 'fp32':
            xproj = ops.linear_fwd_f32(x2, None, b * t, w_ih, b_ih, ops.ACT_NONE)
            x_saved = x2
        else:
            x_saved = ops.cast_pad_bf16(x2)
            xproj = ops.linear_fwd_bf16(x_saved, None, b * t, i_dim, ops.cast_pad_bf16(w_ih), b_ih, 4 * hid,
                                        ops.ACT_NONE, out_f32=True)
            if xproj.shape[1] != 4 * hid:
                xproj = xproj[:, :4 * hid].contiguous()
        ctx.persistent = lstm_persistent(precision, b, t, hid)
        hstate_bf = None
        if ctx.persistent:
            out, hstate, cstate, saved, hstate_bf = ops.lstm_fwd_bf16(xproj.view(b, t, 4 * hid), w_hh.contiguous(), b_hh.contiguous(),
                                                                      seq_len, h0, c0, b, t, hid)
        else:
            out, hstate, cstate, saved = ops.lstm_fwd(xproj.view(b, t, 4 * hid), w_hh.contiguous(), b_hh.contiguous(), seq_len,
                                                      h0, c0, b, t, hid)
        ctx.precision = precision
        ctx.shape = (b, t, i_dim, hid)
        ctx.has_h0, ctx.has_c0 = h0 is not None, c0 is not None
        ctx.save_for_backward(x_saved, seq_len, w_ih, w_hh, hstate, cstate, saved, hstate_bf)
        return out, hstate[:, t].unsqueeze(0).contiguous(), cstate[:, t].unsqueeze(0).contiguous()

    @staticmethod
    def backward(ctx, grad_out, grad_hn, grad_cn):
        x_saved, seq_len, w_ih, w_hh, hstate, cstate, saved, hstate_bf = ctx.saved_tensors
        b, t, i_dim, hid = ctx.shape
        dev = hstate.device
        g_out = grad_out.contiguous() if grad_out is not None else torch.zeros((b, t, hid), dtype=torch.float32, device=dev)
        g_hn = grad_hn.reshape(b, hid).contiguous() if grad_hn is not None else None
        g_cn = grad_cn.reshape(b, hid).contiguous() if grad_cn is not None else None
        dgates_bf = None
        if ctx.persistent:
            dgates, dh0, dc0, dgates_bf = ops.lstm_bwd_bf16(g_out, g_hn, g_cn, cstate, saved, w_hh, seq_len, b, t, hid, want_f32=False)
        else:
            dgates, dh0, dc0 = ops.lstm_bwd(g_out, g_hn, g_cn, cstate, saved, w_hh, seq_len, b, t, hid)
        m = b * t
        dg2 = dgates.view(m, 4 * hid) if dgates is not None else None
        prev_rows = state_rows(b, t, dev)
        hs2 = hstate.view(b * (t + 1), hid)
        need_x = ctx.needs_input_grad[1]
        dx = None
        if ctx.precision == 'fp32':
            dw_ih, db = ops.linear_wgrad_f32(dg2, x_saved, None, 4 * hid, i_dim)
            dw_hh, _ = ops.linear_wgrad_f32(dg2, hs2, prev_rows, 4 * hid, hid, want_bias=False)
            if need_x:
                dx = ops.linear_dgrad_f32(dg2, w_ih, None).view(b, t, i_dim)
        else:
            # the persistent recurrence already wrote the bf16 shadows of the gate gradients and of the states
            dg_bf = dgates_bf.view(m, 4 * hid) if dgates_bf is not None else ops.cast_pad_bf16(dg2)
            hs_bf = hstate_bf.view(b * (t + 1), hid) if hstate_bf is not None else ops.cast_pad_bf16(hs2)
            dw_ih, db = ops.linear_wgrad_bf16(dg_bf, x_saved, None, m, 4 * hid, i_dim)
            dw_hh, _ = ops.linear_wgrad_bf16(dg_bf, hs_bf, prev_rows, m, 4 * hid, hid, want_bias=False)
            if need_x:
                dx = ops.linear_dgrad_bf16(dg_bf, m, 4 * hid, ops.cast_transpose_bf16(w_ih), i_dim, None, out_f32=True)
                if dx.shape[1] != i_dim:
                    dx = dx[:, :i_dim].contiguous()
                dx = dx.view(b, t, i_dim)
        return (None, dx, dh0.view(1, b, hid) if ctx.has_h0 else None, dc0.view(1, b, hid) if ctx.has_c0 else None, None,
                dw_ih, dw_hh, db, db.clone())


# False (MORGANA_LSTM_STACK_BACKWARD=0): the stack's backward runs layer by layer (one persistent launch + three GEMMs per layer)
LSTM_STACK_BACKWARD = os.environ.get('MORGANA_LSTM_STACK_BACKWARD', '1') != '0'


class LSTMStackPersistFn(torch.autograd.Function):
    """L stacked single-layer LSTMs (the 8 x RecurrentCuDNNWrapper(nn.LSTM(512, 512)) of models/RNN_SPSS.py:36-37, or one
    multi-layer nn.LSTM) in bf16 mode: the forward of the whole stack is ONE persistent launch, a wavefront over (layer, time)
    (csrc/lstm_persist.hip, mg_lstm_pstack_fwd_bf16), and so is the backward (mg_lstm_pstack_bwd_bf16: the same wavefront run down in
    time and through the layers, the gradient a layer hands to the one below computed inside the step), followed by the weight-gradient
    GEMMs; where that launch does not cover the shape the layers run top-down, each as one persistent launch (mg_lstm_bwd_persist_bf16)
    between its weight-gradient and input-gradient GEMMs.  Same results as L chained LSTMFn calls up to the summation order of the
    in-step products.

    forward(ctx, x (B,T,I), seq_len, h0s, c0s ((L,B,H) or None), *params) with params = w_ih, w_hh, b_ih, b_hh per layer;
    returns (outputs of the top layer (B,T,H), h_n (L,B,H), c_n (L,B,H))."""

    @staticmethod
    def forward(ctx, x, seq_len, h0s, c0s, *params):
        n_layers = len(params) // 4
        w_ih = [params[4 * l] for l in range(n_layers)]
        w_hh = [params[4 * l + 1].contiguous() for l in range(n_layers)]
        b_ih = [params[4 * l + 2].contiguous() for l in range(n_layers)]
        b_hh = [params[4 * l + 3].contiguous() for l in range(n_layers)]
        x = ops._require(x, torch.float32, 'inputs')
        b, t, i_dim = x.shape
        hid = w_hh[0].shape[1]
        x_saved = ops.cast_pad_bf16(x.view(b * t, i_dim))
        xproj0 = ops.linear_fwd_bf16(x_saved, None, b * t, i_dim, _w_plain([w_ih[0]])[0], b_ih[0], 4 * hid, ops.ACT_NONE,
                                     out_f32=True)
        if xproj0.shape[1] != 4 * hid:
            xproj0 = xproj0[:, :4 * hid].contiguous()
        out, hstate, cstate, saved, hstate_bf = ops.lstm_pstack_fwd(xproj0.view(b, t, 4 * hid), w_ih, w_hh, b_ih, b_hh, seq_len, h0s, c0s,
                                                                    b, t, hid)
        ctx.param_refs = list(params)                    # the Parameter objects: their .grad views (direct gradients), their shadows
        ctx.shape = (b, t, i_dim, hid, n_layers)
        ctx.has_h0, ctx.has_c0 = h0s is not None, c0s is not None
        ctx.save_for_backward(x_saved, seq_len, *w_ih, *w_hh, *cstate, *saved, *hstate_bf)
        hn = torch.stack([hstate[l][:, t] for l in range(n_layers)], dim=0)
        cn = torch.stack([cstate[l][:, t] for l in range(n_layers)], dim=0)
        return out, hn, cn

    @staticmethod
    def backward(ctx, grad_out, grad_hn, grad_cn):
        b, t, i_dim, hid, n_layers = ctx.shape
        sv = ctx.saved_tensors
        x_saved, seq_len = sv[0], sv[1]
        pos = 2
        w_ih = sv[pos:pos + n_layers]; pos += n_layers
        w_hh = sv[pos:pos + n_layers]; pos += n_layers
        cstate = sv[pos:pos + n_layers]; pos += n_layers
        saved = sv[pos:pos + n_layers]; pos += n_layers
        hstate_bf = sv[pos:pos + n_layers]
        dev = x_saved.device
        m = b * t
        prev_rows = state_rows(b, t, dev)                                 # h_{t-1} of hstate (B, T+1, H): row b (T+1) + t
        next_rows = state_rows(b, t, dev, shift=1)                        # h_t: the input of the layer above at step t
        g_out = grad_out.contiguous() if grad_out is not None else torch.zeros((b, t, hid), dtype=torch.float32, device=dev)
        grads = [None] * (4 * n_layers)
        dx = None
        if LSTM_STACK_BACKWARD and ops.lstm_pstack_bwd_ok(b, t, hid, n_layers):
            # one wavefront launch for every layer's recurrence and the input gradients between the layers, then the weight gradients
            g_hn = [grad_hn[l] for l in range(n_layers)] if grad_hn is not None else None
            g_cn = [grad_cn[l] for l in range(n_layers)] if grad_cn is not None else None
            prm = ctx.param_refs
            _, dg_bfs, dh0, dc0 = ops.lstm_pstack_bwd(g_out, g_hn, g_cn, cstate, saved, [prm[4 * l] for l in range(n_layers)],
                                                      [prm[4 * l + 1] for l in range(n_layers)], seq_len, b, t, hid)
            # inside functional.backward the sixteen weight gradients and their bias sums are added straight into the optimiser's flat
            # gradient (both weight-gradient launches of a layer produce the bias sums: b_ih and b_hh receive the same values), and
            # autograd gets None: no AccumulateGrad launch per parameter, no clone of db
            direct = _direct_params(*prm)
            for l in range(n_layers):
                dg_bf = dg_bfs[l].view(m, 4 * hid)
                p_ih, p_hh, pb_ih, pb_hh = prm[4 * l:4 * l + 4]
                kw_ih = dict(out_w=p_ih.grad, out_b=pb_ih.grad, accumulate=True) if direct else {}
                kw_hh = dict(out_w=p_hh.grad, out_b=pb_hh.grad, accumulate=True) if direct else dict(want_bias=False)
                if l == 0:
                    dw_ih, db = ops.linear_wgrad_bf16(dg_bf, x_saved, None, m, 4 * hid, i_dim, **kw_ih)
                else:
                    dw_ih, db = ops.linear_wgrad_bf16(dg_bf, hstate_bf[l - 1].view(b * (t + 1), hid), next_rows, m, 4 * hid, hid, **kw_ih)
                dw_hh, _ = ops.linear_wgrad_bf16(dg_bf, hstate_bf[l].view(b * (t + 1), hid), prev_rows, m, 4 * hid, hid, **kw_hh)
                if not direct:
                    grads[4 * l:4 * l + 4] = [dw_ih, dw_hh, db, db.clone()]
            if ctx.needs_input_grad[0]:
                dx = ops.linear_dgrad_bf16(dg_bfs[0].view(m, 4 * hid), m, 4 * hid, _w_t(prm[0]), i_dim, None, out_f32=True)
                if dx.shape[1] != i_dim:
                    dx = dx[:, :i_dim].contiguous()
                dx = dx.view(b, t, i_dim)
            return (dx, None, dh0 if ctx.has_h0 else None, dc0 if ctx.has_c0 else None, *grads)
        dh0 = torch.empty((n_layers, b, hid), dtype=torch.float32, device=dev)
        dc0 = torch.empty((n_layers, b, hid), dtype=torch.float32, device=dev)
        for l in range(n_layers - 1, -1, -1):
            g_hn = grad_hn[l].reshape(b, hid).contiguous() if grad_hn is not None else None
            g_cn = grad_cn[l].reshape(b, hid).contiguous() if grad_cn is not None else None
            _, dh0_l, dc0_l, dg_bf = ops.lstm_bwd_bf16(g_out, g_hn, g_cn, cstate[l], saved[l], w_hh[l], seq_len, b, t, hid, want_f32=False)
            dh0[l], dc0[l] = dh0_l, dc0_l
            dg_bf = dg_bf.view(m, 4 * hid)
            if l == 0:
                dw_ih, db = ops.linear_wgrad_bf16(dg_bf, x_saved, None, m, 4 * hid, i_dim)
            else:
                dw_ih, db = ops.linear_wgrad_bf16(dg_bf, hstate_bf[l - 1].view(b * (t + 1), hid), next_rows, m, 4 * hid, hid)
            dw_hh, _ = ops.linear_wgrad_bf16(dg_bf, hstate_bf[l].view(b * (t + 1), hid), prev_rows, m, 4 * hid, hid, want_bias=False)
            grads[4 * l:4 * l + 4] = [dw_ih, dw_hh, db, db.clone()]
            if l > 0 or ctx.needs_input_grad[0]:
                k_in = hid if l > 0 else i_dim
                d_in = ops.linear_dgrad_bf16(dg_bf, m, 4 * hid, ops.cast_transpose_bf16(w_ih[l]), k_in, None, out_f32=True)
                if d_in.shape[1] != k_in:
                    d_in = d_in[:, :k_in].contiguous()
                if l > 0:
                    g_out = d_in.view(b, t, hid)
                else:
                    dx = d_in.view(b, t, i_dim)
        return (dx, None, dh0 if ctx.has_h0 else None, dc0 if ctx.has_c0 else None, *grads)


def lstm_stack_persistent(precision, b, t, hid, n_layers):
    return precision == 'bf16' and RECURRENCE_BF16 and n_layers >= 2 and ops.lstm_pstack_ok(b, t, hid, n_layers)


LSTM_STACK_LAG = 32       # steps a layer of a skewed stack runs behind the one below (and frames per projection chunk)


class LSTMStackFn(torch.autograd.Function):
    """L stacked single-layer LSTMs (the 8 x RecurrentCuDNNWrapper(nn.LSTM) of models/RNN_SPSS.py:36-37, or one multi-layer
    nn.LSTM) run skewed in time: layer l works ``lag`` steps behind layer l - 1 and one launch per step serves every layer
    (mg_lstm_stack_fwd_f32 / _bwd_f32), T + (L - 1) lag dependent launches instead of L T.  Every ``lag`` steps the chunk of
    outputs a layer has just finished goes through its upper neighbour's input projection (forward), the chunk of gate
    gradients through its own W_ih towards the layer below (backward).  Same arithmetic as L chained ``LSTMFn`` calls.

    forward(ctx, precision, lag, x (B,T,I), seq_len, h0s, c0s ((L,B,H) or None), *params) with params = w_ih, w_hh, b_ih, b_hh per
    layer; returns (outputs of the top layer (B,T,H), h_n (L,B,H), c_n (L,B,H))."""

    @staticmethod
    def _chunk_rows(b, t, lag, device):
        rows = []
        for c0 in range(0, t, lag):
            tt = torch.arange(c0, min(t, c0 + lag), device=device, dtype=torch.int32)
            rows.append((torch.arange(b, device=device, dtype=torch.int32)[:, None] * t + tt[None, :]).reshape(-1).contiguous())
        return rows

    @staticmethod
    def forward(ctx, precision, lag, x, seq_len, h0s, c0s, *params):
        from . import _lib
        n_layers = len(params) // 4
        w_ih = [params[4 * l] for l in range(n_layers)]
        w_hh = [params[4 * l + 1].contiguous() for l in range(n_layers)]
        b_ih = [params[4 * l + 2] for l in range(n_layers)]
        b_hh = [params[4 * l + 3].contiguous() for l in range(n_layers)]
        x = ops._require(x, torch.float32, 'inputs')
        b, t, i_dim = x.shape
        hid = w_hh[0].shape[1]
        dev = x.device
        x2 = x.view(b * t, i_dim)
        if precision == 'fp32':
            x_saved = x2
            xproj0 = ops.linear_fwd_f32(x2, None, b * t, w_ih[0], b_ih[0], ops.ACT_NONE)
            w_bf = None
        else:
            x_saved = ops.cast_pad_bf16(x2)
            w_bf = [ops.cast_pad_bf16(w) for w in w_ih]
            xproj0 = ops.linear_fwd_bf16(x_saved, None, b * t, i_dim, w_bf[0], b_ih[0], 4 * hid, ops.ACT_NONE, out_f32=True)
            if xproj0.shape[1] != 4 * hid:
                xproj0 = xproj0[:, :4 * hid].contiguous()
        hstate = [torch.empty((b, t + 1, hid), dtype=torch.float32, device=dev) for _ in range(n_layers)]
        cstate = [torch.empty((b, t + 1, hid), dtype=torch.float32, device=dev) for _ in range(n_layers)]
        for l in range(n_layers):
            for state, init in ((hstate[l], h0s), (cstate[l], c0s)):
                if init is None:
                    state[:, 0].zero_()
                else:
                    state[:, 0].copy_(init[l].reshape(b, hid))
        out = [torch.empty((b, t, hid), dtype=torch.float32, device=dev) for _ in range(n_layers)]
        saved = [torch.empty((b, t, 4 * hid), dtype=torch.float32, device=dev) for _ in range(n_layers)]
        rows = LSTMStackFn._chunk_rows(b, t, lag, dev)
        n_chunks = len(rows)
        descs = (_lib.LstmFwdLayer * n_layers)()
        for l in range(n_layers):
            d = descs[l]
            d.w_hh, d.b_hh = w_hh[l].data_ptr(), b_hh[l].data_ptr()
            d.hstate, d.cstate, d.out, d.saved = hstate[l].data_ptr(), cstate[l].data_ptr(), out[l].data_ptr(), saved[l].data_ptr()
        descs[0].xproj, descs[0].x_T, descs[0].x_t0 = xproj0.data_ptr(), t, 0
        chunk = [None] * n_layers                   # keeps the current projection chunk of each upper layer alive
        def project(l, c):
            m = rows[c].numel()
            below = out[l - 1].view(b * t, hid)
            if precision == 'fp32':
                return ops.linear_fwd_f32(below, rows[c], m, w_ih[l], b_ih[l], ops.ACT_NONE)
            a_bf = ops.gather_rows(below, rows[c], out_bf16=True)
            xp = ops.linear_fwd_bf16(a_bf, None, m, hid, w_bf[l], b_ih[l], 4 * hid, ops.ACT_NONE, out_f32=True)
            return xp if xp.shape[1] == 4 * hid else xp[:, :4 * hid].contiguous()

        for s0 in range(0, t + (n_layers - 1) * lag, lag):
            todo = [(l, s0 // lag - l) for l in range(1, n_layers) if 0 <= s0 // lag - l < n_chunks]
            for l, c in todo:
                xp = project(l, c)
                chunk[l] = xp
                descs[l].xproj, descs[l].x_T, descs[l].x_t0 = xp.data_ptr(), rows[c].numel() // b, c * lag
            for l in range(1, n_layers):
                if chunk[l] is None:
                    descs[l].xproj, descs[l].x_T, descs[l].x_t0 = xproj0.data_ptr(), 1, -1      # not reached yet: never read
            ops.lstm_stack_fwd(descs, n_layers, seq_len, b, t, hid, lag, s0, s0 + lag)
        ctx.precision, ctx.lag = precision, lag
        ctx.shape = (b, t, i_dim, hid, n_layers)
        ctx.has_h0, ctx.has_c0 = h0s is not None, c0s is not None
        ctx.save_for_backward(x_saved, seq_len, *w_ih, *w_hh, *hstate, *cstate, *saved, *out[:-1])
        hn = torch.stack([hstate[l][:, t] for l in range(n_layers)], dim=0)
        cn = torch.stack([cstate[l][:, t] for l in range(n_layers)], dim=0)
        return out[-1], hn, cn

    @staticmethod
    def backward(ctx, grad_out, grad_hn, grad_cn):
        from . import _lib
        b, t, i_dim, hid, n_layers = ctx.shape
        precision, lag = ctx.precision, ctx.lag
        sv = ctx.saved_tensors
        x_saved, seq_len = sv[0], sv[1]
        pos = 2
        w_ih = sv[pos:pos + n_layers]; pos += n_layers
        w_hh = sv[pos:pos + n_layers]; pos += n_layers
        hstate = sv[pos:pos + n_layers]; pos += n_layers
        cstate = sv[pos:pos + n_layers]; pos += n_layers
        saved = sv[pos:pos + n_layers]; pos += n_layers
        outs = sv[pos:pos + n_layers - 1]
        dev = hstate[0].device
        g_top = grad_out.contiguous() if grad_out is not None else None
        dgates = [torch.empty((b, t, 4 * hid), dtype=torch.float32, device=dev) for _ in range(n_layers)]
        carry_h = [(grad_hn[l].reshape(b, hid).clone() if grad_hn is not None else torch.zeros((b, hid), dtype=torch.float32, device=dev))
                   for l in range(n_layers)]
        carry_c = [(grad_cn[l].reshape(b, hid).clone() if grad_cn is not None else torch.zeros((b, hid), dtype=torch.float32, device=dev))
                   for l in range(n_layers)]
        dh0 = torch.empty((n_layers, b, hid), dtype=torch.float32, device=dev)
        dc0 = torch.empty((n_layers, b, hid), dtype=torch.float32, device=dev)
        rows = LSTMStackFn._chunk_rows(b, t, lag, dev)
        n_chunks = len(rows)
        w_t_bf = [ops.cast_transpose_bf16(w) for w in w_ih] if precision == 'bf16' else None
        descs = (_lib.LstmBwdLayer * n_layers)()
        for l in range(n_layers):
            d = descs[l]
            d.cstate, d.saved, d.w_hh, d.dgates = cstate[l].data_ptr(), saved[l].data_ptr(), w_hh[l].data_ptr(), dgates[l].data_ptr()
            d.carry_h, d.carry_c = carry_h[l].data_ptr(), carry_c[l].data_ptr()
            d.dh0, d.dc0 = dh0[l].data_ptr(), dc0[l].data_ptr()
            d.grad_out, d.g_T, d.g_t0 = None, 1, 0
        top = descs[n_layers - 1]
        top.grad_out, top.g_T, top.g_t0 = (g_top.data_ptr() if g_top is not None else None), t, 0
        chunk = [None] * n_layers
        t_pad = n_chunks * lag
        u_end = t_pad + (n_layers - 1) * lag + 1
        def back_project(l, c):
            m = rows[c].numel()
            above = dgates[l + 1].view(b * t, 4 * hid)
            if precision == 'fp32':
                return ops.linear_dgrad_f32(ops.gather_rows(above, rows[c]), w_ih[l + 1], None)
            dx = ops.linear_dgrad_bf16(ops.gather_rows(above, rows[c], out_bf16=True), m, 4 * hid, w_t_bf[l + 1], hid, None, out_f32=True)
            return dx if dx.shape[1] == hid else dx[:, :hid].contiguous()

        for u0 in range(0, u_end, lag):
            todo = [(l, n_chunks - 1 - (u0 // lag - (n_layers - 1 - l))) for l in range(n_layers - 2, -1, -1)]
            todo = [(l, c) for l, c in todo if 0 <= c < n_chunks]
            for l, c in todo:
                dx = back_project(l, c)
                chunk[l] = dx
                descs[l].grad_out, descs[l].g_T, descs[l].g_t0 = dx.data_ptr(), rows[c].numel() // b, c * lag
            ops.lstm_stack_bwd(descs, n_layers, seq_len, b, t, hid, lag, u0, min(u0 + lag, u_end))
        # weight gradients and the gradient w.r.t. the stack's input: big GEMMs over all frames, as for a single layer
        m = b * t
        prev_rows = state_rows(b, t, dev)
        grads = []
        for l in range(n_layers):
            dg2 = dgates[l].view(m, 4 * hid)
            hs2 = hstate[l].view(b * (t + 1), hid)
            in_dim = i_dim if l == 0 else hid
            if precision == 'fp32':
                x_in = x_saved if l == 0 else outs[l - 1].view(m, hid)
                dw_ih, db = ops.linear_wgrad_f32(dg2, x_in, None, 4 * hid, in_dim)
                dw_hh, _ = ops.linear_wgrad_f32(dg2, hs2, prev_rows, 4 * hid, hid, want_bias=False)
            else:
                dg_bf = ops.cast_pad_bf16(dg2)
                x_in = x_saved if l == 0 else ops.cast_pad_bf16(outs[l - 1].view(m, hid))
                dw_ih, db = ops.linear_wgrad_bf16(dg_bf, x_in, None, m, 4 * hid, in_dim)
                dw_hh, _ = ops.linear_wgrad_bf16(dg_bf, ops.cast_pad_bf16(hs2), prev_rows, m, 4 * hid, hid, want_bias=False)
            grads += [dw_ih, dw_hh, db, db.clone()]
        dx = None
        if ctx.needs_input_grad[2]:
            dg2 = dgates[0].view(m, 4 * hid)
            if precision == 'fp32':
                dx = ops.linear_dgrad_f32(dg2, w_ih[0], None).view(b, t, i_dim)
            else:
                dx = ops.linear_dgrad_bf16(ops.cast_pad_bf16(dg2), m, 4 * hid, w_t_bf[0], i_dim, None, out_f32=True)
                if dx.shape[1] != i_dim:
                    dx = dx[:, :i_dim].contiguous()
                dx = dx.view(b, t, i_dim)
        return (None, None, dx, None, dh0 if ctx.has_h0 else None, dc0 if ctx.has_c0 else None) + tuple(grads)


class MaskedMSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, predictions, targets, seq_len, kind='mse'):
        loss, grad = ops.masked_mse(predictions, targets, seq_len, want_grad=ctx.needs_input_grad[0], kind=kind)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        (grad,) = ctx.saved_tensors
        if _is_unit_grad(grad_loss):                 # functional.backward's cached one: nothing to scale (a 20 MB elementwise pass at C4)
            return grad, None, None, None
        return grad * grad_loss, None, None, None


class StreamLossFn(torch.autograd.Function):
    """Multi-stream loss of models/RNN_SPSS.py:120-139 in one pass: returns (loss, sigmoid of the BCE stream or None)."""

    @staticmethod
    def forward(ctx, predictions, seq_len, kinds, want_prob, *targets):
        loss, grad, prob = ops.stream_loss(predictions, targets, kinds, seq_len, want_grad=ctx.needs_input_grad[0],
                                           want_prob=want_prob)
        ctx.save_for_backward(grad)
        ctx.n_targets = len(targets)
        ctx.set_materialize_grads(False)
        if prob is not None:
            ctx.mark_non_differentiable(prob)
        return loss, prob

    @staticmethod
    def backward(ctx, grad_loss, _grad_prob):
        (grad,) = ctx.saved_tensors
        if _is_unit_grad(grad_loss):
            return (grad, None, None, None) + (None,) * ctx.n_targets
        return (grad * grad_loss, None, None, None) + (None,) * ctx.n_targets


class GatherRowsFn(torch.autograd.Function):
    """out[m] = x2d[rows[m]] (zero for -1) with distinct non-negative rows: backward is a scatter."""

    @staticmethod
    def forward(ctx, x2d, rows):
        x2d = ops._require(x2d, torch.float32, 'sequence_feature')
        ctx.save_for_backward(rows)
        ctx.n_src = x2d.shape[0]
        return ops.gather_rows(x2d, rows)

    @staticmethod
    def backward(ctx, grad_out):
        (rows,) = ctx.saved_tensors
        return ops.scatter_rows(grad_out.contiguous(), rows, ctx.n_src), None
