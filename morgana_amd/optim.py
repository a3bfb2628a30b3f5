"""Fused Adam over one flat fp32 buffer, with the data-parallel gradient exchange folded in.

Stands where the reference constructs ``torch.optim.Adam(model.parameters(), lr, weight_decay)``
(experiment_builder.py:516) and calls ``zero_grad`` / ``step`` (:468, :474).  Same update rule and defaults.

MI355X design: parameters, gradients and both moments are views into four contiguous buffers, so
  * ``zero_grad`` is one memset - or nothing at all: with ``fused_loop`` the update kernel zeroes the gradient behind its read,
  * the data-parallel exchange is ONE RCCL all-reduce of the whole gradient (1.5 MB for the F0Model) per step (or two buckets, the
    early one overlapped with the backward pass: ``exchange_gradients``), with the 1/world_size mean folded into the update kernel's
    gradient read,
  * ``step`` is one elementwise kernel (mg_adam_step_plan_f32) instead of ~10 foreach passes over 8 tensors - and that kernel is
    also the step's last consumer of everything the backward pass left for it: it sums the split-M slabs of the weight-gradient
    GEMMs itself (no reduce launches, one rank only) and re-casts each weight it has changed into the bf16 operands the next
    step's GEMMs read (no cast launches).
"""
import torch
import torch.distributed as dist

from . import _lib, ops


class Adam(torch.optim.Optimizer):
    """``kernel`` is a test seam: the CPU-only unit tests inject the oracle's update there; on a device it is always
    the HIP kernel.

    ``fused_loop``: the caller runs the reference's loop body - ``zero_grad(); loss = model(...); loss.backward(); step()``
    (experiment_builder.py:468-474) - and nothing reads ``.grad`` between ``backward`` and ``step``.  Then (a) ``step`` leaves the
    gradient buffer zeroed and the following ``zero_grad`` is free, (b) on one rank the backward pass may leave split-M partial
    results of its weight-gradient GEMMs for ``step`` to sum (``defer_slabs``) instead of reducing them into ``.grad`` itself.
    ``ExperimentBuilder`` and ``graphs.GraphedTrainStep`` own exactly that loop and switch it on; off (the default) keeps torch's
    semantics to the letter (``.grad`` complete after ``backward``, untouched by ``step``)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, process_group=None,
                 kernel=None, exchange_always=False, fused_loop=False, exchange_never=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super(Adam, self).__init__(params, defaults)
        self.process_group = process_group
        # exchange_always: run the gradient exchange whenever a process group exists, also at world size 1 (where it is the
        # identity) - lets the one-GPU box exercise the multi-rank code path on a live RCCL communicator
        self.exchange_always = exchange_always
        # exchange_never: no gradient exchange even with several ranks (each rank then trains on its own shard) - a MEASUREMENT
        # switch: bench.py times the step without its exchange beside the step with it (exchange.exposed_us), never a training mode
        self.exchange_never = bool(exchange_never)
        self.fused_loop = bool(fused_loop)
        self._kernel = kernel
        self._flat = []
        for group in self.param_groups:
            plist = [p for p in group['params'] if p.requires_grad]
            if not plist:
                self._flat.append(None)
                continue
            device, total = plist[0].device, sum(p.numel() for p in plist)
            flat_p = torch.empty(total, dtype=torch.float32, device=device)
            flat_g = torch.zeros(total, dtype=torch.float32, device=device)
            off, offsets = 0, {}
            for p in plist:
                if p.dtype != torch.float32 or p.device != device:
                    raise ValueError('Adam: all parameters of a group must be float32 on one device')
                n = p.numel()
                flat_p[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat_p[off:off + n].view_as(p)
                p.grad = flat_g[off:off + n].view_as(p)
                p._mg_direct_grad = True          # functional._deliver_param_grads may add into flat_g directly
                p._mg_optimizer = self            # functional asks it whether slabs may be deferred (weak: the param outlives nothing here)
                offsets[id(p)] = off
                off += n
            self._flat.append({'param': flat_p, 'grad': flat_g, 'exp_avg': torch.zeros_like(flat_p),
                               'exp_avg_sq': torch.zeros_like(flat_p), 'step': 0, 'params': plist, 'offsets': offsets,
                               'pending': [], 'clean': False})

    def flat_buffers(self, group=0):
        return self._flat[group]

    def _on_device(self):
        return self._kernel is None and any(f is not None and f['param'].is_cuda for f in self._flat)

    def zero_grad(self, set_to_none=False):
        for flat in self._flat:
            if flat is not None:
                if flat['clean']:
                    flat['clean'] = False         # the last step's kernel zeroed the buffer; trusted once (see class docstring)
                else:
                    flat['grad'].zero_()
                flat['pending'] = []              # partial results nobody consumed belong to a gradient that is being dropped
                if flat.get('tail') is not None:  # a forward's deferred tail whose update never came: finish it as its own launch
                    ops.finish_deferred_tail(flat['tail'])
                    flat['tail'] = None
                off = 0
                for p in flat['params']:          # re-attach views a caller may have dropped (set_to_none habits)
                    n = p.numel()
                    if p.grad is None or p.grad.data_ptr() != flat['grad'][off:off + n].data_ptr():
                        p.grad = flat['grad'][off:off + n].view_as(p)
                    off += n

    def _world(self):
        if self.exchange_never:
            return 1
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.process_group)
        return 1

    # ---- what the backward pass may leave for the update kernel -------------------------------------------------------------
    def defers_slabs(self):
        """May a weight-gradient GEMM leave its split-M slabs unreduced for ``step``?  Only in the fused loop, on the device,
        and when no exchange needs the finished gradient first."""
        return self.fused_loop and self._on_device() and not self.exchanging()

    def defer_slabs(self, first_param, count, slab, n_slabs, stride):
        """Register partial results: elements [offset of ``first_param``, + count) of the flat gradient are ADDITIONALLY the sum of
        ``n_slabs`` slabs of ``stride`` floats in ``slab`` (kept alive here until the step has consumed them)."""
        for flat in self._flat:
            if flat is not None and id(first_param) in flat['offsets']:
                off = flat['offsets'][id(first_param)]
                if len(flat['pending']) >= _lib.ADAM_MAX_SLABS:
                    # the update kernel's plan holds ADAM_MAX_SLABS sources (a deep stack registers one per leading layer plus
                    # the tail): any further one is summed now, by a reduce launch straight into the flat gradient
                    ops.slab_reduce(slab, n_slabs, stride, count, flat['grad'][off:off + int(count)], accumulate=True)
                else:
                    flat['pending'].append((off, int(count), slab, int(n_slabs), int(stride)))
                return
        raise ValueError('defer_slabs: the parameter is not one of this optimiser\'s')

    def defer_tail(self, first_param, tail):
        """A forward pass's deferred tail (ops.f0_l2tail_rows_expand(defer=True)): the next update launch of the group that holds
        ``first_param`` repeats the prediction and forms the loss in its first blocks (mg_adam_tail).  Only inside a step captured
        whole into a HIP graph: until that launch has run, the loss and the frame-level prediction are not valid."""
        for flat in self._flat:
            if flat is not None and id(first_param) in flat['offsets']:
                if flat.get('tail') is not None:
                    raise RuntimeError('defer_tail: the previous deferred tail was never consumed')
                flat['tail'] = tail
                return
        raise ValueError('defer_tail: the parameter is not one of this optimiser\'s')

    def _shadows(self, flat):
        """(offset, rows, cols, plain, transposed) for every 2-D parameter that carries bf16 operand copies (ops.param_shadows)."""
        out = []
        for p in flat['params']:
            sh = getattr(p, '_mg_shadow', None)
            if sh is not None and p.dim() == 2 and sh['plain'].device == p.device:
                out.append((flat['offsets'][id(p)], p.shape[0], p.shape[1], sh['plain'], sh['t'], p, False))
            pr = getattr(p, '_mg_pair', None)              # [hi | lo] pair planes of precision 'bf16x3' (ops.pair_shadows)
            if pr is not None and p.dim() == 2 and pr['plain'].device == p.device:
                out.append((flat['offsets'][id(p)], p.shape[0], p.shape[1], pr['plain'], pr['t'], p, True))
        return out

    # ---- HIP-graph support (morgana_amd/graphs.py): the captured update reads its step-dependent scalars from device memory
    def _scalar_buffers(self, flat):
        if 'scalars' not in flat:          # slot j (floats 2 j, 2 j + 1): the scalars of the j-th step of a multi-step graph replay
            flat['scalars'] = torch.zeros(2 * ops.STORE_PAIRS_MAX, dtype=torch.float32, device=flat['param'].device)
        return flat['scalars']

    def exchanging(self):
        """True when ``step`` has a gradient exchange to do."""
        if self.exchange_never or not (dist.is_available() and dist.is_initialized()):
            return False
        return self._world() > 1 or self.exchange_always

    def bucket_split(self, group=0):
        """Element index that cuts the flat gradient into the part produced LAST by the backward pass ([:split], the first Linear
        layer's weight and bias: 81 % of the README F0Model's bytes) and the part that is final before that layer's weight-gradient
        kernel starts ([split:]).  0 = no cut (the first layer is less than a quarter of the gradient, or there is nothing else)."""
        flat = self._flat[group]
        if flat is None or len(flat['params']) < 3:
            return 0
        first = flat['params'][0].numel() + (flat['params'][1].numel() if flat['params'][1].dim() == 1 else 0)
        total = flat['grad'].numel()
        return first if 4 * first >= total and first < total else 0

    def exchange_gradients(self, part=None):
        """The step's gradient all-reduce (SUM; the mean is folded into the update kernel).  ``part``: None = the whole flat buffer
        in one collective; 'early' / 'late' = the two sides of ``bucket_split`` (a step that overlaps the early side with the rest
        of its backward pass calls both, so every element is still reduced exactly once).  No-op without a process group."""
        if not self.exchanging():
            return
        for g, flat in enumerate(self._flat):
            if flat is None:
                continue
            split = self.bucket_split(g)
            buf = flat['grad']
            if part == 'early':
                buf = buf[split:] if split else None
            elif part == 'late':
                buf = buf[:split] if split else buf
            if buf is not None and buf.numel():
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.process_group)

    def prepare_capture(self):
        for flat in self._flat:
            if flat is not None:
                self._scalar_buffers(flat)

    def advance(self, n=1):
        """Count ``n`` steps and stage their (step_size, bc2_sqrt) for the captured updates: ONE small launch per group on the current
        stream ahead of the graph replay that consumes them (slot j for the j-th step of the replay), the values carried as kernel
        arguments (the host runs many replays ahead of the device; a copy from a reused host buffer would be read too late).  The
        learning rate is the group's current one for all ``n`` steps."""
        if not 1 <= n <= ops.STORE_PAIRS_MAX:
            raise ValueError('advance: 1 <= n <= %d' % ops.STORE_PAIRS_MAX)
        for group, flat in zip(self.param_groups, self._flat):
            if flat is None:
                continue
            pairs = []
            for _ in range(n):
                flat['step'] += 1
                pairs.append(ops.adam_scalars(group['lr'], group['betas'], flat['step']))
            if n == 1:
                ops.store_pair(self._scalar_buffers(flat), *pairs[0])
            else:
                ops.store_pairs(self._scalar_buffers(flat), pairs)

    def _launch(self, group, flat, world, slot=0):
        """The update kernel with everything this step left for it (see the module docstring)."""
        # the kernel's plan refreshes up to ADAM_MAX_SHADOWS operand copies; the others are re-cast by ONE batched launch right behind
        # it (ops.refresh_shadows), so that every copy is current when the update ends.  (They used to keep a stale stamp for the next
        # forward pass to notice - which a step captured into a HIP graph cannot do: a replay moves no stamp, and a capture that
        # followed a no_grad forward found every stamp current and recorded no cast, so the layers past the plan multiplied by
        # stale bf16 weights from the second replay on.  ADVICE round 3.)
        every = self._shadows(flat)
        pairs = [sh for sh in every if sh[6]]               # pair planes have no batched re-split launch: they go first
        every = pairs + [sh for sh in every if not sh[6]]
        shadows, rest = every[:_lib.ADAM_MAX_SHADOWS], every[_lib.ADAM_MAX_SHADOWS:]
        stale_pairs = [sh for sh in rest if sh[6]]
        rest = [sh for sh in rest if not sh[6]]
        pending, flat['pending'] = flat['pending'], []
        tail, flat['tail'] = flat.get('tail'), None
        ops.adam_step_plan(flat['param'], flat['grad'], flat['exp_avg'], flat['exp_avg_sq'], group['betas'], group['eps'],
                           group['weight_decay'], self._scalar_buffers(flat)[2 * slot:], 1.0 / world, slab_srcs=pending,
                           shadows=[sh[:5] + (sh[6],) for sh in shadows], clear_grad=self.fused_loop, tail=tail)
        if rest:
            ops.refresh_shadows([sh[5] for sh in rest])
        flat['clean'] = self.fused_loop
        for p in flat['params']:
            p._mg_updates = getattr(p, '_mg_updates', 0) + 1
        for sh in every:                                     # refreshed by this update: current again
            if sh in stale_pairs:
                continue                                     # (more pairs than the plan holds: their next reader splits them again)
            p = sh[5]
            (p._mg_pair if sh[6] else p._mg_shadow)['version'] = (p._version, p._mg_updates)
        # what this update refreshed, by identity of the copies: a graph that captured it re-stamps exactly these after a replay
        flat['refreshed'] = [(sh[5], sh[3], sh[4]) for sh in every if sh not in stale_pairs]

    def refreshed_shadows(self):
        """[(parameter, plain copy, transposed copy or None)] the last update kernel launch (or its capture) kept current."""
        out = []
        for flat in self._flat:
            if flat is not None:
                out += flat.get('refreshed', [])
        return out

    def note_replayed(self, refreshed, n_steps=1):
        """Book-keeping for ``n_steps`` updates that ran inside a graph replay (the host saw none of them): every parameter counts
        as updated, and of the bf16 operand copies only those the captured update refreshes (``refreshed``: what ``refreshed_shadows``
        returned right after the capture) are current - a copy allocated since, e.g. a transposed one first asked for by a later
        eager pass, keeps its old stamp and is cast again by its next reader (ops.param_shadows)."""
        for flat in self._flat:
            if flat is not None:
                for p in flat['params']:
                    p._mg_updates = getattr(p, '_mg_updates', 0) + n_steps
        for p, plain, trans in refreshed:
            for sh in (getattr(p, '_mg_shadow', None), getattr(p, '_mg_pair', None)):
                if sh is not None and sh['plain'] is plain and sh['t'] is trans:
                    sh['version'] = (p._version, p._mg_updates)

    @torch.no_grad()
    def step_captured(self, slot=0):
        """The parameter update alone, from device-resident scalars (call ``advance`` before each replay; ``slot`` = which of the
        staged pairs: the position of this step inside a multi-step replay).  No all-reduce here."""
        world = self._world()
        for group, flat in zip(self.param_groups, self._flat):
            if flat is not None:
                self._launch(group, flat, world, slot)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        world = self._world()
        for group, flat in zip(self.param_groups, self._flat):
            if flat is None:
                continue
            if self.exchanging():
                # the single gradient exchange of the step: sum over ranks, mean folded into the kernel below
                dist.all_reduce(flat['grad'], op=dist.ReduceOp.SUM, group=self.process_group)
            flat['step'] += 1
            if self._kernel is not None or not flat['param'].is_cuda:
                kernel = self._kernel if self._kernel is not None else ops.adam_step
                kernel(flat['param'], flat['grad'], flat['exp_avg'], flat['exp_avg_sq'], group['lr'], group['betas'],
                       group['eps'], group['weight_decay'], flat['step'], 1.0 / world)
                continue
            step_size, bc2_sqrt = ops.adam_scalars(group['lr'], group['betas'], flat['step'])
            ops.store_pair(self._scalar_buffers(flat), step_size, bc2_sqrt)
            self._launch(group, flat, world)
        return loss
