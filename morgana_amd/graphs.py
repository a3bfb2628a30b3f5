"""HIP-graph replay of the training step.

The reference's loop body (``experiment_builder.py:468-474``: ``zero_grad``, ``model(features)``, ``backward``, ``optimizer.step``)
is ~35 kernel launches in this package; once the kernels of the README F0Model step sum to 0.3 ms the Python / autograd / launch
path around them (0.5 ms per step) is what bounds the step.  ``GraphedTrainStep`` captures the step once into HIP graphs
(``torch.cuda.CUDAGraph`` = hipGraph on ROCm) and replays it: the same kernels on the same buffers, one launch from the host.

What makes the step capturable: every kernel of the path is launched on torch's current stream with no host synchronisation,
scratch comes from torch's allocator (graph-private pool during capture), and the only step-dependent scalars - Adam's bias
corrections - are read from device memory (``optim.Adam.advance`` / ``step_captured``, ``mg_adam_step_dev_f32``).  With more than
one rank the gradient all-reduce is captured too when the stack can do that (two buckets, the early one overlapped with the first
layer's weight gradient; see ``GraphedTrainStep``), otherwise it stays outside the graph (forward + backward graph, eager RCCL
all-reduce, the one-kernel update launched directly).

The captured step works on fixed buffers: ``features`` must be the same device tensors for every replay (copy a new batch into
them with ``load``; shapes must not change).  Metrics accumulated inside ``loss`` keep accumulating - their accumulators are device
tensors.  The learning rate may change between replays (it enters through ``advance``).
"""
import os

import collections
import warnings

import torch

from . import functional


_CAPTURE_VERDICT = {}


def rccl_capture_report(device, group=None):
    """What the capture probe found, cached per process group (the probe captures - and may fail to capture - a collective, which is
    not something to repeat): {'verdict': use the captured exchange?, 'world', 'backend', 'graph_held_collective': did the captured
    graph contain a node at all (torch warns "The CUDA Graph is empty" when a one-rank collective is elided),
    'replayed_sum_equals_world': did both replays leave world x 1.0 in the buffer (None if nothing was replayed), 'error'}.
    At world size 1 a green verdict says nothing about a LIVE collective - the report says so ('vacuous')."""
    import torch.distributed as dist
    key = (str(device), id(group), dist.get_backend(group), dist.get_world_size(group))
    if key not in _CAPTURE_VERDICT:
        _CAPTURE_VERDICT[key] = _rccl_capture_probe(device, group)
    return _CAPTURE_VERDICT[key]


def rccl_capture_works(device, group=None):
    return rccl_capture_report(device, group)['verdict']


def _leave_failed_capture(entry_stream):
    """A capture that raised inside torch.cuda.graph's block can leave the capture stream current (its context manager's exit fails
    before it restores the stream): every later launch of the thread then fails with hipErrorStreamCaptureInvalidated.  Putting the
    entry stream back makes the thread usable again (measured on this stack with a collective that cannot be captured)."""
    try:
        torch.cuda.set_stream(entry_stream)
        torch.cuda.synchronize()
    except Exception:                                                   # noqa: BLE001 - best effort; the caller reports its own error
        pass


def _rccl_capture_probe(device, group=None):
    """Can this stack capture an RCCL all-reduce into a HIP graph and replay it correctly?  Captures one on a scratch buffer,
    replays it twice and checks the sums; every rank then takes the minimum of the verdicts (one eager all-reduce), so that all ranks
    choose the same exchange mode.  Any exception on the way counts as "no"."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    report = {'verdict': False, 'world': world, 'backend': dist.get_backend(group), 'graph_held_collective': None,
              'replayed_sum_equals_world': None, 'error': None, 'vacuous': world == 1}
    if dist.get_backend(group) != 'nccl':              # gloo moves device tensors through the host: never capturable, and trying leaves
        report['error'] = 'backend %s: not capturable (not tried)' % dist.get_backend(group)      # the thread's capture invalidated
        return report
    ok = 1.0
    entry_stream = torch.cuda.current_stream()
    try:
        buf = torch.empty(1024, dtype=torch.float32, device=device)
        dist.all_reduce(buf.fill_(1.0), group=group)                   # communicator set-up outside the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            buf.fill_(1.0)
            dist.all_reduce(buf, group=group)                           # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter('always')
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                dist.all_reduce(buf, group=group)
        report['graph_held_collective'] = not any('Graph is empty' in str(w.message) for w in caught)
        sums_ok = True
        for _ in range(2):
            buf.fill_(1.0)
            graph.replay()
            torch.cuda.synchronize()
            if not bool((buf == float(world)).all().item()):
                ok, sums_ok = 0.0, False
        report['replayed_sum_equals_world'] = sums_ok
        if world > 1 and not report['graph_held_collective']:
            ok = 0.0                                   # more than one rank and a graph without the collective: never "captured"
    except Exception as exc:                                            # noqa: BLE001 - any failure means "use the eager exchange"
        ok = 0.0
        report['error'] = str(exc).splitlines()[0][:200] if str(exc) else type(exc).__name__
        _leave_failed_capture(entry_stream)
    verdict = torch.full((1,), ok, dtype=torch.float32, device=device)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=group)
    report['verdict'] = bool(verdict.item() > 0.5)
    return report


class GraphedTrainStep(object):
    """``exchange`` (more than one rank only): 'captured' = the gradient all-reduce is part of the graph - the bucket that is final
    before the first layer's weight gradient starts goes out on a side stream underneath that kernel, the rest right behind it,
    then the update: ONE launch per step from the host; 'eager' = forward + backward graph, one eager RCCL all-reduce of the whole
    bucket, the update kernel launched directly (three launches, nothing overlapped); 'auto' (default, or $MG_EXCHANGE) = captured if
    ``rccl_capture_works`` says so on every rank, else eager.  The mode taken is in ``exchange_mode``."""

    def __init__(self, model, optimizer, features, warmup=3, exchange=None, steps_per_replay=1):
        """``steps_per_replay`` = K: K consecutive training steps are captured into ONE graph (``features`` = one batch used by all K,
        or a list of K batches resident on the device) and one call performs K steps.  Between two replays the device idles for the
        graph launch (8-9 us measured) and runs the launch that stages Adam's step-dependent scalars (4.7 us): at a 0.17 ms step
        that is 8 % of the time, and it is paid once per replay, not once per step.  The update of step j reads scalar slot j
        (``optim.Adam.advance(K)`` stages all K with one launch); results are bit-identical to K single-step replays."""
        self.model, self.optimizer = model, optimizer
        self.steps_per_replay = int(steps_per_replay)
        if isinstance(features, (list, tuple)):
            if len(features) != self.steps_per_replay:
                raise ValueError('a list of batches must have steps_per_replay = %d entries' % self.steps_per_replay)
            self.batches = list(features)
        else:
            self.batches = [features] * self.steps_per_replay
        self.features = self.batches[0]
        self.loss = None
        self.losses = []
        self.output = None
        # this object runs exactly the reference's loop body, so the optimiser may treat it as one unit: the update kernel zeroes the
        # gradient behind its read and (one rank) sums the weight-gradient slabs the backward pass leaves for it (optim.Adam.fused_loop)
        optimizer.fused_loop = True
        self._multi = optimizer.exchanging()
        self.exchange_mode = 'none'
        if warmup:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):                # allocator and workspace warm-up, off the capture; these ARE training steps
                    self._eager_step()
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self._multi:
            want = exchange or os.environ.get('MG_EXCHANGE', 'auto')
            if want not in ('auto', 'captured', 'eager'):
                raise ValueError("exchange must be 'auto', 'captured' or 'eager', got %r" % (want,))
            device = optimizer.flat_buffers()['grad'].device
            captured = want != 'eager' and rccl_capture_works(device, optimizer.process_group)
            if want == 'captured' and not captured:
                raise RuntimeError('exchange="captured": this stack cannot capture an RCCL all-reduce into a HIP graph')
            self.exchange_mode = 'captured' if captured else 'eager'
        self._fwd_bwd = torch.cuda.CUDAGraph()
        self.optimizer.prepare_capture()               # capture records the launches, it does not run them
        # the dropout step counter must exist before the capture opens (warmup=0: no eager step has created it) - created inside, it
        # would be reset by every replay (ops.dropout_state)
        from . import ops as _ops
        for p in model.parameters():
            if p.is_cuda:
                _ops.dropout_state(p.device)
                break
        # with a process group alive its watchdog thread polls events while we capture: judge only this thread's calls
        mode = dict(capture_error_mode='thread_local') if self._multi else {}
        if self.steps_per_replay > 1 and self.exchange_mode == 'eager':
            # the collective sits between graph launches: one step per replay.  One batch for all steps: fall back quietly (the caller
            # reads steps_per_replay); a list of batches cannot be honoured
            if any(bt is not self.batches[0] for bt in self.batches):
                raise ValueError('steps_per_replay > 1 needs the whole step inside the graph (one rank, or a captured exchange)')
            self.steps_per_replay = 1
            self.batches = self.batches[:1]
        entry_stream = torch.cuda.current_stream()
        try:
            self._capture(mode)
        except Exception:
            _leave_failed_capture(entry_stream)        # so that the caller's eager fallback finds a usable stream
            raise
        # the bf16 operand copies the captured update keeps current: what a replay re-stamps (optim.Adam.note_replayed)
        self._refreshed = self.optimizer.refreshed_shadows() if hasattr(self.optimizer, 'refreshed_shadows') else []
        self.steps_done = warmup

    def _capture(self, mode):
        # a step captured whole: what the forward leaves for the backward's first launch to finish (functional.DEFER_TAIL) cannot be
        # observed half done - a replay runs forward and backward as one unit
        prev = functional.set_defer_tail(True)
        from . import ops
        ops.begin_capture_epoch()                      # operand splits of 'bf16x3' are recorded afresh in every capture
        try:
            self._capture_steps(mode)
        finally:
            functional.set_defer_tail(prev)

    def _capture_steps(self, mode):
        with torch.cuda.graph(self._fwd_bwd, **mode):
            for j in range(self.steps_per_replay):
                self.optimizer.zero_grad()
                self.loss, self.output = self.model(self.batches[j])
                self.losses.append(self.loss)
                if self.exchange_mode == 'captured':
                    self._capture_backward_and_exchange()
                    self.optimizer.step_captured(slot=j)
                else:
                    functional.backward(self.loss)
                    if not self._multi:
                        self.optimizer.step_captured(slot=j)

    def _capture_backward_and_exchange(self):
        """Backward with the early bucket's all-reduce forked onto a side stream the moment its gradients are final (the hook fires
        inside the backward pass, ahead of the first layer's weight-gradient kernel), joined again before the late bucket's."""
        main = torch.cuda.current_stream()
        comm = torch.cuda.Stream()
        fired = []

        def early_bucket_ready(stack_params):
            # "everything but the first layer's gradient is final" is a statement about the firing STACK.  It is a statement about
            # the model's flat gradient only if that stack's parameters are the optimiser's, in its order, and all of them - a model
            # with further trainable parameters, or two fused stacks, must not cut its exchange here (the late part would go out
            # before it is final, or the early part twice): such steps keep the single collective behind the backward pass.
            if fired or not early_exchange_is_safe(self.optimizer, stack_params):
                return
            comm.wait_stream(main)
            with torch.cuda.stream(comm):
                self.optimizer.exchange_gradients('early')
            fired.append(True)

        split = self.optimizer.bucket_split() > 0
        prev = functional.set_early_grads_hook(early_bucket_ready if split else None)
        try:
            functional.backward(self.loss)
        finally:
            functional.set_early_grads_hook(prev)
        if fired:
            main.wait_stream(comm)
            self.optimizer.exchange_gradients('late')
        else:
            self.optimizer.exchange_gradients()        # a model whose backward has no early point: one collective

    def _eager_step(self):
        self.optimizer.zero_grad()
        loss, _ = self.model(self.features)
        functional.backward(loss)
        self.optimizer.step()

    def load(self, features, slot=0, keys=None, extra_pairs=()):
        """Copy a new batch (same keys, shapes and dtypes) into the captured buffers (of step ``slot`` of a multi-step replay).
        ``keys``: the tensors the captured step reads (``BaseModel.step_input_keys``); None = every tensor of the batch.
        ``extra_pairs``: further (dst, src) device copies that ride in the same launch (GraphedStepCache files the previous step's loss)."""
        pairs, rest = list(extra_pairs), []
        for key, value in features.items():
            if isinstance(value, torch.Tensor) and (keys is None or key in keys):
                dst = self.batches[slot][key]
                ok = (value.is_cuda and dst.is_cuda and value.device == dst.device and value.dtype == dst.dtype and value.is_contiguous()
                      and dst.is_contiguous() and value.numel() == dst.numel())
                (pairs if ok else rest).append((dst, value))
        if pairs:
            from . import ops
            ops.copy_many(pairs)                       # one launch for the batch's tensors (a copy kernel each: 3-15 us apiece)
        for dst, value in rest:
            dst.copy_(value, non_blocking=True)

    def __call__(self):
        """``steps_per_replay`` training steps (one by default); returns the (device, 0-d) loss tensor of the last of them - valid
        until the next call (``losses``: one tensor per step of the replay)."""
        self.optimizer.advance(self.steps_per_replay)
        self._fwd_bwd.replay()
        if self.exchange_mode == 'eager':
            self.optimizer.exchange_gradients()        # the step's one RCCL all-reduce, outside the graph
            self.optimizer.step_captured()             # one kernel: launched directly
        elif hasattr(self.optimizer, 'note_replayed'):
            self.optimizer.note_replayed(self._refreshed, self.steps_per_replay)
        self.steps_done += self.steps_per_replay
        return self.loss


def early_exchange_is_safe(optimizer, stack_params, group=0):
    """May the step all-reduce ``flat['grad'][bucket_split:]`` the moment a stack reports that everything but its first layer's
    gradient is final?  Only when the stack's parameter list is exactly the optimiser's flat parameter list (same objects, same
    order): then "[split:] is final" holds for the buffer that is exchanged."""
    flat = optimizer.flat_buffers(group)
    if flat is None or optimizer.bucket_split(group) <= 0:
        return False
    own = flat['params']
    return len(stack_params) == len(own) and all(a is b for a, b in zip(stack_params, own))


# Load the next batch into a second captured step's buffers on a side stream while the current step runs (GraphedStepCache.prefetch).
# MEASURED (round 5, C2, profiles/r5_train_epoch_timeline.txt) and OFF: the copy does run beside the step, but it takes bandwidth and CUs
# from the weight-gradient kernel it overlaps (26 -> 37-41 us) and the replay's wait for the side stream's event costs the ~20 us the
# copy would have - 0.158 ms per step either way.
PREFETCH = os.environ.get('MORGANA_GRAPH_PREFETCH', '0') != '0'


class GraphedStepCache(object):
    """Graph replay inside a training loop whose batches repeat a few shapes (``ExperimentBuilder(use_graphs=True)``).

    A batch whose (name, shape, dtype) signature is new runs eagerly - an ordinary training step, which also warms up allocator and
    workspaces for that shape; the second batch of a signature is captured into static copies of its tensors (capture runs
    nothing) and replayed, and every later one is copied into those buffers and replayed.  No step is ever run twice or skipped, so
    the loop trains exactly as the eager loop does (bit-identical, tests/test_gpu_parity.py).

    ``prefetch`` (optional, called right AFTER ``step`` with the batch that comes next - what ``ExperimentBuilder.train_epoch`` does with
    one batch of look-ahead): a signature holds TWO captured
    steps with a set of static buffers each, used in turn, and the NEXT batch is copied into the idle set on a side stream while the
    current step's graph runs - the copy of a batch (27.5 MB operand table at BASELINE config C2: ~15 us) then costs the step nothing.
    Without ``prefetch`` the copy runs in line in front of the replay, as before."""

    MAX_SEEN = 1024      # signatures remembered as "ran once": a ragged loader whose frame totals never repeat must not grow this forever

    def __init__(self, model, optimizer, max_graphs=8, max_group_graphs=256):
        self.model, self.optimizer, self.max_graphs, self.max_group_graphs = model, optimizer, max_graphs, max_group_graphs
        self._no_capture = set()                          # signatures whose capture failed: ordinary launches from then on
        self._groups = {}                                 # group_key -> GraphedTrainStep over K resident batches (``step_group``)
        self.group_replays = 0
        self._seen = collections.OrderedDict()
        self._steps = {}                                  # signature -> [GraphedTrainStep, ...] (at most two: the ping-pong pair)
        self._turn = {}                                   # signature -> index of the step object the next batch takes
        self._copy_stream = None
        self._prefetched = None                           # (features object, signature, step object) of the batch loaded ahead
        self._before_replay = None                        # event on the main stream in front of the last replay (see ``prefetch``)
        self._pending_loss = None                         # (slot, loss tensor) of the last replayed step, not yet filed
        self.eager_steps = self.replayed_steps = self.prefetched_steps = 0        # how the steps were run (see ``stats``)
        self._warned = False

    def stats(self):
        """{'eager': steps run as ordinary launches, 'replayed': steps replayed from a graph, 'graphs': graphs held, 'prefetched':
        replays whose batch was loaded ahead on the side stream}.  Mostly eager steps mean the batches do not repeat their signature:
        bucket ragged lengths to a few shapes for ``use_graphs`` to pay."""
        return {'eager': self.eager_steps, 'replayed': self.replayed_steps, 'graphs': sum(len(v) for v in self._steps.values()),
                'prefetched': self.prefetched_steps, 'group_graphs': len(self._groups), 'group_replays': self.group_replays}

    def _keys(self, features):
        keys = self.model.step_input_keys(features) if hasattr(self.model, 'step_input_keys') else None
        return None if keys is None else set(keys)

    def prefetch(self, features):
        """Load ``features`` (the batch ``step`` will be called with NEXT; call this right after the current ``step``) into the idle
        static buffers of its signature on a side stream, beside the step that has just been launched.  A no-op unless the signature
        has its two captured steps already."""
        if features is None or self._prefetched is not None or not PREFETCH:
            return
        key = self.signature(features)
        pair = self._steps.get(key)
        if pair is None or len(pair) < 2:
            return
        graphed = pair[self._turn.get(key, 0)]
        main = torch.cuda.current_stream()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream()
        side = self._copy_stream
        # The copy may start once (a) the loader's kernels that produced this batch and (b) the last replay that read the target buffers
        # have run.  Both were enqueued on `main` BEFORE the replay of the step that is running now (`_before_replay`: recorded by
        # ``step`` right in front of its replay; the caller fetched this batch before it called ``step``) - so the copy waits for that
        # point, not for the running step, and runs beside it.
        if self._before_replay is not None:
            side.wait_event(self._before_replay)
        else:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            graphed.load(features, keys=self._keys(features))
            for v in features.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(side)                  # the batch's memory must outlive the copy, whatever frees it on `main`
            done = torch.cuda.Event()
            done.record(side)
        self._prefetched = (features, key, graphed, done)

    def graph_count(self):
        return sum(len(v) for v in self._steps.values())

    @staticmethod
    def signature(features):
        # host-side integers (n_frames_total sizes the packed-frame layout) are part of what a captured step was built for
        return tuple(sorted((k, tuple(v.shape), str(v.dtype)) if isinstance(v, torch.Tensor) else (k, v)
                            for k, v in features.items() if isinstance(v, (torch.Tensor, int))))

    @staticmethod
    def group_key(batches):
        """What a graph captured over RESIDENT batches was built on: every tensor's address, shape and dtype and every host integer of
        every batch, in order.  (The graph keeps the batches alive, so an address in a key cannot come to mean another tensor.)"""
        return tuple(tuple(sorted((k, v.data_ptr(), tuple(v.shape), str(v.dtype)) if isinstance(v, torch.Tensor) else (k, v)
                                  for k, v in f.items() if isinstance(v, (torch.Tensor, int))))
                     for f in batches)

    def _file_losses(self, pairs):
        from . import ops
        pairs = [(dst, src.detach().reshape(dst.shape)) for dst, src in pairs if dst is not None]
        if pairs:
            ops.copy_many(pairs)                          # MG_COPY_MAX pairs per launch

    def step_group(self, batches, loss_slots=None):
        """K consecutive training steps on K batches that STAY where they are (a resident epoch: the loader is a list of device
        batches, kept by the caller from epoch to epoch - 288 GB of HBM hold the corpora this model family trains on).  The first time
        a group is seen with all of its batch signatures warmed up, the K steps are captured into ONE graph that reads the batches in
        place - no load launch, no static copies - and every later epoch replays it: one graph launch and one scalar-staging launch per
        K steps instead of per step, ragged shapes included (each step of the graph is captured on its own batch's shapes).  A group
        with a signature that has not run yet runs as eager steps (which warm up allocator and workspaces, as in ``step``); beyond
        ``max_group_graphs`` graphs, with more than one rank, or with non-resident batches the steps take the single-step path.
        No step is run twice or skipped.  ``loss_slots``: K 0-d device tensors that receive the steps' losses (one launch per group).
        Returns the list of the K loss tensors (replayed: the graph's own, valid until its next replay)."""
        batches = list(batches)
        slots = list(loss_slots) if loss_slots is not None else [None] * len(batches)
        ahead, self._prefetched = self._prefetched, None
        if ahead is not None:
            torch.cuda.current_stream().wait_event(ahead[3])
        self.flush()
        sigs = [self.signature(f) for f in batches]
        resident = all(v.is_cuda for f in batches for v in f.values() if isinstance(v, torch.Tensor))
        if any(sig not in self._seen for sig in sigs):
            losses = [self._eager(f, sig)[0] for f, sig in zip(batches, sigs)]
            self._file_losses(zip(slots, losses))
            return losses
        key = self.group_key(batches) if resident else None
        graphed = self._groups.get(key) if key is not None else None
        if graphed is None and (key is None or self.optimizer.exchanging() or len(self._groups) >= self.max_group_graphs):
            losses = []
            for f, slot in zip(batches, slots):
                losses.append(self.step(f, clone_loss=slot is None, loss_slot=slot)[0])
            return losses
        if graphed is None:
            graphed = None if any(sig in self._no_capture for sig in sigs) else self._try_capture(sigs[0], batches, len(batches))
            if graphed is None:                           # a step that cannot be captured: ordinary launches from now on
                for sig in sigs:
                    self._no_capture.add(sig)
                losses = [self._eager(f, sig)[0] for f, sig in zip(batches, sigs)]
                self._file_losses(zip(slots, losses))
                return losses
            self._groups[key] = graphed
        graphed()
        self.replayed_steps += len(batches)
        self.group_replays += 1
        self._file_losses(zip(slots, graphed.losses))
        return list(graphed.losses)

    def _try_capture(self, key, features, k):
        """A captured step (of ``k`` batches), or None when the model's step cannot be captured - a forward that reads a device value
        back (``.item()``, a sizing sync), say: HIP refuses the call inside a capture.  The signature then stays on ordinary launches
        (one warning); training goes on - capture runs nothing, so no step is lost."""
        try:
            return GraphedTrainStep(self.model, self.optimizer, features, warmup=0, steps_per_replay=k)
        except Exception as exc:                          # noqa: BLE001 - whatever the capture tripped over, the eager loop does not
            self._no_capture.add(key)
            warnings.warn('GraphedStepCache: this step cannot be captured into a HIP graph (%s) - batches of this shape run as ordinary '
                          'launches' % (str(exc).splitlines()[0][:160] if str(exc) else type(exc).__name__))
            return None

    def _eager(self, features, key):
        self._before_replay = None                        # an eager step: a batch loaded ahead waits for all of it
        self.flush()
        self._seen[key] = True
        self._seen.move_to_end(key)
        while len(self._seen) > self.MAX_SEEN:
            self._seen.popitem(last=False)
        self.eager_steps += 1
        if not self._warned and self.eager_steps >= 64 and self.replayed_steps == 0:
            self._warned = True
            warnings.warn('GraphedStepCache: %d steps and no batch signature has repeated - every step runs as eager launches '
                          '(ragged batches: bucket the lengths to a few shapes, or turn use_graphs off)' % self.eager_steps)
        self.optimizer.zero_grad()
        loss, output = self.model(features)
        functional.backward(loss)
        self.optimizer.step()
        return loss, output

    def flush(self):
        """File the last step's loss (``step(..., loss_slot=)``) if it still waits for a launch to ride in."""
        if self._pending_loss is not None:
            dst, src = self._pending_loss
            self._pending_loss = None
            dst.copy_(src)

    def step(self, features, clone_loss=True, loss_slot=None):
        """zero_grad, forward, backward, optimizer step on ``features``; returns (loss, output_features).  ``clone_loss`` False: the
        replayed graph's own loss tensor is handed out - valid until the next replay of that graph (a caller that files it away at once).
        ``loss_slot``: a 0-d device tensor that receives the step's loss - for a replayed step LATER, in the launch that loads the next
        batch (or in ``flush``): a small copy of its own costs a launch and its gap, 14 us of a 0.1 ms step."""
        key = self.signature(features)
        pair = self._steps.get(key)
        graphed = None
        ahead, self._prefetched = self._prefetched, None
        if ahead is not None and ahead[0] is features:
            graphed = ahead[2]                            # loaded ahead on the side stream: the replay waits for that copy only
            torch.cuda.current_stream().wait_event(ahead[3])
            self.prefetched_steps += 1
        elif ahead is not None:
            torch.cuda.current_stream().wait_event(ahead[3])      # a batch loaded ahead that did not come next: its copy must still land
        if graphed is None and key in self._seen and key not in self._no_capture and (pair is None or len(pair) < 2) \
                and self.graph_count() < 2 * self.max_graphs:
            # the second (third) batch of a signature: capture a step on static copies of its tensors - two per signature, taken in turn
            static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in features.items()}
            graphed = self._try_capture(key, static, 1)
            if graphed is None:
                return self._eager(features, key)         # (capture runs nothing: the step has not been taken yet)
            self._steps.setdefault(key, []).append(graphed)
            self._turn[key] = 0 if len(self._steps[key]) < 2 else 1
        elif graphed is None and pair:
            # only what the captured step reads is copied into the graph's static buffers (BaseModel.step_input_keys); a tensor the step
            # never reads (the fp32 phone feature beside its operand table: 49 MB at C2) stays where it is, and the entry the captured
            # graph holds for it keeps its own static copy (nobody reads that one)
            graphed = pair[self._turn.get(key, 0)]
            extra, self._pending_loss = ([self._pending_loss] if self._pending_loss is not None else []), None
            graphed.load(features, keys=self._keys(features), extra_pairs=extra)
        if graphed is not None:
            if len(self._steps[key]) == 2:
                self._turn[key] = 1 - self._steps[key].index(graphed)
            for k, v in features.items():              # non-tensor entries (utterance names) follow the batch
                if not isinstance(v, torch.Tensor):
                    graphed.features[k] = v
            self.replayed_steps += 1
            self.flush()                                  # (a loss still pending here: this replay did not load a batch - captures)
            if PREFETCH:
                self._before_replay = torch.cuda.Event()
                self._before_replay.record(torch.cuda.current_stream())
            loss = graphed()
            if loss_slot is not None:
                self._pending_loss = (loss_slot, loss.detach().reshape(loss_slot.shape))
            return (loss.clone() if clone_loss else loss), graphed.output      # the loss buffer is rewritten by that graph's next replay
        return self._eager(features, key)
