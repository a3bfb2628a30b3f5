"""HIP-graph replay of the training step.

The reference's loop body (``experiment_builder.py:468-474``: ``zero_grad``, ``model(features)``, ``backward``, ``optimizer.step``)
is ~35 kernel launches in this package; once the kernels of the README F0Model step sum to 0.3 ms the Python / autograd / launch
path around them (0.5 ms per step) is what bounds the step.  ``GraphedTrainStep`` captures the step once into HIP graphs
(``torch.cuda.CUDAGraph`` = hipGraph on ROCm) and replays it: the same kernels on the same buffers, one launch from the host.

What makes the step capturable: every kernel of the path is launched on torch's current stream with no host synchronisation,
scratch comes from torch's allocator (graph-private pool during capture), and the only step-dependent scalars - Adam's bias
corrections - are read from device memory (``optim.Adam.advance`` / ``step_captured``, ``mg_adam_step_dev_f32``).  With more than
one rank the gradient all-reduce stays OUTSIDE the graph (forward + backward graph, eager RCCL all-reduce, the one-kernel update
launched directly).

The captured step works on fixed buffers: ``features`` must be the same device tensors for every replay (copy a new batch into
them with ``load``; shapes must not change).  Metrics accumulated inside ``loss`` keep accumulating - their accumulators are device
tensors.  The learning rate may change between replays (it enters through ``advance``).
"""
import torch

from . import functional


class GraphedTrainStep(object):
    def __init__(self, model, optimizer, features, warmup=3):
        self.model, self.optimizer, self.features = model, optimizer, features
        self.loss = None
        self.output = None
        self._multi = optimizer._world() > 1
        if warmup:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):                # allocator and workspace warm-up, off the capture; these ARE training steps
                    self._eager_step()
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._fwd_bwd = torch.cuda.CUDAGraph()
        self.optimizer.prepare_capture()               # capture records the launches, it does not run them
        # with a process group alive its watchdog thread polls events while we capture: judge only this thread's calls
        mode = dict(capture_error_mode='thread_local') if self._multi else {}
        with torch.cuda.graph(self._fwd_bwd, **mode):
            self.optimizer.zero_grad()
            self.loss, self.output = self.model(self.features)
            functional.backward(self.loss)
            if not self._multi:
                self.optimizer.step_captured()
        self.steps_done = warmup

    def _eager_step(self):
        self.optimizer.zero_grad()
        loss, _ = self.model(self.features)
        functional.backward(loss)
        self.optimizer.step()

    def load(self, features):
        """Copy a new batch (same keys, shapes and dtypes) into the captured buffers."""
        for key, value in features.items():
            if isinstance(value, torch.Tensor):
                self.features[key].copy_(value, non_blocking=True)

    def __call__(self):
        """One training step; returns the (device, 0-d) loss tensor of the captured step - valid until the next call."""
        self.optimizer.advance()
        self._fwd_bwd.replay()
        if self._multi:
            self.optimizer.exchange_gradients()        # the step's one RCCL all-reduce, outside the graph
            self.optimizer.step_captured()             # one kernel: launched directly
        self.steps_done += 1
        return self.loss


class GraphedStepCache(object):
    """Graph replay inside a training loop whose batches repeat a few shapes (``ExperimentBuilder(use_graphs=True)``).

    A batch whose (name, shape, dtype) signature is new runs eagerly - an ordinary training step, which also warms up allocator and
    workspaces for that shape; the second batch of a signature is captured into static copies of its tensors (capture runs
    nothing) and replayed, and every later one is copied into those buffers and replayed.  No step is ever run twice or skipped, so
    the loop trains exactly as the eager loop does (bit-identical, tests/test_gpu_parity.py)."""

    def __init__(self, model, optimizer, max_graphs=8):
        self.model, self.optimizer, self.max_graphs = model, optimizer, max_graphs
        self._seen = set()
        self._steps = {}

    @staticmethod
    def signature(features):
        return tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in features.items() if isinstance(v, torch.Tensor)))

    def step(self, features):
        """zero_grad, forward, backward, optimizer step on ``features``; returns (loss, output_features)."""
        key = self.signature(features)
        graphed = self._steps.get(key)
        if graphed is None and key in self._seen and len(self._steps) < self.max_graphs:
            static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in features.items()}
            graphed = self._steps[key] = GraphedTrainStep(self.model, self.optimizer, static, warmup=0)
        elif graphed is not None:
            graphed.load(features)
        if graphed is not None:
            for k, v in features.items():              # non-tensor entries (utterance names) follow the batch
                if not isinstance(v, torch.Tensor):
                    graphed.features[k] = v
            return graphed().clone(), graphed.output      # the loss buffer is rewritten by the next replay: hand out a copy
        self._seen.add(key)
        self.optimizer.zero_grad()
        loss, output = self.model(features)
        functional.backward(loss)
        self.optimizer.step()
        return loss, output
