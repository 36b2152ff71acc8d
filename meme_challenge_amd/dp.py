"""Data-parallel gradient exchange: one process per GPU, bucketed sum
all-reduce over RCCL/xGMI overlapped with backward.

Replaces the reference's single-process nn.DataParallel (train_template.py:58-59:
per-iteration parameter broadcast + gradient reduce rooted at GPU 0).  Here
every rank keeps a resident replica; the only exchange is the gradient sum.

The flat gradient buffer is laid out in backward-completion order
(ParamStore): [head | layer nl-1 | ... | layer 0 | embeddings], so each bucket
is ONE contiguous slice and its all-reduce is issued the moment the backward
schedule has finished that layer -- on the stream that produced the gradients
(the wgrad side stream), so RCCL starts behind exactly the kernels it depends on
while the dgrad chain of the next layer keeps the CUs busy.  Consecutive layers
are coalesced up to ``bucket_bytes`` so that each collective is large enough to
run at link rate (xGMI is point-to-point: a ring step is bound by one ~153 GB/s
link, so few large transfers beat many small ones).

Averaging (1/world) is folded into the optimizer's ``grad_scale``; the gradient
norm for clipping is computed on the reduced buffer, identical on every rank,
so no second collective is needed.
"""
import contextlib
import os

import torch
import torch.distributed as dist


class GradSync(object):
    def __init__(self, flat_grads, bucket_ranges, group=None, bucket_bytes=64 << 20):
        self.flat = flat_grads
        self.ranges = list(bucket_ranges)         # per ParamStore bucket: (start, end)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.active = True
        self._works = []
        self._pending = None                      # (start, end) accumulated but not yet launched
        self._next = 0
        self.launched = []                        # (start, end) slices issued this step (for tests)

    # -- driven by the trainer ---------------------------------------------------
    def prepare(self, will_step=True):
        """Call before backward.  Gradients are exchanged only on the micro-batch that
        steps; earlier micro-batches accumulate locally."""
        # UNITER_DP_FORCE=1 exercises the collective path on a single rank (testing only)
        self.active = bool(will_step) and (self.world > 1 or os.environ.get('UNITER_DP_FORCE') == '1')
        self._works, self._pending, self._next, self.launched = [], None, 0, []

    def finish(self):
        """Call after backward, before the optimizer step: flush + wait."""
        if not self.active:
            return
        # any bucket the backward did not announce (e.g. frozen parts) is sent now
        while self._next < len(self.ranges):
            self._add_bucket(self._next, None)
        self._flush(None)
        for w in self._works:
            w.wait()
        self._works = []

    # -- driven by the model's backward schedule -----------------------------------
    def hook(self, kind, index, stream):
        """kind: 'begin' (head/pooler grads are complete), 'layer' (layer `index` complete,
        called for nl-1 .. 0), 'embed' (everything complete)."""
        if not self.active:
            return
        if kind == 'begin':
            self._add_bucket(0, stream)
        elif kind == 'layer':
            self._add_bucket(self._next, stream)
        elif kind == 'embed':
            while self._next < len(self.ranges):
                self._add_bucket(self._next, stream)
            self._flush(stream)

    # -- internals ------------------------------------------------------------------
    def _add_bucket(self, i, stream):
        if i != self._next:
            return
        s, e = self.ranges[i]
        self._next += 1
        if self._pending is None:
            self._pending = (s, e)
        elif self._pending[1] == s:
            self._pending = (self._pending[0], e)
        else:
            self._flush(stream)
            self._pending = (s, e)
        if (self._pending[1] - self._pending[0]) * self.flat.element_size() >= self.bucket_bytes:
            self._flush(stream)

    def _flush(self, stream):
        if self._pending is None:
            return
        s, e = self._pending
        self._pending = None
        buf = self.flat[s:e]
        ctx = torch.cuda.stream(stream) if (stream is not None and buf.is_cuda) else contextlib.nullcontext()
        with ctx:
            self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.launched.append((s, e))


def attach(model, group=None, bucket_bytes=64 << 20):
    """Wire a GradSync to a MemeUniter, UniterForPretraining or UniterModel and return it."""
    store = model.param_store() if hasattr(model, 'param_store') else None
    if store is None:
        from .model import ensure_store
        store = ensure_store(model)
    gs = GradSync(store.flat_grads, store.bucket_ranges, group=group, bucket_bytes=bucket_bytes)
    um = getattr(model, 'uniter_model', None) or getattr(model, 'uniter', None) or model   # MemeUniter / UniterForPretraining / UniterModel
    um._grad_hook = gs.hook
    return gs


def broadcast_parameters(model, src=0, group=None):
    """One-off: make every replica start from rank `src`'s weights."""
    store = model.param_store()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(store.flat_params, src=src, group=group)
