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
while the dgrad chain of the next layer keeps the CUs busy.

Bucket policy (xGMI is point-to-point: a ring step is bound by one ~153 GB/s link, so few large
transfers beat many small ones, but whatever is still in flight when backward ends is exposed):
  * consecutive layers are coalesced until a collective carries >= ``bucket_bytes`` of PAYLOAD;
  * layer 0 is flushed with the layers -- it does not wait for the embeddings;
  * the embeddings (89 MB of word-embedding gradient in fp32) are their own collective, issued
    the moment the embedding backward has finished: the only exchange that cannot hide behind
    backward compute.

Payload (``payload='bf16'``): the slice is rounded to bf16 into a communication buffer, summed by
RCCL in bf16 and read by the fused optimizer (and the clip norm) straight from there, widened on the
fly (any other consumer gets it widened back into the fp32 buffer) -- half the bytes on every
link; local accumulation over micro-batches, the clip norm and the optimizer stay fp32.  The
default is 'fp32' (bit-identical to a single-process run on the concatenated batch, up to the
summation order); the bf16 precision mode of the model selects 'bf16'.

Averaging (1/world) is folded into the optimizer's ``grad_scale``; the gradient
norm for clipping is computed on the reduced buffer, identical on every rank,
so no second collective is needed.  Without clipping the optimizer's per-block launches wait
only for the buckets that cover their block (``wait_range``).
"""
import contextlib
import os

import torch
import torch.distributed as dist


class GradSync(object):
    def __init__(self, flat_grads, bucket_ranges, group=None, bucket_bytes=32 << 20, payload='fp32'):
        if payload not in ('fp32', 'bf16'):
            raise ValueError("payload must be 'fp32' or 'bf16'")
        self.flat = flat_grads
        self.ranges = list(bucket_ranges)         # per ParamStore bucket: (start, end)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.payload = payload
        self.comm = torch.empty_like(flat_grads, dtype=torch.bfloat16) if payload == 'bf16' else None
        self.solo = {len(self.ranges) - 1} if len(self.ranges) > 2 else set()     # the embeddings: never coalesced
        # True (set by trainer.sync_step for the fused optimizer): the consumer reads the reduced bf16 sums straight from
        # `comm` (uniter_adam_step_g16 / uniter_grad_sumsq_bf16), wait_range does not widen them back into the fp32 buffer
        self.consumer_reads_comm = False
        self.active = True
        self._inflight = []                       # [work, start, end, unpacked]
        self._pending = None                      # (start, end) accumulated but not yet launched
        self._next = 0
        self.launched = []                        # (start, end) slices issued this step (for tests)
        self.launch_streams = []                  # the stream each of them was issued on (for tests)

    # -- driven by the trainer ---------------------------------------------------
    def prepare(self, will_step=True):
        """Call before backward.  Gradients are exchanged only on the micro-batch that
        steps; earlier micro-batches accumulate locally."""
        # UNITER_DP_FORCE=1 exercises the collective path on a single rank (testing only)
        self.active = bool(will_step) and (self.world > 1 or os.environ.get('UNITER_DP_FORCE') == '1')
        self._inflight, self._pending, self._next, self.launched, self.launch_streams = [], None, 0, [], []

    def finish(self):
        """Call after backward, before anything reads the whole gradient buffer (clip norm, a
        single-launch optimizer step): flush + wait for every bucket."""
        if not self.active:
            return
        self.flush_all()
        self.wait_range(0, self.flat.numel())

    def flush_all(self):
        """Issue whatever the backward did not announce (e.g. frozen parts)."""
        if not self.active:
            return
        while self._next < len(self.ranges):
            self._add_bucket(self._next, None)
        self._flush(None)

    def wait_range(self, lo, hi):
        """Make the CURRENT stream wait for the buckets that overlap flat[lo:hi] (and widen their bf16
        sums back into the fp32 gradient buffer, once)."""
        if not self.active:
            return
        for rec in self._inflight:
            work, s, e, done = rec
            if e <= lo or s >= hi or done:
                continue
            work.wait()
            if self.comm is not None and not self.consumer_reads_comm:
                self.flat[s:e].copy_(self.comm[s:e])
            rec[3] = True

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
            if index == 0:
                self._flush(stream)               # layer 0 goes now, not together with the embeddings
        elif kind == 'embed':
            while self._next < len(self.ranges):
                self._add_bucket(self._next, stream)
            self._flush(stream)

    # -- internals ------------------------------------------------------------------
    def _payload_bytes(self, n):
        return n * (2 if self.payload == 'bf16' else self.flat.element_size())

    def _add_bucket(self, i, stream):
        if i != self._next:
            return
        s, e = self.ranges[i]
        self._next += 1
        if i in self.solo:
            self._flush(stream)
        if self._pending is None:
            self._pending = (s, e)
        elif self._pending[1] == s:
            self._pending = (self._pending[0], e)
        else:
            self._flush(stream)
            self._pending = (s, e)
        if i in self.solo or self._payload_bytes(self._pending[1] - self._pending[0]) >= self.bucket_bytes:
            self._flush(stream)

    def _flush(self, stream):
        if self._pending is None:
            return
        s, e = self._pending
        self._pending = None
        on_gpu = self.flat.is_cuda
        ctx = torch.cuda.stream(stream) if (stream is not None and on_gpu) else contextlib.nullcontext()
        with ctx:
            if self.comm is not None:
                buf = self.comm[s:e]
                buf.copy_(self.flat[s:e])         # round to nearest even, on the stream that produced the gradients
            else:
                buf = self.flat[s:e]
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.launch_streams.append(torch.cuda.current_stream().cuda_stream if on_gpu else None)
        self._inflight.append([work, s, e, False])
        self.launched.append((s, e))


def attach(model, group=None, bucket_bytes=32 << 20, payload=None):
    """Wire a GradSync to a MemeUniter, UniterForPretraining or UniterModel and return it.
    payload None: 'bf16' when the encoder runs in the bf16 precision mode, else 'fp32'."""
    store = model.param_store() if hasattr(model, 'param_store') else None
    if store is None:
        from .model import ensure_store
        store = ensure_store(model)
    um = getattr(model, 'uniter_model', None) or getattr(model, 'uniter', None) or model   # MemeUniter / UniterForPretraining / UniterModel
    if payload is None:
        payload = 'bf16' if getattr(um, 'precision', 'fp32') == 'bf16' else 'fp32'
    gs = GradSync(store.flat_grads, store.bucket_ranges, group=group, bucket_bytes=bucket_bytes, payload=payload)
    um._grad_hook = gs.hook
    return gs


def broadcast_parameters(model, src=0, group=None):
    """One-off: make every replica start from rank `src`'s weights."""
    store = model.param_store()
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(store.flat_params, src=src, group=group)


def shard_indices(indices, rank, world):
    """Rank `rank`'s share of a sample order that every rank computed identically (same seed): every
    world-th index, the list padded by wrapping around so that all ranks run the same number of
    iterations (one collective per step on every rank)."""
    indices = list(indices)
    if world <= 1:
        return indices
    per = (len(indices) + world - 1) // world
    padded = indices + indices[:per * world - len(indices)]
    return padded[rank::world]
