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
so no second collective is needed.  The norm is reduced PIECE BY PIECE (``pieces``: one per
collective, in issue order): the optimizer waits for a collective, reduces its slice into a slot,
goes on to the next -- so when backward ends only the embeddings' collective and the 20-us partial
norm of its slice are still ahead of the update, not a pass over all 440 MB.  Without clipping
the optimizer's per-block launches wait only for the buckets that cover their block (``wait_range``).

Sparse word-embedding exchange (``sparse_embeddings=True``): a fine-tuning step touches at most
B*T of the 28996 rows of the word-embedding table, yet its gradient is 89 of the 98 MB of the one
collective that cannot hide behind backward.  With the flag the table leaves the dense exchange:
every rank sorts its token ids (at ``prepare`` time, during the forward), marks the first
occurrence of each, all-gathers (ids, rows of its local gradient at the first occurrences, zero
rows elsewhere) -- B*T*H elements per rank instead of V*H -- then clears its own touched rows and
adds every rank's rows IN RANK ORDER (each rank's ids are unique among its non-zero rows, so the
adds of one rank never collide; the order over ranks is the same everywhere): all replicas hold
bit-identical sums, as after an all-reduce.  The rest of the embeddings bucket (position / type /
image projections, 2 M parameters) stays a dense collective.  A step whose word-embedding gradient
is dense (the MLM task's tied decoder) passes ``token_ids=None`` and takes the dense path.
"""
import contextlib
import os

import torch
import torch.distributed as dist


class GradSync(object):
    def __init__(self, flat_grads, bucket_ranges, group=None, bucket_bytes=32 << 20, payload='fp32', word_table=None,
                 accum=1, token_capacity=None):
        """word_table: (start, rows, row_len) of the word-embedding gradient inside the LAST bucket (it must open that
        bucket) -- enables the sparse exchange for steps that announce their token ids to ``prepare``.
        accum: micro-batches per optimizer step (gradient_accumulation): the sparse exchange carries the ids of ALL of them,
        so its agreed capacity is sized for a full window (the first exchange of an epoch comes after ONE micro-batch:
        the reference's iteration-0 quirk, train_template.py:95-109).
        token_capacity: token ids ONE micro-batch can hold at most (batch_size x max_txt_len: the reference truncates and pads every
        caption to --max_txt_len, train_uniter.py:98, data/meme_dataset.py:170-177) -- the sparse exchange's capacity is then
        token_capacity x accum from the start, the same on every rank by construction: no agreement collective, no host
        synchronisation, and a later batch that holds more ids than the first cannot exceed it.  None: agreed at the first step."""
        if payload not in ('fp32', 'bf16'):
            raise ValueError("payload must be 'fp32' or 'bf16'")
        self.flat = flat_grads
        self.ranges = list(bucket_ranges)         # per ParamStore bucket: (start, end)
        self.word_table = None
        if word_table is not None:
            ws, V, H = word_table
            ls, le = self.ranges[-1]
            if ws != ls or ws + V * H > le:
                raise ValueError('word_table must open the last (embeddings) bucket')
            self.word_table = (ws, V, H)
            # the table becomes a range of its own (up to the 64-element boundary the next tensor starts on: the padding
            # holds zeros); what follows it in the bucket stays a dense collective
            we = (ws + V * H + 63) // 64 * 64
            self.ranges[-1:] = [(ws, we)] + ([(we, le)] if we < le else [])
        self._tokens = []                         # token ids of the micro-batches accumulated since the last exchange
        self._tokens_dense = False                # one of them had a dense word-embedding gradient
        self.accum = max(1, int(accum))
        # ids per rank and exchange: static (token_capacity x accum), or agreed over the ranks at the first sparse step
        self._cap = int(token_capacity) * self.accum if token_capacity else None
        self.cap_source = 'static' if token_capacity else 'agreed at the first step'
        self._micro = 0                           # micro-batches recorded since the last exchange
        # timing of the collectives (bench.py --gpus N: the `comm` block): events on the issuing / waiting streams
        self.timing = False
        self.timings = []                         # per collective: dict(bytes, issue_to_done_ms, exposed_ms, start, end)
        self._timing_open = []
        self.sparse_steps = 0                     # exchanges that took the sparse path (for tests / the bench line)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_bytes = bucket_bytes
        self.payload = payload
        self.comm = torch.empty_like(flat_grads, dtype=torch.bfloat16) if payload == 'bf16' else None
        # the embeddings (with a word table: the table and the rest of the bucket): never coalesced
        n_emb = 1 if self.word_table is None else len(self.ranges) - len(bucket_ranges) + 1
        self.solo = set(range(len(self.ranges) - n_emb, len(self.ranges))) if len(bucket_ranges) > 2 else set()
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
    def prepare(self, will_step=True, token_ids=None):
        """Call before backward.  Gradients are exchanged only on the micro-batch that
        steps; earlier micro-batches accumulate locally.
        token_ids (sparse word-embedding exchange only): the int64 ids this micro-batch looks up in the word-embedding
        table -- the rows its gradient can touch; None = the gradient of the table is dense this time."""
        # UNITER_DP_FORCE=1 exercises the collective path on a single rank (testing only)
        self.active = bool(will_step) and (self.world > 1 or os.environ.get('UNITER_DP_FORCE') == '1')
        self._inflight, self._pending, self._next, self.launched, self.launch_streams = [], None, 0, [], []
        exchanging = self.world > 1 or os.environ.get('UNITER_DP_FORCE') == '1'
        if self.word_table is not None and exchanging:      # (no exchange possible: nothing to remember -- the lists would only grow)
            if token_ids is None:
                self._tokens_dense = True
            else:
                self._tokens.append(token_ids.reshape(-1).to(torch.int64))
            self._micro += 1
            self._sorted = None
            if self.active and not self._tokens_dense:
                # sorted ids and first-occurrence marks now, next to the forward: nothing of it waits for the backward
                ids = self._tokens[0] if len(self._tokens) == 1 else torch.cat(self._tokens)
                if self._cap is None:             # once: the ranks agree on a capacity (the only host synchronisation)
                    # per micro-batch, times the micro-batches of a full accumulation window: the first exchange may come
                    # after fewer of them than the later ones
                    per = (ids.numel() + self._micro - 1) // max(1, self._micro)
                    cap = torch.tensor([per * max(self.accum, self._micro)], dtype=torch.int64, device=ids.device)
                    if self.world > 1:
                        dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=self.group)
                    self._cap = int(cap.item())
                if ids.numel() > self._cap:
                    raise ValueError('sparse embedding exchange: %d token ids in a step, capacity (%s) is %d (gradient_accumulation '
                                     '= %d passed to dp.attach; pass token_capacity = batch_size x max_txt_len to dp.attach to size it '
                                     'for the largest batch up front)' % (ids.numel(), self.cap_source, self._cap, self.accum))
                srt = ids.sort().values
                first = torch.ones_like(srt, dtype=torch.bool)
                first[1:] = srt[1:] != srt[:-1]
                if srt.numel() < self._cap:       # a short last batch: pad with id 0 / no row
                    pad = self._cap - srt.numel()
                    srt = torch.cat([srt, srt.new_zeros(pad)])
                    first = torch.cat([first, first.new_zeros(pad)])
                self._sorted = (srt, first)

    def pieces(self):
        """(start, end) of every collective issued this step, in issue order (flush_all first): the slices a consumer
        can wait for one by one (``wait_range``) -- the clip norm is reduced that way."""
        return [(s, e) for _, s, e, _ in self._inflight]

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
        self._tokens, self._tokens_dense, self._micro = [], False, 0

    def wait_range(self, lo, hi):
        """Make the CURRENT stream wait for the buckets that overlap flat[lo:hi] (and widen their bf16
        sums back into the fp32 gradient buffer, once)."""
        if not self.active:
            return
        for rec in self._inflight:
            work, s, e, done = rec
            if e <= lo or s >= hi or done:
                continue
            if work is None:                      # the sparse exchange: summed in place on the issuing stream
                rec[3] = True
                continue
            t = self._timing_of(s, e)
            if t is not None:
                t['w0'] = torch.cuda.Event(enable_timing=True)
                t['w0'].record()
            work.wait()
            if t is not None:
                t['w1'] = torch.cuda.Event(enable_timing=True)
                t['w1'].record()
            if self.comm is not None and not self.consumer_reads_comm:
                self.flat[s:e].copy_(self.comm[s:e])
            rec[3] = True

    # -- timing (bench.py) -----------------------------------------------------------
    def _timing_of(self, s, e):
        if not self.timing:
            return None
        return next((t for t in self._timing_open if t['start'] == s and t['end'] == e and 'w1' not in t), None)

    def collect_timings(self):
        """Close the records of the step(s) since the last call (synchronises): per collective its payload bytes, the time
        from its issue (event on the issuing stream, in front of the collective) to the point where the consumer's stream
        had it (event behind the consumer's wait), and the part of that the consumer's stream spent waiting (exposed)."""
        out = []
        if self.flat.is_cuda:
            torch.cuda.synchronize()
        for t in self._timing_open:
            if 'w1' not in t:
                continue
            out.append({'start': t['start'], 'end': t['end'], 'bytes': t['bytes'],
                        'issue_to_done_ms': t['e0'].elapsed_time(t['w1']), 'exposed_ms': t['w0'].elapsed_time(t['w1'])})
        self._timing_open = []
        self.timings.extend(out)
        return out

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
            self._finish_embeddings(stream)

    def _finish_embeddings(self, stream):
        sparse = self.word_table is not None and self._sorted is not None and not self._tokens_dense
        word_idx = None
        if sparse:
            ws = self.word_table[0]
            word_idx = next(i for i, (s, _) in enumerate(self.ranges) if s == ws)
        while self._next < len(self.ranges):
            if self._next == word_idx:
                self._flush(stream)
                self._next += 1                   # the table does not join the dense exchange this step
                continue
            self._add_bucket(self._next, stream)
        self._flush(stream)
        if sparse:
            self._exchange_rows(stream)
        self._tokens, self._tokens_dense, self._micro = [], False, 0

    def _exchange_rows(self, stream):
        """The word-embedding table's gradient: all-gather (sorted ids, rows at first occurrences), then the sum in rank
        order over cleared rows (module docstring)."""
        ws, V, H = self.word_table
        srt, first = self._sorted
        on_gpu = self.flat.is_cuda
        ctx = torch.cuda.stream(stream) if (stream is not None and on_gpu) else contextlib.nullcontext()
        with ctx:
            table = self.flat[ws:ws + V * H].view(V, H)
            rows = table.index_select(0, srt)
            rows.masked_fill_(~first.unsqueeze(1), 0.0)
            if self.comm is not None:
                rows = rows.to(torch.bfloat16)
            n = srt.numel()
            all_ids = torch.empty(self.world * n, dtype=srt.dtype, device=srt.device)
            all_rows = torch.empty(self.world * n, H, dtype=rows.dtype, device=rows.device)
            if self.world > 1 or (dist.is_initialized() and os.environ.get('UNITER_DP_FORCE') == '1'):
                # (UNITER_DP_FORCE: the collectives themselves also on a single rank -- tests on one GPU)
                w_ids = dist.all_gather_into_tensor(all_ids, srt, group=self.group, async_op=True)
                w_rows = dist.all_gather_into_tensor(all_rows, rows, group=self.group, async_op=True)
                w_ids.wait()
                w_rows.wait()
            else:
                all_ids.copy_(srt)
                all_rows.copy_(rows)
            table.index_fill_(0, srt, 0.0)
            for r in range(self.world):           # rank order: the same sequence of fp32 adds on every replica
                table.index_add_(0, all_ids[r * n:(r + 1) * n], all_rows[r * n:(r + 1) * n].to(torch.float32))
            we = (ws + V * H + 63) // 64 * 64
            if self.comm is not None:             # a consumer that reads the bf16 sums finds the table there as well
                self.comm[ws:we].copy_(self.flat[ws:we])
            self.launch_streams.append(torch.cuda.current_stream().cuda_stream if on_gpu else None)
        self._inflight.append([None, ws, we, False])
        self.launched.append((ws, we))
        self.sparse_steps += 1
        self.last_sparse_rows = n

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
            if self.timing and on_gpu:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
                self._timing_open.append({'start': s, 'end': e, 'bytes': self._payload_bytes(e - s), 'e0': e0})
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.launch_streams.append(torch.cuda.current_stream().cuda_stream if on_gpu else None)
        self._inflight.append([work, s, e, False])
        self.launched.append((s, e))


DEFAULT_CU_RESERVE = 16      # CUs left to RCCL's kernels beside the backward pass (world > 1); UNITER_DP_CU_RESERVE overrides


def cu_reserve_default(world):
    """CUs the persistent matrix kernels leave free while a gradient exchange is attached: UNITER_DP_CU_RESERVE, else 16 with more
    than one rank (RCCL runs one workgroup per channel; its kernels -- 256 threads, ~100 registers per lane: they do not fit beside
    a persistent 144-KB GEMM workgroup on the same CU -- otherwise take CUs as GEMM workgroups exit and strand the launch's last
    workgroups behind them), 0 on one rank.  A starting point, not a finding: ``pick_cu_reserve`` measures it on the node."""
    e = os.environ.get('UNITER_DP_CU_RESERVE')
    if e is not None:
        try:
            return max(0, min(128, int(e)))
        except ValueError:
            return 0
    return DEFAULT_CU_RESERVE if world > 1 else 0


def prepare_rccl_env(world, env=None):
    """Before the process group exists.  RCCL's channel count is RCCL's by default: how many workgroups it needs to fill seven xGMI
    links is tuned per topology by its authors, a gradient exchange that is starved of channels is exposed at the end of every
    backward pass (the whole job's scaling), while an exchange that takes CUs from the matrix kernels costs them a few per cent --
    and the reserve is measured against whatever RCCL opens (``pick_cu_reserve``).  UNITER_DP_CAP_CHANNELS=1 caps the channels at
    the reserve instead (NCCL_MAX_NCHANNELS, only when the caller left it alone): RCCL then asks for exactly the CUs that were left
    to it -- the configuration to try when ``comm.collectives`` show the exchange hidden with room to spare.  Returns the reserve."""
    env = os.environ if env is None else env
    r = cu_reserve_default(world)
    if r > 0 and env.get('UNITER_DP_CAP_CHANNELS') == '1':
        env.setdefault('NCCL_MAX_NCHANNELS', str(r))
    return r


def pick_cu_reserve(sync, encoder, one_step, candidates=None, steps=6, warm=2, budget_s=None):
    """Measure instead of believing: run `warm` + `steps` training steps (``one_step()``: forward, backward with the exchange, update)
    with each candidate number of reserved CUs (default: the reserve in force, 0, 16, 48), take the slowest rank's time for each,
    keep the fastest candidate on every rank (ties: the earlier one) and return {'picked': r, 'candidates': [{'cu_reserve': r, 'ms_per_step': t}, ...]}.  What the right
    reserve is depends on how many channels RCCL opens on the node's topology and on how long its kernels sit beside the matrix
    kernels -- neither can be known before the first exchange has run on the real links.  The steps are ordinary steps (their
    updates count); results do not depend on the reserve (tests/test_model_gpu.py::test_cu_reserve_for_a_gradient_exchange...).
    A failure on any rank leaves the attach-time reserve in place on all of them ('error' in the result).  One rank: no-op, None.
    budget_s: wall-time budget -- after every candidate the ranks agree (one 8-byte MAX all-reduce) on the time spent so far, and
    the candidates that do not fit are skipped on every rank alike ('skipped' in the result; the first candidate always runs).  The
    result carries 'wall_s', the slowest rank's time inside this function."""
    import time
    world = sync.world if sync is not None else 1
    if sync is None or world <= 1:
        return None
    start = int(getattr(sync, 'cu_reserve', 0))
    if candidates is None:
        # the reserve in force, none, and three times the default (RCCL's own channel count on eight GPUs is several dozen)
        candidates = [start] + [c for c in (0, DEFAULT_CU_RESERVE, 3 * DEFAULT_CU_RESERVE) if c != start]
    dev = sync.flat.device
    use_cuda = dev.type == 'cuda'

    def fence():
        dist.barrier(group=sync.group)
        if use_cuda:
            torch.cuda.synchronize(dev)

    times, ok, err = [], 1, None
    t_in = time.perf_counter()
    tried = list(candidates)
    try:
        for i, c in enumerate(candidates):
            encoder.cu_reserve = int(c)
            for _ in range(warm):
                one_step()
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                one_step()
            fence()
            times.append((time.perf_counter() - t0) / steps * 1e3)
            if budget_s is not None and i + 1 < len(candidates):
                # every rank must stop at the same candidate: the slowest rank's clock decides
                el = torch.tensor([time.perf_counter() - t_in], dtype=torch.float64, device=dev if use_cuda else 'cpu')
                dist.all_reduce(el, op=dist.ReduceOp.MAX, group=sync.group)
                if float(el.item()) * (i + 2) / (i + 1) > float(budget_s):      # (the next candidate would not fit)
                    tried = list(candidates[:i + 1])
                    break
    except Exception as e:                                       # noqa: BLE001
        ok, err = 0, repr(e)
    skipped = [int(c) for c in candidates[len(tried):]]
    candidates = tried
    res = {'picked': start, 'candidates': []}
    if skipped:
        res['skipped'] = skipped
    try:
        buf = torch.zeros(1 + len(candidates), dtype=torch.float64, device=dev if use_cuda else 'cpu')
        buf[0] = -float(ok)                                      # MAX over ranks of -ok: 0 as soon as one rank failed
        for i, t in enumerate(times):
            buf[1 + i] = t
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=sync.group)
        if buf[0].item() < 0:
            ts = [float(x) for x in buf[1:].tolist()]
            best = min(range(len(ts)), key=lambda i: (ts[i], i))
            res['picked'] = int(candidates[best])
            res['candidates'] = [{'cu_reserve': int(c), 'ms_per_step': round(t, 3)} for c, t in zip(candidates, ts)]
        else:
            res['error'] = err or 'failed on another rank'
    except Exception as e:                                       # noqa: BLE001
        res['error'] = err or repr(e)
    sync.cu_reserve = res['picked']
    encoder.cu_reserve = res['picked']
    sync.collect_timings()
    sync.timings = []
    try:
        w = torch.tensor([time.perf_counter() - t_in], dtype=torch.float64, device=dev if use_cuda else 'cpu')
        dist.all_reduce(w, op=dist.ReduceOp.MAX, group=sync.group)
        res['wall_s'] = round(float(w.item()), 3)
    except Exception:                                            # noqa: BLE001
        res['wall_s'] = round(time.perf_counter() - t_in, 3)
    if budget_s is not None:
        res['budget_s'] = float(budget_s)
    return res


XGMI_LINK_GBS = 153.0        # one xGMI link, one direction (MI355X: 7 links per GPU to its 7 peers)


def predicted_exposed_ms(last_bytes, world, links=1):
    """What the LAST collective of a step (the embeddings': the one no backward compute is left to hide) costs on the wire if it
    runs as a ring all-reduce: every rank sends and receives 2 (N - 1) / N of the payload, at the rate of `links` xGMI links
    (1 = one ring on one link per hop, the pessimistic case; 7 = all links of the fully connected node).  Arithmetic, to read the
    measured ``exposed_ms_per_step`` against -- not a measurement."""
    if world <= 1 or last_bytes <= 0:
        return 0.0
    return 2.0 * (world - 1) / world * last_bytes / (XGMI_LINK_GBS * 1e9 * max(1, links)) * 1e3


def attach(model, group=None, bucket_bytes=32 << 20, payload=None, sparse_embeddings=None, accum=1, cu_reserve=None,
           token_capacity=None):
    """Wire a GradSync to a MemeUniter, UniterForPretraining or UniterModel and return it.
    payload None: 'bf16' when the encoder runs in the bf16 precision mode, else 'fp32'.
    accum: the trainer's gradient_accumulation (sizes the sparse exchange for a full window of micro-batches).
    sparse_embeddings (None: UNITER_DP_SPARSE_EMB=1): exchange the touched rows of the word-embedding gradient instead
    of the table (GradSync docstring); the trainer then passes each micro-batch's token ids to ``prepare``.
    token_capacity: batch_size x max_txt_len -- sizes the sparse exchange statically (GradSync docstring)."""
    store = model.param_store() if hasattr(model, 'param_store') else None
    if store is None:
        from .model import ensure_store
        store = ensure_store(model)
    um = getattr(model, 'uniter_model', None) or getattr(model, 'uniter', None) or model   # MemeUniter / UniterForPretraining / UniterModel
    if payload is None:
        payload = 'bf16' if getattr(um, 'precision', 'fp32') == 'bf16' else 'fp32'
    if sparse_embeddings is None:
        sparse_embeddings = os.environ.get('UNITER_DP_SPARSE_EMB') == '1'
    word_table = None
    if sparse_embeddings:
        name = next((n for n in store.names if n.endswith('embeddings.word_embeddings.weight')), None)
        if name is not None and store.bucket_ranges and store.offsets[name] == store.bucket_ranges[-1][0]:
            V, H = store.params[name].shape
            word_table = (store.offsets[name], int(V), int(H))
    gs = GradSync(store.flat_grads, store.bucket_ranges, group=group, bucket_bytes=bucket_bytes, payload=payload,
                  word_table=word_table, accum=accum, token_capacity=token_capacity)
    um._grad_hook = gs.hook
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    gs.cu_reserve = cu_reserve_default(world) if cu_reserve is None else int(cu_reserve)
    um.cu_reserve = gs.cu_reserve
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
