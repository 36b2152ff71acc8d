"""Training-step semantics of the reference trainer on the HIP path.

Mirrors:
  nn.BCEWithLogitsLoss(pos_weight)            train_template.py:64-65
  TrainerTemplate.calculate_loss              train_template.py:95-126
  TrainerTemplate.average_gradients           train_template.py:89-92
  get_optimizer (param-group split, Adam/AdamW) utils/optim_utils.py:9-46
  init_scheduler (warmup / warmup_cosine / step / multi_step)  train_template.py:72-82
  TrainerUniter.{train,eval,test}_iter_step    train_uniter.py:58-81

The optimizer step (gradient averaging, global-norm clip, Adam/AdamW update,
zero_grad) is ONE fused kernel over the flat parameter buffer
(csrc/optim.hip); the gradient norm stays on the device, so a training
iteration needs no host synchronisation (the reference does `.item()` /
`.cpu()` every iteration, train_template.py:121-124).
"""
import math
import os

import torch

from . import _lib
from ._lib import check, ptr, UniterHipError
from .model import CHUNK, ensure_store


# --------------------------------------------------------------------------- #
# loss
# --------------------------------------------------------------------------- #
class _BceLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, pos_weight):
        _lib.require_gpu_tensor(logits, torch.float32, 'logits')
        x = logits.reshape(-1).contiguous()
        y = labels.reshape(-1).to(torch.int64).contiguous()
        if x.numel() != y.numel():
            raise ValueError('logits/labels size mismatch')
        B = x.numel()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        probs = torch.empty(B, dtype=torch.float32, device=x.device)
        dlog = torch.empty(B, dtype=torch.float32, device=x.device)
        check(_lib.lib().uniter_bce_logits(ptr(x), ptr(y), float(pos_weight), ptr(loss), ptr(probs),
                                           ptr(dlog), 1.0, B, _lib.cur_stream()), 'uniter_bce_logits')
        ctx.save_for_backward(dlog)
        ctx.shape = logits.shape
        ctx.mark_non_differentiable(probs)
        ctx.set_materialize_grads(False)           # no zero tensor for the probabilities' (absent) gradient
        return loss, probs

    @staticmethod
    def backward(ctx, dloss, _dprobs):
        (dlog,) = ctx.saved_tensors
        if dloss is None:
            return None, None, None
        unit = _UNIT.get(dlog.device)
        if unit is not None and dloss.data_ptr() == unit.data_ptr():
            return dlog.view(ctx.shape), None, None          # loss.backward(unit_gradient(..)): d loss = 1, nothing to multiply
        return (dlog * dloss).view(ctx.shape), None, None


_UNIT = {}


def unit_gradient(device):
    """A cached scalar 1.0 to pass as loss.backward(gradient=...): saves the ones_like fill autograd would launch for the
    root and (recognised by its address in _BceLogitsFn.backward) the multiply by it -- two launch-latency-sized kernels
    in the host-bound stretch between forward and backward."""
    device = torch.device(device)
    t = _UNIT.get(device)
    if t is None:
        t = _UNIT[device] = torch.ones((), dtype=torch.float32, device=device)
    return t


def bce_with_logits_loss(logits, labels, pos_weight=1.0, return_probs=False):
    """BCEWithLogitsLoss(pos_weight=[w]) with mean reduction; labels are cast to
    float as train_template.py:99 does.  Also yields sigmoid(logits)."""
    loss, probs = _BceLogitsFn.apply(logits, labels, pos_weight)
    return (loss, probs) if return_probs else loss


# --------------------------------------------------------------------------- #
# optimizer
# --------------------------------------------------------------------------- #
NO_DECAY = ('bias', 'LayerNorm.bias', 'LayerNorm.weight')      # utils/optim_utils.py:16


def no_decay(name):
    return any(nd in name for nd in NO_DECAY)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (coupled L2) / AdamW over the model's flat buffers.

    ``param_groups`` keeps the reference's two-group layout (decay / no-decay) so
    LR schedulers written against torch optimizers keep working; the arithmetic is
    one kernel launch.  ``step()`` also performs ``average_gradients`` (``grad_scale``)
    and ``clip_grad_norm_`` (``max_grad_norm``) when asked to, and zeroes the
    gradients (the reference calls ``zero_grad`` right after ``step``).
    """

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3, adamw=False):
        self.model = model
        self.store = model.param_store() if hasattr(model, 'param_store') else ensure_store(model)
        named = list(model.named_parameters())
        decay = [p for n, p in named if not no_decay(n)]
        nodecay = [p for n, p in named if no_decay(n)]
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__([{'params': decay, 'weight_decay': weight_decay},
                          {'params': nodecay, 'weight_decay': 0.0}], defaults)
        self.adamw = bool(adamw)
        self.step_count = 0
        st = self.store
        self.exp_avg = torch.zeros_like(st.flat_params)
        self.exp_avg_sq = torch.zeros_like(st.flat_params)
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=st.device)
        self._parts = None          # per-slice partial sums of the clip norm (data parallel: one per collective)
        # the clip norm taken during the backward pass (attach_norm_hooks): unreduced partial sums per bucket
        self._np_buf = None
        self._np_blocks = 0         # partial sums written by the backward that just ran (0 = not armed)
        self._np_seen = None
        n_ws = _lib.lib().uniter_grad_sumsq_ws_bytes(st.numel)
        self._ws = torch.empty(n_ws, dtype=torch.uint8, device=st.device)
        self._ws_bytes = n_ws
        self._flags_key = None
        self._flags = None
        self._flags_cache = {}      # touched-set -> device flags (multitask training alternates between a few sets)
        # set to the UniterModel to overlap the update with the next forward (see step()); anything
        # else that reads parameters on the current stream must call join() first
        self.overlap_encoder = None
        # set to the UniterModel: zero_grad (fused into step) skips the encoder layers' weight gradients -- four fifths of all
        # parameters -- and the next backward pass writes them with `=` instead of `+=` (uniter_model_set_wgrad_overwrite):
        # 28 instead of 32 bytes per parameter, no read-modify-write in the weight-gradient epilogues.  Until that backward
        # pass those .grad tensors hold the previous step's values (ParamStore.wgrad_stale); the trainers of this package
        # never read them in between (the reference calls zero_grad right after step, train_template.py:105-107)
        self.lazy_zero_encoder = None
        # grid of the blocks that share the chip with the next forward pass.  256 (one workgroup per CU) where the forward's kernels
        # can share a CU with it (fp32, bf16: DESIGN.md section 4 "Streams"); in the fp32x3 mode a persistent 144-KB, 504-of-512-register
        # GEMM workgroup cannot start on a CU while an optimizer workgroup sits there -- the blocks should be OUT OF THE WAY fast, not
        # thin: round 6 measured 9.52 ms (256), 9.42 (512), 9.35 (1024), 9.47 (no overlap at all) per step, bf16 unchanged
        # (profiles/r06_adam_wgs_ab.txt).  UNITER_ADAM_OVERLAP_WGS fixes it for every mode
        self._overlap_wgs_env = os.environ.get('UNITER_ADAM_OVERLAP_WGS')
        self._overlap_wgs = int(self._overlap_wgs_env) if self._overlap_wgs_env else None
        self._pending = None
        self._plan_cache = None
        # the word-embedding table's update split by rows (round 6, uniter_adam_step_rows): note_tokens / early_word_update / step
        # (built, bit-identical, and OFF by default: same-box A/B 9.53 ms with and without in fp32x3, 4.39 -> 4.44 ms in bf16 --
        # profiles/r06_word_rows_ab.txt: the 100 us the head of the next forward pass no longer waits for are CU-time the ahead-of-time
        # launch takes from the forward pass it runs beside, and behind the table the next forward's own first kernels bound the head)
        self.split_word_rows = os.environ.get('UNITER_ADAM_WORD_ROWS', '0') == '1'
        self._rowmask = None        # one byte per row of the table: 1 = a token of the micro-batches since the last step looks it up
        self._rows_noted = False    # every micro-batch since the last step announced its ids (none had a dense table gradient)
        self._early = None          # the rows without a gradient were updated ahead: (step_count, lr, betas, eps, wd, adamw, event)
        self._word_cache = None
        self._rowmask_clear = None  # event behind the mask's clearing (side stream)
        self._rowmask_ready = None  # event behind the mask's last fill (the stream note_tokens ran on)

    @property
    def overlap_workgroups(self):
        if self._overlap_wgs is not None:
            return self._overlap_wgs
        enc = self.overlap_encoder
        return 1024 if (enc is not None and getattr(enc, 'precision', 'fp32') == 'fp32x3') else 256

    @overlap_workgroups.setter
    def overlap_workgroups(self, v):
        self._overlap_wgs = None if v is None else int(v)

    def _overlap_plan(self, enc):
        """(head ranges, [embeddings, layer 0, ..] ranges) in flat-buffer elements, or None when the
        store's buckets are not [head | layer nl-1 .. 0 | embeddings]."""
        if self._plan_cache is None:
            st, nl = self.store, enc.config.num_hidden_layers
            r = list(st.bucket_ranges)
            ok = len(r) in (nl + 1, nl + 2)
            head = r[:len(r) - nl - 1] if ok else []
            blocks = [r[-1]] + r[-2:-2 - nl:-1] if ok else []
            covered = sorted(head + blocks)
            ok = ok and covered[0][0] == 0 and covered[-1][1] == st.numel and \
                all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
            word = None
            if ok:      # the word-embedding table, when it opens the embeddings' bucket: its update is a launch of its own
                name = next((n for n in st.names if n.endswith('embeddings.word_embeddings.weight')), None)
                if name is not None and st.offsets[name] == blocks[0][0]:
                    end = st.offsets[name] + (st.params[name].numel() + CHUNK - 1) // CHUNK * CHUNK
                    if end < blocks[0][1]:
                        word = (st.offsets[name], end)
            self._plan_cache = (head, blocks, word) if ok else False
        return self._plan_cache or None

    NORM_BLOCKS = 256           # workgroups (= partial sums) per layer slice, on the side stream beside the input-gradient chain
    NORM_BLOCKS_LAST = 2048     # the embeddings' slice runs alone behind the backward pass: the whole chip
    FUSED_SLOTS = 2048          # fp32x3 / bf16: slots a layer's weight-gradient launch writes itself (one per compute wave: up to 8 x 256)

    def attach_norm_hooks(self, encoder):
        """Take the clip norm (clip_grad_norm_, train_template.py:104) slice by slice DURING the backward pass: the
        encoder's backward schedule announces every bucket of the flat gradient buffer the moment it is final (head,
        layer nl-1 .. 0 on the weight-gradient stream, embeddings on the main stream) and this optimizer reduces the
        bucket into partial sums right there -- on a stream that idles half of each layer anyway.  `step` then only
        joins the partial sums (one 1-workgroup launch) instead of streaming all 440 MB of gradients once more
        behind the backward pass.  Single process only: with a data-parallel exchange attached the norm is that of the
        REDUCED gradients (dp.GradSync.pieces)."""
        if getattr(encoder, '_grad_hook', None) is not None:
            return False
        st = self.store
        nb = len(st.bucket_ranges)
        nl = encoder.config.num_hidden_layers
        if nb not in (nl + 1, nl + 2):
            return False
        first_layer = nb - nl - 1            # index of layer nl-1's bucket (0 when there is no head bucket)
        # partial sums per bucket: [head (optional) | layer 0 .. nl-1 (uniform stride: the fp32x3 backward writes layer l's own
        # partial sums at l * stride, uniter_model_set_norm_partials) | embeddings]
        lstride = max(self.NORM_BLOCKS, self.FUSED_SLOTS)
        head_n = self.NORM_BLOCKS if first_layer == 1 else 0
        self._np_buf = torch.zeros(head_n + nl * lstride + self.NORM_BLOCKS_LAST, dtype=torch.float64, device=st.device)
        layers_view = self._np_buf[head_n:head_n + nl * lstride]
        self._np_blocks, self._np_seen = 0, set()
        lib = _lib.lib()
        state = {'fused': False, 'per_layer': 0}

        def region(k):
            """(offset in doubles, slots) of bucket k's partial sums"""
            if k < first_layer:
                return 0, self.NORM_BLOCKS
            if k == nb - 1:
                return head_n + nl * lstride, self.NORM_BLOCKS_LAST
            layer = nl - 1 - (k - first_layer)
            return head_n + layer * lstride, self.NORM_BLOCKS

        def reduce_bucket(k, stream):
            lo, hi = st.bucket_ranges[k]
            off, n = region(k)
            sp = stream.cuda_stream if stream is not None else _lib.cur_stream()
            check(lib.uniter_grad_sumsq_part(st.flat_grads.data_ptr() + 4 * lo, None, hi - lo,
                                             self._np_buf.data_ptr() + 8 * off, n, sp), 'uniter_grad_sumsq_part')
            self._np_seen.add(k)

        def hook(kind, index, stream):
            if kind == 'begin':
                self._np_seen = set()
                self._np_blocks = 0
                # fp32x3: every layer's weight-gradient launch leaves the layer's share itself (bit-reproducible slots); any other
                # precision: one reduction launch per layer bucket on the weight-gradient stream.  A switch of precision between
                # two backward passes (bench.py's native-fp32 leg) changes which slots are written: clear the stale ones
                per_layer = encoder.norm_partials_per_layer()
                fused = 0 < per_layer <= lstride
                #  the slot COUNT matters too: uniter_sumsq_combine joins the whole buffer, and a launch on a smaller
                # grid (a CU reserve, a capped grid, another batch shape) leaves the larger grid's tail slots behind
                if fused != state['fused'] or (fused and per_layer != state['per_layer']):
                    layers_view.zero_()
                    state['fused'] = fused
                state['per_layer'] = per_layer
                if first_layer == 1:
                    reduce_bucket(0, stream)                 # head / pooler: final before the encoder's backward starts
            elif kind == 'layer':
                if state['fused']:
                    self._np_seen.add(first_layer + (nl - 1 - index))
                else:
                    reduce_bucket(first_layer + (nl - 1 - index), stream)
            elif kind == 'embed':
                reduce_bucket(nb - 1, stream)
                if len(self._np_seen) == nb:
                    self._np_blocks = self._np_buf.numel()   # armed: every bucket of THIS backward is in

        encoder._grad_hook = hook
        encoder._norm_parts = (layers_view, lstride)
        self._norm_hook = hook
        return True

    # -- the word-embedding table, row by row ---------------------------------------------------------------------------
    def _word_table(self):
        """(offset, rows, row length) of the word-embedding table in the flat buffers, or None (no such tensor, a row length
        that is no multiple of the optimizer's 64-element chunks, or the split switched off)."""
        if not self.split_word_rows:
            return None
        if self._word_cache is None:
            st = self.store
            name = next((n for n in st.names if n.endswith('embeddings.word_embeddings.weight')), None)
            wt = False
            if name is not None:
                V, H = st.params[name].shape
                if int(H) % CHUNK == 0 and st.offsets[name] % CHUNK == 0:
                    wt = (st.offsets[name], int(V), int(H), name)
            self._word_cache = wt
        return self._word_cache or None

    def note_tokens(self, input_ids):
        """Once per micro-batch, before its backward pass: the token ids it looks up in the word-embedding table -- the only rows its
        gradient can touch (model/model.py:232-236; padding_idx 0 included: harmless).  None = this micro-batch's table gradient is
        dense (the MLM task's tied decoder): no split this step."""
        wt = self._word_table()
        if wt is None:
            return
        if input_ids is None:
            self._rows_noted = None           # dense until the next step
            return
        if self._rows_noted is None:
            return
        if self._rowmask is None:
            self._rowmask = torch.zeros(wt[1], dtype=torch.uint8, device=self.store.device)
        if self._rowmask_clear is not None:        # (the last step cleared the mask on the side stream, behind its last reader)
            torch.cuda.current_stream().wait_event(self._rowmask_clear)
            self._rowmask_clear = None
        ids = input_ids.reshape(-1)
        self._rowmask.index_fill_(0, ids.clamp(0, wt[1] - 1), 1)
        self._rowmask_ready = torch.cuda.Event()
        self._rowmask_ready.record()
        self._rows_noted = True

    def early_word_update(self, stream=None):
        """Before the backward pass of the micro-batch that steps (every micro-batch since the last step called note_tokens): update the
        rows of the word-embedding table that NO token of them looks up.  Their gradient is zero whatever the backward pass computes, and
        0 x (clip coefficient) = 0: torch.optim.Adam's update of such a row (g = wd p, utils/optim_utils.py:33-40) depends on the
        step number and the learning rate alone -- so 93 % of the table (0.58 of the step's 3.18 GB) leaves the head of the NEXT
        forward pass, whose text branch used to wait for all of it, and streams beside this step's own forward / backward instead.
        `step()` then updates the looked-up rows only (with their gradients, clipped).  Parameters bit-identical to the one-launch
        update.  Returns True when the ahead-of-time launch was issued."""
        wt = self._word_table()
        if wt is None or self._rows_noted is not True or self._early is not None:
            return False
        st = self.store
        g0, g1 = self.param_groups
        lr = float(g0['lr'])
        if float(g1['lr']) != lr:
            return False
        off, V, H, name = wt
        flags = self._row_flags(off, V * H, no_decay(name))
        side = stream
        if side is None:
            enc = self.overlap_encoder
            side = getattr(enc, '_side_stream', None) if enc is not None else None
        if side is None:
            side = _lib.shared_stream(st.device, 'side')
        # behind the mask's last fill only -- NOT behind everything queued on the current stream: called in front of the forward pass, the
        # launch then runs beside it like one more of the optimizer's blocks (the side stream holds the previous step's update ahead of it)
        if self._rowmask_ready is not None:
            side.wait_event(self._rowmask_ready)
        b1, b2 = g0['betas']
        import ctypes as C
        wgs = int(os.environ.get('UNITER_ADAM_EARLY_WGS', self.overlap_workgroups))
        check(_lib.lib().uniter_adam_step_rows(st.flat_params.data_ptr() + 4 * off, st.flat_grads.data_ptr() + 4 * off,
                                               self.exp_avg.data_ptr() + 4 * off, self.exp_avg_sq.data_ptr() + 4 * off,
                                               flags.data_ptr(), V * H, None, 1.0, 0.0, lr, float(b1), float(b2), float(g0['eps']),
                                               float(g0['weight_decay']), self.step_count + 1, int(self.adamw), 0,
                                               self._rowmask.data_ptr(), H, 0, wgs, C.c_void_p(side.cuda_stream)),
              'uniter_adam_step_rows')
        ev = torch.cuda.Event()
        ev.record(side)
        self._early = (self.step_count + 1, lr, (float(b1), float(b2)), float(g0['eps']), float(g0['weight_decay']), bool(self.adamw), ev)
        return True

    def _row_flags(self, off, n, nodecay):
        """chunk flags of the word table for the row-split launches: every chunk on the update path (decay as the tensor's group says)"""
        key = ('rows', off, n, bool(nodecay))
        f = self._flags_cache.get(key)
        if f is None:
            f = torch.full((n // CHUNK,), 1 if nodecay else 2, dtype=torch.uint8, device=self.store.device)
            self._flags_cache[key] = f
        return f

    def join(self):
        """Make the current stream wait for an overlapped update still running on the side stream."""
        if self._pending is not None:
            torch.cuda.current_stream().wait_event(self._pending)
            self._pending = None

    LAZY_SUFFIXES = ('attention.self.query.weight', 'attention.self.key.weight', 'attention.self.value.weight',
                     'attention.output.dense.weight', 'intermediate.dense.weight', 'output.dense.weight')

    def _chunk_flags(self, lazy=False):
        st = self.store
        key = (frozenset(st.touched), bool(lazy))
        if key != self._flags_key:
            flags = self._flags_cache.get(key)
            if flags is None:
                host = torch.zeros(st.numel // CHUNK, dtype=torch.uint8)
                for n in st.touched:
                    o = st.offsets[n] // CHUNK
                    k = (st.params[n].numel() + CHUNK - 1) // CHUNK
                    keep = 4 if (lazy and '.encoder.layer.' in '.' + n and n.endswith(self.LAZY_SUFFIXES)) else 0
                    host[o:o + k] = (1 if no_decay(n) else 2) + keep
                flags = host.to(st.device)
                if len(self._flags_cache) < 16:
                    self._flags_cache[key] = flags
            self._flags = flags
            self._flags_key = key
        return self._flags

    def grad_norm(self):
        """L2 norm of all gradients touched since the last zero_grad (device tensor)."""
        st = self.store
        check(_lib.lib().uniter_grad_sumsq(ptr(st.flat_grads), ptr(self._chunk_flags()), st.numel,
                                           ptr(self._sumsq), ptr(self._ws), self._ws_bytes,
                                           _lib.cur_stream()), 'uniter_grad_sumsq')
        return self._sumsq.sqrt()

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, max_grad_norm=0.0, zero_grads=True, grad_ready=None, grad_bf16=None,
             grad_pieces=None):
        """grad_bf16: optional bf16 tensor holding the gradients to apply (the reduced data-parallel payload, dp.GradSync.comm)
        in place of the fp32 gradient buffer, which is still zeroed.
        grad_ready(lo, hi): optional callable that makes the CURRENT stream wait until flat_grads[lo:hi] is final
        (dp.GradSync.wait_range): each block launch then waits only for the gradient buckets that cover it.  Clipping
        needs the norm of everything: with max_grad_norm > 0 it is reduced slice by slice over `grad_pieces` (the
        collectives' slices in issue order, tiling the buffer) -- wait for one, reduce it into its slot, go on -- so
        that behind the backward only the last collective and the norm of ITS slice are ahead of the update; without
        `grad_pieces` the whole buffer is waited for and reduced in one pass."""
        if closure is not None:
            raise UniterHipError('FusedAdam does not support closures')
        st = self.store
        if not st.is_current():
            raise UniterHipError('model parameters were moved after the optimizer was built')
        g0, g1 = self.param_groups
        lr = float(g0['lr'])
        if float(g1['lr']) != lr:
            raise UniterHipError('FusedAdam needs one learning rate for both parameter groups')
        lazy = bool(zero_grads) and self.lazy_zero_encoder is not None and os.environ.get('UNITER_LAZY_ZERO') != '0'
        flags = self._chunk_flags(lazy)
        lib = _lib.lib()
        if max_grad_norm and max_grad_norm > 0:
            def sumsq(lo, hi, out_ptr):
                if grad_bf16 is not None:
                    check(lib.uniter_grad_sumsq_bf16(grad_bf16.data_ptr() + 2 * lo, flags.data_ptr() + lo // CHUNK, hi - lo,
                                                     out_ptr, ptr(self._ws), self._ws_bytes, _lib.cur_stream()),
                          'uniter_grad_sumsq_bf16')
                else:
                    check(lib.uniter_grad_sumsq(st.flat_grads.data_ptr() + 4 * lo, flags.data_ptr() + lo // CHUNK, hi - lo,
                                                out_ptr, ptr(self._ws), self._ws_bytes, _lib.cur_stream()),
                          'uniter_grad_sumsq')

            armed, self._np_blocks = self._np_blocks, 0
            pieces = sorted(grad_pieces) if (grad_ready is not None and grad_pieces) else None
            if pieces is not None and not (pieces[0][0] == 0 and pieces[-1][1] == st.numel and
                                           all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))):
                pieces = None                     # the slices do not tile the buffer: one pass over all of it
            if armed and grad_ready is None and grad_bf16 is None:
                # the backward pass left the norm as partial sums (attach_norm_hooks): join them
                check(lib.uniter_sumsq_combine(ptr(self._np_buf), armed, ptr(self._sumsq), _lib.cur_stream()),
                      'uniter_sumsq_combine')
            elif pieces is None or len(pieces) == 1:
                if grad_ready is not None:
                    grad_ready(0, st.numel)
                sumsq(0, st.numel, self._sumsq.data_ptr())
            else:
                if self._parts is None or self._parts.numel() < len(pieces):
                    self._parts = torch.zeros(max(16, len(pieces)), dtype=torch.float64, device=st.device)
                for k, (lo, hi) in enumerate(grad_pieces):        # issue order: the early collectives have landed long ago
                    grad_ready(lo, hi)
                    sumsq(lo, hi, self._parts.data_ptr() + 8 * k)
                check(lib.uniter_sumsq_combine(ptr(self._parts), len(pieces), ptr(self._sumsq), _lib.cur_stream()),
                      'uniter_sumsq_combine')
            grad_ready = None
        else:
            self._np_blocks = 0
        self.step_count += 1
        b1, b2 = g0['betas']

        mirror = getattr(st, 'mirror', None)
        enc = self.overlap_encoder
        plan = self._overlap_plan(enc) if enc is not None else None
        # the mirror feeds the encoder LAYERS' dense products only: blocks outside them (the 24 M parameters of the embeddings,
        # the heads) skip its 2 - 6 bytes per parameter
        layers_rng = (min(b[0] for b in plan[1][1:]), max(b[1] for b in plan[1][1:])) if plan is not None and len(plan[1]) > 1 else None

        def launch(lo, hi, stream_ptr, max_wgs=0):
            off = lo * 4
            mirror = getattr(st, 'mirror', None)
            if mirror is not None and layers_rng is not None and not (lo >= layers_rng[0] and hi <= layers_rng[1]):
                mirror = None
            # the bf16 weight mirror (precision 'bf16') is written by the same kernel, on the same stream
            # (precision 'fp32x3': its three bf16 pieces, piece p at mirror + p * numel)
            x3 = mirror is not None and getattr(st, 'mirror_pieces', 1) == 3
            dst = st.pair_src() if x3 else None       # fp32x3: the layers' weights go to the mirror in the paired-row layout (the launch walks the mirror's order)
            check(lib.uniter_adam_step_x3p(st.flat_params.data_ptr() + off, st.flat_grads.data_ptr() + off,
                                           (grad_bf16.data_ptr() + lo * 2) if grad_bf16 is not None else None,
                                           self.exp_avg.data_ptr() + off, self.exp_avg_sq.data_ptr() + off,
                                           flags.data_ptr() + lo // CHUNK, hi - lo, ptr(self._sumsq),
                                           float(grad_scale), float(max_grad_norm or 0.0), lr, float(b1), float(b2),
                                           float(g0['eps']), float(g0['weight_decay']), self.step_count,
                                           int(self.adamw), int(bool(zero_grads)),
                                           (mirror.data_ptr() + lo * 2) if mirror is not None else None,
                                           st.numel if x3 else 0,
                                           (dst.data_ptr() + 8 * (lo // CHUNK)) if dst is not None else None,
                                           lo if dst is not None else 0,
                                           max_wgs, stream_ptr),
                  'uniter_adam_step')

        # the word-embedding table's rows without a gradient were updated ahead (early_word_update): its launch takes the looked-up rows
        early, self._early = self._early, None
        wt = self._word_table() if early is not None else None
        if early is not None:
            ok = (wt is not None and early[:6] == (self.step_count, lr, (float(b1), float(b2)), float(g0['eps']), float(g0['weight_decay']),
                                                  bool(self.adamw)) and grad_bf16 is None)
            if not ok:
                raise UniterHipError('FusedAdam.step: the word-embedding rows without a gradient were updated ahead (early_word_update) '
                                     'for step %d / lr %g, but this step runs with other hyper-parameters or a bf16 gradient payload'
                                     % (early[0], early[1]))

        def launch_word_rows(stream_ptr, max_wgs):
            off, V, H, name = wt
            rf = self._row_flags(off, V * H, no_decay(name))
            check(lib.uniter_adam_step_rows(st.flat_params.data_ptr() + 4 * off, st.flat_grads.data_ptr() + 4 * off,
                                            self.exp_avg.data_ptr() + 4 * off, self.exp_avg_sq.data_ptr() + 4 * off, rf.data_ptr(),
                                            V * H, ptr(self._sumsq), float(grad_scale), float(max_grad_norm or 0.0), lr, float(b1),
                                            float(b2), float(g0['eps']), float(g0['weight_decay']), self.step_count, int(self.adamw),
                                            int(bool(zero_grads)), self._rowmask.data_ptr(), H, 1, max_wgs, stream_ptr),
                  'uniter_adam_step_rows')

        if plan is None:
            if grad_ready is not None:
                grad_ready(0, st.numel)
            if early is None:
                launch(0, st.numel, _lib.cur_stream())
            else:
                off, V, H, _ = wt
                we = off + V * H
                torch.cuda.current_stream().wait_event(early[6])
                if off > 0:
                    launch(0, off, _lib.cur_stream())
                launch_word_rows(_lib.cur_stream(), 0)
                if we < st.numel:
                    launch(we, st.numel, _lib.cur_stream())
        else:
            # The update is HBM-bound, the next forward MFMA-bound: run the encoder's blocks on the
            # side stream in the order the forward needs them (embeddings, layer 0, 1, ..), one event
            # per block; the next uniter_model_forward waits block by block instead of for all of it.
            head, blocks, word = plan
            if os.environ.get('UNITER_ADAM_WORD_SPLIT') == '0':
                word = None
            if early is not None and (word is None or word[0] != wt[0]):
                raise UniterHipError('FusedAdam.step: early_word_update needs the word table as its own optimizer launch '
                                     '(UNITER_ADAM_WORD_SPLIT=0 or an unexpected parameter layout)')
            main = torch.cuda.current_stream()
            for lo, hi in head:                                  # pooler / heads: tiny, stay on this stream
                if grad_ready is not None:
                    grad_ready(lo, hi)
                launch(lo, hi, _lib.cur_stream())
            side = enc._side_stream
            if side is None:
                side = enc._side_stream = _lib.shared_stream(st.device, 'side')
            side.wait_stream(main)
            events = []
            import ctypes as C

            def block(lo, hi, max_wgs):
                if grad_ready is not None:
                    with torch.cuda.stream(side):
                        grad_ready(lo, hi)
                if early is not None and word is not None and (lo, hi) == tuple(word):
                    # the table's looked-up rows only (the ahead-of-time launch ran on this stream: stream order), then the 64-element
                    # padding behind the table, if the block has any
                    we = wt[0] + wt[1] * wt[2]
                    launch_word_rows(C.c_void_p(side.cuda_stream), max_wgs)
                    # the row mask belongs to the micro-batches of THIS step: cleared right behind its last reader (NOT behind the layers'
                    # blocks that follow on this stream: the next step's note_tokens waits for this event in front of its forward pass)
                    with torch.cuda.stream(side):
                        self._rowmask.zero_()
                    self._rowmask_clear = torch.cuda.Event()
                    self._rowmask_clear.record(side)
                    if we < hi:
                        launch(we, hi, C.c_void_p(side.cuda_stream), max_wgs)
                else:
                    launch(lo, hi, C.c_void_p(side.cuda_stream), max_wgs)
                ev = torch.cuda.Event()
                ev.record(side)
                return ev

            word_ev = None
            for k, (lo, hi) in enumerate(blocks):
                # the embeddings' block has the chip to itself (the forward waits for it); the layers' blocks share it
                # with the forward of the layers before them: a grid that leaves the forward its wave slots
                if k == 0 and word is not None:
                    # everything but the word table first (2 M parameters): the next forward's image branch reads no word
                    # embedding and starts behind this launch, beside the table's 0.13 ms of streaming (grid capped at half the chip's wave
                    # slots: the text branch of the next forward waits for this launch -- 512 workgroups 4.83, 1024 4.79 ms bf16 step)
                    if os.environ.get('UNITER_ADAM_EMB_MAIN', '1') != '0' and grad_ready is None:
                        # ... on the MAIN stream: the next forward's first kernels then follow it in stream order instead of
                        # behind a cross-stream event that an idle queue picks up 40-150 us late
                        word_ev = block(word[0], word[1], int(os.environ.get('UNITER_ADAM_WORD_WGS', min(2048, self.overlap_workgroups * 4))))
                        launch(word[1], hi, _lib.cur_stream(), 0)
                        ev = torch.cuda.Event()
                        ev.record(main)
                        events.append(ev)
                    else:
                        events.append(block(word[1], hi, 0))
                        word_ev = block(word[0], word[1], int(os.environ.get('UNITER_ADAM_WORD_WGS', min(2048, self.overlap_workgroups * 4))))
                else:
                    events.append(block(lo, hi, 0 if k == 0 else self.overlap_workgroups))
            last = events[-1]                    # the side stream's last launch: what join() waits for
            if word_ev is not None:
                events.append(word_ev)
            enc._set_ready_events(events)
            self._pending = last
        if early is not None or self._rows_noted is not False:
            # the row mask belongs to the micro-batches of THIS step: clear it behind its last reader
            if self._rowmask is not None and not (plan is not None and early is not None):
                self._rowmask.zero_()          # (the overlapped path cleared it behind the looked-up rows' launch, on the side stream)
                self._rowmask_clear = None
            self._rows_noted = False
        if zero_grads:
            st.touched.clear()      # flags stay cached: the same set is touched again next step
            if lazy:
                st.wgrad_stale = True

    def zero_grad(self, set_to_none=False):
        self.store.zero_grads()
        self._flags_key = None


class TorchOptimizerStep(object):
    """`--optimizer adamax | sgd` (utils/optim_utils.py:36-43): the update itself is torch.optim's, on the views
    into the flat buffers; averaging, global-norm clipping and zero_grad around it follow FusedAdam.step's
    contract so the trainer drives both alike.  Not a fused kernel: the reference's recipe trains with Adam."""

    def __init__(self, model, inner):
        self.store = model.param_store() if hasattr(model, 'param_store') else ensure_store(model)
        self.inner = inner
        self.param_groups = inner.param_groups

    def join(self):
        pass

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, max_grad_norm=0.0, zero_grads=True, grad_ready=None):
        st = self.store
        if grad_ready is not None:
            grad_ready(0, st.numel)
        g = st.flat_grads
        if grad_scale != 1.0:
            g.mul_(grad_scale)
        if max_grad_norm and max_grad_norm > 0:
            g.mul_(torch.clamp(float(max_grad_norm) / (g.norm() + 1e-6), max=1.0))
        idle = [p for n, p in st.params.items() if n not in st.touched]     # no gradient this step: no update, no decay
        for p in idle:
            p.grad = None
        self.inner.step()
        st.reattach_grads(full=True)
        st.mirror_dirty = True
        if zero_grads:
            g.zero_()
            st.touched.clear()

    def zero_grad(self, set_to_none=False):
        self.store.zero_grads()


def get_optimizer(model, config, group_param_func=None):
    """utils/optim_utils.py:9-46: two parameter groups (weight decay off for biases and LayerNorm), then the
    optimizer by name.  adam / adamw run as the fused HIP step."""
    if group_param_func is not None:
        raise UniterHipError('custom parameter grouping is not supported (INTEGRATION.md: out of scope)')
    name = config['optimizer']
    if name in ('adam', 'adamw'):
        return FusedAdam(model, lr=config['lr'], betas=(config['beta1'], config['beta2']),
                         weight_decay=config['weight_decay'], adamw=(name == 'adamw'))
    if name not in ('adamax', 'sgd'):
        raise ValueError('invalid optimizer %r' % name)
    store = model.param_store() if hasattr(model, 'param_store') else ensure_store(model)
    named = list(store.params.items())
    groups = [{'params': [p for n, p in named if not no_decay(n)], 'weight_decay': config['weight_decay']},
              {'params': [p for n, p in named if no_decay(n)], 'weight_decay': 0.0}]
    if name == 'adamax':             # utils/optim_utils.py:36-37: torch's default betas, the config's are not passed
        inner = torch.optim.Adamax(groups, lr=config['lr'])
    else:                            # :41-43: momentum = beta1
        inner = torch.optim.SGD(groups, lr=config['lr'], momentum=config['beta1'])
    return TorchOptimizerStep(model, inner)


# --------------------------------------------------------------------------- #
# schedulers (transformers.get_*_schedule_with_warmup restated; train_template.py:72-82)
# --------------------------------------------------------------------------- #
def cosine_warmup_lambda(num_warmup_steps, num_training_steps, num_cycles=0.5):
    def f(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        prog = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * prog)))
    return f


def linear_warmup_lambda(num_warmup_steps, num_training_steps):
    def f(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) /
                   float(max(1, num_training_steps - num_warmup_steps)))
    return f


def get_scheduler(optimizer, config, steps_per_epoch):
    optimizer = getattr(optimizer, 'inner', optimizer)        # TorchOptimizerStep wraps a torch optimizer
    total = steps_per_epoch * config['max_epoch']
    kind = config['scheduler']
    if kind == 'step':
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=config['lr_decay_step'],
                                               gamma=config['lr_decay_factor'])
    if kind == 'multi_step':
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[5, 10, 15, 25, 40],
                                                    gamma=config['lr_decay_factor'])
    if kind == 'warmup':
        return torch.optim.lr_scheduler.LambdaLR(optimizer, linear_warmup_lambda(config['warmup_steps'], total))
    if kind == 'warmup_cosine':
        return torch.optim.lr_scheduler.LambdaLR(optimizer, cosine_warmup_lambda(config['warmup_steps'], total))
    raise ValueError('unknown scheduler %r' % kind)


# --------------------------------------------------------------------------- #
# the step
# --------------------------------------------------------------------------- #
def sync_step(optimizer, grad_sync, accum, max_grad_norm):
    """average_gradients + clip + optimizer step + zero_grad behind the data-parallel exchange
    (train_template.py:89-92,103-107).  With clipping the norm needs every bucket; without it each
    optimizer block waits only for the buckets that cover it."""
    world, ready, g16, kw = 1, None, None, {}
    if grad_sync is not None and grad_sync.active:
        world = grad_sync.world
        fused = isinstance(optimizer, FusedAdam)
        # bf16 payload + the fused optimizer: the reduced sums are consumed where RCCL left them
        grad_sync.consumer_reads_comm = grad_sync.comm is not None and fused
        if grad_sync.consumer_reads_comm:
            kw['grad_bf16'] = grad_sync.comm
        grad_sync.flush_all()
        if fused:
            ready = grad_sync.wait_range
            if max_grad_norm and max_grad_norm > 0:
                kw['grad_pieces'] = grad_sync.pieces()     # the clip norm, slice by slice as the collectives land
        else:
            grad_sync.finish()
    elif grad_sync is not None:
        world = grad_sync.world
    optimizer.step(grad_scale=1.0 / (accum * world), max_grad_norm=max_grad_norm, zero_grads=True, grad_ready=ready, **kw)



class TrainStep(object):
    """`TrainerTemplate.calculate_loss` (train_template.py:95-126) + the forward of
    `TrainerUniter.train_iter_step` (train_uniter.py:67-72).

    Quirk kept on purpose: the modulo test is on the per-epoch iteration index
    starting at 0, so iteration 0 steps immediately with a single micro-batch whose
    gradient is still divided by ``gradient_accumulation``.
    """

    def __init__(self, model, optimizer, scheduler, config, grad_sync=None):
        self.model, self.optimizer, self.scheduler, self.config = model, optimizer, scheduler, config
        self.grad_sync = grad_sync          # DP: object with .finish() called before the optimizer step
        if config.get('loss_func', 'bce_logits') != 'bce_logits':
            raise UniterHipError("only loss_func='bce_logits' is built on the HIP path")
        enc = getattr(model, 'uniter_model', None)
        if grad_sync is None and enc is not None and isinstance(optimizer, FusedAdam) and (config.get('max_grad_norm') or 0) > 0:
            optimizer.attach_norm_hooks(enc)        # the clip norm is reduced bucket by bucket during the backward pass
        if enc is not None and isinstance(optimizer, FusedAdam):
            optimizer.lazy_zero_encoder = enc       # zero_grad skips what the next backward pass overwrites anyway
        self.iters = 0
        self.last_loss = None
        self.last_probs = None

    @staticmethod
    def forward_kwargs(batch):
        return dict(img_feat=batch['img_feat'], img_pos_feat=batch['img_pos_feat'],
                    input_ids=batch['input_ids'], position_ids=batch['position_ids'],
                    attention_mask=batch['attn_mask'], gather_index=batch['gather_index'],
                    output_all_encoded_layers=False, seq_lens=batch.get('seq_lens'))

    def train_iter(self, batch, iters=None):
        if iters is not None:
            self.iters = iters
        cfg = self.config
        accum = cfg['gradient_accumulation']
        stepping = self.iters % accum == 0
        opt = self.optimizer
        if self.grad_sync is None and isinstance(opt, FusedAdam):
            # single process: the rows of the word-embedding table this step cannot touch are updated AHEAD, beside this forward pass
            # (FusedAdam.early_word_update); with a gradient exchange the other ranks' tokens are not known here
            opt.note_tokens(batch.get('input_ids'))
            if stepping:
                opt.early_word_update()
        preds = self.model(**self.forward_kwargs(batch))
        loss, probs = bce_with_logits_loss(preds.squeeze(1), batch['labels'], cfg['pos_wt'],
                                           return_probs=True)
        if self.grad_sync is not None:
            self.grad_sync.prepare(will_step=stepping, token_ids=batch['input_ids'])
        loss.backward(unit_gradient(loss.device))
        if stepping:
            sync_step(self.optimizer, self.grad_sync, accum, cfg['max_grad_norm'])
            self.scheduler.step()
        self.last_loss, self.last_probs = loss.detach(), probs
        self.iters += 1
        return self.last_loss

    @torch.no_grad()
    def eval_iter(self, batch):
        preds = self.model(**self.forward_kwargs(batch))
        loss, probs = bce_with_logits_loss(preds.squeeze(1), batch['labels'], self.config['pos_wt'],
                                           return_probs=True)
        return loss, probs

    @torch.no_grad()
    def test_iter(self, batch):
        return self.model(**self.forward_kwargs(batch)).squeeze()
