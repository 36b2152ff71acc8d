"""Host-side mirror of the reference's UNITER module surface, executing on
libuniter_hip.so.

Mirrors (same names, constructor / forward signatures, state_dict keys, error
behaviour):
  UniterConfig               model/model.py:24-114
  UniterPreTrainedModel      model/model.py:117-214  (init_weights, from_pretrained)
  UniterTextEmbeddings       model/model.py:217-245
  UniterImageEmbeddings      model/model.py:248-272
  UniterEncoder / BertLayer… model/model.py:275-292, model/layer.py:53-170
  BertPooler                 model/layer.py:173-185
  UniterModel                model/model.py:295-367

The nn.Module tree exists to own parameters under the reference's names; the
arithmetic of ``UniterModel.forward`` and of its backward is one kernel schedule
inside the library (csrc/model.cpp).  All parameters (and gradients) live in one
flat fp32 device buffer each (``ParamStore``) so that the optimizer step and the
data-parallel gradient exchange are single-buffer operations; the
``nn.Parameter`` objects are views into it.

There is no CPU / eager fallback: tensors must be on the GPU and the library
must be built, otherwise a ``UniterHipError`` is raised.
"""
import copy
import ctypes as C
import json
import os
import logging

import numpy as np

import torch
from torch import nn

from . import _lib
from ._lib import UniterBatchC, UniterConfigC, UniterHipError, check, ptr

logger = logging.getLogger(__name__)

CHUNK = 64          # floats; granularity of the optimizer's per-chunk flags


class UniterConfig(object):
    """Hyper-parameters of a `UniterModel`.  Same construction surface as the reference's holder
    (model/model.py:24-114: an int vocabulary size plus keyword overrides, or the path of a json
    file; `from_dict`, `from_json_file`, `to_dict`, `to_json_string`) -- the attribute names are the
    json schema of config/uniter-*.json."""

    FIELDS = (('hidden_size', 768), ('num_hidden_layers', 12), ('num_attention_heads', 12),
              ('intermediate_size', 3072), ('hidden_act', 'gelu'), ('hidden_dropout_prob', 0.1),
              ('attention_probs_dropout_prob', 0.1), ('max_position_embeddings', 512),
              ('type_vocab_size', 2), ('initializer_range', 0.02))

    def __init__(self, vocab_size_or_config_json_file, **overrides):
        src = vocab_size_or_config_json_file
        if isinstance(src, bool) or not isinstance(src, (int, str)):
            raise ValueError('UniterConfig needs a vocabulary size (int) or the path of a config json (str), '
                             'got %s' % type(src).__name__)
        unknown = set(overrides) - {k for k, _ in self.FIELDS}
        if unknown:
            raise TypeError('unknown UniterConfig fields: %s' % ', '.join(sorted(unknown)))
        if isinstance(src, str):
            with open(src, encoding='utf-8') as f:
                self._absorb(json.load(f))
        else:
            self.vocab_size = src
            for name, default in self.FIELDS:
                setattr(self, name, overrides.get(name, default))

    def _absorb(self, mapping):
        for key, value in mapping.items():
            setattr(self, key, value)
        return self

    @classmethod
    def from_dict(cls, json_object):
        return cls(-1)._absorb(json_object)

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, encoding='utf-8') as f:
            return cls.from_dict(json.load(f))

    def to_dict(self):
        return copy.deepcopy(vars(self))

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def __repr__(self):
        return self.to_json_string()


# hyper-parameters of the two published UNITER sizes (the reference ships them as
# config/uniter-base.json and config/uniter-large.json); `UniterConfig.from_name('uniter-base')`
BUILTIN_CONFIGS = {
    'uniter-base': dict(vocab_size=28996, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                        intermediate_size=3072, hidden_act='gelu', hidden_dropout_prob=0.1,
                        attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                        initializer_range=0.02),
}
BUILTIN_CONFIGS['uniter-large'] = dict(BUILTIN_CONFIGS['uniter-base'], hidden_size=1024, num_hidden_layers=24,
                                       num_attention_heads=16, intermediate_size=4096)


def resolve_config(path_or_name):
    """A json path, or the name of a built-in size ('uniter-base' / 'uniter-large',
    with or without a directory prefix and '.json')."""
    import os
    if os.path.isfile(path_or_name):
        return UniterConfig.from_json_file(path_or_name)
    key = os.path.basename(path_or_name)
    key = key[:-5] if key.endswith('.json') else key
    if key in BUILTIN_CONFIGS:
        return UniterConfig.from_dict(BUILTIN_CONFIGS[key])
    raise ValueError("[!] ERROR: config JSON path does not exist: %r" % path_or_name)


# --------------------------------------------------------------------------- #
# flat parameter / gradient storage
# --------------------------------------------------------------------------- #
def _round_up(x, a):
    return (x + a - 1) // a * a


class ParamStore(object):
    """One flat fp32 buffer for parameters and one for gradients.

    Layout (backward-completion order, so that a gradient bucket is a contiguous
    slice): [head & pooler | layer nl-1 | ... | layer 0 | embeddings].  Inside a
    layer query/key/value weights are adjacent (one [3H,H] GEMM operand), as are
    their biases.  Every tensor starts on a 64-float boundary (optimizer flag
    granularity).
    """

    def __init__(self, module):
        named = list(module.named_parameters())
        if not named:
            raise UniterHipError('module has no parameters')
        dev = named[0][1].device
        if dev.type != 'cuda':
            raise UniterHipError('parameters must be on the GPU before the first forward '
                                 '(model.cuda()); got %s' % dev)
        for n, p in named:
            if p.dtype != torch.float32 or p.device != dev:
                raise UniterHipError('parameter %s must be float32 on %s' % (n, dev))
        self.device = dev
        order, self.buckets = self._order(named)
        self.names = [n for n, _ in order]
        offs, off = {}, 0
        for n, p in order:
            offs[n] = off
            off += _round_up(p.numel(), CHUNK)
        self.numel = off
        self.offsets = offs
        self.flat_params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.params = {}
        with torch.no_grad():
            for n, p in order:
                o, k = offs[n], p.numel()
                view = self.flat_params[o:o + k].view(p.shape)
                view.copy_(p.data)
                old_grad = p.grad
                p.data = view
                gview = self.flat_grads[o:o + k].view(p.shape)
                if old_grad is not None:
                    gview.copy_(old_grad)
                p.grad = gview
                self.params[n] = p
        # bucket boundaries as (start, end) element offsets
        self.bucket_ranges = []
        for names in self.buckets:
            s = min(offs[n] for n in names)
            e = max(offs[n] + _round_up(self.params[n].numel(), CHUNK) for n in names)
            self.bucket_ranges.append((s, e))
        self.touched = set()
        # True between an optimizer step that did not clear the encoder layers' weight gradients (FusedAdam.lazy_zero) and the
        # backward pass that overwrites them: p.grad of those tensors holds LAST step's values meanwhile
        self.wgrad_stale = False

    @staticmethod
    def _order(named):
        """Group parameter names into backward-order buckets."""
        import re
        by = dict(named)
        head, layers, emb = [], {}, []
        for n, _ in named:
            m = re.search(r'encoder\.layer\.(\d+)\.', n)
            if m:
                layers.setdefault(int(m.group(1)), []).append(n)
            elif 'embeddings.' in n:
                emb.append(n)
            else:
                head.append(n)

        def layer_key(n):
            ranks = ['attention.self.query.weight', 'attention.self.key.weight',
                     'attention.self.value.weight', 'attention.self.query.bias',
                     'attention.self.key.bias', 'attention.self.value.bias']
            for i, r in enumerate(ranks):
                if n.endswith(r):
                    return (0, i)
            return (1, 0)
        buckets, order = [], []
        if head:
            buckets.append(head)
        for l in sorted(layers, reverse=True):
            names = layers[l]
            fused = sorted([n for n in names if layer_key(n)[0] == 0], key=layer_key)
            rest = [n for n in names if layer_key(n)[0] == 1]
            buckets.append(fused + rest)
        if emb:
            buckets.append(emb)
        for b in buckets:
            order += [(n, by[n]) for n in b]
        return order, buckets

    def is_current(self, full=False):
        """Are the module's parameters still views of the flat buffer?  The per-step check
        looks at three sentinels (.to()/.cuda() re-home every parameter at once); `full`
        checks all of them."""
        names = self.names if full else (self.names[0], self.names[len(self.names) // 2], self.names[-1])
        base = self.flat_params.data_ptr()
        for n in names:
            if self.params[n].data_ptr() != base + 4 * self.offsets[n]:
                return False
        return True

    def reattach_grads(self, full=False):
        names = self.names if full else (self.names[0], self.names[len(self.names) // 2], self.names[-1])
        base = self.flat_grads.data_ptr()
        ok = True
        for n in names:
            g = self.params[n].grad
            if g is None or g.data_ptr() != base + 4 * self.offsets[n]:
                ok = False
                break
        if ok and not full:
            return
        for n, p in self.params.items():
            o, k = self.offsets[n], p.numel()
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * o:
                p.grad = self.flat_grads[o:o + k].view(p.shape)

    def touch(self, names):
        self.touched.update(names)

    # -- bf16 mirror of the parameters (precision 'bf16': GEMM weight operands are read from it) ----
    # precision 'fp32x3': the mirror holds the THREE bf16 pieces of every parameter, piece p at mirror[p * numel:]
    def ensure_mirror(self, pieces=1):
        """Allocate / refresh the bf16 copy (or the three bf16 pieces) of the flat parameter buffer.  Stale when a torch
        op wrote a parameter in place (load_state_dict, p.add_(..): every Parameter's version counter is summed,
        ~20 us for 212 tensors) or a raw-pointer kernel did and said so (mirror_dirty; trainer.FusedAdam
        refreshes the mirror itself, block by block)."""
        if (getattr(self, 'mirror', None) is None or self.mirror.device != self.flat_params.device
                or getattr(self, 'mirror_pieces', 1) != pieces):
            self.mirror = torch.empty(pieces * self.numel, dtype=torch.bfloat16, device=self.flat_params.device)
            self.mirror_pieces = pieces
            self.mirror_dirty = True
            self._mirror_version = -1
        version = self.flat_params._version
        for p in self.params.values():
            version += p._version
        if self.mirror_dirty or self._mirror_version != version:
            self.refresh_mirror(0, self.numel, _lib.cur_stream())
            self.mirror_dirty = False
            self._mirror_version = version
        return self.mirror

    PAIRED_SUFFIXES = ('attention.self.query.weight', 'attention.self.key.weight', 'attention.self.value.weight',
                       'attention.output.dense.weight', 'intermediate.dense.weight', 'output.dense.weight')

    def pair_dst(self):
        """Destination table of the PAIRED-ROW layout of the x3 weight mirror (include/uniter_hip.h, uniter_adam_step_x3p): one int32 per
        64-element chunk of the flat buffer -- for the encoder layers' dense weights [N][K] the absolute mirror offset of the chunk's
        first 32-element unit, (n >> 1) * 2 K + (k >> 5) * 64 + (n & 1) * 32 from the tensor's start (rows 2 q, 2 q + 1 interleaved in
        64-byte units: a 32-deep k-tile of a row pair is one 128-byte line for the forward products' loaders), -1 elsewhere.  None:
        pairing off (UNITER_X3_PAIRED=0) or a tensor shape it does not fit (odd rows, a row length that is no multiple of 64)."""
        if not hasattr(self, '_pair_dst'):
            self._pair_dst = None
            if os.environ.get('UNITER_X3_PAIRED', '1') != '0':
                tab = np.full(self.numel // CHUNK, -1, dtype=np.int32)
                ok, any_ = True, False
                for n in self.names:
                    if '.encoder.layer.' not in '.' + n or not n.endswith(self.PAIRED_SUFFIXES):
                        continue
                    N, K = (int(x) for x in self.params[n].shape)
                    off = self.offsets[n]
                    if N % 2 or K % CHUNK or off % CHUNK:
                        ok = False
                        break
                    cr = np.arange(N * K // CHUNK, dtype=np.int64)
                    row, kk = cr // (K // CHUNK), (cr % (K // CHUNK)) * CHUNK
                    tab[off // CHUNK: off // CHUNK + cr.size] = (off + (row >> 1) * 2 * K + (kk >> 5) * 64 + (row & 1) * 32).astype(np.int32)
                    any_ = True
                if ok and any_ and self.numel < 2 ** 31:
                    self._pair_dst = torch.from_numpy(tab).to(self.device)
        return self._pair_dst

    def pair_src(self):
        """The inverse of pair_dst for the optimizer (uniter_adam_step_x3p walks the buffer in the mirror's order): int32 [numel / 64][2],
        the flat-buffer offsets of the parameters behind the two 32-element units of every 64-element chunk of the mirror -- unit u of
        row 2 q and unit u of row 2 q + 1 of a paired tensor; (-1, -1) where a chunk is its own source."""
        if not hasattr(self, '_pair_src'):
            self._pair_src = None
            if self.pair_dst() is not None:
                tab = np.full((self.numel // CHUNK, 2), -1, dtype=np.int32)
                for n in self.names:
                    if '.encoder.layer.' not in '.' + n or not n.endswith(self.PAIRED_SUFFIXES):
                        continue
                    N, K = (int(x) for x in self.params[n].shape)
                    off = self.offsets[n]
                    d = np.arange(N * K // CHUNK, dtype=np.int64) * CHUNK          # mirror offset of the chunk inside the tensor
                    q, u = d // (2 * K), (d % (2 * K)) // 64
                    s0 = off + (2 * q) * K + 32 * u
                    tab[off // CHUNK: off // CHUNK + d.size, 0] = s0.astype(np.int32)
                    tab[off // CHUNK: off // CHUNK + d.size, 1] = (s0 + K).astype(np.int32)
                self._pair_src = torch.from_numpy(tab).to(self.device)
        return self._pair_src

    def mirror_paired(self):
        return getattr(self, 'mirror_pieces', 1) == 3 and self.pair_dst() is not None

    def refresh_mirror(self, lo, hi, stream_ptr):
        if getattr(self, 'mirror_pieces', 1) == 3 and self.pair_dst() is not None:
            check(_lib.lib().uniter_mirror_refresh_x3(self.flat_params.data_ptr(), lo, hi - lo, self.mirror.data_ptr(), self.numel,
                                                      self.pair_dst().data_ptr(), stream_ptr), 'uniter_mirror_refresh_x3')
        elif getattr(self, 'mirror_pieces', 1) == 3:
            check(_lib.lib().uniter_split3(self.flat_params.data_ptr() + 4 * lo, 1, hi - lo, hi - lo,
                                           self.mirror.data_ptr() + 2 * lo, 0, self.numel, stream_ptr), 'uniter_split3')
        else:
            check(_lib.lib().uniter_cast_bf16(self.flat_params.data_ptr() + 4 * lo, self.mirror.data_ptr() + 2 * lo,
                                              hi - lo, stream_ptr), 'uniter_cast_bf16')

    def zero_grads(self):
        self.flat_grads.zero_()
        self.wgrad_stale = False
        self.touched.clear()
        self.reattach_grads(full=True)


def _find_store(module):
    return getattr(module, '_param_store', None)


def ensure_store(root):
    """(Re)build the flat store of `root` if parameters were moved / replaced."""
    st = _find_store(root)
    if st is None or not st.is_current():
        st = ParamStore(root)
        object.__setattr__(root, '_param_store', st)
        for m in root.modules():
            if m is not root and hasattr(m, '_param_store'):
                object.__setattr__(m, '_param_store', None)
            if isinstance(m, UniterModel):
                m._destroy_handle()
                object.__setattr__(m, '_store_root', root)
    else:
        st.reattach_grads()
    return st


# --------------------------------------------------------------------------- #
# parameter containers (reference module names => identical state_dict keys)
# --------------------------------------------------------------------------- #
class _ParamLayerNorm(nn.Module):
    """Parameter holder for a LayerNorm(eps=1e-12) (Apex FusedLayerNorm in the
    reference: model/model.py:229,252,253,258; model/layer.py:108,149)."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.eps = eps


class _ParamLinear(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features))


class _ParamEmbedding(nn.Module):
    def __init__(self, num_embeddings, embedding_dim, padding_idx=None):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.padding_idx = padding_idx
        self.weight = nn.Parameter(torch.empty(num_embeddings, embedding_dim))


class UniterTextEmbeddings(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.word_embeddings = _ParamEmbedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = _ParamEmbedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = _ParamEmbedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = _ParamLayerNorm(config.hidden_size)


class UniterImageEmbeddings(nn.Module):
    def __init__(self, config, img_dim):
        super().__init__()
        self.img_linear = _ParamLinear(img_dim, config.hidden_size)
        self.img_layer_norm = _ParamLayerNorm(config.hidden_size)
        self.pos_layer_norm = _ParamLayerNorm(config.hidden_size)
        self.pos_linear = _ParamLinear(7, config.hidden_size)
        self.mask_embedding = _ParamEmbedding(2, img_dim, padding_idx=0)
        self.LayerNorm = _ParamLayerNorm(config.hidden_size)


class BertSelfAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError(
                "The hidden size (%d) is not a multiple of the number of attention "
                "heads (%d)" % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = _ParamLinear(config.hidden_size, self.all_head_size)
        self.key = _ParamLinear(config.hidden_size, self.all_head_size)
        self.value = _ParamLinear(config.hidden_size, self.all_head_size)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = _ParamLinear(config.hidden_size, config.hidden_size)
        self.LayerNorm = _ParamLayerNorm(config.hidden_size)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        if config.hidden_act != 'gelu':
            raise ValueError('only hidden_act="gelu" (erf form, model/layer.py:31-37) is built')
        self.dense = _ParamLinear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = _ParamLinear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = _ParamLayerNorm(config.hidden_size)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)


class UniterEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])


# --------------------------------------------------------------------------- #
# small autograd bridges (pooler, head)
# --------------------------------------------------------------------------- #
class _PoolerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, weight, bias, module):
        _lib.require_gpu_tensor(hidden, torch.float32, 'hidden')
        hidden = hidden.contiguous()
        B, L, H = hidden.shape
        pooled = torch.empty(B, H, dtype=torch.float32, device=hidden.device)
        check(_lib.lib().uniter_pooler_fwd(ptr(hidden), ptr(weight), ptr(bias), ptr(pooled),
                                           B, L, H, _lib.cur_stream()), 'uniter_pooler_fwd')
        ctx.save_for_backward(hidden, pooled)
        ctx.module = module
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        hidden, pooled = ctx.saved_tensors
        mod = ctx.module
        B, L, H = hidden.shape
        dpooled = dpooled.contiguous()
        w, b = mod.dense.weight, mod.dense.bias
        _ensure_grad(w)
        _ensure_grad(b)
        dhidden = None
        if ctx.needs_input_grad[0]:
            dhidden = torch.zeros_like(hidden)
        check(_lib.lib().uniter_pooler_bwd(ptr(dpooled), ptr(pooled), ptr(hidden), ptr(w),
                                           ptr(w.grad), ptr(b.grad), ptr(dhidden), B, L, H, 0,
                                           _lib.cur_stream()), 'uniter_pooler_bwd')
        _mark_touched(mod, (w, b))
        return dhidden, None, None, None


class _LinearSmallFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, module):
        _lib.require_gpu_tensor(x, torch.float32, 'input')
        x = x.contiguous()
        B, H = x.shape
        Cn = weight.shape[0]
        y = torch.empty(B, Cn, dtype=torch.float32, device=x.device)
        check(_lib.lib().uniter_linear_small_fwd(ptr(x), ptr(weight), ptr(bias), ptr(y), B, H, Cn,
                                                 _lib.cur_stream()), 'uniter_linear_small_fwd')
        ctx.save_for_backward(x)
        ctx.module = module
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        mod = ctx.module
        dy = dy.contiguous()
        B, H = x.shape
        w, b = mod.weight, mod.bias
        Cn = w.shape[0]
        _ensure_grad(w)
        _ensure_grad(b)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        check(_lib.lib().uniter_linear_small_bwd(ptr(dy), ptr(x), ptr(w), ptr(dx), ptr(w.grad),
                                                 ptr(b.grad), B, H, Cn, _lib.cur_stream()),
              'uniter_linear_small_bwd')
        _mark_touched(mod, (w, b))
        return dx, None, None, None


class _PoolHeadFn(torch.autograd.Function):
    """Pooler + classifier (model/layer.py:179-185, model/meme_uniter.py:19-21) as ONE launch each way (csrc/head.hip,
    uniter_pool_head_fwd / _bwd): the stretch between the encoder's forward and backward passes is a chain of launch latencies with
    nothing beside it.  Same results as BertPooler followed by HipLinear."""


    @staticmethod
    def forward(ctx, hidden, anchor, pooler, linear):
        _lib.require_gpu_tensor(hidden, torch.float32, 'hidden')
        hidden = hidden.contiguous()
        B, L, H = hidden.shape
        wl = linear.weight
        Cn = wl.shape[0]
        dev = hidden.device
        # (one set of counters per head module, not per device: two models' forward launches on two streams must not count on the same words)
        ticket = linear.__dict__.get('_pool_head_ticket')
        if ticket is None or ticket.device != dev:
            ticket = torch.zeros(17 * 64, dtype=torch.int32, device=dev)      # UNITER_POOL_HEAD_TICKET_WORDS
            object.__setattr__(linear, '_pool_head_ticket', ticket)
        pooled = torch.empty(B, H, dtype=torch.float32, device=dev)
        logits = torch.empty(B, Cn, dtype=torch.float32, device=dev)
        check(_lib.lib().uniter_pool_head_fwd(ptr(hidden), ptr(pooler.dense.weight), ptr(pooler.dense.bias), ptr(wl), ptr(linear.bias),
                                              ptr(pooled), ptr(logits), ptr(ticket), B, L, H, Cn, _lib.cur_stream()),
              'uniter_pool_head_fwd')
        ctx.save_for_backward(hidden, pooled)
        ctx.pooler, ctx.linear = pooler, linear
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        hidden, pooled = ctx.saved_tensors
        pooler, linear = ctx.pooler, ctx.linear
        B, L, H = hidden.shape
        dlogits = dlogits.contiguous()
        wp, bp, wl, bl = pooler.dense.weight, pooler.dense.bias, linear.weight, linear.bias
        for p_ in (wp, bp, wl, bl):
            _ensure_grad(p_)
        dhidden = None
        if ctx.needs_input_grad[0]:
            # only row 0 of every sample carries a gradient and the launch ASSIGNS it: the zeros around it are written once, not per step
            # (the encoder's backward pass reads this tensor, it never writes it; the cache's own reference keeps autograd from
            # accumulating into it in place)
            key = (hidden.device, tuple(hidden.shape))
            cache = pooler.__dict__.get('_zero_dhidden')         # (per module: two models on two streams do not share the buffer)
            if cache is None:
                cache = {}
                object.__setattr__(pooler, '_zero_dhidden', cache)
            dhidden = cache.get(key) if os.environ.get('UNITER_HEAD_ZERO_CACHE', '1') != '0' else torch.zeros_like(hidden)
            if dhidden is None:
                if len(cache) >= 2:
                    cache.clear()
                dhidden = cache[key] = torch.zeros_like(hidden)
        check(_lib.lib().uniter_pool_head_bwd(ptr(dlogits), ptr(pooled), ptr(hidden), ptr(wp), ptr(wl), ptr(wp.grad), ptr(bp.grad),
                                              ptr(wl.grad), ptr(bl.grad), ptr(dhidden), B, L, H, wl.shape[0], 0, _lib.cur_stream()),
              'uniter_pool_head_bwd')
        _mark_touched(pooler, (wp, bp))
        _mark_touched(linear, (wl, bl))
        return dhidden, None, None, None


def pool_head(hidden, pooler, linear):
    """logits = linear(pooler(hidden)) in one launch; the separate modules when the fused form does not apply (UNITER_FUSED_HEAD=0,
    another dtype / device, more classes than a wave handles comfortably)."""
    if (os.environ.get('UNITER_FUSED_HEAD', '1') == '0' or not hidden.is_cuda or hidden.dtype != torch.float32 or hidden.dim() != 3
            or linear.weight.shape[0] > 16 or hidden.shape[-1] > 4096 or not isinstance(pooler, BertPooler) or not isinstance(linear, HipLinear)):
        return linear(pooler(hidden))
    w = pooler.dense.weight
    anchor = w if torch.is_grad_enabled() else w.detach()
    return _PoolHeadFn.apply(hidden, anchor, pooler, linear)


def _ensure_grad(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)


def _mark_touched(module, params):
    root = getattr(module, '_store_root_ref', None)
    st = _find_store(root) if root is not None else None
    if st is not None:
        key = tuple(id(p) for p in params)
        cache = module.__dict__.get('_touch_cache')
        if cache is None or cache[0] is not st:
            cache = (st, {})
            object.__setattr__(module, '_touch_cache', cache)
        names = cache[1].get(key)
        if names is None:
            ids = set(key)
            names = cache[1][key] = [n for n, p in st.params.items() if id(p) in ids]
        st.touch(names)


class BertPooler(nn.Module):
    """model/layer.py:173-185: tanh(dense(hidden[:, 0]))."""

    def __init__(self, config):
        super().__init__()
        self.dense = _ParamLinear(config.hidden_size, config.hidden_size)

    def forward(self, hidden_states):
        anchor = self.dense.weight if torch.is_grad_enabled() else self.dense.weight.detach()
        return _PoolerFn.apply(hidden_states, anchor, self.dense.bias, self)


class HipLinear(nn.Module):
    """nn.Linear(hidden, n_classes) for tiny n_classes (model/meme_uniter.py:15,20)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        # nn.Linear default init (kaiming_uniform(a=sqrt(5)) / uniform bias)
        import math
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return _LinearSmallFn.apply(x, self.weight, self.bias, self)


# --------------------------------------------------------------------------- #
# the encoder as one autograd node
# --------------------------------------------------------------------------- #
class _UniterFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, batch, keep, all_layers, mode, seed, offset):
        lib = _lib.lib()
        cfg = model.config
        B, L = batch.B, batch.L
        nl, H = cfg.num_hidden_layers, cfg.hidden_size
        dev = anchor.device
        shape = (nl, B, L, H) if all_layers else (B, L, H)
        hidden = torch.empty(shape, dtype=torch.float32, device=dev)
        if model.precision == 'fp32x3' and L > lib.uniter_attn_x3_max_len() and not getattr(model, '_warned_long_attn', False):
            # (thin spot ii: say it once instead of silently running at the older kernels' speed)
            logger.warning('precision fp32x3: joint length %d > %d -- the attention\'s own products run on the fp32 MFMA kernels '
                           '(attention_f32.hip) for this batch; the dense products stay x3 products', L, lib.uniter_attn_x3_max_len())
            model._warned_long_attn = True
        # (before the workspace is sized: the backward pass's k-pieces are planned for the CUs a gradient exchange leaves it)
        check(lib.uniter_model_set_cu_reserve(model._handle, int(getattr(model, 'cu_reserve', 0))), 'uniter_model_set_cu_reserve')
        nbytes = lib.uniter_model_ws_bytes(model._handle, B, batch.T if batch.input_ids else 0,
                                           batch.R if batch.img_feat else 0, L, mode)
        ws = model._get_ws(nbytes, mode)
        if mode != 0 and model.use_side_stream and os.environ.get('UNITER_AUX_STREAM') != '0':
            # the dropout keep flags of the attention are drawn beside the head of the forward pass, on a third stream
            aux = _lib.shared_stream(dev, 'aux')
            check(lib.uniter_model_set_aux_stream(model._handle, C.c_void_p(aux.cuda_stream)), 'uniter_model_set_aux_stream')
        else:
            check(lib.uniter_model_set_aux_stream(model._handle, None), 'uniter_model_set_aux_stream')
        check(lib.uniter_model_forward(model._handle, C.byref(batch), ptr(hidden), int(all_layers),
                                       mode, seed, offset, ptr(ws), nbytes, _lib.cur_stream()),
              'uniter_model_forward')
        ctx.model, ctx.batch, ctx.keep, ctx.ws, ctx.nbytes = model, batch, keep, ws, nbytes
        ctx.generation = lib.uniter_model_generation(model._handle)
        ctx.all_layers, ctx.seed, ctx.offset = all_layers, seed, offset
        return hidden

    @staticmethod
    def backward(ctx, d_hidden):
        model = ctx.model
        if _lib.lib().uniter_model_generation(model._handle) != ctx.generation:
            raise UniterHipError('backward of a forward that is no longer the latest one on this UniterModel: the library '
                                 'keeps the activations of ONE forward per model: run backward before ANY further forward of '
                                 'this model (forwards under torch.no_grad() reuse the same plan and count as well)')
        d_hidden = d_hidden.contiguous()
        model._run_backward(ctx.batch, d_hidden, ctx.all_layers, ctx.seed, ctx.offset, ctx.ws,
                            ctx.nbytes)
        ctx.ws = None
        return (None,) * 8


class UniterPreTrainedModel(nn.Module):
    """Weights initialisation + pretrained loading (mirror of model/model.py:117-214)."""

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        if not isinstance(config, UniterConfig):
            raise ValueError(
                "Parameter config in `{}(config)` should be an instance of "
                "class `UniterConfig`. To create a model from a Google "
                "pretrained model use "
                "`model = {}.from_pretrained(PRETRAINED_MODEL_NAME)`".format(
                    self.__class__.__name__, self.__class__.__name__))
        self.config = config

    def init_weights(self, module):
        """Linear/Embedding weights ~ N(0, initializer_range); biases 0; LN 1/0
        (model/model.py:133-146)."""
        if isinstance(module, (_ParamLinear, _ParamEmbedding, HipLinear)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, _ParamLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, (_ParamLinear, HipLinear)) and module.bias is not None:
            module.bias.data.zero_()

    @classmethod
    def from_pretrained(cls, config_file, state_dict, *inputs, **kwargs):
        """Build the model from a config (json path, built-in name or UniterConfig) and fill it from
        `state_dict` -- the loading contract of model/model.py:148-214: TF-era LayerNorm names
        (`gamma` / `beta`) are accepted, a checkpoint saved under a `bert.` prefix is accepted by a
        model that has no `bert` attribute, missing and unused keys are logged, tensors of the wrong
        shape raise RuntimeError.  The caller's dict is not modified."""
        config = config_file if isinstance(config_file, UniterConfig) else resolve_config(config_file)
        logger.info("Model config {}".format(config))
        model = cls(config, *inputs, **kwargs)
        strip = 'bert.' if (not hasattr(model, 'bert') and any(k.startswith('bert.') for k in state_dict)) else ''
        cleaned, ignored = {}, []
        for key, tensor in state_dict.items():
            if strip and not key.startswith(strip):
                ignored.append(key)              # outside the prefixed sub-tree: the checkpoint's other heads
                continue
            name = key[len(strip):]
            head, _, leaf = name.rpartition('.')
            leaf = {'gamma': 'weight', 'beta': 'bias'}.get(leaf, leaf)
            cleaned[(head + '.' if head else '') + leaf] = tensor
        result = model.load_state_dict(cleaned, strict=False)      # raises RuntimeError on a shape mismatch
        if result.missing_keys:
            logger.info("Weights of {} not initialized from pretrained model: {}".format(
                cls.__name__, list(result.missing_keys)))
        unused = ignored + [strip + k for k in result.unexpected_keys]
        if unused:
            logger.info("Weights from pretrained model not used in {}: {}".format(cls.__name__, unused))
        return model


class UniterModel(UniterPreTrainedModel):
    """Joint vision-language encoder (mirror of model/model.py:295-367)."""

    def __init__(self, config, img_dim):
        super().__init__(config)
        self.img_dim = img_dim
        self.embeddings = UniterTextEmbeddings(config)
        self.img_embeddings = UniterImageEmbeddings(config, img_dim)
        self.encoder = UniterEncoder(config)
        self.pooler = BertPooler(config)
        self.apply(self.init_weights)
        self._handle = None
        self._prefix_names = None
        self._applied_precision = None
        object.__setattr__(self, '_store_root', None)
        self._ws_cache = {}
        self._seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self._offset = 0
        self._grad_hook = None       # callable(kind, index) used by the DP gradient exchange
        # (device double buffer, doubles per layer): where the fp32x3 backward leaves each layer's clip-norm partial sums
        # (uniter_model_set_norm_partials; trainer.FusedAdam.attach_norm_hooks); None = nobody asked
        self._norm_parts = None
        # CUs the persistent matrix kernels leave to the kernels of a data-parallel gradient exchange (dp.attach sets it)
        self.cu_reserve = 0
        self._side_stream = None
        self.use_side_stream = True
        # 'fp32'; 'bf16': bf16 MFMA GEMMs on bf16-resident operands (weight mirror + bf16 activation copies),
        # fp32 master weights / LayerNorm / softmax / optimizer; 'bf16_hybrid': bf16 MFMA, operands converted in flight
        self.precision = 'fp32'
        self.pack_padded = False     # True: compute the valid positions only (see _pack); padded outputs are 0

    # -- plumbing ------------------------------------------------------------
    def set_dropout_seed(self, seed, offset=0):
        self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self._offset = int(offset)

    def _c_config(self):
        c = self.config
        return UniterConfigC(c.hidden_size, c.num_hidden_layers, c.num_attention_heads,
                             c.intermediate_size, c.vocab_size, c.max_position_embeddings,
                             c.type_vocab_size, self.img_dim, float(c.hidden_dropout_prob),
                             float(c.attention_probs_dropout_prob))

    def _root(self):
        return self._store_root if self._store_root is not None else self

    def _ensure_handle(self):
        root = self._root()
        st = _find_store(root)
        if st is None or not st.is_current() or self._handle is None:
            # the outermost module that owns this model decides the flat layout
            st = ensure_store(root)
            for m in root.modules():
                object.__setattr__(m, '_store_root_ref', root)
            lib = _lib.lib()
            cc = self._c_config()
            n = lib.uniter_num_params(C.byref(cc))
            mine = dict(self.named_parameters())
            pa, ga = (C.c_void_p * n)(), (C.c_void_p * n)()
            for i in range(n):
                name = lib.uniter_param_name(C.byref(cc), i).decode()
                p = mine[name]
                pa[i] = p.data_ptr()
                ga[i] = p.grad.data_ptr()
            h = C.c_void_p()
            check(lib.uniter_model_create(C.byref(cc), pa, ga, n, C.byref(h)), 'uniter_model_create')
            self._destroy_handle()
            self._handle = h
            self._prefix_names = None
            self._applied_precision = None
        if self.precision not in ('fp32', 'fp32x3', 'bf16', 'bf16_hybrid'):
            raise ValueError("precision must be 'fp32', 'fp32x3', 'bf16' or 'bf16_hybrid'")
        if self.precision == 'bf16' and (self.config.hidden_size % 64 or self.config.intermediate_size % 64):
            raise ValueError("precision='bf16' needs hidden_size and intermediate_size to be multiples of 64 (the "
                             "bf16-resident GEMMs run on 64-deep k-tiles); use 'bf16_hybrid' or 'fp32' for this model")
        if self.precision == 'fp32x3' and (self.config.hidden_size % 32 or self.config.intermediate_size % 32):
            raise ValueError("precision='fp32x3' needs hidden_size and intermediate_size to be multiples of 32 (its products "
                             "run on 32-deep k-tiles); use 'fp32' for this model")
        if self.precision in ('bf16', 'fp32x3'):
            # refreshed here whenever the parameters changed behind its back
            mirror = st.ensure_mirror(3 if self.precision == 'fp32x3' else 1)
            if self._applied_precision != (self.precision, mirror.data_ptr()):
                check(_lib.lib().uniter_model_set_weight_mirror(self._handle, ptr(st.flat_params), ptr(mirror),
                                                                st.numel), 'uniter_model_set_weight_mirror')
                check(_lib.lib().uniter_model_set_precision(self._handle, 3 if self.precision == 'fp32x3' else 2),
                      'uniter_model_set_precision')
                # fp32x3: the encoder layers' weights sit in the mirror in the paired-row layout (ParamStore.pair_dst)
                check(_lib.lib().uniter_model_set_weight_pairing(self._handle, int(self.precision == 'fp32x3' and st.mirror_paired())),
                      'uniter_model_set_weight_pairing')
                self._applied_precision = (self.precision, mirror.data_ptr())
        elif self._applied_precision != self.precision:
            check(_lib.lib().uniter_model_set_precision(self._handle, 1 if self.precision == 'bf16_hybrid' else 0),
                  'uniter_model_set_precision')
            self._applied_precision = self.precision
        return st

    def _destroy_handle(self):
        if getattr(self, '_handle', None) is not None:
            try:
                _lib.lib().uniter_model_destroy(self._handle)
            except Exception:       # interpreter shutdown
                pass
            self._handle = None

    def __del__(self):
        self._destroy_handle()

    def _get_ws(self, nbytes, mode):
        if mode != 0:
            # activations must survive until backward: fresh buffer per training forward.  Always the
            # largest size seen so far, so the caching allocator hands the same block back instead of
            # freeing / mallocing gigabytes whenever the batch layout (task, lengths) changes
            self._ws_high = max(getattr(self, '_ws_high', 0), nbytes)
            return torch.empty(self._ws_high, dtype=torch.uint8, device=self.embeddings.LayerNorm.weight.device)
        ws = self._ws_cache.get(0)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.embeddings.LayerNorm.weight.device)
            self._ws_cache[0] = ws
        return ws

    def _touched_names(self, batch):
        """Parameter names (in the root's namespace) that receive a gradient."""
        root = self._root()
        if self._prefix_names is None:
            ids = {id(p): n for n, p in root.named_parameters()}
            self._prefix_names = {n: ids[id(p)] for n, p in self.named_parameters()}
            self._touch_lists = {}
        key = (bool(batch.input_ids), bool(batch.img_feat), bool(batch.img_masks))
        if key in self._touch_lists:
            return self._touch_lists[key]
        out = []
        for local, full in self._prefix_names.items():
            if local.startswith('pooler.'):
                continue
            if local.startswith('img_embeddings.'):
                if not batch.img_feat:
                    continue
                if 'mask_embedding' in local and not batch.img_masks:
                    continue
            if local.startswith('embeddings.') and 'token_type' not in local and not batch.input_ids:
                continue
            out.append(full)
        self._touch_lists[key] = out
        return out

    def _run_backward(self, batch, d_hidden, all_layers, seed, offset, ws, nbytes):
        lib = _lib.lib()
        nl = self.config.num_hidden_layers
        st = _find_store(self._root())
        st.reattach_grads()
        main = torch.cuda.current_stream()
        side = None
        if self.use_side_stream:
            if self._side_stream is None:
                self._side_stream = _lib.shared_stream(d_hidden.device, 'side')
            side = self._side_stream
            side.wait_stream(main)
        side_ptr = C.c_void_p(side.cuda_stream) if side is not None else None
        if getattr(st, 'wgrad_stale', False):
            # the optimizer step left the encoder's weight gradients uncleared (trainer.FusedAdam, lazy_zero): this backward
            # pass overwrites them; the flag holds for one pass, later micro-batches accumulate
            check(lib.uniter_model_set_wgrad_overwrite(self._handle, 1), 'uniter_model_set_wgrad_overwrite')
            st.wgrad_stale = False
        np_ = self._norm_parts
        if np_ is not None and not 0 < self.norm_partials_per_layer() <= np_[1]:
            np_ = None           # (more slots than the trainer made room for -- a chip with more CUs: it reduces the buckets itself, as it decided)
        check(lib.uniter_model_set_norm_partials(self._handle, np_[0].data_ptr() if np_ is not None else None,
                                                 np_[1] if np_ is not None else 0), 'uniter_model_set_norm_partials')
        check(lib.uniter_model_backward_begin(self._handle, C.byref(batch), ptr(d_hidden),
                                              int(all_layers), seed, offset, ptr(ws), nbytes,
                                              _lib.cur_stream(), side_ptr), 'uniter_model_backward_begin')
        hook = self._grad_hook
        if hook is not None:
            hook('begin', None, side or main)
        for l in range(nl - 1, -1, -1):
            check(lib.uniter_model_backward_layer(self._handle, l), 'uniter_model_backward_layer')
            if hook is not None:
                hook('layer', l, side or main)
        check(lib.uniter_model_backward_embed(self._handle), 'uniter_model_backward_embed')
        st.touch(self._touched_names(batch))
        if hook is not None:
            hook('embed', None, main)

    def norm_partials_per_layer(self):
        """Partial sums of the clip norm each layer's backward leaves by itself in the CURRENT precision (fp32x3: its
        weight-gradient launch carries them), 0 = none: the caller reduces the layer's gradient bucket itself."""
        self._ensure_handle()          # (applies the precision to the handle)
        return int(_lib.lib().uniter_model_norm_partials_per_layer(self._handle))

    def _set_ready_events(self, events):
        """Per-block 'parameters final' events of an optimizer step overlapped with this forward
        (trainer.FusedAdam); consumed by the next uniter_model_forward."""
        self._ensure_handle()
        self._ready_events = list(events)            # keep the torch events alive until they are consumed
        arr = (C.c_void_p * len(events))(*[C.c_void_p(e.cuda_event) for e in events])
        check(_lib.lib().uniter_model_set_ready_events(self._handle, arr, len(events)),
              'uniter_model_set_ready_events')

    @staticmethod
    def lengths_from_mask(attention_mask):
        """Per-sample lengths of a right-padded attention mask as a host list (ONE device -> host synchronisation, in the
        caller's hands): what `seq_lens` wants when the batch does not carry it."""
        return [int(n) for n in attention_mask.sum(dim=1).to(torch.int64).cpu().tolist()]

    def _pack(self, b, keep, attention_mask, gather_index, seq_lens):
        """Token packing: hand the library the valid positions only.  The
        attention mask must be right-padded (1..1 0..0 per row, what get_attention_mask builds):
        padded keys then carry weight exp(-10000) = 0 in fp32 and padded queries feed nothing
        downstream, so dropping those rows changes no valid output and no gradient."""
        B, L = b.B, b.L
        if seq_lens is None:
            # the row count of the packed layout sizes the workspace on the host: it cannot come from a device tensor
            # without a device -> host synchronisation in the middle of the step, so the lengths are an input
            raise UniterHipError('pack_padded needs seq_lens (host list of tl + nbb per sample: batch["seq_lens"] of '
                                 'data.MemeDataset\'s collate / utils.make_synthetic_batch; for a one-off call '
                                 'UniterModel.lengths_from_mask(attention_mask) reads them back from the mask)')
        lens = np.asarray(torch.as_tensor(seq_lens).cpu().numpy(), dtype=np.int64).reshape(-1)
        if lens.shape[0] != B or (lens < 1).any() or (lens > L).any():
            raise ValueError('seq_lens must hold B values in [1, L]')
        cu = np.zeros(B + 1, dtype=np.int32)
        cu[1:] = np.cumsum(lens)
        dst = np.concatenate([np.arange(int(n), dtype=np.int64) + i * L for i, n in enumerate(lens)])
        dev = attention_mask.device
        # pinned staging + non_blocking copies: a pageable .to(device) waits for everything queued on the stream, i.e. the host
        # would lose its lead over the GPU once per step (same-box A/B of the packed fp32 step: 10.9 ms with pageable copies, 9.65 without)
        cu_h, dst_h = torch.from_numpy(cu).pin_memory(), torch.from_numpy(dst).pin_memory()
        cu_t, dst_t = cu_h.to(dev, non_blocking=True), dst_h.to(dev, non_blocking=True)
        keep.extend([cu_h, dst_h])         # the staging buffers live until the copies have run (with the batch's other tensors)
        S = (b.T if b.input_ids else 0) + (b.R if b.img_feat else 0)
        if gather_index is not None and b.input_ids and b.img_feat:
            flat = (gather_index + torch.arange(B, device=dev, dtype=torch.int64).unsqueeze(1) * S).reshape(-1)
            src_t = flat.index_select(0, dst_t)
        else:
            if S != L:
                raise ValueError('without gather_index the output length must equal T+R')
            src_t = dst_t
        keep.extend([cu_t, dst_t, src_t])
        b.cu_seqlens, b.pack_src, b.pack_dst, b.Mp = cu_t.data_ptr(), src_t.data_ptr(), dst_t.data_ptr(), int(cu[-1])

    # -- the reference's public surface ---------------------------------------
    def forward(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask,
                gather_index=None, img_masks=None, output_all_encoded_layers=True,
                txt_type_ids=None, img_type_ids=None, seq_lens=None):
        """Same signature / return as model/model.py:336-367: list of per-layer
        hidden states, or the last layer's [B, L, H] tensor.

        ``seq_lens`` (extension, host list of tl+nbb per sample) is read (and required) only when
        ``self.pack_padded`` is set: the packed layout's row count sizes the workspace on the host."""
        self._ensure_handle()
        dev = self.embeddings.LayerNorm.weight.device
        keep = []

        def prep(t, dtype, name):
            if t is None:
                return None
            if not torch.is_tensor(t):
                raise UniterHipError('%s must be a tensor' % name)
            if t.device != dev:
                raise UniterHipError('%s is on %s but the model is on %s (no implicit transfers: '
                                     'move the batch with batch_to_device)' % (name, t.device, dev))
            if t.dtype != dtype:
                t = t.to(dtype)
            t = t.contiguous()
            keep.append(t)
            return t

        if input_ids is None and img_feat is None:
            raise ValueError('need input_ids and/or img_feat')
        input_ids = prep(input_ids, torch.int64, 'input_ids')
        position_ids = prep(position_ids, torch.int64, 'position_ids')
        txt_type_ids = prep(txt_type_ids, torch.int64, 'txt_type_ids')
        img_feat = prep(img_feat, torch.float32, 'img_feat')
        img_pos_feat = prep(img_pos_feat, torch.float32, 'img_pos_feat')
        img_type_ids = prep(img_type_ids, torch.int64, 'img_type_ids')
        img_masks = prep(img_masks, torch.int64, 'img_masks')
        attention_mask = prep(attention_mask, torch.float32, 'attention_mask')
        gather_index = prep(gather_index, torch.int64, 'gather_index')
        if input_ids is not None and img_feat is not None and gather_index is None:
            raise ValueError('joint text+image input needs gather_index (model/model.py:327-333)')

        b = UniterBatchC()
        B = (input_ids if input_ids is not None else img_feat).shape[0]
        b.B = B
        b.T = input_ids.shape[1] if input_ids is not None else 0
        b.R = img_feat.shape[1] if img_feat is not None else 0
        b.L = attention_mask.shape[1]
        if attention_mask.shape[0] != B:
            raise ValueError('attention_mask batch size mismatch')
        if input_ids is not None:
            if position_ids is None:
                raise ValueError('position_ids required with input_ids')
            b.pos_bcast = 1 if (position_ids.shape[0] == 1 and B != 1) else 0
            if position_ids.shape[-1] != b.T:
                raise ValueError('position_ids length mismatch')
        if img_feat is not None:
            if img_feat.shape[2] != self.img_dim:
                raise ValueError('img_feat last dim %d != img_dim %d' % (img_feat.shape[2], self.img_dim))
            if img_pos_feat is None or tuple(img_pos_feat.shape) != (B, b.R, 7):
                raise ValueError('img_pos_feat must be [B, R, 7]')
        if gather_index is not None and tuple(gather_index.shape) != (B, b.L):
            raise ValueError('gather_index must be [B, L] with L = attention_mask.shape[1]')
        for fld, t in (('input_ids', input_ids), ('position_ids', position_ids),
                       ('txt_type_ids', txt_type_ids), ('img_feat', img_feat),
                       ('img_pos_feat', img_pos_feat), ('img_type_ids', img_type_ids),
                       ('img_masks', img_masks), ('attention_mask', attention_mask),
                       ('gather_index', gather_index)):
            setattr(b, fld, t.data_ptr() if t is not None else None)

        if self.pack_padded and b.L <= _lib.lib().uniter_attn_varlen_max_len():
            self._pack(b, keep, attention_mask, gather_index, seq_lens)

        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        mode = 0 if not grad else (1 if self.training else 2)
        seed, offset = self._seed, self._offset
        if mode == 1:
            self._offset = (self._offset + 1) & 0xFFFFFFFF
        anchor = self.embeddings.LayerNorm.weight if grad else self.embeddings.LayerNorm.weight.detach()
        hidden = _UniterFn.apply(anchor, self, b, keep, bool(output_all_encoded_layers), mode, seed, offset)
        if output_all_encoded_layers:
            return [hidden[i] for i in range(self.config.num_hidden_layers)]
        return hidden
