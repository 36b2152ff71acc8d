"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Functional, CPU, fp32 restatement of the reference's UNITER forward pass.
Every function cites the reference file:line it follows (paths relative to
/root/reference).  Weights are passed as a plain ``{name: tensor}`` dict using
the reference's ``state_dict`` key names (``uniter_model.`` prefix for
``MemeUniter``).  Backward is obtained by torch autograd over this forward.

Dropout: ``drop=None`` is eval mode.  ``drop=DropSpec(...)`` applies the
counter-based masks of ``oracle/philox.py`` at exactly the reference's dropout
sites, so the HIP path can be compared in train mode with identical masks.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from . import philox


@dataclass
class DropSpec:
    seed: int
    offset: int
    p_hidden: float       # config.hidden_dropout_prob
    p_attn: float         # config.attention_probs_dropout_prob


def _apply_dropout(x, p, drop, site, index=None):
    """nn.Dropout semantics (inverted scaling) with the shared Philox mask.
    ``index``: optional int64 tensor (same shape as x) of linear element
    indices; default is x's own row-major linear index."""
    if drop is None or p <= 0.0:
        return x
    if index is None:
        keep = philox.keep_mask(x.numel(), p, drop.seed, drop.offset, site)
        keep = torch.from_numpy(keep).view(x.shape)
    else:
        n = int(index.max().item()) + 1
        keep_all = torch.from_numpy(
            philox.keep_mask(n, p, drop.seed, drop.offset, site))
        keep = keep_all[index.reshape(-1)].view(x.shape)
    scale = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(
        1.0 - p, dtype=torch.float32)
    return x * keep.to(x.dtype) * scale


def layer_norm(x, w, b):
    # apex FusedLayerNorm(hidden, eps=1e-12): biased variance, eps inside sqrt,
    # affine (model/model.py:229,252,253,258; model/layer.py:108,149)
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-12)


def gelu(x):
    # model/layer.py:31-37 (erf form)
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


# --- the bf16 precision mode of the HIP path (BASELINE configs[2..4]) ----------------------------
# prec='bf16' rounds exactly what the bf16 kernels round (round-to-nearest-even, fp32 accumulation
# everywhere) so that the HIP path can be held to a tight bar in that mode too:
#   * every dense product (img_linear, query/key/value, attention output, intermediate, output):
#     both operands rounded, forward and backward (dX = bf(dY) W_bf, dW = bf(dY)^T X_bf), bias
#     gradient from the unrounded dY;
#   * the fused query|key|value output is stored as bf16;
#   * gelu(u) is stored as bf16, and dU = bf(dH * bf(gelu'(u))) (gelu' is kept as bf16);
#   * attention: probabilities (after dropout) rounded where they multiply V; backward with dO,
#     Pd = p * mask and dS rounded where they become matrix operands, delta = sum(O * dO) in fp32.
#     The forward kernel rounds the UNNORMALISED probabilities of its blocked online softmax
#     (exp(s - running max)): _online_softmax_pv_b16 restates that blocking, because any rounding
#     that is modelled differently decorrelates every later bf16 rounding within a few layers.
# Everything else (LayerNorm, softmax, dropout, residual adds, pooler, head, loss) is fp32 in both.
def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _LinearB16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, round_out):
        xb, wb = bf(x), bf(w)
        y = F.linear(xb, wb, b)
        ctx.save_for_backward(xb, wb)
        ctx.has_bias = b is not None
        return bf(y) if round_out else y

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        dyb = bf(dy)
        dx = dyb @ wb
        dw = dyb.reshape(-1, dyb.shape[-1]).t() @ xb.reshape(-1, xb.shape[-1])
        db = dy.reshape(-1, dy.shape[-1]).sum(0) if ctx.has_bias else None
        return dx, dw, db, None


class _GeluB16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u):
        cdf = 0.5 * (1.0 + torch.erf(u / math.sqrt(2.0)))
        pdf = torch.exp(-0.5 * u * u) / math.sqrt(2.0 * math.pi)
        ctx.save_for_backward(bf(cdf + u * pdf))
        return bf(u * cdf)

    @staticmethod
    def backward(ctx, dh):
        (dg,) = ctx.saved_tensors
        return bf(dh * dg)


def _online_softmax_pv_b16(s, keep, v):
    """O = softmax(s) (x keep) V as csrc/attention_bf16.hip's forward kernel forms it: keys in blocks of 32, the
    key range cut in two halves (first half = ceil(nblk / 2) blocks) that each run an online softmax; block j
    multiplies V with bf16(exp(s - m_j) * keep), m_j = the half's running maximum after block j, and the fp32
    accumulator is rescaled by exp(m_old - m_new) in between; the normaliser sums the UNROUNDED exp(s - m).
    Returns (O, p) with p the exact softmax (what the backward kernel recomputes from the stored LSE)."""
    B, nh, L, _ = s.shape
    Lr = (L + 31) // 32 * 32
    nblk = Lr // 32
    hb = (nblk + 1) // 2
    ninf = float('-inf')
    sb = F.pad(s, (0, Lr - L), value=ninf).view(B, nh, L, nblk, 32)
    bmax = sb.max(-1).values
    halves = [torch.cummax(bmax[..., :hb], dim=-1).values]
    if nblk > hb:
        halves.append(torch.cummax(bmax[..., hb:], dim=-1).values)
    mj = torch.cat(halves, -1)                                        # [B, nh, L, nblk]
    mfin = mj.max(-1, keepdim=True).values
    w = torch.exp(mj - mfin)                                          # rescale of block j's contribution
    mj = torch.where(torch.isinf(mj), torch.zeros_like(mj), mj)      # a half whose keys are all masked: p = 0, w = 0
    pt = torch.exp(sb - mj.unsqueeze(-1))                             # unnormalised probabilities
    l = (pt.sum(-1) * w).sum(-1, keepdim=True)                        # [B, nh, L, 1]
    if keep is not None:
        pt = pt * F.pad(keep, (0, Lr - L), value=0.0).view(B, nh, L, nblk, 32)
    peff = (bf(pt) * w.unsqueeze(-1)).view(B, nh, L, Lr)[..., :L] / l
    return torch.matmul(peff, v), torch.softmax(s, dim=-1)


class _AttnB16(torch.autograd.Function):
    """q, k, v: [B, nh, L, d] holding bf16 values; ext_mask additive [B, 1, 1, L]; keep: dropout
    multiplier (0 or 1/(1-p)) or None."""
    @staticmethod
    def forward(ctx, q, k, v, ext_mask, keep, scale):
        s = torch.matmul(q, k.transpose(-1, -2)) * scale + ext_mask
        o, p = _online_softmax_pv_b16(s, keep, v)
        pd = p if keep is None else p * keep
        ctx.save_for_backward(q, k, v, p, pd, o)
        ctx.keep, ctx.scale = keep, scale
        return o

    @staticmethod
    def backward(ctx, do):
        # attn_b16_dq_kernel / attn_b16_dkv_kernel: p recomputed from the LSE (normalised), Pd = p * keep and
        # dS = p (dP keep - delta) scale rounded to bf16 where they multiply dO, K and Q; dO rounded too
        q, k, v, p, pd, o = ctx.saved_tensors
        keep, scale = ctx.keep, ctx.scale
        dob = bf(do)
        delta = (o * do).sum(-1, keepdim=True)
        dp = torch.matmul(dob, v.transpose(-1, -2))
        ds = p * ((dp if keep is None else dp * keep) - delta) * scale
        dsb, pdb = bf(ds), bf(pd)
        dq = torch.matmul(dsb, k)
        dk = torch.matmul(dsb.transpose(-1, -2), q)
        dv = torch.matmul(pdb.transpose(-1, -2), dob)
        return dq, dk, dv, None, None, None


def linear(x, w, b, prec='fp32', round_out=False):
    if prec == 'fp32':
        return F.linear(x, w, b)
    return _LinearB16.apply(x, w, b, round_out)


def text_embeddings(sd, p, input_ids, position_ids, token_type_ids, cfg, drop):
    # UniterTextEmbeddings.forward, model/model.py:232-245
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    e = (sd[p + 'embeddings.word_embeddings.weight'][input_ids]
         + sd[p + 'embeddings.position_embeddings.weight'][position_ids]
         + sd[p + 'embeddings.token_type_embeddings.weight'][token_type_ids])
    e = layer_norm(e, sd[p + 'embeddings.LayerNorm.weight'],
                   sd[p + 'embeddings.LayerNorm.bias'])
    return _apply_dropout(e, cfg['hidden_dropout_prob'], drop,
                          philox.SITE_TXT_EMB)


def image_embeddings(sd, p, img_feat, img_pos_feat, img_type_ids, img_masks,
                     cfg, drop, prec='fp32'):
    # UniterModel._compute_img_embeddings, model/model.py:311-319 and
    # UniterImageEmbeddings.forward, model/model.py:261-272
    if img_type_ids is None:
        img_type_ids = torch.ones_like(img_feat[:, :, 0].long())
    type_emb = sd[p + 'embeddings.token_type_embeddings.weight'][img_type_ids]
    q = p + 'img_embeddings.'
    if img_masks is not None:
        # model/model.py:262-265: row 0 of mask_embedding is forced to zero
        mw = sd[q + 'mask_embedding.weight']
        mw = torch.cat([torch.zeros_like(mw[:1]), mw[1:]], 0)
        img_feat = img_feat + mw[img_masks.long()]
    t_im = layer_norm(linear(img_feat, sd[q + 'img_linear.weight'],
                             sd[q + 'img_linear.bias'], prec),
                      sd[q + 'img_layer_norm.weight'],
                      sd[q + 'img_layer_norm.bias'])
    t_pos = layer_norm(F.linear(img_pos_feat, sd[q + 'pos_linear.weight'],
                                sd[q + 'pos_linear.bias']),
                       sd[q + 'pos_layer_norm.weight'],
                       sd[q + 'pos_layer_norm.bias'])
    e = layer_norm(t_im + t_pos + type_emb, sd[q + 'LayerNorm.weight'],
                   sd[q + 'LayerNorm.bias'])
    return _apply_dropout(e, cfg['hidden_dropout_prob'], drop,
                          philox.SITE_IMG_EMB)


def self_attention(sd, lp, x, ext_mask, cfg, drop, layer, prec='fp32'):
    # BertSelfAttention.forward, model/layer.py:75-101
    B, L, H = x.shape
    nh = cfg['num_attention_heads']
    d = H // nh

    def heads(t):  # transpose_for_scores, model/layer.py:70-73
        return t.view(B, L, nh, d).permute(0, 2, 1, 3)

    a = lp + 'attention.self.'
    rnd = prec != 'fp32'
    q = heads(linear(x, sd[a + 'query.weight'], sd[a + 'query.bias'], prec, rnd))
    k = heads(linear(x, sd[a + 'key.weight'], sd[a + 'key.bias'], prec, rnd))
    v = heads(linear(x, sd[a + 'value.weight'], sd[a + 'value.bias'], prec, rnd))
    if rnd:
        keep = None
        if drop is not None and cfg['attention_probs_dropout_prob'] > 0.0:
            Lp = (L + 3) // 4 * 4
            idx = (torch.arange(B * nh * L, dtype=torch.int64).view(B, nh, L, 1)
                   * Lp + torch.arange(L, dtype=torch.int64).view(1, 1, 1, L))
            keep = _apply_dropout(torch.ones(B, nh, L, L), cfg['attention_probs_dropout_prob'], drop,
                                  philox.site_attn_probs(layer), index=idx)
        c = _AttnB16.apply(q, k, v, ext_mask, keep, 1.0 / math.sqrt(d))
        return c.permute(0, 2, 1, 3).contiguous().view(B, L, H)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(d)
    s = s + ext_mask
    pr = torch.softmax(s, dim=-1)
    if drop is not None and cfg['attention_probs_dropout_prob'] > 0.0:
        Lp = (L + 3) // 4 * 4
        idx = (torch.arange(B * nh * L, dtype=torch.int64).view(B, nh, L, 1)
               * Lp + torch.arange(L, dtype=torch.int64).view(1, 1, 1, L))
        pr = _apply_dropout(pr, cfg['attention_probs_dropout_prob'], drop,
                            philox.site_attn_probs(layer), index=idx)
    c = torch.matmul(pr, v).permute(0, 2, 1, 3).contiguous().view(B, L, H)
    return c


def bert_layer(sd, lp, x, ext_mask, cfg, drop, layer, prec='fp32'):
    # BertLayer.forward, model/layer.py:166-170
    c = self_attention(sd, lp, x, ext_mask, cfg, drop, layer, prec)
    # BertSelfOutput.forward, model/layer.py:111-115
    o = lp + 'attention.output.'
    h = linear(c, sd[o + 'dense.weight'], sd[o + 'dense.bias'], prec)
    h = _apply_dropout(h, cfg['hidden_dropout_prob'], drop,
                       philox.site_attn_out(layer))
    y1 = layer_norm(h + x, sd[o + 'LayerNorm.weight'], sd[o + 'LayerNorm.bias'])
    # BertIntermediate.forward, model/layer.py:139-142
    u = linear(y1, sd[lp + 'intermediate.dense.weight'],
               sd[lp + 'intermediate.dense.bias'], prec)
    u = gelu(u) if prec == 'fp32' else _GeluB16.apply(u)
    # BertOutput.forward, model/layer.py:152-156
    o = lp + 'output.'
    h = linear(u, sd[o + 'dense.weight'], sd[o + 'dense.bias'], prec)
    h = _apply_dropout(h, cfg['hidden_dropout_prob'], drop,
                       philox.site_ffn_out(layer))
    return layer_norm(h + y1, sd[o + 'LayerNorm.weight'],
                      sd[o + 'LayerNorm.bias'])


def uniter_forward(sd, cfg, input_ids, position_ids, img_feat, img_pos_feat,
                   attention_mask, gather_index=None, img_masks=None,
                   output_all_encoded_layers=True, txt_type_ids=None,
                   img_type_ids=None, drop=None, prefix='', return_embed=False, prec='fp32'):
    """UniterModel.forward, model/model.py:336-367.  prec='bf16': the rounding points of the HIP
    path's bf16 mode (see _LinearB16 above); 'fp32' is the reference arithmetic."""
    assert prec in ('fp32', 'bf16')
    p = prefix
    ext = attention_mask.unsqueeze(1).unsqueeze(2).to(torch.float32)
    ext = (1.0 - ext) * -10000.0                      # model/model.py:342-345
    if input_ids is None:                              # image only, :348-351
        emb = image_embeddings(sd, p, img_feat, img_pos_feat, img_type_ids,
                               img_masks, cfg, drop, prec)
    elif img_feat is None:                             # text only, :352-355
        emb = text_embeddings(sd, p, input_ids, position_ids, txt_type_ids,
                              cfg, drop)
    else:                                              # joint, :321-334
        txt = text_embeddings(sd, p, input_ids, position_ids, txt_type_ids,
                              cfg, drop)
        img = image_embeddings(sd, p, img_feat, img_pos_feat, img_type_ids,
                               img_masks, cfg, drop, prec)
        gi = gather_index.unsqueeze(-1).expand(-1, -1, txt.shape[-1])
        emb = torch.gather(torch.cat([txt, img], dim=1), dim=1, index=gi)
    layers = []
    h = emb
    for i in range(cfg['num_hidden_layers']):          # model/model.py:282-292
        h = bert_layer(sd, p + 'encoder.layer.%d.' % i, h, ext, cfg, drop, i, prec)
        if output_all_encoded_layers:
            layers.append(h)
    out = layers if output_all_encoded_layers else h
    if return_embed:
        return out, emb
    return out


def pooler(sd, prefix, hidden):
    # BertPooler.forward, model/layer.py:179-185
    return torch.tanh(F.linear(hidden[:, 0], sd[prefix + 'pooler.dense.weight'],
                               sd[prefix + 'pooler.dense.bias']))


def meme_uniter_forward(sd, cfg, drop=None, prec='fp32', **kw):
    """MemeUniter.forward, model/meme_uniter.py:17-21 (kwargs as
    train_uniter.py:69-71 passes them)."""
    kw = dict(kw)
    kw.setdefault('output_all_encoded_layers', False)
    h = uniter_forward(sd, cfg, drop=drop, prefix='uniter_model.', prec=prec, **kw)
    if isinstance(h, list):
        h = h[-1]
    pooled = pooler(sd, 'uniter_model.', h)
    return F.linear(pooled, sd['linear.weight'], sd['linear.bias'])


# --- host helpers the path's inputs are defined by ---------------------------
def get_gather_index(txt_lens, num_bbs, batch_size, max_len, out_size):
    # utils/utils.py:111-117
    assert len(txt_lens) == len(num_bbs) == batch_size
    gi = np.tile(np.arange(out_size, dtype=np.int64), (batch_size, 1))
    for i, (tl, nbb) in enumerate(zip(txt_lens, num_bbs)):
        gi[i, tl:tl + nbb] = np.arange(max_len, max_len + nbb, dtype=np.int64)
    return torch.from_numpy(gi)


def get_attention_mask(text_len, img_len):
    # utils/utils.py:120-125 (pad_sequence of ones, padding 0)
    n = [t + i for t, i in zip(text_len, img_len)]
    m = np.zeros((len(n), max(n)), dtype=np.float32)
    for r, k in enumerate(n):
        m[r, :k] = 1.0
    return torch.from_numpy(m)


# --- parameter inventory (names, shapes) -------------------------------------
def param_shapes(cfg, img_dim=2048, n_classes=1, prefix='uniter_model.',
                 with_head=True):
    """Ordered (name, shape) list == reference ``MemeUniter.state_dict()``
    (model/model.py:217-305, model/layer.py:53-185, model/meme_uniter.py:8-15)."""
    H, I = cfg['hidden_size'], cfg['intermediate_size']
    p = prefix
    out = [(p + 'embeddings.word_embeddings.weight', (cfg['vocab_size'], H)),
           (p + 'embeddings.position_embeddings.weight',
            (cfg['max_position_embeddings'], H)),
           (p + 'embeddings.token_type_embeddings.weight',
            (cfg['type_vocab_size'], H)),
           (p + 'embeddings.LayerNorm.weight', (H,)),
           (p + 'embeddings.LayerNorm.bias', (H,)),
           (p + 'img_embeddings.img_linear.weight', (H, img_dim)),
           (p + 'img_embeddings.img_linear.bias', (H,)),
           (p + 'img_embeddings.img_layer_norm.weight', (H,)),
           (p + 'img_embeddings.img_layer_norm.bias', (H,)),
           (p + 'img_embeddings.pos_layer_norm.weight', (H,)),
           (p + 'img_embeddings.pos_layer_norm.bias', (H,)),
           (p + 'img_embeddings.pos_linear.weight', (H, 7)),
           (p + 'img_embeddings.pos_linear.bias', (H,)),
           (p + 'img_embeddings.mask_embedding.weight', (2, img_dim)),
           (p + 'img_embeddings.LayerNorm.weight', (H,)),
           (p + 'img_embeddings.LayerNorm.bias', (H,))]
    for i in range(cfg['num_hidden_layers']):
        lp = p + 'encoder.layer.%d.' % i
        for nm in ('query', 'key', 'value'):
            out += [(lp + 'attention.self.%s.weight' % nm, (H, H)),
                    (lp + 'attention.self.%s.bias' % nm, (H,))]
        out += [(lp + 'attention.output.dense.weight', (H, H)),
                (lp + 'attention.output.dense.bias', (H,)),
                (lp + 'attention.output.LayerNorm.weight', (H,)),
                (lp + 'attention.output.LayerNorm.bias', (H,)),
                (lp + 'intermediate.dense.weight', (I, H)),
                (lp + 'intermediate.dense.bias', (I,)),
                (lp + 'output.dense.weight', (H, I)),
                (lp + 'output.dense.bias', (H,)),
                (lp + 'output.LayerNorm.weight', (H,)),
                (lp + 'output.LayerNorm.bias', (H,))]
    out += [(p + 'pooler.dense.weight', (H, H)), (p + 'pooler.dense.bias', (H,))]
    if with_head:
        out += [('linear.weight', (n_classes, H)), ('linear.bias', (n_classes,))]
    return out


def synth_state_dict(cfg, seed=0, img_dim=2048, n_classes=1,
                     prefix='uniter_model.', ln_jitter=0.0):
    """Deterministic synthetic weights following the reference's init
    distribution (init_weights, model/model.py:133-146: Linear/Embedding
    weights ~ N(0, initializer_range), biases 0, LN weight 1 / bias 0) but
    drawn from a numpy PCG64 stream so that the same tensors can be rebuilt on
    any machine without shipping them.  ``ln_jitter`` > 0 perturbs biases and
    LN affine so that parity tests exercise them."""
    rng = np.random.Generator(np.random.PCG64(seed))
    std = cfg.get('initializer_range', 0.02)
    sd = {}
    for name, shape in param_shapes(cfg, img_dim, n_classes, prefix):
        is_ln = ('LayerNorm' in name or 'layer_norm' in name)
        if name.endswith('.weight') and not is_ln:
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
        elif name.endswith('.weight'):
            w = np.ones(shape, np.float32)
            if ln_jitter:
                w += rng.standard_normal(shape, dtype=np.float32) * np.float32(ln_jitter)
        else:
            w = np.zeros(shape, np.float32)
            if ln_jitter:
                w += rng.standard_normal(shape, dtype=np.float32) * np.float32(ln_jitter)
        sd[name] = torch.from_numpy(w)
    return sd


def synth_batch(B, T, R, seed=1234, vocab=28996, img_dim=2048, txt_lens=None,
                num_bbs=None, pos_label_prob=0.36):
    """Synthetic batch with the reference collate_fn's layout
    (data/meme_dataset.py:152-214).  ``txt_lens``/``num_bbs``
    None = full-length (headline benchmark); lists = ragged/compact batch."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ids = rng.integers(1, vocab, size=(B, T), dtype=np.int64)
    ids[:, 0] = 101 if vocab > 101 else 1
    feat = np.abs(rng.standard_normal((B, R, img_dim), dtype=np.float32))
    xy = rng.random((B, R, 2), dtype=np.float32) * np.float32(0.7)
    wh = rng.random((B, R, 2), dtype=np.float32) * np.float32(0.25) + np.float32(0.05)
    pos = np.concatenate([xy, xy + wh, wh, wh[..., :1] * wh[..., 1:]], axis=-1)
    labels = (rng.random(B) < pos_label_prob).astype(np.int64)
    if txt_lens is None:
        txt_lens = [T] * B
    if num_bbs is None:
        num_bbs = [R] * B
    for b in range(B):                       # pad like the tokenizer / pad_sequence
        ids[b, txt_lens[b]:] = 0
        feat[b, num_bbs[b]:] = 0
        pos[b, num_bbs[b]:] = 0
    attn = get_attention_mask(txt_lens, num_bbs)
    gi = get_gather_index(txt_lens, num_bbs, B, T, attn.shape[1])
    return {'input_ids': torch.from_numpy(ids),
            'position_ids': torch.arange(T, dtype=torch.int64).unsqueeze(0).repeat(B, 1),
            'img_feat': torch.from_numpy(feat),
            'img_pos_feat': torch.from_numpy(pos.astype(np.float32)),
            'attn_mask': attn, 'gather_index': gi,
            'labels': torch.from_numpy(labels)}
