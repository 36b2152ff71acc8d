"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the pretraining heads used by BASELINE config 5 (ITM / MLM / MRFR) and of the
region-classification tasks next to them (MRC / MRC-kl):
UniterForPretraining.forward_{mlm,mrfr,itm,mrc} (model/pretrain.py:107-233),
RegionFeatureRegression (model/pretrain.py:19-33), RegionClassification (:36-47),
BertOnlyMLMHead (model/layer.py:188-233).
Weights use the reference's UniterForPretraining.state_dict() key names ("uniter." prefix).

prec='bf16' (the bf16 precision mode of the HIP path, BASELINE configs[2..4]): the encoder rounds what
uniter_oracle's bf16 mode rounds, and every dense product of the heads that runs on the bf16 matrix pipe --
the head transforms, the tied MLM decoder, the tied MRFR projection -- rounds both operands, forward and
backward (uniter_oracle._LinearB16); GELU, LayerNorm, the losses, the pooler and the two-way ITM head stay
fp32, as in meme_challenge_amd/pretrain.py."""
import torch
import torch.nn.functional as F

from . import uniter_oracle as O


def _masked_hidden(hidden, mask):
    # _compute_masked_hidden, model/pretrain.py:129-133
    m = mask.bool().unsqueeze(-1).expand_as(hidden)
    return hidden[m].contiguous().view(-1, hidden.size(-1))


def _head_transform(sd, p, x, prec='fp32'):
    # dense -> erf-GELU -> LayerNorm(eps 1e-12)   (model/layer.py:188-204; model/pretrain.py:22-25)
    return O.layer_norm(O.gelu(O.linear(x, sd[p[0] + 'weight'], sd[p[0] + 'bias'], prec)),
                        sd[p[1] + 'weight'], sd[p[1] + 'bias'])


def _encode(sd, cfg, batch, drop, img_masks=None, prec='fp32'):
    return O.uniter_forward(sd, cfg, batch['input_ids'], batch['position_ids'], batch['img_feat'],
                            batch['img_pos_feat'], batch['attn_masks'], batch['gather_index'],
                            img_masks=img_masks, output_all_encoded_layers=False, drop=drop, prefix='uniter.',
                            prec=prec)


def forward_mlm(sd, cfg, batch, compute_loss=True, drop=None, prec='fp32'):
    seq = _encode(sd, cfg, batch, drop, prec=prec)
    seq = seq[:, :batch['input_ids'].size(1), :]
    h = _masked_hidden(seq, batch['txt_labels'] != -1)
    h = _head_transform(sd, ('cls.predictions.transform.dense.', 'cls.predictions.transform.LayerNorm.'), h, prec)
    # tied decoder (model/layer.py:212-226): word_embeddings.weight + separate bias
    scores = O.linear(h, sd['uniter.embeddings.word_embeddings.weight'], None, prec) + sd['cls.predictions.bias']
    if not compute_loss:
        return scores
    tl = batch['txt_labels']
    return F.cross_entropy(scores, tl[tl != -1], reduction='none')


def forward_mrfr(sd, cfg, batch, compute_loss=True, drop=None, prec='fp32'):
    seq = _encode(sd, cfg, batch, drop, img_masks=batch['img_masks'], prec=prec)
    h = _masked_hidden(seq, batch['img_mask_tgt'])
    h = _head_transform(sd, ('feat_regress.net.0.', 'feat_regress.net.2.'), h, prec)
    # F.linear(hidden, img_linear.weight.t(), bias), model/pretrain.py:27,32
    pred = O.linear(h, sd['uniter.img_embeddings.img_linear.weight'].t(), sd['feat_regress.bias'], prec)
    if not compute_loss:
        return pred
    return F.mse_loss(pred, batch['feat_targets'], reduction='none')


def forward_itm(sd, cfg, batch, compute_loss=True, drop=None, prec='fp32'):
    seq = _encode(sd, cfg, batch, drop, prec=prec)
    pooled = O.pooler(sd, 'uniter.', seq)
    scores = F.linear(pooled, sd['itm_output.weight'], sd['itm_output.bias'])
    if not compute_loss:
        return scores
    return F.cross_entropy(scores, batch['targets'], reduction='none')


def forward_mrc(sd, cfg, batch, task='mrc', compute_loss=True, drop=None, prec='fp32'):
    # model/pretrain.py:205-233 (the region classifier's products stay fp32 in both modes: its label width, 1601, is
    # no multiple of 4 and the task is not part of BASELINE config 5)
    seq = _encode(sd, cfg, batch, drop, img_masks=batch['img_masks'], prec=prec)
    h = _masked_hidden(seq, batch['img_mask_tgt'])
    h = _head_transform(sd, ('region_classifier.net.0.', 'region_classifier.net.2.'), h)
    scores = F.linear(h, sd['region_classifier.net.3.weight'], sd['region_classifier.net.3.bias'])
    if not compute_loss:
        return scores
    lt = batch['label_targets']
    if 'kl' in task:
        return F.kl_div(F.log_softmax(scores, dim=-1), lt, reduction='none')
    # the background class (column 0) is never the target (:227-230)
    hard = torch.max(lt[:, 1:], dim=-1)[1] + 1
    return F.cross_entropy(scores, hard, ignore_index=0, reduction='none')


def synth_label_targets(n, label_dim, seed):
    """Detector-style soft labels for n masked regions: rows of a softmax, a few exact zeros."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed + 11))
    z = rng.standard_normal((n, label_dim)) * 2.0
    p = np.exp(z - z.max(1, keepdims=True))
    p[rng.random((n, label_dim)) < 0.1] = 0.0          # kl_div's xlogy(0, 0) = 0 branch
    p /= p.sum(1, keepdims=True)
    return torch.from_numpy(p.astype('float32'))


def synth_pretrain_batch(B, T, R, seed, vocab, img_dim, txt_lens, num_bbs, mask_prob=0.3):
    """A config-5 style batch (data/pretrain_mlm.py:95-131, pretrain_mrfr.py:88-131,
    pretrain_itm.py): MLM labels, region masks + regression targets, ITM targets."""
    import numpy as np
    b = O.synth_batch(B, T, R, seed=seed, vocab=vocab, img_dim=img_dim, txt_lens=txt_lens, num_bbs=num_bbs)
    rng = np.random.Generator(np.random.PCG64(seed + 7))
    ids = b['input_ids'].clone()
    txt_labels = torch.full_like(ids, -1)
    for i in range(B):
        picked = False
        for t in range(1, txt_lens[i]):
            if rng.random() < mask_prob:
                txt_labels[i, t] = ids[i, t]
                ids[i, t] = 3            # [MASK]-like id
                picked = True
        if not picked:
            txt_labels[i, 1] = ids[i, 1]
            ids[i, 1] = 3
    img_masks = torch.zeros(B, R, dtype=torch.bool)
    for i in range(B):
        for r in range(num_bbs[i]):
            img_masks[i, r] = rng.random() < mask_prob
        if not img_masks[i].any():
            img_masks[i, 0] = True
    L = b['attn_mask'].shape[1]
    img_mask_tgt = torch.zeros(B, L, dtype=torch.bool)
    for i in range(B):
        img_mask_tgt[i, txt_lens[i]:txt_lens[i] + R] = img_masks[i][:max(0, min(R, L - txt_lens[i]))]
    feat = b['img_feat']
    feat_targets = feat[img_masks.unsqueeze(-1).expand_as(feat)].contiguous().view(-1, feat.size(-1))
    out = {'input_ids': ids, 'input_ids_unmasked': b['input_ids'],
           'position_ids': torch.arange(T, dtype=torch.long).unsqueeze(0),
           'img_feat': feat, 'img_feat_masked': feat.masked_fill(img_masks.unsqueeze(-1), 0),
           'img_pos_feat': b['img_pos_feat'], 'attn_masks': b['attn_mask'], 'gather_index': b['gather_index'],
           'txt_labels': txt_labels, 'img_masks': img_masks, 'img_mask_tgt': img_mask_tgt,
           'feat_targets': feat_targets,
           'targets': torch.from_numpy((rng.random(B) < 0.5).astype('int64'))}
    return out
