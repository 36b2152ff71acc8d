"""Host helpers that define the hot path's inputs (mirror of utils/utils.py:111-141
and of the loss / batch plumbing in train_template.py)."""
import random

import numpy as np
import torch


def get_gather_index(txt_lens, num_bbs, batch_size, max_len, out_size, device=None):
    """Row map of the joint sequence (contract of utils/utils.py:111-117): output position j of
    sample b reads row gi[b, j] of cat([text (max_len rows), image]).  Positions [tl, tl + nbb) read
    the image rows max_len .. max_len + nbb - 1, every other position reads row j (text, or padding
    nothing looks at).  Vectorised: one comparison grid instead of a loop over the batch; can be
    built directly on the device the model lives on."""
    tl = torch.as_tensor(txt_lens, dtype=torch.long, device=device).reshape(-1, 1)
    nbb = torch.as_tensor(num_bbs, dtype=torch.long, device=device).reshape(-1, 1)
    if not (tl.shape[0] == nbb.shape[0] == batch_size):
        raise ValueError('txt_lens / num_bbs must hold batch_size entries')
    col = torch.arange(out_size, dtype=torch.long, device=device).unsqueeze(0)
    in_img = (col >= tl) & (col < tl + nbb)
    return torch.where(in_img, col - tl + max_len, col.expand(batch_size, out_size))


def get_attention_mask(text_len, img_len, device=None):
    """[B, max(tl + nbb)] float mask, 1 on the tl + nbb valid positions of each row (contract of
    utils/utils.py:120-125)."""
    n = (torch.as_tensor(text_len, dtype=torch.long, device=device)
         + torch.as_tensor(img_len, dtype=torch.long, device=device)).reshape(-1, 1)
    col = torch.arange(int(n.max()), dtype=torch.long, device=device).unsqueeze(0)
    return (col < n).to(torch.float32)


def pad_tensors(tensors, lens=None, pad=0):
    """Stack B tensors [n_b, hid] into [B, max n_b, hid], filling with `pad` (contract of
    utils/utils.py:128-141; `lens` may shorten a tensor's used part)."""
    if lens is None:
        lens = [int(t.shape[0]) for t in tensors]
    first = tensors[0]
    out = first.new_full((len(tensors), max(lens), first.shape[-1]), pad)
    for row, (t, n) in enumerate(zip(tensors, lens)):
        out[row, :n] = t[:n]
    return out


def set_seed(seed):
    """utils/utils.py:100-107."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def make_synthetic_batch(B, T, R, seed=1234, vocab=28996, img_dim=2048, txt_lens=None,
                         num_bbs=None, pos_label_prob=0.36, device=None):
    """Synthetic batch with the layout of the reference collate_fn
    (data/meme_dataset.py:152-214): input_ids ~ U{1..V-1}
    with [CLS]=101 first, position_ids = arange(T), img_feat ~ |N(0,1)|,
    7-d box features (x1,y1,x2,y2,w,h,w*h), compact attention mask and
    gather_index, labels ~ Bernoulli(0.36).  Same PCG64 stream as the oracle's
    generator so fixtures can be rebuilt anywhere."""
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
    for b in range(B):
        ids[b, txt_lens[b]:] = 0
        feat[b, num_bbs[b]:] = 0
        pos[b, num_bbs[b]:] = 0
    attn = get_attention_mask(txt_lens, num_bbs)
    gi = get_gather_index(txt_lens, num_bbs, B, T, attn.shape[1])
    batch = {'input_ids': torch.from_numpy(ids),
             'position_ids': torch.arange(T, dtype=torch.int64).unsqueeze(0).repeat(B, 1),
             'img_feat': torch.from_numpy(feat),
             'img_pos_feat': torch.from_numpy(pos.astype(np.float32)),
             'attn_mask': attn, 'gather_index': gi,
             'labels': torch.from_numpy(labels)}
    if device is not None:
        batch = {k: v.to(device) for k, v in batch.items()}
    batch['seq_lens'] = [int(t) + int(n) for t, n in zip(txt_lens, num_bbs)]     # host list (see data.py collate)
    return batch


def make_synthetic_pretrain_batch(task, B, T, R, seed=1234, vocab=28996, img_dim=2048, txt_lens=None,
                                  num_bbs=None, mask_prob=0.15, device=None):
    """Synthetic batch for one pretraining task (BASELINE configs[4]) with the key
    names the reference's collates produce (`attn_masks` sic):

    * ``mlm``  -- BERT masking of pretrain_mlm.py:35-69: each real token is selected with probability
      0.15 (at least one per sample), 80 % -> [MASK]=103, 10 % -> random id, 10 % kept;
      ``txt_labels`` = original id at selected positions, -1 elsewhere;
    * ``mrfr`` -- region masking of pretrain_mrfr.py:29-51: ``img_masks`` ~ Bernoulli(0.15) over the
      real regions (at least one), ``img_mask_tgt`` = the same mask placed behind the text part of
      the gathered sequence, ``feat_targets`` = the masked regions' original features;
    * ``itm``  -- ``targets`` ~ Bernoulli(0.5) (pretrain_itm.py:27-47 swaps in another image for 0).
    """
    if task not in ('mlm', 'mrfr', 'itm'):
        raise ValueError('invalid task')
    base = make_synthetic_batch(B, T, R, seed=seed, vocab=vocab, img_dim=img_dim, txt_lens=txt_lens,
                                num_bbs=num_bbs)
    rng = np.random.Generator(np.random.PCG64(seed + 7919))
    txt_lens = [T] * B if txt_lens is None else list(txt_lens)
    num_bbs = [R] * B if num_bbs is None else list(num_bbs)
    batch = {'input_ids': base['input_ids'], 'position_ids': base['position_ids'][:1].contiguous(),
             'img_feat': base['img_feat'], 'img_pos_feat': base['img_pos_feat'],
             'attn_masks': base['attn_mask'], 'gather_index': base['gather_index']}
    L = base['attn_mask'].shape[1]
    if task == 'mlm':
        ids = batch['input_ids'].numpy().copy()
        labels = np.full((B, T), -1, dtype=np.int64)
        for b in range(B):
            n = txt_lens[b]
            sel = rng.random(n) < mask_prob
            sel[0] = False                                     # [CLS] is never masked
            if n > 1 and not sel.any():
                sel[1 + rng.integers(0, n - 1)] = True
            how = rng.random(n)
            for t in np.nonzero(sel)[0]:
                labels[b, t] = ids[b, t]
                if how[t] < 0.8:
                    ids[b, t] = 103 if vocab > 103 else vocab - 1
                elif how[t] < 0.9:
                    ids[b, t] = rng.integers(1, vocab)
        batch['input_ids'] = torch.from_numpy(ids)
        batch['txt_labels'] = torch.from_numpy(labels)
    elif task == 'mrfr':
        img_masks = np.zeros((B, R), dtype=bool)
        tgt = np.zeros((B, L), dtype=bool)
        for b in range(B):
            n = num_bbs[b]
            m = rng.random(n) < mask_prob
            if not m.any():
                m[rng.integers(0, n)] = True
            img_masks[b, :n] = m
            tgt[b, txt_lens[b]:txt_lens[b] + n] = m
        feat = batch['img_feat']
        batch['img_masks'] = torch.from_numpy(img_masks)
        batch['img_mask_tgt'] = torch.from_numpy(tgt)
        batch['feat_targets'] = feat[torch.from_numpy(img_masks)].contiguous()
    else:
        batch['targets'] = torch.from_numpy((rng.random(B) < 0.5).astype(np.int64))
    if device is not None:
        batch = {k: v.to(device) for k, v in batch.items()}
    batch['seq_lens'] = base['seq_lens']
    return batch
