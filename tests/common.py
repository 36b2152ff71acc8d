"""Shared helpers for the parity tests (test infrastructure)."""
import json

import numpy as np
import torch

TINY = dict(vocab_size=97, hidden_size=128, num_hidden_layers=2,
            num_attention_heads=2, intermediate_size=256, hidden_act='gelu',
            hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
            max_position_embeddings=40, type_vocab_size=2,
            initializer_range=0.02)
TINY_IMG_DIM = 64

BASE = dict(attention_probs_dropout_prob=0.1, hidden_act='gelu',
            hidden_dropout_prob=0.1, hidden_size=768, initializer_range=0.02,
            intermediate_size=3072, max_position_embeddings=512,
            num_attention_heads=12, num_hidden_layers=12, type_vocab_size=2,
            vocab_size=28996)
LARGE = dict(BASE, hidden_size=1024, intermediate_size=4096,
             num_attention_heads=16, num_hidden_layers=24)


def sd_from_npz(z, prefix='sd/'):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files
            if k.startswith(prefix)}


def batch_from_npz(z, prefix='in/'):
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files
            if k.startswith(prefix)}


def model_kwargs(batch):
    # the kwargs train_uniter.py:69-71 passes
    return dict(img_feat=batch['img_feat'], img_pos_feat=batch['img_pos_feat'],
                input_ids=batch['input_ids'], position_ids=batch['position_ids'],
                attention_mask=batch['attn_mask'],
                gather_index=batch['gather_index'],
                output_all_encoded_layers=False)


def maxdiff(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(a).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(b).double()
    return (a - b).abs().max().item()


def simple_tokenizer(texts, max_length=12):
    """Deterministic offline tokenizer with the return fields of the BertTokenizer partial the reference builds
    (train_uniter.py:124-126: input_ids padded to max_length, length, attention_mask, token_type_ids).  Used on BOTH
    sides of the data-pipeline fixture (tests/golden/make_golden.py gen_data_pipeline)."""
    ids = torch.zeros(len(texts), max_length, dtype=torch.long)
    lens = []
    for r, t in enumerate(texts):
        toks = [101] + [1000 + sum(ord(c) * (i + 1) for i, c in enumerate(w)) % 20000 for w in str(t).split()][:max_length - 2] + [102]
        ids[r, :len(toks)] = torch.tensor(toks)
        lens.append(len(toks))
    return {'input_ids': ids, 'length': torch.tensor(lens), 'attention_mask': (ids != 0).long(),
            'token_type_ids': torch.zeros_like(ids)}
