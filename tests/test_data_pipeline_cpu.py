"""CPU: the input pipeline (meme_challenge_amd/data.py) against outputs of the REFERENCE's own
`MemeDataset.__getitem__`, `collate_fn`, `ConfounderSampler` and `_load_img_feature`
(data/meme_dataset.py:27-271, data/dataset_template.py:92-114), captured by
tests/golden/make_golden.py::gen_data_pipeline into tests/golden/data_pipeline.npz.  The fixture
carries the raw dataset content; this test rebuilds the reference-format files from it."""
import json
import os
import random

import numpy as np
import pytest
import torch

from common import simple_tokenizer
from conftest import load_golden


@pytest.fixture(scope='module')
def pipe(tmp_path_factory):
    z = load_golden('data_pipeline.npz')
    root = str(tmp_path_factory.mktemp('ds'))
    fdir = os.path.join(root, 'img_feats')
    os.makedirs(fdir)
    ids, texts, labels = z['raw/ids'].tolist(), z['raw/texts'].tolist(), z['raw/labels'].tolist()
    with open(os.path.join(root, 'train.jsonl'), 'w') as f:
        for i, sid in enumerate(ids):
            name = str(sid).zfill(5)
            info = {'bbox': z['raw/%d/bbox' % i].copy(), 'image_width': int(z['raw/%d/wh' % i][0]),
                    'image_height': int(z['raw/%d/wh' % i][1]), 'objects': z['raw/%d/objects' % i]}
            for key in ('objects_conf', 'cls_prob'):
                if 'raw/%d/%s' % (i, key) in z.files:
                    info[key] = z['raw/%d/%s' % (i, key)]
            np.save(os.path.join(fdir, name + '.npy'), z['raw/%d/feat' % i])
            np.save(os.path.join(fdir, name + '_info.npy'), info, allow_pickle=True)
            f.write(json.dumps({'id': sid, 'img': 'img/%s.png' % name, 'label': labels[i], 'text': texts[i]}) + '\n')
    return z, root, fdir


@pytest.mark.parametrize('thr', [0.0, 0.45])
def test_items_and_batches_match_reference(pipe, thr):
    from meme_challenge_amd.data import MemeDataset
    z, root, fdir = pipe
    tag = 'thr%g' % thr
    ds = MemeDataset(filepath=os.path.join(root, 'train.jsonl'), feature_dir=fdir, preload_images=False,
                     text_padding=simple_tokenizer, return_ids=True, confidence_threshold=thr)
    assert ds.name == 'train' and len(ds) == len(z['raw/ids'])
    for i in range(len(ds)):
        it = ds[i]
        assert torch.equal(it['img_feat'], torch.from_numpy(z['%s/item/%d/img_feat' % (tag, i)])), i
        assert torch.equal(it['img_pos_feat'], torch.from_numpy(z['%s/item/%d/img_pos_feat' % (tag, i)])), i   # bit-exact box features
        assert [int(it['label']), int(it['data_id'])] == z['%s/item/%d/label_id' % (tag, i)].tolist()
    collate = ds.get_collate_fn()
    bi = 0
    while '%s/batch/%d/idxs' % (tag, bi) in z.files:
        idxs = z['%s/batch/%d/idxs' % (tag, bi)].tolist()
        b = collate([ds[i] for i in idxs])
        for k in ('input_ids', 'position_ids', 'img_feat', 'img_pos_feat', 'token_type_ids', 'attn_mask', 'gather_index',
                  'labels', 'ids'):
            ref = torch.from_numpy(z['%s/batch/%d/%s' % (tag, bi, k)])
            assert b[k].dtype == ref.dtype and torch.equal(b[k], ref), (bi, k)
        bi += 1
    assert bi == 3


@pytest.mark.parametrize('rep', [1, 2, 3])
def test_confounder_sampler_matches_reference(pipe, rep):
    """Same confounder set and -- the sampler draws from python's global `random` -- the same epoch orders for the
    same seed (construction shuffles once, every __iter__ again)."""
    from meme_challenge_amd.data import MemeDataset, ConfounderSampler
    z, root, fdir = pipe
    ds = MemeDataset(filepath=os.path.join(root, 'train.jsonl'), feature_dir=fdir, text_padding=simple_tokenizer)
    random.seed(1000 + rep)
    sm = ConfounderSampler(ds, repeat_factor=rep)
    assert sm.confounders == z['sampler/%d/confounders' % rep].tolist()
    assert len(sm) == int(z['sampler/%d/len' % rep])
    assert list(iter(sm)) == z['sampler/%d/epoch0' % rep].tolist()
    assert list(iter(sm)) == z['sampler/%d/epoch1' % rep].tolist()
