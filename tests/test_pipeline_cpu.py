"""CPU: input pipeline (on-disk feature format -> batch dict), sampler, metrics, CLI surface."""
import json
import os

import numpy as np
import pytest
import torch


def test_dataset_reads_reference_disk_format_and_collates(tmp_path):
    from meme_challenge_amd.data import MemeDataset, HashTokenizer, write_synthetic_dataset, ConfounderSampler
    from functools import partial
    root = str(tmp_path)
    feat_dir = write_synthetic_dataset(root, n=9, num_bb=(3, 7), img_dim=16, seed=1, splits=('train',))
    tok = partial(HashTokenizer(vocab_size=2000), max_length=12, padding='max_length', truncation=True,
                  return_tensors='pt', return_length=True)
    # ragged_regions=True: every sample masked at its own region count (the default reproduces the reference's collate,
    # which counts the padded rows: tests/test_data_pipeline_cpu.py)
    ds = MemeDataset(os.path.join(root, 'train.jsonl'), feature_dir=feat_dir, text_padding=tok, return_ids=True,
                     ragged_regions=True)
    assert len(ds) == 9 and ds.name == 'train'
    s = ds[0]
    info = np.load(os.path.join(feat_dir, '00000_info.npy'), allow_pickle=True).item()
    bb = info['bbox'][0]
    exp = np.array([bb[0] / 640, bb[1] / 480, bb[2] / 640, bb[3] / 480, (bb[2] - bb[0]) / 640, (bb[3] - bb[1]) / 480,
                    (bb[2] - bb[0]) / 640 * (bb[3] - bb[1]) / 480], np.float32)
    assert np.allclose(s['img_pos_feat'][0].numpy(), exp, atol=1e-6)      # dataset_template.py:98-113
    batch = ds.get_collate_fn()([ds[i] for i in range(4)])
    B, T = batch['input_ids'].shape
    nbb = [ds[i]['img_feat'].shape[0] for i in range(4)]
    tl = [int((batch['input_ids'][i] != 0).sum()) for i in range(4)]
    L = max(t + n for t, n in zip(tl, nbb))
    assert batch['attn_mask'].shape == (4, L) and batch['gather_index'].shape == (4, L)
    for i in range(4):
        assert batch['attn_mask'][i].sum().item() == tl[i] + nbb[i]
        assert batch['gather_index'][i, tl[i]:tl[i] + nbb[i]].tolist() == list(range(T, T + nbb[i]))
        assert torch.equal(batch['img_feat'][i, nbb[i]:], torch.zeros_like(batch['img_feat'][i, nbb[i]:]))
    assert batch['position_ids'].tolist() == [list(range(T))] * 4
    assert set(batch) >= {'input_ids', 'position_ids', 'img_feat', 'img_pos_feat', 'attn_mask', 'gather_index',
                          'labels', 'ids'}
    # confounders: same text with both labels is repeated repeat_factor times
    ds.data.text[0] = ds.data.text[1] = 'same text'
    ds.data.labels[0], ds.data.labels[1] = 0, 1
    smp = ConfounderSampler(ds, repeat_factor=3)
    order = list(iter(smp))
    assert order.count(0) == 3 and order.count(1) == 3 and len(order) == 7 + 6


def test_metrics_against_reference_golden(host_helpers):
    from meme_challenge_amd.metrics import standard_metrics, find_optimal_threshold, aucroc
    z = host_helpers
    for k in range(int(z['metrics/n'])):
        p, y = torch.from_numpy(z['metrics/%d/probs' % k]), torch.from_numpy(z['metrics/%d/labels' % k])
        m = standard_metrics(p, y, add_optimal_acc=True)
        ref = json.loads(str(z['metrics/%d/ref' % k]))
        for key, v in ref.items():
            assert abs(m[key] - v) < 1e-6, (k, key, m[key], v)
    assert aucroc(torch.tensor([0.2, 0.9]), torch.tensor([1, 1])) == 0.0


def test_cli_flags_match_reference_surface():
    import train_uniter
    from meme_challenge_amd.model import resolve_config
    p = train_uniter.build_parser()
    a = p.parse_args([])
    ref_defaults = dict(data_path='./dataset', model_path='./model_checkpoints', vis_path='./vis_checkpoints',
                        model_save_name='best_model.pt', optimizer='adam', loss_func='bce_logits',
                        optimize_for='aucroc', scheduler='warmup_cosine', confounder_repeat=1,
                        object_conf_thresh=0.0, num_folds=0, crossval_dev_size=300, beta1=0.9, beta2=0.999,
                        batch_size=8, num_workers=0, gradient_accumulation=1, max_grad_norm=5, pos_wt=1,
                        lr=1e-4, warmup_steps=50, weight_decay=1e-3, max_epoch=20, lr_decay_step=3,
                        lr_decay_factor=0.8, patience=5, early_stop_thresh=1e-3, seed=42, log_every=2000,
                        parallel_computing=False, config='./config/uniter-base.json',
                        feature_path='./dataset/img_feats', max_txt_len=60, conf_th=0.2, max_bb=100, min_bb=10,
                        num_bb=36, fc_dim=64, dropout=0.2)
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k
    c = resolve_config('./config/uniter-base.json')
    assert (c.hidden_size, c.num_hidden_layers, c.vocab_size) == (768, 12, 28996)
    assert resolve_config('uniter-large').intermediate_size == 4096
    with pytest.raises(ValueError):
        resolve_config('./config/nope.json')


def test_feature_shard_equals_per_sample_files(tmp_path):
    """The packed, memory-mapped shard yields exactly what the reference's per-sample
    `<id>.npy` / `<id>_info.npy` files yield, sample by sample and after collate."""
    from meme_challenge_amd.data import (MemeDataset, HashTokenizer, write_synthetic_dataset, build_feature_shard,
                                         FeatureShard)
    from functools import partial
    root = str(tmp_path)
    feat_dir = write_synthetic_dataset(root, n=7, num_bb=(2, 6), img_dim=8, seed=4, splits=('train',))
    tok = partial(HashTokenizer(vocab_size=500), max_length=10, padding='max_length', truncation=True,
                  return_tensors='pt', return_length=True)
    plain = MemeDataset(os.path.join(root, 'train.jsonl'), feature_dir=feat_dir, text_padding=tok)
    prefix = os.path.join(root, 'train_shard')
    index = build_feature_shard(feat_dir, plain.data.ids.tolist(), prefix)
    assert index['offsets'][-1] == sum(plain[i]['img_feat'].shape[0] for i in range(len(plain)))
    packed = MemeDataset(os.path.join(root, 'train.jsonl'), feature_dir=None, text_padding=tok, feature_shard=prefix,
                         confidence_threshold=0.0)
    for i in range(len(plain)):
        a, b = plain[i], packed[i]
        assert torch.equal(a['img_feat'], b['img_feat']) and torch.equal(a['img_pos_feat'], b['img_pos_feat'])
    ba = plain.get_collate_fn()([plain[i] for i in range(5)])
    bb = packed.get_collate_fn()([packed[i] for i in range(5)])
    for k in ('input_ids', 'img_feat', 'img_pos_feat', 'attn_mask', 'gather_index', 'labels'):
        assert torch.equal(ba[k], bb[k]), k
    assert ba['seq_lens'] == bb['seq_lens']
    # the confidence filter sees the same detector scores
    thr = float(np.median(FeatureShard(prefix).conf[:]))
    pa = MemeDataset(os.path.join(root, 'train.jsonl'), feature_dir=feat_dir, text_padding=tok, confidence_threshold=thr)
    pb = MemeDataset(os.path.join(root, 'train.jsonl'), text_padding=tok, feature_shard=prefix, confidence_threshold=thr)
    for i in range(len(pa)):
        assert torch.equal(pa[i]['img_feat'], pb[i]['img_feat'])
    with pytest.raises(AssertionError):
        build_feature_shard(feat_dir, plain.data.ids.tolist()[:3], prefix + '_small')
        MemeDataset(os.path.join(root, 'train.jsonl'), text_padding=tok, feature_shard=prefix + '_small')


def test_rank_shards_are_disjoint_cover_the_epoch_and_have_equal_length():
    """train_uniter.RankShard / dp.shard_indices: with the same seed every rank draws the same epoch order from the
    inner sampler and takes every world-th index of it -- disjoint, covering, padded to a common length."""
    import random
    from meme_challenge_amd.dp import shard_indices
    import train_uniter
    order = list(range(37))
    random.Random(5).shuffle(order)
    for world in (2, 3, 8):
        shards = [shard_indices(order, r, world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1 and len(shards[0]) == -(-37 // world)
        flat = [i for s in shards for i in s]
        assert set(flat) == set(order)
        core = [i for k in range(len(shards[0])) for s in shards for i in [s[k]]][:37]
        assert core == order                                   # rank-major interleave reproduces the order: no sample skipped
        assert len(flat) - len(set(flat)) == world * len(shards[0]) - 37      # only the wrap-around padding repeats

    class Inner(object):
        def __iter__(self):
            random.seed(11)
            o = list(range(10))
            random.shuffle(o)
            return iter(o)

        def __len__(self):
            return 10
    a, b = train_uniter.RankShard(Inner(), 0, 2), train_uniter.RankShard(Inner(), 1, 2)
    assert len(a) == len(b) == 5 and set(a).isdisjoint(set(b)) and set(a) | set(b) == set(range(10))


def test_runtime_options_survive_the_final_reload():
    """--precision / --pack_padded belong to the encoder object; TrainerTemplate.end_training rebuilds the model through
    load_model(), which must apply them again (the final validation / test scoring would otherwise run in fp32, padded)."""
    import train_uniter
    t = object.__new__(train_uniter.TrainerUniter)
    t.config = {'config': 'uniter-base', 'n_classes': 1, 'precision': 'bf16', 'pack_padded': True}
    t.model_file = '/nonexistent/best_model.pt'
    import torch
    base = dict(train_uniter.resolve_config('uniter-base').to_dict(), num_hidden_layers=1, vocab_size=200)
    real = train_uniter.resolve_config
    train_uniter.resolve_config = lambda _: train_uniter.UniterConfig.from_dict(base)
    try:
        t.load_model()
    finally:
        train_uniter.resolve_config = real
    assert t.model.uniter_model.precision == 'bf16' and t.model.uniter_model.pack_padded is True
