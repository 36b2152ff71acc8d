"""GPU: the train_uniter.py CLI end to end on a synthetic dataset written in the reference's
on-disk format: data loader -> trainer loop -> fused step -> early-stopping bookkeeping ->
checkpoint ({'model_state_dict': ...}) -> CSV / JSON exports."""
import json
import os

import pytest
import torch

from common import TINY

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('ragged', [True, False])
def test_cli_trains_exports_and_checkpoint_reloads(tmp_path, ragged):
    """ragged=False is the reference's collate (every sample masked at the batch's largest region count: the zero rows
    are attended, data/meme_dataset.py:161,191); with it the synthetic signal -- a shift of the region features -- is
    diluted and three epochs do not learn it (AUROC 0.47 - 0.61 over seeds, 0.97 - 0.99 with per-sample masks)."""
    import train_uniter
    cfg = tmp_path / 'tiny.json'
    cfg.write_text(json.dumps(dict(TINY, vocab_size=28996, max_position_embeddings=64)))
    data_dir, model_dir = str(tmp_path / 'data'), str(tmp_path / 'ckpt')
    best, test_metrics = train_uniter.main([
        '--config', str(cfg), '--data_path', data_dir, '--model_path', model_dir, '--vis_path', str(tmp_path / 'vis'),
        '--synthetic', '48', '--batch_size', '8', '--max_epoch', '3', '--lr', '1e-3', '--warmup_steps', '2',
        '--gradient_accumulation', '2', '--pos_wt', '1.8', '--max_txt_len', '16', '--seed', '1', '--log_every', '3']
        + (['--ragged_regions'] if ragged else []))
    assert (0.9 if ragged else 0.0) < best['aucroc'] <= 1.0          # the synthetic labels are learnable from the features
    ck = torch.load(os.path.join(model_dir, 'best_model.pt'))
    assert set(ck) == {'model_state_dict'} and 'uniter_model.encoder.layer.1.output.dense.weight' in ck['model_state_dict']
    metrics = json.load(open(os.path.join(model_dir, 'best_model_metrics.json')))
    assert {'dev', 'train'} <= set(metrics) and 'loss' in metrics['dev']
    csv = open(os.path.join(model_dir, 'best_model_dev_seen_preds.csv')).read().splitlines()
    assert csv[0] == 'id,proba,label,gt' and len(csv) == 49
    assert 'test_seen' in test_metrics
    # the saved checkpoint reproduces the exported probabilities
    from meme_challenge_amd.model import UniterModel, UniterConfig
    from meme_challenge_amd.meme_uniter import MemeUniter
    c = UniterConfig.from_json_file(str(cfg))
    m = MemeUniter(UniterModel(c, 2048), c.hidden_size, 1)
    m.load_state_dict(ck['model_state_dict'])
    m = m.cuda().eval()
    from meme_challenge_amd.data import MemeDataset, HashTokenizer
    from functools import partial
    tok = partial(HashTokenizer(max_length=16), max_length=16)
    ds = MemeDataset(os.path.join(data_dir, 'dev_seen.jsonl'), os.path.join(data_dir, 'img_feats'), text_padding=tok,
                     return_ids=True, ragged_regions=ragged)
    b = ds.get_collate_fn()([ds[i] for i in range(8)])
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in b.items()}
    with torch.no_grad():
        p = torch.sigmoid(m(img_feat=b['img_feat'], img_pos_feat=b['img_pos_feat'], input_ids=b['input_ids'],
                            position_ids=b['position_ids'], attention_mask=b['attn_mask'],
                            gather_index=b['gather_index'], output_all_encoded_layers=False)).reshape(-1).cpu()
    exported = torch.tensor([float(l.split(',')[1]) for l in csv[1:9]])
    assert (p - exported).abs().max() < 1e-4


def test_device_prefetcher_yields_the_loader_batches_on_the_gpu():
    """DevicePrefetcher: same batches, same order as the wrapped DataLoader, tensors on the GPU, non-tensor
    entries (seq_lens) passed through; safe to consume while the next copy is in flight."""
    import torch
    from torch.utils import data
    from meme_challenge_amd.data import DevicePrefetcher

    class DS(data.Dataset):
        name = 'toy'

        def __len__(self):
            return 23

        def __getitem__(self, i):
            return {'x': torch.full((5, 7), float(i)), 'i': torch.tensor(i)}

    def collate(samples):
        return {'x': torch.stack([s['x'] for s in samples]), 'i': torch.stack([s['i'] for s in samples]),
                'seq_lens': [int(s['i']) for s in samples], 'none': None}

    loader = data.DataLoader(DS(), batch_size=4, collate_fn=collate, pin_memory=True)
    for threaded in (True, False):                    # (round 6: the loader's work in a background thread, the default; and inline)
        pf = DevicePrefetcher(loader, 'cuda', threaded=threaded)
        assert len(pf) == len(loader) and pf.dataset.name == 'toy' and pf.threaded is threaded
        seen = []
        for ref, got in zip(loader, pf):
            assert got['x'].is_cuda and got['i'].is_cuda and got['none'] is None and got['seq_lens'] == ref['seq_lens']
            y = (got['x'] * 2).sum()                      # consume on the compute stream
            assert torch.equal(got['x'].cpu(), ref['x']) and torch.equal(got['i'].cpu(), ref['i'])
            assert y.item() == ref['x'].sum().item() * 2
            seen.extend(got['i'].tolist())
        assert seen == list(range(23))
        assert sum(1 for _ in pf) == len(loader)          # re-iterable
        # leaving the loop early stops the background thread (no batch is fetched for nobody)
        for k, got in enumerate(pf):
            if k == 1:
                break
        import threading
        import time
        time.sleep(0.3)
        assert not any(t.name == 'uniter-prefetch' and t.is_alive() for t in threading.enumerate())

    class Boom(DS):
        def __getitem__(self, i):
            if i == 9:
                raise RuntimeError('broken sample')
            return super().__getitem__(i)
    pf = DevicePrefetcher(data.DataLoader(Boom(), batch_size=4, collate_fn=collate), 'cuda', threaded=True)
    try:
        for _ in pf:
            pass
        raised = False
    except RuntimeError as e:
        raised = 'broken sample' in str(e)
    assert raised                                     # a failure in the loader's thread reaches the trainer


def test_cli_with_shards_prefetch_and_packing(tmp_path):
    """The same run through the packed feature shard, the device prefetcher (default) and token packing."""
    import train_uniter
    cfg = tmp_path / 'tiny.json'
    cfg.write_text(json.dumps(dict(TINY, vocab_size=28996, max_position_embeddings=64)))
    data_dir, model_dir = str(tmp_path / 'data'), str(tmp_path / 'ckpt')
    best, _ = train_uniter.main([
        '--config', str(cfg), '--data_path', data_dir, '--model_path', model_dir, '--vis_path', str(tmp_path / 'vis'),
        '--synthetic', '48', '--batch_size', '8', '--max_epoch', '3', '--lr', '1e-3', '--warmup_steps', '2',
        '--pos_wt', '1.8', '--max_txt_len', '16', '--seed', '1', '--log_every', '3',
        '--feature_shards', '--pack_padded', '--ragged_regions'])
    assert os.path.isfile(os.path.join(data_dir, 'train_shard.feat.npy'))
    assert 0.5 < best['aucroc'] <= 1.0


def test_cli_cross_validation_folds_and_ensemble(tmp_path):
    """--num_folds 2: splits written in the reference's layout, one checkpoint and one set of prediction files per
    fold, and the ensemble of the folds' dev_seen predictions (utils/crossval.py:132-215)."""
    import train_uniter
    cfg = tmp_path / 'tiny.json'
    cfg.write_text(json.dumps(dict(TINY, vocab_size=28996, max_position_embeddings=64)))
    data_dir, model_dir = str(tmp_path / 'data'), str(tmp_path / 'ckpt')
    metrics = train_uniter.main([
        '--config', str(cfg), '--data_path', data_dir, '--model_path', model_dir, '--vis_path', str(tmp_path / 'vis'),
        '--synthetic', '64', '--batch_size', '8', '--max_epoch', '2', '--lr', '1e-3', '--warmup_steps', '2',
        '--max_txt_len', '16', '--seed', '1', '--log_every', '50', '--num_folds', '2', '--crossval_dev_size', '16',
        '--precision', 'bf16'])
    assert len(metrics) == 2 and all('aucroc' in m for m in metrics)
    cv = os.path.join(data_dir, 'crossval_16')
    assert {'train_00.jsonl', 'dev_00.jsonl', 'train_01.jsonl', 'dev_01.jsonl'} <= set(os.listdir(cv))
    for k in (0, 1):
        assert os.path.isfile(os.path.join(model_dir, 'best_model_fold_%d.pt' % k))
        assert os.path.isfile(os.path.join(model_dir, 'best_model_fold_%d_dev_seen_preds.csv' % k))
    ens = open(os.path.join(model_dir, 'best_model_dev_seen_ensemble.csv')).read().splitlines()
    assert ens[0] == 'id,gt,proba,label' and len(ens) == 65


def test_cli_two_ranks_shard_the_data_and_keep_their_replicas_equal(tmp_path):
    """train_uniter.py --parallel_computing on two ranks (gloo, both on cuda:0): rank 0 writes the dataset while rank 1
    waits, every rank draws its own batches, gradients are exchanged bucket by bucket, the replicas end with identical
    parameters (TrainerTemplate._check_replicas raises otherwise), rank 0 alone exports, nobody leaves early."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = tmp_path / 'tiny.json'
    cfg.write_text(json.dumps(dict(TINY, vocab_size=28996, max_position_embeddings=64)))
    data_dir, model_dir = str(tmp_path / 'data'), str(tmp_path / 'ckpt')
    args = [sys.executable, os.path.join(repo, 'train_uniter.py'), '--config', str(cfg), '--data_path', data_dir,
            '--model_path', model_dir, '--vis_path', str(tmp_path / 'vis'), '--synthetic', '64', '--batch_size', '8',
            '--max_epoch', '4', '--lr', '1e-3', '--warmup_steps', '2', '--pos_wt', '1.8', '--max_txt_len', '16',
            '--seed', '1', '--log_every', '3', '--ragged_regions', '--parallel_computing', 'True']
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29551', WORLD_SIZE='2', LOCAL_RANK='0',
               UNITER_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen(args, env=dict(env, RANK=str(r)), cwd=repo, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    assert 'data-parallel replicas agree' in outs[0] + outs[1]
    assert os.path.isfile(os.path.join(model_dir, 'best_model.pt'))
    json.load(open(os.path.join(model_dir, 'best_model_metrics.json')))
    # two ranks x batch 8 = one process x batch 16: the validation loss follows the same trajectory epoch by epoch (the
    # ranks draw the same 16 samples per step the single process draws; only the dropout streams differ)
    import re
    single = subprocess.run([a if a != '8' else '16' for a in args[:-2]] + ['--data_path', str(tmp_path / 'data1'),
                             '--model_path', str(tmp_path / 'ckpt1')],
                            env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE')}, cwd=repo,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert single.returncode == 0, single.stdout[-4000:]
    ev = lambda text: [float(x) for x in re.findall(r'eval_loss = ([0-9.]+)', text)]
    dp_loss, sp_loss = ev(outs[0]), ev(single.stdout)
    assert len(dp_loss) == len(sp_loss) == 4
    assert max(abs(a - b) for a, b in zip(dp_loss, sp_loss)) < 0.02, (dp_loss, sp_loss)
    csv = open(os.path.join(model_dir, 'best_model_dev_seen_preds.csv')).read().splitlines()
    assert csv[0] == 'id,proba,label,gt' and len(csv) == 65
