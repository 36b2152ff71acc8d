#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the authoring container (needs /root/reference).  The reference
is imported unmodified; the only shims are import stubs for packages that are
absent here:
  * apex.normalization.fused_layer_norm.FusedLayerNorm -> torch.nn.LayerNorm
    (Apex's own CPU fallback is F.layer_norm; this IS the reference CPU path)
  * torch.utils.tensorboard.SummaryWriter, seaborn  -> empty stubs so that
    train_template.py imports (neither is touched by calculate_loss)

Outputs (all small):
  tiny_model.npz     reference-initialised tiny UNITER (2 layers, H=128) with
                     inputs, every intermediate the path exposes, logits, loss
                     and ALL parameter gradients
  shapes_base.npz    UNITER-base (config/uniter-base.json) at BASELINE config-1
                     and config-2 shapes with PCG64-synthesised weights
                     (oracle.synth_state_dict): logits, loss, per-parameter
                     gradient norms and a few gradient slices
  data_pipeline.npz  MemeDataset.__getitem__ / collate_fn / ConfounderSampler of the reference on a small synthetic
                     dataset in its on-disk format (raw content included so the test rebuilds the files)
  shapes_large.npz   UNITER-large (config 4 shape): logits, loss, gradient norms and slices
  host_helpers.npz   get_gather_index / get_attention_mask outputs for ragged
                     lists; state_dict key names; LR-schedule values from
                     transformers; param-group split from get_optimizer
  trainer_steps.npz  TrainerTemplate.calculate_loss driven for several
                     iterations (gradient_accumulation=2, clip, Adam/AdamW,
                     warm-up cosine): losses and parameters after every iteration
"""
import os
import sys
import types
import json

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'


def install_stubs():
    apex = types.ModuleType('apex')
    norm = types.ModuleType('apex.normalization')
    fln = types.ModuleType('apex.normalization.fused_layer_norm')
    fln.FusedLayerNorm = torch.nn.LayerNorm
    sys.modules.update({'apex': apex, 'apex.normalization': norm,
                        'apex.normalization.fused_layer_norm': fln})
    tb = types.ModuleType('torch.utils.tensorboard')
    tb.SummaryWriter = object
    sys.modules['torch.utils.tensorboard'] = tb
    sys.modules['seaborn'] = types.ModuleType('seaborn')


install_stubs()
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

from model.model import UniterModel, UniterConfig          # noqa: E402
from model.meme_uniter import MemeUniter                    # noqa: E402
from utils.utils import get_gather_index, get_attention_mask  # noqa: E402
from utils.optim_utils import get_optimizer                 # noqa: E402
import train_template                                       # noqa: E402
from transformers import (get_cosine_schedule_with_warmup,  # noqa: E402
                          get_linear_schedule_with_warmup)

from oracle import uniter_oracle as O                       # noqa: E402

torch.set_num_threads(8)

TINY = dict(vocab_size=97, hidden_size=128, num_hidden_layers=2,
            num_attention_heads=2, intermediate_size=256, hidden_act='gelu',
            hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
            max_position_embeddings=40, type_vocab_size=2,
            initializer_range=0.02)
TINY_IMG_DIM = 64


def np_sd(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def kwargs_of(batch):
    # exactly the kwargs train_uniter.py:69-71 passes
    return dict(img_feat=batch['img_feat'], img_pos_feat=batch['img_pos_feat'],
                input_ids=batch['input_ids'], position_ids=batch['position_ids'],
                attention_mask=batch['attn_mask'], gather_index=batch['gather_index'],
                output_all_encoded_layers=False)


def build_ref(cfg_dict, img_dim, seed=None, sd=None):
    cfg = UniterConfig.from_dict(cfg_dict)
    if seed is not None:
        torch.manual_seed(seed)
    um = UniterModel(cfg, img_dim=img_dim)
    m = MemeUniter(uniter_model=um, hidden_size=cfg.hidden_size, n_classes=1)
    if sd is not None:
        m.load_state_dict(sd)
    return m


def jitter_(model, seed):
    """Make biases / LN affine non-trivial (reference init has them at 0/1, which
    would hide bias / gamma / beta bugs).  Uses torch RNG; values are stored."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('bias') or 'LayerNorm' in n or 'layer_norm' in n:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)


def gen_tiny():
    m = build_ref(TINY, TINY_IMG_DIM, seed=7)
    jitter_(m, 11)
    m.eval()
    B, T, R = 3, 10, 6
    tl, nbb = [10, 4, 7], [6, 6, 3]
    batch = O.synth_batch(B, T, R, seed=5, vocab=TINY['vocab_size'],
                          img_dim=TINY_IMG_DIM, txt_lens=tl, num_bbs=nbb)
    # cross-check the oracle's helpers against the reference's own
    assert torch.equal(batch['gather_index'],
                       get_gather_index(tl, nbb, B, T, batch['attn_mask'].shape[1]))
    assert torch.equal(batch['attn_mask'], get_attention_mask(tl, nbb))
    out = {('sd/' + k): v for k, v in np_sd(m).items()}
    for k, v in batch.items():
        out['in/' + k] = v.numpy()
    um = m.uniter_model
    kw = kwargs_of(batch)
    with torch.no_grad():
        txt = um._compute_txt_embeddings(batch['input_ids'], batch['position_ids'])
        img = um._compute_img_embeddings(batch['img_feat'], batch['img_pos_feat'])
        emb = um._compute_img_txt_embeddings(batch['input_ids'], batch['position_ids'],
                                             batch['img_feat'], batch['img_pos_feat'],
                                             batch['gather_index'])
        kw_all = dict(kw)
        kw_all['output_all_encoded_layers'] = True
        layers = um(**kw_all)
        pooled = um.pooler(layers[-1])
        logits = m(**kw)
        # text-only / image-only modes (model/model.py:348-355)
        txt_only = um(batch['input_ids'], batch['position_ids'], None, None,
                      torch.ones(B, T), output_all_encoded_layers=False)
        img_only = um(None, None, batch['img_feat'], batch['img_pos_feat'],
                      torch.ones(B, R), output_all_encoded_layers=False)
        # img_masks path (model/model.py:262-265)
        img_masks = torch.tensor([[0, 1, 0, 0, 1, 0]] * B)
        kw_m = dict(kw)
        kw_m['img_masks'] = img_masks
        masked = um(**kw_m)
    out.update({'out/txt_emb': txt.numpy(), 'out/img_emb': img.numpy(),
                'out/emb': emb.numpy(), 'out/pooled': pooled.numpy(),
                'out/logits': logits.numpy(), 'out/txt_only': txt_only.numpy(),
                'out/img_only': img_only.numpy(), 'in/img_masks': img_masks.numpy(),
                'out/masked': masked.numpy()})
    for i, l in enumerate(layers):
        out['out/layer%d' % i] = l.numpy()
    # loss + all grads (eval mode => dropout off, deterministic)
    crit = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([1.8]))
    m.zero_grad()
    loss = crit(m(**kw).squeeze(1), batch['labels'].float())
    loss.backward()
    out['out/loss'] = np.array(loss.item(), np.float64)
    for n, p in m.named_parameters():
        out['grad/' + n] = (p.grad if p.grad is not None
                            else torch.zeros_like(p)).numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'tiny_model.npz'), **out)
    print('tiny_model.npz: logits', logits.view(-1).tolist(), 'loss', loss.item())


def gen_shapes(cfg_path, fname, cases, with_grads=True):
    cfg = json.load(open(cfg_path))
    sd = O.synth_state_dict(cfg, seed=0, ln_jitter=0.02)
    m = build_ref(cfg, 2048, sd=sd)
    m.eval()
    out = {}
    crit = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([1.8]))
    for name, (B, T, R, tl, nbb, seed) in cases.items():
        batch = O.synth_batch(B, T, R, seed=seed, txt_lens=tl, num_bbs=nbb)
        kw = kwargs_of(batch)
        if with_grads:
            m.zero_grad()
            logits = m(**kw)
            loss = crit(logits.squeeze(1), batch['labels'].float())
            loss.backward()
            out[name + '/loss'] = np.array(loss.item(), np.float64)
            names, norms = [], []
            for n, p in m.named_parameters():
                names.append(n)
                norms.append(p.grad.double().norm().item() if p.grad is not None else 0.0)
            out[name + '/grad_norms'] = np.array(norms, np.float64)
            out['param_names'] = np.array(names)
            for n in ('linear.weight',
                      'uniter_model.encoder.layer.0.attention.self.query.weight',
                      'uniter_model.img_embeddings.img_linear.weight',
                      'uniter_model.encoder.layer.%d.output.dense.weight' % (cfg['num_hidden_layers'] - 1)):
                g = dict(m.named_parameters())[n].grad
                out[name + '/gslice/' + n] = g.reshape(-1)[:4096].numpy().copy()
            we = m.uniter_model.embeddings.word_embeddings.weight.grad
            rows = torch.unique(batch['input_ids'])[:8]
            out[name + '/word_rows'] = rows.numpy()
            out[name + '/word_grad_rows'] = we[rows].numpy().copy()
        else:
            with torch.no_grad():
                logits = m(**kw)
        out[name + '/logits'] = logits.detach().numpy().copy()
        out[name + '/shape'] = np.array([B, T, R, seed], np.int64)
        if tl is not None:
            out[name + '/txt_lens'] = np.array(tl, np.int64)
            out[name + '/num_bbs'] = np.array(nbb, np.int64)
        print(fname, name, 'logits[:4]', logits.view(-1)[:4].tolist())
    np.savez_compressed(os.path.join(HERE, fname), **out)


def gen_host_helpers():
    out = {}
    cases = {'a': ([64, 40, 10, 55], [36, 20, 36, 12], 64),
             'b': ([5, 1, 9], [10, 12, 3], 9),
             'c': ([128] * 2, [36] * 2, 128)}
    for k, (tl, nbb, T) in cases.items():
        am = get_attention_mask(tl, nbb)
        gi = get_gather_index(tl, nbb, len(tl), T, am.shape[1])
        out['gi/%s/tl' % k] = np.array(tl)
        out['gi/%s/nbb' % k] = np.array(nbb)
        out['gi/%s/T' % k] = np.array(T)
        out['gi/%s/gather_index' % k] = gi.numpy()
        out['gi/%s/attn_mask' % k] = am.numpy()
    # state_dict key inventory + param-group split of the reference optimizer factory
    cfg = json.load(open(os.path.join(REF, 'config/uniter-base.json')))
    tiny_like = dict(cfg)
    tiny_like.update(vocab_size=50, hidden_size=64, intermediate_size=128,
                     num_attention_heads=1, max_position_embeddings=16)
    m = build_ref(tiny_like, 32, seed=0)
    out['state_dict_keys'] = np.array(list(m.state_dict().keys()))
    out['state_dict_numel_base'] = np.array(
        sum(int(np.prod(s)) for _, s in O.param_shapes(cfg)), np.int64)
    opt = get_optimizer(m, dict(weight_decay=1e-3, optimizer='adam', beta1=0.9,
                                beta2=0.999, lr=1e-4))
    ids = {id(p): n for n, p in m.named_parameters()}
    out['decay_names'] = np.array([ids[id(p)] for p in opt.param_groups[0]['params']])
    out['no_decay_names'] = np.array([ids[id(p)] for p in opt.param_groups[1]['params']])
    # LR schedules
    for nm, fn, (w, t) in (('cos_500_3000', get_cosine_schedule_with_warmup, (500, 3000)),
                           ('cos_2_10', get_cosine_schedule_with_warmup, (2, 10)),
                           ('lin_50_400', get_linear_schedule_with_warmup, (50, 400))):
        p = torch.nn.Parameter(torch.zeros(1))
        o = torch.optim.Adam([p], lr=1.0)
        s = fn(o, num_warmup_steps=w, num_training_steps=t)
        lrs = [o.param_groups[0]['lr']]
        for _ in range(t + 5):
            o.step()
            s.step()
            lrs.append(o.param_groups[0]['lr'])
        out['lr/' + nm] = np.array(lrs, np.float64)
    # data/metrics.py: standard_metrics (binary) incl. optimal threshold, on seeded probabilities
    from data import metrics as RM
    rng = np.random.default_rng(0)
    cases = [(10, 0.4), (57, 0.3), (300, 0.5), (8, 0.9)]
    out['metrics/n'] = np.array(len(cases))
    for k, (n, pr) in enumerate(cases):
        p = rng.random(n).astype(np.float32)
        p[::7] = p[0]                                   # ties
        y = (rng.random(n) < pr).astype(np.int64)
        y[0], y[1] = 0, 1
        m = RM.standard_metrics(torch.from_numpy(p), torch.from_numpy(y), add_optimal_acc=True)
        out['metrics/%d/probs' % k] = p
        out['metrics/%d/labels' % k] = y
        out['metrics/%d/ref' % k] = np.array(json.dumps({kk: float(vv) for kk, vv in m.items()}))
    np.savez_compressed(os.path.join(HERE, 'host_helpers.npz'), **out)
    print('host_helpers.npz written;', len(out['state_dict_keys']), 'keys')


class _Loader(list):
    pass


def gen_trainer_steps():
    """Drive the REAL TrainerTemplate.calculate_loss (train_template.py:95-126)."""
    out = {}
    B, T, R = 3, 10, 6
    n_iters = 5
    batches = [O.synth_batch(B, T, R, seed=100 + i, vocab=TINY['vocab_size'],
                             img_dim=TINY_IMG_DIM, txt_lens=[10, 6, 8], num_bbs=[6, 4, 5])
               for i in range(n_iters)]
    for b in batches:   # make sure both classes appear
        b['labels'] = torch.tensor([1, 0, 1])
    for k, v in batches[0].items():
        pass
    for i, b in enumerate(batches):
        for k, v in b.items():
            out['batch%d/%s' % (i, k)] = v.numpy()
    for optname in ('adam', 'adamw'):
        m = build_ref(TINY, TINY_IMG_DIM, seed=7)
        jitter_(m, 11)
        m.eval()        # dropout off so the run is reproducible on another backend
        if optname == 'adam':
            for k, v in np_sd(m).items():
                out['sd0/' + k] = v
        cfg = dict(weight_decay=1e-3, optimizer=optname, beta1=0.9, beta2=0.999,
                   lr=1e-3, loss_func='bce_logits', gradient_accumulation=2,
                   max_grad_norm=1, pos_wt=1.8, parallel_computing=False)
        tr = object.__new__(train_template.TrainerTemplate)
        tr.config = cfg
        tr.device = torch.device('cpu')
        tr.model = m
        tr.optimizer = get_optimizer(m, cfg)
        tr.scheduler = get_cosine_schedule_with_warmup(tr.optimizer, num_warmup_steps=1,
                                                       num_training_steps=6)
        tr.criterion = torch.nn.BCEWithLogitsLoss(pos_weight=torch.tensor([cfg['pos_wt']]))
        tr.probs_list, tr.preds_list, tr.labels_list = [], [], []
        tr.loss_list, tr.short_loss_list = [], []
        for it, b in enumerate(batches):
            tr.iters = it
            preds = m(**kwargs_of(b))
            tr.calculate_loss(preds, b['labels'], grad_step=True)
            for k, v in np_sd(m).items():
                if k in ('linear.weight', 'linear.bias',
                         'uniter_model.encoder.layer.1.output.dense.weight',
                         'uniter_model.encoder.layer.0.attention.self.query.bias',
                         'uniter_model.img_embeddings.img_layer_norm.weight',
                         'uniter_model.embeddings.LayerNorm.weight',
                         'uniter_model.embeddings.word_embeddings.weight'):
                    out['%s/it%d/%s' % (optname, it, k)] = v
            out['%s/it%d/lr' % (optname, it)] = np.array(
                tr.optimizer.param_groups[0]['lr'], np.float64)
        out[optname + '/losses'] = np.array(tr.loss_list, np.float64)
        out[optname + '/probs'] = np.concatenate(tr.probs_list)
        if optname == 'adam':
            for k, v in np_sd(m).items():
                out['%s/final/%s' % (optname, k)] = v
        print('trainer', optname, 'losses', tr.loss_list)
    out['cfg'] = np.array(json.dumps(dict(lr=1e-3, weight_decay=1e-3, beta1=0.9, beta2=0.999,
                                          gradient_accumulation=2, max_grad_norm=1,
                                          pos_wt=1.8, warmup=1, total=6)))
    np.savez_compressed(os.path.join(HERE, 'trainer_steps.npz'), **out)


def gen_pretrain():
    """UniterForPretraining (model/pretrain.py) on the tiny config: mlm / mrfr / itm outputs,
    per-sample losses and selected gradients (incl. both tied weights)."""
    from model.pretrain import UniterForPretraining
    from oracle import pretrain_oracle as PO
    TINY_PRE = dict(TINY, vocab_size=100)      # the HIP MLM decoder needs vocab % 4 == 0 (28996 is)
    cfg = UniterConfig.from_dict(TINY_PRE)
    torch.manual_seed(21)
    m = UniterForPretraining(cfg, img_dim=TINY_IMG_DIM, img_label_dim=11)
    jitter_(m, 5)
    m.eval()
    out = {}
    for k, v in m.state_dict().items():
        out['sd/' + k] = v.detach().numpy().copy()
    B, T, R = 3, 10, 6
    tl, nbb = [10, 4, 7], [6, 6, 3]
    b = PO.synth_pretrain_batch(B, T, R, seed=9, vocab=TINY_PRE['vocab_size'], img_dim=TINY_IMG_DIM,
                                txt_lens=tl, num_bbs=nbb)
    b['label_targets'] = PO.synth_label_targets(int(b['img_masks'].sum()), 11, seed=9)
    for k, v in b.items():
        out['in/' + k] = v.numpy()
    keep = ['uniter.embeddings.word_embeddings.weight', 'uniter.img_embeddings.img_linear.weight',
            'uniter.img_embeddings.mask_embedding.weight', 'cls.predictions.bias',
            'cls.predictions.transform.dense.weight', 'cls.predictions.transform.LayerNorm.weight',
            'feat_regress.bias', 'feat_regress.net.0.weight', 'feat_regress.net.2.bias', 'itm_output.weight',
            'itm_output.bias', 'uniter.pooler.dense.weight', 'uniter.encoder.layer.0.attention.self.query.weight',
            'region_classifier.net.0.weight', 'region_classifier.net.2.weight', 'region_classifier.net.3.weight',
            'region_classifier.net.3.bias',
            'uniter.encoder.layer.1.output.LayerNorm.bias', 'uniter.embeddings.position_embeddings.weight']
    params = dict(m.named_parameters())
    for task in ('mlm', 'mrfr', 'itm', 'mrc', 'mrc-kl'):
        batch = dict(b)
        if task in ('mrfr', 'mrc', 'mrc-kl'):
            batch['img_feat'] = b['img_feat_masked']
        scores = m(batch, task, compute_loss=False)
        m.zero_grad()
        loss = m(batch, task, compute_loss=True)
        loss.mean().backward()
        out['%s/scores' % task] = scores.detach().numpy().copy()
        out['%s/loss' % task] = loss.detach().numpy().copy()
        for n in keep:
            g = params[n].grad
            out['%s/grad/%s' % (task, n)] = (g if g is not None else torch.zeros_like(params[n])).numpy().copy()
        print('pretrain', task, 'loss mean', loss.mean().item(), tuple(scores.shape))
    out['state_dict_keys'] = np.array(list(m.state_dict().keys()))
    np.savez_compressed(os.path.join(HERE, 'pretrain_tiny.npz'), **out)


def gen_crossval_ensemble():
    """utils/crossval.py generate_crossval_splits and utils/ensemble.py (create_ensemble_prediction, find_ensemble
    without deap) run on small synthetic inputs; inputs and the files they wrote are stored as arrays.
    seaborn (imported by data/metrics.py for a plot helper) is not installed: an empty stand-in module is registered,
    in this script only."""
    import csv, glob, json, tempfile, types
    sys.modules.setdefault('seaborn', types.ModuleType('seaborn'))
    from utils.crossval import generate_crossval_splits
    from utils import ensemble as E
    rng = np.random.Generator(np.random.PCG64(77))
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        n_train, n_dev = 97, 20
        ids = rng.permutation(90000)[:n_train + n_dev] + 1000
        labels = (rng.random(n_train + n_dev) < 0.4).astype(np.int64)
        for name, sl in (('train', slice(0, n_train)), ('dev_seen', slice(n_train, None))):
            with open(os.path.join(tmp, name + '.jsonl'), 'w') as f:
                f.write('\n'.join(json.dumps({'id': int(i), 'img': 'img/%05d.png' % i, 'label': int(l), 'text': 't%d' % i})
                                  for i, l in zip(ids[sl], labels[sl])))
        out['cv/ids'], out['cv/labels'], out['cv/n_train'] = ids, labels, np.int64(n_train)
        generate_crossval_splits(tmp, dev_size=16)
        files = sorted(glob.glob(os.path.join(tmp, 'crossval_16', '*.jsonl')))
        out['cv/files'] = np.array([os.path.basename(f) for f in files])
        for f in files:
            with open(f) as fh:
                out['cv/out/' + os.path.basename(f)] = np.array([json.loads(l)['id'] for l in fh.read().split('\n')])
        # ---- use_dev_set variant: dev_seen with text confounders, rotated through the folds ----
        tmp2 = os.path.join(tmp, 'usedev')
        os.makedirs(tmp2)
        n_dev2 = 26
        rng2 = np.random.Generator(np.random.PCG64(78))        # its own stream: the inputs of the other parts stay as they were
        ids2 = rng2.permutation(90000)[:n_train + n_dev2] + 100000
        labels2 = (rng2.random(n_train + n_dev2) < 0.45).astype(np.int64)
        texts2 = ['t%d' % i for i in ids2]
        for a, b in ((n_train + 1, n_train + 7), (n_train + 3, n_train + 12), (n_train + 4, n_train + 20)):
            texts2[b] = texts2[a]                      # three confounder groups inside dev_seen
        texts2[n_train + 21] = texts2[n_train + 3]     # ... one of them with three members
        for name, sl in (('train', slice(0, n_train)), ('dev_seen', slice(n_train, None))):
            with open(os.path.join(tmp2, name + '.jsonl'), 'w') as f:
                f.write('\n'.join(json.dumps({'id': int(i), 'img': 'img/%05d.png' % i, 'label': int(l), 'text': t})
                                  for i, l, t in zip(ids2[sl], labels2[sl], texts2[sl])))
        out['cvd/ids'], out['cvd/labels'], out['cvd/texts'] = ids2, labels2, np.array(texts2)
        # two folds: with more, the reference's float32 choice probabilities (e.g. 2/3 + 1/3) are rejected by numpy
        generate_crossval_splits(tmp2, dev_size=20, use_dev_set=True)
        files = sorted(glob.glob(os.path.join(tmp2, 'crossval_20_usedevtest', '*.jsonl')))
        out['cvd/files'] = np.array([os.path.basename(f) for f in files])
        for f in files:
            with open(f) as fh:
                out['cvd/out/' + os.path.basename(f)] = np.array([json.loads(l)['id'] for l in fh.read().split('\n')])
        print('use_dev_set files:', len(files))
        # ---- ensemble ----
        n = 60
        eid = rng.permutation(5000)[:n] + 1
        gt = (rng.random(n) < 0.45).astype(np.int64)
        out['ens/id'], out['ens/gt'] = eid, gt
        tid = rng.permutation(5000)[:40] + 6000
        tgt = (rng.random(40) < 0.45).astype(np.int64)
        out['ens/tid'], out['ens/tgt'] = tid, tgt
        dev_files, test_files = [], []
        for k in range(3):
            keep = np.sort(rng.permutation(n)[:50])            # every fold misses 10 of the dev samples
            proba = np.clip(0.5 + (gt[keep] - 0.5) * rng.uniform(0.1, 0.6) + rng.normal(0, 0.25, keep.size), 0.001, 0.999)
            proba = np.round(proba, 6)
            out['ens/fold%d/keep' % k], out['ens/fold%d/proba' % k] = keep, proba
            fn = os.path.join(tmp, 'uniter_fold_%d_dev_seen_preds.csv' % k)
            with open(fn, 'w') as f:
                f.write('id,proba,label,gt\n' + ''.join('%i,%f,%i,%i\n' % (eid[j], p, p > 0.5, gt[j]) for j, p in zip(keep, proba)))
            dev_files.append(fn)
            tp = np.round(np.clip(0.5 + (tgt - 0.5) * rng.uniform(0.1, 0.6) + rng.normal(0, 0.25, 40), 0.001, 0.999), 6)
            out['ens/fold%d/tproba' % k] = tp
            fn = os.path.join(tmp, 'uniter_fold_%d_test_seen_preds.csv' % k)
            with open(fn, 'w') as f:
                f.write('id,proba,label,gt\n' + ''.join('%i,%f,%i,%i\n' % (tid[j], tp[j], tp[j] > 0.5, tgt[j]) for j in range(40)))
            test_files.append(fn)
        preds = np.stack([np.where(np.isin(np.arange(n), out['ens/fold%d/keep' % k]), 0.3 + 0.1 * k, -1.0) for k in range(3)])
        for ol in (False, True):
            out['ens/direct/%d' % ol] = E.create_ensemble_prediction(preds.copy(), [0.5, 1.0, 2.0], on_logits=ol)
        E.find_ensemble(dev_files=dev_files, test_files=test_files, weight_range=(0.0, 0.5, 1.0, 2.0), max_weights=10000)
        for name in ('uniter_dev_seen_ensemble.csv', 'uniter_test_seen_ensemble.csv'):
            with open(os.path.join(tmp, name)) as f:
                rows = list(csv.reader(f))
            out['ens/out/%s/header' % name] = np.array(rows[0])
            out['ens/out/%s/body' % name] = np.array([[float(v) for v in r] for r in rows[1:]])
            print(name, rows[0], len(rows) - 1)
    np.savez_compressed(os.path.join(HERE, 'crossval_ensemble.npz'), **out)


def simple_tokenizer(texts, max_length=12):
    """Offline stand-in for the BertTokenizer partial of train_uniter.py:124-126 (same return fields); the test
    applies the SAME function (tests/common.py) on the other side."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE)))
    from common import simple_tokenizer as tok
    return tok(texts, max_length=max_length)


def gen_data_pipeline():
    """data/meme_dataset.py (MemeDataset.__getitem__, collate_fn, ConfounderSampler) and
    data/dataset_template.py:_load_img_feature run on a small synthetic dataset in the reference's on-disk format.
    The fixture holds the dataset's RAW content (so the test can rebuild the files) and what the reference made of it."""
    import random
    import tempfile
    from data.meme_dataset import MemeDataset, ConfounderSampler
    import torch.utils.data as tdata
    tdata.Sampler.__init__ = lambda self, *a, **k: None     # torch 1.6's Sampler took the data source (meme_dataset.py:224)
    rng = np.random.Generator(np.random.PCG64(2024))
    n, dim = 14, 8
    texts = ['look at this cat', 'when you see it', 'nobody : me', 'look at this cat', 'sky tree car', 'when you see it',
             'a b c d e f g h i j k l m n', 'x', 'funny dog meme', 'nobody : me', 'tree', 'car car car', 'you and me', 'sky']
    labels = [0, 1, 0, 1, 0, 0, 1, 0, 1, 1, 0, 1, 0, 0]        # 'look at this cat' and 'nobody : me' are confounders
    out = {'raw/texts': np.array(texts), 'raw/labels': np.array(labels, np.int64), 'raw/ids': np.arange(100, 100 + n)}
    with tempfile.TemporaryDirectory() as tmp:
        fdir = os.path.join(tmp, 'img_feats')
        os.makedirs(fdir)
        with open(os.path.join(tmp, 'train.jsonl'), 'w') as f:
            for i in range(n):
                sid = str(100 + i).zfill(5)
                nbb = int(rng.integers(2, 7))
                feat = rng.standard_normal((nbb, dim)).astype(np.float32)
                W, H = int(rng.integers(200, 800)), int(rng.integers(200, 800))
                x1 = rng.random((nbb, 1)) * 0.6 * W; y1 = rng.random((nbb, 1)) * 0.6 * H
                bw = (rng.random((nbb, 1)) * 0.3 + 0.05) * W; bh = (rng.random((nbb, 1)) * 0.3 + 0.05) * H
                bbox = np.concatenate([x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
                info = {'bbox': bbox.copy(), 'image_width': W, 'image_height': H, 'objects': rng.integers(0, 1600, nbb)}
                if i % 2 == 0:
                    info['objects_conf'] = rng.random(nbb).astype(np.float32)
                else:                       # the other detector export: class probabilities (dataset_template.py:103-106)
                    info['cls_prob'] = rng.random((nbb, 5)).astype(np.float32)
                np.save(os.path.join(fdir, sid + '.npy'), feat)
                np.save(os.path.join(fdir, sid + '_info.npy'), info, allow_pickle=True)
                f.write(json.dumps({'id': 100 + i, 'img': 'img/%s.png' % sid, 'label': labels[i], 'text': texts[i]}) + '\n')
                out['raw/%d/feat' % i] = feat
                out['raw/%d/bbox' % i] = bbox
                out['raw/%d/wh' % i] = np.array([W, H], np.int64)
                out['raw/%d/objects' % i] = info['objects']
                out['raw/%d/%s' % (i, 'objects_conf' if i % 2 == 0 else 'cls_prob')] = info.get('objects_conf', info.get('cls_prob'))
        for thr in (0.0, 0.45):
            ds = MemeDataset(filepath=os.path.join(tmp, 'train.jsonl'), feature_dir=fdir, preload_images=False, debug=False,
                             text_padding=simple_tokenizer, return_ids=True, confidence_threshold=thr)
            tag = 'thr%g' % thr
            for i in range(n):
                it = ds[i]
                out['%s/item/%d/img_feat' % (tag, i)] = it['img_feat'].numpy()
                out['%s/item/%d/img_pos_feat' % (tag, i)] = it['img_pos_feat'].numpy()
                out['%s/item/%d/label_id' % (tag, i)] = np.array([int(it['label']), int(it['data_id'])])
            collate = ds.get_collate_fn()
            for bi, idxs in enumerate(([0, 1, 2, 3], [6, 7, 10], [13, 12, 11, 9, 8, 5])):
                b = collate([ds[i] for i in idxs])
                out['%s/batch/%d/idxs' % (tag, bi)] = np.array(idxs)
                for k, v in b.items():
                    if v is not None:
                        out['%s/batch/%d/%s' % (tag, bi, k)] = v.numpy()
        ds = MemeDataset(filepath=os.path.join(tmp, 'train.jsonl'), feature_dir=fdir, preload_images=False,
                         text_padding=simple_tokenizer)
        for rep in (1, 2, 3):
            random.seed(1000 + rep)
            sm = ConfounderSampler(ds, repeat_factor=rep)
            out['sampler/%d/confounders' % rep] = np.array(sm.confounders)
            out['sampler/%d/len' % rep] = np.array(len(sm))
            out['sampler/%d/epoch0' % rep] = np.array(list(iter(sm)))
            out['sampler/%d/epoch1' % rep] = np.array(list(iter(sm)))
    np.savez_compressed(os.path.join(HERE, 'data_pipeline.npz'), **out)
    print('data_pipeline.npz:', len(out), 'arrays; batch0 attn_mask', out['thr0/batch/0/attn_mask'].sum(1))


if __name__ == '__main__':
    which = sys.argv[1:] or ['tiny', 'host', 'trainer', 'base', 'large', 'pretrain', 'crossval', 'data']
    if 'data' in which:
        gen_data_pipeline()
    if 'crossval' in which:
        gen_crossval_ensemble()
    if 'pretrain' in which:
        gen_pretrain()
    if 'tiny' in which:
        gen_tiny()
    if 'host' in which:
        gen_host_helpers()
    if 'trainer' in which:
        gen_trainer_steps()
    if 'base' in which:
        gen_shapes(os.path.join(REF, 'config/uniter-base.json'), 'shapes_base.npz', {
            'cfg1_full': (4, 64, 36, None, None, 1234),
            'cfg1_ragged': (4, 64, 36, [64, 40, 10, 55], [36, 20, 36, 12], 1235),
            'cfg2_full': (16, 128, 36, None, None, 1234),
        })
    if 'large' in which:
        gen_shapes(os.path.join(REF, 'config/uniter-large.json'), 'shapes_large.npz', {
            'cfg4_full': (8, 128, 50, None, None, 1234),
        })
