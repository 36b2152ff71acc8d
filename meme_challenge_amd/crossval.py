"""Cross-validation driver (host side; mirrors utils/crossval.py of the reference).

`generate_crossval_splits(data_path, dev_size)` writes `crossval_<dev_size>/{train,dev}_<kk>.jsonl`: the training
list is shuffled once with `random.seed(42)`, split by label, and fold k takes `dev_size // 2` samples of each
label as its validation set (utils/crossval.py:24-47,112-123) -- the same files the reference writes for the same
input.  `train_crossval` runs one trainer per fold (seed + fold, `<name>_fold_<k>.<ext>` checkpoints), averages the
folds' validation metrics and hands the per-fold prediction files to `ensemble.find_ensemble` (:132-215).
The `use_dev_set` variant (:49-110: half of dev_seen joins each fold's training set, the other half is that fold's test
set `dev_seen_<kk>.jsonl`, with balanced re-use and confounder groups kept together) writes `crossval_<n>_usedevtest/`;
both variants are pinned on files the reference itself wrote (tests/golden/crossval_ensemble.npz).
"""
import json
import logging
import math
import os
import random
from glob import glob
from statistics import mean

import numpy as np

from .ensemble import find_ensemble
from .utils import set_seed

logger = logging.getLogger('CrossValLog')


def export_jsonl(filepath, dict_list):
    with open(filepath, 'w') as f:
        f.write('\n'.join(json.dumps(d) for d in dict_list))


def _read_jsonl(path):
    with open(path) as f:
        return [json.loads(line) for line in f.readlines()]


def crossval_dir(data_path, dev_size, use_dev_set=False):
    return os.path.join(data_path, 'crossval_%i%s' % (dev_size, '_usedevtest' if use_dev_set else ''))


def _dev_seen_rotation(dev_list, num_splits):
    """`use_dev_set` (utils/crossval.py:49-110): every fold tests on HALF of dev_seen and trains on the other half, such
    that over the folds each dev_seen sample is tested about equally often and text confounders (same text, several
    memes) stay together.  Returns (test index lists, train index lists), one per fold.

    The draws come from numpy's GLOBAL generator (seeded with 42 by the caller) in the reference's order -- one
    two-way choice per confounder group that is not forced, one shuffle when more samples are due than fit, one
    weighted choice without replacement for the rest -- because the files written from them are the interface."""
    n = len(dev_list)
    half = n // 2
    due = np.zeros(n, dtype=np.float32) + int(math.ceil(num_splits / 2.0))     # test appearances each sample still owes
    by_text = {}
    for idx, item in enumerate(dev_list):
        by_text.setdefault(item['text'], []).append(idx)
    groups = [np.array(v, dtype=np.int32) for v in by_text.values() if len(v) > 1]
    grouped = np.array([i for g in groups for i in g], dtype=np.int32)
    logger.info('Number of confounders: %i (sum: %i)' % (len(groups), grouped.shape[0]))
    tests = []
    for k in range(num_splits):
        left = num_splits - k
        weights = np.copy(due)                       # snapshot BEFORE this fold's confounder decisions
        taken_groups = np.array([], dtype=np.int32)
        for g in groups:
            owed = float(due[g[0]])                  # a group is booked on its first member
            # (float64 probabilities: the reference forms them in float32 and numpy rejects e.g. 2/3 + 1/3 as "do not
            # sum to 1", so its own call only survives fold counts with exact fractions -- where both agree)
            if owed >= left or np.random.choice(2, size=1, p=[(left - owed) / left, owed / left]) == 1:
                taken_groups = np.concatenate([taken_groups, g])
                due[g[0]] -= 1
        weights[grouped] = 0
        forced = np.where(weights >= left)[0]        # must be tested in every remaining fold to catch up
        room = half - taken_groups.shape[0]
        if forced.shape[0] > room:
            np.random.shuffle(forced)
            forced = forced[np.argsort(due[forced][::-1])][:room]      # (the reference's ordering, reversed keys and all)
        room -= forced.shape[0]
        weights[forced] = 0
        if weights.sum() == 0:
            drawn = np.zeros((0,))
        else:
            drawn = np.random.choice(n, size=room, replace=False, p=weights / weights.sum())
            due[drawn] = due[drawn] - 1
        due[forced] = due[forced] - 1
        tests.append(drawn.tolist() + np.arange(n)[forced].tolist() + taken_groups.tolist())
    trains = [[i for i in range(n) if i not in t] for t in tests]
    return tests, trains


def generate_crossval_splits(data_path, dev_size=300, use_dev_set=False):
    random.seed(42)
    np.random.seed(42)
    data_list, dev_list = [], []
    for name in ('train.jsonl', 'dev_seen.jsonl'):
        path = os.path.join(data_path, name)
        assert os.path.isfile(path), 'Tried to create cross validation splits, but file could not be found at %s' % path
        items = _read_jsonl(path)
        if name == 'dev_seen.jsonl' and use_dev_set:
            dev_list = items            # kept in file order, rotated through the folds below
            continue
        random.shuffle(items)           # one shuffle per file, in this order (the RNG stream is part of the format)
        data_list += items
    by_label = {l: [d for d in data_list if d['label'] == l] for l in (0, 1)}
    num_splits = min(len(v) for v in by_label.values()) // dev_size
    dev_tests = dev_trains = None
    if use_dev_set:
        dev_tests, dev_trains = _dev_seen_rotation(dev_list, num_splits)
        logger.info('Test set lengths: %s' % str([len(t) for t in dev_tests]))
    out_dir = crossval_dir(data_path, dev_size, use_dev_set)
    os.makedirs(out_dir, exist_ok=True)
    half = dev_size // 2
    for k in range(num_splits):
        lo, hi = k * half, (k + 1) * half
        dev_set = by_label[0][lo:hi] + by_label[1][lo:hi]
        train_set = by_label[0][:lo] + by_label[0][hi:] + by_label[1][:lo] + by_label[1][hi:]
        tag = str(k).zfill(2)
        if use_dev_set:
            train_set = train_set + [dev_list[i] for i in dev_trains[k]]
            export_jsonl(os.path.join(out_dir, 'dev_seen_%s.jsonl' % tag), [dev_list[int(i)] for i in dev_tests[k]])
        export_jsonl(os.path.join(out_dir, 'train_%s.jsonl' % tag), train_set)
        export_jsonl(os.path.join(out_dir, 'dev_%s.jsonl' % tag), dev_set)
        logger.info('Exported split %i with %4.2f%% hateful memes in validation set.'
                    % (k, 100.0 * sum(d['label'] for d in dev_set) / max(len(dev_set), 1)))
    return num_splits


def _dataset_name(loader):
    ds = getattr(loader, 'dataset', None)
    if ds is None and hasattr(loader, 'loader'):
        ds = getattr(loader.loader, 'dataset', None)
    return getattr(ds, 'name', '')


def train_crossval(trainer_class, config, data_loader_funcs, num_folds=0, dev_size=300, use_dev_set=False):
    """num_folds = 0: one ordinary run on train.jsonl / dev_seen.jsonl.  Otherwise: the first num_folds folds
    (-1 = all).  Returns the list of per-fold validation metrics (the single run's metrics for num_folds = 0)."""
    if num_folds == 0:
        config['train_loader'] = data_loader_funcs['train'](os.path.join(config['data_path'], 'train.jsonl'))
        config['val_loader'] = data_loader_funcs['val'](os.path.join(config['data_path'], 'dev_seen.jsonl'))
        return trainer_class(config).train_main()
    cv = crossval_dir(config['data_path'], dev_size, use_dev_set)
    import torch.distributed as dist
    ddp = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not ddp or dist.get_rank() == 0:          # one writer: the files are opened with 'w' (truncated) while they are written
        if not os.path.isdir(cv) or not glob(os.path.join(cv, '*.jsonl')):
            logger.info('Creating cross-validation splits for dev size %i' % dev_size)
            generate_crossval_splits(config['data_path'], dev_size=dev_size, use_dev_set=use_dev_set)
    if ddp:                                      # nobody globs or opens a fold file before rank 0 has closed them all
        dist.barrier()
    train_sets = sorted(glob(os.path.join(cv, 'train_??.jsonl')))
    dev_sets = sorted(glob(os.path.join(cv, 'dev_??.jsonl')))
    test_sets = sorted(glob(os.path.join(cv, 'dev_seen_??.jsonl')))        # use_dev_set: the fold's half of dev_seen
    assert len(train_sets) == len(dev_sets), 'Found an inequal number of training and validation sets'
    folds = len(dev_sets) if num_folds == -1 else min(num_folds, len(dev_sets))
    if use_dev_set:
        assert len(test_sets) >= folds, 'Could not find enough test sets.'
    base, ext = config['model_save_name'].rsplit('.', 1)
    fixed_tests = list(config.get('test_loader', []))
    if use_dev_set:             # the whole dev_seen is no test set any more: parts of it are trained on
        fixed_tests = [t for t in fixed_tests if _dataset_name(t) != 'dev_seen']
    val_metrics = []
    for k in range(folds):
        set_seed(config['seed'] + k)
        logger.info('Starting fold %i of %i' % (k, folds))
        config['train_loader'] = data_loader_funcs['train'](train_sets[k])
        config['val_loader'] = data_loader_funcs['val'](dev_sets[k])
        config['test_loader'] = fixed_tests + ([data_loader_funcs['test'](test_sets[k])] if use_dev_set else [])
        config['model_save_name'] = '%s_fold_%i.%s' % (base, k, ext)
        fold_metrics, _ = trainer_class(config).train_main()
        val_metrics.append(fold_metrics)
    config['model_save_name'] = base + '.' + ext
    if not val_metrics:
        return val_metrics
    means = {key: mean(v[key] for v in val_metrics) for key in val_metrics[0]}
    logger.info('Cross validation finished. Mean scores of validation folds:\n' + '\n'.join(
        '%s: %s' % (k, ('%5.4f' % v) if k == 'loss' else ('%4.2f%%' % (100.0 * v))) for k, v in means.items()))
    if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
        return val_metrics                 # the prediction files are rank 0's (written before end_training's barrier)
    names = [_dataset_name(t) for t in config.get('test_loader', [])]
    dev_names = sorted(n for n in names if n.startswith('dev'))
    if not dev_names:
        logger.warning('Skipping ensemble calculation as no predictions for a validation set could be found')
        return val_metrics
    pattern = os.path.join(config['model_path'], base + '_fold_*')
    if use_dev_set:             # every fold predicted its own half of dev_seen: the ensemble weights are fitted on all of them
        dev_files = sorted(glob(pattern + '_dev_seen_??_preds.csv'))
        test_files = [sorted(glob(pattern + '_%s_preds.csv' % _dataset_name(t))) for t in fixed_tests]
    else:
        dev_files = sorted(glob(pattern + '_%s_preds.csv' % dev_names[0]))
        test_files = [sorted(glob(pattern + '_%s_preds.csv' % n)) for n in names if n != dev_names[0]]
    if dev_files:
        find_ensemble(dev_files=dev_files, test_files=[t for t in test_files if t])
    return val_metrics
