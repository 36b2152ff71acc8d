"""Ensembling of per-fold prediction files (host side; mirrors utils/ensemble.py of the reference).

Prediction files are the trainer's CSV exports (`id,proba,label[,gt]`, train_template.py:155-186).
`find_ensemble(dev_files, test_files)` aligns the files of the folds on their ids, searches per-model
weights -- on probabilities and on logits -- that maximise AUROC on the dev predictions
(utils/ensemble.py:35-112, brute force :180-203), picks the accuracy-optimal threshold and writes
`<model>_<set>_ensemble.csv` next to the inputs.  The reference adds an evolutionary search when the
`deap` package is installed (:63-71, :206-284); like the reference without that package, this
module stops at the brute-force result.
"""
import csv
import logging
import os
import random
from itertools import product

import numpy as np

from .metrics import aucroc, find_optimal_threshold

logger = logging.getLogger('EnsembleLogger')


def load_csv(csv_file):
    """Columns of a prediction file as numpy arrays: `proba` float, everything else int (utils/ensemble.py:115-127)."""
    with open(csv_file, 'r', newline='') as f:
        rows = list(csv.reader(f, delimiter=','))
    header, body = rows[0], rows[1:]
    return {col: np.array([float(r[i]) if col == 'proba' else int(r[i]) for r in body]) for i, col in enumerate(header)}


def align_ids(csv_dicts):
    """Bring several prediction files onto the sorted union of their ids; a file that lacks an id gets
    proba = label = -1 there (utils/ensemble.py:130-141).  The ground truth must agree between files."""
    all_ids = np.array(sorted({int(e) for d in csv_dicts for e in d['id'].tolist()}))
    gt = np.full(all_ids.shape, -1, dtype=np.int64)
    out = []
    for d in csv_dicts:
        pos = np.searchsorted(all_ids, d['id'])
        have = gt[pos] >= 0
        if np.any(gt[pos][have] != d['gt'][have]):
            raise AssertionError('Label mismatch in the predictions. Something must be wrong with the predictions.')
        gt[pos] = d['gt']
        proba = np.full(all_ids.shape, -1.0)
        label = np.full(all_ids.shape, -1, dtype=np.int64)
        proba[pos], label[pos] = d['proba'], d['label']
        out.append({'orig': d, 'id': all_ids, 'proba': proba, 'label': label})
    for d in out:
        d['gt'] = gt
    return out


def export_csv(csv_dict, csv_file):
    """Write the columns back (floats as %f, ints as %i; utils/ensemble.py:144-154)."""
    cols = [k for k in csv_dict if k != 'orig']
    n = len(csv_dict[cols[0]])
    with open(csv_file, 'w') as f:
        f.write(','.join(cols) + '\n')
        for i in range(n):
            f.write(','.join(('%f' % csv_dict[k][i]) if isinstance(csv_dict[k][i], (float, np.floating))
                             else ('%i' % csv_dict[k][i]) for k in cols) + '\n')


def create_ensemble_prediction(predictions, weights, on_logits=False):
    """Weighted mean of the models' probabilities (or of their logits, mapped back through the sigmoid).
    Entries equal to -1 are missing: they get no weight, and a sample nobody predicted gets 0.5
    (utils/ensemble.py:157-177).  The input is not modified."""
    p = np.array(np.stack(predictions, axis=0) if isinstance(predictions, (list, tuple)) else predictions, dtype=np.float64)
    w = np.asarray(weights, dtype=np.float64)
    missing = p == -1
    p[missing] = 0.5
    present = 1 - missing
    if on_logits:
        p = np.log(np.clip(p, 1e-8, 1.0)) - np.log(np.clip(1 - p, 1e-8, 1.0))
    wsum = (w[:, None] * present).sum(axis=0)
    out = (w[:, None] * p * present).sum(axis=0) / np.clip(wsum, 1e-4, 1e5)
    out[wsum == 0.0] = 0.5
    if on_logits:
        out = 1.0 / (1.0 + np.exp(-out))
    return out


def brute_force_finder(eval_func, num_weights, weight_range, max_weights=1e5):
    """Best (weights, on_logits) over the grid weight_range^num_weights, capped at max_weights tuples (a seeded
    shuffle of the grid, or seeded random draws when the grid itself is too large; utils/ensemble.py:180-203).
    The first maximum wins."""
    max_weights = int(max_weights)
    if np.log(len(weight_range)) * num_weights < np.log(2e7):
        tuples = list(product(weight_range, repeat=num_weights))
        if len(tuples) > max_weights:
            random.seed(42)
            random.shuffle(tuples)
            tuples = tuples[:max_weights]
    else:
        np.random.seed(42)
        idx = np.random.randint(0, len(weight_range), size=(max_weights, num_weights))
        tuples = [[weight_range[idx[m, n]] for n in range(num_weights)] for m in range(max_weights)]
    best_score, best_config = -1, None
    for weights in tuples:
        for on_logits in (True, False):
            score, = eval_func(weights, on_logits=on_logits)
            if score > best_score:
                best_score, best_config = score, {'weights': weights, 'on_logits': on_logits}
    return best_score, best_config


def _names(dev_file):
    """(model name, dev-set name) out of `<model>_fold_<k>_<set>_preds.csv` (utils/ensemble.py:42-47)."""
    base = dev_file.split('/')[-1]
    if dev_file.endswith('_00_preds.csv'):          # per-fold test files `..._dev_seen_00_preds.csv`
        return base.rsplit('_', 6)[0], '_'.join(dev_file.rsplit('_', 4)[-4:-1])
    return base.rsplit('_', 5)[0], '_'.join(dev_file.rsplit('_', 3)[-3:-1])


def find_ensemble(dev_files, test_files, weight_range=(0.0, 0.5, 1.0, 2.0), max_weights=10000):
    """utils/ensemble.py:35-112.  Returns {'score', 'config', 'threshold', 'files'} (the reference returns None)."""
    dev_preds = align_ids([load_csv(f) for f in dev_files])
    dev_gt = dev_preds[0]['gt']
    dev_scores = [aucroc(d['orig']['proba'], d['orig']['gt']) for d in dev_preds]
    logger.info('Individual scores: ' + ', '.join('%4.2f%%' % (100.0 * s) for s in dev_scores))
    output_dir = os.path.dirname(dev_files[0]) or '.'
    model_name, dev_name = _names(dev_files[0])
    predictions = np.stack([d['proba'] for d in dev_preds], axis=0)

    def eval_func(weights, on_logits=True):
        score = float(aucroc(create_ensemble_prediction(predictions, weights, on_logits), dev_gt))
        # The reference's create_ensemble_prediction fills the missing entries of ITS INPUT with 0.5 in place
        # (utils/ensemble.py:163-164), so from the second evaluation on a fold that did not predict a sample votes
        # 0.5 for it with full weight.  Kept, so that the search picks the same weights as the reference does.
        predictions[predictions == -1] = 0.5
        return score,

    best_score, best = brute_force_finder(eval_func, len(dev_preds), weight_range, max_weights)
    proba = create_ensemble_prediction(predictions, best['weights'], best['on_logits'])
    threshold = find_optimal_threshold(proba, dev_gt)
    written = []
    # column order of the reference's export: id, gt, proba, label
    out = {'id': dev_preds[0]['id'], 'gt': dev_gt, 'proba': proba, 'label': (proba > threshold).astype(np.int32)}
    written.append(os.path.join(output_dir, model_name + '_' + dev_name + '_ensemble.csv'))
    export_csv(out, written[-1])
    logger.info('Best score on %s: %4.2f%% (accuracy=%4.2f%%)' % (dev_name, best_score * 100.0,
                                                                 100.0 * (out['label'] == dev_gt).mean()))
    if test_files and not isinstance(test_files[0], (list, tuple)):
        test_files = [test_files]
    for test_list in test_files or []:
        if not test_list:
            continue
        test_name = '_'.join(test_list[0].rsplit('_', 3)[-3:-1])
        test_model = test_list[0].split('/')[-1].rsplit('_', 5)[0]
        preds = [load_csv(f) for f in test_list]
        p = create_ensemble_prediction([d['proba'] for d in preds], best['weights'], best['on_logits'])
        d = {k: v for k, v in preds[0].items()}
        d['proba'], d['label'] = p, (p > threshold).astype(np.int32)
        if 'gt' in d:
            logger.info('New ensemble score on %s: %4.2f%%' % (test_name, 100.0 * aucroc(p, d['gt'])))
        written.append(os.path.join(output_dir, test_model + '_' + test_name + '_ensemble.csv'))
        export_csv(d, written[-1])
    return {'score': best_score, 'config': best, 'threshold': threshold, 'files': written}
