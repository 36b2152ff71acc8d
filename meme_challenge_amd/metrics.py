"""Binary classification metrics used by the trainer (counterpart of data/metrics.py:16-167:
accuracy / recall / precision / F1 at a threshold, AUROC, optimal-accuracy threshold).

Host-side post-processing of probabilities once per epoch; the threshold search is vectorised
(one sort + cumulative sums) instead of re-evaluating every candidate threshold."""
import logging

import numpy as np
import torch

LOGGER = logging.getLogger('MetricLogger')


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def aucroc(probs, labels):
    """ROC AUC (the challenge metric).  0.0 when only one class is present (data/metrics.py:151-167)."""
    p, y = _np(probs).astype(np.float64), _np(labels).astype(np.int64)
    assert np.all((p <= 1.0) & (p >= 0.0)), "Probabilities must be between 0 and 1"
    assert np.all((y == 0) | (y == 1)), "Labels must be binary (0 or 1)"
    n1 = int(y.sum())
    n0 = y.size - n1
    if n0 == 0 or n1 == 0:
        LOGGER.warning("ROC AUC calculation got only one label. Score not defined here, setting it to 0.")
        return 0.0
    # Mann-Whitney U with average ranks for ties == sklearn.metrics.roc_auc_score
    order = np.argsort(p, kind='mergesort')
    ps = p[order]
    ranks = np.empty(p.size, dtype=np.float64)
    i = 0
    while i < p.size:
        j = i
        while j + 1 < p.size and ps[j + 1] == ps[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y == 1].sum() - n1 * (n1 + 1) / 2.0) / (n0 * n1))


def standard_metrics_binary(probs, labels, threshold=0.5, add_aucroc=True, add_optimal_acc=False, **kwargs):
    p, y = _np(probs).astype(np.float64), _np(labels).astype(np.int64)
    assert np.all((p <= 1.0) & (p >= 0.0)), "Probabilities must be between 0 and 1, but are as follows: " + str(p)
    assert np.all((y == 0) | (y == 1)), "Labels must be binary (0 or 1), but are as follows: " + str(y)
    pred = (p > threshold).astype(np.int64)
    tp = float(((pred == 1) & (y == 1)).sum())
    tn = float(((pred == 0) & (y == 0)).sum())
    fp = float(((pred == 1) & (y == 0)).sum())
    fn = float(((pred == 0) & (y == 1)).sum())
    m = {'accuracy': (tp + tn) / max(p.size, 1),
         'recall': tp / max(tp + fn, 1e-4),
         'precision': tp / max(tp + fp, 1e-4)}
    m['F1'] = 0.0 if m['recall'] == 0.0 or m['precision'] == 0.0 else \
        2 * m['precision'] * m['recall'] / (m['precision'] + m['recall'])
    if add_aucroc:
        m['aucroc'] = aucroc(p, y)
    if add_optimal_acc:
        t = find_optimal_threshold(p, y, metric='accuracy')
        m['optimal_threshold'] = t
        m['optimal_accuracy'] = standard_metrics_binary(p, y, threshold=t, add_aucroc=False)['accuracy']
    return m


standard_metrics = standard_metrics_binary


def find_optimal_threshold(probs, labels, metric='accuracy', show_plot=False):
    """Threshold maximising `metric` over the candidates {0, sorted probs, 1}; the midpoint to the
    next candidate is returned for interior optima (data/metrics.py:98-148)."""
    p, y = _np(probs).astype(np.float64), _np(labels).astype(np.int64)
    cands = np.concatenate([[0.0], np.sort(p), [1.0]])
    if metric == 'accuracy':
        # prediction = p > t: positives are the elements strictly above t
        order = np.argsort(p)
        ps, ys = p[order], y[order]
        n1 = ys.sum()
        # for threshold t: tp = #(y=1, p>t), tn = #(y=0, p<=t)
        le = np.searchsorted(ps, cands, side='right')           # elements <= t
        cum1 = np.concatenate([[0], np.cumsum(ys)])
        tn = le - cum1[le]
        tp = n1 - cum1[le]
        scores = (tp + tn) / max(p.size, 1)
    else:
        scores = np.array([standard_metrics_binary(p, y, t, add_aucroc=False)[metric] for t in cands])
    k = int(scores.argmax())
    if k != len(cands) - 1 and k != 0:
        return float((cands[k] + cands[k + 1]) / 2)
    return float(cands[k])
