"""CPU: pin the pretraining-heads oracle against the reference's UniterForPretraining outputs."""
import numpy as np
import pytest
import torch

from oracle import pretrain_oracle as PO
from common import TINY as _TINY, sd_from_npz, batch_from_npz, maxdiff
from conftest import load_golden

TINY = dict(_TINY, vocab_size=100)      # vocab % 4 == 0 (the tied-decoder GEMM's alignment rule)


@pytest.fixture(scope='module')
def pre():
    return load_golden('pretrain_tiny.npz')


@pytest.mark.parametrize('task', ['mlm', 'mrfr', 'itm', 'mrc', 'mrc-kl'])
def test_pretrain_heads_match_reference(pre, task):
    sd = {k: v.clone().requires_grad_(True) for k, v in sd_from_npz(pre).items()}
    b = batch_from_npz(pre)
    if task in ('mrfr', 'mrc', 'mrc-kl'):
        b['img_feat'] = b['img_feat_masked']
    if task.startswith('mrc'):
        import functools
        fn = functools.partial(PO.forward_mrc, task=task)
    else:
        fn = getattr(PO, 'forward_' + task)
    with torch.no_grad():
        scores = fn(sd, TINY, b, compute_loss=False)
    assert maxdiff(scores, pre[task + '/scores']) < 2e-5
    loss = fn(sd, TINY, b, compute_loss=True)
    assert maxdiff(loss, pre[task + '/loss']) < 2e-5
    loss.mean().backward()
    for k in pre.files:
        if k.startswith(task + '/grad/'):
            n = k[len(task + '/grad/'):]
            ref = torch.from_numpy(pre[k])
            g = sd[n].grad if sd[n].grad is not None else torch.zeros_like(sd[n])
            assert maxdiff(g, ref) <= 1e-6 + 2e-4 * ref.abs().max().item(), (task, n)
