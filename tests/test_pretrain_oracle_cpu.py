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


@pytest.mark.parametrize('task', ['mlm', 'mrfr', 'itm'])
def test_pretrain_oracle_bf16_mode_is_the_fp32_oracle_plus_rounding(pre, task):
    """prec='bf16' (the arithmetic of the HIP path's bf16 mode) only rounds matrix operands: against the
    reference-pinned fp32 mode it moves the outputs by bf16-sized noise (2^-9 relative per operand), not more, not
    nothing -- and with operands that ARE bf16 numbers already a head product is exact."""
    from oracle import uniter_oracle as O
    sd = sd_from_npz(pre)
    b = batch_from_npz(pre)
    if task == 'mrfr':
        b['img_feat'] = b['img_feat_masked']
    fn = getattr(PO, 'forward_' + task)
    with torch.no_grad():
        sf = fn(sd, TINY, b, compute_loss=False, prec='fp32')
        sb = fn(sd, TINY, b, compute_loss=False, prec='bf16')
    rel = (sb - sf).pow(2).mean().sqrt().item() / sf.pow(2).mean().sqrt().item()
    assert 5e-5 < rel < 3e-2, rel          # (the two-way ITM scores of the tiny model average most of it out: 1.4e-4)
    x, w = O.bf(torch.randn(5, 64)), O.bf(torch.randn(7, 64))
    assert torch.equal(O.linear(x, w, None, 'bf16'), torch.nn.functional.linear(x, w))
    # gradients flow through the rounding as through the identity, operands rounded: dW = bf(dy)^T bf(x)
    xw = torch.randn(3, 64, requires_grad=True)
    ww = torch.randn(4, 64, requires_grad=True)
    O.linear(xw, ww, None, 'bf16').sum().backward()
    assert torch.allclose(ww.grad, torch.ones(4, 3) @ O.bf(xw.detach()))
    assert torch.allclose(xw.grad, torch.ones(3, 4) @ O.bf(ww.detach()))
