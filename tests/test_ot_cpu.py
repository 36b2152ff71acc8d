"""CPU: the oracle's restatement of the reference's optimal-transport distance (model/ot.py) against the golden vectors the
reference itself produced (tests/golden/make_ot_golden.py): distance, transport plan, cost matrix, both gradients."""
import os

import numpy as np
import pytest
import torch

from oracle import ot_oracle as OT

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ot_golden.npz'))


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_ot_oracle_matches_reference_golden(case, dtype):
    x = torch.from_numpy(G[case + '_x']).to(dtype).requires_grad_(True)
    y = torch.from_numpy(G[case + '_y']).to(dtype).requires_grad_(True)
    tp, ip = torch.from_numpy(G[case + '_txt_pad']), torch.from_numpy(G[case + '_img_pad'])
    dist, T, cost = OT.optimal_transport_dist(x, y, tp, ip, float(G[case + '_beta']), int(G[case + '_iteration']), int(G[case + '_k']))
    dist.sum().backward()
    tol = 2e-5 if dtype == torch.float32 else 5e-5          # (the golden vectors are the reference's fp32 run)
    for got, name in ((dist, '_dist'), (T, '_T'), (cost, '_cost'), (x.grad, '_dx'), (y.grad, '_dy')):
        ref = torch.from_numpy(G[case + name]).double()
        assert (got.detach().double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), (case, name)
    # padded rows carry no gradient; a plan's padded entries are zero
    assert x.grad[tp].abs().max().item() == 0 if tp.any() else True
    assert T.detach()[(tp.unsqueeze(-1) | ip.unsqueeze(-2)).transpose(1, 2)].abs().max().item() == 0 if (tp.any() or ip.any()) else True


def test_ot_oracle_rejects_k_above_one_like_the_reference():
    x = torch.randn(1, 3, 4); y = torch.randn(1, 2, 4)
    pad = torch.zeros(1, 3, dtype=torch.bool); ipad = torch.zeros(1, 2, dtype=torch.bool)
    with pytest.raises(ValueError):
        OT.optimal_transport_dist(x, y, pad, ipad, k=2)
