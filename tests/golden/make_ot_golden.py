"""Golden vectors of the optimal-transport distance (IPOT) of the reference: /root/reference/model/ot.py imported as it is
(run in the build container only; the reference does not travel to the GPU box).  Inputs and the reference's outputs -- the
distance, the transport plan T of ipot(), the cost matrix and the gradients of sum(distance) w.r.t. both embeddings -- go to
tests/golden/ot_golden.npz.  ot.py's trace() selects the diagonal with a uint8 mask, which this torch no longer accepts in
masked_select: the script patches torch.eye's dtype request to bool for that call (the same diagonal), nothing else."""
import importlib.util, os, sys
import numpy as np
import torch

REF = '/root/reference/model/ot.py'
spec = importlib.util.spec_from_file_location('ref_ot', REF)
ot = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ot)

_eye = torch.eye
def _eye_bool(*a, **k):
    if k.get('dtype') is torch.uint8:
        k['dtype'] = torch.bool
    return _eye(*a, **k)


def run(case, B, M, N, D, txt_lens, img_lens, seed, beta=0.5, iteration=50, k=1, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(B, M, D, generator=g) * scale).requires_grad_(True)
    y = (torch.randn(B, N, D, generator=g) * scale + 0.3 * x.detach()[:, :1, :].expand(B, N, D)).requires_grad_(True)
    txt_pad = torch.zeros(B, M, dtype=torch.bool); img_pad = torch.zeros(B, N, dtype=torch.bool)
    for b in range(B):
        txt_pad[b, txt_lens[b]:] = True
        img_pad[b, img_lens[b]:] = True
    torch.eye = _eye_bool
    try:
        dist = ot.optimal_transport_dist(x, y, txt_pad, img_pad, beta, iteration, k)
        cost = ot.cost_matrix_cosine(x.detach(), y.detach())
        joint = txt_pad.unsqueeze(-1) | img_pad.unsqueeze(-2)
        cost.masked_fill_(joint, 0)
        tl = (M - txt_pad.sum(1)).float(); il = (N - img_pad.sum(1)).float()
        T = ot.ipot(cost, tl, txt_pad, il, img_pad, joint, beta, iteration, k)
    finally:
        torch.eye = _eye
    dist.sum().backward()
    return {case + '_x': x.detach().numpy(), case + '_y': y.detach().numpy(), case + '_txt_pad': txt_pad.numpy(), case + '_img_pad': img_pad.numpy(),
            case + '_beta': np.float32(beta), case + '_iteration': np.int32(iteration), case + '_k': np.int32(k),
            case + '_dist': dist.detach().numpy(), case + '_T': T.numpy(), case + '_cost': cost.numpy(),
            case + '_dx': x.grad.numpy(), case + '_dy': y.grad.numpy()}


if __name__ == '__main__':
    out = {}
    out.update(run('a', 3, 12, 7, 16, [12, 5, 9], [7, 7, 2], seed=1))
    out.update(run('b', 2, 40, 36, 64, [40, 17], [36, 20], seed=2, scale=3.0))
    out.update(run('c', 2, 9, 5, 8, [9, 1], [5, 1], seed=3, beta=0.3, iteration=20, k=1))      # one-token / one-region sample (k > 1: ot.py:62 raises a shape error)
    out['cases'] = np.array(['a', 'b', 'c'])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ot_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items() if k.endswith('_dist')}, out['a_dist'])
