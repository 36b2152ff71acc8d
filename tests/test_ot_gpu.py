"""GPU: meme_challenge_amd.ot.optimal_transport_dist (csrc/ot.hip, through the C ABI) against the golden vectors the reference's
model/ot.py produced (tests/golden/ot_golden.npz) and against the oracle on larger seeded inputs: distance, transport plan,
gradients w.r.t. both embeddings; the reference's error behaviour for k > 1; the pretraining module's opt-in use of it."""
import os

import numpy as np
import pytest
import torch

from oracle import ot_oracle as OT

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ot_golden.npz'))


def _close(got, ref, tol, what):
    ref = ref.double().cpu(); got = got.detach().double().cpu()
    assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), (what, (got - ref).abs().max().item())


@pytest.mark.parametrize('case', ['a', 'b', 'c'])
def test_ot_dist_matches_reference_golden(case):
    from meme_challenge_amd.ot import optimal_transport_dist
    from meme_challenge_amd import _lib as Lb
    x = torch.from_numpy(G[case + '_x']).cuda().requires_grad_(True)
    y = torch.from_numpy(G[case + '_y']).cuda().requires_grad_(True)
    tp, ip = torch.from_numpy(G[case + '_txt_pad']).cuda(), torch.from_numpy(G[case + '_img_pad']).cuda()
    beta, it = float(G[case + '_beta']), int(G[case + '_iteration'])
    dist = optimal_transport_dist(x, y, tp, ip, beta, it, 1)
    dist.sum().backward()
    _close(dist, torch.from_numpy(G[case + '_dist']), 5e-5, 'dist')
    _close(x.grad, torch.from_numpy(G[case + '_dx']), 1e-4, 'dx')
    _close(y.grad, torch.from_numpy(G[case + '_dy']), 1e-4, 'dy')
    assert x.grad[tp].abs().max().item() == 0 if tp.any() else True        # padded rows: no gradient
    # the transport plan itself, through the C entry point
    B, M, D = x.shape; N = y.shape[1]
    T = torch.empty(B, N, M, device='cuda'); d2 = torch.empty(B, device='cuda')
    xd, yd, tp8, ip8 = x.detach(), y.detach(), tp.to(torch.uint8), ip.to(torch.uint8)      # (alive across the asynchronous launch)
    Lb.check(Lb.lib().uniter_ot_dist_fwd(Lb.ptr(xd), Lb.ptr(yd), Lb.ptr(tp8), Lb.ptr(ip8), Lb.ptr(d2), Lb.ptr(T), B, M, N, D, beta, it,
                                         Lb.cur_stream()))
    torch.cuda.synchronize()
    _close(T, torch.from_numpy(G[case + '_T']), 1e-4, 'T')
    assert torch.equal(d2, dist.detach())


@pytest.mark.parametrize('B,M,N,D,seed', [(16, 60, 36, 768, 5), (4, 128, 64, 768, 6), (3, 1, 1, 8, 7), (2, 33, 100, 1024, 8)])
def test_ot_dist_matches_oracle_at_model_sizes(B, M, N, D, seed):
    from meme_challenge_amd.ot import optimal_transport_dist
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, M, D, generator=g); y = torch.randn(B, N, D, generator=g) + 0.2 * x[:, :1, :]
    tp = torch.zeros(B, M, dtype=torch.bool); ip = torch.zeros(B, N, dtype=torch.bool)
    for b in range(1, B):
        tp[b, max(1, M - (3 * b) % M):] = True
        ip[b, max(1, N - (5 * b) % N):] = True
    w = torch.randn(B, generator=g)                      # a non-uniform gradient into the distances
    xo, yo = x.double().requires_grad_(True), y.double().requires_grad_(True)
    do, To, _ = OT.optimal_transport_dist(xo, yo, tp, ip)
    (do * w.double()).sum().backward()
    xd, yd = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    d = optimal_transport_dist(xd, yd, tp.cuda(), ip.cuda())
    (d * w.cuda()).sum().backward()
    _close(d, do, 1e-4, 'dist')
    _close(xd.grad, xo.grad, 2e-4, 'dx')
    _close(yd.grad, yo.grad, 2e-4, 'dy')
    with torch.no_grad():                                 # no gradient requested: no plan stored, same distances
        assert torch.equal(optimal_transport_dist(x.cuda(), y.cuda(), tp.cuda(), ip.cuda()), d.detach())


def test_ot_dist_backward_through_non_contiguous_slices():
    """forward_itm hands slices ctx[:, :tl] / ctx[:, tl:tl + il] of one tensor (non-contiguous for B >= 2): the transport plan
    must be saved for them too (the decision was taken on the contiguous copies, which never require grad)."""
    from meme_challenge_amd.ot import optimal_transport_dist
    g = torch.Generator().manual_seed(11)
    B, tl, il, D = 3, 20, 12, 64
    ctx = torch.randn(B, tl + il + 1, D, generator=g)
    tp = torch.zeros(B, tl, dtype=torch.bool); ip = torch.zeros(B, il, dtype=torch.bool)
    tp[1, 15:] = True; ip[2, 7:] = True
    w = torch.randn(B, generator=g)
    co = ctx.double().requires_grad_(True)
    do, _, _ = OT.optimal_transport_dist(co[:, :tl], co[:, tl:tl + il], tp, ip)
    (do * w.double()).sum().backward()
    cd = ctx.cuda().requires_grad_(True)
    xs, ys = cd[:, :tl, :], cd[:, tl:tl + il, :]
    assert not xs.is_contiguous() and not ys.is_contiguous()
    d = optimal_transport_dist(xs, ys, tp.cuda(), ip.cuda())
    (d * w.cuda()).sum().backward()
    _close(d, do, 1e-4, 'dist')
    _close(cd.grad, co.grad, 2e-4, 'd ctx')
    assert cd.grad[:, tl + il].abs().max().item() == 0


def test_ot_dist_error_behaviour():
    from meme_challenge_amd.ot import optimal_transport_dist
    from meme_challenge_amd._lib import UniterHipError
    x = torch.randn(1, 4, 8).cuda(); y = torch.randn(1, 3, 8).cuda()
    tp = torch.zeros(1, 4, dtype=torch.bool).cuda(); ip = torch.zeros(1, 3, dtype=torch.bool).cuda()
    with pytest.raises(ValueError):
        optimal_transport_dist(x, y, tp, ip, k=2)        # the reference raises for k > 1 too (ot.py:62)
    with pytest.raises(ValueError):
        optimal_transport_dist(x, y[:, :, :4], tp, ip)
    with pytest.raises(RuntimeError):
        optimal_transport_dist(x.cpu(), y.cpu(), tp.cpu(), ip.cpu())      # no CPU path
    for m_, n_ in ((128, 96), ):          # 12288 entries fit the registers, not the 160-KB LDS
        with pytest.raises(UniterHipError):
            optimal_transport_dist(torch.randn(1, m_, 8).cuda(), torch.randn(1, n_, 8).cuda(), torch.zeros(1, m_, dtype=torch.bool).cuda(),
                                   torch.zeros(1, n_, dtype=torch.bool).cuda())
    big = torch.randn(1, 200, 8).cuda(); bigy = torch.randn(1, 100, 8).cuda()
    with pytest.raises(UniterHipError):
        optimal_transport_dist(big, bigy, torch.zeros(1, 200, dtype=torch.bool).cuda(), torch.zeros(1, 100, dtype=torch.bool).cuda())
