"""GPU: fp32-accurate GEMM from pre-split bf16 planes (split once, six MFMA products)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _planes(L, lib, x, transpose=False):
    rows, cols = x.shape
    pr, pc = (cols, rows) if transpose else (rows, cols)
    planes = torch.empty(3, pr, pc, dtype=torch.bfloat16, device='cuda')
    L.check(lib.uniter_split_planes(L.ptr(x), rows, cols, cols, L.ptr(planes), pc, pr * pc, int(transpose), L.cur_stream()))
    return planes


def test_split_planes_is_exact():
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    x = (torch.randn(72, 96, device='cuda') * torch.logspace(-6, 2, 96, device='cuda')).contiguous()
    for tr in (False, True):
        p = _planes(L, lib, x, tr).float()
        rec = (p[0].double() + p[1].double() + p[2].double()).float()
        ref = x.t() if tr else x
        assert torch.equal(rec, ref), tr                       # x1 + x2 + x3 == x bit for bit


@pytest.mark.parametrize('cfg', [0, 1, 2, 3, 4])
def test_gemm_planes_accuracy(cfg):
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    for (M, N, K, epi) in ((164, 192, 128, 0), (320, 256, 192, 1), (2624, 768, 768, 4), (2624, 3072, 768, 2)):
        g = torch.Generator().manual_seed(M)
        A, W = torch.randn(M, K, generator=g).cuda(), torch.randn(N, K, generator=g).cuda()
        bias, aux = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
        ref = A.double() @ W.double().t()
        if epi in (1, 2):
            ref = ref + bias.double()
        pre = ref.clone()
        if epi == 2:
            ref = ref * 0.5 * (1.0 + torch.erf(ref / math.sqrt(2.0)))
        if epi == 4:
            ref = ref + aux.double()
        Ap, Wp = _planes(L, lib, A), _planes(L, lib, W)
        C = torch.empty(M, N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
        L.check(lib.uniter_gemm_planes_cfg(cfg, M, N, K, L.ptr(Ap), K, M * K, L.ptr(Wp), K, N * K, L.ptr(C), N, epi,
                                           L.ptr(bias), L.ptr(aux), L.ptr(auxo), N, 0, L.cur_stream()))
        Cn = torch.empty(M, N, device='cuda')
        L.check(lib.uniter_gemm_f32(0, 0, M, N, K, L.ptr(A), K, L.ptr(W), K, L.ptr(Cn), N, epi, L.ptr(bias), L.ptr(aux),
                                    None, N, 0, L.cur_stream()))
        torch.cuda.synchronize()
        e_pl = (C.double() - ref).abs().max().item()
        e_nat = (Cn.double() - ref).abs().max().item()
        assert e_pl <= 3.0 * e_nat + 1e-6, (cfg, M, N, K, e_pl, e_nat)
        if epi == 2:
            assert (auxo.double() - pre).abs().max().item() <= 3.0 * e_nat + 1e-5
    # dgrad form: dX = dY @ W  ==  dY @ (W^T)^T with transposed weight planes
    g = torch.Generator().manual_seed(1)
    dY, W = torch.randn(300, 256, generator=g).cuda(), torch.randn(256, 128, generator=g).cuda()    # W [N=256, K=128]
    dYp, WTp = _planes(L, lib, dY), _planes(L, lib, W, transpose=True)                                 # W^T [128, 256]
    dX = torch.empty(300, 128, device='cuda')
    L.check(lib.uniter_gemm_planes_cfg(cfg, 300, 128, 256, L.ptr(dYp), 256, 300 * 256, L.ptr(WTp), 256, 128 * 256,
                                       L.ptr(dX), 128, 0, None, None, None, 0, 0, L.cur_stream()))
    assert (dX.double() - dY.double() @ W.double()).abs().max().item() < 1e-4
