"""GPU: fp32 MFMA GEMM (uniter_gemm_f32) against a float64 host product."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib, L, cfg, akm, bkm, M, N, K, epi, beta, seed=0):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((K, M) if akm else (M, K), generator=g)
    B = torch.randn((K, N) if bkm else (N, K), generator=g)
    bias = torch.randn(N, generator=g)
    aux = torch.randn(M, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    Am = A.t() if akm else A
    Bm = B if bkm else B.t()
    ref = Am.double() @ Bm.double()
    aux_out_ref = None
    if epi in (1, 2, 5):
        ref = ref + bias.double()
    if epi == 2:
        aux_out_ref = ref.clone()
        ref = ref * 0.5 * (1.0 + torch.erf(ref / math.sqrt(2.0)))
    if epi == 3:
        x = aux.double()
        ref = ref * (0.5 * (1 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi))
    if epi == 4:
        ref = ref + aux.double()
    if epi == 5:      # forward hands gelu'(u) to the backward pass
        x = ref
        aux_out_ref = 0.5 * (1 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        ref = x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if epi == 6:
        ref = ref * aux.double()
    if beta:
        ref = ref + C0.double()
    dA, dB, dbias, daux, dC = (t.cuda().contiguous() for t in (A, B, bias, aux, C0))
    daux_out = torch.empty(M, N, device='cuda')
    rc = lib.uniter_gemm_f32_cfg(cfg, int(akm), int(bkm), M, N, K, L.ptr(dA), dA.shape[1],
                                 L.ptr(dB), dB.shape[1], L.ptr(dC), N, epi, L.ptr(dbias),
                                 L.ptr(daux), L.ptr(daux_out), N, beta, L.cur_stream())
    L.check(rc, 'gemm')
    torch.cuda.synchronize()
    scale = math.sqrt(K)
    err = (dC.cpu().double() - ref).abs().max().item()
    assert err < 2e-5 * scale * 4, (cfg, akm, bkm, M, N, K, epi, beta, err)
    if epi in (2, 5):
        assert (daux_out.cpu().double() - aux_out_ref).abs().max().item() < 2e-5 * scale * 4


@pytest.mark.parametrize('cfg', [0, 1, 2, 3, 4, 21, 24])
@pytest.mark.parametrize('layout', [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_layouts_and_edges(cfg, layout):
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    akm, bkm = layout
    # ragged M (not a tile multiple), K not a multiple of the 32-deep k-tile
    _run(lib, L, cfg, akm, bkm, M=164, N=192, K=72, epi=0, beta=0)
    _run(lib, L, cfg, akm, bkm, M=48, N=128, K=48 if (akm or bkm) else 64, epi=0, beta=1)


@pytest.mark.parametrize('epi', [1, 2, 3, 4, 5, 6])
def test_gemm_epilogues(epi):
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    _run(lib, L, 0, 0, 0 if epi in (1, 2, 5) else 1, M=300, N=256, K=128, epi=epi, beta=0)
    _run(lib, L, 0, 0, 0 if epi in (1, 2, 5) else 1, M=300, N=256, K=32, epi=epi, beta=0)     # one k-tile: no aux prefetch
    _run(lib, L, 21, 0, 0 if epi in (1, 2, 5) else 1, M=300, N=256, K=96, epi=epi, beta=0)
    _run(lib, L, 4, 0, 0 if epi in (1, 2, 5) else 1, M=100, N=64, K=40, epi=epi, beta=0)      # v1 fallback


@pytest.mark.parametrize('cfg', [0, 1, 4, 21, 24])
def test_gemm_model_shapes(cfg):
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    _run(lib, L, cfg, 0, 0, M=2624, N=3072, K=768, epi=2, beta=0)     # FFN up
    _run(lib, L, cfg, 0, 1, M=2624, N=768, K=3072, epi=4, beta=0)     # dgrad
    _run(lib, L, cfg, 1, 1, M=768, N=3072, K=2624, epi=0, beta=1)     # wgrad (accumulate)
    _run(lib, L, cfg, 0, 0, M=576, N=768, K=2048, epi=1, beta=0)      # img_linear


@pytest.mark.parametrize('cfg', [21, 24])
@pytest.mark.parametrize('layout', [(1, 1), (0, 0), (0, 1), (1, 0)])
def test_gemm_streamk_accumulate(cfg, layout):
    """C += A.B with few output tiles and a long K takes the stream-K kernel (partial tiles are
    added with float atomics): weight-gradient shapes plus ragged edges."""
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    akm, bkm = layout
    _run(lib, L, cfg, akm, bkm, M=768, N=768, K=2624, epi=0, beta=1)
    _run(lib, L, cfg, akm, bkm, M=2304, N=768, K=2624, epi=0, beta=1)
    _run(lib, L, cfg, akm, bkm, M=200, N=136, K=8192, epi=0, beta=1)


@pytest.mark.parametrize('epi', [0, 1, 2, 4, 5, 6])
@pytest.mark.parametrize('K', [32, 64, 416, 768])
def test_gemm_many_tiles_all_epilogues(epi, K):
    """More 64x64 tiles than workgroup slots (two tiles per workgroup), every epilogue, ragged edges
    handled by the buffer range check."""
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    bkm = 0 if epi in (1, 2, 5) else 1
    _run(lib, L, 24, 0, bkm, M=2624, N=3072, K=K, epi=epi, beta=0)
    if K == 64:
        _run(lib, L, 24, 0, bkm, M=2500, N=3000, K=K, epi=epi, beta=1)      # ragged edges, accumulate


def test_gemm_rejects_bad_args():
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    x = torch.zeros(64, 64, device='cuda')
    rc = lib.uniter_gemm_f32(0, 0, 64, 64, 62, L.ptr(x), 64, L.ptr(x), 64, L.ptr(x), 64, 0,
                             None, None, None, 0, 0, L.cur_stream())
    assert rc == -2 and b'multiples of 4' in lib.uniter_last_error()


@pytest.mark.parametrize('K,shapes', [(2624, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),
                                      (100, [(64, 128), (132, 68)]), (37, [(256, 4)])])
def test_grouped_weight_gradients(K, shapes):
    """uniter_wgrad_f32_group: up to four dW_p (+)= A_p^T B_p of one reduction length as ONE launch of whole-K tiles --
    the encoder layer's shapes, ragged shapes (tile tails in both dimensions, a reduction length that is no multiple of the
    32-deep k-tile), overwrite and accumulate; every product against float64."""
    import ctypes as C
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(K)
    n = len(shapes)
    As = [torch.randn(K, m, generator=g).cuda() for m, _ in shapes]
    Bs = [torch.randn(K, nn, generator=g).cuda() for _, nn in shapes]
    init = [torch.randn(m, nn, generator=g).cuda() for m, nn in shapes]
    Ms = (C.c_int * n)(*[m for m, _ in shapes]); Ns = (C.c_int * n)(*[nn for _, nn in shapes])
    pa = (C.c_void_p * n)(*[a.data_ptr() for a in As]); pb = (C.c_void_p * n)(*[b.data_ptr() for b in Bs])
    for overwrite in (1, 0):
        outs = [t.clone() for t in init]
        pc = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
        L.check(lib.uniter_wgrad_f32_group(n, Ms, Ns, K, pa, pb, pc, overwrite, L.cur_stream()), 'uniter_wgrad_f32_group')
        torch.cuda.synchronize()
        for a, b, o, i0 in zip(As, Bs, outs, init):
            ref = a.double().t() @ b.double() + (0 if overwrite else i0.double())
            err = (o.double() - ref).abs().max().item()
            assert err <= 2e-6 * K ** 0.5 * max(1.0, ref.abs().max().item()) / 10 + 1e-5, (overwrite, tuple(o.shape), err)
