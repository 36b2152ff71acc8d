"""GPU: fp32-accurate GEMM on the bf16 pipe (3 x bf16 split, 6 products).  Accuracy is measured
against float64 on the UNROUNDED inputs and compared with the native fp32 MFMA kernel."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(cfg, akm, bkm, M, N, K, epi=0, beta=0, seed=0, scale=1.0):
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((K, M) if akm else (M, K), generator=g) * scale
    B = torch.randn((K, N) if bkm else (N, K), generator=g)
    bias, aux, C0 = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    ref = (A.double().t() if akm else A.double()) @ (B.double() if bkm else B.double().t())
    if epi == 1:
        ref = ref + bias.double()
    if epi == 4:
        ref = ref + aux.double()
    if beta:
        ref = ref + C0.double()
    dA, dB, dbias, daux = (t.cuda().contiguous() for t in (A, B, bias, aux))
    errs = []
    for fn in (lib.uniter_gemm_f32x3_cfg, lib.uniter_gemm_f32_cfg):
        dC = C0.cuda().contiguous()
        L.check(fn(cfg if fn is lib.uniter_gemm_f32x3_cfg else 0, int(akm), int(bkm), M, N, K, L.ptr(dA), dA.shape[1],
                   L.ptr(dB), dB.shape[1], L.ptr(dC), N, epi, L.ptr(dbias), L.ptr(daux), None, N, beta, L.cur_stream()))
        torch.cuda.synchronize()
        errs.append((dC.cpu().double() - ref).abs().max().item())
    return errs, ref.abs().max().item()


@pytest.mark.parametrize('cfg', [1, 2, 3, 4])
@pytest.mark.parametrize('layout', [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_split_gemm_is_fp32_accurate(cfg, layout):
    akm, bkm = layout
    for (M, N, K, epi, beta) in ((164, 192, 128, 0, 0), (320, 256, 192, 1 if not bkm else 4, 0), (64, 128, 64, 0, 1)):
        (e_split, e_native), mag = _both(cfg, akm, bkm, M, N, K, epi, beta)
        assert e_split < 3e-6 * math.sqrt(K) * 4, (cfg, layout, M, N, K, e_split)
        assert e_split <= 3.0 * e_native + 1e-6, (cfg, layout, M, N, K, e_split, e_native)


def test_split_gemm_model_shapes_and_dynamic_range():
    for args in ((0, 0, 2624, 3072, 768), (0, 1, 2624, 768, 3072), (1, 1, 768, 3072, 2624)):
        (e_split, e_native), mag = _both(4, *args)
        assert e_split <= 3.0 * e_native + 1e-6 and e_split < 2e-3, (args, e_split, e_native)
    # operands spanning many binades (gradients are ~1e-6 .. 1e-2)
    (e_split, e_native), mag = _both(4, 0, 0, 256, 256, 512, scale=1e-5)
    assert e_split <= 3.0 * e_native + 1e-12, (e_split, e_native)
