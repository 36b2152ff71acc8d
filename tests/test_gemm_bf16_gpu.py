"""GPU: mixed-precision GEMM (bf16 MFMA, fp32 storage).  The reference product uses the SAME
bf16-rounded operands in float64, so the tolerance only has to cover fp32 accumulation order:
any layout / fragment-mapping error shows up at O(1)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(cfg, akm, bkm, M, N, K, epi, beta, seed=0):
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((K, M) if akm else (M, K), generator=g)
    B = torch.randn((K, N) if bkm else (N, K), generator=g)
    bias, aux, C0 = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    Ar, Br = A.bfloat16().double(), B.bfloat16().double()
    ref = (Ar.t() if akm else Ar) @ (Br if bkm else Br.t())
    if epi in (1, 2, 5):
        ref = ref + bias.double()
    pre = ref.clone()
    if epi == 5:      # gelu(u) out, gelu'(u) to aux_out
        x = ref
        pre = 0.5 * (1 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        ref = x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if epi == 6:
        ref = ref * aux.double()
    if epi == 2:
        ref = ref * 0.5 * (1.0 + torch.erf(ref / math.sqrt(2.0)))
    if epi == 3:
        x = aux.double()
        ref = ref * (0.5 * (1 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi))
    if epi == 4:
        ref = ref + aux.double()
    if beta:
        ref = ref + C0.double()
    dA, dB, dbias, daux, dC = (t.cuda().contiguous() for t in (A, B, bias, aux, C0))
    dauxo = torch.empty(M, N, device='cuda')
    L.check(lib.uniter_gemm_bf16_cfg(cfg, int(akm), int(bkm), M, N, K, L.ptr(dA), dA.shape[1], L.ptr(dB), dB.shape[1],
                                     L.ptr(dC), N, epi, L.ptr(dbias), L.ptr(daux), L.ptr(dauxo), N, beta,
                                     L.cur_stream()), 'gemm_bf16')
    torch.cuda.synchronize()
    err = (dC.cpu().double() - ref).abs().max().item()
    assert err < 1e-4 * math.sqrt(K), (cfg, akm, bkm, M, N, K, epi, beta, err)
    if epi in (2, 5):
        assert (dauxo.cpu().double() - pre).abs().max().item() < 1e-4 * math.sqrt(K)


@pytest.mark.parametrize('cfg', [0, 1, 2, 3, 4])
@pytest.mark.parametrize('layout', [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_bf16_layouts(cfg, layout):
    akm, bkm = layout
    _run(cfg, akm, bkm, M=164, N=192, K=128, epi=0, beta=0)        # ragged M
    _run(cfg, akm, bkm, M=320, N=256, K=192, epi=1 if not bkm else 4, beta=0)
    _run(cfg, akm, bkm, M=64, N=128, K=64, epi=0, beta=1)


@pytest.mark.parametrize('cfg', [0, 1, 4])
def test_gemm_bf16_model_shapes(cfg):
    _run(cfg, 0, 0, M=2624, N=3072, K=768, epi=2, beta=0)
    _run(cfg, 0, 1, M=2624, N=768, K=3072, epi=4, beta=0)
    _run(cfg, 0, 1, M=2624, N=3072, K=768, epi=3, beta=0)
    _run(cfg, 1, 1, M=768, N=3072, K=2624, epi=0, beta=1)          # stream-K (atomics)
    _run(cfg, 1, 1, M=768, N=768, K=2624, epi=0, beta=1)
    _run(cfg, 0, 0, M=2624, N=3072, K=768, epi=5, beta=0)
    _run(cfg, 0, 1, M=2624, N=3072, K=768, epi=6, beta=0)
    _run(cfg, 1, 1, M=768, N=256, K=1458, epi=0, beta=1)           # ragged K (packed batches): k-major operands only


def test_gemm_bf16_falls_back_to_fp32_when_k_not_multiple_of_64():
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    A, B = torch.randn(48, 48), torch.randn(64, 48)
    dA, dB, dC = A.cuda(), B.cuda(), torch.empty(48, 64, device='cuda')
    L.check(lib.uniter_gemm_bf16_cfg(0, 0, 0, 48, 64, 48, L.ptr(dA), 48, L.ptr(dB), 48, L.ptr(dC), 64, 0, None, None,
                                     None, 0, 0, L.cur_stream()))
    # exact fp32 product (no bf16 rounding of the operands)
    assert (dC.cpu().double() - A.double() @ B.double().t()).abs().max() < 1e-4
