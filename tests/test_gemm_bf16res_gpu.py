"""GPU: GEMM with bf16-resident operands (bf16 in HBM, fp32 accumulate, fp32 and/or bf16 output).
Reference: the same bf16 values multiplied in float64."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cast(L, lib, x):
    d = x.cuda().contiguous()
    o = torch.empty(d.shape, dtype=torch.bfloat16, device='cuda')
    L.check(lib.uniter_cast_bf16(L.ptr(d), L.ptr(o), d.numel(), L.cur_stream()))
    return o


def _run(cfg, akm, bkm, M, N, K, epi, beta, want_bf16=True, seed=0):
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(seed)
    A = torch.randn((K, M) if akm else (M, K), generator=g)
    B = torch.randn((K, N) if bkm else (N, K), generator=g)
    bias, aux, C0 = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    dA, dB = _cast(L, lib, A), _cast(L, lib, B)
    assert torch.equal(dA.cpu(), A.bfloat16()) and torch.equal(dB.cpu(), B.bfloat16())     # RNE cast kernel
    Ar, Br = A.bfloat16().double(), B.bfloat16().double()
    ref = (Ar.t() if akm else Ar) @ (Br if bkm else Br.t())
    if epi in (1, 5):
        ref = ref + bias.double()
    pre = None
    if epi == 5:
        x = ref
        pre = 0.5 * (1 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        ref = x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))
    if epi == 4:
        ref = ref + aux.double()
    if epi == 6:
        ref = ref * aux.double()
    if beta:
        ref = ref + C0.double()
    dbias, daux, dC = (t.cuda().contiguous() for t in (bias, aux, C0))
    dauxo = torch.empty(M, N, device='cuda')
    dCb = torch.empty(M, N, dtype=torch.bfloat16, device='cuda') if want_bf16 else None
    L.check(lib.uniter_gemm_bf16res_cfg(cfg, int(akm), int(bkm), M, N, K, L.ptr(dA), dA.shape[1], L.ptr(dB), dB.shape[1],
                                        L.ptr(dC), N, L.ptr(dCb) if want_bf16 else None, N, epi, L.ptr(dbias),
                                        L.ptr(daux), L.ptr(dauxo), N, beta, L.cur_stream()), 'gemm_bf16res')
    torch.cuda.synchronize()
    err = (dC.cpu().double() - ref).abs().max().item()
    assert err < 1e-4 * math.sqrt(K), (cfg, akm, bkm, M, N, K, epi, beta, err)
    if want_bf16:
        assert torch.equal(dCb.cpu(), dC.cpu().bfloat16())          # the bf16 copy is the rounded fp32 output
    if epi == 5:
        assert (dauxo.cpu().double() - pre).abs().max().item() < 1e-4 * math.sqrt(K)


@pytest.mark.parametrize('cfg', [0, 1, 4])
@pytest.mark.parametrize('layout', [(0, 0), (0, 1), (1, 1)])
def test_gemm_bf16res_layouts(cfg, layout):
    akm, bkm = layout
    _run(cfg, akm, bkm, M=168, N=192, K=128, epi=0, beta=0)         # ragged M
    _run(cfg, akm, bkm, M=320, N=256, K=192, epi=1 if not bkm else 4, beta=0)
    _run(cfg, akm, bkm, M=64, N=128, K=64, epi=0, beta=1, want_bf16=False)


@pytest.mark.parametrize('cfg', [0, 1, 4])
def test_gemm_bf16res_model_shapes(cfg):
    _run(cfg, 0, 0, M=2624, N=3072, K=768, epi=5, beta=0)
    _run(cfg, 0, 1, M=2624, N=768, K=3072, epi=4, beta=0)
    _run(cfg, 0, 1, M=2624, N=3072, K=768, epi=6, beta=0)
    _run(cfg, 1, 1, M=768, N=3072, K=2624, epi=0, beta=1, want_bf16=False)      # stream-K
    _run(cfg, 1, 1, M=768, N=256, K=1458, epi=0, beta=1, want_bf16=False)       # ragged K


def test_gemm_bf16res_rejects_beta_with_bf16_copy():
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    A = torch.zeros(64, 64, dtype=torch.bfloat16, device='cuda'); C = torch.zeros(64, 64, device='cuda')
    rc = lib.uniter_gemm_bf16res_cfg(0, 0, 0, 64, 64, 64, L.ptr(A), 64, L.ptr(A), 64, L.ptr(C), 64, L.ptr(A), 64, 0, None, None,
                                     None, 0, 1, L.cur_stream())
    assert rc != 0


def test_gemm_bf16res_rejects_misaligned():
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    x = torch.zeros(64, 64, dtype=torch.bfloat16, device='cuda')
    c = torch.zeros(64, 64, device='cuda')
    rc = lib.uniter_gemm_bf16res_cfg(0, 0, 0, 64, 64, 60, L.ptr(x), 64, L.ptr(x), 64, L.ptr(c), 64, None, 0, 0, None,
                                     None, None, 0, 0, L.cur_stream())
    assert rc != 0 and b'gemm_bf16res' in lib.uniter_last_error()
