"""Glitch screen: resident bf16 GEMM epilogues, repeated runs against a float64 reference."""
import sys, math, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
M, N, K = 2624, 3072, 768
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).bfloat16(); B = torch.randn(N, K, generator=g).bfloat16(); bias = torch.randn(N, generator=g); aux = torch.randn(M, N, generator=g)
pre = (A.double() @ B.double().t())
preb = pre + bias.double()
dg = 0.5 * (1 + torch.erf(preb / math.sqrt(2))) + preb * torch.exp(-0.5 * preb * preb) / math.sqrt(2 * math.pi)
gl = preb * 0.5 * (1 + torch.erf(preb / math.sqrt(2)))
dA, dB, db, daux = A.cuda(), B.cuda(), bias.cuda(), aux.cuda()
for epi in (5, 6, 1, 4):
    for cfg in (1,) * 6 + (4,) * 6:
        C = torch.full((M, N), 7.0, device='cuda'); Cb = torch.full((M, N), 7.0, dtype=torch.bfloat16, device='cuda'); auxo = torch.full((M, N), 7.0, device='cuda')
        L.check(lib.uniter_gemm_bf16res_cfg(cfg, 0, 0, M, N, K, L.ptr(dA), K, L.ptr(dB), K, L.ptr(C), N, L.ptr(Cb), N, epi, L.ptr(db), L.ptr(daux), L.ptr(auxo), N, 0, L.cur_stream()))
        torch.cuda.synchronize()
        ref = {5: gl, 6: pre * aux.double(), 1: preb, 4: pre + aux.double()}[epi]
        e = (C.cpu().double() - ref).abs()
        nbad = (e > 1e-2 * (1 + ref.abs())).sum().item()
        eb = (Cb.float().cpu().double() - ref).abs()
        nbadb = (eb > 2e-2 * (1 + ref.abs())).sum().item()
        nx = ((auxo.cpu().double() - dg).abs() > 1e-2).sum().item() if epi == 5 else 0
        print('epi', epi, 'cfg', cfg, 'badC', nbad, 'badCb', nbadb, 'badAux', nx)
