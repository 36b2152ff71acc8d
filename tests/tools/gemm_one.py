"""Run one GEMM shape/config repeatedly (for rocprofv3 --pmc runs)."""
import sys
import torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
cfg, akm, bkm, M, N, K, epi, iters = [int(x) for x in sys.argv[1:9]]
A = torch.randn((K, M) if akm else (M, K), device='cuda')
B = torch.randn((K, N) if bkm else (N, K), device='cuda')
C = torch.zeros(M, N, device='cuda'); bias = torch.randn(N, device='cuda'); aux = torch.randn(M, N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
for _ in range(iters):
    L.check(lib.uniter_gemm_f32_cfg(cfg, akm, bkm, M, N, K, L.ptr(A), A.shape[1], L.ptr(B), B.shape[1], L.ptr(C), N, epi, L.ptr(bias), L.ptr(aux), L.ptr(auxo), N, 0, L.cur_stream()))
torch.cuda.synchronize()
