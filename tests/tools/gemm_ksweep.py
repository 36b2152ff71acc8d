"""Fixed cost of a GEMM launch: time vs K at the model's output shapes (cfg 0)."""
import sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
def bench(akm, bkm, M, N, K, epi, beta, iters=50):
    A = torch.randn((K, M) if akm else (M, K), device='cuda')
    B = torch.randn((K, N) if bkm else (N, K), device='cuda')
    C = torch.zeros(M, N, device='cuda'); bias = torch.randn(N, device='cuda'); aux = torch.randn(M, N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
    def run():
        L.check(lib.uniter_gemm_f32_cfg(0, akm, bkm, M, N, K, L.ptr(A), A.shape[1], L.ptr(B), B.shape[1], L.ptr(C), N, epi, L.ptr(bias), L.ptr(aux), L.ptr(auxo), N, beta, L.cur_stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, epi in [('N3072 bias+gelu (2 outputs)', 2624, 3072, 2), ('N3072 bias+gelu+dgelu fast', 2624, 3072, 5), ('N3072 dgelu', 2624, 3072, 3), ('N3072 mul', 2624, 3072, 6), ('N3072 bias', 2624, 3072, 1), ('N2304 bias', 2624, 2304, 1), ('N768 bias', 2624, 768, 1), ('N768 add', 2624, 768, 4)]:
    row = []
    for K in (32, 256, 768, 3072):
        us = bench(0, 0, M, N, K, epi, 0)
        row.append('K%d %.1fus' % (K, us))
    print('%-28s %s' % (name, ' | '.join(row)), flush=True)
