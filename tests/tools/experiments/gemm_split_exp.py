import sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
def bench(fn, cfg, akm, bkm, M, N, K, epi=0, iters=20):
    A = torch.randn((K, M) if akm else (M, K), device='cuda'); B = torch.randn((K, N) if bkm else (N, K), device='cuda')
    C = torch.zeros(M, N, device='cuda'); bias = torch.randn(N, device='cuda'); aux = torch.randn(M, N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
    run = lambda: L.check(fn(cfg, akm, bkm, M, N, K, L.ptr(A), A.shape[1], L.ptr(B), B.shape[1], L.ptr(C), N, epi, L.ptr(bias), L.ptr(aux), L.ptr(auxo), N, 0, L.cur_stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9
for name, akm, bkm, M, N, K, epi in [('qkv', 0, 0, 2624, 2304, 768, 1), ('attno', 0, 0, 2624, 768, 768, 1), ('ffnup', 0, 0, 2624, 3072, 768, 2), ('ffndn', 0, 0, 2624, 768, 3072, 1),
                                     ('dgrad1', 0, 1, 2624, 3072, 768, 3), ('dgrad2', 0, 1, 2624, 768, 3072, 4), ('wgrad', 1, 1, 3072, 768, 2624, 0), ('wgrad_o', 1, 1, 768, 768, 2624, 0), ('sq4096', 0, 0, 4096, 4096, 4096, 0)]:
    nat = bench(lib.uniter_gemm_f32_cfg, 0, akm, bkm, M, N, K, epi)
    print(name, 'native %.4fms %.0fTF | ' % nat + ' | '.join('x3cfg%d %.4fms %.0fTF' % ((c,) + bench(lib.uniter_gemm_f32x3_cfg, c, akm, bkm, M, N, K, epi)) for c in (1, 2, 3, 4)), flush=True)
