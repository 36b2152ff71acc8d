"""bf16-resident GEMM vs the convert-in-flight one on the model shapes."""
import sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
def timeit(run, iters=30):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
shapes = [('qkv_fwd', 0, 0, 2624, 2304, 768, 1, 0), ('attnout_fwd', 0, 0, 2624, 768, 768, 1, 0), ('ffnup_fwd', 0, 0, 2624, 3072, 768, 5, 0), ('ffndown_fwd', 0, 0, 2624, 768, 3072, 1, 0),
          ('ffndown_dgrad', 0, 1, 2624, 3072, 768, 6, 0), ('ffnup_dgrad', 0, 1, 2624, 768, 3072, 4, 0), ('attnout_dgrad', 0, 1, 2624, 768, 768, 0, 0), ('qkv_dgrad', 0, 1, 2624, 768, 2304, 4, 0),
          ('ffn1_wgrad', 1, 1, 3072, 768, 2624, 0, 1), ('ffn2_wgrad', 1, 1, 768, 3072, 2624, 0, 1), ('qkv_wgrad', 1, 1, 2304, 768, 2624, 0, 1), ('o_wgrad', 1, 1, 768, 768, 2624, 0, 1)]
tot = [0.0, 0.0]
for name, akm, bkm, M, N, K, epi, beta in shapes:
    A = torch.randn((K, M) if akm else (M, K), device='cuda'); B = torch.randn((K, N) if bkm else (N, K), device='cuda')
    C = torch.zeros(M, N, device='cuda'); Cb = torch.zeros(M, N, dtype=torch.bfloat16, device='cuda')
    bias = torch.randn(N, device='cuda'); aux = torch.randn(M, N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
    Ab, Bb = A.bfloat16(), B.bfloat16()
    row = []
    for cfg in (1, 4):
        ms = timeit(lambda: L.check(lib.uniter_gemm_bf16_cfg(cfg, akm, bkm, M, N, K, L.ptr(A), A.shape[1], L.ptr(B), B.shape[1], L.ptr(C), N, epi, L.ptr(bias), L.ptr(aux), L.ptr(auxo), N, beta, L.cur_stream())))
        mr = timeit(lambda: L.check(lib.uniter_gemm_bf16res_cfg(cfg, akm, bkm, M, N, K, L.ptr(Ab), Ab.shape[1], L.ptr(Bb), Bb.shape[1], L.ptr(C), N, L.ptr(Cb) if not beta else None, N, epi, L.ptr(bias), L.ptr(aux), L.ptr(auxo), N, beta, L.cur_stream())))
        row.append('cfg%d hybrid %.4fms %4.0fTF | resident %.4fms %4.0fTF' % (cfg, ms, 2.0 * M * N * K / ms / 1e9, mr, 2.0 * M * N * K / mr / 1e9))
    print('%-14s %s' % (name, '  ||  '.join(row)), flush=True)
