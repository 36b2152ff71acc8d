"""Lab timing of the resident bf16 GEMM on a few model shapes (used with UNITER_LIB_VARIANT builds)."""
import os, sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
def timeit(run, iters=40):
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
MM = int(os.environ.get('LAB_M', '2624'))
HH = int(os.environ.get('LAB_H', '768')); II = 4 * HH
shapes = [('qkv_fwd', 0, 0, MM, 3 * HH, HH, 1, 0), ('attnout_fwd', 0, 0, MM, HH, HH, 1, 0), ('ffnup_fwd', 0, 0, MM, II, HH, 5, 0), ('ffndown_fwd', 0, 0, MM, HH, II, 1, 0),
          ('ffndown_dgrad', 0, 1, MM, II, HH, 6, 0), ('ffnup_dgrad', 0, 1, MM, HH, II, 4, 0), ('qkv_dgrad', 0, 1, MM, HH, 3 * HH, 4, 0), ('ffn1_wgrad', 1, 1, II, HH, MM, 0, 1), ('ffn2_wgrad', 1, 1, HH, II, MM, 0, 1)]
cfgs = [int(c) for c in os.environ.get('LAB_CFGS', '1,4').split(',')]
for name, akm, bkm, M, N, K, epi, beta in shapes:
    A = torch.randn((K, M) if akm else (M, K), device='cuda').bfloat16(); B = torch.randn((K, N) if bkm else (N, K), device='cuda').bfloat16()
    C = torch.zeros(M, N, device='cuda'); Cb = torch.zeros(M, N, dtype=torch.bfloat16, device='cuda')
    bias = torch.randn(N, device='cuda'); aux = torch.randn(M, N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
    row = []
    for cfg in cfgs:
        args = (cfg, akm, bkm, M, N, K, L.ptr(A), A.shape[1], L.ptr(B), B.shape[1], L.ptr(C) if epi != 5 else None, N, L.ptr(Cb) if not beta else None, N, epi, L.ptr(bias), L.ptr(aux), L.ptr(auxo), N, beta, L.cur_stream())
        f = lib.uniter_gemm_bf16res_cfg
        mr = timeit(lambda: f(*args))
        row.append('cfg%d %.4fms %4.0fTF' % (cfg, mr, 2.0 * M * N * K / mr / 1e9))
    print('%-8s %-14s %s' % (os.environ.get('UNITER_LIB_VARIANT', 'cur'), name, '  |  '.join(row)), flush=True)
