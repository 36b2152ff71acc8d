"""Probe: a deterministic two-way split of K for the fp32 GEMM on N = 768 shapes -- zero C, then the two K halves as
two launches (epilogue on the first, C += by atomics on both) on two streams, against the single launch."""
import sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
s2 = torch.cuda.Stream()
def timeit(run, iters=40):
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, M, N, K, bkm, epi in (('attnout_fwd', 2624, 768, 768, 0, 1), ('ffndown_fwd', 2624, 768, 3072, 0, 1), ('ffnup_dgrad', 2624, 768, 3072, 1, 4), ('attnout_dgrad', 2624, 768, 768, 1, 0), ('qkv_dgrad', 2624, 768, 2304, 1, 4)):
    A = torch.randn(M, K, device='cuda'); B = torch.randn((K, N) if bkm else (N, K), device='cuda')
    bias = torch.randn(N, device='cuda'); aux = torch.randn(M, N, device='cuda')
    C1 = torch.empty(M, N, device='cuda'); C2 = torch.empty(M, N, device='cuda')
    ldb = B.shape[1]; h = K // 2
    pa, pb, pc1, pc2, pbias, paux = L.ptr(A), L.ptr(B), L.ptr(C1), L.ptr(C2), L.ptr(bias), L.ptr(aux)
    import ctypes
    pa2 = ctypes.c_void_p(A.data_ptr() + h * 4)
    pb2 = ctypes.c_void_p(B.data_ptr() + (h * ldb * 4 if bkm else h * 4))
    f = lib.uniter_gemm_f32_cfg
    def single():
        f(0, 0, bkm, M, N, K, pa, K, pb, ldb, pc1, N, epi, pbias, paux, None, N, 0, L.cur_stream())
    ev0 = torch.cuda.Event(); ev1 = torch.cuda.Event()
    def split():
        C2.zero_()
        ev0.record()
        s2.wait_event(ev0)
        f(0, 0, bkm, M, N, h, pa, K, pb, ldb, pc2, N, epi, pbias, paux, None, N, 1, L.cur_stream())
        f(0, 0, bkm, M, N, K - h, pa2, K, pb2, ldb, pc2, N, 0, None, None, None, N, 1, s2.cuda_stream)
        ev1.record(s2)
        torch.cuda.current_stream().wait_event(ev1)
    t1, t2 = timeit(single), timeit(split)
    single(); split(); torch.cuda.synchronize()
    err = (C1 - C2).abs().max().item()
    print('%-14s single %.1f us | zero + two halves on two streams %.1f us | max diff %.2e' % (name, t1, t2, err))
