"""Isolated timing of the balanced walk (UNITER_X3_SK) against the classic one: a layer's grouped weight gradients and the QKV forward
product of UNITER-base at configs[1], HIP events over 30 launches each."""
import ctypes, os, sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from meme_challenge_amd import _lib as L
from test_gemm_x3_gpu import split3
lib = L.lib()
nb = lib.uniter_gemm_x3_balanced_ws_bytes()
ws = torch.zeros(nb // 4, dtype=torch.int32, device='cuda')
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
K = 2624
shapes = [(3072, 768), (768, 3072), (2304, 768), (768, 768)]
As = [split3(torch.randn(K, m, device='cuda')) for m, n in shapes]; Bs = [split3(torch.randn(K, n, device='cuda')) for m, n in shapes]
Cs = [torch.zeros(m, n, device='cuda') for m, n in shapes]
IA, PA = ctypes.c_int * 4, ctypes.c_void_p * 4
Ms, Ns = IA(*[m for m, _ in shapes]), IA(*[n for _, n in shapes])
pa, pb, pc = PA(*[a.data_ptr() for a in As]), PA(*[b.data_ptr() for b in Bs]), PA(*[c.data_ptr() for c in Cs])
for wgs in (0, 240):
    for bal in (0, 1):
        t = timeit(lambda: L.check(lib.uniter_wgrad_x3_group_ws(4, 4, Ms, Ns, K, pa, pb, pc, 1, wgs, None, L.ptr(ws) if bal else None, nb if bal else 0, L.cur_stream())))
        print('weight gradients of a layer, max_wgs %3d, %s walk: %.1f us' % (wgs, 'balanced' if bal else 'classic ', t), flush=True)
for (M, N, Kf) in [(2624, 2304, 768), (2624, 3072, 768)]:
    A3 = split3(torch.randn(M, Kf, device='cuda')); B3 = split3(torch.randn(N, Kf, device='cuda')); bias = torch.randn(N, device='cuda')
    C = torch.empty(M, N, device='cuda')
    for bal in (0, 1):
        t = timeit(lambda: L.check(lib.uniter_gemm_x3_cfg_ws(4, 1, 0, 0, M, N, Kf, L.ptr(A3), 3 * Kf, Kf, L.ptr(B3), 3 * Kf, Kf, L.ptr(C), N, M * N, None, 3 * N, N,
                                                         1, L.ptr(bias), None, None, N, L.ptr(ws) if bal else None, nb if bal else 0, L.cur_stream())))
        print('forward %d x %d x %d + bias, %s walk: %.1f us' % (M, N, Kf, 'balanced' if bal else 'classic ', t), flush=True)
