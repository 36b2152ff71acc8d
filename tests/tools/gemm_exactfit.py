"""Steady state of the fp32 GEMM without tile quantisation: shapes whose 64 x 64 tiles fill the 1024 persistent slots
exactly (one or two tiles per slot), K swept -- what the k-loop itself delivers (slope) and what a launch costs (intercept)."""
import sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
def bench(akm, bkm, M, N, K, epi=1, beta=0, iters=30):
    A = torch.randn((K, M) if akm else (M, K), device='cuda')
    B = torch.randn((K, N) if bkm else (N, K), device='cuda')
    C = torch.zeros(M, N, device='cuda'); bias = torch.randn(N, device='cuda')
    def run():
        L.check(lib.uniter_gemm_f32_cfg(0, akm, bkm, M, N, K, L.ptr(A), A.shape[1], L.ptr(B), B.shape[1], L.ptr(C), N, epi, L.ptr(bias), None, None, N, beta, L.cur_stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, akm, bkm in (('x @ W^T (forward)', 0, 0), ('dy @ W (input gradient)', 0, 1)):
    for M, N in ((2048, 2048), (4096, 2048), (2624, 3072), (2624, 768)):
        ts = {K: bench(akm, bkm, M, N, K) for K in (768, 3072, 6144)}
        slope = (ts[6144] - ts[3072]) / 3072.0                       # us per unit of K
        print('%-24s M=%d N=%d tiles=%d: ' % (name, M, N, ((M + 63) // 64) * ((N + 63) // 64)) +
              ' | '.join('K%d %.1fus %.1fTF' % (K, t, 2.0 * M * N * K / t / 1e6) for K, t in ts.items()) +
              ' | steady %.1f TF, intercept %.1f us' % (2.0 * M * N / slope / 1e6, ts[3072] - slope * 3072), flush=True)
