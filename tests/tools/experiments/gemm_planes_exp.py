import sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
def planes(x):
    r, c = x.shape
    p = torch.empty(3, r, c, dtype=torch.bfloat16, device='cuda')
    L.check(lib.uniter_split_planes(L.ptr(x), r, c, c, L.ptr(p), c, r * c, 0, L.cur_stream()))
    return p
def timeit(run, iters=20):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, M, N, K, epi in [('qkv', 2624, 2304, 768, 1), ('attno', 2624, 768, 768, 1), ('ffnup', 2624, 3072, 768, 2), ('ffndn', 2624, 768, 3072, 1),
                           ('dgrad1', 2624, 3072, 768, 3), ('dgrad2', 2624, 768, 3072, 4), ('sq4096', 4096, 4096, 4096, 0)]:
    A = torch.randn(M, K, device='cuda'); W = torch.randn(N, K, device='cuda'); C = torch.zeros(M, N, device='cuda')
    bias = torch.randn(N, device='cuda'); aux = torch.randn(M, N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
    Ap, Wp = planes(A), planes(W)
    row = []
    for cfg in (1, 2, 3, 4):
        ms = timeit(lambda: L.check(lib.uniter_gemm_planes_cfg(cfg, M, N, K, L.ptr(Ap), K, M * K, L.ptr(Wp), K, N * K, L.ptr(C), N, epi, L.ptr(bias), L.ptr(aux), L.ptr(auxo), N, 0, L.cur_stream())))
        row.append('cfg%d %.4fms %.0fTF' % (cfg, ms, 2.0 * M * N * K / ms / 1e9))
    sp = timeit(lambda: L.check(lib.uniter_split_planes(L.ptr(A), M, K, K, L.ptr(Ap), K, M * K, 0, L.cur_stream())))
    print(name, ' | '.join(row), '| split(A) %.4fms' % sp, flush=True)
