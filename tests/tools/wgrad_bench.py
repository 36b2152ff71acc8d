"""Times the weight-gradient GEMMs of one UNITER-base layer (C += A^T B, K = tokens)."""
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
K = 2624
for name, M, N in [('dW2', 768, 3072), ('dW1', 3072, 768), ('dWo', 768, 768), ('dWqkv', 2304, 768), ('dWimg', 768, 2048)]:
    k = 576 if name == 'dWimg' else K
    A = torch.randn(k, M, device='cuda'); B = torch.randn(k, N, device='cuda'); C = torch.zeros(M, N, device='cuda')
    row = []
    for cfg in (24, 21):
        ms = timeit(lambda: L.check(lib.uniter_gemm_f32_cfg(cfg, 1, 1, M, N, k, L.ptr(A), M, L.ptr(B), N, L.ptr(C), N, 0, None, None, None, 0, 1, L.cur_stream())))
        row.append('cfg%d %.4fms %.0fTF' % (cfg, ms, 2.0 * M * N * k / ms / 1e9))
    print(name, ' | '.join(row), flush=True)
