"""How long plain device writes of the GEMM output sizes take (fill kernels), as a floor for the epilogue."""
import torch
def timeit(run, iters=50):
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in (4, 8, 12, 24, 36, 48, 96, 256):
    x = torch.empty(mb * 1024 * 1024 // 4, device='cuda')
    y = torch.empty_like(x)
    t = timeit(lambda: x.fill_(1.0))
    t2 = timeit(lambda: y.copy_(x))
    print('%4d MB fill %.1f us %.2f TB/s | copy %.1f us %.2f TB/s (r+w)' % (mb, t * 1e3, mb * 1.048576 / t / 1e3, t2 * 1e3, 2 * mb * 1.048576 / t2 / 1e3))
