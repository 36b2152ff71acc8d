"""Measuring stick only (not product): what the vendor bf16 GEMM reaches on the model's shapes, next to
uniter_gemm_bf16res.  Usage: python tests/tools/gemm_bf16_ceiling.py"""
import sys, torch
sys.path.insert(0, '.')
M, H, I = 2624, 768, 3072
shapes = [('qkv_fwd', M, 3 * H, H, 'nt'), ('attnout_fwd', M, H, H, 'nt'), ('ffnup_fwd', M, I, H, 'nt'), ('ffndown_fwd', M, H, I, 'nt'),
          ('ffnup_dgrad', M, H, I, 'nn'), ('ffndown_dgrad', M, I, H, 'nn'), ('ffn1_wgrad', I, H, M, 'tn'), ('qkv_wgrad', 3 * H, H, M, 'tn')]
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, m, n, k, kind in shapes:
    if kind == 'nt':
        a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(n, k, device='cuda').bfloat16(); f = lambda: torch.mm(a, b.t())
    elif kind == 'nn':
        a = torch.randn(m, k, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16(); f = lambda: torch.mm(a, b)
    else:
        a = torch.randn(k, m, device='cuda').bfloat16(); b = torch.randn(k, n, device='cuda').bfloat16(); f = lambda: torch.mm(a.t(), b)
    ms = timeit(f)
    print('%-14s %5dx%5dx%5d %s  vendor bf16->bf16 %.4f ms %6.0f TF' % (name, m, n, k, kind, ms, 2.0 * m * n * k / ms / 1e9))
