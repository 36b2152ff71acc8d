"""Measuring stick only (not product): what the vendor bf16 GEMM (torch.mm -> hipBLASLt) reaches on the model's
shapes, bf16 in / bf16 out, no epilogue.  Launches are replayed from a graph so that the host launch rate (~19 us per
torch.mm call) does not floor the figure.  Usage: python tests/tools/gemm_bf16_ceiling.py [fp32]"""
import sys
import torch
DT = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'fp32') else torch.bfloat16
M, H, I = 2624, 768, 3072
shapes = [('qkv_fwd', M, 3 * H, H, 'nt'), ('attnout_fwd', M, H, H, 'nt'), ('ffnup_fwd', M, I, H, 'nt'), ('ffndown_fwd', M, H, I, 'nt'),
          ('ffndown_dgrad', M, I, H, 'nn'), ('ffnup_dgrad', M, H, I, 'nn'), ('qkv_dgrad', M, H, 3 * H, 'nn'),
          ('ffn1_wgrad', I, H, M, 'tn'), ('ffn2_wgrad', H, I, M, 'tn'), ('qkv_wgrad', 3 * H, H, M, 'tn')]
REP = 20
for name, m, n, k, kind in shapes:
    if kind == 'nt':
        a = torch.randn(m, k, device='cuda').to(DT); b = torch.randn(n, k, device='cuda').to(DT); f = lambda: torch.mm(a, b.t())
    elif kind == 'nn':
        a = torch.randn(m, k, device='cuda').to(DT); b = torch.randn(k, n, device='cuda').to(DT); f = lambda: torch.mm(a, b)
    else:
        a = torch.randn(k, m, device='cuda').to(DT); b = torch.randn(k, n, device='cuda').to(DT); f = lambda: torch.mm(a.t(), b)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP): f()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (5 * REP)
    print('%-14s %5dx%5dx%5d %s  vendor %s %.4f ms %6.0f TF' % (name, m, n, k, kind, 'fp32' if DT == torch.float32 else 'bf16->bf16', ms, 2.0 * m * n * k / ms / 1e9), flush=True)
