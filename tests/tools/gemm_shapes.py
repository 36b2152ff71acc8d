"""fp32 GEMM rates on every UNITER-base (B=16, L=164) shape, as the model calls them (cfg 0 = auto)."""
import sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
def bench(cfg, akm, bkm, M, N, K, epi, beta, iters=30):
    A = torch.randn((K, M) if akm else (M, K), device='cuda')
    B = torch.randn((K, N) if bkm else (N, K), device='cuda')
    C = torch.zeros(M, N, device='cuda'); bias = torch.randn(N, device='cuda'); aux = torch.randn(M, N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
    def run():
        L.check(lib.uniter_gemm_f32_cfg(cfg, akm, bkm, M, N, K, L.ptr(A), A.shape[1], L.ptr(B), B.shape[1], L.ptr(C), N, epi, L.ptr(bias), L.ptr(aux), L.ptr(auxo), N, beta, L.cur_stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * M * N * K / ms / 1e9
cfgs = [int(c) for c in sys.argv[1:]] or [0]
shapes = [('qkv_fwd', 0, 0, 2624, 2304, 768, 1, 0), ('attnout_fwd', 0, 0, 2624, 768, 768, 1, 0), ('ffnup_fwd', 0, 0, 2624, 3072, 768, 2, 0), ('ffndown_fwd', 0, 0, 2624, 768, 3072, 1, 0),
          ('ffndown_dgrad', 0, 1, 2624, 3072, 768, 3, 0), ('ffnup_dgrad', 0, 1, 2624, 768, 3072, 4, 0), ('attnout_dgrad', 0, 1, 2624, 768, 768, 0, 0), ('qkv_dgrad', 0, 1, 2624, 768, 2304, 4, 0),
          ('ffn1_wgrad', 1, 1, 3072, 768, 2624, 0, 1), ('ffn2_wgrad', 1, 1, 768, 3072, 2624, 0, 1), ('qkv_wgrad', 1, 1, 2304, 768, 2624, 0, 1), ('o_wgrad', 1, 1, 768, 768, 2624, 0, 1),
          ('img_fwd', 0, 0, 576, 768, 2048, 1, 0), ('img_wgrad', 1, 1, 768, 2048, 576, 0, 1)]
tot_ms = tot_gf = 0.0
for name, akm, bkm, M, N, K, epi, beta in shapes:
    row = []
    for cfg in cfgs:
        ms, tf = bench(cfg, akm, bkm, M, N, K, epi, beta)
        row.append('cfg%d %.4fms %5.1fTF' % (cfg, ms, tf))
    if not name.startswith('img'):
        tot_ms += ms; tot_gf += 2.0 * M * N * K / 1e9
    print('%-14s M%4d N%4d K%4d | ' % (name, M, N, K) + ' | '.join(row), flush=True)
print('layer total (last cfg): %.3f ms, %.1f GF -> %.1f TF; x12 = %.2f ms' % (tot_ms, tot_gf, tot_gf / tot_ms, 12 * tot_ms))
