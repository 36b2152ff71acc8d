import torch, time, sys
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
def bench(cfg, akm, bkm, M, N, K, epi=0, beta=0, iters=20):
    A = torch.randn((K, M) if akm else (M, K), device='cuda')
    B = torch.randn((K, N) if bkm else (N, K), device='cuda')
    Cc = torch.zeros(M, N, device='cuda'); bias = torch.randn(N, device='cuda'); aux = torch.randn(M, N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
    def run():
        L.check(lib.uniter_gemm_f32_cfg(cfg, akm, bkm, M, N, K, L.ptr(A), A.shape[1], L.ptr(B), B.shape[1], L.ptr(Cc), N, epi, L.ptr(bias), L.ptr(aux), L.ptr(auxo), N, beta, L.cur_stream()))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    return ms, tf
print(lib.uniter_build_info().decode())
shapes = [('qkv_fwd', 0, 0, 2624, 2304, 768, 1), ('attnout_fwd', 0, 0, 2624, 768, 768, 1), ('ffnup_fwd', 0, 0, 2624, 3072, 768, 2), ('ffndown_fwd', 0, 0, 2624, 768, 3072, 1),
          ('ffndown_dgrad', 0, 1, 2624, 3072, 768, 3), ('ffnup_dgrad', 0, 1, 2624, 768, 3072, 4), ('qkv_dgrad', 0, 1, 2624, 768, 2304, 4),
          ('ffn_wgrad', 1, 1, 3072, 768, 2624, 0), ('ffn2_wgrad', 1, 1, 768, 3072, 2624, 0), ('qkv_wgrad', 1, 1, 2304, 768, 2624, 0), ('o_wgrad', 1, 1, 768, 768, 2624, 0),
          ('sq4096', 0, 0, 4096, 4096, 4096, 0)]
for name, akm, bkm, M, N, K, epi in shapes:
    row = []
    for cfg in (1, 4, 21, 24):
        ms, tf = bench(cfg, akm, bkm, M, N, K, epi)
        row.append('cfg%d %.3fms %.1fTF' % (cfg, ms, tf))
    print('%-14s M%d N%d K%d | ' % (name, M, N, K) + ' | '.join(row), flush=True)
print('--- K sweep (M=2624, N=3072, NT, bias epilogue) ---')
for cfg in (1, 21, 24):
    for K in (256, 768, 1536, 3072, 6144):
        ms, tf = bench(cfg, 0, 0, 2624, 3072, K, 1)
        print('cfg%d K=%5d %.4f ms %.1f TF' % (cfg, K, ms, tf), flush=True)
print('--- epilogue cost at K=768 ---')
for epi in (0, 1, 2):
    for cfg in (1, 4):
        ms, tf = bench(cfg, 0, 0, 2624, 3072, 768, epi)
        print('epi%d cfg%d %.4f ms %.1f TF' % (epi, cfg, ms, tf), flush=True)
