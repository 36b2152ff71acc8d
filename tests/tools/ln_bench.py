"""Lab: LayerNorm row passes alone (graph replay), M x H fp32."""
import os, sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
M, H = int(os.environ.get('LAB_M', '2624')), int(os.environ.get('LAB_H', '768'))
x = torch.randn(M, H, device='cuda'); r = torch.randn(M, H, device='cuda'); g = torch.ones(H, device='cuda'); b = torch.zeros(H, device='cuda')
z = torch.empty(M, H, device='cuda'); y = torch.empty(M, H, device='cuda'); yb = torch.empty(M, H, dtype=torch.bfloat16, device='cuda')
mean = torch.empty(M, device='cuda'); rstd = torch.empty(M, device='cuda')
dz = torch.empty(M, H, device='cuda'); dx = torch.empty(M, H, device='cuda'); dxb = torch.empty(M, H, dtype=torch.bfloat16, device='cuda')
nws = lib.uniter_ln_bwd_ws_bytes(M, H); ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
def fwd(p): L.check(lib.uniter_ln_fwd_b16(L.ptr(x), L.ptr(r), L.ptr(g), L.ptr(b), L.ptr(z), L.ptr(y), L.ptr(yb), L.ptr(mean), L.ptr(rstd), M, H, p, 1, 2, 3, L.cur_stream()))
def bwd(p): L.check(lib.uniter_ln_bwd_rows(L.ptr(x), L.ptr(z), L.ptr(mean), L.ptr(rstd), L.ptr(g), L.ptr(dz), L.ptr(dx), L.ptr(dxb), 1, M, H, p, 1, 2, 3, L.ptr(ws), nws, L.cur_stream()))
def timeit(run, iters=30):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        run(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for _ in range(iters): run()
    res = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(res)[2]
fwd(0.1)
print('waves=%s rows=%s  M=%d H=%d:' % (os.environ.get('UNITER_LNB_WAVES', '4'), os.environ.get('UNITER_LNB_ROWS', '2'), M, H),
      'fwd p=0 %.1f us  p=.1 %.1f us | bwd p=0 %.1f us  p=.1 %.1f us' % (timeit(lambda: fwd(0.0)), timeit(lambda: fwd(0.1)), timeit(lambda: bwd(0.0)), timeit(lambda: bwd(0.1))))
