"""In-kernel clock of the x3 GEMM (measurement switch 16; needs the -DUNITER_X3_LAB variant build, see gemm_x3_lab.py): shader cycles / real time over each workgroup's lifetime, after a
warm-up of back-to-back launches on random data."""
import os, sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
M, N, K = 2624, 3072, int(os.environ.get('LAB_K', '1536'))
A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda') * 0.05
def split3(x):
    o = torch.empty(x.shape[0], 3, x.shape[1], dtype=torch.bfloat16, device='cuda')
    L.check(lib.uniter_split3(L.ptr(x), x.shape[0], x.shape[1], x.shape[1], L.ptr(o), 3 * x.shape[1], x.shape[1], L.cur_stream())); return o
A3, B3 = split3(A), split3(B)
C = torch.empty(M, N, device='cuda')
for tok in os.environ.get('LAB_CFGS', '1,1d3,1d4').split(','):
    base, _, d = tok.partition('d')
    cfg = int(base) | ((int(d or 0) | 16) << 8)
    buf = torch.zeros(256 * 4, dtype=torch.int64, device='cuda')
    def run():
        L.check(lib.uniter_gemm_x3_cfg(cfg, 1, 0, 0, M, N, K, L.ptr(A3), 3 * K, K, L.ptr(B3), 3 * K, K, L.ptr(C), N, M * N, None, 3 * N, N, 0, L.ptr(buf), None, None, N, L.cur_stream()))
    for _ in range(300): run()
    torch.cuda.synchronize()
    b = buf.cpu().view(256, 4)
    cyc = (b[:, 1] - b[:, 0]).double(); rt = (b[:, 3] - b[:, 2]).double()
    ok = rt > 0
    ghz = (cyc[ok] / rt[ok] * 0.1)
    st, en = b[:, 2].double(), b[:, 3].double()
    t0 = st[ok].min()
    conc = [int(((st <= t) & (en > t) & ok).sum()) for t in torch.linspace(float(t0), float(en[ok].max()), 9, dtype=torch.float64)[1:-1]]
    print('         launch span %.1f us; workgroups running at 7 instants across it: %s; starts within %.1f us' % (
        (en[ok].max() - t0).item() / 100, conc, (st[ok].max() - t0).item() / 100))
    print('%-8s workgroup lifetime %.1f us (median), %.0f cycles, clock %.2f GHz (min %.2f max %.2f)' % (
        tok, rt[ok].median().item() / 100, cyc[ok].median().item(), ghz.median().item(), ghz.min().item(), ghz.max().item()))
