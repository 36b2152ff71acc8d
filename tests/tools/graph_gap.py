"""How much of a GEMM launch's fixed cost is dispatch gap?  30 dependent launches, plain vs hipGraph replay."""
import sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
def run_plain(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name, M, N, K, epi in [('ffnup', 2624, 3072, 768, 5), ('attnout', 2624, 768, 768, 1), ('tiny', 64, 64, 64, 0)]:
    A = torch.randn(M, K, device='cuda'); B = torch.randn(N, K, device='cuda'); C = torch.zeros(M, N, device='cuda')
    bias = torch.randn(N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
    def chain():
        for _ in range(30):
            L.check(lib.uniter_gemm_f32(0, 0, M, N, K, L.ptr(A), K, L.ptr(B), K, L.ptr(C), N, epi, L.ptr(bias), None, L.ptr(auxo), N, 0, L.cur_stream()))
    t_plain = run_plain(chain) / 30
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain()
        t_graph = run_plain(g.replay) / 30
    print('%-8s plain %.2f us/launch   graph %.2f us/launch' % (name, t_plain * 1e3, t_graph * 1e3), flush=True)
