"""Lab: why is the LayerNorm backward row pass 2-3x slower inside the training step than alone?
rocprofv3 --pmc serialises kernels, so counters cannot see a neighbour; this times the pass (HIP events on its own stream)
while ANOTHER stream keeps the chip busy with one kind of kernel: the stream-K weight gradient (64x64 and 128x128 tiles), the
whole-K one, an HBM copy (bandwidth only, no matrix work), or nothing."""
import sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
M, H, I = 2624, 768, 3072
x = torch.randn(M, H, device='cuda'); z = torch.randn(M, H, device='cuda'); g = torch.ones(H, device='cuda')
mean = torch.zeros(M, device='cuda'); rstd = torch.ones(M, device='cuda')
dz = torch.empty(M, H, device='cuda'); dx = torch.empty(M, H, device='cuda')
nws = lib.uniter_ln_bwd_ws_bytes(M, H); ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
A = torch.randn(M, I, device='cuda'); Bm = torch.randn(M, H, device='cuda'); Cw = torch.zeros(I, H, device='cuda')
big = torch.empty(64 << 20, device='cuda'); big2 = torch.empty_like(big)
def ln(): L.check(lib.uniter_ln_bwd_rows(L.ptr(x), L.ptr(z), L.ptr(mean), L.ptr(rstd), L.ptr(g), L.ptr(dz), L.ptr(dx), None, 1, M, H, 0.1, 1, 2, 3, L.ptr(ws), nws, L.cur_stream()))
def wgrad(cfg): L.check(lib.uniter_gemm_f32_cfg(cfg, 1, 1, I, H, M, L.ptr(A), I, L.ptr(Bm), H, L.ptr(Cw), H, 0, None, None, None, 0, 1, L.cur_stream()))
side, main = torch.cuda.Stream(), torch.cuda.Stream()
def run(name, neighbour, reps=40):
    for _ in range(3): ln()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        if neighbour is not None:
            for _ in range(reps * 3 + 20): neighbour()
    ts = []
    with torch.cuda.stream(main):
        torch.cuda._sleep(200000)              # let the neighbour get going
        for _ in range(reps):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); ln(); e1.record(); ts.append((e0, e1))
            torch.cuda._sleep(150000)          # the pass starts at a random phase of the neighbour's launches
    torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts)
    print('%-44s ln_bwd median %.1f us  (min %.1f, max %.1f)' % (name, v[len(v) // 2], v[0], v[-1]), flush=True)
run('alone', None)
run('beside stream-K 64x64 weight gradients', lambda: wgrad(24))
run('beside whole-K 64x64 weight gradients', lambda: wgrad(25))
run('beside stream-K 128x128 weight gradients', lambda: wgrad(21))
run('beside a 256 MB HBM copy (no matrix work)', lambda: big2.copy_(big))
