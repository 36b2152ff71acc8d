"""Repeated-launch screen of the GEMM epilogues at model size (fp32 kernel, bf16 convert-in-flight, bf16 resident;
128x128 and 64x64 tiles; epilogues BIAS, ADD, GELU_D, MUL) against float64 -- counts elements off by more than the
arithmetic's tolerance in C, the bf16 copy and the aux output.  Written after a packed-math sequence dropped a term on
a few lanes per launch in one kernel variant only; expected output: all zeros."""
import sys, math, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
M, N, K = 2624, 3072, 768
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 6
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g); Bnt = torch.randn(N, K, generator=g); bias = torch.randn(N, generator=g); aux = torch.randn(M, N, generator=g)
def refs(Ar, Br):
    pre = Ar.double() @ Br.double().t()
    pb = pre + bias.double()
    cdf = 0.5 * (1 + torch.erf(pb / math.sqrt(2)))
    return {1: (pb, None), 4: (pre + aux.double(), None), 6: (pre * aux.double(), None),
            5: (pb * cdf, cdf + pb * torch.exp(-0.5 * pb * pb) / math.sqrt(2 * math.pi))}
r32, r16 = refs(A, Bnt), refs(A.bfloat16(), Bnt.bfloat16())
dA, dB, db, daux = A.cuda(), Bnt.cuda(), bias.cuda(), aux.cuda()
dAb, dBb = dA.bfloat16(), dB.bfloat16()
total = 0
for kind, cfgs in (('fp32', (21, 24)), ('hybrid', (1, 4)), ('resident', (1, 4))):
    for cfg in cfgs:
        for epi in (1, 4, 5, 6):
            bad = [0, 0, 0]
            for _ in range(REP):
                C = torch.full((M, N), 7.0, device='cuda'); Cb = torch.full((M, N), 7.0, dtype=torch.bfloat16, device='cuda'); X = torch.full((M, N), 7.0, device='cuda')
                if kind == 'fp32':
                    L.check(lib.uniter_gemm_f32_cfg(cfg, 0, 0, M, N, K, L.ptr(dA), K, L.ptr(dB), K, L.ptr(C), N, epi, L.ptr(db), L.ptr(daux), L.ptr(X), N, 0, L.cur_stream()))
                elif kind == 'hybrid':
                    L.check(lib.uniter_gemm_bf16_cfg(cfg, 0, 0, M, N, K, L.ptr(dA), K, L.ptr(dB), K, L.ptr(C), N, epi, L.ptr(db), L.ptr(daux), L.ptr(X), N, 0, L.cur_stream()))
                else:
                    L.check(lib.uniter_gemm_bf16res_cfg(cfg, 0, 0, M, N, K, L.ptr(dAb), K, L.ptr(dBb), K, L.ptr(C), N, L.ptr(Cb), N, epi, L.ptr(db), L.ptr(daux), L.ptr(X), N, 0, L.cur_stream()))
                torch.cuda.synchronize()
                ref, refx = (r32 if kind == 'fp32' else r16)[epi]
                tol = 1e-2 * (1 + ref.abs())
                bad[0] += ((C.cpu().double() - ref).abs() > tol).sum().item()
                if kind == 'resident':
                    bad[1] += ((Cb.float().cpu().double() - ref).abs() > 2 * tol).sum().item()
                if refx is not None:
                    bad[2] += ((X.cpu().double() - refx).abs() > 1e-2).sum().item()
            total += sum(bad)
            print('%-8s cfg%-2d epi %d  bad C / bf16 copy / aux over %d launches: %s' % (kind, cfg, epi, REP, bad), flush=True)
print('TOTAL BAD', total)
