"""Lab: the fp32-accurate six-product bf16 GEMM (uniter_gemm_x3_cfg) on the model's shapes -- accuracy against float64
next to the native fp32 MFMA kernel (uniter_gemm_f32_cfg), and time per launch per tile geometry next to that kernel.
Interleaved rounds in one process; median over rounds; hipGraph replay of ITERS back-to-back launches.

    python tests/tools/gemm_x3_lab.py            # accuracy + timing
    LAB_ACC=0 / LAB_TIME=0 skip a part; LAB_ONLY=name,name; LAB_CFGS=1,2,3; LAB_KSWEEP=32,768,1536 (fixed cost and slope)
    measurement switches (cfg tokens 1d3 ..) need a variant build of the library:
        UNITER_EXTRA_HIPCC_FLAGS=-DUNITER_X3_LAB python -c "from meme_challenge_amd import build; build.build(variant='x3lab')"
        UNITER_LIB_VARIANT=x3lab LAB_CFGS=1,1d1,1d3,1d4 python tests/tools/gemm_x3_lab.py
"""
import math, os, sys, statistics, ctypes
import torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
MM = int(os.environ.get('LAB_M', '2624')); HH = int(os.environ.get('LAB_H', '768')); II = 4 * HH
ROUNDS = int(os.environ.get('LAB_ROUNDS', '5')); ITERS = int(os.environ.get('LAB_ITERS', '20'))
def _cfg(tok):      # '1' or '1d3' = cfg 1 with measurement switches 3 (1 = no LDS-DMA, 2 = no LDS reads, 4 = no MFMAs)
    base, _, dbg = tok.partition('d')
    return int(base) | (int(dbg or 0) << 8)
CFG_TOKS = os.environ.get('LAB_CFGS', '1,2,3').split(',')
CFGS = [_cfg(t) for t in CFG_TOKS]
NSPLITS = [int(c) for c in os.environ.get('LAB_NSPLIT', '1,2,4').split(',')]
only = os.environ.get('LAB_ONLY')


def split3(x):
    """fp32 [rows, cols] cuda -> x3 [rows, 3, cols] bf16 by the library's kernel"""
    rows, cols = x.shape
    o = torch.empty(rows, 3, cols, dtype=torch.bfloat16, device='cuda')
    L.check(lib.uniter_split3(L.ptr(x), rows, cols, cols, L.ptr(o), 3 * cols, cols, L.cur_stream()), 'split3')
    return o


def x3_gemm(cfg, ns, akm, bkm, M, N, K, A3, B3, C, Cx, epi, bias, aux_in, aux_out):
    return lib.uniter_gemm_x3_cfg(cfg, ns, akm, bkm, M, N, K, L.ptr(A3), 3 * A3.shape[2], A3.shape[2], L.ptr(B3), 3 * B3.shape[2], B3.shape[2],
                                  L.ptr(C), N, M * N, L.ptr(Cx), 3 * N, N, epi, L.ptr(bias), L.ptr(aux_in), L.ptr(aux_out), N, L.cur_stream())


def f32_gemm(akm, bkm, M, N, K, A, B, C, epi, bias, aux_in, aux_out, beta=0):
    return lib.uniter_gemm_f32_cfg(0, akm, bkm, M, N, K, L.ptr(A), A.shape[1], L.ptr(B), B.shape[1], L.ptr(C), N, epi, L.ptr(bias),
                                   L.ptr(aux_in), L.ptr(aux_out), N, beta, L.cur_stream())


def errs(got, ref):
    d = (got.double().cpu() - ref).abs()
    return d.max().item(), d.pow(2).mean().sqrt().item()


# name, akm, bkm, M, N, K, epi
shapes = [('plain_fwd', 0, 0, MM, II, HH, 0), ('qkv_fwd', 0, 0, MM, 3 * HH, HH, 1), ('attnout_fwd', 0, 0, MM, HH, HH, 1), ('ffnup_fwd', 0, 0, MM, II, HH, 5),
          ('ffndown_fwd', 0, 0, MM, HH, II, 1), ('ffndown_dgrad', 0, 1, MM, II, HH, 6), ('ffnup_dgrad', 0, 1, MM, HH, II, 4),
          ('attnout_dgrad', 0, 1, MM, HH, HH, 0), ('qkv_dgrad', 0, 1, MM, HH, 3 * HH, 4),
          ('w2_wgrad', 1, 1, HH, II, MM, 0), ('w1_wgrad', 1, 1, II, HH, MM, 0), ('wo_wgrad', 1, 1, HH, HH, MM, 0),
          ('wqkv_wgrad', 1, 1, 3 * HH, HH, MM, 0)]


if os.environ.get('LAB_KSWEEP'):       # K values for every shape of LAB_ONLY
    ks = [int(k) for k in os.environ['LAB_KSWEEP'].split(',')]
    shapes = [(sh[0] + '_K%d' % k,) + sh[1:5] + (k, sh[6]) for sh in shapes if only and sh[0] in only.split(',') for k in ks]
    only = None


def operands(akm, bkm, M, N, K, scale_a=1.0, scale_b=1.0, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    A = torch.randn((K, M) if akm else (M, K), device='cuda', generator=g) * scale_a
    B = torch.randn((K, N) if bkm else (N, K), device='cuda', generator=g) * scale_b
    return A, B


def ref64(akm, bkm, A, B):
    a = A.double().cpu(); b = B.double().cpu()
    a = a.t() if akm else a
    b = b if bkm else b.t()
    return a @ b


if os.environ.get('LAB_ACC', '1') == '1':
    print('accuracy against float64 (max abs / rms abs error; the x3 kernel on the library\'s pieces of the SAME fp32 operands)')
    # the round trip itself
    x = torch.randn(300, 512, device='cuda') * torch.exp(torch.randn(300, 512, device='cuda') * 4)
    x3 = split3(x)
    back = torch.empty_like(x)
    L.check(lib.uniter_join3(L.ptr(x3), 300, 512, 3 * 512, 512, L.ptr(back), 512, L.cur_stream()), 'join3')
    torch.cuda.synchronize()
    nz = x != 0
    print('split3 -> join3 round trip: bit-identical %s; max |diff| / |x| = %.3g (0 = exact; %d zeros, %d non-finite values back)'
          % (torch.equal(back, x), ((back - x).abs()[nz] / x.abs()[nz]).max().item(), int((~nz).sum()), int((~torch.isfinite(back)).sum())))
    p64 = x3.double().sum(1)
    print('   float64 sum of the pieces vs x: max rel %.3g' % ((p64 - x.double()).abs()[nz] / x.double().abs()[nz]).max().item())
    for name, akm, bkm, M, N, K, epi in shapes:
        if only and name not in only.split(','): continue
        A, B = operands(akm, bkm, M, N, K, 1.0, 0.05)
        ref = ref64(akm, bkm, A, B)
        A3, B3 = split3(A), split3(B)
        C = torch.empty(M, N, device='cuda'); C32 = torch.empty(M, N, device='cuda')
        L.check(f32_gemm(akm, bkm, M, N, K, A, B, C32, 0, None, None, None), 'gemm_f32')
        e32 = errs(C32, ref)
        line = '%-14s %5dx%5dx%5d  native fp32 %.3g / %.3g ' % (name, M, N, K, e32[0], e32[1])
        for tok, cfg in zip(CFG_TOKS, CFGS):
            if cfg >> 8: continue
            C.fill_(float('nan'))
            rc = x3_gemm(cfg, 1, akm, bkm, M, N, K, A3, B3, C, None, 0, None, None, None)
            if rc != 0:
                line += ' c%s: %s' % (tok, lib.uniter_last_error().decode()[:40]); continue
            torch.cuda.synchronize()
            e = errs(C, ref)
            line += ' c%s %.3g / %.3g (x%.2f / x%.2f)' % (tok, e[0], e[1], e[0] / e32[0], e[1] / e32[1])
        print(line, flush=True)

if os.environ.get('LAB_TIME', '1') == '1':
    def make_graph(run):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for _ in range(ITERS): run()
        return g

    def timeit(g):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / ITERS

    print('time per launch (us) and fp32-equivalent TFLOP/s; x3 outputs: fp32, or x3 where the model wants pieces (ffnup_fwd, ffndown_dgrad)')
    for name, akm, bkm, M, N, K, epi in shapes:
        if only and name not in only.split(','): continue
        A, B = operands(akm, bkm, M, N, K, 1.0, 0.05)
        A3, B3 = split3(A), split3(B)
        bias = torch.randn(N, device='cuda'); aux = torch.randn(M, N, device='cuda'); auxo = torch.empty(M, N, device='cuda')
        C = torch.empty(4, M, N, device='cuda'); C32 = torch.empty(M, N, device='cuda'); Cx = torch.empty(M, 3, N, dtype=torch.bfloat16, device='cuda')
        want_x3 = name in ('ffnup_fwd', 'ffndown_dgrad')
        runs = {}
        if akm:
            C32.zero_()
            runs['f32'] = lambda: L.check(lib.uniter_gemm_f32_cfg(25, 1, 1, M, N, K, L.ptr(A), M, L.ptr(B), N, L.ptr(C32), N, 0, None, None, None, 0, 0, L.cur_stream()))
        else:
            runs['f32'] = lambda: L.check(f32_gemm(akm, bkm, M, N, K, A, B, C32, epi, bias, aux, auxo))
        for tok, cfg in zip(CFG_TOKS, CFGS):
            for ns in (NSPLITS if (N <= 1024 and not akm) else (1,)):
                runs['c%s%s' % (tok, 's%d' % ns if ns > 1 else '')] = (
                    lambda cfg=cfg, ns=ns: L.check(x3_gemm(cfg, ns, akm, bkm, M, N, K, A3, B3, None if want_x3 else C, Cx if want_x3 else None,
                                                           epi, bias, aux, auxo)))
        ok = {}
        for k, r in runs.items():
            try:
                r(); torch.cuda.synchronize(); ok[k] = r
            except Exception as e:
                print('  %s %s: %s' % (name, k, str(e)[:100]))
        graphs = {k: make_graph(r) for k, r in ok.items()}
        res = {k: [] for k in ok}
        for _ in range(ROUNDS):
            for k in ok: res[k].append(timeit(graphs[k]))
        fl = 2.0 * M * N * K
        print('%-14s %5dx%5dx%5d  ' % (name, M, N, K) + '  '.join('%s %.1f/%.0f' % (k, statistics.median(v) * 1e3, fl / statistics.median(v) / 1e9) for k, v in res.items()), flush=True)

    # the layer's four weight gradients as one launch
    if not only or 'wgrad_group' in only:
        shp = [(II, HH), (HH, II), (3 * HH, HH), (HH, HH)]
        As = [split3(torch.randn(MM, m, device='cuda')) for m, n in shp]
        Bs = [split3(torch.randn(MM, n, device='cuda')) for m, n in shp]
        Cs = [torch.zeros(m, n, device='cuda') for m, n in shp]
        IA = ctypes.c_int * 4; PA = ctypes.c_void_p * 4
        fl = sum(2.0 * m * n * MM for m, n in shp)
        for cfg in [int(c) for c in os.environ.get('LAB_WG_CFGS', '1,2,3').split(',')]:
            for wgs in (0, 256):
                run = lambda cfg=cfg, wgs=wgs: L.check(lib.uniter_wgrad_x3_group(cfg, 4, IA(*[m for m, n in shp]), IA(*[n for m, n in shp]), MM,
                                                       PA(*[a.data_ptr() for a in As]), PA(*[b.data_ptr() for b in Bs]),
                                                       PA(*[c.data_ptr() for c in Cs]), 1, wgs, L.cur_stream()))
                run(); torch.cuda.synchronize()
                g = make_graph(run)
                ts = [timeit(g) for _ in range(ROUNDS)]
                print('wgrad_group cfg %d max_wgs %d: %.1f us  %.0f TF' % (cfg, wgs, statistics.median(ts) * 1e3, fl / statistics.median(ts) / 1e9), flush=True)
