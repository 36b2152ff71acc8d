"""Lab timing: second-generation bf16 GEMM (uniter_gemm_bf16v2_cfg) against the first-generation resident
kernel and the vendor library on the model's shapes.  Interleaved rounds in one process; median over rounds."""
import os, sys, statistics, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
MM = int(os.environ.get('LAB_M', '2624')); HH = int(os.environ.get('LAB_H', '768')); II = 4 * HH
ROUNDS = int(os.environ.get('LAB_ROUNDS', '7')); ITERS = int(os.environ.get('LAB_ITERS', '30'))
# name, bkm, M, N, K, epi, fp32 out?, bf16 out?, aux bf16 flags (in, out)
shapes = [('qkv_fwd', 0, MM, 3 * HH, HH, 1, 0, 1, 0, 0), ('attnout_fwd', 0, MM, HH, HH, 1, 1, 0, 0, 0),
          ('ffnup_fwd', 0, MM, II, HH, 5, 0, 1, 0, 1), ('ffndown_fwd', 0, MM, HH, II, 1, 1, 0, 0, 0),
          ('ffndown_dgrad', 1, MM, II, HH, 6, 0, 1, 1, 0), ('ffnup_dgrad', 1, MM, HH, II, 4, 1, 0, 0, 0),
          ('attnout_dgrad', 1, MM, HH, HH, 0, 1, 0, 0, 0), ('qkv_dgrad', 1, MM, HH, 3 * HH, 4, 1, 0, 0, 0)]
only = os.environ.get('LAB_ONLY')
if os.environ.get('LAB_EPI'):      # override the epilogue of every shape (fixed-cost ablations)
    shapes = [sh[:5] + (int(os.environ['LAB_EPI']),) + sh[6:] for sh in shapes]
GRAPH = os.environ.get('LAB_GRAPH', '1') == '1'
def make_graph(run):
    """ITERS back-to-back launches captured into one hipGraph: the replay is not bound by the host's launch rate"""
    if not GRAPH: return None
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(ITERS): run()
    return g
def timeit(run, g):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    if g is not None: g.replay()
    else:
        for _ in range(ITERS): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS
variants = [('v1', None, 1)] + [('v2c%d' % c, c, 1) for c in (1, 2, 3, 4)] + [('v2c%ds2' % c, c, 2) for c in (1, 2, 4)] + [('v2c%ds4' % c, c, 4) for c in (1, 2)]
if os.environ.get('LAB_VARIANTS'):     # e.g. v1,c1,c1s2,c1+256 (cfg 1 | dbg 1: stores dropped),c1+512 (k-loop skipped)
    variants = []
    for tok in os.environ['LAB_VARIANTS'].split(','):
        if tok == 'v1': variants.append(('v1', None, 1)); continue
        body, _, dbg = tok.partition('+')
        cfgs, _, ns = body[1:].partition('s')
        variants.append(('v2' + tok, int(cfgs) | int(dbg or 0), int(ns or 1)))
if os.environ.get('LAB_KSWEEP'):       # K values for the first shape of LAB_ONLY
    ks = [int(k) for k in os.environ['LAB_KSWEEP'].split(',')]
    base = [sh for sh in shapes if not only or sh[0] in only.split(',')][0]
    shapes = [(base[0] + '_K%d' % k,) + base[1:4] + (k,) + base[5:] for k in ks]
    only = None
for name, bkm, M, N, K, epi, wc, wcb, abi, abo in shapes:
    if only and name not in only.split(','): continue
    A = torch.randn(M, K, device='cuda').bfloat16(); B = torch.randn((K, N) if bkm else (N, K), device='cuda').bfloat16()
    C = torch.zeros(4, M, N, device='cuda'); Cb = torch.zeros(M, N, dtype=torch.bfloat16, device='cuda')
    bias = torch.randn(N, device='cuda'); aux32 = torch.randn(M, N, device='cuda'); aux16 = aux32.bfloat16()
    auxo32 = torch.empty(M, N, device='cuda'); auxo16 = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    runs = {}
    for vn, cfg, ns in variants:
        if cfg is None:
            args = (0, 0, bkm, M, N, K, L.ptr(A), K, L.ptr(B), B.shape[1], L.ptr(C) if wc else None, N, L.ptr(Cb) if wcb else None, N, epi,
                    L.ptr(bias), L.ptr(aux32), L.ptr(auxo32), N, 0)
            runs[vn] = (lambda a=args: lib.uniter_gemm_bf16res_cfg(*a, L.cur_stream()))
        else:
            if ns > 1 and (wcb or N > 1024): continue
            if (cfg & 0xff) == 3 and N >= 3072: continue
            args = (cfg, ns, 0, bkm, M, N, K, L.ptr(A), K, L.ptr(B), B.shape[1], L.ptr(C) if wc else None, N, M * N, L.ptr(Cb) if wcb else None, N,
                    epi, L.ptr(bias), L.ptr(aux16 if abi else aux32), abi, L.ptr(auxo16 if abo else auxo32), abo, N, 0)
            runs[vn] = (lambda a=args: L.check(lib.uniter_gemm_bf16v2_cfg(*a, L.cur_stream())))
    Bt = B if bkm else B.t().contiguous()
    out_v = torch.empty(M, N, dtype=torch.bfloat16, device='cuda')
    runs['vendor'] = (lambda: torch.mm(A, Bt, out=out_v)) if bkm else (lambda: torch.mm(A, B.t(), out=out_v))
    for r in runs.values(): r()
    torch.cuda.synchronize()
    graphs = {k: make_graph(r) for k, r in runs.items()}
    res = {k: [] for k in runs}
    for _ in range(ROUNDS):
        for k, r in runs.items(): res[k].append(timeit(r, graphs[k]))
    fl = 2.0 * M * N * K
    print('%-14s %5dx%5dx%5d  ' % (name, M, N, K) + '  '.join('%s %.1fus %4.0fTF' % (k, statistics.median(v) * 1e3, fl / statistics.median(v) / 1e9) for k, v in res.items()), flush=True)

# ---- weight gradients: both operands k-major; v1 = stream-K + atomics, v2 = split-K slabs + reduce pass ----
if not only or 'wgrad' in only:
    for name, M, N in (('ffn1_wgrad', II, HH), ('ffn2_wgrad', HH, II), ('qkv_wgrad', 3 * HH, HH), ('attnout_wgrad', HH, HH)):
        K = MM
        A = torch.randn(K, M, device='cuda').bfloat16(); B = torch.randn(K, N, device='cuda').bfloat16()
        C = torch.zeros(M, N, device='cuda'); slabs = torch.zeros(8, M, N, device='cuda')
        runs = {'v1': (lambda: lib.uniter_gemm_bf16res_cfg(0, 1, 1, M, N, K, L.ptr(A), M, L.ptr(B), N, L.ptr(C), N, None, 0, 0, None, None, None, 0, 1, L.cur_stream()))}
        for cfg in (1, 4):
            for ns in (1, 2, 3, 4, 6, 8):
                def run(cfg=cfg, ns=ns):
                    L.check(lib.uniter_gemm_bf16v2_cfg(cfg, ns, 1, 1, M, N, K, L.ptr(A), M, L.ptr(B), N, L.ptr(slabs), N, M * N, None, 0, 0, None, None, 0, None, 0, 0, 0, L.cur_stream()))
                    L.check(lib.uniter_slab_reduce_add(L.ptr(slabs), ns, M * N, L.ptr(C), M * N, L.cur_stream()))
                runs['v2c%ds%d' % (cfg, ns)] = run
        runs['v2c1atomic'] = lambda: L.check(lib.uniter_gemm_bf16v2_cfg(1, 1, 1, 1, M, N, K, L.ptr(A), M, L.ptr(B), N, L.ptr(C), N, M * N, None, 0, 0, None, None, 0, None, 0, 0, 1, L.cur_stream()))
        for r in runs.values(): r()
        torch.cuda.synchronize()
        graphs = {k: make_graph(r) for k, r in runs.items()}
        res = {k: [] for k in runs}
        for _ in range(ROUNDS):
            for k, r in runs.items(): res[k].append(timeit(r, graphs[k]))
        fl = 2.0 * M * N * K
        print('%-14s %5dx%5dx%5d  ' % (name, M, N, K) + '  '.join('%s %.1fus %4.0fTF' % (k, statistics.median(v) * 1e3, fl / statistics.median(v) / 1e9) for k, v in res.items()), flush=True)

# ---- the four weight gradients of a layer: separate stream-K launches vs one grouped whole-K-tile launch ----
if not only or 'wgroup' in only:
    import ctypes
    K = MM
    shp = [(II, HH), (HH, II), (3 * HH, HH), (HH, HH)]
    As = [torch.randn(K, m, device='cuda').bfloat16() for m, n in shp]; Bs = [torch.randn(K, n, device='cuda').bfloat16() for m, n in shp]
    Cs = [torch.zeros(m, n, device='cuda') for m, n in shp]
    IA = ctypes.c_int * 4; PA = ctypes.c_void_p * 4
    Ms = IA(*[m for m, n in shp]); Ns = IA(*[n for m, n in shp])
    pa = PA(*[a.data_ptr() for a in As]); pb = PA(*[b.data_ptr() for b in Bs]); pc = PA(*[c.data_ptr() for c in Cs])
    def sep():
        for (m, n), a, b, c in zip(shp, As, Bs, Cs):
            lib.uniter_gemm_bf16res_cfg(0, 1, 1, m, n, K, L.ptr(a), m, L.ptr(b), n, L.ptr(c), n, None, 0, 0, None, None, None, 0, 1, L.cur_stream())
    runs = {'v1_4launches': sep,
            'group_c1': lambda: L.check(lib.uniter_wgrad_bf16_group(1, 4, Ms, Ns, K, pa, pb, pc, L.cur_stream())),
            'group_c4': lambda: L.check(lib.uniter_wgrad_bf16_group(4, 4, Ms, Ns, K, pa, pb, pc, L.cur_stream())),
            'group_c7_persistent_128x256': lambda: L.check(lib.uniter_wgrad_bf16_group(7, 4, Ms, Ns, K, pa, pb, pc, L.cur_stream()))}
    for r in runs.values(): r()
    torch.cuda.synchronize()
    graphs = {k: make_graph(r) for k, r in runs.items()}
    res = {k: [] for k in runs}
    for _ in range(ROUNDS):
        for k, r in runs.items(): res[k].append(timeit(r, graphs[k]))
    fl = sum(2.0 * m * n * K for m, n in shp)
    print('%-14s K=%d  ' % ('layer_wgrads', K) + '  '.join('%s %.1fus %4.0fTF' % (k, statistics.median(v) * 1e3, fl / statistics.median(v) / 1e9) for k, v in res.items()), flush=True)
