"""GPU: second-generation bf16-resident GEMM (LDS-DMA ring, transposed accumulator, split-K slabs).
Reference: the same bf16 values multiplied in float64 (layouts and epilogues of model/layer.py:76-78,112,140,153
forward and input-gradient products)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _ref_dgelu(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2.0))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


ALL_CFGS = [1, 2, 3, 4, 5, 6, 7, 8]      # 6..8: the persistent loader / compute kernels (csrc/gemm_bf16_p.hip, round 6)


def _run(cfg, bkm, M, N, K, epi, nsplit=1, out='both', aux_in_bf16=False, aux_out_bf16=False, beta=0, seed=0):
    if cfg >= 6 and (epi in (2, 3) or beta or (cfg == 8 and bkm) or (bkm and epi in (1, 5)) or (not bkm and epi in (4, 6))):
        return      # not built there: pre-activation / libm-GELU' epilogues, accumulate, 128 x 192 tiles for k-major weights, epilogues
                    # of the other layout (bias with k-major weights, aux with k-contiguous ones: no product of the model has them)
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g).bfloat16()
    B = torch.randn((K, N) if bkm else (N, K), generator=g).bfloat16()
    bias = torch.randn(N, generator=g)
    aux = torch.randn(M, N, generator=g)
    if aux_in_bf16:
        aux = aux.bfloat16().float()
    C0 = torch.randn(M, N, generator=g)
    ref = A.double() @ (B.double() if bkm else B.double().t())
    pre = None
    if epi in (1, 2, 5):
        ref = ref + bias.double()
    if epi == 2:
        pre, ref = ref, _ref_gelu(ref)
    if epi == 5:
        pre, ref = _ref_dgelu(ref), _ref_gelu(ref)
    if epi == 3:
        ref = ref * _ref_dgelu(aux.double())
    if epi == 4:
        ref = ref + aux.double()
    if epi == 6:
        ref = ref * aux.double()
    if beta:
        ref = ref + C0.double()
    dA, dB, dbias = A.cuda(), B.cuda(), bias.cuda()
    daux = (aux.bfloat16() if aux_in_bf16 else aux).cuda().contiguous()
    want_c = out in ('both', 'f32') or nsplit > 1 or beta
    want_cb = out in ('both', 'bf16') and nsplit == 1 and not beta
    dC = (C0.cuda().contiguous() if beta else torch.full((nsplit, M, N), float('nan'), device='cuda')) if want_c else None
    dCb = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device='cuda') if want_cb else None
    dauxo = torch.full((M, N), float('nan'), dtype=torch.bfloat16 if aux_out_bf16 else torch.float32, device='cuda')
    L.check(lib.uniter_gemm_bf16v2_cfg(cfg, nsplit, 0, int(bkm), M, N, K, L.ptr(dA), K, L.ptr(dB), dB.shape[1],
                                       L.ptr(dC), N, M * N, L.ptr(dCb), N, epi, L.ptr(dbias), L.ptr(daux),
                                       int(aux_in_bf16), L.ptr(dauxo), int(aux_out_bf16), N, beta, L.cur_stream()),
            'gemm_bf16v2')
    torch.cuda.synchronize()
    tol = 1e-4 * math.sqrt(K) * (1 + 0.1 * nsplit)
    tag = (cfg, bkm, M, N, K, epi, nsplit, out, aux_in_bf16, aux_out_bf16, beta)
    if want_c:
        got = dC.cpu().double() if beta else dC.cpu().double().sum(0)
        assert not torch.isnan(got).any(), tag
        err = (got - ref).abs().max().item()
        assert err < tol, tag + (err,)
    if want_cb:
        gb = dCb.cpu().double()
        assert not torch.isnan(gb).any(), tag
        rel = ((gb - ref).abs() / (ref.abs() + 1.0)).max().item()
        assert rel < 2.0 ** -8, tag + (rel,)
        if want_c:
            assert torch.equal(dCb.cpu(), dC[0].cpu().bfloat16()), tag      # the bf16 copy is the rounded fp32 output
    if epi in (2, 5):
        ga = dauxo.cpu().double()
        assert not torch.isnan(ga).any(), tag
        if aux_out_bf16:
            assert ((ga - pre).abs() / (pre.abs() + 1.0)).max().item() < 2.0 ** -8, tag
        else:
            assert (ga - pre).abs().max().item() < tol, tag


@pytest.mark.parametrize('cfg', ALL_CFGS)
@pytest.mark.parametrize('bkm', [0, 1])
def test_layouts_and_edges(cfg, bkm):
    _run(cfg, bkm, M=168, N=192, K=128, epi=0)                       # ragged M, N not a tile multiple
    _run(cfg, bkm, M=320, N=264, K=192, epi=1 if not bkm else 4)     # N % 8 == 0 only
    _run(cfg, bkm, M=64, N=128, K=64, epi=0, out='f32')
    _run(cfg, bkm, M=1, N=8, K=64, epi=0)                            # one row, one 8-column group
    _run(cfg, bkm, M=257, N=520, K=320, epi=1, out='bf16')


@pytest.mark.parametrize('cfg', ALL_CFGS)
def test_epilogues(cfg):
    for epi in (0, 1, 2, 5):
        _run(cfg, 0, M=200, N=256, K=128, epi=epi, aux_out_bf16=False)
        _run(cfg, 0, M=200, N=256, K=128, epi=epi, aux_out_bf16=True, out='bf16')
    for epi in (3, 4, 6):
        _run(cfg, 1, M=200, N=256, K=128, epi=epi, aux_in_bf16=False)
        _run(cfg, 1, M=200, N=256, K=128, epi=epi, aux_in_bf16=True, out='bf16')


@pytest.mark.parametrize('cfg', ALL_CFGS)
@pytest.mark.parametrize('nsplit', [2, 3, 4])
def test_split_k_slabs(cfg, nsplit):
    _run(cfg, 0, M=300, N=256, K=640, epi=1, nsplit=nsplit)
    _run(cfg, 1, M=300, N=256, K=640, epi=4, nsplit=nsplit)
    _run(cfg, 0, M=130, N=128, K=128, epi=1, nsplit=nsplit)          # fewer k-tiles than pieces: empty pieces store zeros


@pytest.mark.parametrize('cfg', ALL_CFGS)
def test_model_shapes(cfg):
    _run(cfg, 0, M=2624, N=3072, K=768, epi=5, out='bf16', aux_out_bf16=True)     # FFN up
    _run(cfg, 0, M=2624, N=2304, K=768, epi=1, out='bf16')                         # QKV
    _run(cfg, 0, M=2624, N=768, K=3072, epi=1, out='f32', nsplit=2)                # FFN down
    _run(cfg, 1, M=2624, N=3072, K=768, epi=6, out='bf16', aux_in_bf16=True)       # FFN down dgrad
    _run(cfg, 1, M=2624, N=768, K=3072, epi=4, out='f32', nsplit=2)                # FFN up dgrad
    _run(cfg, 1, M=1424, N=1024, K=3072, epi=4, out='f32')                         # UNITER-large
    _run(cfg, 0, M=128, N=128, K=128, epi=0, beta=1)


def test_rejects():
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    x = torch.zeros(64, 64, dtype=torch.bfloat16, device='cuda')
    c = torch.zeros(64, 64, device='cuda')

    def call(K=64, N=64, nsplit=1, cb=None, beta=0, akm=0):
        return lib.uniter_gemm_bf16v2_cfg(1, nsplit, akm, 0, 64, N, K, L.ptr(x), 64, L.ptr(x), 64, L.ptr(c), 64, 64 * 64,
                                          cb, 64, 0, None, None, 0, None, 0, 64, beta, L.cur_stream())
    assert call() == 0
    assert call(K=60) != 0 and b'gemm_bf16v2' in lib.uniter_last_error()
    assert call(N=60) != 0
    assert call(nsplit=2, cb=L.ptr(x)) != 0          # split-K has no bf16 output
    assert call(beta=1, cb=L.ptr(x)) != 0
    assert call(akm=1) != 0


def _run_wgrad(cfg, M, N, K, nsplit, beta, seed=0):
    """Weight-gradient layout: A [K, M] and B [K, N] both k-major, C [M, N] = A^T B (K = rows of the batch)."""
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(K, M, generator=g).bfloat16()
    B = torch.randn(K, N, generator=g).bfloat16()
    C0 = torch.randn(M, N, generator=g)
    ref = A.double().t() @ B.double()
    dA, dB = A.cuda(), B.cuda()
    tol = 1e-4 * math.sqrt(K) * (1 + 0.1 * nsplit)
    if beta:
        dC = C0.cuda().contiguous()
        L.check(lib.uniter_gemm_bf16v2_cfg(cfg, 1, 1, 1, M, N, K, L.ptr(dA), M, L.ptr(dB), N, L.ptr(dC), N, M * N, None, 0, 0,
                                           None, None, 0, None, 0, 0, 1, L.cur_stream()), 'gemm_bf16v2 wgrad atomics')
        torch.cuda.synchronize()
        err = (dC.cpu().double() - (ref + C0.double())).abs().max().item()
    else:
        slabs = torch.full((nsplit, M, N), float('nan'), device='cuda')
        L.check(lib.uniter_gemm_bf16v2_cfg(cfg, nsplit, 1, 1, M, N, K, L.ptr(dA), M, L.ptr(dB), N, L.ptr(slabs), N, M * N, None, 0,
                                           0, None, None, 0, None, 0, 0, 0, L.cur_stream()), 'gemm_bf16v2 wgrad slabs')
        out = C0.cuda().contiguous()
        L.check(lib.uniter_slab_reduce_add(L.ptr(slabs), nsplit, M * N, L.ptr(out), M * N, L.cur_stream()), 'slab_reduce_add')
        torch.cuda.synchronize()
        assert not torch.isnan(slabs).any()
        err = (out.cpu().double() - (ref + C0.double())).abs().max().item()
    assert err < tol, (cfg, M, N, K, nsplit, beta, err)


@pytest.mark.parametrize('cfg', [1, 4])
def test_weight_gradient_layout(cfg):
    _run_wgrad(cfg, M=128, N=128, K=128, nsplit=1, beta=0)
    _run_wgrad(cfg, M=256, N=384, K=640, nsplit=3, beta=0)
    _run_wgrad(cfg, M=256, N=128, K=200, nsplit=2, beta=0)          # ragged K (rows beyond K must read as zero)
    _run_wgrad(cfg, M=128, N=256, K=1458, nsplit=4, beta=0)         # ragged K, several pieces
    _run_wgrad(cfg, M=136, N=200, K=256, nsplit=1, beta=0)          # output edges
    _run_wgrad(cfg, M=128, N=128, K=192, nsplit=1, beta=1)          # C += by atomics
    _run_wgrad(cfg, M=768, N=3072, K=2624, nsplit=3, beta=0)        # FFN weight gradient of the model
    _run_wgrad(cfg, M=2304, N=768, K=2624, nsplit=4, beta=0)


def _group_call(lib, L, cfg, Ms, Ns, K, As, Bs, Cs):
    import ctypes
    n = len(Ms)
    IA = ctypes.c_int * n
    PA = ctypes.c_void_p * n
    return lib.uniter_wgrad_bf16_group(cfg, n, IA(*Ms), IA(*Ns), K, PA(*[a.data_ptr() for a in As]),
                                       PA(*[b.data_ptr() for b in Bs]), PA(*[c.data_ptr() for c in Cs]), L.cur_stream())


@pytest.mark.parametrize('cfg', [1, 4, 7])
@pytest.mark.parametrize('shapes,K', [([(128, 128)], 64), ([(136, 200), (256, 128), (8, 8)], 200),
                                      ([(3072, 768), (768, 3072), (2304, 768), (768, 768)], 2624),
                                      ([(256, 64), (64, 256), (192, 64), (64, 64)], 1458)])
def test_weight_gradient_group(cfg, shapes, K):
    """Up to four dW += dY^T X products of one reduction length in ONE launch, whole-K tiles, read-modify-write of dW:
    against fp64 on the same bf16 operands, and bit-identical when repeated (no atomics)."""
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(len(shapes) * 1000 + K)
    As = [torch.randn(K, M, generator=g).bfloat16().cuda() for M, N in shapes]
    Bs = [torch.randn(K, N, generator=g).bfloat16().cuda() for M, N in shapes]
    C0 = [torch.randn(M, N, generator=g) for M, N in shapes]
    outs = []
    for rep in range(2):
        Cs = [c.cuda().contiguous() for c in C0]
        L.check(_group_call(lib, L, cfg, [m for m, _ in shapes], [n for _, n in shapes], K, As, Bs, Cs), 'wgrad group')
        torch.cuda.synchronize()
        outs.append([c.cpu() for c in Cs])
    for (M, N), a, b, c0, c, c2 in zip(shapes, As, Bs, C0, outs[0], outs[1]):
        ref = a.cpu().double().t() @ b.cpu().double() + c0.double()
        assert (c.double() - ref).abs().max().item() < 1e-4 * math.sqrt(K), (M, N, K)
        assert torch.equal(c, c2)


@pytest.mark.parametrize('shapes,K,wgs', [([(3072, 768), (768, 3072), (2304, 768), (768, 768)], 2624, 0),
                                          ([(256, 64), (64, 256), (192, 64), (64, 64)], 1458, 0),
                                          ([(136, 200), (256, 128), (8, 8)], 200, 0),
                                          ([(3072, 768), (768, 3072), (2304, 768), (768, 768)], 333, 216)])
@pytest.mark.parametrize('overwrite', [0, 1])
@pytest.mark.parametrize('cfg', [0, 7])
def test_weight_gradient_group_riders(shapes, K, wgs, overwrite, cfg):
    """The riders of the grouped bf16 weight-gradient launch (uniter_wgrad_bf16_group_riders; the same block as the fp32x3
    launch's, tests/test_gemm_x3_gpu.py): dW bit-identical to the plain launch; colsum_out += column sums of product 0's A operand;
    three column-reduction jobs; the sum of squares of everything written as 4 x grid partial sums; reproducible."""
    import ctypes
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    g = torch.Generator().manual_seed(len(shapes) * 77 + K)
    As = [torch.randn(K, M, generator=g).bfloat16().cuda() for M, N in shapes]
    Bs = [(torch.randn(K, N, generator=g) * 0.05).bfloat16().cuda() for M, N in shapes]
    C0 = [torch.randn(M, N, generator=g).cuda() for M, N in shapes]
    Ms, Ns = [m for m, _ in shapes], [n for _, n in shapes]
    n = len(shapes)
    H = 192
    parts = [torch.randn(328, 3 * H, generator=g).cuda(), torch.randn(5, 3 * H, generator=g).cuda(), torch.randn(70, 3 * H, generator=g).cuda()]
    job_n, job_seg, job_nout = [3 * H, 3 * H, H + 8], [H, 3 * H, H], [3, 1, 2]
    out0 = [[torch.randn(H, generator=g).cuda() for _ in range(3)], [torch.randn(3 * H, generator=g).cuda()],
            [torch.randn(H, generator=g).cuda(), torch.randn(H, generator=g).cuda()]]
    cs0 = torch.randn(Ms[0], generator=g).cuda()
    IA, PA = ctypes.c_int * n, ctypes.c_void_p * n
    # (cfg 7: the persistent 128 x 256-tile launch of gemm_bf16_p.hip -- 8 slots per workgroup, no colsum_out rider)
    per_wg = 8 if cfg == 7 else 4
    slots = lib.uniter_wgrad_bf16_group_slots_cfg(cfg, n, IA(*Ms), IA(*Ns), wgs)
    assert slots > 0 and slots % 32 == 0
    if cfg != 7:
        assert slots == lib.uniter_wgrad_bf16_group_slots(n, IA(*Ms), IA(*Ns), wgs)

    def run(with_riders):
        Cs = [c.clone() for c in C0]
        outs = [[o.clone() for o in job] for job in out0]
        cs = cs0.clone()
        ssq = torch.full((slots,), float('nan'), dtype=torch.float64, device='cuda')
        x = L.X3RidersC()
        if with_riders:
            x.ssq, x.colsum_out, x.njobs = ssq.data_ptr(), (None if cfg == 7 else cs.data_ptr()), 3
            for j in range(3):
                x.part[j] = parts[j].data_ptr(); x.nparts[j] = parts[j].shape[0]; x.stride[j] = 3 * H
                x.n[j] = job_n[j]; x.seg[j] = job_seg[j]
                for o in range(job_nout[j]):
                    x.out[j][o] = outs[j][o].data_ptr()
        L.check(lib.uniter_wgrad_bf16_group_riders(cfg, n, IA(*Ms), IA(*Ns), K, PA(*[a.data_ptr() for a in As]),
                                                   PA(*[b.data_ptr() for b in Bs]), PA(*[c.data_ptr() for c in Cs]), overwrite,
                                                   wgs, ctypes.byref(x) if with_riders else None, L.cur_stream()), 'wgrad_bf16_group_riders')
        torch.cuda.synchronize()
        if with_riders:
            assert x.grid * per_wg == slots and x.nred == sum((k + 63) // 64 for k in job_n)
        return Cs, outs, cs, ssq

    plain = run(False)
    a, b = run(True), run(True)
    for c_plain, c_a, c_b in zip(plain[0], a[0], b[0]):
        assert torch.equal(c_plain, c_a) and torch.equal(c_a, c_b)
    assert torch.equal(a[3], b[3]) and torch.isfinite(a[3]).all()
    total = 0.0
    if cfg != 7:
        ref = cs0.double().cpu() + As[0].double().cpu().sum(0)
        assert (a[2].double().cpu() - ref).abs().max().item() < 2e-6 * math.sqrt(K) * 4 and torch.equal(a[2], b[2])
        total += float((a[2].double() ** 2).sum())
    else:
        assert torch.equal(a[2], cs0)            # untouched
    for j in range(3):
        full = parts[j].double().cpu().sum(0)[:job_n[j]]
        for o in range(job_nout[j]):
            seg = full[o * job_seg[j]:(o + 1) * job_seg[j]]
            exp = out0[j][o].double().cpu().clone()
            exp[:seg.numel()] += seg
            assert (a[1][j][o].double().cpu() - exp).abs().max().item() < 1e-5 * math.sqrt(parts[j].shape[0]), (j, o)
            if seg.numel() < exp.numel():
                assert torch.equal(a[1][j][o][seg.numel():], out0[j][o][seg.numel():])
            total += float((a[1][j][o][:seg.numel()].double() ** 2).sum())
    for c in a[0]:
        total += float((c.double() ** 2).sum())
    got = float(a[3].sum())
    assert abs(got - total) <= 1e-6 * total, (got, total)


def test_weight_gradient_group_rejects_bad_arguments():
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    a = torch.zeros(64, 12, dtype=torch.bfloat16, device='cuda')
    c = torch.zeros(12, 12, device='cuda')
    assert _group_call(lib, L, 1, [12], [12], 64, [a], [a], [c]) != 0          # M % 8
    a = torch.zeros(64, 16, dtype=torch.bfloat16, device='cuda')
    c = torch.zeros(16, 16, device='cuda')
    assert _group_call(lib, L, 1, [16] * 5, [16] * 5, 64, [a] * 5, [a] * 5, [c] * 5) != 0     # more than four products
