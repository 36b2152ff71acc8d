"""GPU: fused dropout+residual+LayerNorm fwd/bwd and column sums vs float64 host math."""
import numpy as np
import pytest
import torch

from oracle import philox

pytestmark = pytest.mark.gpu


def _ref_fwd(x, res, g, b, p, seed, offset, site):
    x = x.double()
    if p > 0:
        keep = torch.from_numpy(philox.keep_mask(x.numel(), p, seed, offset, site)).view(x.shape)
        x = x * keep.double() * float(np.float32(1.0) / np.float32(1.0 - p))
    z = x + (res.double() if res is not None else 0)
    mu = z.mean(-1, keepdim=True)
    var = ((z - mu) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + 1e-12)
    return z, (z - mu) * rstd * g.double() + b.double(), mu.squeeze(-1), rstd.squeeze(-1)


@pytest.mark.parametrize('M,H,p', [(37, 768, 0.0), (37, 768, 0.1), (9, 128, 0.1), (50, 1024, 0.1), (5, 256, 0.3)])
def test_ln_fwd_bwd(M, H, p):
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(M + H)
    x, res, dy = (torch.randn(M, H, generator=gen) for _ in range(3))
    g = 1 + 0.1 * torch.randn(H, generator=gen)
    b = 0.1 * torch.randn(H, generator=gen)
    seed, offset, site = 0x1234567890AB, 7, 13
    z_ref, y_ref, mu_ref, rstd_ref = _ref_fwd(x, res, g, b, p, seed, offset, site)
    d = lambda t: t.cuda().contiguous()
    dx_, dres, dg_, db_ = d(x), d(res), d(g), d(b)
    z, y = torch.empty(M, H, device='cuda'), torch.empty(M, H, device='cuda')
    mu, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
    L.check(lib.uniter_ln_fwd(L.ptr(dx_), L.ptr(dres), L.ptr(dg_), L.ptr(db_), L.ptr(z), L.ptr(y),
                              L.ptr(mu), L.ptr(rstd), M, H, p, seed, offset, site, L.cur_stream()))
    torch.cuda.synchronize()
    assert (z.cpu().double() - z_ref).abs().max() < 1e-5
    assert (y.cpu().double() - y_ref).abs().max() < 2e-5
    assert (mu.cpu().double() - mu_ref).abs().max() < 1e-5
    assert ((rstd.cpu().double() - rstd_ref) / rstd_ref).abs().max() < 1e-5
    # backward via autograd of the float64 reference
    xr = x.double().requires_grad_(True)
    rr = res.double().requires_grad_(True)
    gr = g.double().requires_grad_(True)
    br = b.double().requires_grad_(True)
    xx = xr
    if p > 0:
        keep = torch.from_numpy(philox.keep_mask(x.numel(), p, seed, offset, site)).view(x.shape)
        xx = xr * keep.double() * float(np.float32(1.0) / np.float32(1.0 - p))
    yy = torch.nn.functional.layer_norm(xx + rr, (H,), gr, br, 1e-12)
    yy.backward(dy.double())
    ddy = d(dy)
    dz, dxo = torch.empty(M, H, device='cuda'), torch.empty(M, H, device='cuda')
    dgam = torch.full((H,), 0.5, device='cuda')      # accumulate semantics
    dbet = torch.full((H,), -0.25, device='cuda')
    ws_n = lib.uniter_ln_bwd_ws_bytes(M, H)
    ws = torch.empty(ws_n, dtype=torch.uint8, device='cuda')
    dbias = torch.full((H,), 0.125, device='cuda')
    L.check(lib.uniter_ln_bwd(L.ptr(ddy), L.ptr(z), L.ptr(mu), L.ptr(rstd), L.ptr(dg_), L.ptr(dz),
                              L.ptr(dxo), L.ptr(dgam), L.ptr(dbet), L.ptr(dbias), M, H, p, seed, offset, site,
                              L.ptr(ws), ws_n, L.cur_stream()))
    torch.cuda.synchronize()
    assert (dbias.cpu().double() - 0.125 - xr.grad.sum(0)).abs().max() < 1e-4
    assert (dz.cpu().double() - rr.grad).abs().max() < 5e-5
    assert (dxo.cpu().double() - xr.grad).abs().max() < 5e-5
    assert (dgam.cpu().double() - 0.5 - gr.grad).abs().max() < 1e-4
    assert (dbet.cpu().double() + 0.25 - br.grad).abs().max() < 1e-4


def test_ln_inference_mode_null_outputs():
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    x = torch.randn(6, 768, device='cuda')
    g, b = torch.ones(768, device='cuda'), torch.zeros(768, device='cuda')
    y = torch.empty_like(x)
    L.check(lib.uniter_ln_fwd(L.ptr(x), None, L.ptr(g), L.ptr(b), None, L.ptr(y), None, None,
                              6, 768, 0.0, 0, 0, 0, L.cur_stream()))
    ref = torch.nn.functional.layer_norm(x.cpu(), (768,), None, None, 1e-12)
    assert (y.cpu() - ref).abs().max() < 1e-5


@pytest.mark.parametrize('M,N', [(2624, 3072), (7, 128), (576, 768)])
def test_colsum(M, N):
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    X = torch.randn(M, N)
    out = torch.full((N,), 2.0, device='cuda')
    n = lib.uniter_colsum_ws_bytes(M, N)
    ws = torch.empty(n, dtype=torch.uint8, device='cuda')
    dX = X.cuda()
    L.check(lib.uniter_colsum_f32(L.ptr(dX), M, N, N, L.ptr(out), 1, L.ptr(ws), n, L.cur_stream()))
    assert (out.cpu().double() - 2.0 - X.double().sum(0)).abs().max() < 2e-4
    L.check(lib.uniter_colsum_f32(L.ptr(dX), M, N, N, L.ptr(out), 0, L.ptr(ws), n, L.cur_stream()))
    assert (out.cpu().double() - X.double().sum(0)).abs().max() < 2e-4


@pytest.mark.parametrize('M,H,p,nslab', [(37, 768, 0.1, 1), (2624, 768, 0.1, 2), (9, 128, 0.0, 3), (50, 1024, 0.1, 1)])
def test_ln_x3_copies_are_the_exact_pieces_of_the_fp32_outputs(M, H, p, nslab):
    """fp32x3 mode: the operand copy the row passes hand to the next product is the three bf16 pieces [M][3][H] of the
    fp32 output they also write -- their sum reproduces it bit for bit (forward y, backward dx after dropout), and the
    fp32 outputs are those of the plain entry points (slabs summed in the pass)."""
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    gen = torch.Generator(device='cuda').manual_seed(M + H + nslab)
    slabs = torch.randn(nslab, M, H, device='cuda', generator=gen)
    res, dy = (torch.randn(M, H, device='cuda', generator=gen) for _ in range(2))
    g = 1 + 0.1 * torch.randn(H, device='cuda', generator=gen)
    b = 0.1 * torch.randn(H, device='cuda', generator=gen)
    seed, offset, site = 0xABCDEF, 3, 5

    def join(x3):
        o = torch.empty(M, H, device='cuda')
        L.check(lib.uniter_join3(L.ptr(x3), M, H, 3 * H, H, L.ptr(o), H, L.cur_stream()))
        return o
    z, y, z0, y0 = (torch.empty(M, H, device='cuda') for _ in range(4))
    mu, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
    yx = torch.full((M, 3, H), float('nan'), dtype=torch.bfloat16, device='cuda')
    L.check(lib.uniter_ln_fwd_slabs_x3(L.ptr(slabs), nslab, M * H, L.ptr(res), L.ptr(g), L.ptr(b), L.ptr(z), L.ptr(y), L.ptr(yx),
                                       L.ptr(mu), L.ptr(rstd), M, H, p, seed, offset, site, L.cur_stream()))
    L.check(lib.uniter_ln_fwd_slabs(L.ptr(slabs), nslab, M * H, L.ptr(res), L.ptr(g), L.ptr(b), L.ptr(z0), L.ptr(y0), None,
                                    L.ptr(mu), L.ptr(rstd), M, H, p, seed, offset, site, L.cur_stream()))
    assert torch.equal(y, y0) and torch.equal(z, z0) and torch.equal(join(yx), y)
    ws_n = lib.uniter_ln_bwd_ws_bytes(M, H)
    ws = torch.empty(ws_n, dtype=torch.uint8, device='cuda')
    dys = torch.randn(nslab, M, H, device='cuda', generator=gen)
    dz, dx, dz0, dx0 = (torch.empty(M, H, device='cuda') for _ in range(4))
    dxx = torch.full((M, 3, H), float('nan'), dtype=torch.bfloat16, device='cuda')
    L.check(lib.uniter_ln_bwd_rows_slabs_x3(L.ptr(dys), nslab, M * H, L.ptr(z), L.ptr(mu), L.ptr(rstd), L.ptr(g), L.ptr(dz), L.ptr(dx),
                                            L.ptr(dxx), 1, M, H, p, seed, offset, site, L.ptr(ws), ws_n, L.cur_stream()))
    L.check(lib.uniter_ln_bwd_rows_slabs(L.ptr(dys), nslab, M * H, L.ptr(z), L.ptr(mu), L.ptr(rstd), L.ptr(g), L.ptr(dz0), L.ptr(dx0),
                                         None, 1, M, H, p, seed, offset, site, L.ptr(ws), ws_n, L.cur_stream()))
    assert torch.equal(dz, dz0) and torch.equal(dx, dx0) and torch.equal(join(dxx), dx)
    # the pieces alone (no fp32 dx: what the model's schedule asks for)
    dxx2 = torch.full((M, 3, H), float('nan'), dtype=torch.bfloat16, device='cuda')
    L.check(lib.uniter_ln_bwd_rows_slabs_x3(L.ptr(dys), nslab, M * H, L.ptr(z), L.ptr(mu), L.ptr(rstd), L.ptr(g), L.ptr(dz), None,
                                            L.ptr(dxx2), 1, M, H, p, seed, offset, site, L.ptr(ws), ws_n, L.cur_stream()))
    assert torch.equal(dxx2, dxx)


@pytest.mark.parametrize('M,H,p', [(37, 768, 0.1), (2624, 768, 0.1), (50, 1024, 0.3), (9, 128, 0.1)])
def test_keep_flags_drawn_ahead_are_the_row_passes_own(M, H, p):
    """uniter_hidden_keep_bits_gen + uniter_ln_set_next_keep_bits: the forward and backward row passes that READ the keep flags
    (one nibble per 4-element group, drawn ahead for several sites in one launch) give the outputs of the passes that draw them
    with Philox themselves, bit for bit; the flags are the oracle's (oracle/philox.py); the hand-over holds for ONE launch."""
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    gen = torch.Generator().manual_seed(M * 3 + H)
    x, res, dy = (torch.randn(M, H, generator=gen).cuda() for _ in range(3))
    g = (1 + 0.1 * torch.randn(H, generator=gen)).cuda(); b = (0.1 * torch.randn(H, generator=gen)).cuda()
    seed, offset = 0x1234567890AB, 7
    nsites, site_a0, site_b0, step = 6, 3, 4, 4            # the sites of three encoder layers: 3, 4, 7, 8, 11, 12
    stride = (lib.uniter_hidden_keep_bits_bytes(M * H) + 255) // 256 * 256
    bits = torch.zeros(nsites * stride, dtype=torch.uint8, device='cuda')
    L.check(lib.uniter_hidden_keep_bits_gen(L.ptr(bits), stride, nsites, site_a0, site_b0, step, M * H, p, seed, offset, L.cur_stream()))
    torch.cuda.synchronize()
    for s in (0, 3, 5):
        site = (site_b0 if s & 1 else site_a0) + (s >> 1) * step
        # the flags themselves: the oracle's mask
        keep = philox.keep_mask(M * H, p, seed, offset, site).astype(np.uint8)
        nib = bits[s * stride:s * stride + (M * H // 4 + 1) // 2].cpu().numpy()
        got = np.stack([(nib >> sh) & 1 for sh in range(8)], 1).reshape(-1)[:M * H]         # byte -> 8 flags: low nibble first
        assert np.array_equal(got, keep), site
        site_ptr = bits.data_ptr() + s * stride
        outs = []
        for ahead in (False, True, False):
            z, y = torch.empty(M, H, device='cuda'), torch.empty(M, H, device='cuda')
            mu, rstd = torch.empty(M, device='cuda'), torch.empty(M, device='cuda')
            if ahead:
                L.check(lib.uniter_ln_set_next_keep_bits(site_ptr))
            L.check(lib.uniter_ln_fwd(L.ptr(x), L.ptr(res), L.ptr(g), L.ptr(b), L.ptr(z), L.ptr(y), L.ptr(mu), L.ptr(rstd), M, H, p,
                                      seed, offset, site, L.cur_stream()))
            dz, dx = torch.empty(M, H, device='cuda'), torch.empty(M, H, device='cuda')
            dgam, dbet, dbias = (torch.zeros(H, device='cuda') for _ in range(3))
            ws_n = lib.uniter_ln_bwd_ws_bytes(M, H)
            ws = torch.empty(ws_n, dtype=torch.uint8, device='cuda')
            if ahead:
                L.check(lib.uniter_ln_set_next_keep_bits(site_ptr))
            L.check(lib.uniter_ln_bwd(L.ptr(dy), L.ptr(z), L.ptr(mu), L.ptr(rstd), L.ptr(g), L.ptr(dz), L.ptr(dx), L.ptr(dgam),
                                      L.ptr(dbet), L.ptr(dbias), M, H, p, seed, offset, site, L.ptr(ws), ws_n, L.cur_stream()))
            torch.cuda.synchronize()
            outs.append((z, y, dz, dx, dbias))
        for a, c in zip(outs[0], outs[1]):
            assert torch.equal(a, c)
        for a, c in zip(outs[0], outs[2]):          # (the third run drew its own flags again: the hand-over was for one launch)
            assert torch.equal(a, c)
