"""GPU: the fp32x3 mode's attention with its products on the bf16 matrix pipe (csrc/attention_x3.hip: three bf16 pieces per
operand value, six MFMA products per block) vs the float64 host implementation of BertSelfAttention's math
(model/layer.py:80-100) -- at the fp32 kernels' tolerances, and with errors no larger than theirs."""
import pytest
import torch

from test_attention_gpu import _ref

pytestmark = pytest.mark.gpu

SEED, OFFSET, SITE = 0xABCDEF0123, 5, 2


def _pieces_sum(x3):
    """[rows][3][n] bf16 -> fp32 sum, added small to large in float64 (exact)."""
    d = x3.double()
    return (d[:, 2] + d[:, 1] + d[:, 0])


def _keep(lib, Lb, B, L, nh, p):
    nbytes = lib.uniter_attn_keep_bits_bytes(B, L, nh)
    keep = torch.zeros(max(nbytes, 2) // 2, dtype=torch.int16, device='cuda')
    if p > 0:
        Lb.check(lib.uniter_attn_keep_bits_gen(Lb.ptr(keep), 0, 1, B, L, nh, p, SEED, OFFSET, SITE, 0, Lb.cur_stream()))
    return keep


def _inputs(B, L, nh, ragged=True):
    H = nh * 64
    g = torch.Generator().manual_seed(B * 1000 + L)
    qkv = torch.randn(B * L, 3 * H, generator=g)
    mask = torch.ones(B, L)
    if ragged:
        for b in range(1, B):
            mask[b, L - (b * 7) % L:] = 0
    dctx = torch.randn(B * L, H, generator=g)
    return qkv, mask, dctx


@pytest.mark.parametrize('B,L,nh,p', [(2, 164, 2, 0.0), (2, 164, 2, 0.1), (3, 16, 2, 0.1), (1, 33, 1, 0.25), (2, 100, 12, 0.1),
                                       (1, 178, 16, 0.0), (2, 192, 2, 0.1), (2, 1, 1, 0.0), (16, 164, 12, 0.1)])
def test_attention_x3_fwd_bwd_matches_float64_and_the_fp32_kernels(B, L, nh, p):
    from meme_challenge_amd import _lib as Lb
    lib = Lb.lib()
    H = nh * 64
    qkv, mask, dctx = _inputs(B, L, nh)
    qr = qkv.double().requires_grad_(True)
    ctx_ref, lse_ref = _ref(qr, mask.double(), B, L, nh, p, SEED, OFFSET, SITE)
    ctx_ref.backward(dctx.double())
    dq, dm, dd = qkv.cuda(), mask.cuda(), dctx.cuda()
    keep = _keep(lib, Lb, B, L, nh, p)
    kp = Lb.ptr(keep) if p > 0 else None
    nan = float('nan')

    def fwd(x3_products):
        ctx = torch.full((B * L, H), nan, device='cuda')
        ctx3 = torch.full((B * L, 3, H), nan, dtype=torch.bfloat16, device='cuda')
        lse = torch.full((B, nh, L), nan, device='cuda')
        if x3_products:
            Lb.check(lib.uniter_attn_x3_fwd(Lb.ptr(dq), Lb.ptr(dm), None, Lb.ptr(ctx), Lb.ptr(ctx3), Lb.ptr(lse), kp, B, L, nh, p,
                                            Lb.cur_stream()))
        else:
            Lb.check(lib.uniter_attn_fwd_pre_x3(Lb.ptr(dq), Lb.ptr(dm), None, Lb.ptr(ctx), Lb.ptr(ctx3), Lb.ptr(lse), Lb.ptr(keep),
                                                1 if p > 0 else 0, B, L, nh, p, SEED, OFFSET, SITE, Lb.cur_stream()))
        torch.cuda.synchronize()
        return ctx, ctx3, lse

    def bwd(x3_products, ctx, lse):
        dqkv = torch.full((B * L, 3 * H), nan, device='cuda')
        dqkv3 = torch.full((B * L, 3, 3 * H), nan, dtype=torch.bfloat16, device='cuda')
        part = torch.full((B, 3 * H), nan, device='cuda')
        delta = torch.full((B, nh, L), nan, device='cuda')
        if x3_products:
            Lb.check(lib.uniter_attn_x3_bwd(Lb.ptr(dq), Lb.ptr(dm), None, Lb.ptr(ctx), Lb.ptr(lse), Lb.ptr(dd), 1, 0, Lb.ptr(dqkv),
                                            Lb.ptr(dqkv3), Lb.ptr(part), kp, Lb.ptr(delta), B, L, nh, p, Lb.cur_stream()))
        else:
            wsb = lib.uniter_attn_bwd_ws_bytes(B, L, nh)
            ws = torch.empty(max(wsb, 4) // 4, device='cuda')
            Lb.check(lib.uniter_attn_bwd_ex_x3(Lb.ptr(dq), Lb.ptr(dm), None, Lb.ptr(ctx), Lb.ptr(lse), Lb.ptr(dd), Lb.ptr(dqkv),
                                               Lb.ptr(dqkv3), Lb.ptr(part), Lb.ptr(keep), Lb.ptr(delta), B, L, nh, p, SEED, OFFSET,
                                               SITE, Lb.ptr(ws), wsb, Lb.cur_stream()))
        torch.cuda.synchronize()
        return dqkv, dqkv3, part, delta

    ctx, ctx3, lse = fwd(True)
    ctx_f, _, lse_f = fwd(False)
    e_x3 = (ctx.cpu().double() - ctx_ref.detach()).abs().max().item()
    e_f32 = (ctx_f.cpu().double() - ctx_ref.detach()).abs().max().item()
    assert e_x3 < 2e-5 and (lse.cpu().double() - lse_ref.detach()).abs().max() < 2e-5
    assert e_x3 <= 2 * e_f32 + 2e-7, (e_x3, e_f32)
    # the pieces are the exact split of the fp32 output
    assert torch.equal(_pieces_sum(ctx3).float(), ctx) and torch.equal(ctx3[:, 0], ctx.bfloat16())

    dqkv, dqkv3, part, delta = bwd(True, ctx, lse)
    dqkv_f, _, part_f, delta_f = bwd(False, ctx_f, lse_f)
    ref = qr.grad
    scale = max(1.0, ref.abs().max().item())
    for name, sl in (('dq', slice(0, H)), ('dk', slice(H, 2 * H)), ('dv', slice(2 * H, 3 * H))):
        ex = (dqkv.cpu().double() - ref)[:, sl].abs().max().item()
        ef = (dqkv_f.cpu().double() - ref)[:, sl].abs().max().item()
        assert ex < 1e-4, (name, ex)
        assert ex <= 2 * ef + 2e-7 * scale, (name, ex, ef)
    assert torch.equal(_pieces_sum(dqkv3).float(), dqkv) and torch.equal(dqkv3[:, 0], dqkv.bfloat16())
    colsum = dqkv.view(B, L, 3 * H).double().sum(1)
    assert (part.double() - colsum).abs().max().item() < 1e-4 * max(1.0, colsum.abs().max().item())
    dref = (ctx_ref.detach() * dctx.double()).view(B, L, nh, 64).sum(-1).permute(0, 2, 1)
    assert (delta.cpu().double() - dref).abs().max() < 1e-4
    assert (delta - delta_f).abs().max() < 1e-5

    # outputs are optional one by one: pieces only (what the model asks for), fp32 only
    ctx_only = torch.full((B * L, H), nan, device='cuda')
    Lb.check(lib.uniter_attn_x3_fwd(Lb.ptr(dq), Lb.ptr(dm), None, Lb.ptr(ctx_only), None, None, kp, B, L, nh, p, Lb.cur_stream()))
    d3_only = torch.full((B * L, 3, 3 * H), nan, dtype=torch.bfloat16, device='cuda')
    Lb.check(lib.uniter_attn_x3_bwd(Lb.ptr(dq), Lb.ptr(dm), None, Lb.ptr(ctx), Lb.ptr(lse), Lb.ptr(dd), 1, 0, None, Lb.ptr(d3_only),
                                    None, kp, Lb.ptr(delta), B, L, nh, p, Lb.cur_stream()))
    torch.cuda.synchronize()
    assert torch.equal(ctx_only, ctx) and torch.equal(d3_only, dqkv3)
    # dctx as two k-pieces of the attention-output input gradient: summed while read, the same bits as the summed tensor
    g2 = torch.Generator().manual_seed(7)
    a_ = torch.randn(B * L, H, generator=g2).cuda()
    slabs = torch.stack([a_, dd - a_]).contiguous()
    summed = (slabs[0] + slabs[1]).contiguous()
    outs = []
    for ptr_, ns, stride in ((Lb.ptr(summed), 1, 0), (Lb.ptr(slabs), 2, B * L * H)):
        o3 = torch.full((B * L, 3, 3 * H), nan, dtype=torch.bfloat16, device='cuda')
        dl = torch.full((B, nh, L), nan, device='cuda')
        Lb.check(lib.uniter_attn_x3_bwd(Lb.ptr(dq), Lb.ptr(dm), None, Lb.ptr(ctx), Lb.ptr(lse), ptr_, ns, stride, None, Lb.ptr(o3),
                                        None, kp, Lb.ptr(dl), B, L, nh, p, Lb.cur_stream()))
        torch.cuda.synchronize()
        outs.append((o3, dl))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize('lens,nh,p', [([164, 40, 1, 97], 2, 0.1), ([33, 192], 3, 0.0), ([5], 1, 0.2)])
def test_attention_x3_packed_batches_match_the_padded_form(lens, nh, p):
    """cu_seqlens form (token-packed batches): sample b owns rows cu[b] .. cu[b+1]-1, no mask; same numbers as the padded batch."""
    from meme_challenge_amd import _lib as Lb
    lib = Lb.lib()
    B, L, H = len(lens), max(lens), nh * 64
    qkv, _, dctx = _inputs(B, L, nh, ragged=False)
    mask = torch.zeros(B, L)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    rows = torch.cat([torch.arange(n) + b * L for b, n in enumerate(lens)])
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32).cuda()
    keep = _keep(lib, Lb, B, L, nh, p)
    kp = Lb.ptr(keep) if p > 0 else None
    dctx = dctx * mask.view(-1, 1)        # padded queries carry no gradient (they would reach dK / dV of the valid keys)
    dq, dm, dd = qkv.cuda(), mask.cuda(), dctx.cuda()
    pq, pd_ = qkv[rows].contiguous().cuda(), dctx[rows].contiguous().cuda()
    T = rows.numel()

    def run(q, m, c, d, n):
        ctx = torch.zeros(n, H, device='cuda'); lse = torch.zeros(B, nh, L, device='cuda')
        dqkv = torch.zeros(n, 3 * H, device='cuda'); d3 = torch.zeros(n, 3, 3 * H, dtype=torch.bfloat16, device='cuda')
        part = torch.zeros(B, 3 * H, device='cuda'); delta = torch.zeros(B, nh, L, device='cuda')
        Lb.check(lib.uniter_attn_x3_fwd(Lb.ptr(q), m, c, Lb.ptr(ctx), None, Lb.ptr(lse), kp, B, L, nh, p, Lb.cur_stream()))
        Lb.check(lib.uniter_attn_x3_bwd(Lb.ptr(q), m, c, Lb.ptr(ctx), Lb.ptr(lse), Lb.ptr(d), 1, 0, Lb.ptr(dqkv), Lb.ptr(d3), Lb.ptr(part),
                                        kp, Lb.ptr(delta), B, L, nh, p, Lb.cur_stream()))
        torch.cuda.synchronize()
        return ctx, lse, dqkv, part

    ctx_p, lse_p, dqkv_p, part_p = run(pq, None, Lb.ptr(cu), pd_, T)
    ctx_d, lse_d, dqkv_d, part_d = run(dq, Lb.ptr(dm), None, dd, B * L)
    r = rows.cuda()
    # padded keys carry exp(-10000 - max) = 0 exactly in fp32, so the two forms differ only by the order of exact zeros
    assert (ctx_p - ctx_d[r]).abs().max() < 1e-6 and (dqkv_p - dqkv_d[r]).abs().max() < 1e-5
    for b, n in enumerate(lens):
        assert (lse_p[b, :, :n] - lse_d[b, :, :n]).abs().max() < 1e-5
    # the padded form's bias partials also sum its padded rows' (non-zero) gradients: compare against the packed rows directly
    off = 0
    for b, n in enumerate(lens):
        assert (part_p[b].double() - dqkv_p[off:off + n].double().sum(0)).abs().max() < 1e-4 * max(1.0, dqkv_p.abs().max().item())
        off += n


def test_attention_x3_fully_masked_sample_and_rejects():
    from meme_challenge_amd import _lib as Lb
    lib = Lb.lib()
    B, L, nh = 1, 40, 1
    qkv = torch.randn(B * L, 192)
    mask = torch.zeros(B, L)
    ctx_ref, _ = _ref(qkv, mask, B, L, nh, 0.0, 0, 0, 0)
    ctx = torch.empty(B * L, 64, device='cuda')
    dq, dm = qkv.cuda(), mask.cuda()
    Lb.check(lib.uniter_attn_x3_fwd(Lb.ptr(dq), Lb.ptr(dm), None, Lb.ptr(ctx), None, None, None, B, L, nh, 0.0, Lb.cur_stream()))
    assert (ctx.cpu() - ctx_ref).abs().max() < 1e-3      # score - 10000 is quantised to the ulp of 1e4, as in the reference
    assert lib.uniter_attn_x3_max_len() == 192
    big = torch.zeros(193, 192, device='cuda'); m193 = torch.ones(1, 193, device='cuda'); c193 = torch.zeros(193, 64, device='cuda')
    assert lib.uniter_attn_x3_fwd(Lb.ptr(big), Lb.ptr(m193), None, Lb.ptr(c193), None, None, None, 1, 193, 1, 0.0, Lb.cur_stream()) != 0
    assert 'L 193' in lib.uniter_last_error().decode()
    # dropout without the keep flags drawn ahead; mask AND cu_seqlens; no output at all
    assert lib.uniter_attn_x3_fwd(Lb.ptr(dq), Lb.ptr(dm), None, Lb.ptr(ctx), None, None, None, B, L, nh, 0.1, Lb.cur_stream()) != 0
    assert 'keep flags' in lib.uniter_last_error().decode()
    cu = torch.tensor([0, 40], dtype=torch.int32).cuda()
    assert lib.uniter_attn_x3_fwd(Lb.ptr(dq), Lb.ptr(dm), Lb.ptr(cu), Lb.ptr(ctx), None, None, None, B, L, nh, 0.0, Lb.cur_stream()) != 0
    assert lib.uniter_attn_x3_fwd(Lb.ptr(dq), Lb.ptr(dm), None, None, None, None, None, B, L, nh, 0.0, Lb.cur_stream()) != 0
    torch.cuda.synchronize()
