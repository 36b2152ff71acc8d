"""GPU: attention on the bf16 matrix pipe (uniter_attn_bf16_fwd/bwd) against a float64 reference that
applies the same roundings (q, k, v to bf16; probabilities and score gradients to bf16 where they feed
a product) and the same Philox dropout mask."""
import numpy as np
import pytest
import torch

from oracle import philox

pytestmark = pytest.mark.gpu


def _r(x):
    return x.float().bfloat16().double()


def _reference(qkv, dctx, lens, B, L, nh, p, seed, offset, site, emulate):
    H = nh * 64
    q, k, v = [_r(t).view(B, L, nh, 64).permute(0, 2, 1, 3) for t in qkv.split(H, dim=1)]
    do = _r(dctx).view(B, L, nh, 64).permute(0, 2, 1, 3)
    valid = torch.zeros(B, L, dtype=torch.bool)
    for b, n in enumerate(lens):
        valid[b, :n] = True
    do = do * valid.view(B, 1, L, 1)          # padded queries carry no gradient (the model never reads them)
    s = q @ k.transpose(-1, -2) / 8.0
    s = s.masked_fill(~valid.view(B, 1, 1, L), float('-inf'))
    lse = torch.logsumexp(s, -1)
    pr = torch.softmax(s, -1)
    keep = torch.ones_like(pr)
    if p > 0:
        Lp = (L + 3) // 4 * 4
        idx = (torch.arange(B * nh * L).view(B, nh, L, 1) * Lp + torch.arange(L).view(1, 1, 1, L))
        keep = torch.from_numpy(philox.keep_mask(int(idx.max()) + 1, p, seed, offset, site))[idx.reshape(-1)].view(idx.shape).double()
        keep = keep * float(np.float32(1.0) / np.float32(1.0 - p))
    pd = pr * keep
    if emulate:
        # P rounded as the MFMA operand: the kernel rounds exp(s - running max) block by block (oracle's restatement;
        # padded batches only: with cu_seqlens the two key halves are cut per sample)
        from oracle.uniter_oracle import _online_softmax_pv_b16
        ctx = _online_softmax_pv_b16(s.float(), keep.float() if p > 0 else None, v.float())[0].double()
    else:
        ctx = _r(pd) @ v
    delta = (ctx * do).sum(-1, keepdim=True)
    dp = do @ v.transpose(-1, -2)
    ds = pr * (dp * keep - delta) / 8.0
    dq = _r(ds) @ k
    dk = _r(ds).transpose(-1, -2) @ q
    dv = _r(pd).transpose(-1, -2) @ do
    back = lambda t: t.permute(0, 2, 1, 3).reshape(B * L, H)
    return back(ctx), lse, torch.cat([back(dq), back(dk), back(dv)], dim=1), valid.view(-1)


@pytest.mark.parametrize('B,L,nh,p,lens,varlen', [
    (2, 164, 2, 0.0, [164, 164], False), (2, 164, 2, 0.1, [164, 90], False), (3, 100, 12, 0.1, [100, 1, 37], False),
    (2, 20, 1, 0.25, [20, 7], False), (3, 164, 2, 0.1, [164, 40, 97], True), (2, 192, 1, 0.0, [192, 130], True)])
@pytest.mark.parametrize('kind', ['bf16', 'b16x'])
def test_attention_bf16_fwd_bwd(B, L, nh, p, lens, varlen, kind):
    """kind 'bf16': csrc/attention_bf16.hip (two waves per 32-row block, Pd / dS through a scratch); 'b16x': the same arithmetic in
    csrc/attention_x3.hip's decomposition (one wave per 16 rows, no scratch, keep flags drawn ahead) -- one reference, one set of
    tolerances."""
    from meme_challenge_amd import _lib as Lb
    lib = Lb.lib()
    H = nh * 64
    g = torch.Generator().manual_seed(31 * B + L + nh)
    qkv = torch.randn(B * L, 3 * H, generator=g)
    dctx = torch.randn(B * L, H, generator=g)
    seed, offset, site = 0xBEEF1234, 7, 10
    ctx_ref, lse_ref, dqkv_ref, valid = _reference(qkv, dctx, lens, B, L, nh, p, seed, offset, site, emulate=not varlen)
    rows = torch.nonzero(valid).view(-1)
    mask = valid.view(B, L).float()
    if varlen:
        qd, dd = qkv[rows].cuda().contiguous(), dctx[rows].cuda().contiguous()
        cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32).cuda()
        mptr, cptr = None, Lb.ptr(cu)
    else:
        qd, dd = qkv.cuda(), (dctx * mask.view(-1, 1)).cuda()
        md = mask.cuda()
        mptr, cptr = Lb.ptr(md), None
    M = qd.shape[0]
    ctx = torch.zeros(M, H, device='cuda'); ctxb = torch.zeros(M, H, dtype=torch.bfloat16, device='cuda')
    lse = torch.zeros(B, nh, L, device='cuda'); delta = torch.zeros(B, nh, L, device='cuda')
    dqkv = torch.zeros(M, 3 * H, device='cuda'); dqkvb = torch.zeros(M, 3 * H, dtype=torch.bfloat16, device='cuda')
    bpart = torch.full((B, 3 * H), float('nan'), device='cuda')
    wsb = lib.uniter_attn_bf16_bwd_ws_bytes(B, L, nh)
    ws = torch.full((max(wsb, 4) // 2,), float('nan'), dtype=torch.bfloat16, device='cuda')
    keep = torch.zeros(lib.uniter_attn_keep_bits_bytes(B, L, nh) // 2, dtype=torch.int16, device='cuda')

    if kind == 'b16x' and p > 0:
        Lb.check(lib.uniter_attn_keep_bits_gen(Lb.ptr(keep), 0, 1, B, L, nh, p, seed, offset, site, 0, Lb.cur_stream()))

    def run_x(kp):
        kp = Lb.ptr(keep) if p > 0 else None
        Lb.check(lib.uniter_attn_b16x_fwd(Lb.ptr(qsrc), qb16, mptr, cptr, Lb.ptr(ctx), Lb.ptr(ctxb), Lb.ptr(lse), kp, B, L, nh, p,
                                          Lb.cur_stream()))
        Lb.check(lib.uniter_attn_b16x_bwd(Lb.ptr(qsrc), qb16, mptr, cptr, Lb.ptr(ctx), Lb.ptr(lse), Lb.ptr(dd), Lb.ptr(dqkv),
                                          Lb.ptr(dqkvb), Lb.ptr(bpart), kp, Lb.ptr(delta), B, L, nh, p, Lb.cur_stream()))
        torch.cuda.synchronize()

    def run(kp):
        if kind == 'b16x':
            return run_x(kp)
        Lb.check(lib.uniter_attn_bf16_fwd(Lb.ptr(qsrc), qb16, mptr, cptr, Lb.ptr(ctx), Lb.ptr(ctxb), Lb.ptr(lse), kp, B, L, nh, p,
                                          seed, offset, site, Lb.cur_stream()))
        Lb.check(lib.uniter_attn_bf16_bwd(Lb.ptr(qsrc), qb16, mptr, cptr, Lb.ptr(ctx), Lb.ptr(lse), Lb.ptr(dd), Lb.ptr(dqkv),
                                          Lb.ptr(dqkvb), Lb.ptr(bpart), kp, Lb.ptr(delta), B, L, nh, p, seed, offset, site,
                                          Lb.ptr(ws), wsb, Lb.cur_stream()))
        torch.cuda.synchronize()

    qsrc, qb16 = qd, 0
    run(None)                       # dQ evaluates Philox again
    plain = (ctx.clone(), dqkv.clone())
    run(Lb.ptr(keep))               # dQ reads the keep flags the forward pass stored: the same masks, bit for bit
    assert torch.equal(plain[0], ctx) and torch.equal(plain[1], dqkv)
    # Q, K, V handed over as bf16 (the QKV GEMM's bf16 output): the kernels round fp32 input to exactly these values
    qsrc, qb16 = qd.bfloat16(), 1
    run(Lb.ptr(keep))
    assert torch.equal(plain[0], ctx) and torch.equal(plain[1], dqkv)
    # bf16-only output (what the model asks for: nothing reads the fp32 gradient in precision mode 2)
    only_b = torch.zeros_like(dqkvb)
    if kind == 'b16x':
        Lb.check(lib.uniter_attn_b16x_bwd(Lb.ptr(qsrc), qb16, mptr, cptr, Lb.ptr(ctx), Lb.ptr(lse), Lb.ptr(dd), None, Lb.ptr(only_b),
                                          Lb.ptr(bpart), Lb.ptr(keep) if p > 0 else None, Lb.ptr(delta), B, L, nh, p, Lb.cur_stream()))
    else:
      Lb.check(lib.uniter_attn_bf16_bwd(Lb.ptr(qsrc), qb16, mptr, cptr, Lb.ptr(ctx), Lb.ptr(lse), Lb.ptr(dd), None, Lb.ptr(only_b),
                                      Lb.ptr(bpart), Lb.ptr(keep), Lb.ptr(delta), B, L, nh, p, seed, offset, site,
                                      Lb.ptr(ws), wsb, Lb.cur_stream()))
    torch.cuda.synchronize()
    assert torch.equal(only_b, dqkvb)
    torch.cuda.synchronize()
    sel = slice(None) if varlen else rows
    got_ctx, got_d = ctx.cpu().double()[sel], dqkv.cpu().double()[sel]
    assert torch.isfinite(got_ctx).all() and torch.isfinite(got_d).all()
    # the reference rounds the UNNORMALISED exp(s - running max) block by block as the kernel does; what is left are
    # the few probabilities whose rounding flips on the last bits of the fast exponential
    # (with cu_seqlens the reference rounds the normalised probabilities: bf16-accurate, not the same bits)
    cmax, cmean = (8e-3, 1e-3) if varlen else (1e-3, 2e-5)
    assert (got_ctx - ctx_ref[rows]).abs().max() < cmax * max(1.0, ctx_ref[rows].abs().max().item())
    assert (got_ctx - ctx_ref[rows]).abs().mean() < cmean
    for b, n in enumerate(lens):
        assert (lse.cpu().double()[b, :, :n] - lse_ref[b, :, :n]).abs().max() < 2e-4
    ref_d = dqkv_ref[rows]
    for name, sl in (('dq', slice(0, H)), ('dk', slice(H, 2 * H)), ('dv', slice(2 * H, 3 * H))):
        err = (got_d[:, sl] - ref_d[:, sl]).abs().max().item()
        assert err < 1.5e-2 * max(1.0, ref_d[:, sl].abs().max().item()), (name, err)
    # per-sample column sums of dqkv (the fused QKV-bias gradient before the sum over the batch)
    full = dqkv.cpu().double()
    start = 0
    for bb, n in enumerate(lens):
        blk = full[start:start + n] if varlen else full[bb * L:(bb + 1) * L]
        start += n
        assert (bpart.cpu().double()[bb] - blk.sum(0)).abs().max() < 1e-3 * max(1.0, blk.abs().sum(0).max().item()), bb
    # the bf16 copies are the rounded fp32 outputs
    assert torch.equal(ctxb.cpu()[sel], ctx.cpu()[sel].bfloat16())
    assert torch.equal(dqkvb.cpu()[sel], dqkv.cpu()[sel].bfloat16())
