"""Diagnostic: uniter_attn_bf16_fwd against oracle._online_softmax_pv_b16 on the same bf16 q, k, v."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as Lb
from oracle import uniter_oracle as O, philox
lib = Lb.lib()
for B, L, nh, p in ((2, 100, 2, 0.0), (2, 164, 12, 0.0), (2, 164, 2, 0.1), (2, 178, 2, 0.0)):
    H = nh * 64
    g = torch.Generator().manual_seed(L)
    qkv = torch.randn(B * L, 3 * H, generator=g).bfloat16()
    mask = torch.ones(B, L); mask[1, L - 7:] = 0
    ctx = torch.zeros(B * L, H, device='cuda'); ctxb = torch.zeros(B * L, H, dtype=torch.bfloat16, device='cuda')
    lse = torch.zeros(B, nh, L, device='cuda')
    qd, md = qkv.cuda(), mask.cuda()
    seed, offset, site = 0xBEEF1234, 7, 10
    Lb.check(lib.uniter_attn_bf16_fwd(Lb.ptr(qd), 1, Lb.ptr(md), None, Lb.ptr(ctx), Lb.ptr(ctxb), Lb.ptr(lse), None, B, L, nh, p,
                                      seed, offset, site, Lb.cur_stream()))
    torch.cuda.synchronize()
    q, k, v = [t.float().view(B, L, nh, 64).permute(0, 2, 1, 3) for t in qkv.split(H, dim=1)]
    ext = ((1.0 - mask) * -10000.0).view(B, 1, 1, L)
    s = torch.matmul(q, k.transpose(-1, -2)) * 0.125 + ext
    keep = None
    if p > 0:
        Lp = (L + 3) // 4 * 4
        idx = (torch.arange(B * nh * L).view(B, nh, L, 1) * Lp + torch.arange(L).view(1, 1, 1, L))
        keep = torch.from_numpy(philox.keep_mask(int(idx.max()) + 1, p, seed, offset, site))[idx.reshape(-1)].view(idx.shape).float()
        keep = keep * (torch.tensor(1.0) / torch.tensor(1.0 - p))
    o, pr = O._online_softmax_pv_b16(s, keep, v)
    o2 = torch.matmul(O.bf(pr if keep is None else pr * keep), v)
    back = lambda t: t.permute(0, 2, 1, 3).reshape(B * L, H)
    got = ctx.cpu()
    print('B%d L%d nh%d p%.2f: |hip - online emul| max %.3e  |hip - normalised rounding| max %.3e   ctx max %.2f' % (
        B, L, nh, p, (got - back(o)).abs().max().item(), (got - back(o2)).abs().max().item(), got.abs().max().item()))
