import sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
B, Lq, nh = 16, 164, 12
H = nh * 64
for p in (0.0, 0.1):
    qkv = torch.randn(B * Lq, 3 * H, device='cuda'); mask = torch.ones(B, Lq, device='cuda')
    ctx = torch.empty(B * Lq, H, device='cuda'); lse = torch.empty(B, nh, Lq, device='cuda')
    dctx = torch.randn(B * Lq, H, device='cuda'); dqkv = torch.empty(B * Lq, 3 * H, device='cuda'); delta = torch.empty(B, nh, Lq, device='cuda')
    wsb = lib.uniter_attn_bwd_ws_bytes(B, Lq, nh); ws = torch.empty(max(wsb, 4) // 4, device='cuda')
    keep = torch.zeros(lib.uniter_attn_keep_bits_bytes(B, Lq, nh) // 2, dtype=torch.int16, device='cuda'); kp = L.ptr(keep)
    qkvb = qkv.bfloat16()
    def fwd(): L.check(lib.uniter_attn_fwd(L.ptr(qkv), L.ptr(mask), L.ptr(ctx), L.ptr(lse), B, Lq, nh, p, 1, 2, 3, L.cur_stream()))
    def bwd(): L.check(lib.uniter_attn_bwd(L.ptr(qkv), L.ptr(mask), L.ptr(ctx), L.ptr(lse), L.ptr(dctx), L.ptr(dqkv), L.ptr(delta), B, Lq, nh, p, 1, 2, 3, L.ptr(ws), wsb, L.cur_stream()))
    def fwd16(): L.check(lib.uniter_attn_bf16_fwd(L.ptr(qkvb), 1, L.ptr(mask), None, L.ptr(ctx), None, L.ptr(lse), kp, B, Lq, nh, p, 1, 2, 3, L.cur_stream()))
    def bwd16(): L.check(lib.uniter_attn_bf16_bwd(L.ptr(qkvb), 1, L.ptr(mask), None, L.ptr(ctx), L.ptr(lse), L.ptr(dctx), L.ptr(dqkv), None, None, kp, L.ptr(delta), B, Lq, nh, p, 1, 2, 3, L.ptr(ws), wsb, L.cur_stream()))
    if p > 0: L.check(lib.uniter_attn_keep_bits_gen(kp, 0, 1, B, Lq, nh, p, 1, 2, 3, 0, L.cur_stream()))
    ctx3 = torch.empty(B * Lq, 3, H, dtype=torch.bfloat16, device='cuda'); dqkv3 = torch.empty(B * Lq, 3, 3 * H, dtype=torch.bfloat16, device='cuda')
    part = torch.empty(B, 3 * H, device='cuda')
    def fwdx(): L.check(lib.uniter_attn_x3_fwd(L.ptr(qkv), L.ptr(mask), None, L.ptr(ctx), L.ptr(ctx3), L.ptr(lse), kp if p > 0 else None, B, Lq, nh, p, L.cur_stream()))
    def bwdx(): L.check(lib.uniter_attn_x3_bwd(L.ptr(qkv), L.ptr(mask), None, L.ptr(ctx), L.ptr(lse), L.ptr(dctx), 1, 0, None, L.ptr(dqkv3), L.ptr(part), kp if p > 0 else None, L.ptr(delta), B, Lq, nh, p, L.cur_stream()))
    def fwd3(): L.check(lib.uniter_attn_fwd_pre_x3(L.ptr(qkv), L.ptr(mask), None, L.ptr(ctx), L.ptr(ctx3), L.ptr(lse), kp, 1 if p > 0 else 0, B, Lq, nh, p, 1, 2, 3, L.cur_stream()))
    def bwd3(): L.check(lib.uniter_attn_bwd_ex_x3(L.ptr(qkv), L.ptr(mask), None, L.ptr(ctx), L.ptr(lse), L.ptr(dctx), None, L.ptr(dqkv3), L.ptr(part), kp, L.ptr(delta), B, Lq, nh, p, 1, 2, 3, L.ptr(ws), wsb, L.cur_stream()))
    def fwdbx(): L.check(lib.uniter_attn_b16x_fwd(L.ptr(qkvb), 1, L.ptr(mask), None, L.ptr(ctx), L.ptr(ctx3), L.ptr(lse), kp if p > 0 else None, B, Lq, nh, p, L.cur_stream()))
    def bwdbx(): L.check(lib.uniter_attn_b16x_bwd(L.ptr(qkvb), 1, L.ptr(mask), None, L.ptr(ctx), L.ptr(lse), L.ptr(dctx), None, L.ptr(dqkv3), L.ptr(part), kp if p > 0 else None, L.ptr(delta), B, Lq, nh, p, L.cur_stream()))
    for name, f in (('fwd', fwd), ('bwd(dq+dkv)', bwd), ('fwd pieces out', fwd3), ('bwd pieces out', bwd3), ('x3 fwd', fwdx), ('x3 bwd', bwdx), ('bf16 fwd', fwd16), ('bf16 bwd', bwd16), ('b16x fwd', fwdbx), ('b16x bwd', bwdbx)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print('p=%.1f %-15s %.1f us' % (p, name, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
