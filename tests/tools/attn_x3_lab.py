"""Where the x3 attention backward spends its time: the lab build (-DUNITER_X3_LAB, UNITER_LIB_VARIANT=x3lab) compiles measurement
forms of attn_x3_bwd_kernel selected by UNITER_ATTN_X3_LAB (1 = no pass-1 loop, 2 = no pass-2 loop, 4 = fragments read but no MFMAs,
8 = Q, K, V staged from a [rows][3][ld] bf16 piece layout instead of split in the kernel: 6 instead of 4 bytes per value, no split
instructions -- the values are meaningless, the time is the real thing's)."""
import os, sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
B, Lq, nh = 16, int(os.environ.get('LAB_L', 164)), 12
H = nh * 64
p = 0.1
# (1.5 x the rows: forms 8 / 11 read the buffer as [rows][3][3H] bf16 pieces)
qkv = torch.randn(B * Lq * 3 // 2, 3 * H, device='cuda'); mask = torch.ones(B, Lq, device='cuda')
ctx = torch.empty(B * Lq, H, device='cuda'); lse = torch.empty(B, nh, Lq, device='cuda')
dctx = torch.randn(B * Lq, H, device='cuda'); delta = torch.empty(B, nh, Lq, device='cuda')
keep = torch.zeros(lib.uniter_attn_keep_bits_bytes(B, Lq, nh) // 2, dtype=torch.int16, device='cuda'); kp = L.ptr(keep)
L.check(lib.uniter_attn_keep_bits_gen(kp, 0, 1, B, Lq, nh, p, 1, 2, 3, 0, L.cur_stream()))
ctx3 = torch.empty(B * Lq, 3, H, dtype=torch.bfloat16, device='cuda'); dqkv3 = torch.empty(B * Lq, 3, 3 * H, dtype=torch.bfloat16, device='cuda')
part = torch.empty(B, 3 * H, device='cuda')
def fwdx(): L.check(lib.uniter_attn_x3_fwd(L.ptr(qkv), L.ptr(mask), None, L.ptr(ctx), L.ptr(ctx3), L.ptr(lse), kp, B, Lq, nh, p, L.cur_stream()))
def bwdx(): L.check(lib.uniter_attn_x3_bwd(L.ptr(qkv), L.ptr(mask), None, L.ptr(ctx), L.ptr(lse), L.ptr(dctx), 1, 0, None, L.ptr(dqkv3), L.ptr(part), kp, L.ptr(delta), B, Lq, nh, p, L.cur_stream()))
fwdx()
names = {0: 'complete', 1: 'no pass-1 loop', 2: 'no pass-2 loop', 3: 'no loops (staging, stores)', 4: 'no MFMAs', 5: 'pass 2 without MFMAs', 6: 'pass 1 without MFMAs',
         8: 'Q, K, V read as pieces', 11: 'no loops, Q, K, V as pieces'}
for lab in [int(x) for x in os.environ.get('LAB_FORMS', '0,1,2,3,4,5,6,8,11').split(',')]:
    os.environ['UNITER_ATTN_X3_LAB'] = str(lab)
    for _ in range(3): bwdx()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): bwdx()
    e1.record(); torch.cuda.synchronize()
    print('x3 bwd L=%d form %d %-28s %.1f us' % (Lq, lab, names[lab], e0.elapsed_time(e1) / 20 * 1e3), flush=True)
