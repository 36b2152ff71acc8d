"""Where the time of the fp32 attention backward (dQ kernel) goes: per-workgroup phase clocks of an instrumented
variant build (-DATTN_STAMPS).  Usage on the GPU box:
  UNITER_EXTRA_HIPCC_FLAGS=-DATTN_STAMPS python -c "from meme_challenge_amd import build; build.build(variant='stamps')"
  UNITER_LIB_VARIANT=stamps UNITER_DEV_PARTIAL_LIB=1 python tests/tools/attn_phase_lab.py"""
import ctypes, sys, numpy as np, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib as L
lib = L.lib()
B, Lq, nh, H = 16, 164, 12, 768
g = torch.Generator(device='cuda').manual_seed(0)
qkv = torch.randn(B * Lq, 3 * H, device='cuda', generator=g)
mask = torch.ones(B, Lq, device='cuda')
ctx = torch.empty(B * Lq, H, device='cuda'); lse = torch.empty(B, nh, Lq, device='cuda')
keep = torch.zeros(lib.uniter_attn_keep_bits_bytes(B, Lq, nh), dtype=torch.uint8, device='cuda')
L.check(lib.uniter_attn_fwd_ex(L.ptr(qkv), L.ptr(mask), None, L.ptr(ctx), None, L.ptr(lse), L.ptr(keep), B, Lq, nh, 0.1, 1, 0, 3, L.cur_stream()))
dctx = torch.randn_like(ctx); dqkv = torch.empty_like(qkv); delta = torch.empty(B, nh, Lq, device='cuda')
part = torch.empty(B, 3 * H, device='cuda')
wsb = lib.uniter_attn_bwd_ws_bytes(B, Lq, nh); ws = torch.empty(wsb, dtype=torch.uint8, device='cuda')
def bwd():
    L.check(lib.uniter_attn_bwd_ex(L.ptr(qkv), L.ptr(mask), None, L.ptr(ctx), L.ptr(lse), L.ptr(dctx), L.ptr(dqkv), None, L.ptr(part),
                                   L.ptr(keep), L.ptr(delta), B, Lq, nh, 0.1, 1, 0, 3, L.ptr(ws), wsb, L.cur_stream()))
for _ in range(5): bwd()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); [bwd() for _ in range(20)]; e1.record(); torch.cuda.synchronize()
print('dq + dkv: %.1f us per call' % (e0.elapsed_time(e1) / 20 * 1e3))
lib.uniter_dbg_attn_stamps.restype = ctypes.c_int
out = np.zeros(3 * 8 * 1024, dtype=np.uint64)
L.check(lib.uniter_dbg_attn_stamps(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.size)))
st = out.reshape(3, 1024, 8)[1, :B * nh, :7].astype(np.int64)      # dq kernel, 100 MHz clock
t0 = st[:, 0].min()
rel = (st - t0) / 100.0
names = ['start', 'loads issued', 'staged (barrier)', 'loop done', 'barrier', 'exchange', 'end']
for k, n in enumerate(names):
    print('%-18s median %6.2f us   min %6.2f   max %6.2f' % (n, np.median(rel[:, k]), rel[:, k].min(), rel[:, k].max()))
d = np.diff(rel, axis=1)
print('phase durations (median us):', ['%.2f' % x for x in np.median(d, axis=0)])
