"""Pooler + classifier as one launch each way (csrc/head.hip: uniter_pool_head_fwd / _bwd) against the separate launches they
replace (uniter_pooler_fwd + uniter_linear_small_fwd; uniter_linear_small_bwd + uniter_pooler_bwd -- themselves checked against the
reference's golden logits and gradients in test_model_gpu.py) and against a float64 restatement of model/layer.py:179-185 followed by
model/meme_uniter.py:19-21."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from meme_challenge_amd import _lib
    return _lib


def _inputs(B, L, H, Cn, seed):
    g = torch.Generator().manual_seed(seed)
    hidden = torch.randn(B, L, H, generator=g)
    Wp = torch.randn(H, H, generator=g) / H ** 0.5
    bp = torch.randn(H, generator=g) * 0.1
    Wl = torch.randn(Cn, H, generator=g) / H ** 0.5
    bl = torch.randn(Cn, generator=g) * 0.1
    dlog = torch.randn(B, Cn, generator=g)
    return [t.cuda() for t in (hidden, Wp, bp, Wl, bl, dlog)]


def _separate(hidden, Wp, bp, Wl, bl, dlog, grads0):
    L_ = _lib()
    lib, ptr, cs = L_.lib(), L_.ptr, L_.cur_stream()
    B, L, H = hidden.shape
    Cn = Wl.shape[0]
    pooled = torch.empty(B, H, device='cuda')
    logits = torch.empty(B, Cn, device='cuda')
    L_.check(lib.uniter_pooler_fwd(ptr(hidden), ptr(Wp), ptr(bp), ptr(pooled), B, L, H, cs))
    L_.check(lib.uniter_linear_small_fwd(ptr(pooled), ptr(Wl), ptr(bl), ptr(logits), B, H, Cn, cs))
    dWp, dbp, dWl, dbl = [g.clone() for g in grads0]
    dpooled = torch.empty(B, H, device='cuda')
    dhidden = torch.zeros_like(hidden)
    L_.check(lib.uniter_linear_small_bwd(ptr(dlog), ptr(pooled), ptr(Wl), ptr(dpooled), ptr(dWl), ptr(dbl), B, H, Cn, cs))
    L_.check(lib.uniter_pooler_bwd(ptr(dpooled), ptr(pooled), ptr(hidden), ptr(Wp), ptr(dWp), ptr(dbp), ptr(dhidden), B, L, H, 0, cs))
    return pooled, logits, dWp, dbp, dWl, dbl, dhidden


def _fused(hidden, Wp, bp, Wl, bl, dlog, grads0, ticket, dhidden=None, beta=0):
    L_ = _lib()
    lib, ptr, cs = L_.lib(), L_.ptr, L_.cur_stream()
    B, L, H = hidden.shape
    Cn = Wl.shape[0]
    pooled = torch.full((B, H), float('nan'), device='cuda')
    logits = torch.full((B, Cn), float('nan'), device='cuda')
    L_.check(lib.uniter_pool_head_fwd(ptr(hidden), ptr(Wp), ptr(bp), ptr(Wl), ptr(bl), ptr(pooled), ptr(logits), ptr(ticket),
                                      B, L, H, Cn, cs))
    dWp, dbp, dWl, dbl = [g.clone() for g in grads0]
    if dhidden is None:
        dhidden = torch.zeros_like(hidden)
    L_.check(lib.uniter_pool_head_bwd(ptr(dlog), ptr(pooled), ptr(hidden), ptr(Wp), ptr(Wl), ptr(dWp), ptr(dbp), ptr(dWl), ptr(dbl),
                                      ptr(dhidden), B, L, H, Cn, beta, cs))
    return pooled, logits, dWp, dbp, dWl, dbl, dhidden


@pytest.mark.parametrize('B,L,H,Cn', [(16, 164, 768, 1), (5, 7, 768, 3), (1, 3, 64, 1), (9, 4, 72, 2), (70, 2, 128, 4), (16, 50, 1024, 1)])
def test_one_launch_each_way_equals_the_separate_launches(B, L, H, Cn):
    hidden, Wp, bp, Wl, bl, dlog = _inputs(B, L, H, Cn, seed=B * 131 + H)
    g = torch.Generator().manual_seed(3)
    grads0 = [torch.randn(*s, generator=g).cuda() for s in ((H, H), (H,), (Cn, H), (Cn,))]      # the launches ACCUMULATE into these
    ticket = torch.zeros(17 * 64, dtype=torch.int32, device='cuda')       # UNITER_POOL_HEAD_TICKET_WORDS
    ref = _separate(hidden, Wp, bp, Wl, bl, dlog, grads0)
    for rep in range(3):                       # (the ticket counter is left zero by every launch)
        got = _fused(hidden, Wp, bp, Wl, bl, dlog, grads0, ticket)
        assert int(ticket.abs().sum().item()) == 0
        names = ('pooled', 'logits', 'dWp', 'dbp', 'dWl', 'dbl', 'dhidden')
        for nm, a, b in zip(names, got, ref):
            assert torch.isfinite(a).all(), nm
            # same sums in the same order; the compiler may contract a multiply-add differently in the two kernels
            torch.testing.assert_close(a, b, rtol=2e-6, atol=2e-6, msg=lambda m, nm=nm: '%s (launch %d): %s' % (nm, rep, m))
    # float64 restatement
    h0 = hidden[:, 0].double()
    pooled64 = torch.tanh(h0 @ Wp.double().t() + bp.double())
    logits64 = pooled64 @ Wl.double().t() + bl.double()
    dpooled = dlog.double() @ Wl.double()
    dpre = dpooled * (1 - pooled64 ** 2)
    torch.testing.assert_close(got[0].double(), pooled64, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got[1].double(), logits64, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(got[2].double(), grads0[0].double() + dpre.t() @ h0, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got[3].double(), grads0[1].double() + dpre.sum(0), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got[4].double(), grads0[2].double() + dlog.double().t() @ pooled64, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got[5].double(), grads0[3].double() + dlog.double().sum(0), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(got[6][:, 0].double(), dpre @ Wp.double(), rtol=1e-4, atol=1e-4)
    assert float(got[6][:, 1:].abs().max()) == 0.0 if L > 1 else True


def test_backward_assigns_or_adds_row_zero_and_leaves_the_other_rows():
    B, L, H, Cn = 4, 5, 128, 1
    hidden, Wp, bp, Wl, bl, dlog = _inputs(B, L, H, Cn, seed=11)
    grads0 = [torch.zeros(H, H).cuda(), torch.zeros(H).cuda(), torch.zeros(Cn, H).cuda(), torch.zeros(Cn).cuda()]
    ticket = torch.zeros(17 * 64, dtype=torch.int32, device='cuda')       # UNITER_POOL_HEAD_TICKET_WORDS
    base = torch.randn(B, L, H).cuda()
    ref = _fused(hidden, Wp, bp, Wl, bl, dlog, grads0, ticket)[6]
    assigned = _fused(hidden, Wp, bp, Wl, bl, dlog, grads0, ticket, dhidden=base.clone(), beta=0)[6]
    added = _fused(hidden, Wp, bp, Wl, bl, dlog, grads0, ticket, dhidden=base.clone(), beta=1)[6]
    assert torch.equal(assigned[:, 0], ref[:, 0]) and torch.equal(assigned[:, 1:], base[:, 1:])
    torch.testing.assert_close(added[:, 0], base[:, 0] + ref[:, 0], rtol=1e-6, atol=1e-6)
    assert torch.equal(added[:, 1:], base[:, 1:])


def test_bad_arguments_fail_loudly():
    L_ = _lib()
    lib = L_.lib()
    assert lib.uniter_pool_head_fwd(None, None, None, None, None, None, None, None, 1, 1, 1, 1, None) != 0
    assert lib.uniter_pool_head_bwd(None, None, None, None, None, None, None, None, None, None, 1, 1, 1, 1, 0, None) != 0


@pytest.mark.parametrize('train', [False, True])
def test_memeuniter_with_the_fused_head_equals_the_separate_modules(train, monkeypatch):
    """MemeUniter.forward (model/meme_uniter.py:17-21): logits, loss and every gradient with the fused head against
    UNITER_FUSED_HEAD=0, same weights, same dropout stream; two steps (the cached zero gradient of the encoder output is reused)."""
    from common import TINY, TINY_IMG_DIM
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import bce_with_logits_loss, TrainStep
    from meme_challenge_amd.utils import make_synthetic_batch
    cfg = UniterConfig.from_dict(TINY)
    torch.manual_seed(5)
    model = MemeUniter(UniterModel(cfg, img_dim=TINY_IMG_DIM), cfg.hidden_size, 1).cuda()
    model.train(train)
    enc = model.uniter_model
    batch = make_synthetic_batch(4, 12, 5, seed=2, vocab=TINY['vocab_size'], img_dim=TINY_IMG_DIM, device='cuda')
    kw = TrainStep.forward_kwargs(batch)

    def run(fused):
        monkeypatch.setenv('UNITER_FUSED_HEAD', '1' if fused else '0')
        out = []
        for step in range(2):
            enc.set_dropout_seed(7, step)
            st = model.param_store()
            st.zero_grads()
            logits = model(**kw)
            loss = bce_with_logits_loss(logits, batch['labels'], 1.8)
            loss.backward()
            torch.cuda.synchronize()
            out.append((logits.detach().clone(), loss.detach().clone(), st.flat_grads.clone()))
        return out

    a, b = run(True), run(False)
    for (l1, s1, g1), (l2, s2, g2) in zip(a, b):
        torch.testing.assert_close(l1, l2, rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(s1, s2, rtol=2e-6, atol=2e-6)
        torch.testing.assert_close(g1, g2, rtol=2e-5, atol=2e-6)
        assert float(g1.abs().max()) > 0
