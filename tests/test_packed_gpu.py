"""GPU: token packing -- computing the valid positions only must reproduce the
padded path at every valid position, in every parameter gradient, and against the reference's
golden vectors for the ragged BASELINE config-1 batch."""
import numpy as np
import pytest
import torch

from oracle import uniter_oracle as O
from common import TINY, TINY_IMG_DIM, BASE, sd_from_npz, batch_from_npz, model_kwargs, maxdiff
from test_model_gpu import build, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,L,nh,p,lens', [(3, 164, 2, 0.0, [164, 40, 97]), (3, 164, 2, 0.1, [164, 40, 97]),
                                           (4, 100, 12, 0.1, [1, 33, 100, 64]), (2, 20, 1, 0.2, [20, 7])])
def test_varlen_attention_equals_padded_attention(B, L, nh, p, lens):
    from meme_challenge_amd import _lib as Lb
    lib = Lb.lib()
    H = nh * 64
    g = torch.Generator().manual_seed(11 * B + L)
    qkv = torch.randn(B * L, 3 * H, generator=g)
    dctx = torch.randn(B * L, H, generator=g)
    mask = torch.zeros(B, L)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    rows = torch.cat([torch.arange(n) + b * L for b, n in enumerate(lens)])
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    seed, offset, site = 77, 3, 6

    def run(varlen):
        q = (qkv[rows] if varlen else qkv).cuda().contiguous()
        d = (dctx[rows] if varlen else dctx).cuda().contiguous()
        M = q.shape[0]
        ctx = torch.zeros(M, H, device='cuda'); lse = torch.zeros(B, nh, L, device='cuda')
        dqkv = torch.zeros(M, 3 * H, device='cuda'); delta = torch.zeros(B, nh, L, device='cuda')
        wsb = lib.uniter_attn_bwd_ws_bytes(B, L, nh)
        ws = torch.full((max(wsb, 4) // 4,), float('nan'), device='cuda')
        if varlen:
            cud = cu.cuda()
            Lb.check(lib.uniter_attn_fwd_varlen(Lb.ptr(q), Lb.ptr(cud), Lb.ptr(ctx), Lb.ptr(lse), B, L, nh, p, seed,
                                                offset, site, Lb.cur_stream()))
            Lb.check(lib.uniter_attn_bwd_varlen(Lb.ptr(q), Lb.ptr(cud), Lb.ptr(ctx), Lb.ptr(lse), Lb.ptr(d),
                                                Lb.ptr(dqkv), Lb.ptr(delta), B, L, nh, p, seed, offset, site,
                                                Lb.ptr(ws), wsb, Lb.cur_stream()))
        else:
            m = mask.cuda()
            Lb.check(lib.uniter_attn_fwd(Lb.ptr(q), Lb.ptr(m), Lb.ptr(ctx), Lb.ptr(lse), B, L, nh, p, seed, offset,
                                         site, Lb.cur_stream()))
            # the padded path also back-propagates the (ignored) padded rows' dctx: zero them, as the model does
            d = d * m.reshape(-1, 1)
            Lb.check(lib.uniter_attn_bwd(Lb.ptr(q), Lb.ptr(m), Lb.ptr(ctx), Lb.ptr(lse), Lb.ptr(d), Lb.ptr(dqkv),
                                         Lb.ptr(delta), B, L, nh, p, seed, offset, site, Lb.ptr(ws), wsb,
                                         Lb.cur_stream()))
        torch.cuda.synchronize()
        return ctx.cpu(), dqkv.cpu()

    ctx_p, dq_p = run(False)
    ctx_v, dq_v = run(True)
    assert torch.isfinite(ctx_v).all() and torch.isfinite(dq_v).all()
    assert (ctx_v - ctx_p[rows]).abs().max() < 2e-5
    assert (dq_v - dq_p[rows]).abs().max() < 2e-4


def _grads(m):
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def test_packed_tiny_model_equals_padded_everywhere_valid(tiny):
    from meme_challenge_amd.trainer import bce_with_logits_loss
    sd = sd_from_npz(tiny)
    b = to_dev(batch_from_npz(tiny))
    lens = b['attn_mask'].sum(1).long().tolist()
    assert min(lens) < b['attn_mask'].shape[1]          # the fixture is ragged
    out = {}
    for packed in (False, True):
        m = build(TINY, TINY_IMG_DIM, sd).eval()
        m.uniter_model.pack_padded = packed
        kw = model_kwargs(b)
        kw['output_all_encoded_layers'] = True
        kw['seq_lens'] = lens
        layers = m.uniter_model(**kw)                      # one forward: the module keeps one set of activations
        logits = m.linear(m.uniter_model.pooler(layers[-1]))
        loss = bce_with_logits_loss(logits, b['labels'], 1.8)
        loss.backward()
        torch.cuda.synchronize()
        out[packed] = ([l.detach().clone() for l in layers], logits.detach().clone(), _grads(m))
    valid = b['attn_mask'].bool()
    for lp, lv in zip(out[False][0], out[True][0]):
        assert maxdiff(lv[valid], lp[valid]) < 2e-5
        assert lv[~valid].abs().max().item() == 0.0      # padded positions come back as zeros
    assert maxdiff(out[True][1], out[False][1]) < 1e-5
    assert maxdiff(out[True][1], tiny['out/logits']) < 1e-5          # and still the reference's logits
    for n, gp in out[False][2].items():
        gv = out[True][2][n]
        tol = 2e-6 + 2e-4 * gp.abs().max().item()
        assert maxdiff(gv, gp) <= tol, (n, maxdiff(gv, gp), tol)
    for n in out[True][2]:
        ref = torch.from_numpy(tiny['grad/' + n])
        assert maxdiff(out[True][2][n], ref) <= 2e-6 + 2e-4 * ref.abs().max().item(), n


def test_packed_all_layer_gradients_flow(tiny):
    """output_all_encoded_layers=True with a loss on every layer: the padded gradient is gathered to the
    valid rows layer by layer; gradients placed on padded positions are ignored."""
    sd = sd_from_npz(tiny)
    b = to_dev(batch_from_npz(tiny))
    valid = b['attn_mask'].bool().unsqueeze(-1)
    res = {}
    for packed in (False, True):
        m = build(TINY, TINY_IMG_DIM, sd).eval()
        m.uniter_model.pack_padded = packed
        kw = model_kwargs(b)
        kw['output_all_encoded_layers'] = True
        kw['seq_lens'] = m.uniter_model.lengths_from_mask(b['attn_mask'])
        layers = m.uniter_model(**kw)
        loss = sum(((l * valid) ** 2).sum() * (0.1 + i) for i, l in enumerate(layers))
        loss.backward()
        torch.cuda.synchronize()
        res[packed] = _grads(m)
    for n, gp in res[False].items():
        assert maxdiff(res[True][n], gp) <= 1e-5 + 3e-4 * gp.abs().max().item(), n


def test_packed_base_ragged_matches_reference_golden(shapes_base):
    from meme_challenge_amd.trainer import bce_with_logits_loss
    from meme_challenge_amd.utils import make_synthetic_batch
    z, name = shapes_base, 'cfg1_ragged'
    sd = O.synth_state_dict(BASE, seed=0, ln_jitter=0.02)
    m = build(BASE, 2048, sd).eval()
    m.uniter_model.pack_padded = True
    B, T, R, seed = z[name + '/shape'].tolist()
    tl, nbb = z[name + '/txt_lens'].tolist(), z[name + '/num_bbs'].tolist()
    b = make_synthetic_batch(B, T, R, seed=seed, txt_lens=tl, num_bbs=nbb, device='cuda')
    kw = model_kwargs(b)
    logits = m(seq_lens=[a + c for a, c in zip(tl, nbb)], **kw)         # host lengths: no device sync
    assert maxdiff(logits, z[name + '/logits']) < 5e-5
    loss = bce_with_logits_loss(logits, b['labels'], 1.8)
    loss.backward()
    torch.cuda.synchronize()
    params = dict(m.named_parameters())
    for n, ref in zip(list(z['param_names']), z[name + '/grad_norms']):
        got = params[n].grad.double().norm().item()
        assert abs(got - ref) <= 1e-6 + 2e-3 * ref, (n, got, ref)


def test_packed_training_step_runs_and_tracks_padded_loss():
    """Dropout on: the two layouts draw different masks (the Philox element index is the row in
    the computed layout), so only the statistics agree -- the loss trajectories stay close."""
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler
    from meme_challenge_amd.utils import make_synthetic_batch
    cfg = UniterConfig.from_dict(TINY)
    config = dict(optimizer='adam', lr=1e-3, beta1=0.9, beta2=0.999, weight_decay=1e-3, gradient_accumulation=1,
                  max_grad_norm=5, pos_wt=1.8, loss_func='bce_logits', scheduler='warmup_cosine', warmup_steps=2,
                  max_epoch=2)
    b = make_synthetic_batch(8, 24, 10, seed=3, vocab=TINY['vocab_size'], img_dim=TINY_IMG_DIM,
                             txt_lens=[24, 5, 9, 17, 3, 24, 11, 8], num_bbs=[10, 4, 10, 2, 7, 1, 10, 5], device='cuda')
    losses = {}
    for packed in (False, True):
        torch.manual_seed(0)
        m = MemeUniter(UniterModel(cfg, img_dim=TINY_IMG_DIM), cfg.hidden_size, 1).cuda().train()
        m.uniter_model.pack_padded = packed
        m.uniter_model.set_dropout_seed(5, 0)
        opt = FusedAdam(m, lr=config['lr'], weight_decay=config['weight_decay'])
        step = TrainStep(m, opt, get_scheduler(opt, config, steps_per_epoch=10), config)
        ls = []
        for it in range(8):
            step.train_iter(b, iters=it)
            ls.append(float(step.last_loss.item()))
        losses[packed] = ls
    assert all(np.isfinite(losses[True]))
    assert abs(losses[True][-1] - losses[False][-1]) < 0.25
    assert losses[True][-1] < losses[True][0]


@pytest.mark.parametrize('precision', ['fp32', 'bf16'])
@pytest.mark.parametrize('mode', ['txt', 'img', 'joint', 'joint_masks'])
def test_single_modality_and_mask_inputs_in_every_layout(tiny, precision, mode):
    """Text-only / image-only / joint (+ img_masks) inputs through the padded and the packed layout, in fp32 and
    in the bf16-resident mode: packed == padded at the valid positions (fp32: round-off; bf16: bf16 accuracy)."""
    sd = sd_from_npz(tiny)
    b = to_dev(batch_from_npz(tiny))
    B, T = b['input_ids'].shape
    R = b['img_feat'].shape[1]
    tl = (b['input_ids'] != 0).sum(1)
    nbb = b['attn_mask'].sum(1).long() - tl
    ar = lambda n: torch.arange(n, device='cuda').unsqueeze(0)
    if mode == 'txt':
        kw = dict(input_ids=b['input_ids'], position_ids=b['position_ids'], img_feat=None, img_pos_feat=None,
                  attention_mask=(ar(T) < tl.unsqueeze(1)).float())
    elif mode == 'img':
        kw = dict(input_ids=None, position_ids=None, img_feat=b['img_feat'], img_pos_feat=b['img_pos_feat'],
                  attention_mask=(ar(R) < nbb.unsqueeze(1)).float())
    else:
        kw = model_kwargs(b)
        kw.pop('output_all_encoded_layers')
        if mode == 'joint_masks':
            kw['img_masks'] = b['img_masks']
    outs = {}
    for packed in (False, True):
        m = build(TINY, TINY_IMG_DIM, sd).eval()
        m.uniter_model.precision = precision
        m.uniter_model.pack_padded = packed
        h = m.uniter_model(output_all_encoded_layers=False, seq_lens=m.uniter_model.lengths_from_mask(kw['attention_mask']), **kw)
        (h * h).sum().backward()
        torch.cuda.synchronize()
        outs[packed] = (h.detach().clone(), _grads(m))
    valid = kw['attention_mask'].bool()
    tol = 2e-5 if precision == 'fp32' else 3e-2
    assert maxdiff(outs[True][0][valid], outs[False][0][valid]) < tol * max(1.0, outs[False][0].abs().max().item())
    assert outs[True][0][~valid].abs().max().item() == 0.0 if (~valid).any() else True
    # the loss above also sums the padded positions of the padded layout, so only compare where both define it:
    # gradients flow from valid rows only once the padded rows are excluded
    for packed in (False, True):
        m = build(TINY, TINY_IMG_DIM, sd).eval()
        m.uniter_model.precision = precision
        m.uniter_model.pack_padded = packed
        h = m.uniter_model(output_all_encoded_layers=False, seq_lens=m.uniter_model.lengths_from_mask(kw['attention_mask']), **kw)
        ((h * valid.unsqueeze(-1)) ** 2).sum().backward()
        torch.cuda.synchronize()
        outs[packed] = _grads(m)
    gtol = 3e-4 if precision == 'fp32' else 6e-2
    for n, gp in outs[False].items():
        assert maxdiff(outs[True][n], gp) <= 1e-5 + gtol * max(gp.abs().max().item(), 1e-3), (n, maxdiff(outs[True][n], gp))


def test_packing_needs_host_lengths(tiny):
    """No hidden device -> host synchronisation: the packed layout's row count sizes the workspace on the host, so the
    lengths are an input (batch['seq_lens']); asking for packing without them raises instead of reading the mask back."""
    from meme_challenge_amd._lib import UniterHipError
    sd = sd_from_npz(tiny)
    b = to_dev(batch_from_npz(tiny))
    m = build(TINY, TINY_IMG_DIM, sd).eval()
    m.uniter_model.pack_padded = True
    kw = model_kwargs(b)
    with pytest.raises(UniterHipError, match='seq_lens'):
        m(**kw)
    lens = m.uniter_model.lengths_from_mask(b['attn_mask'])
    assert lens == b['attn_mask'].sum(1).long().tolist()
    assert maxdiff(m(seq_lens=lens, **kw), tiny['out/logits']) < 1e-5
