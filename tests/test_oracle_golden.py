"""CPU: pin the oracle (oracle/*.py) against vectors produced by running the
reference itself (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import uniter_oracle as O
from oracle import step_oracle as S
from oracle import philox
from common import (TINY, TINY_IMG_DIM, BASE, LARGE, sd_from_npz, batch_from_npz,
                    model_kwargs, maxdiff)


def test_tiny_forward_intermediates(tiny):
    sd = sd_from_npz(tiny)
    b = batch_from_npz(tiny)
    p = 'uniter_model.'
    txt = O.text_embeddings(sd, p, b['input_ids'], b['position_ids'], None, TINY, None)
    img = O.image_embeddings(sd, p, b['img_feat'], b['img_pos_feat'], None, None, TINY, None)
    assert maxdiff(txt, tiny['out/txt_emb']) < 2e-6
    assert maxdiff(img, tiny['out/img_emb']) < 2e-6
    kw = model_kwargs(b)
    kw['output_all_encoded_layers'] = True
    layers, emb = O.uniter_forward(sd, TINY, prefix=p, return_embed=True, **kw)
    assert maxdiff(emb, tiny['out/emb']) < 2e-6
    for i, l in enumerate(layers):
        assert maxdiff(l, tiny['out/layer%d' % i]) < 5e-6
    assert maxdiff(O.pooler(sd, p, layers[-1]), tiny['out/pooled']) < 2e-6
    logits = O.meme_uniter_forward(sd, TINY, **model_kwargs(b))
    assert maxdiff(logits, tiny['out/logits']) < 2e-6


def test_tiny_modes(tiny):
    sd = sd_from_npz(tiny)
    b = batch_from_npz(tiny)
    B, T = b['input_ids'].shape
    R = b['img_feat'].shape[1]
    p = 'uniter_model.'
    t = O.uniter_forward(sd, TINY, b['input_ids'], b['position_ids'], None, None,
                         torch.ones(B, T), output_all_encoded_layers=False, prefix=p)
    assert maxdiff(t, tiny['out/txt_only']) < 5e-6
    i = O.uniter_forward(sd, TINY, None, None, b['img_feat'], b['img_pos_feat'],
                         torch.ones(B, R), output_all_encoded_layers=False, prefix=p)
    assert maxdiff(i, tiny['out/img_only']) < 5e-6
    kw = model_kwargs(b)
    m = O.uniter_forward(sd, TINY, prefix=p, img_masks=b['img_masks'], **kw)
    assert maxdiff(m, tiny['out/masked']) < 5e-6


def test_tiny_loss_and_grads(tiny):
    sd = {k: v.clone().requires_grad_(True) for k, v in sd_from_npz(tiny).items()}
    b = batch_from_npz(tiny)
    logits = O.meme_uniter_forward(sd, TINY, **model_kwargs(b))
    loss = S.bce_with_logits(logits, b['labels'], 1.8)
    assert abs(loss.item() - float(tiny['out/loss'])) < 1e-6
    loss.backward()
    for k, v in sd.items():
        ref = torch.from_numpy(tiny['grad/' + k])
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        tol = 1e-6 + 1e-4 * ref.abs().max().item()
        assert maxdiff(g, ref) <= tol, k
    # padding_idx=0 rows never receive gradient (model/model.py:220-221)
    assert sd['uniter_model.embeddings.word_embeddings.weight'].grad is not None


def test_host_helpers(host_helpers):
    z = host_helpers
    for k in ('a', 'b', 'c'):
        tl, nbb, T = z['gi/%s/tl' % k].tolist(), z['gi/%s/nbb' % k].tolist(), int(z['gi/%s/T' % k])
        am = O.get_attention_mask(tl, nbb)
        gi = O.get_gather_index(tl, nbb, len(tl), T, am.shape[1])
        assert np.array_equal(am.numpy(), z['gi/%s/attn_mask' % k])
        assert np.array_equal(gi.numpy(), z['gi/%s/gather_index' % k])
    # SURVEY 8(c) example row
    gi = O.get_gather_index([64, 40, 10, 55], [36, 20, 36, 12], 4, 64, 100)
    assert gi[2].tolist() == list(range(10)) + list(range(64, 100)) + list(range(46, 100))
    tiny_like = dict(BASE, vocab_size=50, hidden_size=64, intermediate_size=128,
                     num_attention_heads=1, max_position_embeddings=16)
    names = [n for n, _ in O.param_shapes(tiny_like, img_dim=32)]
    assert names == list(z['state_dict_keys'])
    assert len(names) == 212
    assert sum(int(np.prod(s)) for _, s in O.param_shapes(BASE)) == 109899521 == int(z['state_dict_numel_base'])
    dec = set(z['decay_names'].tolist())
    nod = set(z['no_decay_names'].tolist())
    for n in names:
        assert (n in nod) == S.no_decay(n)
        assert (n in dec) != (n in nod)
    assert 'uniter_model.img_embeddings.img_layer_norm.weight' in dec   # the quirk


def test_lr_schedules(host_helpers):
    z = host_helpers
    for nm, fn, (w, t) in (('cos_500_3000', S.cosine_warmup_lambda, (500, 3000)),
                           ('cos_2_10', S.cosine_warmup_lambda, (2, 10)),
                           ('lin_50_400', S.linear_warmup_lambda, (50, 400))):
        ref = z['lr/' + nm]
        mine = np.array([fn(i, w, t) for i in range(len(ref))])
        assert np.abs(mine - ref).max() < 1e-12


@pytest.mark.parametrize('optname', ['adam', 'adamw'])
def test_trainer_steps(trainer_steps, optname):
    z = trainer_steps
    sd = {k: v.clone().requires_grad_(True) for k, v in sd_from_npz(z, 'sd0/').items()}
    n_it = len(z[optname + '/losses'])
    batches = [batch_from_npz(z, 'batch%d/' % i) for i in range(n_it)]
    opt = S.AdamOracle(list(sd.items()), lr=1e-3, betas=(0.9, 0.999),
                       weight_decay=1e-3, adamw=(optname == 'adamw'))
    lam = lambda s: S.cosine_warmup_lambda(s, 1, 6)
    losses = []
    # run one iteration at a time so that parameters can be compared after each
    for it in range(n_it):
        pass
    state = {'acc': None}
    losses = S.train_iterations(
        lambda b: O.meme_uniter_forward(sd, TINY, **model_kwargs(b)), sd, opt,
        batches, [b['labels'] for b in batches], pos_wt=1.8,
        gradient_accumulation=2, max_grad_norm=1, lr_lambda=lam)
    assert np.abs(np.array(losses) - z[optname + '/losses']).max() < 2e-5
    if optname == 'adam':
        for k, v in sd.items():
            assert maxdiff(v, z['adam/final/' + k]) < 2e-5, k
    else:
        for k in ('linear.weight', 'uniter_model.encoder.layer.1.output.dense.weight',
                  'uniter_model.embeddings.word_embeddings.weight'):
            assert maxdiff(sd[k], z['adamw/it%d/%s' % (n_it - 1, k)]) < 2e-5, k


def test_base_shapes(shapes_base):
    z = shapes_base
    sd = O.synth_state_dict(BASE, seed=0, ln_jitter=0.02)
    for name in ('cfg1_full', 'cfg1_ragged'):
        B, T, R, seed = z[name + '/shape'].tolist()
        tl = z[name + '/txt_lens'].tolist() if name + '/txt_lens' in z.files else None
        nbb = z[name + '/num_bbs'].tolist() if name + '/num_bbs' in z.files else None
        b = O.synth_batch(B, T, R, seed=seed, txt_lens=tl, num_bbs=nbb)
        with torch.no_grad():
            logits = O.meme_uniter_forward(sd, BASE, **model_kwargs(b))
        assert maxdiff(logits, z[name + '/logits']) < 2e-5, name


def test_philox_known_answer():
    # Random123 known-answer vectors for Philox4x32-10
    r = philox.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = philox.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff,
                             0xffffffff, 0xffffffff)
    assert [int(x) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = philox.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344,
                             0xa4093822, 0x299f31d0)
    assert [int(x) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    m = philox.keep_mask(100000, 0.1, seed=42, offset=3, site=5)
    assert abs(m.mean() - 0.9) < 0.005
