"""CPU: host-side mirror of the reference interface (no GPU compute): state_dict
keys, config, from_pretrained renames, input helpers, schedules, C-ABI surface,
and the fail-loudly contract."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from common import TINY, TINY_IMG_DIM, BASE

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_model():
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    cfg = UniterConfig.from_dict(TINY)
    return MemeUniter(UniterModel(cfg, img_dim=TINY_IMG_DIM), cfg.hidden_size, 1)


def test_library_builds_loads_and_exports_every_declared_symbol():
    from meme_challenge_amd import build, _lib
    build.build(verbose=False)
    h = ctypes.CDLL(_lib.LIB_PATH)
    hdr = open(os.path.join(REPO, 'include', 'uniter_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(uniter_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(h, name), 'library does not export %s' % name
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    lib = _lib.lib()
    assert lib.uniter_abi_version() == 1
    assert b'gfx950' in lib.uniter_build_info()


def test_param_table_of_the_library_matches_reference_state_dict(host_helpers):
    from meme_challenge_amd import _lib
    lib = _lib.lib()
    cc = _lib.UniterConfigC(64, 12, 1, 128, 50, 16, 2, 32, 0.1, 0.1)
    n = lib.uniter_num_params(ctypes.byref(cc))
    names = ['uniter_model.' + lib.uniter_param_name(ctypes.byref(cc), i).decode() for i in range(n)]
    ref = [k for k in host_helpers['state_dict_keys'].tolist() if k.startswith('uniter_model.')]
    assert names == ref
    r, c = ctypes.c_int64(), ctypes.c_int64()
    assert lib.uniter_param_shape(ctypes.byref(cc), 5, ctypes.byref(r), ctypes.byref(c)) == 0
    assert (r.value, c.value) == (64, 32)          # img_linear.weight [H, img_dim]
    # invalid arguments are reported, never crash
    assert lib.uniter_gemm_f32(0, 0, 0, 4, 4, None, 4, None, 4, None, 4, 0, None, None, None, 0, 0, None) == -1
    assert b'gemm' in lib.uniter_last_error()


def test_state_dict_keys_and_shapes_match_reference(host_helpers, tiny):
    m = _tiny_model()
    tiny_like = dict(BASE, vocab_size=50, hidden_size=64, intermediate_size=128,
                     num_attention_heads=1, max_position_embeddings=16)
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    m2 = MemeUniter(UniterModel(UniterConfig.from_dict(tiny_like), 32), 64, 1)
    assert list(m2.state_dict().keys()) == host_helpers['state_dict_keys'].tolist()
    sd = m.state_dict()
    for k in tiny.files:
        if k.startswith('sd/'):
            assert tuple(sd[k[3:]].shape) == tiny[k].shape, k
    # reference checkpoints load unchanged (utils/save.py:57-63 format)
    ref_sd = {k[3:]: torch.from_numpy(tiny[k]) for k in tiny.files if k.startswith('sd/')}
    m.load_state_dict(ref_sd, strict=True)
    assert torch.equal(m.linear.weight.data, ref_sd['linear.weight'])


def test_init_weights_statistics():
    torch.manual_seed(0)
    m = _tiny_model()
    um = m.uniter_model
    assert abs(um.embeddings.word_embeddings.weight.std().item() - 0.02) < 0.002
    assert um.encoder.layer[0].attention.self.query.bias.abs().max().item() == 0
    assert torch.all(um.embeddings.LayerNorm.weight == 1) and torch.all(um.img_embeddings.img_layer_norm.bias == 0)
    # layers are initialised independently (model/model.py:278-280,304)
    assert not torch.equal(um.encoder.layer[0].output.dense.weight, um.encoder.layer[1].output.dense.weight)


def test_config_roundtrip_and_errors(tmp_path):
    from meme_challenge_amd.model import UniterConfig, UniterModel
    p = tmp_path / 'c.json'
    p.write_text(json.dumps(BASE))
    c = UniterConfig.from_json_file(str(p))
    assert c.hidden_size == 768 and c.to_dict() == BASE
    assert json.loads(c.to_json_string()) == BASE
    assert UniterConfig(str(p)).num_hidden_layers == 12
    assert UniterConfig(100, hidden_size=64).vocab_size == 100
    with pytest.raises(ValueError):
        UniterConfig(3.5)
    with pytest.raises(ValueError):
        UniterModel({'hidden_size': 64}, 32)
    bad = UniterConfig.from_dict(dict(TINY, hidden_size=100, num_attention_heads=3))
    with pytest.raises(ValueError):
        UniterModel(bad, 32)


def test_from_pretrained_renames_and_prefix(tmp_path, tiny):
    from meme_challenge_amd.model import UniterModel
    p = tmp_path / 'c.json'
    p.write_text(json.dumps(TINY))
    sd = {}
    for k in tiny.files:
        if k.startswith('sd/uniter_model.'):
            name = 'bert.' + k[len('sd/uniter_model.'):]
            if 'LayerNorm' in name or 'layer_norm' in name:
                name = name.replace('.weight', '.gamma').replace('.bias', '.beta')
            sd[name] = torch.from_numpy(tiny[k])
    m = UniterModel.from_pretrained(str(p), sd, img_dim=TINY_IMG_DIM)
    assert torch.equal(m.embeddings.LayerNorm.weight.data,
                       torch.from_numpy(tiny['sd/uniter_model.embeddings.LayerNorm.weight']))
    assert torch.equal(m.encoder.layer[1].output.dense.weight.data,
                       torch.from_numpy(tiny['sd/uniter_model.encoder.layer.1.output.dense.weight']))
    sd_bad = dict(sd)
    sd_bad['bert.pooler.dense.weight'] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError):
        UniterModel.from_pretrained(str(p), sd_bad, img_dim=TINY_IMG_DIM)


def test_input_helpers_match_reference(host_helpers):
    from meme_challenge_amd.utils import get_gather_index, get_attention_mask, make_synthetic_batch
    from oracle import uniter_oracle as O
    z = host_helpers
    for k in ('a', 'b', 'c'):
        tl, nbb, T = z['gi/%s/tl' % k].tolist(), z['gi/%s/nbb' % k].tolist(), int(z['gi/%s/T' % k])
        am = get_attention_mask(tl, nbb)
        gi = get_gather_index(tl, nbb, len(tl), T, am.shape[1])
        assert np.array_equal(am.numpy(), z['gi/%s/attn_mask' % k])
        assert np.array_equal(gi.numpy(), z['gi/%s/gather_index' % k])
    a = make_synthetic_batch(3, 12, 5, seed=9, txt_lens=[12, 3, 7], num_bbs=[5, 5, 2])
    b = O.synth_batch(3, 12, 5, seed=9, txt_lens=[12, 3, 7], num_bbs=[5, 5, 2])
    assert a.pop('seq_lens') == [17, 8, 9]          # host-side lengths (extension key for token packing)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_lr_lambdas_match_transformers(host_helpers):
    from meme_challenge_amd.trainer import cosine_warmup_lambda, linear_warmup_lambda, no_decay
    z = host_helpers
    for nm, fn in (('cos_500_3000', cosine_warmup_lambda(500, 3000)), ('cos_2_10', cosine_warmup_lambda(2, 10)),
                   ('lin_50_400', linear_warmup_lambda(50, 400))):
        ref = z['lr/' + nm]
        assert np.abs(np.array([fn(i) for i in range(len(ref))]) - ref).max() < 1e-12
    for n in z['decay_names'].tolist():
        assert not no_decay(n)
    for n in z['no_decay_names'].tolist():
        assert no_decay(n)


def test_product_path_fails_loudly_without_gpu_or_library():
    """No CPU fallback: CPU tensors are rejected, and a missing .so raises."""
    from meme_challenge_amd._lib import UniterHipError
    m = _tiny_model()
    from oracle import uniter_oracle as O
    b = O.synth_batch(2, 6, 3, vocab=TINY['vocab_size'], img_dim=TINY_IMG_DIM)
    with pytest.raises(UniterHipError):
        m(img_feat=b['img_feat'], img_pos_feat=b['img_pos_feat'], input_ids=b['input_ids'],
          position_ids=b['position_ids'], attention_mask=b['attn_mask'], gather_index=b['gather_index'],
          output_all_encoded_layers=False)
    code = ("import meme_challenge_amd._lib as L\n"
            "L.LIB_PATH = '/nonexistent/libuniter_hip.so'\n"
            "try:\n    L.lib()\nexcept L.UniterHipError as e:\n    print('RAISED', e)\n")
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=REPO)
    assert 'RAISED' in out.stdout and 'no fallback' in out.stdout


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(REPO, 'meme_challenge_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert '/root/reference' not in src


@pytest.mark.parametrize('task', ['mlm', 'mrfr', 'itm'])
def test_synthetic_pretrain_batch_layout(task):
    """Key names / shapes of the reference's pretraining collates (pretrain_mlm.py:72-130,
    pretrain_mrfr.py:54-130, pretrain_itm.py:50-117): `attn_masks` (sic), [1,T] position ids,
    -1 labels off the masked positions, feat_targets = masked regions' features."""
    from meme_challenge_amd.utils import make_synthetic_pretrain_batch, make_synthetic_batch
    B, T, R = 3, 12, 5
    tl, nb = [12, 7, 4], [5, 3, 2]
    b = make_synthetic_pretrain_batch(task, B, T, R, seed=5, vocab=200, img_dim=8, txt_lens=tl, num_bbs=nb)
    base = make_synthetic_batch(B, T, R, seed=5, vocab=200, img_dim=8, txt_lens=tl, num_bbs=nb)
    assert b['position_ids'].shape == (1, T) and 'attn_masks' in b and 'attn_mask' not in b
    assert torch.equal(b['attn_masks'], base['attn_mask']) and torch.equal(b['gather_index'], base['gather_index'])
    if task == 'mlm':
        lab = b['txt_labels']
        assert lab.shape == (B, T) and (lab[:, 0] == -1).all()
        for i in range(B):
            assert (lab[i, tl[i]:] == -1).all() and (lab[i] != -1).sum() >= 1
        sel = lab != -1
        assert torch.equal(lab[sel], base['input_ids'][sel])               # label = original token
        assert torch.equal(b['input_ids'][~sel], base['input_ids'][~sel])   # untouched elsewhere
    elif task == 'mrfr':
        m, tgt = b['img_masks'], b['img_mask_tgt']
        assert m.dtype == torch.bool and m.shape == (B, R) and tgt.shape == b['attn_masks'].shape
        for i in range(B):
            assert m[i].sum() >= 1 and not m[i, nb[i]:].any()
            assert torch.equal(tgt[i, tl[i]:tl[i] + nb[i]], m[i, :nb[i]]) and tgt[i].sum() == m[i].sum()
        assert torch.equal(b['feat_targets'], base['img_feat'][m])
    else:
        assert b['targets'].shape == (B,) and set(b['targets'].tolist()) <= {0, 1}


def test_package_import_sets_the_hardware_queue_default():
    """Main / side stream overlap needs more HIP hardware queues than the default once RCCL has created its own
    streams (meme_challenge_amd/__init__.py); a value chosen by the user is left alone."""
    import os, subprocess, sys
    code = "import os; import meme_challenge_amd; print(os.environ['GPU_MAX_HW_QUEUES'])"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != 'GPU_MAX_HW_QUEUES'}
    assert subprocess.check_output([sys.executable, '-c', code], cwd=root, env=env).decode().strip() == '8'
    env['GPU_MAX_HW_QUEUES'] = '4'
    assert subprocess.check_output([sys.executable, '-c', code], cwd=root, env=env).decode().strip() == '4'


def test_loss_func_bce_and_ce_follow_the_reference():
    """train_template.py:64-69,96-99: 'bce' = nn.BCELoss on torch.sigmoid(preds) (the head emits logits), 'ce' =
    nn.CrossEntropyLoss on two logits.  (binary_cross_entropy on raw logits asserts on the device.)"""
    import torch
    from types import SimpleNamespace
    from meme_challenge_amd.train_template import TrainerTemplate
    torch.manual_seed(0)
    logits = torch.randn(6, 1) * 3.0
    labels = torch.tensor([0, 1, 1, 0, 1, 0])
    stub = SimpleNamespace(config={'loss_func': 'bce', 'pos_wt': 1.0})
    loss, probs = TrainerTemplate._loss_and_probs(stub, logits, labels)
    ref = torch.nn.BCELoss()(torch.sigmoid(logits).squeeze(1), labels.float())
    assert torch.allclose(loss, ref) and torch.allclose(probs, torch.sigmoid(logits.squeeze(1)))
    assert float(probs.min()) >= 0.0 and float(probs.max()) <= 1.0
    two = torch.randn(6, 2)
    stub.config['loss_func'] = 'ce'
    loss, probs = TrainerTemplate._loss_and_probs(stub, two, labels)
    assert torch.allclose(loss, torch.nn.CrossEntropyLoss()(two, labels))
    assert torch.allclose(probs, torch.softmax(two, dim=1)[:, 1])


def test_adamax_and_sgd_take_the_reference_arguments():
    """utils/optim_utils.py:36-43: Adamax with torch's default betas (the config's are not passed), SGD with momentum = beta1."""
    import inspect
    from meme_challenge_amd import trainer
    src = inspect.getsource(trainer.get_optimizer)
    assert "torch.optim.Adamax(groups, lr=config['lr'])" in src
    assert "torch.optim.SGD(groups, lr=config['lr'], momentum=config['beta1'])" in src


def test_x3_plan_fits_the_cus_it_is_given():
    """uniter_gemm_x3_plan (host arithmetic, no launch): geometry and k-pieces of the fp32x3 forward / input-gradient products.  On the
    whole chip the choices measured in profiles/r05_gemm_x3_lab.txt (two k-pieces of 128 x 128 tiles for the N = hidden products,
    128 x 256 tiles for the wide ones); on the 240 CUs a data-parallel exchange leaves, no form of 252 work items (two rounds there)."""
    import ctypes as C
    from meme_challenge_amd import _lib as L
    lib = L.lib()

    def plan(M, N, K, avail, fixed=0):
        c, n = C.c_int(), C.c_int()
        assert lib.uniter_gemm_x3_plan(M, N, K, avail, fixed, C.byref(c), C.byref(n)) == 0
        return c.value, n.value
    M, H, I = 2624, 768, 3072
    assert plan(M, H, I, 256) == (3, 2) and plan(M, H, H, 256) == (3, 2) and plan(M, H, 3 * H, 256) == (3, 2)
    assert plan(M, I, H, 256, 1) == (4, 1) and plan(M, 3 * H, H, 256, 1) == (4, 1)
    for shape in ((M, H, I), (M, H, H), (M, H, 3 * H)):
        cfg, ns = plan(*shape, 240)
        tiles = ((shape[0] + 127) // 128) * ((shape[1] + (255 if cfg == 4 else 127)) // (256 if cfg == 4 else 128))
        assert tiles * ns <= 240, (shape, cfg, ns)                      # one round on what is left
    assert plan(M, I, H, 240, 1)[0] == 3                                 # 252 tiles of 128 x 256 would take two rounds: 504 persistent ones
    assert plan(M, 3 * H, H, 240, 1) == (4, 1)                           # 189 tiles: still one round
    assert plan(64, 8, 64, 256) == (3, 1)                                # tiny: nothing to choose
    assert lib.uniter_gemm_x3_plan(0, 8, 64, 256, 0, None, None) != 0


def test_balanced_walk_slot_counts_are_host_arithmetic():
    """uniter_wgrad_x3_group_slots_ws / uniter_gemm_x3_balanced_ws_bytes (no launch): a layer of UNITER-base is 216 tiles of 128 x 256 --
    the classic walk writes 8 x 216 sum-of-squares slots, the balanced walk (a workspace of the advertised size) one per compute wave
    of every CU it may use; a grid capped to the tile count, a workspace too small, or no reduction length: the classic count."""
    import ctypes as C
    from meme_challenge_amd import _lib as L
    lib = L.lib()
    nb = lib.uniter_gemm_x3_balanced_ws_bytes()
    assert nb == 16384 + 256 * 128 * 256 * 4                              # flag words + one partial-sum tile per CU (no device: 256)
    IA = C.c_int * 4
    Ms, Ns = IA(3072, 768, 2304, 768), IA(768, 3072, 768, 768)
    plain = lib.uniter_wgrad_x3_group_slots(4, 4, Ms, Ns, 0)
    assert plain == 8 * 216
    assert lib.uniter_wgrad_x3_group_slots_ws(4, 4, Ms, Ns, 2624, 0, nb) == 8 * 256
    assert lib.uniter_wgrad_x3_group_slots_ws(4, 4, Ms, Ns, 2624, 240, nb) == 8 * 240
    assert lib.uniter_wgrad_x3_group_slots_ws(4, 4, Ms, Ns, 2624, 216, nb) == plain          # full rounds: nothing to balance
    assert lib.uniter_wgrad_x3_group_slots_ws(4, 4, Ms, Ns, 2624, 0, nb - 1) == plain         # workspace too small
    assert lib.uniter_wgrad_x3_group_slots_ws(4, 4, Ms, Ns, 0, 0, nb) == plain
    assert lib.uniter_wgrad_x3_group_slots_ws(3, 4, Ms, Ns, 2624, 0, nb) == lib.uniter_wgrad_x3_group_slots(3, 4, Ms, Ns, 0)   # 128 x 128 tiles: classic
