"""GPU: optimizer-step semantics of the reference trainer (driven by the REAL
TrainerTemplate.calculate_loss when the golden was made): gradient-accumulation
quirk, averaging, clipping, Adam / AdamW with the name-based decay split, cosine
warm-up, skipped no-grad parameters."""
import numpy as np
import pytest
import torch

from common import TINY, TINY_IMG_DIM, sd_from_npz, batch_from_npz, maxdiff

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('precision', ['fp32', 'fp32x3'])
@pytest.mark.parametrize('optname', ['adam', 'adamw'])
def test_trainer_steps_match_reference(trainer_steps, optname, precision):
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import get_optimizer, TrainStep, cosine_warmup_lambda
    z = trainer_steps
    cfg = UniterConfig.from_dict(TINY)
    m = MemeUniter(UniterModel(cfg, img_dim=TINY_IMG_DIM), cfg.hidden_size, 1)
    m.load_state_dict(sd_from_npz(z, 'sd0/'))
    m = m.cuda().eval()          # the golden run had dropout off
    m.uniter_model.precision = precision     # fp32x3: the optimizer refreshes the three-piece weight mirror every step
    config = dict(optimizer=optname, lr=1e-3, beta1=0.9, beta2=0.999, weight_decay=1e-3,
                  gradient_accumulation=2, max_grad_norm=1, pos_wt=1.8, loss_func='bce_logits')
    opt = get_optimizer(m, config)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, cosine_warmup_lambda(1, 6))
    step = TrainStep(m, opt, sched, config)
    n_it = len(z[optname + '/losses'])
    for it in range(n_it):
        b = {k: v.cuda() for k, v in batch_from_npz(z, 'batch%d/' % it).items()}
        loss = step.train_iter(b, iters=it)
        assert abs(loss.item() - z[optname + '/losses'][it]) < 2e-5, it
        assert abs(opt.param_groups[0]['lr'] - float(z['%s/it%d/lr' % (optname, it)])) < 1e-12
        sd = m.state_dict()
        for k in z.files:
            pre = '%s/it%d/' % (optname, it)
            if k.startswith(pre) and not k.endswith('/lr'):
                assert maxdiff(sd[k[len(pre):]], z[k]) < 2e-5, (it, k)
    assert maxdiff(step.last_probs, z[optname + '/probs'][-3:]) < 2e-5
    if optname == 'adam':
        sd = m.state_dict()
        for k in z.files:
            if k.startswith('adam/final/'):
                assert maxdiff(sd[k[len('adam/final/'):]], z[k]) < 2e-5, k
        # mask_embedding never receives a gradient -> torch skips it (no decay either)
        assert torch.equal(sd['uniter_model.img_embeddings.mask_embedding.weight'].cpu(),
                           torch.from_numpy(z['sd0/uniter_model.img_embeddings.mask_embedding.weight']))


def test_grad_norm_and_clip_coefficient():
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import FusedAdam
    torch.manual_seed(1)
    cfg = UniterConfig.from_dict(TINY)
    m = MemeUniter(UniterModel(cfg, img_dim=TINY_IMG_DIM), cfg.hidden_size, 1).cuda()
    opt = FusedAdam(m, lr=1e-2, weight_decay=0.0)
    st = opt.store
    for p in m.parameters():             # (padding between tensors stays zero, as in real use)
        p.grad.copy_(torch.randn_like(p))
    g = st.flat_grads.clone()
    st.touch(st.names)
    ref = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())).item()
    assert abs(opt.grad_norm().item() - ref) < 1e-6 * ref
    p0 = st.flat_params.clone()
    opt.step(grad_scale=0.5, max_grad_norm=1.0)
    # first Adam step moves every touched parameter by lr * sign(g) (bias-corrected), whatever the clip
    moved = (st.flat_params - p0)
    for n in ('linear.weight', 'uniter_model.encoder.layer.0.output.dense.weight'):
        o, k = st.offsets[n], st.params[n].numel()
        big = g[o:o + k].abs() > 0.1          # (eps=1e-8 matters only for vanishing gradients)
        assert torch.allclose(moved[o:o + k][big], -1e-2 * torch.sign(g[o:o + k][big]), atol=5e-6)
    assert st.flat_grads.abs().max().item() == 0      # zero_grad fused into the step


def test_overlapped_optimizer_step_matches_single_launch():
    """FusedAdam.overlap_encoder: the update runs block by block on the side stream beside the next
    forward (uniter_model_set_ready_events gates each layer); parameters after several steps must equal
    the single-launch optimizer's up to the run-to-run noise of the embedding-gradient atomics, and
    join() must make them readable."""
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler
    from meme_challenge_amd.utils import make_synthetic_batch
    from common import TINY, TINY_IMG_DIM
    cfg = UniterConfig.from_dict(dict(TINY, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0))
    config = dict(optimizer='adam', lr=1e-3, beta1=0.9, beta2=0.999, weight_decay=1e-3, gradient_accumulation=1,
                  max_grad_norm=1.0, pos_wt=1.8, loss_func='bce_logits', scheduler='warmup_cosine', warmup_steps=2,
                  max_epoch=2)
    b = make_synthetic_batch(4, 16, 6, seed=3, vocab=TINY['vocab_size'], img_dim=TINY_IMG_DIM, device='cuda')
    finals = {}
    for overlap in (False, True, 'again'):
        torch.manual_seed(0)
        m = MemeUniter(UniterModel(cfg, img_dim=TINY_IMG_DIM), cfg.hidden_size, 1).cuda().train()
        m.uniter_model.use_side_stream = False        # wgrads on the main stream: no atomics-order noise between the runs
        opt = FusedAdam(m, lr=config['lr'], weight_decay=config['weight_decay'])
        if overlap is True:
            opt.overlap_encoder = m.uniter_model
        step = TrainStep(m, opt, get_scheduler(opt, config, steps_per_epoch=10), config)
        for it in range(5):
            step.train_iter(b, iters=it)
        opt.join()
        torch.cuda.synchronize()
        finals[overlap] = m.param_store().flat_params.clone()
    noise = (finals['again'] - finals[False]).abs().max().item()        # two identical single-launch runs
    diff = (finals[True] - finals[False]).abs().max().item()
    # (Adam turns an atomics-order difference of a near-zero gradient into a visible step: the bound is the larger of a
    # multiple of what two identical runs show and a few 1e-6 -- five steps at lr 1e-3 move the parameters by ~5e-3)
    assert diff <= max(4 * noise, 5e-6), (diff, noise)


@pytest.mark.parametrize('precision', ['fp32', 'fp32x3', 'bf16', 'switch', 'fp32x3_long'])
def test_clip_norm_taken_during_backward_equals_the_full_pass(precision):
    """(fp32x3: every layer's share of the norm is left by the layer's own weight-gradient launch, its riders -- no reduction
    launch per layer; 'switch': the precision changes between steps, as in bench.py's native-fp32 leg: the slots the other form
    wrote must not be counted.)
    FusedAdam.attach_norm_hooks: the clip norm reduced bucket by bucket while the backward pass runs (partial sums
    joined by uniter_sumsq_combine) is the norm of the full pass over the gradient buffer -- same double-precision sum up
    to its order -- with gradient accumulation (the partial sums of the STEPPING backward see the accumulated buffer) and
    with the weight gradients on the side stream."""
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler
    from meme_challenge_amd.utils import make_synthetic_batch
    # 'fp32x3_long' (ADVICE r05): a joint length beyond the attention kernels' fused query|key|value bias partials (L = 200 > 192) --
    # that bias gradient is then written by a launch BEHIND the riders' launch, so the plan announces no slots and the trainer
    # reduces the layer buckets itself: the norm must still be the full pass's
    long_ = precision == 'fp32x3_long'
    precision = 'fp32x3' if long_ else precision
    cfg = UniterConfig.from_dict(dict(TINY, max_position_embeddings=160) if long_ else TINY)
    config = dict(optimizer='adam', lr=1e-3, beta1=0.9, beta2=0.999, weight_decay=1e-3, gradient_accumulation=2,
                  max_grad_norm=0.05, pos_wt=1.8, loss_func='bce_logits', scheduler='warmup_cosine', warmup_steps=2,
                  max_epoch=2)
    bs = [make_synthetic_batch(2 if long_ else 4, 130 if long_ else 16, 70 if long_ else 6, seed=3 + k, vocab=TINY['vocab_size'],
                               img_dim=TINY_IMG_DIM, device='cuda') for k in range(2)]
    finals, norms = {}, {}
    for hooked in (False, True):
        torch.manual_seed(0)
        m = MemeUniter(UniterModel(cfg, img_dim=TINY_IMG_DIM), cfg.hidden_size, 1).cuda().train()
        m.uniter_model.set_dropout_seed(5, 0)
        opt = FusedAdam(m, lr=config['lr'], weight_decay=config['weight_decay'])
        step = TrainStep(m, opt, get_scheduler(opt, config, steps_per_epoch=10), config)
        assert m.uniter_model._grad_hook is not None          # TrainStep attached the hooks (clipping on, no exchange)
        if not hooked:
            m.uniter_model._grad_hook = None
        seen = []
        for it in range(5):
            m.uniter_model.precision = {'switch': ('fp32x3', 'bf16', 'fp32', 'fp32x3', 'bf16')[it]}.get(precision, precision)
            step.train_iter(bs[it % 2], iters=it)
            seen.append(opt._sumsq.clone())
        torch.cuda.synchronize()
        if hooked and precision == 'fp32x3':
            # the riders carried the layers' shares -- except where the plan cannot (long_)
            assert (m.uniter_model.norm_partials_per_layer() == 0) if long_ else (m.uniter_model.norm_partials_per_layer() > 0)
        finals[hooked], norms[hooked] = m.param_store().flat_params.clone(), torch.cat(seen)
    assert float(norms[True].min()) > 0 and float(norms[True].sqrt().min()) > config['max_grad_norm']     # the clip was active
    assert torch.allclose(norms[True], norms[False], rtol=1e-5, atol=0), (norms[True], norms[False])
    assert (finals[True] - finals[False]).abs().max().item() <= 5e-6


@pytest.mark.parametrize('precision', ['fp32', 'fp32x3', 'bf16'])
def test_lazy_zero_grad_overwrite_equals_clearing(precision):
    """FusedAdam.lazy_zero_encoder: zero_grad leaves the encoder layers' weight gradients alone (28 instead of 32 bytes per
    parameter in the optimizer step) and the next backward pass writes them with `=` (uniter_model_set_wgrad_overwrite);
    later micro-batches accumulate.  With gradient_accumulation = 2 (the reference's recipe, including its iteration-0
    quirk) the parameters after five iterations equal those of the clearing optimizer."""
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler
    from meme_challenge_amd.utils import make_synthetic_batch
    cfgd = dict(TINY, hidden_size=128, intermediate_size=256, num_attention_heads=2) if precision == 'bf16' else TINY
    cfg = UniterConfig.from_dict(cfgd)
    config = dict(optimizer='adam', lr=1e-3, beta1=0.9, beta2=0.999, weight_decay=1e-3, gradient_accumulation=2,
                  max_grad_norm=1.0, pos_wt=1.8, loss_func='bce_logits', scheduler='warmup_cosine', warmup_steps=2,
                  max_epoch=2)
    bs = [make_synthetic_batch(4, 16, 6, seed=3 + k, vocab=cfgd['vocab_size'], img_dim=TINY_IMG_DIM, device='cuda')
          for k in range(2)]
    finals = {}
    for lazy in (False, True):
        torch.manual_seed(0)
        m = MemeUniter(UniterModel(cfg, img_dim=TINY_IMG_DIM), cfg.hidden_size, 1).cuda().train()
        m.uniter_model.precision = precision
        m.uniter_model.set_dropout_seed(5, 0)
        m.uniter_model.use_side_stream = False            # (no atomics-order noise between the two runs)
        opt = FusedAdam(m, lr=config['lr'], weight_decay=config['weight_decay'])
        step = TrainStep(m, opt, get_scheduler(opt, config, steps_per_epoch=10), config)
        assert opt.lazy_zero_encoder is m.uniter_model    # TrainStep turns it on
        if not lazy:
            opt.lazy_zero_encoder = None
        st = m.param_store()
        for it in range(5):
            step.train_iter(bs[it % 2], iters=it)
            stepped = it % 2 == 0
            w = m.uniter_model.encoder.layer[0].intermediate.dense.weight.grad
            b = m.uniter_model.encoder.layer[0].intermediate.dense.bias.grad
            if stepped:
                assert b.abs().max().item() == 0.0                      # cleared as ever
                assert (w.abs().max().item() > 0.0) == lazy             # left alone: the next backward pass overwrites it
                assert st.wgrad_stale == lazy
            else:
                assert not st.wgrad_stale and w.abs().max().item() > 0.0
        torch.cuda.synchronize()
        finals[lazy] = st.flat_params.clone()
    assert (finals[True] - finals[False]).abs().max().item() <= 5e-6


@pytest.mark.parametrize('n_other', [0, 7, 15])
def test_side_stream_runs_beside_the_main_stream_whatever_else_is_in_use(n_other):
    """_lib.shared_stream: HIP hands out its hardware queues round-robin to the streams in use, and a side stream that lands on
    the main stream's queue serialises the two halves of the backward pass (16.0 instead of 13.5 ms per fp32 step with seven
    other streams in use).  The package's ONE side stream per device is chosen by measurement: whatever number of other
    streams a process has put to use, a spin kernel on it overlaps one on the current stream (child process: the choice is
    made once per process)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, 'tests/tools/stream_queue_probe.py', str(n_other)], cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'overlaps the current stream: True' in r.stdout, r.stdout


def test_two_host_threads_drive_two_model_handles():
    """The library's "next launch" side channel (in-kernel stamp slot, wave priority: csrc/common.h) is per host thread: two
    threads that each drive their own model handle on their own stream (cross-validation folds in threads, a
    DataParallel-style caller) get the gradients of a solo run, and each handle's launch stamps count ITS
    launches -- nothing is taken by the other thread's launches."""
    import ctypes as C
    import threading
    from meme_challenge_amd import _lib as L
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import bce_with_logits_loss
    from meme_challenge_amd.utils import make_synthetic_batch
    from common import model_kwargs
    lib = L.lib()
    cfg = UniterConfig.from_dict(TINY)
    NK, STEPS = 11, 6

    def make(seed, precision):
        torch.manual_seed(seed)
        m = MemeUniter(UniterModel(cfg, img_dim=TINY_IMG_DIM), cfg.hidden_size, 1).cuda().eval()
        m.uniter_model.precision = precision
        m.uniter_model.use_side_stream = False
        b = make_synthetic_batch(3 + seed, 12, 5, seed=10 + seed, vocab=TINY['vocab_size'], img_dim=TINY_IMG_DIM, device='cuda')
        return m, b

    def run(m, b, stream, out):
        try:
            with torch.cuda.stream(stream):
                m(**model_kwargs(b))                                  # builds the handle
                h = m.uniter_model._handle
                L.check(lib.uniter_prof_enable_stamps(h, 1, None))
                for _ in range(STEPS):
                    for p in m.parameters():
                        p.grad = None
                    m.param_store().zero_grads()
                    loss = bce_with_logits_loss(m(**model_kwargs(b)), b['labels'], 1.8)
                    loss.backward()
                stream.synchronize()
                n, ms = (C.c_int * NK)(), (C.c_double * NK)()
                L.check(lib.uniter_prof_collect_stamps(h, n, ms, NK))
                L.check(lib.uniter_prof_enable_stamps(h, 0, None))
                out['counts'] = list(n)
                out['grads'] = m.param_store().flat_grads.clone()
        except Exception as e:                                        # noqa: BLE001 -- surfaces in the main thread's assert
            out['error'] = repr(e)

    solo, both = [{}, {}], [{}, {}]
    specs = [(0, 'fp32'), (1, 'fp32x3')]
    for k, (seed, prec) in enumerate(specs):
        m, b = make(seed, prec)
        run(m, b, torch.cuda.Stream(), solo[k])
        assert 'error' not in solo[k], solo[k]
    models = [make(seed, prec) for seed, prec in specs]
    threads = [threading.Thread(target=run, args=(m, b, torch.cuda.Stream(), both[k])) for k, (m, b) in enumerate(models)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(2):
        assert 'error' not in both[k], both[k]
        assert both[k]['counts'] == solo[k]['counts'] and sum(both[k]['counts']) > 0, (k, both[k]['counts'], solo[k]['counts'])
        # (the embedding tables' gradients are summed with float atomics: equal up to their order)
        assert (both[k]['grads'] - solo[k]['grads']).abs().max().item() <= 1e-6 * solo[k]['grads'].abs().max().item(), k


@pytest.mark.parametrize('optimizer', ['adam', 'adamw'])
@pytest.mark.parametrize('overlap', [False, True])
@pytest.mark.parametrize('precision', ['fp32', 'fp32x3'])
def test_word_table_rows_updated_ahead_equal_the_one_launch_update(optimizer, overlap, precision):
    """Round 6 (uniter_adam_step_rows, FusedAdam.early_word_update): the rows of the word-embedding table that no token of the step
    looks up are updated AHEAD of the backward pass -- their gradient is zero whatever it computes, and 0 x clip coefficient = 0, so
    torch.optim.Adam's update of such a row (g = wd p; utils/optim_utils.py:33-40) depends on the step number and the learning rate
    alone -- and the optimizer step then takes the looked-up rows only.  Same arithmetic per element: parameters AND both moment
    buffers are bit-identical to the one-launch update, with gradient accumulation (the reference's iteration-0 quirk included),
    clipping on, Adam and AdamW, with and without the update overlapping the next forward."""
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler
    from meme_challenge_amd.utils import make_synthetic_batch
    cfg = UniterConfig.from_dict(TINY)
    config = dict(optimizer=optimizer, lr=1e-3, beta1=0.9, beta2=0.999, weight_decay=1e-2, gradient_accumulation=2,
                  max_grad_norm=0.05, pos_wt=1.8, loss_func='bce_logits', scheduler='warmup_cosine', warmup_steps=2, max_epoch=2)
    bs = [make_synthetic_batch(4, 16, 6, seed=3 + k, vocab=TINY['vocab_size'], img_dim=TINY_IMG_DIM, device='cuda') for k in range(3)]
    res = {}
    for split in (False, True):
        torch.manual_seed(0)
        m = MemeUniter(UniterModel(cfg, img_dim=TINY_IMG_DIM), cfg.hidden_size, 1).cuda().train()
        m.uniter_model.precision = precision
        m.uniter_model.set_dropout_seed(5, 0)
        opt = FusedAdam(m, lr=config['lr'], weight_decay=config['weight_decay'], adamw=(optimizer == 'adamw'))
        opt.split_word_rows = split
        opt._word_cache = None
        if overlap:
            opt.overlap_encoder = m.uniter_model
        step = TrainStep(m, opt, get_scheduler(opt, config, steps_per_epoch=10), config)
        early = 0
        for it in range(7):
            before = opt._early
            step.train_iter(bs[it % 3], iters=it)
            early += int(split and before is None and opt._early is None and it % 2 == 0)
        opt.join()
        torch.cuda.synchronize()
        assert opt._early is None and (opt._rowmask is None or int(opt._rowmask.sum()) == 0)      # nothing pending, the mask is clear
        res[split] = (m.param_store().flat_params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone())
    st = m.param_store()
    name = 'uniter_model.embeddings.word_embeddings.weight'
    V, H = st.params[name].shape
    off = st.offsets[name]
    looked_up = torch.zeros(V, dtype=torch.bool, device='cuda')
    for b in bs:
        looked_up[b['input_ids'].reshape(-1)] = True
    never = ~looked_up
    assert int(never.sum()) >= 8              # rows of the table that none of these batches looks up
    for a, b in zip(res[False], res[True]):
        # the rows updated ahead: their update depends on the step number and the learning rate alone -- bit for bit the one-launch update's
        ta, tb = a[off:off + V * H].view(V, H), b[off:off + V * H].view(V, H)
        assert torch.equal(ta[never], tb[never])
        # everything else: the same arithmetic on gradients that carry the embedding backward's float-atomics order noise
        assert (a - b).abs().max().item() <= 5e-6
    # ... and they DID move (weight decay + the moments' decay): not a comparison of two untouched tables
    assert (res[True][0][off:off + V * H].view(V, H)[never] - st.params[name].detach().new_zeros(1)).abs().max().item() > 0
