"""Training-step semantics of the reference trainer on the HIP path.

Mirrors:
  nn.BCEWithLogitsLoss(pos_weight)            train_template.py:64-65
  TrainerTemplate.calculate_loss              train_template.py:95-126
  TrainerTemplate.average_gradients           train_template.py:89-92
  get_optimizer (param-group split, Adam/AdamW) utils/optim_utils.py:9-46
  init_scheduler (warmup / warmup_cosine / step / multi_step)  train_template.py:72-82
  TrainerUniter.{train,eval,test}_iter_step    train_uniter.py:58-81

The optimizer step (gradient averaging, global-norm clip, Adam/AdamW update,
zero_grad) is ONE fused kernel over the flat parameter buffer
(csrc/optim.hip); the gradient norm stays on the device, so a training
iteration needs no host synchronisation (the reference does `.item()` /
`.cpu()` every iteration, train_template.py:121-124).
"""
import math

import torch

from . import _lib
from ._lib import check, ptr, UniterHipError
from .model import CHUNK, ensure_store


# --------------------------------------------------------------------------- #
# loss
# --------------------------------------------------------------------------- #
class _BceLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, pos_weight):
        _lib.require_gpu_tensor(logits, torch.float32, 'logits')
        x = logits.reshape(-1).contiguous()
        y = labels.reshape(-1).to(torch.int64).contiguous()
        if x.numel() != y.numel():
            raise ValueError('logits/labels size mismatch')
        B = x.numel()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        probs = torch.empty(B, dtype=torch.float32, device=x.device)
        dlog = torch.empty(B, dtype=torch.float32, device=x.device)
        check(_lib.lib().uniter_bce_logits(ptr(x), ptr(y), float(pos_weight), ptr(loss), ptr(probs),
                                           ptr(dlog), 1.0, B, _lib.cur_stream()), 'uniter_bce_logits')
        ctx.save_for_backward(dlog)
        ctx.shape = logits.shape
        ctx.mark_non_differentiable(probs)
        return loss, probs

    @staticmethod
    def backward(ctx, dloss, _dprobs):
        (dlog,) = ctx.saved_tensors
        return (dlog * dloss).view(ctx.shape), None, None


def bce_with_logits_loss(logits, labels, pos_weight=1.0, return_probs=False):
    """BCEWithLogitsLoss(pos_weight=[w]) with mean reduction; labels are cast to
    float as train_template.py:99 does.  Also yields sigmoid(logits)."""
    loss, probs = _BceLogitsFn.apply(logits, labels, pos_weight)
    return (loss, probs) if return_probs else loss


# --------------------------------------------------------------------------- #
# optimizer
# --------------------------------------------------------------------------- #
NO_DECAY = ('bias', 'LayerNorm.bias', 'LayerNorm.weight')      # utils/optim_utils.py:16


def no_decay(name):
    return any(nd in name for nd in NO_DECAY)


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam (coupled L2) / AdamW over the model's flat buffers.

    ``param_groups`` keeps the reference's two-group layout (decay / no-decay) so
    LR schedulers written against torch optimizers keep working; the arithmetic is
    one kernel launch.  ``step()`` also performs ``average_gradients`` (``grad_scale``)
    and ``clip_grad_norm_`` (``max_grad_norm``) when asked to, and zeroes the
    gradients (the reference calls ``zero_grad`` right after ``step``).
    """

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3, adamw=False):
        self.model = model
        self.store = model.param_store() if hasattr(model, 'param_store') else ensure_store(model)
        named = list(model.named_parameters())
        decay = [p for n, p in named if not no_decay(n)]
        nodecay = [p for n, p in named if no_decay(n)]
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__([{'params': decay, 'weight_decay': weight_decay},
                          {'params': nodecay, 'weight_decay': 0.0}], defaults)
        self.adamw = bool(adamw)
        self.step_count = 0
        st = self.store
        self.exp_avg = torch.zeros_like(st.flat_params)
        self.exp_avg_sq = torch.zeros_like(st.flat_params)
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=st.device)
        n_ws = _lib.lib().uniter_grad_sumsq_ws_bytes(st.numel)
        self._ws = torch.empty(n_ws, dtype=torch.uint8, device=st.device)
        self._ws_bytes = n_ws
        self._flags_key = None
        self._flags = None
        self._flags_cache = {}      # touched-set -> device flags (multitask training alternates between a few sets)

    def _chunk_flags(self):
        st = self.store
        key = frozenset(st.touched)
        if key != self._flags_key:
            flags = self._flags_cache.get(key)
            if flags is None:
                host = torch.zeros(st.numel // CHUNK, dtype=torch.uint8)
                for n in st.touched:
                    o = st.offsets[n] // CHUNK
                    k = (st.params[n].numel() + CHUNK - 1) // CHUNK
                    host[o:o + k] = 1 if no_decay(n) else 2
                flags = host.to(st.device)
                if len(self._flags_cache) < 16:
                    self._flags_cache[key] = flags
            self._flags = flags
            self._flags_key = key
        return self._flags

    def grad_norm(self):
        """L2 norm of all gradients touched since the last zero_grad (device tensor)."""
        st = self.store
        check(_lib.lib().uniter_grad_sumsq(ptr(st.flat_grads), ptr(self._chunk_flags()), st.numel,
                                           ptr(self._sumsq), ptr(self._ws), self._ws_bytes,
                                           _lib.cur_stream()), 'uniter_grad_sumsq')
        return self._sumsq.sqrt()

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0, max_grad_norm=0.0, zero_grads=True):
        if closure is not None:
            raise UniterHipError('FusedAdam does not support closures')
        st = self.store
        if not st.is_current():
            raise UniterHipError('model parameters were moved after the optimizer was built')
        g0, g1 = self.param_groups
        lr = float(g0['lr'])
        if float(g1['lr']) != lr:
            raise UniterHipError('FusedAdam needs one learning rate for both parameter groups')
        flags = self._chunk_flags()
        lib = _lib.lib()
        if max_grad_norm and max_grad_norm > 0:
            check(lib.uniter_grad_sumsq(ptr(st.flat_grads), ptr(flags), st.numel, ptr(self._sumsq),
                                        ptr(self._ws), self._ws_bytes, _lib.cur_stream()),
                  'uniter_grad_sumsq')
        self.step_count += 1
        b1, b2 = g0['betas']
        check(lib.uniter_adam_step(ptr(st.flat_params), ptr(st.flat_grads), ptr(self.exp_avg),
                                   ptr(self.exp_avg_sq), ptr(flags), st.numel, ptr(self._sumsq),
                                   float(grad_scale), float(max_grad_norm or 0.0), lr, float(b1), float(b2),
                                   float(g0['eps']), float(g0['weight_decay']), self.step_count,
                                   int(self.adamw), int(bool(zero_grads)), _lib.cur_stream()),
              'uniter_adam_step')
        if zero_grads:
            st.touched.clear()      # flags stay cached: the same set is touched again next step

    def zero_grad(self, set_to_none=False):
        self.store.zero_grads()
        self._flags_key = None


def get_optimizer(model, config, group_param_func=None):
    """utils/optim_utils.py:9-46 for the optimizers the HIP step implements."""
    if group_param_func is not None:
        raise UniterHipError('custom parameter grouping is not supported by the fused optimizer')
    name = config['optimizer']
    if name not in ('adam', 'adamw'):
        raise ValueError('invalid optimizer' if name not in ('adamax', 'sgd') else
                         'optimizer %r is not built on the HIP path (adam / adamw are)' % name)
    return FusedAdam(model, lr=config['lr'], betas=(config['beta1'], config['beta2']),
                     weight_decay=config['weight_decay'], adamw=(name == 'adamw'))


# --------------------------------------------------------------------------- #
# schedulers (transformers.get_*_schedule_with_warmup restated; train_template.py:72-82)
# --------------------------------------------------------------------------- #
def cosine_warmup_lambda(num_warmup_steps, num_training_steps, num_cycles=0.5):
    def f(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        prog = float(step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * prog)))
    return f


def linear_warmup_lambda(num_warmup_steps, num_training_steps):
    def f(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) /
                   float(max(1, num_training_steps - num_warmup_steps)))
    return f


def get_scheduler(optimizer, config, steps_per_epoch):
    total = steps_per_epoch * config['max_epoch']
    kind = config['scheduler']
    if kind == 'step':
        return torch.optim.lr_scheduler.StepLR(optimizer, step_size=config['lr_decay_step'],
                                               gamma=config['lr_decay_factor'])
    if kind == 'multi_step':
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[5, 10, 15, 25, 40],
                                                    gamma=config['lr_decay_factor'])
    if kind == 'warmup':
        return torch.optim.lr_scheduler.LambdaLR(optimizer, linear_warmup_lambda(config['warmup_steps'], total))
    if kind == 'warmup_cosine':
        return torch.optim.lr_scheduler.LambdaLR(optimizer, cosine_warmup_lambda(config['warmup_steps'], total))
    raise ValueError('unknown scheduler %r' % kind)


# --------------------------------------------------------------------------- #
# the step
# --------------------------------------------------------------------------- #
class TrainStep(object):
    """`TrainerTemplate.calculate_loss` (train_template.py:95-126) + the forward of
    `TrainerUniter.train_iter_step` (train_uniter.py:67-72).

    Quirk kept on purpose: the modulo test is on the per-epoch iteration index
    starting at 0, so iteration 0 steps immediately with a single micro-batch whose
    gradient is still divided by ``gradient_accumulation``.
    """

    def __init__(self, model, optimizer, scheduler, config, grad_sync=None):
        self.model, self.optimizer, self.scheduler, self.config = model, optimizer, scheduler, config
        self.grad_sync = grad_sync          # DP: object with .finish() called before the optimizer step
        if config.get('loss_func', 'bce_logits') != 'bce_logits':
            raise UniterHipError("only loss_func='bce_logits' is built on the HIP path")
        self.iters = 0
        self.last_loss = None
        self.last_probs = None

    @staticmethod
    def forward_kwargs(batch):
        return dict(img_feat=batch['img_feat'], img_pos_feat=batch['img_pos_feat'],
                    input_ids=batch['input_ids'], position_ids=batch['position_ids'],
                    attention_mask=batch['attn_mask'], gather_index=batch['gather_index'],
                    output_all_encoded_layers=False, seq_lens=batch.get('seq_lens'))

    def train_iter(self, batch, iters=None):
        if iters is not None:
            self.iters = iters
        cfg = self.config
        preds = self.model(**self.forward_kwargs(batch))
        loss, probs = bce_with_logits_loss(preds.squeeze(1), batch['labels'], cfg['pos_wt'],
                                           return_probs=True)
        accum = cfg['gradient_accumulation']
        stepping = self.iters % accum == 0
        if self.grad_sync is not None:
            self.grad_sync.prepare(will_step=stepping)
        loss.backward()
        if stepping:
            if self.grad_sync is not None:
                self.grad_sync.finish()
            world = self.grad_sync.world if self.grad_sync is not None else 1
            self.optimizer.step(grad_scale=1.0 / (accum * world), max_grad_norm=cfg['max_grad_norm'],
                                zero_grads=True)
            self.scheduler.step()
        self.last_loss, self.last_probs = loss.detach(), probs
        self.iters += 1
        return self.last_loss

    @torch.no_grad()
    def eval_iter(self, batch):
        preds = self.model(**self.forward_kwargs(batch))
        loss, probs = bce_with_logits_loss(preds.squeeze(1), batch['labels'], self.config['pos_wt'],
                                           return_probs=True)
        return loss, probs

    @torch.no_grad()
    def test_iter(self, batch):
        return self.model(**self.forward_kwargs(batch)).squeeze()
