"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the reference's training-step semantics around the model:
loss, gradient-accumulation quirk, gradient averaging, global-norm clipping,
parameter-group split, Adam / AdamW update and LR schedules.
"""
import math

import torch


def bce_with_logits(logits, labels, pos_wt):
    # nn.BCEWithLogitsLoss(pos_weight=[pos_wt]) mean-reduced,
    # train_template.py:64-65,98-99 (preds.squeeze(1), labels.float())
    x = logits.squeeze(1) if logits.dim() == 2 else logits
    y = labels.to(torch.float32)
    lw = 1.0 + (pos_wt - 1.0) * y
    loss = (1.0 - y) * x + lw * (torch.log1p(torch.exp(-x.abs()))
                                 + torch.clamp(-x, min=0.0))
    return loss.mean()


def no_decay(name):
    # utils/optim_utils.py:16 -- substring match on the parameter NAME; note
    # img_layer_norm.weight / pos_layer_norm.weight do NOT match and are decayed.
    return any(nd in name for nd in ('bias', 'LayerNorm.bias', 'LayerNorm.weight'))


def cosine_warmup_lambda(step, warmup, total, num_cycles=0.5):
    # transformers.get_cosine_schedule_with_warmup (train_template.py:80-82)
    if step < warmup:
        return float(step) / float(max(1, warmup))
    prog = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * prog)))


def linear_warmup_lambda(step, warmup, total):
    # transformers.get_linear_schedule_with_warmup (train_template.py:77-79)
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))


class AdamOracle:
    """torch.optim.Adam (coupled L2: g += wd*p) or AdamW (decoupled) exactly as
    utils/optim_utils.py:9-46 configures them (eps 1e-8, no amsgrad)."""

    def __init__(self, named_params, lr, betas=(0.9, 0.999), weight_decay=1e-3,
                 adamw=False, eps=1e-8):
        self.names = [n for n, _ in named_params]
        self.p = {n: p for n, p in named_params}
        self.m = {n: torch.zeros_like(p) for n, p in named_params}
        self.v = {n: torch.zeros_like(p) for n, p in named_params}
        self.t = 0
        self.base_lr = lr
        self.lr = lr
        self.b1, self.b2 = betas
        self.wd = weight_decay
        self.adamw = adamw
        self.eps = eps

    @torch.no_grad()
    def step(self, grads):
        self.t += 1
        b1, b2 = self.b1, self.b2
        bc1 = 1.0 - b1 ** self.t
        bc2 = 1.0 - b2 ** self.t
        for n in self.names:
            if n not in grads:      # torch optimizers skip params with grad None
                continue            # (e.g. mask_embedding when img_masks is None)
            p, g = self.p[n], grads[n]
            wd = 0.0 if no_decay(n) else self.wd
            if self.adamw:
                p.mul_(1.0 - self.lr * wd)
            elif wd != 0.0:
                g = g + wd * p
            self.m[n].mul_(b1).add_(g, alpha=1.0 - b1)
            self.v[n].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            denom = (self.v[n].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[n], denom, value=-self.lr / bc1)


def clip_coef(total_norm, max_norm):
    # torch.nn.utils.clip_grad_norm_: coef = max_norm/(norm+1e-6) clamped to 1
    c = float(max_norm) / (float(total_norm) + 1e-6)
    return min(c, 1.0)


def train_iterations(model_fn, params, opt, batches, labels, pos_wt,
                     gradient_accumulation, max_grad_norm, lr_lambda, iters0=0):
    """TrainerTemplate.calculate_loss semantics, train_template.py:95-109:
    the modulo test is on the per-epoch iteration index starting at 0, so
    iteration 0 steps immediately; accumulated grads are divided by
    ``gradient_accumulation`` (average_gradients, :89-92); then clip, step,
    scheduler.step, zero_grad.  ``model_fn(batch) -> logits`` must be
    differentiable w.r.t. the tensors in ``params`` (dict name->leaf tensor).
    Returns list of losses."""
    acc = {}
    losses = []
    sched_step = 0
    # LambdaLR applies lambda(0) at construction: with warm-up the first
    # optimizer step runs at lr = base_lr * lambda(0) (= 0 for warmup > 0).
    opt.lr = opt.base_lr * lr_lambda(0)
    for it, (batch, y) in enumerate(zip(batches, labels)):
        iters = iters0 + it
        logits = model_fn(batch)
        loss = bce_with_logits(logits, y, pos_wt)
        gs = torch.autograd.grad(loss, [params[n] for n in params], allow_unused=True)
        for n, g in zip(params, gs):
            if g is not None:
                acc[n] = acc[n] + g if n in acc else g.clone()
        if iters % gradient_accumulation == 0:
            grads = {n: a / gradient_accumulation for n, a in acc.items()}
            total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).item()
            c = clip_coef(total, max_grad_norm)
            if c < 1.0:
                grads = {n: g * c for n, g in grads.items()}
            opt.step(grads)
            sched_step += 1
            opt.lr = opt.base_lr * lr_lambda(sched_step)
            acc = {}
        losses.append(loss.item())
    return losses
