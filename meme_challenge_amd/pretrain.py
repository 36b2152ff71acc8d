"""UNITER pretraining heads on the HIP path (BASELINE config 5: ITM / MLM / MRFR; plus MRC / MRC-kl).

Mirror of model/pretrain.py:19-233 and model/layer.py:188-233: `UniterForPretraining(config,
img_dim, img_label_dim)` with the reference's module / state_dict names (`uniter.*`,
`cls.predictions.*` with the decoder tied to `word_embeddings.weight`, `feat_regress.*` tied to
`img_linear.weight` used transposed, `region_classifier.*`, `itm_output.*`) and
`forward(batch, task, compute_loss=True)` for task in {'mlm', 'mrfr', 'itm', 'mrc', 'mrc-kl'}.  The OT
loss (a value the reference computes and discards) is out of scope and raises.

The heads are composed from small autograd nodes whose forward / backward are C-ABI calls
(GEMM, LayerNorm, row gather, cross-entropy, KL divergence, MSE); parameter gradients accumulate directly into
the flat gradient buffer, including both tied weights.
"""
import logging
from collections import defaultdict

import torch
from torch import nn

from . import _lib
from ._lib import check, ptr, UniterHipError
from .model import (UniterModel, UniterPreTrainedModel, HipLinear, ensure_store, _ensure_grad,
                    _mark_touched, _ParamLinear, _ParamLayerNorm)

logger = logging.getLogger(__name__)

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_DGELU, EPI_ADD = range(5)


# Precision of the heads' dense products: follows the encoder's.  It travels with the OWNING head module (attribute
# `head_precision`, set by UniterForPretraining on its own heads whenever its encoder's precision changes; each autograd node
# keeps the value it was built with for its backward) -- not through a module global: two models with different precisions in
# one process, or a direct call of forward_mlm / forward_itm, see their own setting.  'bf16': both operands rounded to bf16
# while they are staged, products on the bf16 matrix pipe, fp32 accumulation / epilogue / output (uniter_gemm_bf16_cfg) -- the
# tied MLM decoder [n_masked, 768] x [768, 28996] is 82 GFLOP per MLM step at B = 32.
def _head_precision(owner):
    return getattr(owner, 'head_precision', 'fp32')


def _gemm(akm, bkm, M, N, K, A, lda, B, ldb, Cc, ldc, epi=EPI_NONE, bias=None, aux_in=None, aux_out=None,
          ld_aux=0, beta=0, prec='fp32'):
    if M == 0 or N == 0 or K == 0:
        return
    lib, st = _lib.lib(), _lib.cur_stream()
    if prec != 'bf16':
        check(lib.uniter_gemm_f32(akm, bkm, M, N, K, ptr(A), lda, ptr(B), ldb, ptr(Cc), ldc, epi, ptr(bias),
                                  ptr(aux_in), ptr(aux_out), ld_aux, beta, st), 'uniter_gemm_f32')
        return
    K1 = K if (K % 64 == 0 or (akm and bkm)) else K // 64 * 64
    if K1 != K and not (epi == EPI_NONE and not akm and K1 > 0 and A.is_contiguous() and B.is_contiguous()):
        K1 = K              # (uniter_gemm_bf16_cfg itself runs such a shape on the exact fp32 kernel)
    check(lib.uniter_gemm_bf16_cfg(0, akm, bkm, M, N, K1, ptr(A), lda, ptr(B), ldb, ptr(Cc), ldc, epi, ptr(bias),
                                   ptr(aux_in), ptr(aux_out), ld_aux, beta, st), 'uniter_gemm_bf16_cfg')
    if K1 != K:
        # the bf16 kernel stages 64-deep k-tiles of a k-contiguous operand: the last K % 64 terms of the contraction (the
        # vocabulary size 28996 = 453 * 64 + 4 in the decoder's input gradient) are added by the fp32 kernel
        a_off = A.data_ptr() + 4 * K1
        b_off = B.data_ptr() + 4 * K1 * (ldb if bkm else 1)
        check(lib.uniter_gemm_f32(akm, bkm, M, N, K - K1, a_off, lda, b_off, ldb, ptr(Cc), ldc, EPI_NONE, None, None, None,
                                  0, 1, st), 'uniter_gemm_f32')


def _colsum(X, M, N, out):
    if M == 0:
        return
    lib = _lib.lib()
    n = lib.uniter_colsum_ws_bytes(M, N)
    ws = torch.empty(n, dtype=torch.uint8, device=X.device)
    check(lib.uniter_colsum_f32(ptr(X), M, N, N, ptr(out), 1, ptr(ws), n, _lib.cur_stream()), 'uniter_colsum_f32')


class _LinearFn(torch.autograd.Function):
    """y = act(x @ W^T + b)  (w_t=False, W stored [out,in]) or x @ W + b (w_t=True, W stored [in,out]:
    the MRFR output layer uses img_linear.weight transposed, model/pretrain.py:27,32)."""

    @staticmethod
    def forward(ctx, x, weight, bias, owner, names, gelu, w_t):
        x = x.contiguous()
        M, K = x.shape
        N = weight.shape[1] if w_t else weight.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        u = torch.empty_like(y) if gelu else None
        prec = ctx.prec = _head_precision(owner)
        if w_t:
            _gemm(0, 1, M, N, K, x, K, weight, N, y, N, EPI_BIAS, bias, prec=prec)
        else:
            _gemm(0, 0, M, N, K, x, K, weight, K, y, N, EPI_BIAS_GELU if gelu else EPI_BIAS, bias,
                  aux_out=u, ld_aux=N, prec=prec)
        ctx.save_for_backward(x, u)
        ctx.owner, ctx.names, ctx.w_t = owner, names, w_t
        ctx.wb = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, u = ctx.saved_tensors
        weight, bias = ctx.wb
        w_t = ctx.w_t
        dy = dy.contiguous()
        M, K = x.shape
        N = dy.shape[1]
        _ensure_grad(weight)
        _ensure_grad(bias)
        if u is not None:       # through the activation: dy <- dy * gelu'(u)
            g = torch.empty_like(dy)
            check(_lib.lib().uniter_dgelu_mul(ptr(dy), ptr(u), ptr(g), dy.numel(), _lib.cur_stream()),
                  'uniter_dgelu_mul')
            dy = g
        dx = None
        prec = ctx.prec
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if w_t:      # dx[m,k] = sum_n dy[m,n] W[k,n]
                _gemm(0, 0, M, K, N, dy, N, weight, N, dx, K, prec=prec)
            else:        # dx[m,k] = sum_n dy[m,n] W[n,k]
                _gemm(0, 1, M, K, N, dy, N, weight, K, dx, K, prec=prec)
        if w_t:          # dW[k,n] += sum_m x[m,k] dy[m,n]
            _gemm(1, 1, K, N, M, x, K, dy, N, weight.grad, N, beta=1, prec=prec)
        else:            # dW[n,k] += sum_m dy[m,n] x[m,k]
            _gemm(1, 1, N, K, M, dy, N, x, K, weight.grad, K, beta=1, prec=prec)
        _colsum(dy, M, N, bias.grad)
        _mark_touched_names(ctx.owner, ctx.names)
        return dx, None, None, None, None, None, None


def _mark_touched_names(owner, params):
    _mark_touched(owner, params)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, owner):
        x = x.contiguous()
        M, H = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        check(_lib.lib().uniter_ln_fwd(ptr(x), None, ptr(weight), ptr(bias), None, ptr(y), ptr(mean), ptr(rstd),
                                       M, H, 0.0, 0, 0, 0, _lib.cur_stream()), 'uniter_ln_fwd')
        ctx.save_for_backward(x, mean, rstd)
        ctx.owner, ctx.wb = owner, (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        weight, bias = ctx.wb
        dy = dy.contiguous()
        M, H = x.shape
        _ensure_grad(weight)
        _ensure_grad(bias)
        dx = torch.empty_like(x)
        lib = _lib.lib()
        n = lib.uniter_ln_bwd_ws_bytes(M, H)
        ws = torch.empty(n, dtype=torch.uint8, device=x.device)
        if M > 0:
            check(lib.uniter_ln_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(weight), ptr(dx), ptr(dx),
                                    ptr(weight.grad), ptr(bias.grad), None, M, H, 0.0, 0, 0, 0, ptr(ws), n,
                                    _lib.cur_stream()), 'uniter_ln_bwd')
        _mark_touched(ctx.owner, (weight, bias))
        return dx, None, None, None


class _GatherRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, idx):
        src2 = src.contiguous().view(-1, src.shape[-1])
        n, H = idx.numel(), src2.shape[1]
        out = torch.empty(n, H, dtype=torch.float32, device=src.device)
        check(_lib.lib().uniter_row_gather(ptr(src2), ptr(idx), ptr(out), n, H, src2.shape[0], _lib.cur_stream()),
              'uniter_row_gather')
        ctx.save_for_backward(idx)
        ctx.shape = src.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        dsrc = torch.zeros(ctx.shape, dtype=torch.float32, device=dout.device)
        d2 = dsrc.view(-1, ctx.shape[-1])
        dout = dout.contiguous()
        check(_lib.lib().uniter_row_scatter_add(ptr(dout), ptr(idx), ptr(d2), idx.numel(), d2.shape[1], d2.shape[0],
                                                _lib.cur_stream()), 'uniter_row_scatter_add')
        return dsrc, None


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        logits = logits.contiguous()
        targets = targets.contiguous().to(torch.int64)
        n, Cn = logits.shape
        loss = torch.empty(n, dtype=torch.float32, device=logits.device)
        lse = torch.empty_like(loss)
        check(_lib.lib().uniter_cross_entropy_fwd(ptr(logits), ptr(targets), ptr(loss), ptr(lse), n, Cn, Cn,
                                                  _lib.cur_stream()), 'uniter_cross_entropy_fwd')
        ctx.save_for_backward(logits, targets, lse)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, targets, lse = ctx.saved_tensors
        n, Cn = logits.shape
        dl = torch.empty_like(logits)
        check(_lib.lib().uniter_cross_entropy_bwd(ptr(logits), ptr(targets), ptr(lse), ptr(dloss.contiguous()),
                                                  ptr(dl), n, Cn, Cn, _lib.cur_stream()), 'uniter_cross_entropy_bwd')
        return dl, None


class _MseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        pred, target = pred.contiguous(), target.contiguous().to(torch.float32)
        if pred.shape != target.shape:
            raise ValueError('mse_loss: shape mismatch %s vs %s' % (tuple(pred.shape), tuple(target.shape)))
        loss = torch.empty_like(pred)
        check(_lib.lib().uniter_mse_fwd(ptr(pred), ptr(target), ptr(loss), pred.numel(), _lib.cur_stream()),
              'uniter_mse_fwd')
        ctx.save_for_backward(pred, target)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        pred, target = ctx.saved_tensors
        dp = torch.empty_like(pred)
        check(_lib.lib().uniter_mse_bwd(ptr(pred), ptr(target), ptr(dloss.contiguous()), ptr(dp), pred.numel(),
                                        _lib.cur_stream()), 'uniter_mse_bwd')
        return dp, None


class _LinearPadNFn(torch.autograd.Function):
    """y = x @ W^T + b for an output width that is not a multiple of 4 (the region classifier's label
    dimension, 1601 for the UNITER detectors): the GEMM's k-major operands need 4-element rows, so the
    products run on copies padded with zero rows / columns; the padding never reaches the caller."""

    @staticmethod
    def forward(ctx, x, weight, bias, owner):
        x = x.contiguous()
        M, K = x.shape
        N = weight.shape[0]
        Np = (N + 3) // 4 * 4
        wp = torch.zeros(Np, K, dtype=torch.float32, device=x.device)
        wp[:N].copy_(weight.detach())
        bp = torch.zeros(Np, dtype=torch.float32, device=x.device)
        bp[:N].copy_(bias.detach())
        yp = torch.empty(M, Np, dtype=torch.float32, device=x.device)
        _gemm(0, 0, M, Np, K, x, K, wp, K, yp, Np, EPI_BIAS, bp)
        ctx.save_for_backward(x, wp)
        ctx.owner, ctx.wb, ctx.N = owner, (weight, bias), N
        return yp[:, :N].contiguous()

    @staticmethod
    def backward(ctx, dy):
        x, wp = ctx.saved_tensors
        weight, bias = ctx.wb
        N, (M, K) = ctx.N, x.shape
        Np = wp.shape[0]
        _ensure_grad(weight)
        _ensure_grad(bias)
        dyp = torch.zeros(M, Np, dtype=torch.float32, device=x.device)
        dyp[:, :N].copy_(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _gemm(0, 1, M, K, Np, dyp, Np, wp, K, dx, K)
        dwp = torch.zeros(Np, K, dtype=torch.float32, device=x.device)
        _gemm(1, 1, Np, K, M, dyp, Np, x, K, dwp, K, beta=1)
        weight.grad.add_(dwp[:N])
        dbp = torch.zeros(Np, dtype=torch.float32, device=x.device)
        _colsum(dyp, M, Np, dbp)
        bias.grad.add_(dbp[:N])
        _mark_touched_names(ctx.owner, (weight, bias))
        return dx, None, None, None


class _KlDivFn(torch.autograd.Function):
    """F.kl_div(F.log_softmax(logits, -1), target, reduction='none')   (model/pretrain.py:222-226)."""

    @staticmethod
    def forward(ctx, logits, target):
        logits, target = logits.contiguous(), target.contiguous().to(torch.float32)
        if logits.shape != target.shape:
            raise ValueError('kl_div: shape mismatch %s vs %s' % (tuple(logits.shape), tuple(target.shape)))
        n, Cn = logits.shape
        loss = torch.empty_like(logits)
        lse = torch.empty(n, dtype=torch.float32, device=logits.device)
        check(_lib.lib().uniter_kl_div_fwd(ptr(logits), ptr(target), ptr(loss), ptr(lse), n, Cn, Cn,
                                           _lib.cur_stream()), 'uniter_kl_div_fwd')
        ctx.save_for_backward(logits, target, lse)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        logits, target, lse = ctx.saved_tensors
        n, Cn = logits.shape
        dl = torch.empty_like(logits)
        check(_lib.lib().uniter_kl_div_bwd(ptr(logits), ptr(target), ptr(lse), ptr(dloss.contiguous()), ptr(dl),
                                           n, Cn, Cn, _lib.cur_stream()), 'uniter_kl_div_bwd')
        return dl, None


def _hard_region_labels(label_targets):
    """torch.max(label_targets[:, 1:], dim=-1)[1] + 1: the most likely non-background class
    (model/pretrain.py:227-228; the first maximum on ties)."""
    lt = label_targets.contiguous().to(torch.float32)
    n, Cn = lt.shape
    out = torch.empty(n, dtype=torch.int64, device=lt.device)
    check(_lib.lib().uniter_row_argmax(ptr(lt), n, Cn, Cn, 1, ptr(out), _lib.cur_stream()), 'uniter_row_argmax')
    return out


def hip_linear(x, lin, owner, gelu=False):
    return _LinearFn.apply(x, lin.weight, lin.bias, owner, (lin.weight, lin.bias), gelu, False)


# --------------------------------------------------------------------------- #
# modules (reference names)
# --------------------------------------------------------------------------- #
class GELU(nn.Module):
    pass            # placeholder so that nn.Sequential indices match (net.0 / net.1 / net.2)


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = _ParamLinear(config.hidden_size, config.hidden_size)
        self.LayerNorm = _ParamLayerNorm(config.hidden_size)

    def forward(self, h):
        h = hip_linear(h, self.dense, self, gelu=True)
        return _LayerNormFn.apply(h, self.LayerNorm.weight, self.LayerNorm.bias, self)


class _TiedDecoder(nn.Module):
    def __init__(self, weight):
        super().__init__()
        self.weight = weight            # the SAME Parameter as word_embeddings.weight (model/layer.py:212-215)


class BertLMPredictionHead(nn.Module):
    def __init__(self, config, bert_model_embedding_weights):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = _TiedDecoder(bert_model_embedding_weights)
        self.bias = nn.Parameter(torch.zeros(bert_model_embedding_weights.size(0)))

    def forward(self, h):
        h = self.transform(h)
        return _LinearFn.apply(h, self.decoder.weight, self.bias, self, (self.decoder.weight, self.bias), False, False)


class BertOnlyMLMHead(nn.Module):
    def __init__(self, config, bert_model_embedding_weights):
        super().__init__()
        self.predictions = BertLMPredictionHead(config, bert_model_embedding_weights)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


class RegionFeatureRegression(nn.Module):
    """model/pretrain.py:19-33: Linear -> GELU -> LN, then F.linear(h, img_linear.weight.t(), bias)."""

    def __init__(self, hidden_size, feat_dim, img_linear_weight):
        super().__init__()
        self.net = nn.Sequential(_ParamLinear(hidden_size, hidden_size), GELU(), _ParamLayerNorm(hidden_size))
        self.weight = img_linear_weight        # tied (registers as feat_regress.weight, shared storage)
        self.bias = nn.Parameter(torch.zeros(feat_dim))

    def forward(self, x):
        h = hip_linear(x, self.net[0], self, gelu=True)
        h = _LayerNormFn.apply(h, self.net[2].weight, self.net[2].bias, self)
        return _LinearFn.apply(h, self.weight, self.bias, self, (self.weight, self.bias), False, True)


class RegionClassification(nn.Module):
    """model/pretrain.py:36-47: Linear -> GELU -> LN -> Linear(hidden, label_dim), for MRC / MRC-kl."""

    def __init__(self, hidden_size, label_dim):
        super().__init__()
        self.net = nn.Sequential(_ParamLinear(hidden_size, hidden_size), GELU(), _ParamLayerNorm(hidden_size),
                                 _ParamLinear(hidden_size, label_dim))

    def forward(self, x):
        h = hip_linear(x, self.net[0], self, gelu=True)
        h = _LayerNormFn.apply(h, self.net[2].weight, self.net[2].bias, self)
        out = self.net[3]
        if out.weight.shape[0] % 4 == 0:
            return hip_linear(h, out, self)
        return _LinearPadNFn.apply(h, out.weight, out.bias, self)


class UniterForPretraining(UniterPreTrainedModel):
    def __init__(self, config, img_dim, img_label_dim):
        super().__init__(config)
        self.uniter = UniterModel(config, img_dim)
        self.cls = BertOnlyMLMHead(config, self.uniter.embeddings.word_embeddings.weight)
        self.feat_regress = RegionFeatureRegression(config.hidden_size, img_dim,
                                                    self.uniter.img_embeddings.img_linear.weight)
        self.region_classifier = RegionClassification(config.hidden_size, img_label_dim)
        self.itm_output = HipLinear(config.hidden_size, 2)
        self.apply(self.init_weights)

    def param_store(self):
        st = ensure_store(self)
        self.uniter._ensure_handle()
        return st

    def _sync_head_precision(self):
        """The heads' dense products follow the encoder's precision: stamped on this model's own head modules (the `owner`
        every _LinearFn node is built with), so forward_mlm / forward_itm / .. called directly see it as well."""
        p = 'bf16' if self.uniter.precision == 'bf16' else 'fp32'
        if getattr(self, '_heads_precision', None) != p:
            for mod in self.modules():
                if mod is not self.uniter and not any(mod is u for u in self.uniter.modules()):
                    mod.head_precision = p
            self._heads_precision = p

    def forward(self, batch, task, compute_loss=True):
        ensure_store(self)
        self._sync_head_precision()
        batch = defaultdict(lambda: None, batch)
        common = (batch['input_ids'], batch['position_ids'], batch['img_feat'], batch['img_pos_feat'],
                  batch['attn_masks'], batch['gather_index'])
        self._seq_lens = batch['seq_lens']        # host lengths for token packing (UniterModel.pack_padded), when the batch has them
        if task == 'mlm':
            return self.forward_mlm(*common, batch['txt_labels'], compute_loss)
        elif task == 'mrfr':
            return self.forward_mrfr(*common, batch['img_masks'], batch['img_mask_tgt'], batch['feat_targets'],
                                     compute_loss)
        elif task == 'itm':
            return self.forward_itm(*common, batch['targets'], batch['ot_inputs'], compute_loss)
        elif task.startswith('mrc'):
            return self.forward_mrc(*common, batch['img_masks'], batch['img_mask_tgt'], batch['label_targets'], task,
                                    compute_loss)
        raise ValueError('invalid task')

    @staticmethod
    def _masked_rows(hidden, mask):
        """_compute_masked_hidden (model/pretrain.py:129-133): rows of `hidden` [B,L,H] where
        `mask` [B,L'] (L' <= L) is set, in row-major order."""
        return UniterForPretraining._rows_at(hidden, UniterForPretraining._mask_positions(mask))

    @staticmethod
    def _mask_positions(mask):
        """[n, 2] (sample, position) of the set entries of `mask`, row-major: the ONE host synchronisation of a masked task's
        step (the count sizes everything behind it; the reference's boolean indexing synchronises the same way).  It depends
        on the batch only, so the forward_* methods take it BEFORE the encoder is enqueued -- the host then waits for the
        previous step's tail instead of for this step's forward -- and reuse it for the labels (a second boolean indexing
        would be a second synchronisation)."""
        return torch.nonzero(mask.bool(), as_tuple=False)

    @staticmethod
    def _rows_at(hidden, nz):
        L = hidden.shape[1]
        return _GatherRowsFn.apply(hidden, (nz[:, 0] * L + nz[:, 1]).contiguous())

    def forward_mlm(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                    txt_labels, compute_loss=True):
        self._sync_head_precision()
        nz = self._mask_positions(txt_labels != -1)
        seq = self.uniter(input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                          output_all_encoded_layers=False, seq_lens=getattr(self, '_seq_lens', None))
        T = input_ids.size(1)
        if seq.shape[1] < T:
            raise ValueError('sequence output shorter than the text length (model/pretrain.py:116)')
        masked = self._rows_at(seq, nz)                         # == seq[:, :T][mask]
        scores = self.cls(masked)
        if not compute_loss:
            return scores
        return _CrossEntropyFn.apply(scores, txt_labels[nz[:, 0], nz[:, 1]])      # == txt_labels[txt_labels != -1]

    def forward_mrfr(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                     img_masks, img_mask_tgt, feat_targets, compute_loss=True):
        self._sync_head_precision()
        nz = self._mask_positions(img_mask_tgt)
        seq = self.uniter(input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                          output_all_encoded_layers=False, img_masks=img_masks, seq_lens=getattr(self, '_seq_lens', None))
        masked = self._rows_at(seq, nz)
        pred = self.feat_regress(masked)
        if not compute_loss:
            return pred
        return _MseFn.apply(pred, feat_targets)

    def forward_mrc(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                    img_masks, img_mask_tgt, label_targets, task, compute_loss=True):
        """model/pretrain.py:205-233: region classification on the masked regions; 'mrc' trains against the most
        likely non-background detector class, 'mrc-kl' against the detector's soft labels."""
        self._sync_head_precision()
        nz = self._mask_positions(img_mask_tgt)
        seq = self.uniter(input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                          output_all_encoded_layers=False, img_masks=img_masks, seq_lens=getattr(self, '_seq_lens', None))
        masked = self._rows_at(seq, nz)
        scores = self.region_classifier(masked)
        if not compute_loss:
            return scores
        if 'kl' in task:
            return _KlDivFn.apply(scores, label_targets)
        # ignore_index=0 (:231) never fires: the targets are 1 + an argmax over the non-background classes
        return _CrossEntropyFn.apply(scores, _hard_region_labels(label_targets))

    def forward_itm(self, input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                    targets, ot_inputs=None, compute_loss=True):
        self._sync_head_precision()
        seq = self.uniter(input_ids, position_ids, img_feat, img_pos_feat, attention_mask, gather_index,
                          output_all_encoded_layers=False, seq_lens=getattr(self, '_seq_lens', None))
        # model/pretrain.py:168-193 computes the optimal-transport distance of the batch and then returns the ITM loss / scores
        # alone (both `return ..., ot_loss` lines are commented out there): nothing a caller sees depends on it.  Here it is
        # computed only on request (`compute_ot_loss = True`) and left in `self.ot_loss` as (positive pairs, negative pairs).
        self.ot_loss = None
        if ot_inputs is not None and getattr(self, 'compute_ot_loss', False):
            from .ot import optimal_transport_dist
            b, tl, il = seq.size(0), input_ids.size(1), img_feat.size(1)
            max_l = max(int(ot_inputs['scatter_max']) + 1, tl + il)
            idx = ot_inputs['ot_scatter'].unsqueeze(-1).expand_as(seq)
            ctx_emb = torch.zeros(b, max_l, seq.size(-1), dtype=seq.dtype, device=seq.device).scatter_(dim=1, index=idx, src=seq)
            ot_dist = optimal_transport_dist(ctx_emb[:, :tl, :].float(), ctx_emb[:, tl:tl + il, :].float(), ot_inputs['txt_pad'],
                                             ot_inputs['img_pad']).to(seq.dtype)
            self.ot_loss = (ot_dist.masked_select(targets == 1), ot_dist.masked_select(targets == 0))
        elif ot_inputs is not None and not getattr(self, '_warned_ot', False):
            logger.info('forward_itm: ot_inputs given -- the reference discards the OT distance it computes from them; '
                        'set compute_ot_loss = True to have it in self.ot_loss')
            self._warned_ot = True
        pooled = self.uniter.pooler(seq)
        scores = self.itm_output(pooled)
        if not compute_loss:
            return scores
        return _CrossEntropyFn.apply(scores, targets)
