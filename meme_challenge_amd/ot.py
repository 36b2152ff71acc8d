"""Optimal-transport distance of the ITM pretraining task: the reference's model/ot.py (`optimal_transport_dist`, :69-85, with
`cost_matrix_cosine` :11-21 and `ipot` :36-66) on the HIP library -- same name, arguments and defaults, same gradient (through
the cost matrix only: the reference detaches the transport plan).  The reference's ipot runs with k == 1 only (ot.py:62 raises
a shape error for k > 1); so does this one, with a ValueError that says so."""
import torch

from . import _lib
from ._lib import check, ptr


class _OtDist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, txt_emb, img_emb, txt_pad, img_pad, beta, iteration):
        lib = _lib.lib()
        B, M, D = txt_emb.shape
        N = img_emb.shape[1]
        x, y = txt_emb.contiguous(), img_emb.contiguous()
        xp, yp = txt_pad.to(torch.uint8).contiguous(), img_pad.to(torch.uint8).contiguous()
        dist = torch.empty(B, dtype=torch.float32, device=x.device)
        # (grad mode is off in here: a .contiguous() copy of a non-contiguous input has requires_grad False -- ask the node)
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        T = torch.empty(B, N, M, dtype=torch.float32, device=x.device) if need else None
        check(lib.uniter_ot_dist_fwd(ptr(x), ptr(y), ptr(xp), ptr(yp), ptr(dist), ptr(T) if T is not None else None, B, M, N, D,
                                     float(beta), int(iteration), _lib.cur_stream()), 'uniter_ot_dist_fwd')
        ctx.save_for_backward(x, y, T if T is not None else dist)
        ctx.has_T = T is not None
        return dist

    @staticmethod
    def backward(ctx, g):
        x, y, T = ctx.saved_tensors
        if not ctx.has_T:
            raise RuntimeError('optimal_transport_dist: backward without a saved transport plan')
        B, M, D = x.shape
        N = y.shape[1]
        dx, dy = torch.empty_like(x), torch.empty_like(y)
        g = g.to(torch.float32).contiguous()
        check(_lib.lib().uniter_ot_dist_bwd(ptr(x), ptr(y), ptr(T), ptr(g), ptr(dx), ptr(dy), B, M, N, D, _lib.cur_stream()),
              'uniter_ot_dist_bwd')
        return dx, dy, None, None, None, None


def optimal_transport_dist(txt_emb, img_emb, txt_pad, img_pad, beta=0.5, iteration=50, k=1):
    """[B, M, D], [B, N, D], [B, M] bool, [B, N] bool -> [B] (model/ot.py:69-85)."""
    if k != 1:
        raise ValueError('optimal_transport_dist: k must be 1 (the reference\'s ipot fails for k > 1, model/ot.py:62)')
    if txt_emb.dim() != 3 or img_emb.dim() != 3 or txt_emb.size(0) != img_emb.size(0) or txt_emb.size(2) != img_emb.size(2):
        raise ValueError('optimal_transport_dist: expected [B, M, D] and [B, N, D]')      # the asserts of ot.py:14-16
    if not txt_emb.is_cuda:
        raise RuntimeError('optimal_transport_dist: the HIP library computes on the GPU (no CPU path)')
    return _OtDist.apply(txt_emb.float(), img_emb.float(), txt_pad, img_pad, beta, iteration)
