"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's optimal-transport distance (IPOT), the checker of
meme_challenge_amd/ot.py.  Nothing in the product path imports this file.

Restates /root/reference/model/ot.py: cost_matrix_cosine (:11-21), ipot (:36-66) and optimal_transport_dist (:69-85), reached
from model/pretrain.py:168-193 (forward_itm with ot_inputs).  Pinned by tests/golden/ot_golden.npz, which
tests/golden/make_ot_golden.py wrote by importing the reference itself (distance, transport plan, cost matrix and the gradients
w.r.t. both embeddings; tests/test_ot_cpu.py).  The reference's ipot only runs with k == 1 (with k > 1 its second inner
iteration multiplies a [b, 1, m] sigma, ot.py:62, and raises): this restatement is written for k == 1 and says so.
"""
import torch


def cost_matrix_cosine(x, y, eps=1e-5):
    # ot.py:11-21: rows normalised with F.normalize (norm clamped at eps), cosine distance of every pair
    xn = x / x.norm(dim=-1, keepdim=True).clamp_min(eps)
    yn = y / y.norm(dim=-1, keepdim=True).clamp_min(eps)
    return 1 - xn @ yn.transpose(1, 2)


def ipot(C, x_len, x_pad, y_len, y_pad, joint_pad, beta, iteration, k=1):
    # ot.py:36-66 (no gradient): C [B, M, N]; returns the transport plan T [B, N, M]
    if k != 1:
        raise ValueError('ipot: the reference runs with k == 1 only (ot.py:62 fails for k > 1)')
    b, m, n = C.shape
    sigma = torch.ones(b, m, dtype=C.dtype) / x_len.unsqueeze(1)
    T = torch.ones(b, n, m, dtype=C.dtype)
    A = torch.exp(-C.transpose(1, 2) / beta)
    sigma = sigma.masked_fill(x_pad, 0)
    jp = joint_pad.transpose(1, 2)
    T = T.masked_fill(jp, 0)
    A = A.masked_fill(jp, 0)
    xl, yl = x_len.view(b, 1, 1), y_len.view(b, 1, 1)
    x_mask = (x_pad.to(C.dtype) * 1e4).unsqueeze(1)           # [b, 1, m]
    y_mask = (y_pad.to(C.dtype) * 1e4).unsqueeze(1)           # [b, 1, n]
    delta = None
    for _ in range(iteration):
        Q = A * T                                             # [b, n, m]
        sigma = sigma.view(b, m, 1)
        delta = 1 / (yl * Q.matmul(sigma).view(b, 1, n) + y_mask)
        sigma = 1 / (xl * delta.matmul(Q) + x_mask)           # [b, 1, m]
        T = delta.view(b, n, 1) * Q * sigma
    return T.masked_fill(jp, 0)


def optimal_transport_dist(txt_emb, img_emb, txt_pad, img_pad, beta=0.5, iteration=50, k=1):
    # ot.py:69-85: the gradient reaches the embeddings through the cost matrix only (T is detached)
    cost = cost_matrix_cosine(txt_emb, img_emb)
    joint_pad = txt_pad.unsqueeze(-1) | img_pad.unsqueeze(-2)
    cost = cost.masked_fill(joint_pad, 0)
    txt_len = (txt_pad.size(1) - txt_pad.sum(dim=1)).to(cost.dtype)
    img_len = (img_pad.size(1) - img_pad.sum(dim=1)).to(cost.dtype)
    with torch.no_grad():
        T = ipot(cost.detach(), txt_len, txt_pad, img_len, img_pad, joint_pad, beta, iteration, k)
    return torch.diagonal(cost.matmul(T), dim1=1, dim2=2).sum(-1), T, cost
