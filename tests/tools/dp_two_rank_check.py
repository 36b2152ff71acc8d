"""Child process of tests/test_dp_gpu.py::test_two_ranks_share_the_batch: one of TWO ranks (both on cuda:0, backend gloo --
RCCL wants one device per rank, gloo moves GPU tensors through the host) running the HIP backward with its hooks and
dp.GradSync on ITS HALF of a batch.  Rank 0 also runs the whole batch alone and prints how far the exchanged, averaged
gradients are from the big-batch gradients (the data-parallel invariant, on the real kernels)."""
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def build(cfgd):
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    torch.manual_seed(0)
    cfg = UniterConfig.from_dict(cfgd)
    model = MemeUniter(UniterModel(cfg, img_dim=2048), cfg.hidden_size, 1).cuda().train()
    model.uniter_model.set_dropout_seed(11, 0)
    return model


def grads_of(model, batch, sync):
    from meme_challenge_amd.trainer import TrainStep, bce_with_logits_loss
    store = model.param_store()
    store.zero_grads()
    if sync is not None:
        sync.prepare(True, token_ids=batch['input_ids'])
    preds = model(**TrainStep.forward_kwargs(batch))
    loss = bce_with_logits_loss(preds.squeeze(1), batch['labels'], 1.8)
    loss.backward()
    if sync is not None:
        sync.finish()
    torch.cuda.synchronize()
    return store.flat_grads.detach().clone(), float(loss.item())


def main():
    from common import TINY
    from meme_challenge_amd import dp
    from meme_challenge_amd.utils import make_synthetic_batch
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    # no dropout: a mask is a function of the row index inside the batch a rank sees, so shards and the whole batch would
    # draw different masks; everything else (ragged lengths, masks, gather) is exercised
    cfgd = dict(TINY, vocab_size=28996, max_position_embeddings=512, num_hidden_layers=3, hidden_size=256,
                num_attention_heads=4, intermediate_size=512, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    B, T, R = 4, 24, 12
    full = make_synthetic_batch(B, T, R, seed=5, device='cuda', txt_lens=[24, 17, 9, 24], num_bbs=[12, 5, 12, 8])
    per = B // world
    shard = {k: (v[rank * per:(rank + 1) * per] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == B else v)
             for k, v in full.items()}
    if 'seq_lens' in shard and not torch.is_tensor(shard['seq_lens']):
        shard['seq_lens'] = list(full['seq_lens'])[rank * per:(rank + 1) * per]
    out = {}
    for name, prec, payload, sparse in (('fp32', 'fp32', 'fp32', False), ('bf16', 'bf16', 'fp32', False),
                                        ('bf16_payload', 'bf16', 'bf16', False), ('fp32_sparse', 'fp32', 'fp32', True),
                                        ('bf16_payload_sparse', 'bf16', 'bf16', True)):
        model = build(cfgd)
        model.uniter_model.precision = prec
        sync = dp.attach(model, payload=payload, sparse_embeddings=sparse)
        g, loss = grads_of(model, shard, sync)
        g = g / world                                       # what the optimizer's grad_scale = 1 / world applies
        losses = [None] * world
        dist.all_gather_object(losses, loss)
        if rank == 0:
            ref_model = build(cfgd)
            ref_model.uniter_model.precision = prec
            g_ref, loss_ref = grads_of(ref_model, full, None)
            scale = g_ref.abs().max().item()
            out[name] = dict(maxdiff=(g - g_ref).abs().max().item(), scale=scale,
                             rel_rms=((g - g_ref).norm() / g_ref.norm()).item(),
                             loss_mean=sum(losses) / world, loss_ref=loss_ref, buckets=len(sync.launched),
                             sparse_steps=sync.sparse_steps)
        dist.barrier()
    if rank == 0:
        print('DP2CHECK ' + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
