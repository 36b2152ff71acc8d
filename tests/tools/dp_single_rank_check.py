"""Child process of tests/test_dp_gpu.py: one rank, backend nccl (= RCCL), UNITER_DP_FORCE=1.
Runs the same two training steps (a) without a process group and (b) through the model's backward
hooks + dp.GradSync + RCCL all-reduce, and prints one JSON line with what the parent asserts on."""
import json
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def run(payload, precision, cfgd, B, T, R, max_grad_norm, sparse=False):
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler
    from meme_challenge_amd.utils import make_synthetic_batch
    from meme_challenge_amd import dp
    torch.manual_seed(0)
    cfg = UniterConfig.from_dict(cfgd)
    model = MemeUniter(UniterModel(cfg, img_dim=2048), cfg.hidden_size, 1).cuda().train()
    enc = model.uniter_model
    enc.precision = precision
    enc.set_dropout_seed(77, 0)
    config = dict(optimizer='adam', lr=1e-3, beta1=0.9, beta2=0.999, weight_decay=1e-3, gradient_accumulation=1,
                  max_grad_norm=max_grad_norm, pos_wt=1.8, loss_func='bce_logits', scheduler='warmup_cosine',
                  warmup_steps=2, max_epoch=1)
    opt = FusedAdam(model, lr=config['lr'], weight_decay=config['weight_decay'])
    opt.overlap_encoder = enc
    sched = get_scheduler(opt, config, steps_per_epoch=10)
    sync = dp.attach(model, payload=payload, sparse_embeddings=sparse) if payload else None
    step = TrainStep(model, opt, sched, config, grad_sync=sync)
    batch = make_synthetic_batch(B, T, R, seed=5, device='cuda')
    info = {}
    # step 1: keep the gradients (zero_grads happens inside the optimizer step, so snapshot through a hook-free rerun)
    store = model.param_store()
    for it in range(2):
        if it == 1 and sync is not None:
            pass
        step.train_iter(batch, iters=0)
    opt.join()
    torch.cuda.synchronize()
    if sync is not None:
        side = enc._side_stream.cuda_stream
        info.update(launched=sync.launched, n_buckets=len(store.bucket_ranges),
                    ranges=[list(r) for r in store.bucket_ranges],
                    on_side_stream=[s == side for s in sync.launch_streams], payload=sync.payload,
                    sparse_steps=sync.sparse_steps, sync_ranges=[list(r) for r in sync.ranges])
    return store.flat_params.detach().clone(), float(step.last_loss.item()), info


def main():
    from common import TINY
    cfgd = dict(TINY, vocab_size=28996, max_position_embeddings=512, num_hidden_layers=3, hidden_size=256,
                num_attention_heads=4, intermediate_size=512)
    shape = (4, 24, 12)
    out = {}
    ref_clip, loss_ref, _ = run(None, 'fp32', cfgd, *shape, max_grad_norm=5)
    ref_clip2, _, _ = run(None, 'fp32', cfgd, *shape, max_grad_norm=5)
    out['plain_rerun_maxdiff'] = float((ref_clip2 - ref_clip).abs().max().item())      # float-atomic weight gradients: order noise
    ref_noclip, _, _ = run(None, 'fp32', cfgd, *shape, max_grad_norm=0)
    ref_b16, _, _ = run(None, 'bf16', cfgd, *shape, max_grad_norm=5)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ['UNITER_DP_FORCE'] = '1'
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    got, loss, info = run('fp32', 'fp32', cfgd, *shape, max_grad_norm=5)
    out['fp32_clip_equal'] = bool(torch.equal(got, ref_clip))
    out['fp32_clip_maxdiff'] = float((got - ref_clip).abs().max().item())
    out['loss_diff'] = abs(loss - loss_ref)
    out['info'] = info
    got, _, info2 = run('fp32', 'fp32', cfgd, *shape, max_grad_norm=0)      # per-block waits instead of finish()
    out['fp32_noclip_maxdiff'] = float((got - ref_noclip).abs().max().item())
    got, _, info3 = run('bf16', 'bf16', cfgd, *shape, max_grad_norm=5)
    d = (got - ref_b16).abs().max().item()
    out['bf16_payload_maxdiff'] = float(d)
    out['bf16_payload_scale'] = float((ref_b16 - ref_clip).abs().max().item())
    out['bf16_info'] = info3
    # sparse word-embedding exchange: RCCL all-gather of (ids, rows), summed in rank order over cleared rows
    got, _, info4 = run('fp32', 'fp32', cfgd, *shape, max_grad_norm=5, sparse=True)
    out['fp32_sparse_maxdiff'] = float((got - ref_clip).abs().max().item())
    out['sparse_info'] = info4
    got, _, info5 = run('bf16', 'bf16', cfgd, *shape, max_grad_norm=5, sparse=True)
    out['bf16_sparse_maxdiff'] = float((got - ref_b16).abs().max().item())
    out['bf16_sparse_steps'] = info5['sparse_steps']
    dist.destroy_process_group()
    print('DPCHECK ' + json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
