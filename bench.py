#!/usr/bin/env python
"""Headline benchmark: train samples/sec of UNITER-base fine-tuning
(BASELINE.json: batch 16 per GPU, 36 regions x 2048-d, 128 text tokens,
12 layers, fp32) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = forward + BCE loss + backward (+ RCCL gradient all-reduce for N>1)
+ global-norm clip + Adam + LR schedule, dropout ON (p=0.1), synthetic inputs
resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')      # before the HIP runtime initialises: see meme_challenge_amd/__init__.py

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

BASE = dict(attention_probs_dropout_prob=0.1, hidden_act='gelu', hidden_dropout_prob=0.1,
            hidden_size=768, initializer_range=0.02, intermediate_size=3072,
            max_position_embeddings=512, num_attention_heads=12, num_hidden_layers=12,
            type_vocab_size=2, vocab_size=28996)       # == reference config/uniter-base.json
LARGE = dict(BASE, hidden_size=1024, intermediate_size=4096, num_attention_heads=16, num_hidden_layers=24)

PEAK_TFLOPS = {'f32': 157.3, 'bf16': 2500.0}       # MI355X_MICROARCH.md: fp32 matrix (v_mfma_f32_32x32x2_f32)


def flops_per_step(cfg, B, T, R, L=None, lens=None):
    """Algorithmic FLOPs: 2MNK per GEMM, bwd = 2x fwd, attention
    QK^T and PV included, elementwise excluded.  Returns (total, ffn_only, ffn_up_fwd)."""
    H, I, nl = cfg['hidden_size'], cfg['intermediate_size'], cfg['num_hidden_layers']
    L = T + R if L is None else L         # compacted joint length (max over the batch of tl + nbb)
    M = B * L
    sq = B * L * L
    if lens is not None:                  # token packing: only the valid positions are computed
        M = sum(lens)
        sq = sum(n * n for n in lens)
    per_layer = 2 * M * H * 3 * H + 2 * M * H * H + 2 * 2 * M * H * I + 2 * 2 * sq * H
    fwd = nl * per_layer + 2 * B * R * 2048 * H + 2 * B * H * H + 2 * B * H
    ffn_up_fwd = 2 * M * H * I
    return 3 * fwd, 3 * nl * 2 * ffn_up_fwd, ffn_up_fwd


def cpu_baseline(seconds_budget=25.0):
    """Reported baseline only: the CPU oracle (port of the reference forward; autograd
    backward) on BASELINE config 1 (B=4, T=64, R=36, UNITER-base, dropout on), timed on this
    box's host cores.  A bounded sample: 1 warm-up + up to 10 steps within the budget."""
    from oracle import uniter_oracle as O
    from oracle import step_oracle as S
    nthreads = min(torch.get_num_threads(), 32)      # more threads than ~32 slow the small CPU ops down
    torch.set_num_threads(nthreads)
    sd = {k: v.requires_grad_(True) for k, v in O.synth_state_dict(BASE, seed=0).items()}
    b = O.synth_batch(4, 64, 36, seed=1234)
    kw = dict(img_feat=b['img_feat'], img_pos_feat=b['img_pos_feat'], input_ids=b['input_ids'],
              position_ids=b['position_ids'], attention_mask=b['attn_mask'],
              gather_index=b['gather_index'], output_all_encoded_layers=False)

    def step(i):
        drop = O.DropSpec(1234, i, 0.1, 0.1)
        loss = S.bce_with_logits(O.meme_uniter_forward(sd, BASE, drop=drop, **kw), b['labels'], 1.8)
        grads = torch.autograd.grad(loss, [v for v in sd.values()], allow_unused=True)
        return grads
    step(0)
    t0 = time.perf_counter()
    n = 0
    while n < 10 and (time.perf_counter() - t0) < seconds_budget:
        step(n + 1)
        n += 1
    dt = (time.perf_counter() - t0) / max(n, 1)
    return {'value': round(4.0 / dt, 3), 'unit': 'samples/s', 'cores': nthreads, 'kind': 'port',
            'sample': 'config 1: UNITER-base B=4 T=64 R=36 fwd+bwd, %d steps after 1 warm-up, '
                      'oracle/uniter_oracle.py (torch CPU fp32, dropout on), %.3f s/step' % (n, dt)}


def pmc_traffic(args, M, cfgd):
    """Memory-side bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes
    (separate FETCH_SIZE / WRITE_SIZE runs of this same command, tests/tools/run_profile.sh ->
    tests/tools/pmc_to_traffic.py; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
    Counters cannot be read from inside a timed run, so the figure is a profile artefact: it is
    reported only when the profiled shape is the one being benchmarked, else null."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r01_pmc_traffic.json')
    try:
        rec = json.load(open(path))['ffn_up_fwd']
    except (OSError, KeyError, ValueError):
        return None
    shape = rec.get('shape', {})
    if args.precision != 'fp32' or (shape.get('M'), shape.get('N'), shape.get('K')) != (
            M, cfgd['intermediate_size'], cfgd['hidden_size']):
        return None
    return int(rec['traffic_bytes'])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch')
    ap.add_argument('--txt_len', type=int, default=128)
    ap.add_argument('--num_bb', type=int, default=36)
    ap.add_argument('--model', choices=['base', 'large'], default='base')
    ap.add_argument('--precision', choices=['fp32', 'bf16', 'bf16_hybrid'], default='fp32',
                    help="fp32 (BASELINE configs[1], default) or bf16: bf16 MFMA for the dense GEMMs, fp32 elsewhere (configs[2])")
    ap.add_argument('--workload', choices=['finetune', 'multitask'], default='finetune',
                    help='finetune = MemeUniter step (BASELINE configs[1-3], default); multitask = UNITER + ITM/MLM/MRFR '
                         'heads, one task drawn per step (configs[4]; use --batch 32)')
    ap.add_argument('--ragged', action='store_true',
                    help='per-sample lengths tl ~ U{8..T}, nbb ~ U{10..R} (SURVEY 8(d) D1: correctness/reporting variant, not the headline)')
    ap.add_argument('--packed', action='store_true',
                    help='token packing: compute the valid positions only (pays off with --ragged; identical results)')
    ap.add_argument('--no_cpu_baseline', action='store_true')
    ap.add_argument('--no_side_stream', action='store_true')
    ap.add_argument('--no_adam_overlap', action='store_true',
                    help='run the optimizer step as one launch on the main stream instead of block by block beside the next forward')
    ap.add_argument('--prof_kind', type=int, default=1, help='UNITER_K_* kind timed with HIP events (1 = FFN-up fwd GEMM)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if rank == 0:
            print('warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE' % (args.gpus, world), file=sys.stderr)
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or 'RANK' in os.environ
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from meme_challenge_amd import _lib
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler, sync_step
    from meme_challenge_amd.utils import make_synthetic_batch, make_synthetic_pretrain_batch
    from meme_challenge_amd import dp
    import ctypes as C

    cfgd = BASE if args.model == 'base' else LARGE
    torch.manual_seed(0)                                   # identical init on every rank
    cfg = UniterConfig.from_dict(cfgd)
    B, T, R = args.batch, args.txt_len, args.num_bb
    lens = {}
    if args.ragged:
        import numpy as np
        rng = np.random.Generator(np.random.PCG64(4321 + rank))
        lens = dict(txt_lens=[int(x) for x in rng.integers(8, T + 1, size=B)],
                    num_bbs=[int(x) for x in rng.integers(10, R + 1, size=B)])
    config = dict(optimizer='adam', lr=3e-5, beta1=0.9, beta2=0.999, weight_decay=1e-3,
                  gradient_accumulation=1, max_grad_norm=5, pos_wt=1.8, loss_func='bce_logits',
                  scheduler='warmup_cosine', warmup_steps=500, max_epoch=30)
    if args.workload == 'finetune':
        model = MemeUniter(UniterModel(cfg, img_dim=2048), cfg.hidden_size, 1).to(dev).train()
        encoder = model.uniter_model
        batch = make_synthetic_batch(B, T, R, seed=1234 + rank, device=dev, **lens)
    else:
        from meme_challenge_amd.pretrain import UniterForPretraining
        model = UniterForPretraining(cfg, img_dim=2048, img_label_dim=1601).to(dev).train()
        encoder = model.uniter
        tasks = ('itm', 'mlm', 'mrfr')
        batches = {t: make_synthetic_pretrain_batch(t, B, T, R, seed=1234 + rank, device=dev, **lens) for t in tasks}
        import random
        task_rng = random.Random(99)                       # same task sequence on every rank
    encoder.use_side_stream = not args.no_side_stream
    encoder.precision = args.precision
    encoder.pack_padded = args.packed
    encoder.set_dropout_seed(1234 + rank, 0)
    opt = FusedAdam(model, lr=config['lr'], weight_decay=config['weight_decay'])
    if not args.no_adam_overlap:
        opt.overlap_encoder = encoder
    sched = get_scheduler(opt, config, steps_per_epoch=1000)
    sync = None
    if use_dist:
        dp.broadcast_parameters(model)
        sync = dp.attach(model)
    if args.workload == 'finetune':
        step = TrainStep(model, opt, sched, config, grad_sync=sync)

        def one_step():
            step.train_iter(batch, iters=0)
    else:
        last = {}

        def one_step():
            task = task_rng.choice(tasks)
            if sync is not None:
                sync.prepare(True)
            loss = model(batches[task], task, compute_loss=True).mean()
            loss.backward()
            sync_step(opt, sync, 1, config['max_grad_norm'])
            sched.step()
            last['loss'] = loss.detach()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    lib = _lib.lib()
    handle = encoder._handle
    barrier()
    if args.prof_kind:
        _lib.check(lib.uniter_prof_enable(handle, args.prof_kind))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    n_launch, tot_ms = C.c_int(0), C.c_double(0.0)
    if args.prof_kind:
        _lib.check(lib.uniter_prof_collect(handle, C.byref(n_launch), C.byref(tot_ms)))
        _lib.check(lib.uniter_prof_enable(handle, 0))
    loss = float((step.last_loss if args.workload == 'finetune' else last['loss']).item())
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = B * world * args.steps / dt
        L_eff = int((batch if args.workload == 'finetune' else batches['itm'])
                    ['attn_mask' if args.workload == 'finetune' else 'attn_masks'].shape[1])
        cur = batch if args.workload == 'finetune' else batches['itm']
        total, ffn, ffn_up = flops_per_step(cfgd, B, T, R, L_eff, cur['seq_lens'] if args.packed else None)
        M_eff = sum(cur['seq_lens']) if args.packed else B * L_eff
        dt_name = 'f32' if args.precision == 'fp32' else 'bf16'
        peak = PEAK_TFLOPS[dt_name]
        out = {
            'metric': 'train samples/sec UNITER-%s (%d regions, %d tok)' % (args.model, R, T),
            'value': round(value, 2), 'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': dt_name, 'data': 'synthetic',
            'config': {'workload': ('UNITER-%s fine-tune step (fwd + BCE + bwd + clip + Adam, dropout 0.1), '
                                    if args.workload == 'finetune' else
                                    'UNITER-%s + ITM/MLM/MRFR heads, one task per step (fwd + loss + bwd + clip + Adam, '
                                    'dropout 0.1; BASELINE configs[4]; FLOP fractions count the encoder only), ') % args.model +
                                   'batch %d per GPU, %d regions x 2048, %d text tokens%s, %s'
                                   % (B, R, T, (' (ragged: joint length %d%s)' % (L_eff, ', packed' if args.packed else '')) if args.ragged else '',
                                      'fp32 (BASELINE configs[1])' if args.precision == 'fp32'
                                      else 'bf16 MFMA GEMMs / fp32 storage (BASELINE configs[2])'),
                       'global_batch': B * world, 'parallelism': 'dp%d' % world,
                       'side_stream_wgrad': not args.no_side_stream,
                       'optimizer_overlaps_next_forward': not args.no_adam_overlap,
                       'hip_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES')},
            'step_mfma_frac': round(total / (ms * 1e-3) / world * world / (peak * 1e12), 4),
            'ffn_roofline_frac': round(ffn / (ms * 1e-3) / (peak * 1e12), 4),
            'final_loss': round(loss, 5),
        }
        if args.prof_kind and n_launch.value > 0:
            avg_ms = tot_ms.value / n_launch.value
            ach = ffn_up / (avg_ms * 1e-3) / 1e12 if args.prof_kind == 1 else None
            out['roofline'] = {'bound': 'mfma', 'achieved': round(ach, 2) if ach else None, 'peak': peak,
                               'unit': 'TFLOP/s', 'frac': round(ach / peak, 4) if ach else None,
                               'traffic': pmc_traffic(args, M_eff, cfgd),
                               'kernel': ('gemm_f32_v3_kernel<64,64,false,false,TAG=1>' if args.precision == 'fp32'
                                          else 'gemm_bf16_kernel<...,false,false>') +
                                         ' (FFN-up fwd: M=%d N=%d K=%d, bias+GELU epilogue)'
                                         % (M_eff, cfgd['intermediate_size'], cfgd['hidden_size']),
                               'launches': n_launch.value, 'avg_ms': round(avg_ms, 4)}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
