import time, sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd.model import UniterConfig, UniterModel
from meme_challenge_amd.meme_uniter import MemeUniter
from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler
from meme_challenge_amd.utils import make_synthetic_batch
import bench
for prec in ('bf16', 'fp32'):
    torch.manual_seed(0)
    cfg = UniterConfig.from_dict(bench.BASE)
    dev = torch.device('cuda:0')
    model = MemeUniter(UniterModel(cfg, img_dim=2048), cfg.hidden_size, 1).to(dev).train()
    enc = model.uniter_model; enc.precision = prec; enc.set_dropout_seed(1, 0)
    batch = make_synthetic_batch(16, 128, 36, seed=1, device=dev)
    config = dict(optimizer='adam', lr=3e-5, beta1=0.9, beta2=0.999, weight_decay=1e-3, gradient_accumulation=1, max_grad_norm=5, pos_wt=1.8,
                  loss_func='bce_logits', scheduler='warmup_cosine', warmup_steps=500, max_epoch=30)
    opt = FusedAdam(model, lr=3e-5, weight_decay=1e-3); opt.overlap_encoder = enc
    step = TrainStep(model, opt, get_scheduler(opt, config, steps_per_epoch=1000), config)
    for _ in range(20): step.train_iter(batch, iters=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step.train_iter(batch, iters=0)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(prec, 'host enqueue %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) / 50 * 1e3, (t2 - t0) / 50 * 1e3))
    # pure host cost: one step enqueued into an empty queue
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); step.train_iter(batch, iters=0); ts.append(time.perf_counter() - t0)
    ts.sort(); print(prec, 'one step into an empty queue: median %.2f ms, min %.2f ms' % (ts[10] * 1e3, ts[0] * 1e3))
    import cProfile, pstats, io
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(20): step.train_iter(batch, iters=0)
    pr.disable(); torch.cuda.synchronize()
    sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats('tottime').print_stats(18); print(sio.getvalue()[:3500])
