"""HIP gives every stream that has been used one of GPU_MAX_HW_QUEUES hardware queues, round-robin; a stream that shares the
main stream's queue runs behind it, not beside it.  This probe puts N other streams to use first (data loaders, other models,
a framework's own), then asks the package for its side stream and reports whether a spin kernel on it overlaps one on the
current stream -- and, with --step, what the fp32 training step takes.  With the package's stream chosen by measurement
(_lib.shared_stream) the answer must not depend on N; with a fresh torch.cuda.Stream() per model it did (N = 7: 16.0 instead
of 13.5 ms per step).   Usage: python tests/tools/stream_queue_probe.py N [--step]"""
import sys, time, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device('cuda:0')
others = [torch.cuda.Stream(device=dev) for _ in range(n)]
for s in others:
    with torch.cuda.stream(s):
        torch.zeros(8, device=dev).add_(1)
torch.cuda.synchronize()
side = _lib.shared_stream(dev, 'side')
print('streams in use before: %d  side stream overlaps the current stream: %s' % (n, _lib._overlaps(side, dev)), flush=True)
if '--step' in sys.argv:
    import bench
    from meme_challenge_amd.model import UniterConfig, UniterModel
    from meme_challenge_amd.meme_uniter import MemeUniter
    from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler
    from meme_challenge_amd.utils import make_synthetic_batch
    torch.manual_seed(0)
    cfg = UniterConfig.from_dict(bench.BASE)
    model = MemeUniter(UniterModel(cfg, img_dim=2048), cfg.hidden_size, 1).to(dev).train()
    enc = model.uniter_model; enc.set_dropout_seed(1, 0)
    batch = make_synthetic_batch(16, 128, 36, seed=1, device=dev)
    config = dict(optimizer='adam', lr=3e-5, beta1=0.9, beta2=0.999, weight_decay=1e-3, gradient_accumulation=1, max_grad_norm=5,
                  pos_wt=1.8, loss_func='bce_logits', scheduler='warmup_cosine', warmup_steps=500, max_epoch=30)
    opt = FusedAdam(model, lr=3e-5, weight_decay=1e-3); opt.overlap_encoder = enc
    step = TrainStep(model, opt, get_scheduler(opt, config, steps_per_epoch=1000), config)
    for _ in range(15): step.train_iter(batch, iters=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step.train_iter(batch, iters=0)
    torch.cuda.synchronize()
    assert enc._side_stream is side
    print('fp32 step %.3f ms' % ((time.perf_counter() - t0) / 30 * 1e3), flush=True)
