"""Potential of replaying the whole training step as one hipGraph (dropout offset / lr frozen: speed probe only)."""
import sys, time, torch
sys.path.insert(0, '.')
from meme_challenge_amd.model import UniterConfig, UniterModel
from meme_challenge_amd.meme_uniter import MemeUniter
from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler
from meme_challenge_amd.utils import make_synthetic_batch
from bench import BASE
dev = torch.device('cuda')
torch.manual_seed(0)
cfg = UniterConfig.from_dict(BASE)
model = MemeUniter(UniterModel(cfg, img_dim=2048), cfg.hidden_size, 1).to(dev).train()
model.uniter_model.set_dropout_seed(1234, 0)
if len(sys.argv) > 1:
    model.uniter_model.precision = sys.argv[1]
batch = make_synthetic_batch(16, 128, 36, seed=1234, device=dev)
config = dict(optimizer='adam', lr=3e-5, beta1=0.9, beta2=0.999, weight_decay=1e-3, gradient_accumulation=1, max_grad_norm=5,
              pos_wt=1.8, loss_func='bce_logits', scheduler='warmup_cosine', warmup_steps=500, max_epoch=30)
opt = FusedAdam(model, lr=config['lr'], weight_decay=config['weight_decay'])
step = TrainStep(model, opt, get_scheduler(opt, config, steps_per_epoch=1000), config)
def timed(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for _ in range(5): step.train_iter(batch, iters=0)
print('plain  %.3f ms/step' % timed(lambda: step.train_iter(batch, iters=0)), flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step.train_iter(batch, iters=0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step.train_iter(batch, iters=0)
    print('captured', flush=True)
    print('graph  %.3f ms/step' % timed(g.replay), flush=True)
