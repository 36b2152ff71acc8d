"""Per-task step time of UniterForPretraining (BASELINE configs[4] shapes) + kernel breakdown hints."""
import sys, time, torch
sys.path.insert(0, '.')
from meme_challenge_amd.model import UniterConfig
from meme_challenge_amd.pretrain import UniterForPretraining
from meme_challenge_amd.trainer import FusedAdam
from meme_challenge_amd.utils import make_synthetic_pretrain_batch
from bench import BASE
dev = torch.device('cuda')
torch.manual_seed(0)
cfg = UniterConfig.from_dict(BASE)
model = UniterForPretraining(cfg, img_dim=2048, img_label_dim=1601).to(dev).train()
model.uniter.set_dropout_seed(1, 0)
opt = FusedAdam(model, lr=3e-5, weight_decay=1e-3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for task in ('itm', 'mlm', 'mrfr'):
    batch = make_synthetic_pretrain_batch(task, B, 128, 36, device=dev)
    def step():
        loss = model(batch, task).mean(); loss.backward(); opt.step(grad_scale=1.0, max_grad_norm=5)
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print('%-5s %.2f ms/step  %.0f samples/s' % (task, dt * 1e3, B / dt), flush=True)
