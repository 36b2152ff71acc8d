"""Where the GEMMs of consecutive training steps sit on the GPU's clock, from the in-kernel launch stamps (no profiler
attached): per step the time from the last backward GEMM's end to the next step's first forward GEMM's start (the step
boundary: embedding backward, clip norm, optimizer head, next step's embeddings), and the step's length.
Usage: python tests/tools/step_boundary.py [bf16|fp32]"""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
from meme_challenge_amd import _lib
from meme_challenge_amd.model import UniterConfig, UniterModel
from meme_challenge_amd.meme_uniter import MemeUniter
from meme_challenge_amd.trainer import FusedAdam, TrainStep, get_scheduler
from meme_challenge_amd.utils import make_synthetic_batch
import bench
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
torch.manual_seed(0)
cfg = UniterConfig.from_dict(bench.BASE)
dev = torch.device('cuda:0')
model = MemeUniter(UniterModel(cfg, img_dim=2048), cfg.hidden_size, 1).to(dev).train()
enc = model.uniter_model; enc.precision = prec; enc.set_dropout_seed(1, 0)
batch = make_synthetic_batch(16, 128, 36, seed=1, device=dev)
config = dict(optimizer='adam', lr=3e-5, beta1=0.9, beta2=0.999, weight_decay=1e-3, gradient_accumulation=1, max_grad_norm=5,
              pos_wt=1.8, loss_func='bce_logits', scheduler='warmup_cosine', warmup_steps=500, max_epoch=30)
opt = FusedAdam(model, lr=3e-5, weight_decay=1e-3); opt.overlap_encoder = enc
step = TrainStep(model, opt, get_scheduler(opt, config, steps_per_epoch=1000), config)
for _ in range(30): step.train_iter(batch, iters=0)
torch.cuda.synchronize()
lib = _lib.lib()
_lib.check(lib.uniter_prof_enable_stamps(enc._handle, 1, None))
NS = 8
for _ in range(NS): step.train_iter(batch, iters=0)
cap = 4096
kinds = (C.c_int * cap)(); t0 = (C.c_double * cap)(); t1 = (C.c_double * cap)(); n = C.c_int(0)
_lib.check(lib.uniter_prof_stamp_spans(enc._handle, kinds, t0, t1, cap, C.byref(n)))
_lib.check(lib.uniter_prof_enable_stamps(enc._handle, 0, None))
spans = [(kinds[i], t0[i], t1[i]) for i in range(n.value)]
FWD = (1, 2, 3, 4)
# a step's first forward GEMM: a forward kind behind a backward kind
firsts = [i for i in range(1, len(spans)) if spans[i][0] in FWD and spans[i - 1][0] not in FWD]
firsts = [0] + firsts
for a, b in zip(firsts, firsts[1:]):
    last_bwd_end = max(e for k, s, e in spans[a:b] if k not in FWD)
    print('%s step %.1f us (first forward GEMM to the next step\'s); boundary (last backward GEMM end -> next forward GEMM start) %.1f us'
          % (prec, spans[b][1] - spans[a][1], spans[b][1] - last_bwd_end))
