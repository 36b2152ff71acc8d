"""bf16 mode: how closely the two attention kernel families (UNITER_ATTN_B16X=1/0, read once per process) track the bf16 oracle on
the config-1-shape training step: logits and a few gradients against the oracle's bf16 mode and against the fp32 oracle."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_parity_configs_gpu as T
from oracle import uniter_oracle as O
from meme_challenge_amd.trainer import bce_with_logits_loss
cfg, B, Tt, R, seed = T.BASE, 4, 64, 36, int(sys.argv[1]) if len(sys.argv) > 1 else 1234
sd = O.synth_state_dict(cfg, seed=0, ln_jitter=0.02)
b = O.synth_batch(B, Tt, R, seed=seed)
m = T._build(cfg, sd, 'bf16', train=True)
m.uniter_model.set_dropout_seed(0xB16B16, 3)
bd = {k: v.cuda() for k, v in b.items()}
logits = m(**T.model_kwargs(bd))
bce_with_logits_loss(logits.squeeze(1), bd['labels'], 1.8).backward()
torch.cuda.synchronize()
drop = O.DropSpec(0xB16B16, 3, cfg['hidden_dropout_prob'], cfg['attention_probs_dropout_prob'])
lb, _, gb = T._oracle_step(sd, cfg, b, drop, 'bf16')
lf, _, gf = T._oracle_step(sd, cfg, b, drop, 'fp32')
print('logits: |hip - bf16 oracle| %.3e  |hip - fp32| %.3e  |bf16 oracle - fp32| %.3e' % (T.maxdiff(logits, lb), T.maxdiff(logits, lf), T.maxdiff(lb, lf)))
worst = []
for n, p in m.named_parameters():
    if gf[n].abs().max().item() == 0.0 or n.endswith('key.bias'): continue
    eh, eo, ehb = T._rms_rel(p.grad, gf[n]), T._rms_rel(gb[n], gf[n]), T._rms_rel(p.grad, gb[n])
    worst.append((eh / max(eo, 1e-30), n, eh, eo, ehb))
worst.sort(reverse=True)
for r, n, eh, eo, ehb in worst[:6]:
    print('%-60s hip/fp32 %.4f  oracle/fp32 %.4f  ratio %.2f  hip/oracle %.4f' % (n, eh, eo, r, ehb))
import statistics
print('median ratio %.3f over %d tensors; median hip/oracle %.4f' % (statistics.median(w[0] for w in worst), len(worst), statistics.median(w[4] for w in worst)))
