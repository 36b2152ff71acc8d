"""Diagnostic: per-layer hidden states, HIP bf16 vs oracle bf16-rounding vs oracle fp32 (eval mode)."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import uniter_oracle as O
from common import BASE, model_kwargs
from test_parity_configs_gpu import _build
cfg = dict(BASE, num_hidden_layers=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
sd = O.synth_state_dict(cfg, seed=0, ln_jitter=0.02)
b = O.synth_batch(4, 64, 36, seed=1234)
bd = {k: v.cuda() for k, v in b.items()}
kw = model_kwargs(bd); kw['output_all_encoded_layers'] = True
m = _build(cfg, sd, 'bf16')
with torch.no_grad():
    hb = [h.cpu() for h in m.uniter_model(**kw)]
m.uniter_model.precision = 'fp32'
with torch.no_grad():
    hf = [h.cpu() for h in m.uniter_model(**kw)]
kwc = model_kwargs(b); kwc['output_all_encoded_layers'] = True
with torch.no_grad():
    ob = O.uniter_forward(sd, cfg, prefix='uniter_model.', prec='bf16', **kwc)
    of = O.uniter_forward(sd, cfg, prefix='uniter_model.', prec='fp32', **kwc)
def rms(a, c): return ((a - c).double().pow(2).mean().sqrt() / c.double().pow(2).mean().sqrt()).item()
for l in range(cfg['num_hidden_layers']):
    print('layer %d  rms rel: hip_b16-ora_b16 %.2e  hip_b16-ora_f32 %.2e  ora_b16-ora_f32 %.2e  hip_f32-ora_f32 %.2e' % (
        l, rms(hb[l], ob[l]), rms(hb[l], of[l]), rms(ob[l], of[l]), rms(hf[l], of[l])))
