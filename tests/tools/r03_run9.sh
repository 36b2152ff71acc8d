#!/bin/bash
python -m pytest tests/test_layernorm_gpu.py tests/test_model_gpu.py tests/test_pretrain_gpu.py tests/test_trainer_gpu.py -q -x 2>&1 | tail -3
python tests/tools/ln_bench.py; UNITER_LNB_LEAN=0 python tests/tools/ln_bench.py
python tests/tools/ln_beside_lab.py 2>&1 | grep ln_bwd
echo "--- register-resident form"; UNITER_LNB_LEAN=0 python tests/tools/ln_beside_lab.py 2>&1 | grep ln_bwd
one() { python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], [(f['family'][5:17],f['avg_us']) for f in d['roofline_families'] if 'norm' in f['family']])"; }
for i in 1 2 3; do
  echo -n "fp32 lean  "; one
  echo -n "fp32 regs  "; UNITER_LNB_LEAN=0 one
  echo -n "bf16 lean  "; one --precision bf16
  echo -n "bf16 regs  "; UNITER_LNB_LEAN=0 one --precision bf16
done
