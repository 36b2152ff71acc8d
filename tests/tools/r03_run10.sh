#!/bin/bash
python tests/tools/ln_beside_lab.py 2>&1 | grep ln_bwd
one() { python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], [(f['family'][5:17],f['avg_us']) for f in d['roofline_families'] if 'norm' in f['family']])"; }
for i in 1 2 3; do
  echo -n "fp32 prio3  "; one
  echo -n "bf16 prio3  "; one --precision bf16
done
