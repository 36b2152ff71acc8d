#!/bin/bash
one() { python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d.get('backward_gemms_together',{}).get('frac'), [(f['family'][5:17],f['launches_per_step'],f['avg_us']) for f in d['roofline_families'][5:]])"; }
for i in 1 2; do
  for sl in 1024 768 512 1728; do echo -n "grouped slots=$sl "; UNITER_WGRAD_GROUP_F32_SLOTS=$sl one; done
  echo -n "4 launches "; UNITER_WGRAD_GROUP_F32=0 one
done
