#!/bin/bash
one() { python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('backward_gemms_together',{}).get('frac'), [(f['family'][5:17],f['avg_us']) for f in d['roofline_families'][5:]])"; }
for i in 1 2; do
for mask in 0 3 15 12 1 2; do echo -n "WHOLE=$mask "; UNITER_WGRAD_WHOLE=$mask one; done
done
