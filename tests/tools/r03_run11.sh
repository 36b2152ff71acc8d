#!/bin/bash
one() { python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], [(f['family'][5:17],f['avg_us']) for f in d['roofline_families'][4:]])"; }
for i in 1 2 3; do
  echo -n "fp32 main2 attn2 "; one
  echo -n "fp32 main0 attn0 "; UNITER_MAIN_PRIO=0 UNITER_ATTN_PRIO=0 one
  echo -n "fp32 main0 attn2 "; UNITER_MAIN_PRIO=0 one
  echo -n "fp32 main2 attn0 "; UNITER_ATTN_PRIO=0 one
  echo -n "fp32 main1 attn3 "; UNITER_MAIN_PRIO=1 UNITER_ATTN_PRIO=3 one
done
for i in 1 2 3; do
  echo -n "bf16 main2 attn2 "; one --precision bf16
  echo -n "bf16 main0 attn0 "; UNITER_MAIN_PRIO=0 UNITER_ATTN_PRIO=0 one --precision bf16
  echo -n "bf16 main0 attn2 "; UNITER_MAIN_PRIO=0 one --precision bf16
done
