#!/bin/bash
# same-box A/B: current library vs libuniter_hip_<variant>.so, alternating, N rounds.  usage: ab_bench.sh variant rounds [bench args]
v=$1; n=$2; shift 2
for i in $(seq $n); do
  echo -n "cur  "; python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"
  echo -n "$v "; UNITER_LIB_VARIANT=$v UNITER_DEV_PARTIAL_LIB=1 python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"
done
