#!/bin/bash
cd $GRAFT_REPO_ROOT
pr() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for q in default 8 16 default 8; do
  if [ "$q" != default ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  echo -n "plain queues=$q "; python3 bench.py --no_cpu_baseline --steps 30 --warmup 5 "$@" 2>/dev/null | pr
done
