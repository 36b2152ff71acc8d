#!/bin/bash
# single-rank forced DP path (UNITER_DP_FORCE=1) under different GPU_MAX_HW_QUEUES, next to the plain step
cd $GRAFT_REPO_ROOT
pr() { grep '^{' | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for q in default 2 8 16; do
  [ "$q" != default ] && export GPU_MAX_HW_QUEUES=$q
  echo -n "dp-forced queues=$q "; UNITER_DP_FORCE=1 python3 bench.py --gpus 1 --no_cpu_baseline --steps 30 --warmup 5 "$@" 2>/dev/null | pr
done
unset GPU_MAX_HW_QUEUES
echo -n "plain "; python3 bench.py --no_cpu_baseline --steps 30 --warmup 5 "$@" 2>/dev/null | pr
