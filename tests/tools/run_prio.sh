cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "--- $*"; timeout 600 "$@" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'])
"; }
{ for rep in 1 2 3; do for f in 0 1; do export UNITER_FIN_LATE=$f; echo "== FIN_LATE=$f"
  run python bench.py --precision bf16 --no_cpu_baseline
  run python bench.py --no_cpu_baseline --steps 40
  done; done
} > gpurun_out/prio.log 2>&1
cat gpurun_out/prio.log
