cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "--- $*"; timeout 600 "$@" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'])
"; }
{ rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
  run python bench.py --steps 20 --no_cpu_baseline
  run python bench.py --no_cpu_baseline
  run python bench.py
  run python bench.py --no_cpu_baseline
  rocm-smi --showpower --showtemp 2>/dev/null | grep -i "power\|temp" | head -6
} > gpurun_out/prio.log 2>&1
cat gpurun_out/prio.log
