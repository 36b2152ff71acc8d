cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ timeout 900 python -m pytest tests/test_bench_gpu.py tests/test_pretrain_gpu.py -q -x 2>&1 | tail -3
  timeout 300 python bench.py --steps 20 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], json.dumps(d.get('traffic_from_profile'))[:600])
"
} > gpurun_out/prio.log 2>&1
cat gpurun_out/prio.log
