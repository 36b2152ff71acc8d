cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ timeout 600 python -m pytest tests/test_gemm_bf16v2_gpu.py -q -x -k "group" 2>&1 | tail -5
  LAB_ONLY=wgroup timeout 600 python tests/tools/gemm_v2_lab.py
  for g in 0 1 4; do echo "--- UNITER_WGRAD_GROUP=$g"; UNITER_WGRAD_GROUP=$g timeout 600 python bench.py --precision bf16 --no_cpu_baseline --steps 50 --warmup 10 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], [(f['family'], f['avg_us'], f['achieved']) for f in d['roofline_families'] if f['family'] in ('gemm_wgrad', 'gemm_dgrad')])
"; done
} > gpurun_out/wgroup.log 2>&1
cat gpurun_out/wgroup.log
