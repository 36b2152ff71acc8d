cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "--- $*"; timeout 600 "$@" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], [(f['family'], f['avg_us']) for f in d['roofline_families'] if 'attention' in f['family']])
"; }
{ UNITER_LIB_VARIANT=stamps UNITER_DEV_PARTIAL_LIB=1 python tests/tools/attn_phase_lab.py
  timeout 1200 python -m pytest tests/test_attention_gpu.py tests/test_attention_bf16_gpu.py tests/test_packed_gpu.py tests/test_parity_configs_gpu.py -q -x 2>&1 | tail -3
  run python bench.py --precision bf16 --no_cpu_baseline
  run python bench.py --no_cpu_baseline
} > gpurun_out/prio.log 2>&1
cat gpurun_out/prio.log
