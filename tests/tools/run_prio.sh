cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
run() { echo "--- $*"; timeout 600 "$@" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'])
"; }
{ for g in 0 1; do
    UNITER_WGRAD_GROUP=$g run python bench.py --precision bf16 --model large --batch 8 --num_bb 50 --no_cpu_baseline
    UNITER_WGRAD_GROUP=$g run python bench.py --precision bf16 --workload multitask --batch 32 --no_cpu_baseline
    UNITER_WGRAD_GROUP=$g run python bench.py --precision bf16 --ragged --packed --no_cpu_baseline
  done
  timeout 1200 python -m pytest tests/test_trainer_gpu.py tests/test_gemm_bf16v2_gpu.py tests/test_parity_configs_gpu.py tests/test_model_gpu.py tests/test_packed_gpu.py tests/test_dp_gpu.py -q -x 2>&1 | tail -5
} > gpurun_out/prio.log 2>&1
cat gpurun_out/prio.log
