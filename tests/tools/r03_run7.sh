#!/bin/bash
mkdir -p gpurun_out/r03
python -m pytest tests/test_trainer_gpu.py tests/test_dp_gpu.py tests/test_model_gpu.py tests/test_cli_gpu.py tests/test_packed_gpu.py tests/test_gemm_gpu.py tests/test_gemm_bf16v2_gpu.py tests/test_pretrain_gpu.py -q -x > gpurun_out/r03/tests7.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03/tests7.log
grep -v Warning gpurun_out/r03/tests7.log | tail -12
one() { python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('optimizer',{}).get('ms'), d.get('optimizer',{}).get('bytes_per_parameter'))"; }
for i in 1 2 3; do
  echo -n "fp32 lazy      "; one
  echo -n "fp32 no lazy   "; UNITER_LAZY_ZERO=0 one
  echo -n "fp32 streamK   "; UNITER_LAZY_ZERO=0 UNITER_WGRAD_WHOLE=0 one
  echo -n "bf16 lazy      "; one --precision bf16
  echo -n "bf16 no lazy   "; UNITER_LAZY_ZERO=0 one --precision bf16
done
