#!/bin/bash
mkdir -p gpurun_out/r03
python -m pytest tests/test_attention_gpu.py tests/test_attention_bf16_gpu.py tests/test_model_gpu.py tests/test_packed_gpu.py tests/test_parity_configs_gpu.py tests/test_trainer_gpu.py -q -x > gpurun_out/r03/tests8.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03/tests8.log
grep -v Warning gpurun_out/r03/tests8.log | tail -8
one() { python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], [(f['family'][5:17],f['avg_us']) for f in d['roofline_families'] if 'attention' in f['family']])"; }
for i in 1 2 3; do
  echo -n "fp32 pregen    "; one
  echo -n "fp32 in-kernel "; UNITER_KEEP_PREGEN=0 one
  echo -n "bf16 pregen    "; one --precision bf16
  echo -n "bf16 in-kernel "; UNITER_KEEP_PREGEN=0 one --precision bf16
done
