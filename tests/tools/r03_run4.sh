#!/bin/bash
mkdir -p gpurun_out/r03
python -m pytest tests/test_trainer_gpu.py tests/test_dp_gpu.py tests/test_model_gpu.py tests/test_cli_gpu.py tests/test_packed_gpu.py -q -x > gpurun_out/r03/tests4.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03/tests4.log
tail -8 gpurun_out/r03/tests4.log
one() { python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('optimizer',{}).get('ms'))"; }
for i in 1 2 3; do
  echo -n "fp32 new            "; one
  echo -n "fp32 no-word-split  "; UNITER_ADAM_WORD_SPLIT=0 one
  echo -n "bf16 new            "; one --precision bf16
  echo -n "bf16 no-word-split  "; UNITER_ADAM_WORD_SPLIT=0 one --precision bf16
done
echo -n "bf16 no side stream "; one --precision bf16 --no_side_stream
echo -n "fp32 no side stream "; one --no_side_stream
