#!/bin/bash
# round 3, GPU call 1: full GPU suite, then same-box A/B of the fp32 weight-gradient tile size
mkdir -p gpurun_out/r03
python -m pytest tests -m gpu -x -q > gpurun_out/r03/tests1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03/tests1.log
tail -5 gpurun_out/r03/tests1.log
python tests/tools/wgrad_bench.py 2>&1 | tee gpurun_out/r03/wgrad_bench.txt
python tests/tools/attn_bench.py 2>&1 | tee gpurun_out/r03/attn_bench.txt
for i in 1 2 3; do
  for cfg in 0 21; do
    echo -n "WGRAD_CFG=$cfg "; UNITER_WGRAD_CFG=$cfg python bench.py --no_cpu_baseline --steps 40 --warmup 10 2>/dev/null | tee -a gpurun_out/r03/ab_wgrad_$cfg.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('backward_gemms_together',{}).get('frac'), [(f['family'],f['avg_us']) for f in d['roofline_families']])"
  done
done
python bench.py --precision bf16 --no_cpu_baseline --steps 40 --warmup 10 2>/dev/null | tee gpurun_out/r03/bf16_base.json | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('bf16', d['ms_per_step'], d['value'])"
