#!/bin/bash
mkdir -p gpurun_out/r03
python -m pytest tests/test_parity_configs_gpu.py -q -k "config5" > gpurun_out/r03/tests3.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03/tests3.log
tail -30 gpurun_out/r03/tests3.log
python -m pytest tests/test_pretrain_gpu.py tests/test_gemm_bf16v2_gpu.py -q 2>&1 | tail -3
one() { python bench.py --precision bf16 --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], [(f['family'][:12],f['avg_us']) for f in d['roofline_families']])"; }
for i in 1 2; do
  for wgs in 0 256 192 128; do echo -n "GROUP_WGS=$wgs "; UNITER_WGRAD_GROUP_WGS=$wgs one; done
  echo -n "GROUP=4 "; UNITER_WGRAD_GROUP=4 one
  echo -n "GROUP=4 WGS=256 "; UNITER_WGRAD_GROUP=4 UNITER_WGRAD_GROUP_WGS=256 one
done
for wgs in 0 256; do echo -n "multitask GROUP_WGS=$wgs "; UNITER_WGRAD_GROUP_WGS=$wgs one --workload multitask --batch 32; done
for wgs in 0 256; do echo -n "large GROUP_WGS=$wgs "; UNITER_WGRAD_GROUP_WGS=$wgs one --model large --batch 8 --num_bb 50; done
