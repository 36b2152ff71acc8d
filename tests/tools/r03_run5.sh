#!/bin/bash
one() { python bench.py --no_cpu_baseline --steps 40 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d.get('backward_gemms_together',{}).get('frac'), [(f['family'][5:17],f['avg_us']) for f in d['roofline_families']])"; }
for i in 1 2; do
echo -n "default          "; one
echo -n "SK=0 (whole tiles)"; UNITER_GEMM_SK=0 one
echo -n "no side stream   "; one --no_side_stream
echo -n "slots 512        "; UNITER_WGRAD_SLOTS_F32=512 one
done
python -m pytest tests/test_packed_gpu.py tests/test_trainer_gpu.py -q 2>&1 | tail -2
