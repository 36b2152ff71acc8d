#!/bin/bash
# refresh the bf16-mode bench lines and the kernel stats kept under profiles/ (run on the GPU box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
o=gpurun_out/r01c; mkdir -p $o
B="python3 bench.py --no_cpu_baseline --steps 50 --warmup 10"
$B --precision bf16 | tail -1 > $o/r01_bench_bf16.json
$B --precision bf16_hybrid | tail -1 > $o/r01_bench_bf16_hybrid.json
$B --precision bf16 --ragged --packed | tail -1 > $o/r01_bench_bf16_ragged_packed.json
$B --precision bf16 --workload multitask --batch 32 | tail -1 > $o/r01_bench_bf16_multitask.json
$B --precision bf16 --model large --batch 8 --num_bb 50 | tail -1 > $o/r01_bench_large_bf16.json
$B | tail -1 > $o/r01_bench_fp32_check.json
python3 tests/tools/gemm_lab.py > $o/r01_gemm_bf16_resident_tiles.txt 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/stats -o st -- python3 bench.py --no_cpu_baseline --precision bf16 > $o/bench_bf16_under_rocprof.json 2>$o/err.txt
cp $(find $o/stats -name "*kernel_stats.csv" | head -1) $o/r01_bench_bf16_kernel_stats.csv
find $o -name "*kernel_trace.csv" -delete; find $o -name "*.db" -delete; rm -rf $o/stats
for f in $o/*.json; do echo $f; python3 -c "import sys,json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done
