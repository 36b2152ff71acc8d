# Round-2 profile set: bench lines, rocprofv3 kernel stats (fp32 + bf16), PMC passes (separate runs; never with a
# trace domain besides --kernel-trace), traffic of the FFN-up forward GEMM and of the time-dominant stream-K weight
# gradient kernel.  Usage on the GPU box: bash tests/tools/run_profile_r02.sh
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02
mkdir -p $O
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 300 python bench.py --precision bf16 --no_cpu_baseline > $O/bench_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --model large --batch 8 --num_bb 50 --no_cpu_baseline > $O/bench_large.json 2>> $O/bench.err
timeout 300 python bench.py --model large --batch 8 --num_bb 50 --precision bf16 --no_cpu_baseline > $O/bench_large_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --workload multitask --batch 32 --no_cpu_baseline > $O/bench_multitask.json 2>> $O/bench.err
timeout 300 python bench.py --workload multitask --batch 32 --precision bf16 --no_cpu_baseline > $O/bench_bf16_multitask.json 2>> $O/bench.err
timeout 300 python bench.py --ragged --packed --no_cpu_baseline > $O/bench_ragged_packed.json 2>> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --no_cpu_baseline --steps 25 --warmup 5 > $O/bench_under_rocprof.json 2>$O/rocprof.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -o st -- python3 bench.py --precision bf16 --no_cpu_baseline --steps 25 --warmup 5 > $O/bench_bf16_under_rocprof.json 2>>$O/rocprof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
find $O/stats_bf16 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_bf16_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 bench.py --no_cpu_baseline --steps 3 --warmup 1 --prof_kind 0 > /dev/null 2>$O/pmc1.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 bench.py --no_cpu_baseline --steps 3 --warmup 1 --prof_kind 0 > /dev/null 2>$O/pmc2.err
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_mfma -o m -- python3 bench.py --no_cpu_baseline --steps 3 --warmup 1 --prof_kind 0 > /dev/null 2>$O/pmc3.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_bf16 -o f -- python3 bench.py --precision bf16 --no_cpu_baseline --steps 3 --warmup 1 --prof_kind 0 > /dev/null 2>$O/pmc4.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_bf16 -o w -- python3 bench.py --precision bf16 --no_cpu_baseline --steps 3 --warmup 1 --prof_kind 0 > /dev/null 2>$O/pmc5.err
for d in fetch write mfma fetch_bf16 write_bf16; do python tests/tools/pmc_summary.py $O/pmc_$d $O/pmc_$d.csv; head -6 $O/pmc_$d.csv | cut -c1-200; done
python tests/tools/pmc_to_traffic.py $O $O/pmc_traffic.json
python tests/tools/pmc_table.py $O/ > $O/kernel_table.md; python tests/tools/pmc_table.py $O/ bf16 > $O/kernel_table_bf16.md
# keep the merge-back small
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
du -sh $O
