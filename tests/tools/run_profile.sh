set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r01
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py > gpurun_out/r01/bench.json 2> gpurun_out/r01/bench.err; tail -1 gpurun_out/r01/bench.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01/stats -o st -- python3 bench.py --no_cpu_baseline > gpurun_out/r01/bench_rocprof.json 2>gpurun_out/r01/rocprof.err
find gpurun_out/r01/stats -name "*kernel_stats.csv" | head
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r01/pmc_fetch -o f -- python3 bench.py --no_cpu_baseline --steps 3 --warmup 1 > /dev/null 2>gpurun_out/r01/pmc1.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r01/pmc_write -o w -- python3 bench.py --no_cpu_baseline --steps 3 --warmup 1 > /dev/null 2>gpurun_out/r01/pmc2.err
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/r01/pmc_mfma -o m -- python3 bench.py --no_cpu_baseline --steps 3 --warmup 1 > /dev/null 2>gpurun_out/r01/pmc3.err
for d in fetch write mfma; do python tests/tools/pmc_summary.py gpurun_out/r01/pmc_$d gpurun_out/r01/pmc_$d.csv; head -8 gpurun_out/r01/pmc_$d.csv; done
# keep the merge-back small
find gpurun_out/r01 -name "*counter_collection.csv" -delete; find gpurun_out/r01 -name "*kernel_trace.csv" -delete; find gpurun_out/r01 -name "*.db" -delete
du -sh gpurun_out/r01
