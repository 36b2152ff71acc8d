# Round-3 profile set.  Part 1 (bench lines + kernel stats + timelines + lab tools), part 2 (PMC passes: separate runs, never
# with a trace domain besides --kernel-trace).  Usage on the GPU box: bash tests/tools/run_profile_r03.sh [1|2]
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p
mkdir -p $O
part=${1:-1}
if [ "$part" = "1" ]; then
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
timeout 300 python bench.py --precision bf16 --no_cpu_baseline > $O/bench_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --model large --batch 8 --num_bb 50 --no_cpu_baseline > $O/bench_large.json 2>> $O/bench.err
timeout 300 python bench.py --model large --batch 8 --num_bb 50 --precision bf16 --no_cpu_baseline > $O/bench_large_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --workload multitask --batch 32 --no_cpu_baseline > $O/bench_multitask.json 2>> $O/bench.err
timeout 300 python bench.py --workload multitask --batch 32 --precision bf16 --no_cpu_baseline > $O/bench_bf16_multitask.json 2>> $O/bench.err
timeout 300 python bench.py --ragged --packed --no_cpu_baseline > $O/bench_ragged_packed.json 2>> $O/bench.err
timeout 300 python bench.py --ragged --packed --precision bf16 --no_cpu_baseline > $O/bench_bf16_ragged_packed.json 2>> $O/bench.err
UNITER_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --prewarm_s 0 --prof_kind 0 --no_cpu_baseline > $O/bench_gpus2_gloo_one_gpu.json 2>> $O/bench.err
UNITER_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --no_cpu_baseline > $O/bench_rccl_one_rank_forced.json 2>> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --no_cpu_baseline --steps 25 --warmup 5 > $O/bench_under_rocprof.json 2>$O/rocprof.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -o st -- python3 bench.py --precision bf16 --no_cpu_baseline --steps 25 --warmup 5 > $O/bench_bf16_under_rocprof.json 2>>$O/rocprof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
find $O/stats_bf16 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_bf16_kernel_stats.csv
bash tests/tools/run_timeline.sh f32; bash tests/tools/run_timeline.sh bf16
cp gpurun_out/tl/timeline_f32.txt $O/timeline_f32.txt; cp gpurun_out/tl/timeline_bf16.txt $O/timeline_bf16.txt
python tests/tools/attn_bench.py > $O/attention_isolated.txt 2>&1
python tests/tools/wgrad_bench.py > $O/wgrad_tiles_isolated.txt 2>&1
python tests/tools/gemm_exactfit.py > $O/gemm_f32_exactfit.txt 2>&1
./tests/tools/mfma_f32_pricelist.bin > $O/mfma_f32_pricelist.txt 2>&1
./tests/tools/mfma_valu_coissue.bin > $O/mfma_valu_coissue.txt 2>&1
./tests/tools/mfma_peak.bin > $O/mfma_peak_clock.txt 2>&1
python tests/tools/ln_bench.py > $O/ln_isolated.txt 2>&1
(python tests/tools/step_boundary.py bf16; python tests/tools/step_boundary.py fp32) 2>&1 | grep -v amdgpu.ids > $O/step_boundary.txt
(for n in 0 7; do python tests/tools/stream_queue_probe.py $n --step; done) 2>&1 | grep -v amdgpu.ids > $O/stream_queue_probe.txt
(python tests/tools/cli_throughput.py fp32; python tests/tools/cli_throughput.py bf16; python tests/tools/cli_throughput.py bf16 --feature_shards) 2>&1 | grep 'samples/s' > $O/cli_throughput_raw.txt
fi
if [ "$part" = "2" ]; then
pm() { name=$1; shift; ctr=$1; shift; timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$name -o p -- python3 bench.py --no_cpu_baseline --steps 3 --warmup 1 --prof_kind 0 "$@" > /dev/null 2>$O/pmc_$name.err; python tests/tools/pmc_summary.py $O/pmc_$name $O/pmc_$name.csv; }
pm fetch FETCH_SIZE
pm write WRITE_SIZE
pm mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"
pm fetch_bf16 FETCH_SIZE --precision bf16
pm write_bf16 WRITE_SIZE --precision bf16
# weight-gradient tile forms: stream-K 64 x 64, stream-K 128 x 128, whole-K 64 x 64 (default)
UNITER_WGRAD_WHOLE=0 UNITER_LAZY_ZERO=0 pm fetch_sk64 FETCH_SIZE
UNITER_WGRAD_WHOLE=0 UNITER_LAZY_ZERO=0 UNITER_WGRAD_CFG=21 pm fetch_sk128 FETCH_SIZE
# LayerNorm backward beside the weight gradients: waits and L2 hits, whole-K vs stream-K neighbours
pm ln_waits "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY TCC_HIT_sum TCC_MISS_sum"
UNITER_WGRAD_WHOLE=0 UNITER_LAZY_ZERO=0 pm ln_waits_sk "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY TCC_HIT_sum TCC_MISS_sum"
pm ln_waits_noside "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY TCC_HIT_sum TCC_MISS_sum" --no_side_stream
python tests/tools/pmc_to_traffic.py $O $O/pmc_traffic.json
python tests/tools/pmc_table.py $O/ > $O/kernel_table.md; python tests/tools/pmc_table.py $O/ bf16 > $O/kernel_table_bf16.md
fi
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
