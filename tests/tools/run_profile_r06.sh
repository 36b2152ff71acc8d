# Round-6 profile set.  Part 1 (bench lines + kernel stats + timelines + lab tools), part 2 (PMC passes: separate runs, never
# with a trace domain besides --kernel-trace; the kernel stats of part 2's own timing pass feed the tables, so each part stands
# alone).  Usage on the GPU box: bash tests/tools/run_profile_r06.sh [1|2]
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p
mkdir -p $O
part=${1:-1}
python -c "from meme_challenge_amd import _lib; print(_lib.lib().uniter_build_info().decode())" > $O/build_info.txt 2>/dev/null
if [ "$part" = "1" ]; then
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
timeout 300 python bench.py --precision fp32 --no_cpu_baseline > $O/bench_native_fp32.json 2>> $O/bench.err
timeout 300 python bench.py --precision bf16 --no_cpu_baseline > $O/bench_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --model large --batch 8 --num_bb 50 --no_cpu_baseline --no_bf16_leg > $O/bench_large.json 2>> $O/bench.err
timeout 300 python bench.py --model large --batch 8 --num_bb 50 --precision bf16 --no_cpu_baseline > $O/bench_large_bf16.json 2>> $O/bench.err
timeout 300 python bench.py --workload multitask --batch 32 --no_cpu_baseline > $O/bench_multitask.json 2>> $O/bench.err
timeout 300 python bench.py --workload multitask --batch 32 --precision bf16 --no_cpu_baseline > $O/bench_bf16_multitask.json 2>> $O/bench.err
timeout 300 python bench.py --ragged --packed --no_cpu_baseline --no_bf16_leg > $O/bench_ragged_packed.json 2>> $O/bench.err
timeout 300 python bench.py --ragged --packed --precision bf16 --no_cpu_baseline > $O/bench_bf16_ragged_packed.json 2>> $O/bench.err
UNITER_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --prewarm_s 0 --prof_kind 0 --no_cpu_baseline > $O/bench_gpus2_gloo_one_gpu.json 2>> $O/bench.err
UNITER_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --no_cpu_baseline --no_bf16_leg > $O/bench_rccl_one_rank_forced.json 2>> $O/bench.err
UNITER_DP_FORCE=1 UNITER_DP_CU_RESERVE=16 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 timeout 300 python bench.py --no_cpu_baseline --no_bf16_leg > $O/bench_rccl_one_rank_forced_reserve16.json 2>> $O/bench.err
# RCCL's own kernels beside the step: their LDS / register footprint from a kernel trace of the forced one-rank exchange
UNITER_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29535 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/rccl_trace -o t -- python3 bench.py --no_cpu_baseline --no_native_leg --no_bf16_leg --steps 4 --warmup 2 --prof_kind 0 > /dev/null 2>>$O/rocprof.err
python tests/tools/kernel_footprints.py $O/rccl_trace > $O/rccl_kernel_footprint.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python3 bench.py --no_cpu_baseline --no_native_leg --no_bf16_leg --steps 25 --warmup 5 > $O/bench_under_rocprof.json 2>>$O/rocprof.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bf16 -o st -- python3 bench.py --precision bf16 --no_cpu_baseline --steps 25 --warmup 5 > $O/bench_bf16_under_rocprof.json 2>>$O/rocprof.err
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
find $O/stats_bf16 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_bf16_kernel_stats.csv
bash tests/tools/run_timeline.sh f32 --precision fp32x3 --no_bf16_leg; bash tests/tools/run_timeline.sh bf16
cp gpurun_out/tl/timeline_f32.txt $O/timeline_f32x3.txt; cp gpurun_out/tl/timeline_bf16.txt $O/timeline_bf16.txt
python tests/tools/attn_bench.py 2>&1 | grep -v amdgpu.ids > $O/attention_isolated.txt
LAB_CFGS=3,4 LAB_NSPLIT=1,2,4 LAB_WG_CFGS=3,4 timeout 500 python tests/tools/gemm_x3_lab.py 2>&1 | grep -v amdgpu.ids > $O/gemm_x3_lab.txt
python tests/tools/ln_bench.py > $O/ln_isolated.txt 2>&1
python tests/tools/step_boundary.py fp32x3 > $O/step_boundary.txt 2>&1
(python tests/tools/cli_throughput.py fp32x3; python tests/tools/cli_throughput.py fp32; python tests/tools/cli_throughput.py bf16) 2>&1 | grep 'samples/s' > $O/cli_throughput.txt
fi
if [ "$part" = "2" ]; then
pm() { name=$1; shift; ctr=$1; shift; timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$name -o p -- python3 bench.py --no_cpu_baseline --no_native_leg --no_bf16_leg --steps 3 --warmup 1 --prof_kind 0 "$@" > /dev/null 2>$O/pmc_$name.err; python tests/tools/pmc_summary.py $O/pmc_$name $O/pmc_$name.csv; }
# (this part's own timing pass: the tables need durations from the same box and build)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -o st -- python3 bench.py --no_cpu_baseline --no_native_leg --no_bf16_leg --steps 25 --warmup 5 > /dev/null 2>$O/rocprof2.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2_bf16 -o st -- python3 bench.py --precision bf16 --no_cpu_baseline --steps 25 --warmup 5 > /dev/null 2>>$O/rocprof2.err
find $O/stats2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/pmc_pass_kernel_stats.csv
find $O/stats2_bf16 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/pmc_pass_bf16_kernel_stats.csv
pm fetch FETCH_SIZE
pm write WRITE_SIZE
pm mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES"
pm fetch_bf16 FETCH_SIZE --precision bf16
pm write_bf16 WRITE_SIZE --precision bf16
pm mfma_bf16 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" --precision bf16
bash tests/tools/attn_x3_pmc.sh > $O/attn_x3_pmc.txt 2>&1
python tests/tools/pmc_to_traffic_r06.py $O $O/pmc_traffic.json > $O/pmc_traffic.txt
python tests/tools/pmc_table.py $O/ "" pmc_pass > $O/kernel_table.md; python tests/tools/pmc_table.py $O/ bf16 pmc_pass > $O/kernel_table_bf16.md
fi
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
