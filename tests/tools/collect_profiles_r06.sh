#!/bin/bash
# copy the summaries of gpurun_out/r06p (tests/tools/run_profile_r06.sh 1 and 2) into profiles/ under the round's prefix
O=gpurun_out/r06p; P=profiles
for f in attention_isolated.txt bench.json bench_bf16.json bench_bf16_kernel_stats.csv bench_bf16_multitask.json bench_bf16_ragged_packed.json bench_bf16_under_rocprof.json bench_gpus2_gloo_one_gpu.json bench_kernel_stats.csv bench_large.json bench_large_bf16.json bench_multitask.json bench_native_fp32.json bench_ragged_packed.json bench_under_rocprof.json build_info.txt gemm_x3_lab.txt ln_isolated.txt rccl_kernel_footprint.txt timeline_bf16.txt timeline_f32x3.txt step_boundary.txt cli_throughput.txt pmc_fetch.csv pmc_fetch_bf16.csv pmc_mfma.csv pmc_mfma_bf16.csv pmc_traffic.json pmc_traffic.txt pmc_write.csv pmc_write_bf16.csv pmc_pass_kernel_stats.csv pmc_pass_bf16_kernel_stats.csv attn_x3_pmc.txt kernel_table.md kernel_table_bf16.md; do
  [ -f $O/$f ] && cp $O/$f $P/r06_$f || echo "missing $f"
done
# (the forced one-rank RCCL lines: RCCL prints its banner behind the JSON line when the process exits -- keep the line)
for f in bench_rccl_one_rank_forced bench_rccl_one_rank_forced_reserve16; do [ -f $O/$f.json ] && head -1 $O/$f.json > $P/r06_$f.json; done
[ -f $O/bench_final.json ] && cp $O/bench_final.json $P/r06_bench_driver_style.json
[ -f gpurun_out/full_gpu_r06.txt ] && tail -1 gpurun_out/full_gpu_r06.txt > $P/r06_gpu_tests.txt
