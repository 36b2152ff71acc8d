# copy the summaries of gpurun_out/r04p (tests/tools/run_profile_r04.sh 1 and 2) into profiles/ under the round's prefix
O=gpurun_out/r04p; P=profiles
for f in attention_isolated.txt attn_x3_lab.txt attn_x3_pmc.txt b16x_vs_oracle.txt bench.json bench_attention_on_fp32_mfma.json bench_bf16.json bench_bf16_kernel_stats.csv bench_bf16_multitask.json bench_bf16_ragged_packed.json bench_bf16_under_rocprof.json bench_gpus2_gloo_one_gpu.json bench_kernel_stats.csv bench_large.json bench_large_bf16.json bench_multitask.json bench_native_fp32.json bench_ragged_packed.json bench_rccl_one_rank_forced.json bench_rccl_one_rank_forced_sparse.json bench_under_rocprof.json build_info.txt gemm_x3_ablation.txt gemm_x3_clock.txt gemm_x3_ksweep.txt gemm_x3_lab.txt kernel_table.md kernel_table_bf16.md ln_isolated.txt pmc_fetch.csv pmc_fetch_bf16.csv pmc_mfma.csv pmc_traffic.json pmc_traffic.txt pmc_write.csv pmc_write_bf16.csv timeline_bf16.txt timeline_f32x3.txt step_boundary.txt; do
  [ -f $O/$f ] && cp $O/$f $P/r04_$f || echo "missing $f"
done
grep 'samples/s' $O/cli_throughput_raw.txt > $P/r04_cli_throughput.txt
cp gpurun_out/full_gpu_r04.txt $P/r04_gpu_tests.txt
python tests/tools/results_table.py profiles r04_
