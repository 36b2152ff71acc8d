cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/gemm_pmc
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/gemm_pmc/p1 -o a -- python3 tests/tools/gemm_shapes.py 0 > /dev/null 2> gpurun_out/gemm_pmc/p1.err
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/gemm_pmc/p2 -o b -- python3 tests/tools/gemm_shapes.py 0 > /dev/null 2> gpurun_out/gemm_pmc/p2.err
for d in p1 p2; do python tests/tools/pmc_summary.py gpurun_out/gemm_pmc/$d gpurun_out/gemm_pmc/$d.csv; done
find gpurun_out/gemm_pmc -name "*counter_collection.csv" -delete; find gpurun_out/gemm_pmc -name "*kernel_trace.csv" -delete
tail -3 gpurun_out/gemm_pmc/p1.err
