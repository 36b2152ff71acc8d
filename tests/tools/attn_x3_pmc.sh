# PMC passes over the isolated x3 attention kernels (separate runs, --kernel-trace only beside --pmc)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/attn_x3_pmc; mkdir -p $O
i=0
for ctr in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  LAB_FORMS=0 timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/p$i -o p -- python3 tests/tools/attn_x3_lab.py > /dev/null 2> $O/p$i.err
  python tests/tools/pmc_summary.py $O/p$i $O/p$i.csv > /dev/null 2>&1
  grep -i "attn_x3" $O/p$i.csv | head -4
done
head -1 $O/p1.csv
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info.csv" -delete
