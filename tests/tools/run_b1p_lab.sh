# lab: the persistent bf16 kernels (cfg 6 / 7 / 8, csrc/gemm_bf16_p.hip) against gemm_bf16_dma.hip's (cfg 1 / 5) and the vendor library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
  LAB_VARIANTS=c1,c6,c7,c8 LAB_ONLY=qkv_fwd,ffnup_fwd,ffndown_dgrad timeout 300 python tests/tools/gemm_v2_lab.py
  LAB_VARIANTS=c1s2,c6s2,c6s3,c7s2,c7s4 LAB_ONLY=ffndown_fwd,ffnup_dgrad,qkv_dgrad timeout 300 python tests/tools/gemm_v2_lab.py
  LAB_VARIANTS=c5,c1s2,c6,c6s2 LAB_ONLY=attnout_fwd,attnout_dgrad timeout 300 python tests/tools/gemm_v2_lab.py
  LAB_ONLY=wgroup timeout 300 python tests/tools/gemm_v2_lab.py
} > gpurun_out/r06_b1p_lab.log 2>&1
cat gpurun_out/r06_b1p_lab.log
