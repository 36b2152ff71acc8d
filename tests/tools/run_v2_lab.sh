cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{ echo "--- fixed cost: qkv-like (bias, bf16 out) N=3072 via LAB_EPI"; 
  LAB_ONLY=ffnup_fwd LAB_EPI=1 LAB_VARIANTS=c1,c1+256,c1+512,c1+768 python tests/tools/gemm_v2_lab.py
  LAB_ONLY=ffnup_fwd LAB_EPI=0 LAB_VARIANTS=c1,c1+256,c1+512,c1+768 python tests/tools/gemm_v2_lab.py
  LAB_ONLY=ffnup_fwd LAB_EPI=5 LAB_VARIANTS=c1,c1+256,c1+512,c1+768 python tests/tools/gemm_v2_lab.py
  echo "--- M sweep (ffnup, c1)"; for m in 256 1312 2624 5248; do LAB_M=$m LAB_ONLY=ffnup_fwd LAB_VARIANTS=c1,c1+768,c2 python tests/tools/gemm_v2_lab.py; done
} > gpurun_out/v2_lab3.log 2>&1
cat gpurun_out/v2_lab3.log
