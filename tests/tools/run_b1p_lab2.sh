# lab: slope per k-tile and fixed cost of the persistent bf16 kernels (K sweep)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
  echo "--- K sweep, FFN-up shape (N = 3072), bias + GELU epilogue"
  LAB_KSWEEP=256,768,1536,3072 LAB_VARIANTS=c1,c6,c7 LAB_ONLY=ffnup_fwd timeout 300 python tests/tools/gemm_v2_lab.py
  echo "--- the same, bias epilogue only (bf16 out)"
  LAB_EPI=1 LAB_KSWEEP=256,768,1536,3072 LAB_VARIANTS=c1,c6,c7 LAB_ONLY=ffnup_fwd timeout 300 python tests/tools/gemm_v2_lab.py
  echo "--- K sweep, QKV shape (N = 2304), bias"
  LAB_KSWEEP=256,768,1536,3072 LAB_VARIANTS=c1,c7,c8 LAB_ONLY=qkv_fwd timeout 300 python tests/tools/gemm_v2_lab.py
  echo "--- K sweep, N = 768 (two k-pieces)"
  LAB_KSWEEP=768,1536,3072,6144 LAB_VARIANTS=c1s2,c6s2,c7s4 LAB_ONLY=ffndown_fwd timeout 300 python tests/tools/gemm_v2_lab.py
} > gpurun_out/r06_b1p_lab2.log 2>&1
cat gpurun_out/r06_b1p_lab2.log
