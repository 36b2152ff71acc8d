#!/bin/bash
mkdir -p gpurun_out/r03
./tests/tools/mfma_valu_coissue.bin 2>&1 | tee gpurun_out/r03/mfma_valu_coissue.txt
python -m pytest tests -m gpu -q > gpurun_out/r03/tests2.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r03/tests2.log
tail -15 gpurun_out/r03/tests2.log
