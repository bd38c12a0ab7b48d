#!/bin/bash
set -e
mkdir -p gpurun_out/r04_lib
timeout -k 10 500 python3 tools/exp_gemm_vs_library.py > gpurun_out/r04_lib/gemm_vs_library.txt 2>&1 || { tail -20 gpurun_out/r04_lib/gemm_vs_library.txt; exit 1; }
cat gpurun_out/r04_lib/gemm_vs_library.txt
