#!/bin/bash
set -e
mkdir -p gpurun_out/r04_ac
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "simce or ce_combine" > gpurun_out/r04_ac/t_kernels.log 2>&1 || { tail -40 gpurun_out/r04_ac/t_kernels.log; exit 1; }
tail -2 gpurun_out/r04_ac/t_kernels.log
