#!/bin/bash
set -e
mkdir -p gpurun_out/r04_p
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "dropout" > gpurun_out/r04_p/t_kernels.log 2>&1 || { tail -40 gpurun_out/r04_p/t_kernels.log; exit 1; }
tail -3 gpurun_out/r04_p/t_kernels.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_models.py -x -q -m gpu -k "graphed or dropout" > gpurun_out/r04_p/t_models.log 2>&1 || { tail -60 gpurun_out/r04_p/t_models.log; exit 1; }
tail -3 gpurun_out/r04_p/t_models.log
