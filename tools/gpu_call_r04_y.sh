#!/bin/bash
set -e
mkdir -p gpurun_out/r04_y
timeout -k 10 600 python3 -m pytest tests/test_gpu_models.py -x -q -m gpu -k "graphed" > gpurun_out/r04_y/t_models.log 2>&1 || { tail -60 gpurun_out/r04_y/t_models.log; exit 1; }
tail -2 gpurun_out/r04_y/t_models.log
