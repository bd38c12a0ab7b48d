#!/bin/bash
# the non-default switches still work: rotate-half RoPE path (CLIPK_ROPE_INTERLEAVED=0), lazy bf16 refresh (CLIPK_BATCH_REFRESH=0)
set -e
mkdir -p gpurun_out/r04_z
CLIPK_ROPE_INTERLEAVED=0 timeout -k 10 900 python3 -m pytest tests/test_gpu_models.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r04_z/t_il0.log 2>&1 || { tail -40 gpurun_out/r04_z/t_il0.log; exit 1; }
tail -2 gpurun_out/r04_z/t_il0.log
CLIPK_BATCH_REFRESH=0 timeout -k 10 900 python3 -m pytest tests/test_gpu_models.py -x -q -m gpu > gpurun_out/r04_z/t_lazy.log 2>&1 || { tail -40 gpurun_out/r04_z/t_lazy.log; exit 1; }
tail -2 gpurun_out/r04_z/t_lazy.log
