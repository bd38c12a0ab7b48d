#!/bin/bash
# head-major addressing probe for the hd-24 whole-head attention kernels
set -e
mkdir -p gpurun_out/r04_h
timeout -k 10 300 python3 tools/exp_attn_headmajor.py 1024 > gpurun_out/r04_h/headmajor.txt 2>&1
cat gpurun_out/r04_h/headmajor.txt
