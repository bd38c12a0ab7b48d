#!/bin/bash
set -e
mkdir -p gpurun_out/r04_v
timeout -k 10 400 python3 tools/exp_skinny_graph.py > gpurun_out/r04_v/skinny_graph.txt 2>&1 || { tail -20 gpurun_out/r04_v/skinny_graph.txt; exit 1; }
cat gpurun_out/r04_v/skinny_graph.txt
