#!/bin/bash
set -e
mkdir -p gpurun_out/r04_t
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention_f32 or attn_f32 or dropout or gemm_f32" > gpurun_out/r04_t/t_kernels.log 2>&1 || { tail -40 gpurun_out/r04_t/t_kernels.log; exit 1; }
tail -3 gpurun_out/r04_t/t_kernels.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_models.py -x -q -m gpu -k "notebook or trimodal or f32" > gpurun_out/r04_t/t_models.log 2>&1 || { tail -60 gpurun_out/r04_t/t_models.log; exit 1; }
tail -3 gpurun_out/r04_t/t_models.log
timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04_t/nb_graph.json 2> gpurun_out/r04_t/nb_graph.err
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/r04_t/nb_graph.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['value'], j['parity']['loss_abs_err'])
print({k: v for k, v in j['kernels'].items() if 'attn' in k})
PY
