#!/bin/bash
# one-launch f32 parameter gradients: kernel test, f32-model tests, notebook bench (graph + eager)
set -e
mkdir -p gpurun_out/r04_j
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wgrad_f32 or gemm_f32 or colsum" > gpurun_out/r04_j/t_kernels.log 2>&1 || { tail -30 gpurun_out/r04_j/t_kernels.log; exit 1; }
tail -3 gpurun_out/r04_j/t_kernels.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_models.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r04_j/t_models.log 2>&1 || { tail -30 gpurun_out/r04_j/t_models.log; exit 1; }
tail -3 gpurun_out/r04_j/t_models.log
timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 > gpurun_out/r04_j/nb_graph.json 2> gpurun_out/r04_j/nb_graph.err
timeout -k 10 200 python3 bench.py --config notebook --eager --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r04_j/nb_eager.json 2> gpurun_out/r04_j/nb_eager.err
timeout -k 10 200 python3 bench.py --config c1 --steps 200 --warmup 20 > gpurun_out/r04_j/c1.json 2> gpurun_out/r04_j/c1.err
python3 - <<'PY'
import json
for f in ('nb_graph', 'nb_eager', 'c1'):
    j = json.loads(open('gpurun_out/r04_j/%s.json' % f).read().strip().splitlines()[-1]); print(f, j['ms_per_step'], j['value'], j.get('parity', {}).get('loss_abs_err'))
PY
