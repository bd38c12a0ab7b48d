#!/bin/bash
set -e
mkdir -p gpurun_out/r04_aa
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "simce or ce_" > gpurun_out/r04_aa/t_kernels.log 2>&1 || { tail -40 gpurun_out/r04_aa/t_kernels.log; exit 1; }
tail -2 gpurun_out/r04_aa/t_kernels.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_models.py tests/test_gpu_configs.py tests/test_gpu_rccl.py -x -q -m gpu > gpurun_out/r04_aa/t_models.log 2>&1 || { tail -60 gpurun_out/r04_aa/t_models.log; exit 1; }
tail -2 gpurun_out/r04_aa/t_models.log
timeout -k 10 200 python3 bench.py --config c1 --steps 300 --warmup 30 > gpurun_out/r04_aa/c1.json 2> gpurun_out/r04_aa/c1.err
timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 > gpurun_out/r04_aa/nb.json 2> gpurun_out/r04_aa/nb.err
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_aa/c2.json 2> gpurun_out/r04_aa/c2.err
python3 - <<'PY'
import json
for f in ('c1', 'nb', 'c2'):
    j = json.loads(open('gpurun_out/r04_aa/%s.json' % f).read().strip().splitlines()[-1]); print(f, j['ms_per_step'], j['value'], j.get('parity', {}).get('loss_abs_err'))
PY
