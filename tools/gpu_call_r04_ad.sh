#!/bin/bash
set -e
mkdir -p gpurun_out/r04_ad
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "layernorm" > gpurun_out/r04_ad/t_kernels.log 2>&1 || { tail -40 gpurun_out/r04_ad/t_kernels.log; exit 1; }
tail -2 gpurun_out/r04_ad/t_kernels.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_models.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r04_ad/t_models.log 2>&1 || { tail -60 gpurun_out/r04_ad/t_models.log; exit 1; }
tail -2 gpurun_out/r04_ad/t_models.log
for d in 1 0; do
CLIPK_DEFER_LN_GRADS=$d timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04_ad/nb_d$d.json 2> gpurun_out/r04_ad/nb_d$d.err
CLIPK_DEFER_LN_GRADS=$d timeout -k 10 200 python3 bench.py --config c1 --steps 500 --warmup 50 --no-cpu-baseline --single-stream > gpurun_out/r04_ad/c1_1s_d$d.json 2> gpurun_out/r04_ad/c1_d$d.err
CLIPK_DEFER_LN_GRADS=$d timeout -k 10 200 python3 bench.py --config c1 --steps 500 --warmup 50 --no-cpu-baseline > gpurun_out/r04_ad/c1_2s_d$d.json 2> gpurun_out/r04_ad/c1_d$d.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_ad/*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j['ms_per_step'], j['value'], j.get('parity', {}).get('loss_abs_err'))
PY
