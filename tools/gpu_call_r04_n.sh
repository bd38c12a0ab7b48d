#!/bin/bash
set -e
mkdir -p gpurun_out/r04_n
timeout -k 10 900 python3 -m pytest tests/test_gpu_models.py -x -q -m gpu -k "icnn" > gpurun_out/r04_n/t_models.log 2>&1 || { tail -40 gpurun_out/r04_n/t_models.log; exit 1; }
tail -3 gpurun_out/r04_n/t_models.log
for rnd in 1 2; do
timeout -k 10 200 python3 bench.py --config c5 --steps 300 --warmup 30 > gpurun_out/r04_n/c5_3s_r$rnd.json 2> gpurun_out/r04_n/c5_3s.err
timeout -k 10 200 python3 bench.py --config c5 --steps 300 --warmup 30 --single-stream > gpurun_out/r04_n/c5_1s_r$rnd.json 2> gpurun_out/r04_n/c5_1s.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_n/c5_*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j['ms_per_step'], j['value'], j.get('roofline', {}).get('frac'), j.get('parity'))
PY
