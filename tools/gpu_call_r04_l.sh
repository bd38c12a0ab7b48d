#!/bin/bash
set -e
mkdir -p gpurun_out/r04_l
timeout -k 10 900 python3 -m pytest tests/test_gpu_models.py -x -q -m gpu -k "graphed or side_streams or notebook or trimodal" > gpurun_out/r04_l/t_models.log 2>&1 || { tail -40 gpurun_out/r04_l/t_models.log; exit 1; }
tail -3 gpurun_out/r04_l/t_models.log
timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 > gpurun_out/r04_l/nb_graph_2s.json 2> gpurun_out/r04_l/nb_graph_2s.err
timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 --single-stream --no-cpu-baseline > gpurun_out/r04_l/nb_graph_1s.json 2> gpurun_out/r04_l/nb_graph_1s.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_l/nb_*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j['ms_per_step'], j['value'], j.get('parity', {}).get('loss_abs_err'), j['config'].get('hip_streams'))
PY
