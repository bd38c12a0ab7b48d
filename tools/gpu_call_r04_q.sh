#!/bin/bash
set -e
mkdir -p gpurun_out/r04_q
timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04_q/nb_graph_p0.json 2> gpurun_out/r04_q/nb_graph_p0.err
timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 --no-cpu-baseline --dropout 0.1 > gpurun_out/r04_q/nb_graph_p01.json 2> gpurun_out/r04_q/nb_graph_p01.err
timeout -k 10 200 python3 bench.py --config notebook --steps 100 --warmup 20 --no-cpu-baseline --dropout 0.1 --eager > gpurun_out/r04_q/nb_eager_p01.json 2> gpurun_out/r04_q/nb_eager_p01.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_q/nb_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j['ms_per_step'], j['value'], j.get('parity', {}).get('loss_abs_err'), j['loss'], j['config'].get('dropout'))
    except Exception as e:
        print(f, 'ERR', e); print(open(f.replace('.json', '.err')).read()[-2000:])
PY
