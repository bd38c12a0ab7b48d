#!/bin/bash
set -e
mkdir -p gpurun_out/r04_ab
for rnd in 1 2 3 4 5; do
timeout -k 10 200 python3 bench.py --config c1 --steps 500 --warmup 50 --no-cpu-baseline --no-parity > gpurun_out/r04_ab/c1_2s_r$rnd.json 2> gpurun_out/r04_ab/c1_2s.err
timeout -k 10 200 python3 bench.py --config c1 --steps 500 --warmup 50 --single-stream --no-cpu-baseline --no-parity > gpurun_out/r04_ab/c1_1s_r$rnd.json 2> gpurun_out/r04_ab/c1_1s.err
done
python3 - <<'PY'
import json, glob
for k in ('1s', '2s'):
    v = [json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'] for f in sorted(glob.glob('gpurun_out/r04_ab/c1_%s_r*.json' % k))]
    print(k, v)
PY
