#!/bin/bash
set -e
mkdir -p gpurun_out/r04_r
for rnd in 1 2; do
timeout -k 10 300 python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/r04_r/c2_1s_r$rnd.json 2> gpurun_out/r04_r/c2_1s.err
timeout -k 10 300 python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-parity --dual-stream > gpurun_out/r04_r/c2_2s_r$rnd.json 2> gpurun_out/r04_r/c2_2s.err
done
timeout -k 10 300 python3 bench.py --steps 15 --warmup 5 --no-cpu-baseline --no-parity --dual-stream --micro-batches 2 > gpurun_out/r04_r/c2_2s_mb2.json 2> gpurun_out/r04_r/c2_2s_mb2.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_r/c2_*.json')):
    j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j['ms_per_step'], j['value'], j['roofline']['frac'])
PY
