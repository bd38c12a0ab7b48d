#!/bin/bash
set -e
mkdir -p gpurun_out/r04_ae
for rnd in 1 2 3 4; do
for pr in normal high; do
CLIPK_BRANCH_PRIORITY=$pr timeout -k 10 200 python3 bench.py --config c1 --steps 500 --warmup 50 --no-cpu-baseline --no-parity > gpurun_out/r04_ae/c1_${pr}_r$rnd.json 2> gpurun_out/r04_ae/c1.err
CLIPK_BRANCH_PRIORITY=$pr timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 --no-cpu-baseline --no-parity > gpurun_out/r04_ae/nb_${pr}_r$rnd.json 2> gpurun_out/r04_ae/nb.err
done
done
python3 - <<'PY'
import json, glob
for k in ('c1_normal', 'c1_high', 'nb_normal', 'nb_high'):
    v = [json.loads(open(f).read().strip().splitlines()[-1])['ms_per_step'] for f in sorted(glob.glob('gpurun_out/r04_ae/%s_r*.json' % k))]
    print(k, v)
PY
