#!/bin/bash
set -e
mkdir -p gpurun_out/r04_c4
timeout -k 10 600 python3 bench.py --config c4 --steps 5 --warmup 2 > gpurun_out/r04_c4/bench_c4.json 2> gpurun_out/r04_c4/bench_c4.err || { tail -20 gpurun_out/r04_c4/bench_c4.err; exit 1; }
timeout -k 10 400 python3 bench.py --lengths ragged-packed --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_c4/bench_ragged_packed.json 2> gpurun_out/r04_c4/bench_ragged_packed.err || { tail -20 gpurun_out/r04_c4/bench_ragged_packed.err; exit 1; }
timeout -k 10 400 python3 bench.py --lengths ragged-padded --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r04_c4/bench_ragged_padded.json 2> gpurun_out/r04_c4/bench_ragged_padded.err || { tail -20 gpurun_out/r04_c4/bench_ragged_padded.err; exit 1; }
python3 - <<'PY'
import json
for f in ('bench_c4', 'bench_ragged_packed', 'bench_ragged_padded'):
    j = json.loads(open('gpurun_out/r04_c4/%s.json' % f).read().strip().splitlines()[-1]); print(f, j['ms_per_step'], j['value'], j.get('parity', {}).get('loss_abs_err'), j['roofline']['frac'])
PY
