#!/bin/bash
# full GPU suite + smoke + default bench at HEAD
set -e
mkdir -p gpurun_out/r04_o
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > gpurun_out/r04_o/gputest.log 2>&1 || { tail -40 gpurun_out/r04_o/gputest.log; exit 1; }
tail -3 gpurun_out/r04_o/gputest.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04_o/smoke.log 2>&1 || { tail -20 gpurun_out/r04_o/smoke.log; exit 1; }
tail -2 gpurun_out/r04_o/smoke.log
date +%s > gpurun_out/r04_o/t0
timeout -k 10 400 python3 bench.py > gpurun_out/r04_o/bench_default.json 2> gpurun_out/r04_o/bench_default.err
date +%s > gpurun_out/r04_o/t1
python3 - <<'PY'
import json
j = json.loads(open('gpurun_out/r04_o/bench_default.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline'].get('traffic'), j['parity']['loss_abs_err'], j['cpu_baseline']['value'])
print('wall', int(open('gpurun_out/r04_o/t1').read()) - int(open('gpurun_out/r04_o/t0').read()), 's')
PY
