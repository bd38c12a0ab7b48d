#!/bin/bash
set -e
mkdir -p gpurun_out/r04_x
timeout -k 10 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wgrad" > gpurun_out/r04_x/t_kernels.log 2>&1 || { tail -40 gpurun_out/r04_x/t_kernels.log; exit 1; }
tail -2 gpurun_out/r04_x/t_kernels.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_models.py -x -q -m gpu -k "reproducible or c2_full" > gpurun_out/r04_x/t_models.log 2>&1 || { tail -40 gpurun_out/r04_x/t_models.log; exit 1; }
tail -2 gpurun_out/r04_x/t_models.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_x/prof -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/gpurun_out/r04_x/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04_x/bench.err
cd $GRAFT_REPO_ROOT
find gpurun_out/r04_x/prof -name '*kernel_trace.csv' -delete; find gpurun_out/r04_x/prof -name '*.db' -delete
python3 - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/r04_x/prof/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'wgrad' in r['Name'] or 'colreduce' in r['Name']:
        print(r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3)
j = json.loads(open('gpurun_out/r04_x/bench.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['kernels']['gemm_wgrad'])
PY
