#!/bin/bash
set -e
mkdir -p gpurun_out/r04_nbprof
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_nbprof/prof -o nb -- python3 $GRAFT_REPO_ROOT/bench.py --config notebook --steps 40 --warmup 10 --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/gpurun_out/r04_nbprof/nb_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04_nbprof/nb_prof.err
cd $GRAFT_REPO_ROOT
find gpurun_out/r04_nbprof/prof -name '*kernel_trace.csv' -delete; find gpurun_out/r04_nbprof/prof -name '*.db' -delete
timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 > gpurun_out/r04_nbprof/nb_graph.json 2> gpurun_out/r04_nbprof/nb_graph.err
timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 --dropout 0.1 --no-cpu-baseline > gpurun_out/r04_nbprof/nb_graph_p01.json 2> gpurun_out/r04_nbprof/nb_graph_p01.err
python3 - <<'PY'
import json
for f in ('nb_graph', 'nb_graph_p01'):
    j = json.loads(open('gpurun_out/r04_nbprof/%s.json' % f).read().strip().splitlines()[-1]); print(f, j['ms_per_step'], j['value'], j.get('parity', {}).get('loss_abs_err'), j['step_hbm_floor']['frac_of_floor'])
PY
