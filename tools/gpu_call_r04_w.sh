#!/bin/bash
# N > 1 rehearsals of bench.py at HEAD: two ranks on the one GPU over gloo, and one rank over RCCL
set -e
mkdir -p gpurun_out/r04_w
export HSA_ENABLE_IPC_MODE_LEGACY=0
CLIPK_REHEARSE_ONE_GPU=1 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --batch 256 > gpurun_out/r04_w/bench_2ranks_gloo.json 2> gpurun_out/r04_w/bench_2ranks_gloo.err || { tail -30 gpurun_out/r04_w/bench_2ranks_gloo.err; exit 1; }
CLIPK_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity > gpurun_out/r04_w/bench_1rank_rccl.json 2> gpurun_out/r04_w/bench_1rank_rccl.err || { tail -30 gpurun_out/r04_w/bench_1rank_rccl.err; exit 1; }
python3 - <<'PY'
import json
for f in ('bench_2ranks_gloo', 'bench_1rank_rccl'):
    j = json.loads(open('gpurun_out/r04_w/%s.json' % f).read().strip().splitlines()[-1]); print(f, j['n_gpus'], j['ms_per_step'], j['value'], j.get('rccl'))
PY
