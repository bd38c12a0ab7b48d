#!/bin/bash
# notebook model: in-place LayerNorm parameter gradients + block-level f32 autograd nodes; towers on two HIP streams (A/B)
set -e
mkdir -p gpurun_out/r04_k
timeout -k 10 600 python3 -m pytest tests/test_gpu_models.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r04_k/t_models.log 2>&1 || { tail -30 gpurun_out/r04_k/t_models.log; exit 1; }
tail -3 gpurun_out/r04_k/t_models.log
for rnd in 1 2; do
  timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04_k/nb_graph_1s_r$rnd.json 2> gpurun_out/r04_k/nb_graph_1s_r$rnd.err
  timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 --no-cpu-baseline --dual-stream > gpurun_out/r04_k/nb_graph_2s_r$rnd.json 2> gpurun_out/r04_k/nb_graph_2s_r$rnd.err
done
timeout -k 10 200 python3 bench.py --config notebook --eager --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r04_k/nb_eager_1s.json 2> gpurun_out/r04_k/nb_eager_1s.err
timeout -k 10 200 python3 bench.py --config notebook --eager --steps 100 --warmup 20 --no-cpu-baseline --dual-stream > gpurun_out/r04_k/nb_eager_2s.json 2> gpurun_out/r04_k/nb_eager_2s.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_k/nb_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j['ms_per_step'], j['value'], j.get('parity', {}).get('loss_abs_err'), j['loss'])
    except Exception as e:
        print(f, 'ERR', e); print(open(f.replace('.json', '.err')).read()[-1500:])
PY
