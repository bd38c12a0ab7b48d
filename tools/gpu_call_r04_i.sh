#!/bin/bash
# f32 parameter gradients on a side stream: A/B on the notebook model (graph replay and eager), then the f32-model tests with it on
set -e
mkdir -p gpurun_out/r04_i
for rnd in 1 2; do
  for side in 0 1; do
    CLIPK_F32_WGRAD_SIDE_STREAM=$side timeout -k 10 200 python3 bench.py --config notebook --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04_i/nb_graph_side${side}_r${rnd}.json 2> gpurun_out/r04_i/nb_graph_side${side}_r${rnd}.err
    CLIPK_F32_WGRAD_SIDE_STREAM=$side timeout -k 10 200 python3 bench.py --config notebook --eager --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r04_i/nb_eager_side${side}_r${rnd}.json 2> gpurun_out/r04_i/nb_eager_side${side}_r${rnd}.err
  done
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r04_i/nb_*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], j['ms_per_step'], j['value'], j.get('parity', {}).get('loss_abs_err'))
    except Exception as e:
        print(f, 'ERR', e)
PY
CLIPK_F32_WGRAD_SIDE_STREAM=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_models.py -x -q -m gpu -k "notebook or trimodal or graphed or f32 or clip_opt or slice" > gpurun_out/r04_i/tests_side1.log 2>&1 || true
tail -5 gpurun_out/r04_i/tests_side1.log
