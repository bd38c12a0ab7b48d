set -o pipefail
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "graphed or adamw or trajectory" > $O/gputest_d.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gputest_d.log
python3 bench.py --config notebook --steps 50 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_notebook_sliced_f32_graph.json 2> $O/bench_notebook_graph.err; echo "rc=$?"; tail -3 $O/bench_notebook_graph.err
python3 bench.py --config notebook --eager --steps 50 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_notebook_sliced_f32_eager.json 2> /dev/null; echo "rc=$?"
python3 - <<'P'
import json
for f in ("graph","eager"):
    j=json.load(open(f'gpurun_out/r04/bench_notebook_sliced_f32_{f}.json'))
    print(f, j['ms_per_step'], j['value'], j['step_hbm_floor'], sum(v['ms_per_step'] for v in j['kernels'].values()))
P
