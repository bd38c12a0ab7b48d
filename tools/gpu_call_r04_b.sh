set -o pipefail
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm or gelu" > $O/gputest_b.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gputest_b.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_B1024_b.json 2> $O/bench_c2_B1024_b.err; echo "bench rc=$?"
python3 - <<'P'
import json
j=json.load(open('gpurun_out/r04/bench_c2_B1024_b.json'))
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['parity']['loss_abs_err'], j['parity'].get('trajectory_max_abs_err'))
for k,v in j['roofline']['by_epilogue'].items(): print(k, v)
print({k:v['ms_per_step'] for k,v in j['kernels'].items() if isinstance(v,dict)})
P
