set -o pipefail
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gelu_grad_code or layernorm" > $O/gputest_c.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputest_c.log
for i in 1 2; do
for v in 0 1; do
CLIPK_LN_BWD_FROM_OUTPUT=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_ln_fromy${v}_$i.json 2> $O/bench_ln_fromy${v}_$i.err; echo "bench fromy=$v rc=$?"
done; done
CLIPK_LN_BWD_FROM_OUTPUT=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_ln_fromy1_parity.json 2> $O/bench_ln_fromy1_parity.err
python3 - <<'P'
import json
for v in (0,1):
  for i in (1,2):
    j=json.load(open(f'gpurun_out/r04/bench_ln_fromy{v}_{i}.json'))
    k=j['kernels']
    print(v, i, j['ms_per_step'], k['layernorm_bwd']['ms_per_step'], k['layernorm_fwd']['ms_per_step'], k['gemm_nt']['ms_per_step'])
j=json.load(open('gpurun_out/r04/bench_ln_fromy1_parity.json'))
print('parity with fromy:', j['parity']['loss_abs_err'], j['parity'].get('trajectory_max_abs_err'), [t['abs_err'] for t in j['parity']['trajectory']])
P
