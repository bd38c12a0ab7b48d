set -o pipefail
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/gputest_il.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputest_il.log; grep -n "^E " $O/gputest_il.log | head -5
for i in 1 2; do for v in 0 1; do
CLIPK_ROPE_INTERLEAVED=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_il${v}_$i.json 2>/dev/null; echo "il=$v rc=$?"
done; done
python3 - <<'P'
import json
for v in (0,1):
  for i in (1,2):
    j=json.load(open(f'gpurun_out/r04/bench_il{v}_{i}.json')); k=j['kernels']
    print(v, i, j['ms_per_step'], 'attn fwd', k['attn_fwd']['ms_per_step'], 'bwd', k['attn_bwd']['ms_per_step'], 'gemm', k['gemm_nt']['ms_per_step'], 'wgrad', k['gemm_wgrad']['ms_per_step'])
P
