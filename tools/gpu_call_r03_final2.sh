set -o pipefail
O=gpurun_out/r03_final2; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/gputest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
bash tools/profile_bench.sh r03_prof5 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -3 $O/profile.log
cp gpurun_out/r03_prof5/traffic_gemm_nt.json profiles/traffic_gemm_nt.json
python3 bench.py --steps 20 --warmup 5 > $O/bench_c2_B1024.json 2> $O/bench_c2_B1024.err; echo "bench rc=$?"
python3 - <<'P'
import json
j=json.load(open('gpurun_out/r03_final2/bench_c2_B1024.json'))
r=j['roofline']
print(j['value'], j['ms_per_step'], r['frac'], r['traffic'], r.get('traffic_stale'), j['parity']['loss_abs_err'], {k:v['ms_per_step'] for k,v in j['kernels'].items() if isinstance(v,dict)})
P
