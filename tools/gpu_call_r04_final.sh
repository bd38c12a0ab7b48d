set -o pipefail
O=gpurun_out/r04_final; mkdir -p $O
bash tools/profile_bench.sh r04_prof > $O/profile.log 2>&1; echo "profile rc=$?"; tail -3 $O/profile.log
cp gpurun_out/r04_prof/traffic_gemm_nt.json profiles/traffic_gemm_nt.json
python3 bench.py --steps 20 --warmup 5 > $O/bench_c2_B1024.json 2> $O/bench_c2_B1024.err; echo "bench rc=$?"
python3 bench.py --config c2 --batch 512 --steps 20 --warmup 5 > $O/bench_c2_B512.json 2> $O/bench_c2_B512.err; echo "bench512 rc=$?"
python3 bench.py --config c3sim --steps 50 --warmup 5 > $O/bench_c3sim.json 2> $O/bench_c3sim.err; echo "c3sim rc=$?"
python3 bench.py --config c5 --steps 50 --warmup 5 > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
python3 bench.py --config c1 --steps 50 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err; echo "c1 rc=$?"
python3 bench.py --config notebook --steps 50 --warmup 5 > $O/bench_notebook_sliced_f32_graph.json 2> $O/bench_notebook.err; echo "nb rc=$?"
python3 bench.py --config notebook --dropout 0.1 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_notebook_dropout01_graph.json 2>/dev/null; echo "nb dropout rc=$?"
python3 bench.py --config notebook --eager --steps 50 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_notebook_sliced_f32_eager.json 2>/dev/null; echo "nb eager rc=$?"
python3 bench.py --config notebook --variant full-bf16 --eager --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_notebook_full_bf16.json 2>/dev/null; echo "nb full rc=$?"
python3 - <<'P'
import json
j=json.load(open('gpurun_out/r04_final/bench_c2_B1024.json'))
r=j['roofline']
print(j['value'], j['ms_per_step'], r['frac'], r['traffic'], r.get('frac_hbm'), r.get('traffic_stale'), j['parity']['loss_abs_err'], j['parity']['trajectory_max_abs_err'], j['parity']['trajectory_max_forward_abs_err'])
j=json.load(open('gpurun_out/r04_final/bench_c2_B512.json')); print('B512', j['value'], j['ms_per_step'], j['parity']['loss_abs_err'], j.get('cpu_baseline',{}).get('value'))
for f in ('c3sim','c5','c1','notebook_sliced_f32_graph','notebook_dropout01_graph','notebook_sliced_f32_eager','notebook_full_bf16'):
    j=json.load(open(f'gpurun_out/r04_final/bench_{f}.json')); print(f, j['value'], j['ms_per_step'], j['roofline']['bound'], j['roofline']['frac'], j.get('parity',{}).get('loss_abs_err', j.get('parity',{}).get('max_abs_err')))
P
