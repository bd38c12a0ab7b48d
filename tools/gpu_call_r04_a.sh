set -o pipefail
O=gpurun_out/r04; mkdir -p $O
python -m pytest tests -m gpu -q -s > $O/gputest_a.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gputest_a.log
grep -E "trajectory:|notebook|tri-modal" $O/gputest_a.log | head -20
python3 bench.py --config notebook --variant full-bf16 --steps 5 --warmup 2 > $O/bench_notebook_full_bf16.json 2> $O/bench_notebook_full_bf16.err; echo "nb full rc=$?"
python3 bench.py --config notebook --steps 20 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_notebook_sliced_f32.json 2> $O/bench_notebook_sliced_f32.err; echo "nb sliced rc=$?"
python3 bench.py --steps 10 --warmup 3 > $O/bench_c2_B1024_a.json 2> $O/bench_c2_B1024_a.err; echo "bench rc=$?"
python3 - <<'P'
import json
for f in ("bench_notebook_full_bf16","bench_notebook_sliced_f32"):
    j=json.load(open(f'gpurun_out/r04/{f}.json'))
    print(f, j['value'], j['ms_per_step'], j['roofline']['frac'], j.get('parity',{}).get('loss_abs_err'), j.get('step_hbm_floor',{}).get('frac_of_floor'))
    print({k:(v['launches_per_step'], v['ms_per_step']) for k,v in j['kernels'].items()})
j=json.load(open('gpurun_out/r04/bench_c2_B1024_a.json'))
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline'].get('frac_hbm'), j['parity']['loss_abs_err'])
for t in j['parity']['trajectory']: print(t)
print(j['parity']['trajectory_sign_flips_step0'])
P
