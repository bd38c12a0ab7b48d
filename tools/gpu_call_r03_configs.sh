set -o pipefail
O=gpurun_out/r03_cfg; mkdir -p $O
for c in c1 c3sim c5; do python3 bench.py --config $c --steps 50 --warmup 10 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; done
python3 bench.py --config c4 --steps 5 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"
python3 bench.py --lengths ragged-packed --steps 8 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_ragged_packed.json 2> $O/bench_ragged_packed.err; echo "packed rc=$?"
python3 bench.py --lengths ragged-padded --steps 8 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_ragged_padded.json 2> $O/bench_ragged_padded.err; echo "padded rc=$?"
python3 bench.py --batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_c2_B512.json 2> $O/bench_c2_B512.err; echo "b512 rc=$?"
python3 - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_cfg/bench_*.json')):
    j=json.load(open(f)); print(f.split('/')[-1], j['value'], j['unit'], j['ms_per_step'], (j.get('parity') or {}).get('loss_abs_err', (j.get('parity') or {}).get('max_abs_err')))
P
