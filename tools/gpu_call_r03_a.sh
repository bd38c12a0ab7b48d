set -o pipefail
O=gpurun_out/r03_a; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gputest.log
bash tools/profile_bench.sh r03_prof > $O/profile.log 2>&1; echo "profile rc=$?"; tail -3 $O/profile.log
for c in c1 c3sim c5; do python3 bench.py --config $c --steps 50 --warmup 10 > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; done
python3 bench.py --config c4 --steps 5 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"
python3 bench.py --lengths ragged-packed --steps 8 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_ragged_packed.json 2> $O/bench_ragged_packed.err; echo "packed rc=$?"
python3 bench.py --lengths ragged-padded --steps 8 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_ragged_padded.json 2> $O/bench_ragged_padded.err; echo "padded rc=$?"
python3 bench.py --batch 512 --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_c2_B512.json 2> $O/bench_c2_B512.err; echo "b512 rc=$?"
for f in $O/bench_*.json; do echo "== $f"; head -c 600 $f; echo; done
