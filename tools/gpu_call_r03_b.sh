set -o pipefail
O=gpurun_out/r03_b; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gputest.log
bash tools/profile_bench.sh r03_prof2 > $O/profile.log 2>&1; echo "profile rc=$?"; tail -3 $O/profile.log
python3 bench.py --steps 20 --warmup 5 > $O/bench_c2_B1024.json 2> $O/bench_c2_B1024.err; echo "bench rc=$?"
head -c 1500 $O/bench_c2_B1024.json; echo
