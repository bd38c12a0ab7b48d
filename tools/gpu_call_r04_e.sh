set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nbstats -o nb -- python3 bench.py --config notebook --steps 50 --warmup 5 --no-cpu-baseline --no-parity > $O/bench_notebook_prof.json 2> $O/bench_notebook_prof.err; echo "rc=$?"
find $O/nbstats -name '*kernel_trace.csv' -delete; find $O/nbstats -name '*.db' -delete
F=$(find $O/nbstats -name '*kernel_stats.csv' | head -1); echo $F; head -30 $F | cut -c1-220
