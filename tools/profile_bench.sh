#!/bin/bash
# Profile the default bench.py workload on the GPU box: (1) rocprofv3 --kernel-trace --stats summary, (2) + (3) the two
# PMC passes (FETCH_SIZE, WRITE_SIZE) reduced to profiles/traffic_gemm_nt.json by tools/pmc_traffic.py.
#   bash tools/profile_bench.sh <tag>      (run through gpurun; results under gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
ARGS="--steps 6 --warmup 2 --no-cpu-baseline --no-parity"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o c2 -- python3 bench.py $ARGS > $OUT/bench_stats.json 2> $OUT/bench_stats.err || exit 1
echo "stats pass done" 
PARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-timers"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o c2 -- python3 bench.py $PARGS > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err || exit 2
echo "fetch pass done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o c2 -- python3 bench.py $PARGS > $OUT/bench_write.json 2> $OUT/bench_write.err || exit 3
echo "write pass done"
F=$(find $OUT/pmc_fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/pmc_write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py --fetch $F --write $W --key c2_B1024_L256_full --out $OUT/traffic_gemm_nt.json \
  --cmd "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py $PARGS" || exit 4
# keep the transfer small: only the stats CSV and the reduced JSON travel back whole; drop the big per-dispatch tables
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*counter_collection.csv' -size +20M -delete
find $OUT -name '*.db' -delete
ls -la $OUT $OUT/stats/* | head -40
