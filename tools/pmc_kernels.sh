#!/bin/bash
# Counter passes over a kernel driver (default tools/prof_gemm.py): one --stats pass and separate --pmc passes (8 SQ slots /
# 4 TCC slots per pass; FETCH_SIZE and WRITE_SIZE cannot share one), reduced to a markdown table by tools/pmc_table.py.
#     bash tools/pmc_kernels.sh <tag> [B] [driver.py] [kernel-name match]
set -o pipefail
TAG=${1:-pmc_gemm}
B=${2:-1024}
PROG=${3:-tools/prof_gemm.py}
MATCH=${4:-_kernel}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
run() {   # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python3 $PROG $B 6 \
      > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; return 1; }
  echo "pass $name done"
}
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $PROG $B 6 > $OUT/stats.log 2>&1 || exit 1
echo "stats pass done"
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT || exit 2
run sq2 SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE || echo "sq2 skipped"
run fetch FETCH_SIZE TCC_HIT_sum || exit 3
run write WRITE_SIZE TCC_MISS_sum || exit 4
FILES=$(find $OUT -name '*counter_collection.csv' | sort)
python3 tools/pmc_table.py --match "$MATCH" --skip 1 $FILES > $OUT/pmc_table.md
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*.db' -delete
cat $OUT/pmc_table.md
