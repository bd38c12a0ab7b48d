set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemm_f32" > $O/t_f32e.log 2>&1; tail -2 $O/t_f32e.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/skstats -o sk -- python3 tools/exp_gemm_f32_skinny.py 32 > $O/sk.log 2>&1; echo "rc=$?"
python3 - <<'P'
import csv, glob, collections
f=glob.glob('/root/repo/gpurun_out/r04/skstats/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
# group consecutive by kernel name order: print per kernel name & grid
agg=collections.OrderedDict()
for r in rows:
    k=(r['Kernel_Name'][:60], r.get('Grid_Size_X',r.get('Grid_Size','')), r.get('Grid_Size_Y',''))
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    agg.setdefault(k,[]).append(d)
for k,v in agg.items():
    if 'gemm_f32' in k[0]:
        v=sorted(v); print(k, len(v), 'median us', v[len(v)//2]/1e3)
P
find $O/skstats -name '*kernel_trace.csv' -delete; find $O/skstats -name '*.db' -delete
