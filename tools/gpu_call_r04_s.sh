#!/bin/bash
set -e
mkdir -p gpurun_out/r04_s
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_s/prof -o nb -- python3 $GRAFT_REPO_ROOT/bench.py --config notebook --steps 40 --warmup 10 --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/gpurun_out/r04_s/nb_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04_s/nb_prof.err
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r04_s/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# keep the last ~25 steps: find adamw launches as step delimiters
idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
# 3 adamw per step; take a window of 10 steps near the end (graph replays)
ends = idx[2::3]
lo, hi = ends[-12], ends[-2]
win = rows[lo + 1:hi + 1]
t0, t1 = int(win[0]['Start_Timestamp']), int(win[-1]['End_Timestamp'])
print('10 steps wall', (t1 - t0) / 1e6, 'ms; launches per step', len(win) / 10)
busy = collections.Counter(); cnt = collections.Counter()
for r in win:
    n = r['Kernel_Name'].split('(')[0][-60:]
    busy[n] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); cnt[n] += 1
tot = sum(busy.values())
print('sum of kernel time per step', tot / 1e7, 'ms')
for n, b in busy.most_common(28):
    print(f'{b/1e7*1e3:8.1f} us/step {cnt[n]/10:6.1f} x {b/cnt[n]/1e3:7.1f} us  {n}')
# union of busy intervals (any kernel running) -> idle time
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in win)
cur_s, cur_e, union = iv[0][0], iv[0][1], 0
for s, e in iv[1:]:
    if s > cur_e:
        union += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print('GPU busy (any kernel) per step', union / 1e7, 'ms; idle', (t1 - t0 - union) / 1e7, 'ms')
open('gpurun_out/r04_s/window.csv', 'w').write('\n'.join(','.join([r['Kernel_Name'].split('(')[0][-50:].replace(',', ';'), r['Start_Timestamp'], r['End_Timestamp'], r.get('Stream_Id', r.get('Queue_Id', ''))]) for r in rows[ends[-3] + 1:ends[-2] + 1]))
PY
find gpurun_out/r04_s/prof -name '*kernel_trace.csv' -delete; find gpurun_out/r04_s/prof -name '*.db' -delete
