#!/bin/bash
# GPU timeline of the agent loop, kernel by kernel, over two learn periods (rocprofv3 --kernel-trace):
#   tools/loop_timeline.sh <agent> [agent_loop_bench switches...]
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/q; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
A=${1:-rainbow}; shift
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/tools/agent_loop_bench.py 3000 $A "$@" > $OUT/kt.log 2>&1 < /dev/null
grep "agent loop" $OUT/kt.log
python - <<P
import csv, glob
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]
qk = 'Queue_Id' if 'Queue_Id' in rows[0] else None
def short(n):
  n = n.replace('(anonymous namespace)::', '').replace('void ', '')
  return n[:34]
# a cycle starts at the decision kernel that is followed (within the next 6 kernels) by a sample/gather launch
idx = [i for i, r in enumerate(rows) if 'act_one' in r['Kernel_Name']]
learn = [i for i in idx if any('sample' in rows[j]['Kernel_Name'] or 'adam' in rows[j]['Kernel_Name'] for j in range(i + 1, min(i + 16, len(rows))) if j not in idx[idx.index(i) + 1:idx.index(i) + 2])]
starts = []
for i in idx:
  nxt = [j for j in idx if j > i]
  end = nxt[0] if nxt else len(rows)
  if any('adam' in rows[j]['Kernel_Name'] for j in range(i, end)): starts.append(i)
if len(starts) >= 4:
  a, b = starts[1], starts[3]
  t0 = int(rows[a]['Start_Timestamp'])
  prev_end = int(rows[a - 1]['End_Timestamp'])
  for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%8.1f  +%5.1f  gap %5.1f  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get(qk, '') if qk else '', short(r['Kernel_Name'])))
    prev_end = max(prev_end, e)
  print('two learn periods: %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
P
