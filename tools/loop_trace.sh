#!/bin/bash
# GPU timeline of the Rainbow agent loop: busy time and idle gaps per learn period (4 frames)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/q; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/tools/agent_loop_bench.py 3000 ${1:-rainbow} > $OUT/kt.log 2>&1 < /dev/null
grep "agent loop" $OUT/kt.log
python - <<P
import csv, glob
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]          # steady state
def short(n):
  for k in ('act_one', 'adam', 'replay_insert', 'insert', 'sample_gather', 'ConvFwdOp', 'ConvWgrad', 'fc_stream', 'fc_epilogue', 'head_loss', 'FcFwdOp', 'fc2_bwd', 'fc1_dgrad', 'finalize', 'dense_head', 'dense_fc1'):
    if k in n: return k
  return n[:24]
t0 = int(rows[0]['Start_Timestamp']); t1 = int(rows[-1]['End_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
acts = [r for r in rows if 'act_one' in r['Kernel_Name']]
print('frames', len(acts), 'span/frame %.1f us' % ((t1 - t0) / 1e3 / len(acts)), 'GPU busy/frame %.1f us' % (busy / 1e3 / len(acts)))
# gap in front of every decision kernel (GPU idle while the host prepares the frame)
gaps = []
for i, r in enumerate(rows):
  if 'act_one' in r['Kernel_Name'] and i > 0:
    gaps.append((int(r['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp'])) / 1e3)
gaps.sort()
print('idle in front of a decision: median %.1f us, mean %.1f, p90 %.1f' % (gaps[len(gaps) // 2], sum(gaps) / len(gaps), gaps[int(0.9 * len(gaps))]))
from collections import defaultdict
d = defaultdict(float)
for r in rows: d[short(r['Kernel_Name'])] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:8]: print('  %-16s %.1f us/frame' % (k, v / len(acts)))
P
