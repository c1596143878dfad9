#!/bin/bash
# The shipped optimiser kernel back to back (cache-warm, without the next sample's blocks) against its duration
# in the step: a variant build with -DDZ_ADAM_REPEAT=3 (tools/build_variants.sh dz_rainbow adam_rep:"-DDZ_ADAM_REPEAT=3 ...")
# enqueues it three more times behind itself -- wrong parameters, right durations -- and this script reads the four
# launches of every step off a rocprofv3 kernel trace of the bench loop.  Further timing-only switches of the same
# build: -DDZ_ADAM_CHEAP (the per-element update as three additions), -DDZ_ADAM_NOG (one eighth of the tile's FMA chains).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/q; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cp $R/dqn_zoo_amd/libdqnzoo_hip.so /tmp/keep.so; cp $R/tools/ab/adam_rep.so $R/dqn_zoo_amd/libdqnzoo_hip.so
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python $R/bench.py --steps 200 --warmup 50 --cpu-seconds 0 --prof-steps 0 --other-configs 0 --sustain-steps 0 --agent-form-steps 0 --agent-loop-frames 0 > $OUT/kt.log 2>&1 < /dev/null
cp /tmp/keep.so $R/dqn_zoo_amd/libdqnzoo_hip.so
python - <<P
import csv, glob
f = glob.glob('$OUT/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ad = [(i, r) for i, r in enumerate(rows) if 'adam_onfly' in r['Kernel_Name']]
ad = ad[len(ad) // 2:]
# group consecutive adam launches
groups = []; cur = []
for i, r in ad:
  if cur and i != cur[-1][0] + 1: groups.append(cur); cur = []
  cur.append((i, r))
groups.append(cur)
groups = [g for g in groups if len(g) == 4]
import statistics
for k in range(4):
  d = [(int(g[k][1]['End_Timestamp']) - int(g[k][1]['Start_Timestamp'])) / 1e3 for g in groups]
  print('adam launch %d of its group: median %.2f us  mean %.2f  (n=%d)' % (k, statistics.median(d), sum(d) / len(d), len(d)))
P
