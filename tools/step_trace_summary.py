"""Per-step summary of a rocprofv3 --kernel-trace CSV of bench.py (sequential or
two-stream pipelined loop): step span = distance between consecutive adam_kernel
ends; per-kernel average duration and launches per step; busy time per queue; how
much of the side queue's busy time overlaps main-queue kernels.

    python tools/step_trace_summary.py <kernel_trace.csv> [steps=100]
"""
import csv
import sys
from collections import defaultdict


def main():
  rows = list(csv.DictReader(open(sys.argv[1])))
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
  for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
  rows.sort(key=lambda r: r['s'])
  import os
  marker = os.environ.get('DZ_STEP_MARKER', 'adam_')   # last kernel of a step (dense RMSProp learners: finalize_grads)
  adam = [i for i, r in enumerate(rows) if marker in r['Kernel_Name'] and 'kernel' in r['Kernel_Name']]
  if len(adam) < n + 1:
    n = len(adam) - 1
  a, b = adam[-n - 1], adam[-1]
  t0, t1 = rows[a]['e'], rows[b]['e']
  seg = [r for r in rows if r['e'] > t0 and r['e'] <= t1]
  qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else ('Stream_Id' if 'Stream_Id' in rows[0] else None)
  print('last %d steps: span/step %.1f us, launches/step %.1f' % (
      n, (t1 - t0) / n / 1e3, len(seg) / n))
  main_q = rows[b].get(qkey) if qkey else None
  byq = defaultdict(list)
  for r in seg:
    byq[r.get(qkey) if qkey else 0].append(r)
  for q, rs in byq.items():
    busy = sum(r['e'] - r['s'] for r in rs)
    print('queue %s%s: %.1f launches/step, busy %.1f us/step' % (
        q, ' (main)' if q == main_q else '', len(rs) / n, busy / n / 1e3))
  if qkey and len(byq) > 1:
    mains = sorted((r['s'], r['e']) for r in byq[main_q])
    ov = 0
    import bisect
    starts = [m[0] for m in mains]
    for q, rs in byq.items():
      if q == main_q:
        continue
      for r in rs:
        i = max(0, bisect.bisect_left(starts, r['s']) - 1)
        while i < len(mains) and mains[i][0] < r['e']:
          ov += max(0, min(r['e'], mains[i][1]) - max(r['s'], mains[i][0]))
          i += 1
    print('side-queue busy time overlapping main-queue kernels: %.1f us/step' % (ov / n / 1e3))
    # main-queue gaps (idle between consecutive main kernels)
    gaps = sum(max(0, mains[i + 1][0] - mains[i][1]) for i in range(len(mains) - 1))
    print('main-queue gaps: %.1f us/step' % (gaps / n / 1e3))
  if len(sys.argv) > 3:   # timeline of the last k steps
    k = int(sys.argv[3])
    a2 = adam[-k - 1]
    tl0 = rows[a2]['e']
    print('--- timeline of the last %d steps (us from the end of the previous Adam) ---' % k)
    for r in rows:
      if r['e'] > tl0 and r['e'] <= t1:
        nm = r['Kernel_Name']
        for pre in ('void ', '(anonymous namespace)::'):
          nm = nm.replace(pre, '')
        print('%s %8.1f -> %8.1f  (%5.1f)  %s' % (
            'M' if (not qkey or r.get(qkey) == main_q) else ' s', (r['s'] - tl0) / 1e3,
            (r['e'] - tl0) / 1e3, (r['e'] - r['s']) / 1e3, nm[:70]))
  d = defaultdict(lambda: [0, 0])
  for r in seg:
    k = (r.get(qkey) == main_q if qkey else True, r['Kernel_Name'][:100])
    d[k][0] += 1; d[k][1] += r['e'] - r['s']
  for (is_main, name), (c, t) in sorted(d.items(), key=lambda kv: (-kv[0][0], -kv[1][1])):
    print('%s %6.2f us/step  x%.2f  avg %6.2f  %s' % (
        'M' if is_main else 's', t / n / 1e3, c / n, t / c / 1e3, name))


if __name__ == '__main__':
  main()
