"""Copies the outputs of tools/profile_round.sh from gpurun_out/ into profiles/
(python tools/collect_profiles.py r2): the bench line, the kernel-trace stats and
per-step summary, the FETCH_SIZE / WRITE_SIZE medians merged into one table, and the
SQ counter summary.  profiles/<tag>_hbm_traffic.json is derived from the merged table
(the rows bench.py reads for roofline.traffic)."""
import csv, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r2'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out'), os.path.join(root, 'profiles')
line = open(os.path.join(src, 'bench_%s.json' % tag)).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(dst, '%s_bench_line.json' % tag), 'w').write(line + '\n')
for name in ('kernel_stats.csv', 'kernel_step_summary.txt', 'pmc_sq_rainbow.txt'):
  shutil.copy(os.path.join(src, '%s_%s' % (tag, name)), os.path.join(dst, '%s_%s' % (tag, name)))
rows = {}
for col, c in enumerate(('FETCH_SIZE', 'WRITE_SIZE')):
  for r in csv.reader(open(os.path.join(src, '%s_pmc_%s.csv' % (tag, c)))):
    if r[0] == 'kernel':
      continue
    rows.setdefault(r[0], [None, None])[col] = float(r[1])
with open(os.path.join(dst, '%s_pmc_fetch_write.csv' % tag), 'w') as f:
  f.write('kernel,FETCH_SIZE_KB,WRITE_SIZE_KB\n')
  for k, (a, b) in sorted(rows.items(), key=lambda kv: -((kv[1][0] or 0) + (kv[1][1] or 0))):
    f.write('"%s",%s,%s\n' % (k, '' if a is None else '%.1f' % a, '' if b is None else '%.1f' % b))
path = os.path.join(dst, '%s_hbm_traffic.json' % tag)
doc = json.load(open(path))
pick = {'adam': 'adam_kernel', 'fc1_fwd': 'dz_fc_stream_fwd3', 'fc1_dgrad+wgrad': 'FcWgradOp<2, 2, 1, 2, 5>, FcDgradOp'}
for key, pat in pick.items():
  k = next(k for k in rows if pat in k)
  fetch, write = rows[k][0] * 1024, rows[k][1] * 1024
  e = doc['kernels'][key]
  e['fetch_size_bytes_raw'] = fetch; e['write_size_bytes'] = write
  e['hbm_bytes_corrected'] = 2 * fetch + write if key == 'adam' else None
json.dump(doc, open(path, 'w'), indent=1)
d = json.loads(line)
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'x cpu', d['value'] / d['cpu_baseline']['value'])
print({k: (v['value'], v['ms_per_step']) for k, v in d.get('other_configs', {}).items()})
print('roofline', {k: d['roofline'][k] for k in ('kernel', 'avg_us', 'achieved', 'frac', 'traffic')})
print(open(os.path.join(dst, '%s_kernel_step_summary.txt' % tag)).read().splitlines()[1])
