"""Prints the per-kernel event timings of bench.py JSON lines side by side:
python tools/show_kernels.py a.json [b.json ...]"""
import json, sys
runs = []
for f in sys.argv[1:]:
  d = json.loads([l for l in open(f) if l.startswith('{')][0])
  runs.append((f, d))
names = []
for _, d in runs:
  for k in d['roofline']['per_kernel']:
    if k not in names:
      names.append(k)
print('%-22s' % 'kernel' + ''.join('%12s' % f.split('/')[-1][:11] for f, _ in runs))
for k in names:
  print('%-22s' % k + ''.join('%12s' % ('%.2f' % d['roofline']['per_kernel'][k]['us']
                                         if k in d['roofline']['per_kernel'] else '-') for _, d in runs))
print('%-22s' % 'sum' + ''.join('%12.1f' % d['roofline']['learn_kernels_us'] for _, d in runs))
print('%-22s' % 'steps/s' + ''.join('%12.1f' % d['value'] for _, d in runs))
