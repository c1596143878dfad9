"""Copies the outputs of tools/profile_round.sh from gpurun_out/<tag>/ into profiles/
(python tools/collect_profiles.py r3) and derives the two tables bench.py reads:

  profiles/<tag>_hbm_traffic.json  HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE
      passes, FETCH_SIZE corrected with the factor measured IN THE SAME SESSION on
      tools/micro/stream_micro.bin (known byte counts, dword-per-lane and float4-per-lane
      reads -- MI355X_MICROARCH.md: "calibrate on a known byte count in your own access
      pattern"); WRITE_SIZE is checked against fc1's slab store (exactly 12 582 912 B).
  profiles/<tag>_mfma_util.json    MFMA-pipe utilisation per kernel =
      SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz), durations
      from the kernel trace of the sequential loop in the same session.
"""
import ast
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r6'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, 'gpurun_out', tag), os.path.join(root, 'profiles')

# bench mark name -> substring of the kernel's demangled name
KERNELS = {
    'adam': 'adam_', 'adam+next_sample': 'adam_',
    'fc1_fwd': 'dz_fc_stream_fwd3', 'fc1_dgrad+wgrad': 'fc1_dgrad_mfma_kernel',
    # (round 6: the convolutions on the LDS-DMA kernels; conv2's weight gradient rides with conv1's)
    'conv1_fwd': 'dz_conv1_dma_kernel', 'conv2_fwd': 'dz_conv_dma_fwd_kernel<(anonymous namespace)::ConvDmaCfg<20, 20',
    'conv3_fwd': 'dz_conv_dma_fwd_kernel<(anonymous namespace)::ConvDmaCfg<9, 9', 'fc2_fwd': 'dz_mfma_gemm<FcFwdOp',
    'fc2_wgrad+dgrad': 'fc2_bwd_rows_kernel', 'head_chain': 'rainbow_head_chain_kernel',
    'conv3_wgrad+dgrad': 'ConvWgDmaOp<9, 9', 'conv2_dgrad': 'dz_dmaop_kernel<(anonymous namespace)::ConvDgDmaOp<20, 20',
    'conv_wgrads': 'dz_conv_wgrad3_kernel', 'head_loss': 'rainbow_head_loss_kernel',
    'fc1_epilogue': 'fc_epilogue_kernel',
    'finalize_grads': 'finalize_grads_kernel',
    'sample+gather': 'prioritized_sample_gather_kernel',
}


def last_json_line(path):
  line = [l for l in open(path).read().splitlines() if l.startswith('{')][-1]
  json.loads(line)
  return line


def pmc_csv(path):
  rows = {}
  if os.path.exists(path):
    for r in csv.reader(open(path)):
      if r and r[0] != 'kernel':
        rows[r[0]] = float(r[1]) * 1024.0   # KB -> bytes
  return rows


def find(rows, pat):
  for k, v in rows.items():
    if pat in k:
      return v
  return None


def step_durations(path):
  """kernel-name prefix -> average us, from tools/step_trace_summary.py output."""
  out = {}
  for l in open(path):
    m = re.match(r'^[Ms ]\s+[\d.]+ us/step\s+x[\d.]+\s+avg\s+([\d.]+)\s+(.*)$', l.rstrip())
    if m:
      out[m.group(2)] = float(m.group(1))
  return out


for name, out in (('bench.json', 'bench_line.json'), ('bench_40k.json', 'bench_line_40k_steps.json')):
  open(os.path.join(dst, '%s_%s' % (tag, out)), 'w').write(last_json_line(os.path.join(src, name)) + '\n')
for name in ('kernel_stats_fused.csv', 'kernel_stats_sequential.csv',
             'kernel_step_summary_fused.txt', 'kernel_step_summary_sequential.txt',
             'pmc_sq_rainbow.txt', 'pmc_FETCH_SIZE.csv', 'pmc_WRITE_SIZE.csv',
             'pmc_cal_FETCH_SIZE.csv', 'pmc_cal_WRITE_SIZE.csv',
             'pmc_dqn_FETCH_SIZE.csv', 'pmc_dqn_WRITE_SIZE.csv', 'pmc_double_q_FETCH_SIZE.csv',
             'pmc_double_q_WRITE_SIZE.csv', 'agent_loop_rainbow.json', 'agent_loop_dqn.json',
             'kernel_step_summary_double_q.txt', 'kernel_step_summary_dqn.txt', 'act_decision.txt',
             'head_chain_stamps.txt', 'agent_loop_iqn.json', 'kernel_step_summary_separate_launches.txt',
             'iqn_step_launches.txt', 'pmc_sq_iqn.txt', 'dense_c51_qr_steps.txt', 'kernel_step_summary_a18.txt'):
  p = os.path.join(src, name)
  if os.path.exists(p):
    shutil.copy(p, os.path.join(dst, '%s_%s' % (tag, name)))

# ---- FETCH_SIZE calibration on known byte counts ------------------------------------
fetch = pmc_csv(os.path.join(src, 'pmc_FETCH_SIZE.csv'))
write = pmc_csv(os.path.join(src, 'pmc_WRITE_SIZE.csv'))
cal = pmc_csv(os.path.join(src, 'pmc_cal_FETCH_SIZE.csv'))
known = 2.0 * 2 * 3200 * 1024 * 4   # stream_micro: 52.4 MB per launch
factors = {}
for key, pat in (('dword_per_lane', 'stream_dword'), ('float4_per_lane_tiles', 'stream_x4'),
                 ('float4_flat', 'stream_flat')):
  v = find(cal, pat)
  if v:
    factors[key] = round((known * (0.96 if key == 'float4_per_lane_tiles' else 1.0)) / v, 4)
doc = {
    '_method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (each with '
               '--kernel-trace only) on tools/run_fwd.py (12 full Rainbow learner steps, B=32, '
               'A=6), median per launch (tools/profile_round.sh).  read bytes = FETCH_SIZE x the '
               'factor measured in the same session on tools/micro/stream_micro.bin, whose '
               'kernels read a known 52.4 MB per launch in this library\'s own access patterns '
               '(fetch_calibration below: known bytes / reported FETCH_SIZE; the guide\'s '
               '"exactly 1/2 for wide coalesced reads" would be 2.0); WRITE_SIZE as reported '
               '(fc1\'s slab store is exactly 12 582 912 B algorithmic: write_check).  '
               'Infinity-Cache hits are counted by these counters (guide), so "HBM bytes" means '
               'bytes that crossed the L2\'s memory side.',
    'fetch_calibration': factors, 'kernels': {}}
pattern_of = {'adam': 'float4_flat', 'fc1_fwd': 'dword_per_lane',
              'fc1_dgrad+wgrad': 'float4_flat'}
for key in ('adam', 'fc1_fwd', 'fc1_dgrad+wgrad', 'conv1_fwd', 'conv2_fwd', 'conv3_fwd', 'head_chain',
            'conv3_wgrad+dgrad', 'conv2_dgrad', 'conv_wgrads'):
  f, w = find(fetch, KERNELS[key]), find(write, KERNELS[key])
  if f is None or w is None:
    continue
  fac = factors.get(pattern_of.get(key, 'float4_per_lane_tiles'))
  doc['kernels'][key] = {
      'kernel': KERNELS[key], 'fetch_size_bytes_raw': f, 'write_size_bytes': w,
      'fetch_factor': fac, 'hbm_bytes_corrected': None if fac is None else fac * f + w}
w1 = find(write, KERNELS['fc1_fwd'])
doc['write_check'] = {'fc1_fwd_slab_store_algorithmic': 12582912, 'WRITE_SIZE_reported': w1}
# the dense learners' dominant launch (finalize + RMSProp): float4 streams
doc['dense'] = {}
for cfg in ('dqn', 'double_q'):
  f = find(pmc_csv(os.path.join(src, 'pmc_%s_FETCH_SIZE.csv' % cfg)), 'finalize_grads')
  w = find(pmc_csv(os.path.join(src, 'pmc_%s_WRITE_SIZE.csv' % cfg)), 'finalize_grads')
  fac = factors.get('float4_flat')
  if f is not None and w is not None and fac:
    doc['dense'][cfg] = {'kernel': 'finalize_grads(_sg)_kernel (finalize + RMSProp)',
                         'fetch_size_bytes_raw': f, 'write_size_bytes': w, 'fetch_factor': fac,
                         'hbm_bytes_corrected': fac * f + w}
json.dump(doc, open(os.path.join(dst, '%s_hbm_traffic.json' % tag), 'w'), indent=1)

# ---- MFMA utilisation --------------------------------------------------------------
dur = step_durations(os.path.join(src, 'kernel_step_summary_sequential.txt'))
busy = {}
for l in open(os.path.join(src, 'pmc_sq_rainbow.txt')):
  m = re.match(r'^(.*?) (\{.*\})\s*$', l.rstrip())
  if m and 'SQ_VALU_MFMA_BUSY_CYCLES' in m.group(2):
    busy[m.group(1)] = ast.literal_eval(m.group(2))['SQ_VALU_MFMA_BUSY_CYCLES']
util = {'_method': 'SQ_VALU_MFMA_BUSY_CYCLES (median per launch, tools/pmc_rainbow.sh) / (1024 '
                   'SIMDs x average kernel duration in the sequential loop\'s kernel trace x '
                   '2400 MHz); same session', 'kernels': {}}
for key, pat in KERNELS.items():
  b = next((v for k, v in busy.items() if pat.replace('(anonymous namespace)::', '') in k), None)
  d = next((v for k, v in dur.items() if pat in k), None)
  if b is not None and d and b > 0:
    util['kernels'][key] = round(b / (1024 * d * 2400.0), 3)
json.dump(util, open(os.path.join(dst, '%s_mfma_util.json' % tag), 'w'), indent=1)

d = json.loads(last_json_line(os.path.join(src, 'bench.json')))
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'sustained', d.get('sustained'),
      'x cpu', d.get('speedup_vs_cpu_baseline'))
print({k: (v['value'], v['ms_per_step']) for k, v in d.get('other_configs', {}).items()})
print('roofline', {k: d['roofline'].get(k) for k in ('kernel', 'avg_us', 'achieved', 'frac', 'traffic')})
print('step', d['roofline'].get('step'))
print('calibration', factors, 'write check', doc['write_check'])
print('mfma util', util['kernels'])
print(open(os.path.join(src, 'kernel_step_summary_fused.txt')).read().splitlines()[0])
