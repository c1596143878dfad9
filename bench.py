"""Headline benchmark: Rainbow gradient-steps/sec on MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps 2000 --warmup 200
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic transitions:
  PrioritizedTransitionReplay.sample (sum-tree descent, IS weights, 2x32 state
  gather from the 1M-transition HBM store) -> Rainbow update (3 noisy dueling
  C51 network applies, categorical double-Q loss, backward, global-norm clip,
  Adam) -> priority write-back into the sum tree
(ref: rainbow/agent.py:181-198).  Inputs are resident in HBM before the timed
region.  N > 1 runs N independent replicas (one process per GPU, different
seeds; SURVEY.md 8e "replicas only") and all-reduces episode-style statistics
once over RCCL; `value` is the whole-job steps/s.

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (HIP
event timing of the dominant kernel against its algorithmic bytes/flops) and
`cpu_baseline` (the CPU oracle port timed on the host cores, N=1 only).
"""

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

NUM_ACTIONS = 6            # Pong
NUM_ATOMS = 51
VMAX = 10.0
N_STEP_DISCOUNT = 0.99 ** 3
PEAK_HBM = 8.0e12          # B/s   (MI355X_MICROARCH.md, spec)
PEAK_F32_MFMA = 157.3e12   # FLOP/s (MI355X_MICROARCH.md, f32-in MFMA = vector peak)


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=2000)
  ap.add_argument('--warmup', type=int, default=200)
  ap.add_argument('--capacity', type=int, default=1000000)
  ap.add_argument('--batch', type=int, default=32)
  ap.add_argument('--seed', type=int, default=1)
  ap.add_argument('--cpu-seconds', type=float, default=15.0,
                  help='CPU-baseline time budget (0 disables)')
  ap.add_argument('--prof-steps', type=int, default=100)
  ap.add_argument('--sustain-steps', type=int, default=2000,
                  help='after the timed --steps window, also report the rate over this many '
                       'further steps (`sustained`; 0 disables)')
  ap.add_argument('--other-configs', type=int, default=1,
                  help='also measure BASELINE configs 2 and 3 (DQN + uniform replay; '
                       'double-Q + prioritized) after the headline (0 disables)')
  ap.add_argument('--mode', default='fused', choices=('fused', 'sequential'),
                  help="how the steps are enqueued.  'sequential': sample launch, then the "
                       "learner step with the write-back inside Adam -- what Rainbow._learn "
                       "enqueues per learn period.  'fused' (default): the same operations in "
                       "the same order, but the replay is static between two learner steps "
                       "here, so sample(k+1)+gather(k+1) ride in step k's optimiser launch "
                       "(after write-back(k)): one launch fewer on the dependent chain.  "
                       "Both are bit-identical (tests/test_fused_step_gpu.py); the sequential "
                       "form's rate is reported in the same JSON line as `agent_form`")
  ap.add_argument('--graphs', action='store_true',
                  help='replay the learner launches from hipGraphs (sequential mode; '
                       'measured 3 %% slower than eager launches for this step, kept '
                       'for host-bound callers such as the agent loop)')
  ap.add_argument('--no-graphs', action='store_true', help='(default now; accepted for old scripts)')
  ap.add_argument('--sequential', action='store_true', help="same as --mode sequential")
  ap.add_argument('--fused-sample', action='store_true', help="same as --mode fused")
  ap.add_argument('--prime-steps', type=int, default=-1,
                  help='untimed steps before the warm-up (default: one per ring slot)')
  ap.add_argument('--stored-gradients', action='store_true',
                  help='keep every gradient block in memory (learner.keep_all_grads: the '
                       'stored-gradient form of the step, for A/B against the default, '
                       'which forms fc1\'s weight gradient inside the optimiser)')
  ap.add_argument('--agent-form-steps', type=int, default=1000,
                  help='steps of the sequential form (what Rainbow._learn enqueues) timed '
                       'after the headline window for `agent_form` (0 disables)')
  ap.add_argument('--separate-launches', action='store_true',
                  help='one launch per stage of the head chain (learner.separate_launches) instead '
                       'of the default multi-role launch: same-box A/B')
  ap.add_argument('--agent-loop-frames', type=int, default=6000,
                  help='frames of the whole drop-in loop (parts.run_loop: act -> insert -> learn '
                       'every 4th frame; Rainbow and DQN agents on a synthetic environment) timed '
                       'in THIS run for `agent_loop` (0 disables)')
  return ap.parse_args()


def fill_synthetic(replay, cap, device, seed, discount, priority=None):
  """Synthetic transitions: states from a pool of 256 seeded random
  84x84x4 uint8 frames, a ~ U{0..A-1}, r in {-1,0,1}, discount in {0, `discount`};
  written straight into the HBM store (equivalent to `cap` add() calls)."""
  g = torch.Generator(device=device)
  g.manual_seed(seed)
  pool = torch.randint(0, 256, (256, 84, 84, 4), dtype=torch.uint8,
                       device=device, generator=g)
  chunk = 4096
  done = 0
  while done < cap:
    n = min(chunk, cap - done)
    i1 = torch.randint(0, 256, (n,), device=device, generator=g)
    i2 = torch.randint(0, 256, (n,), device=device, generator=g)
    a = torch.randint(0, NUM_ACTIONS, (n,), device=device, generator=g)
    r = (torch.randint(0, 3, (n,), device=device, generator=g) - 1).double()
    d = torch.randint(0, 2, (n,), device=device, generator=g).double() * discount
    fields = [pool[i1], a, r, d, pool[i2]]
    if priority is None:
      replay.bulk_fill(fields)
    else:
      replay.bulk_fill(fields, priority=priority)
    done += n
  return pool


def build_workload(args, device, seed):
  from dqn_zoo_amd import learner as learner_lib
  from dqn_zoo_amd import networks
  from dqn_zoo_amd import parts
  from dqn_zoo_amd import replay as replay_lib

  cap, b = args.capacity, args.batch
  rs = np.random.RandomState(seed)
  # IS exponent schedule of rainbow/run_atari.py:180-188
  beta = parts.LinearSchedule(begin_t=int(0.02 * cap), end_t=200 * 250000,
                              begin_value=0.4, end_value=1.0)
  replay = replay_lib.PrioritizedTransitionReplay(
      capacity=cap,
      structure=replay_lib.Transition(None, None, None, None, None),
      priority_exponent=0.5, importance_sampling_exponent=beta,
      uniform_sample_probability=1e-3, normalize_weights=True,
      random_state=rs, device=device)
  fill_synthetic(replay, cap, device, seed, N_STEP_DISCOUNT, priority=1.0)
  support = np.linspace(-VMAX, VMAX, NUM_ATOMS).astype(np.float32)
  net = networks.RainbowNetwork(NUM_ACTIONS, support, 0.1)
  learner = learner_lib.RainbowLearner(net, learner_lib.AdamConfig(), b,
                                       seed=seed, device=device)
  learner.keep_all_grads = bool(getattr(args, 'stored_gradients', False))
  learner.separate_launches = bool(getattr(args, 'separate_launches', False))
  torch.cuda.synchronize(device)
  return replay, learner, None


def make_step(replay, learner, batch, fused_write_back=True, fused_next_sample=False):
  """The step: sample -> update -> priority write-back (rainbow/agent.py:181-198),
  as `Rainbow._learn` enqueues it.  fused_write_back=False keeps the write-back
  as its own kernel after the update (the reference's literal order).
  fused_next_sample: the replay is static between steps, so sample(k+1) + gather(k+1)
  ride in step k's optimiser launch (after write-back(k), which moves into an earlier
  backward launch): same operations, same order, one launch fewer on the chain."""
  # Host work of the loop, in program order per step: ENQUEUE the step, then make the host RNG draws
  # and the descriptor of the sample the NEXT step's launch will carry.  (Round 5 prepared the
  # descriptor in front of the enqueue: the same work per step, but after the mandatory
  # synchronize() the first timed step then started ~20 us of NumPy later than it had to.)  The
  # draws are consumed in the reference's order either way: sample k + 1's before sample k + 2's.
  pend = {}

  def step_fused():
    if not pend:
      s = replay.sample_device(batch)
      desc, nxt = replay.prepare_next_sample(batch)
    else:
      s, desc, nxt = pend['s'], pend['desc'], pend['nxt']
      if pend['t'] != replay.insertions:   # (what take_prepared() checks)
        raise RuntimeError('the replay changed between prepare_next_sample and its use')
    t = s.transitions
    learner.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32,
                 priority_sink=replay.priority_sink(s.ids), next_sample=desc)
    # the batch `desc` described exists once this step has run; the one after it is described now
    d2, n2 = replay.prepare_next_sample(batch)
    pend.update(s=nxt, desc=d2, nxt=n2, t=replay.insertions)

  if fused_next_sample:
    return step_fused

  def step():
    s = replay.sample_device(batch)
    t = s.transitions
    if fused_write_back:  # the write-back rides inside the backward launches
      learner.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32,
                   priority_sink=replay.priority_sink(s.ids))
    else:
      learner.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32)
      replay.update_priorities(s.ids, learner.priorities)

  return step


# ---- algorithmic work per kernel (SURVEY.md 8d figures, per launch) ----------
def kernel_work(b, a=NUM_ACTIONS, k=NUM_ATOMS):
  """name -> (flops, bytes) of ONE launch at batch b."""
  na = a * k
  p_ref = 77984 + 2 * (3136 * 512 * 2 + 1024) + (512 * na * 2 + na) + \
      (512 * k * 2 + k)
  g = 3
  f = lambda m, n, kk: 2.0 * m * n * kk
  w = {}
  w['conv1_fwd'] = (f(g * b * 400, 32, 256), g * b * 28224 + g * b * 400 * 32 * 4)
  w['conv1_fwd+noise'] = w['conv1_fwd']   # the noise draw rides in the same launch
  w['conv2_fwd'] = (f(g * b * 81, 64, 512), g * b * (12800 + 5184) * 4)
  w['conv3_fwd'] = (f(g * b * 49, 64, 576), g * b * (5184 + 3136) * 4)
  # noisy layers in the reference's two-GEMM form; weights of each parameter set
  # (online, target) streamed once
  w['fc1_fwd'] = (2 * f(g * b, 1024, 3136), 2 * 2 * 3136 * 1024 * 4)
  w['fc2_fwd'] = (2 * f(g * b, na + k, 512), 2 * 2 * 512 * (na + k) * 4)
  # weight gradient + input gradient of a layer are ONE fused launch
  w['fc1_wgrad'] = (f(3136, 1024, b), 2 * 3136 * 1024 * 4)     # write dW mu,sigma
  w['fc1_dgrad'] = (2 * f(b, 3136, 1024), 2 * 3136 * 1024 * 4)  # read W mu,sigma
  # (the one-call step never stores fc1's weight gradient: this launch is the input gradient
  # against W_eff, depth N, a row-owning weight stream -- csrc/dz_row_dgrad.h)
  w['fc1_dgrad+wgrad'] = (f(b, 3136, 1024), w['fc1_dgrad'][1])
  w['fc2_wgrad+dgrad'] = (f(512, na + k, b) + f(b, 1024, (na + k) / 2.0),
                          2 * 2 * 512 * (na + k) * 4)
  # the multi-role head launch (csrc/dz_head_chain.h): fold of fc1's 32 slabs, noisy fc2 (W_eff
  # form: depth K), loss, fc2 backward.  Bytes: the slabs read once, h1 / fc2 slabs / dlogits
  # written and read back through the seams, fc2's parameters (mu, sigma; online + target) and
  # its gradient -- a latency chain, neither figure is its bound
  w['head_chain'] = (f(g * b, na + k, 512) + f(512, na + k, b) + f(b, 1024, (na + k) / 2.0),
                     32 * g * b * 1024 * 4 + 2 * g * b * 1024 * 4 + 2 * 4 * g * b * (na + k) * 4 +
                     2 * 2 * 512 * (na + k) * 4 + 2 * 512 * (na + k) * 4)
  w['conv3_wgrad+dgrad'] = (f(576, 64, b * 49) + f(b * 81, 64, 576),
                            2 * b * (5184 + 3136) * 4)
  w['conv2_wgrad+dgrad'] = (f(512, 64, b * 81) + f(b * 400, 32, 256),
                            2 * b * (12800 + 5184) * 4)
  w['conv1_wgrad'] = (f(256, 32, b * 400), b * 28224 + b * 12800 * 4)
  # round 6: the conv3 / conv2 launches carry the input gradient only; conv1's and conv2's weight
  # gradients share the chain's last launch (conv3's stays with its input gradient)
  w['conv3_dgrad'] = (f(b * 81, 64, 576), b * (5184 + 3136) * 4)
  w['conv2_dgrad'] = (f(b * 400, 32, 256), b * (12800 + 5184) * 4)
  w['conv_wgrads'] = (f(512, 64, b * 81) + f(256, 32, b * 400),       # conv2's + conv1's (the shipped cut)
                      b * (12800 + 5184) * 4 + b * 28224 + b * 12800 * 4)
  w['conv_wgrads3'] = (f(576, 64, b * 49) + w['conv_wgrads'][0],       # ... + conv3's (DZ_CONV_BWD_SPLIT 1)
                       b * (5184 + 3136) * 4 + w['conv_wgrads'][1])
  # SURVEY.md 8d's figure: read g,p,m,v; write p,m,v.  (The launch itself moves less: it
  # forms fc1's 2 x 3.2 M gradient entries from L2-resident factors instead of reading
  # them -- ADAM_BYTES_MOVED, reported next to `achieved` as `achieved_moved`.)
  w['adam'] = (0.0, 7.0 * p_ref * 4)
  # fused mode: the next step's sample + gather (2 x 32 states read and written) rides along
  w['adam+next_sample'] = (0.0, 7.0 * p_ref * 4 + 2.0 * b * (2 * 28224 + 20))
  w['grad_sumsq'] = (0.0, 1.0 * p_ref * 4)
  return w


def adam_bytes_moved(a=NUM_ACTIONS, k=NUM_ATOMS):
  """HBM bytes the Rainbow optimiser launch actually has to move: p, m, v read and written
  for every parameter, the stored gradient only outside the two fc1 matrices."""
  na = a * k
  p_ref = 77984 + 2 * (3136 * 512 * 2 + 1024) + (512 * na * 2 + na) + (512 * k * 2 + k)
  return 6.0 * p_ref * 4 + (p_ref - 2 * 3136 * 1024) * 4


def dense_kernel_work(b, g, a=NUM_ACTIONS):
  """name -> (flops, bytes) of ONE launch of the dense-head (NatureDQN) learner
  at batch b with g network applies per step (2: DQN; 3: double-Q)."""
  p_ref = 77984 + (3136 * 512 + 512) + (512 * a + a)
  f = lambda m, n, kk: 2.0 * m * n * kk
  sets = 2  # online + target weights are each streamed once
  w = {}
  w['conv1_fwd'] = (f(g * b * 400, 32, 256), g * b * 28224 + g * b * 400 * 32 * 4)
  w['conv2_fwd'] = (f(g * b * 81, 64, 512), g * b * (12800 + 5184) * 4)
  w['conv3_fwd'] = (f(g * b * 49, 64, 576), g * b * (5184 + 3136) * 4)
  w['fc1_fwd'] = (f(g * b, 512, 3136), sets * 3136 * 512 * 4)
  w['head+loss'] = (f(g * b, a, 512), sets * 512 * a * 4 + 32 * g * b * 512 * 4)  # slab sums + second layer + TD loss
  w['fc1_wgrad+dgrad'] = (f(3136, 512, b) + f(b, 3136, 512), 2 * 3136 * 512 * 4)
  w['fc2_wgrad+dgrad'] = (f(512, a, b) + f(b, 512, a), 2 * 512 * a * 4)
  w['conv3_wgrad+dgrad'] = (f(576, 64, b * 49) + f(b * 81, 64, 576),
                            2 * b * (5184 + 3136) * 4)
  w['conv2_wgrad+dgrad'] = (f(512, 64, b * 81) + f(b * 400, 32, 256),
                            2 * b * (12800 + 5184) * 4)
  w['conv1_wgrad'] = (f(256, 32, b * 400), b * 28224 + b * 12800 * 4)
  w['conv3_dgrad'] = (f(b * 81, 64, 576), b * (5184 + 3136) * 4)
  w['conv2_dgrad'] = (f(b * 400, 32, 256), b * (12800 + 5184) * 4)
  w['conv_wgrads'] = (f(512, 64, b * 81) + f(256, 32, b * 400),       # conv2's + conv1's (the shipped cut)
                      b * (12800 + 5184) * 4 + b * 28224 + b * 12800 * 4)
  w['conv_wgrads3'] = (f(576, 64, b * 49) + w['conv_wgrads'][0],       # ... + conv3's (DZ_CONV_BWD_SPLIT 1)
                       b * (5184 + 3136) * 4 + w['conv_wgrads'][1])
  w['rmsprop'] = (0.0, 7.0 * p_ref * 4)  # read g,p,mu,nu; write p,mu,nu
  w['finalize+rmsprop'] = w['rmsprop']   # RMSProp rides in the finalize launch (+ the next sample in fused mode)
  return w


def profile_kernels(step, n):
  """Average per-kernel durations (s) of `step` from the library's HIP events."""
  from dqn_zoo_amd import _lib
  lib = _lib.load()
  lib.dz_prof_enable(1)
  ms = (ctypes.c_float * 96)()
  names = ctypes.create_string_buffer(96 * 32)
  acc = {}
  for _ in range(n):
    step()
    torch.cuda.synchronize()
    k = lib.dz_prof_read(96, ctypes.addressof(ms), ctypes.addressof(names))
    for i in range(k):
      nm = names.raw[32 * i:32 * i + 32].split(b'\0')[0].decode()
      acc.setdefault(nm, []).append(ms[i] * 1e-3)
  lib.dz_prof_enable(0)
  return {k: float(np.mean(v)) for k, v in acc.items()}


def roofline_of(avg, work):
  """Dominant kernel of a step and its roofline fraction (as `roofline`)."""
  dom = max(avg, key=avg.get)
  out = {'kernel': dom, 'avg_us': round(avg[dom] * 1e6, 2), 'traffic': None}
  if dom in work:
    flops, nbytes = work[dom]
    if flops / PEAK_F32_MFMA > nbytes / PEAK_HBM:
      out.update(bound='mfma', achieved=round(flops / avg[dom] / 1e12, 3),
                 peak=PEAK_F32_MFMA / 1e12, unit='TFLOP/s')
    else:
      out.update(bound='hbm', achieved=round(nbytes / avg[dom] / 1e9, 3),
                 peak=PEAK_HBM / 1e9, unit='GB/s')
    out['frac'] = round(out['achieved'] / out['peak'], 4)
  out['learn_kernels_us'] = round(sum(avg.values()) * 1e6, 1)
  return out


def measure_rainbow_actions(args, device, replay, num_actions, steps, warmup, prof_steps):
  """The headline's learner step with another action count (same replay, same loop)."""
  from dqn_zoo_amd import learner as learner_lib
  from dqn_zoo_amd import networks
  support = np.linspace(-VMAX, VMAX, NUM_ATOMS).astype(np.float32)
  ln = learner_lib.RainbowLearner(networks.RainbowNetwork(num_actions, support, 0.1),
                                  learner_lib.AdamConfig(), args.batch, seed=args.seed, device=device)
  ln.use_graphs = False
  fused = args.mode == 'fused'
  st = make_step(replay, ln, args.batch, fused_next_sample=fused)
  for _ in range(max(warmup, replay.SAMPLE_RING_DEPTH)):
    st()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    st()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  replay.check_status()
  ln.check_status()
  e = {'metric': 'gradient-steps/sec (Rainbow, %d actions, batch %d)' % (num_actions, args.batch),
       'value': round(steps / dt, 2), 'unit': 'steps/s', 'steps': steps,
       'ms_per_step': round(1e3 * dt / steps, 4), 'dtype': 'f32',
       'config': {'workload': 'the headline\'s Rainbow learner step with the %d-action head '
                              '(advantage layer 512 -> %d)' % (num_actions, num_actions * NUM_ATOMS),
                  'num_actions': num_actions, 'num_atoms': NUM_ATOMS, 'mode': args.mode,
                  'replay_capacity': args.capacity, 'global_batch': args.batch}}
  if prof_steps > 0:
    avg = profile_kernels(st, prof_steps)
    e['roofline'] = roofline_of(avg, kernel_work(args.batch, a=num_actions))
    e['per_kernel_us'] = {k: round(v * 1e6, 2) for k, v in sorted(avg.items(), key=lambda kv: -kv[1])}
  # the prepared-sample hand-off belongs to this loop's learner: the headline's loop is over
  del st, ln
  torch.cuda.empty_cache()
  return e


def measure_other_configs(args, device, steps, warmup, prof_steps):
  """BASELINE.json configs[1] and configs[2] at full size (1M-transition store in
  HBM, batch 32): the whole sample -> update (-> priority write-back) step, as
  `Dqn._learn` / `PrioritizedDqn._learn` enqueue it (ref: dqn/agent.py:179-189,
  dqn/run_atari.py:201-219; prioritized/agent.py:187-206,
  prioritized/run_atari.py:104-113,234-250).  One store is alive at a time."""
  from dqn_zoo_amd import learner as learner_lib
  from dqn_zoo_amd import networks
  from dqn_zoo_amd import parts
  from dqn_zoo_amd import replay as replay_lib

  cap, b = args.capacity, args.batch
  T = replay_lib.Transition(None, None, None, None, None)
  out = {}
  fused = args.mode == 'fused'   # sample(k+1) rides in step k's optimiser launch (eager)

  def run(name, replay, learner, step, work, desc):
    learner.use_graphs = args.other_graphs and not fused
    for _ in range(max(warmup, replay.SAMPLE_RING_DEPTH)):
      step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    replay.check_status()
    e = {'metric': 'gradient-steps/sec (%s, batch %d)' % (name, b),
         'value': round(steps / dt, 2), 'unit': 'steps/s', 'steps': steps,
         'ms_per_step': round(1e3 * dt / steps, 4),
         'replay_samples_per_sec': round(steps / dt * b, 1), 'dtype': 'f32',
         'config': dict(desc, replay_capacity=cap, global_batch=b,
                        num_actions=NUM_ACTIONS,
                        mode=args.mode,
                        launch='eager' if (fused or not args.other_graphs) else
                               'hipGraph replay (learner) + 1 eager sample launch')}
    if prof_steps > 0:
      learner.use_graphs = False
      e['roofline'] = roofline_of(profile_kernels(step, prof_steps), work)
      # HBM bytes of the dominant launch from the committed PMC passes of this learner
      # (tools/run_dense.py under --pmc FETCH_SIZE / WRITE_SIZE; profiles/r5_hbm_traffic.json)
      doc, _ = _profile_json('hbm_traffic')
      t = (doc or {}).get('dense', {}).get(desc.get('pmc_key'))
      if t and e['roofline'].get('kernel', '').startswith('finalize'):
        e['roofline']['traffic'] = t['hbm_bytes_corrected']
    return e

  # ---- configs[1]: DQN, NatureDQN net, uniform replay ------------------------
  rep = replay_lib.TransitionReplay(cap, T, np.random.RandomState(args.seed),
                                    device=device)
  fill_synthetic(rep, cap, device, args.seed, 0.99)
  ln = learner_lib.DenseLearner(
      networks.DenseNetwork('dqn', NUM_ACTIONS), 'q',
      learner_lib.RmsPropConfig(learning_rate=0.00025, decay=0.95,
                                eps=0.01 / 32 ** 2), b, seed=args.seed,
      device=device)

  primed = [False]

  def step_dqn():
    if not fused:
      t, _ = rep.sample_device(b)
      ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, None)
      return
    t, _ = rep.take_prepared() if primed[0] else rep.sample_device(b)
    desc, _ = rep.prepare_next_sample(b)
    primed[0] = True
    ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, None, next_sample=desc)

  out['dqn_uniform_1m'] = run(
      'DQN + uniform replay', rep, ln, step_dqn, dense_kernel_work(b, 2),
      {'workload': 'dqn learner step: uniform sample (positions -> ids -> gather, '
                   'one launch) + 2x NatureDQN apply + Q-learning loss + backward '
                   '+ centred RMSProp', 'baseline_config': 1, 'pmc_key': 'dqn'})
  del rep, ln, step_dqn
  torch.cuda.empty_cache()

  # ---- configs[2]: double-Q + prioritized replay (exponent 0.6) ---------------
  beta = parts.LinearSchedule(begin_t=int(0.05 * cap), end_t=200 * 250000,
                              begin_value=0.4, end_value=1.0)
  rep = replay_lib.PrioritizedTransitionReplay(
      cap, T, priority_exponent=0.6, importance_sampling_exponent=beta,
      uniform_sample_probability=1e-3, normalize_weights=True,
      random_state=np.random.RandomState(args.seed), device=device)
  fill_synthetic(rep, cap, device, args.seed, 0.99, priority=1.0)
  ln = learner_lib.DenseLearner(
      networks.DenseNetwork('double_dqn', NUM_ACTIONS), 'double_q',
      learner_lib.RmsPropConfig(learning_rate=0.00025 / 4, decay=0.95,
                                eps=(0.01 / 32 ** 2) * (1.0 / 4) ** 2), b,
      seed=args.seed, device=device)

  primed2 = [False]

  def step_prio():
    sm = rep.take_prepared() if (fused and primed2[0]) else rep.sample_device(b)
    t = sm.transitions
    desc = None
    if fused:
      desc, _ = rep.prepare_next_sample(b)
      primed2[0] = True
    ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, sm.weights32,
            priority_sink=rep.priority_sink(sm.ids), next_sample=desc)

  out['double_q_prioritized_1m'] = run(
      'double-Q + prioritized replay', rep, ln, step_prio, dense_kernel_work(b, 3),
      {'workload': 'prioritized-DQN learner step: sum-tree sample (exponent 0.6) + '
                   'IS weights + gather (one launch) + 3x NatureDQN apply (shared '
                   'bias) + double-Q loss + backward (+ |td| priority write-back as '
                   'a side block) + centred RMSProp', 'baseline_config': 2, 'pmc_key': 'double_q'})
  del rep, ln
  torch.cuda.empty_cache()
  return out


def _profile_json(name):
  """A committed profile table (profiles/r6_<name>.json, else the latest earlier round's)."""
  for tag in ('r6', 'r5', 'r4', 'r3', 'r2'):
    try:
      with open(os.path.join(ROOT, 'profiles', '%s_%s.json' % (tag, name))) as f:
        return json.load(f), tag
    except (OSError, ValueError):
      continue
  return None, None


def pmc_traffic(kernel):
  """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
  (profiles/r5_hbm_traffic.json: FETCH_SIZE / WRITE_SIZE collected in separate passes and
  corrected as MI355X_MICROARCH.md prescribes, the dword-operand kernels with a factor
  calibrated on a known byte count in the same access pattern); None if not collected."""
  doc, _ = _profile_json('hbm_traffic')
  if not doc:
    return None
  k = doc.get('kernels', {}).get(kernel.split('+next_sample')[0])
  return None if k is None else k.get('hbm_bytes_corrected')


def pmc_mfma_util():
  """mark name -> MFMA-pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x
  kernel cycles)) from the committed SQ counter pass (profiles/r5_mfma_util.json)."""
  doc, tag = _profile_json('mfma_util')
  return ({}, None) if not doc else (doc.get('kernels', {}), tag)


def measure_roofline(step, prof_steps, batch):
  """Per-kernel average durations from HIP events recorded on the launch
  stream (dz_prof_*), then the roofline fraction of the dominant kernel."""
  from dqn_zoo_amd import _lib
  lib = _lib.load()
  lib.dz_prof_enable(1)
  ms = (ctypes.c_float * 96)()
  names = ctypes.create_string_buffer(96 * 32)
  acc = {}
  for _ in range(prof_steps):
    step()
    torch.cuda.synchronize()
    n = lib.dz_prof_read(96, ctypes.addressof(ms), ctypes.addressof(names))
    for i in range(n):
      nm = names.raw[32 * i:32 * i + 32].split(b'\0')[0].decode()
      acc.setdefault(nm, []).append(ms[i] * 1e-3)
  lib.dz_prof_enable(0)
  avg = {k: float(np.mean(v)) for k, v in acc.items()}
  work = kernel_work(batch)
  dom = max(avg, key=avg.get)
  out = {'kernel': dom, 'avg_us': round(avg[dom] * 1e6, 2),
         'traffic': pmc_traffic(dom)}
  if dom in work:
    flops, nbytes = work[dom]
    t_m, t_h = flops / PEAK_F32_MFMA, nbytes / PEAK_HBM
    if t_m > t_h:
      out.update(bound='mfma', achieved=flops / avg[dom] / 1e12,
                 peak=PEAK_F32_MFMA / 1e12, unit='TFLOP/s')
    else:
      out.update(bound='hbm', achieved=nbytes / avg[dom] / 1e9,
                 peak=PEAK_HBM / 1e9, unit='GB/s')
    out['frac'] = out['achieved'] / out['peak']
    out['achieved'] = round(out['achieved'], 3)
    out['frac'] = round(out['frac'], 4)
    if dom.startswith('adam'):
      moved = adam_bytes_moved() + (nbytes - work['adam'][1])
      out['bytes_moved'] = moved
      out['achieved_moved'] = round(moved / avg[dom] / 1e9, 3)
      out['note'] = ('achieved = SURVEY 8d algorithmic bytes (7 words per parameter) / time; '
                     'achieved_moved = the bytes this launch has to move (fc1\'s weight gradient '
                     'is formed in the launch, not read) / time')
  util, util_tag = pmc_mfma_util()
  table = {}
  for k2, v in sorted(avg.items(), key=lambda kv: -kv[1]):
    e = {'us': round(v * 1e6, 2)}
    if k2 in work:
      fl, nb = work[k2]
      e['tflops'] = round(fl / v / 1e12, 2)
      e['gbps'] = round(nb / v / 1e9, 1)
    u = util.get(k2.split('+next_sample')[0].split('+noise')[0])
    if u is not None:
      e['mfma_util'] = u     # from the committed SQ counter pass, not from this run
    t = pmc_traffic(k2)
    if t is not None:
      e['hbm_bytes_pmc'] = t
      if k2 in work and work[k2][1]:
        e['traffic_over_algorithmic'] = round(t / work[k2][1], 2)
    table[k2] = e
  out['per_kernel'] = table
  out['mfma_util_source'] = None if util_tag is None else 'profiles/%s_mfma_util.json' % util_tag
  out['learn_kernels_us'] = round(sum(avg.values()) * 1e6, 1)
  out['launches'] = len(avg)
  return out


def step_roofline(roofline, ms_per_step, batch, extra_launches=0):
  """The whole step against the chip: 4.438 GFLOP of fp32 matrix work and
  194 MB of compulsory traffic per Rainbow gradient step at batch 32."""
  flops = batch * 138700800.0          # 3 applies + backward (two-GEMM noisy form)
  p_ref = 6868485
  nbytes = 7.0 * p_ref * 4 + batch * 2 * 28224
  sec = ms_per_step * 1e-3
  return {'flops': flops, 'bytes': nbytes,
          'frac_mfma': round(flops / sec / PEAK_F32_MFMA, 4),
          'frac_hbm': round(nbytes / sec / PEAK_HBM, 4),
          'busy_us': roofline['learn_kernels_us'],   # HIP-event sum (adds ~2-3 us per launch)
          'launches': roofline['launches'] + extra_launches,
          'floor_us': round(max(flops / PEAK_F32_MFMA, nbytes / PEAK_HBM) * 1e6, 1),
          'note': 'frac_* = algorithmic work / measured step time / chip peak; the step is a '
                  'chain of latency-bound launches (DESIGN.md 4), only Adam is near a roofline'}


def measure_replay(replay, learner, batch, n=100):
  """Device time of the replay kernels from HIP event pairs recorded INSIDE the
  C entry points (dz_prof_read_replay), medians over n rounds (SURVEY.md 8d:
  gather against HBM bandwidth; sum-tree sample / update as latency per query --
  their bytes are trivial).  In the measured step sample+gather are ONE launch and
  the write-back rides inside a backward launch; here each is also timed alone."""
  from dqn_zoo_amd import _lib
  lib = _lib.load()
  lib.dz_prof_enable(1)
  ms = (ctypes.c_float * 3)()
  stream = torch.cuda.current_stream(replay._device).cuda_stream  # pylint: disable=protected-access
  fused, gather, update = [], [], []
  for _ in range(n):
    s = replay.sample_device(batch)                       # pair 0: sample + gather launch
    replay._ring.gather(s.ids, batch, stream)             # pair 1: the gather alone  # pylint: disable=protected-access
    replay.update_priorities(s.ids, learner.priorities)   # pair 2: the write-back alone
    torch.cuda.synchronize()
    lib.dz_prof_read_replay(ctypes.addressof(ms))
    fused.append(ms[0] * 1e3); gather.append(ms[1] * 1e3); update.append(ms[2] * 1e3)
  lib.dz_prof_enable(0)
  t_fused, t_gather, t_update = (float(np.median(a)) for a in (fused, gather, update))
  gather_bytes = 2 * batch * (2 * 28224 + 4 + 8 + 8)  # read + written
  return {
      'sample_plus_gather_us': round(t_fused, 2), 'gather_alone_us': round(t_gather, 2),
      'sumtree_update_us': round(t_update, 2),
      'sumtree_sample_ns_per_query': round(1e3 * t_fused / batch, 1),
      'sumtree_update_ns_per_leaf': round(1e3 * t_update / batch, 1),
      'gather_GBps': round(gather_bytes / t_gather / 1e3, 1),
      'gather_frac_of_hbm_peak': round(gather_bytes / (t_gather * 1e-6) / PEAK_HBM, 4),
      'note': 'event pairs include ~2-3 us of event overhead each; sample+gather is one '
              'launch whose gather blocks re-derive their tree index (20 dependent loads, '
              'capacity 1e6): dependent-load latency, not bytes'}


class SyntheticFrames:
  """A pre-processed environment (84x84x4 uint8 observations from a seeded pool, episodes of
  `n` frames): what parts.run_loop drives the agents on in `agent_loop`."""

  def __init__(self, seed, n=1000):
    rs = np.random.RandomState(seed)
    self.pool = rs.randint(0, 256, (64, 84, 84, 4)).astype(np.uint8)
    self.rs, self.n, self.t = rs, n, 0

  def _obs(self):
    return self.pool[self.rs.randint(64)]

  def reset(self):
    from dqn_zoo_amd import dm_env_shim as dm_env
    self.t = 0
    return dm_env.restart(self._obs())

  def step(self, action):
    from dqn_zoo_amd import dm_env_shim as dm_env
    self.t += 1
    r = float(self.rs.randint(-1, 2))
    if self.t == self.n:
      return dm_env.termination(r, self._obs())
    return dm_env.transition(r, self._obs(), 0.99)


def make_loop_agent(which, learn_period=4, capacity=100000):
  """The drop-in agents exactly as a run script builds them (rainbow/run_atari.py:196-262,
  dqn/run_atari.py:180-236), on an identity preprocessor."""
  from dqn_zoo_amd import learner as learner_lib, networks, parts, processors
  from dqn_zoo_amd import replay as replay_lib
  structure = replay_lib.Transition(None, None, None, None, None)
  if which == 'dqn':
    from dqn_zoo_amd.dqn import agent as dqn_lib
    rep = replay_lib.TransitionReplay(capacity, structure, np.random.RandomState(1))
    ag = dqn_lib.Dqn(
        preprocessor=processors.Identity(), sample_network_input=np.zeros((84, 84, 4), np.uint8),
        network=networks.DenseNetwork('dqn', NUM_ACTIONS), optimizer=learner_lib.RmsPropConfig(),
        transition_accumulator=replay_lib.TransitionAccumulator(), replay=rep, batch_size=32,
        exploration_epsilon=lambda t: 0.1, min_replay_capacity_fraction=0.005,
        learn_period=learn_period, target_network_update_period=2000, rng_key=1,
        grad_error_bound=1.0 / 32)
    return ag, rep
  if which == 'iqn':   # iqn/run_atari.py:97-100, 170-240: the reference's defaults, 64 / 64 / 64 tau samples, Adam
    from dqn_zoo_amd.iqn import agent as iqn_lib
    rep = replay_lib.TransitionReplay(capacity, structure, np.random.RandomState(1))
    ag = iqn_lib.Iqn(
        preprocessor=processors.Identity(),
        sample_network_input=iqn_lib.IqnInputs(state=np.zeros((84, 84, 4), np.uint8),
                                               taus=np.zeros(1, np.float32)),
        network=networks.IqnNetwork(NUM_ACTIONS, 64),
        optimizer=learner_lib.AdamConfig(learning_rate=5e-5, eps=0.01 / 32, max_global_grad_norm=0.0),
        transition_accumulator=replay_lib.TransitionAccumulator(), replay=rep, batch_size=32,
        exploration_epsilon=lambda t: 0.1, min_replay_capacity_fraction=0.005,
        learn_period=learn_period, target_network_update_period=2000, huber_param=1.0,
        tau_samples_policy=64, tau_samples_s_tm1=64, tau_samples_s_t=64, rng_key=1)
    return ag, rep
  from dqn_zoo_amd.rainbow import agent as rainbow_lib
  support = np.linspace(-VMAX, VMAX, NUM_ATOMS).astype(np.float32)
  rep = replay_lib.PrioritizedTransitionReplay(
      capacity, structure, 0.5,
      parts.LinearSchedule(begin_t=2000, end_t=10 ** 7, begin_value=0.4, end_value=1.0),
      1e-3, True, np.random.RandomState(1))
  ag = rainbow_lib.Rainbow(
      preprocessor=processors.Identity(), sample_network_input=np.zeros((84, 84, 4), np.uint8),
      network=networks.RainbowNetwork(NUM_ACTIONS, support, 0.1), support=support,
      optimizer=learner_lib.AdamConfig(),
      transition_accumulator=replay_lib.NStepTransitionAccumulator(3), replay=rep,
      batch_size=32, min_replay_capacity_fraction=0.005, learn_period=learn_period,
      target_network_update_period=2000, rng_key=1)
  return ag, rep


def measure_agent_loop(which, frames, learn_period=4, warm_frames=1000, setup=None, on_warm=None):
  """Agent steps/s of the whole drop-in loop (ref: parts.py:70-122 run_loop driving
  rainbow/agent.py:135-160 / dqn/agent.py:133-160): per frame the acting decision is awaited, the
  transition inserted, and every `learn_period`-th frame a batch sampled and learned from."""
  from dqn_zoo_amd import parts
  ag, rep = make_loop_agent(which, learn_period)
  if setup is not None:
    setup(ag, rep)
  loop = parts.run_loop(ag, SyntheticFrames(3), max_steps_per_episode=0)
  for _ in range(warm_frames):   # past min replay; graphs captured, ring slots touched
    next(loop)
  torch.cuda.synchronize()
  if on_warm is not None:
    on_warm()
  t0 = time.perf_counter()
  for _ in range(frames):
    next(loop)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  rep.check_status()
  out = {'agent': which, 'measured_in_run': True, 'frames': frames,
         'agent_steps_per_sec': round(frames / dt, 1),
         'us_per_agent_step': round(1e6 * dt / frames, 2), 'learn_period': learn_period,
         'learner_steps_per_sec': round(frames / dt / learn_period, 1)}
  del loop, ag, rep
  torch.cuda.empty_cache()
  return out


def usable_cpus():
  """Host cores this process may actually use (affinity mask and cgroup CPU
  quota), which can be far fewer than os.cpu_count() inside a container."""
  n = os.cpu_count() or 1
  try:
    n = min(n, len(os.sched_getaffinity(0)))
  except (AttributeError, OSError):
    pass
  for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try:
      txt = open(path).read().split()
      if path.endswith('cpu.max'):
        if txt[0] != 'max':
          n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
      else:
        q = int(txt[0])
        if q > 0:
          per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
          n = min(n, max(1, q // per))
    except (OSError, ValueError, IndexError):
      pass
  return n


def cpu_baseline(args, seed, budget_s):
  """The CPU oracle port of the same step (reference replay algorithms in
  Python + torch-CPU update on all host cores), on a bounded sample."""
  from oracle import qnet_oracle as qo
  from oracle import qnet_torch_cpu
  from oracle import replay_oracle as ro
  from dqn_zoo_amd import parts
  from dqn_zoo_amd import replay as replay_lib

  cap, b = args.capacity, args.batch
  rs = np.random.RandomState(seed)
  beta = parts.LinearSchedule(begin_t=int(0.02 * cap), end_t=200 * 250000,
                              begin_value=0.4, end_value=1.0)
  T = replay_lib.Transition
  rep = ro.PrioritizedReplayOracle(cap, T(None, None, None, None, None), 0.5,
                                   beta, 1e-3, True, rs)
  frs = np.random.RandomState(seed + 7)
  pool = frs.randint(0, 256, (256, 84, 84, 4)).astype(np.uint8)
  i1 = frs.randint(0, 256, cap)
  i2 = frs.randint(0, 256, cap)
  acts = frs.randint(0, NUM_ACTIONS, cap)
  rew = frs.randint(0, 3, cap) - 1.0
  dis = frs.randint(0, 2, cap) * N_STEP_DISCOUNT
  rep.bulk_fill(cap, lambda i: T(pool[i1[i]], int(acts[i]), float(rew[i]),
                                 float(dis[i]), pool[i2[i]]), 1.0)
  support = np.linspace(-VMAX, VMAX, NUM_ATOMS).astype(np.float32)
  prs = np.random.RandomState(seed)
  params = qo.init_params('rainbow', NUM_ACTIONS, prs)
  port = qnet_torch_cpu.RainbowTorchCpu(params, params, support, NUM_ACTIONS)
  nrs = np.random.RandomState(seed + 3)

  def one():
    tr, ids, w = rep.sample(b)
    noises = [qo.sample_noise(nrs, NUM_ACTIONS) for _ in range(3)]
    out = port.update((tr.s_tm1, tr.a_tm1, tr.r_t, tr.discount_t, tr.s_t), w,
                      noises)
    rep.update_priorities(ids, out['priorities'])

  # Thread count: "all cores" is not the fastest setting for batch-32 layers
  # (and a container's CPU quota can be far below os.cpu_count()), so probe a
  # few counts and keep the best: the baseline gets its most favourable setting.
  ncpu = usable_cpus()
  best_threads, best_dt = 1, float('inf')
  for nt in sorted({max(1, min(ncpu, c)) for c in (4, 8, 16, 32, 64)}):
    torch.set_num_threads(nt)
    one()  # warm-up at this setting
    t0 = time.perf_counter()
    one()
    one()
    dt = (time.perf_counter() - t0) / 2
    if dt < best_dt:
      best_threads, best_dt = nt, dt
  torch.set_num_threads(best_threads)
  n, t0 = 0, time.perf_counter()
  while time.perf_counter() - t0 < budget_s:
    one()
    n += 1
  dt = time.perf_counter() - t0
  return {'value': round(n / dt, 2), 'unit': 'steps/s',
          'cores': int(torch.get_num_threads()), 'kind': 'port',
          'replay': 'oracle port (oracle/replay_oracle.py: the reference\'s algorithms '
                    'restated; /root/reference itself does not travel to the GPU box)',
          'update': 'torch-CPU fp32 port (oracle/qnet_torch_cpu.py); JAX/XLA-CPU is not '
                    'installable here',
          'host_cpus_usable': ncpu, 'host_cpus_total': os.cpu_count(),
          'sample': '%d Rainbow steps (oracle replay sample + torch-CPU update '
                    '+ priority write-back), capacity %d, batch %d, %.1f s' %
                    (n, cap, b, dt)}


def main():
  args = parse_args()
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X; there is no CPU fallback')
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  # `bench.py --gpus N` started by hand (not under torchrun): spawn the ranks
  # ourselves, degrading to the GPUs this box has (gpurun boxes have one)
  if args.gpus > 1 and 'RANK' not in os.environ:
    have = torch.cuda.device_count()
    n = min(args.gpus, have)
    if n < args.gpus:
      print('bench.py: --gpus %d requested, %d visible: running %d replica(s)' %
            (args.gpus, have, n), file=sys.stderr)
    if n > 1:
      import socket
      sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
      argv = [a for a in sys.argv[1:]]
      for i, a in enumerate(argv):
        if a == '--gpus':
          argv[i + 1] = str(n)
        elif a.startswith('--gpus='):
          argv[i] = '--gpus=%d' % n
      os.execv(sys.executable, [
          sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
          '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
          '--master-port', str(port), os.path.abspath(__file__)] + argv)
    args.gpus = n
  dist = None
  if 'RANK' in os.environ:  # under torchrun, world 1 included: RCCL is exercised
    import torch.distributed as dist  # RCCL (backend "nccl" on ROCm)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    # NCCL_DEBUG=VERSION (set in this image) makes RCCL print a version banner on
    # STDOUT; stdout carries the one JSON line only
    if os.environ.get('NCCL_DEBUG', '').upper() == 'VERSION':
      os.environ['NCCL_DEBUG'] = 'NONE'
    dist.init_process_group('nccl', rank=rank, world_size=world,
                            device_id=device)

  from dqn_zoo_amd import distributed as dz_dist
  # one host thread per replica, pinned to its own slice of the node's cores
  local_world = int(os.environ.get('LOCAL_WORLD_SIZE', str(world)))
  cpus = dz_dist.pin_rank(local_rank, local_world)
  replay, learner, _ = build_workload(args, device, dz_dist.replica_seed(args.seed, rank))
  # hipGraph capture is illegal on the legacy default stream: everything from
  # here on runs on an explicit stream.
  torch.cuda.synchronize(device)
  torch.cuda.set_stream(torch.cuda.Stream(device))
  if args.sequential:
    args.mode = 'sequential'
  if args.fused_sample:
    args.mode = 'fused'
  args.other_graphs = not args.no_graphs   # configs 2/3 keep hipGraph replay unless --no-graphs
  args.no_graphs = not args.graphs or args.mode == 'fused'
  learner.use_graphs = not args.no_graphs
  step = make_step(replay, learner, args.batch, fused_next_sample=args.mode == 'fused')
  seq_step = make_step(replay, learner, args.batch)

  # ---- setup that is NOT a step, all of it before the clock starts ----------
  # (round 1 had these inside the timed window: at the driver's --steps 20 the
  # first-use costs of torch.distributed / ReplicaStats / the .cpu() reads were
  # 75 % of the measurement)
  # the statistics path once, end to end (allocations, first-use kernels, the
  # all-reduce's communicator set-up, the two device->host reads)
  dry = dz_dist.ReplicaStats(device)
  dry.add(grad_steps=0, loss_sum=learner.losses.double().sum())
  dry.all_reduce()
  stats = dz_dist.ReplicaStats(device)
  torch.cuda.synchronize()
  # Untimed priming steps, directly in front of the warm-up (nothing idles the GPU in
  # between): every ring slot / graph is touched, and the chip reaches its steady state
  # -- the first ~30 steps after an idle period run ~4 % slower each (clock ramp-up; the
  # per-step spans of a kernel trace fall from 172 to 163 us), which a 20-step window
  # would otherwise measure instead of the step.  `sustained` reports the long-run rate.
  prime = max(replay.SAMPLE_RING_DEPTH, 64)
  if args.prime_steps >= 0:
    prime = args.prime_steps
  for _ in range(prime):
    step()

  for _ in range(args.warmup):
    step()
  torch.cuda.synchronize()
  if dist is not None:
    dist.barrier()
  torch.cuda.synchronize()
  # ---- the timed region: EXACTLY args.steps steps ---------------------------
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  dt_host = time.perf_counter() - t0   # host enqueue time (the GPU may still be running)
  torch.cuda.synchronize()
  if dist is not None:
    dist.barrier()
    torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  # statistics boundary (not a gradient step, outside the clock): one RCCL
  # all-reduce of packed sums over xGMI (keys of
  # EpisodeTracker/StepRateTracker, parts.py:239-284)
  totals = dz_dist.reduce_run(stats, dt, args.steps, learner.losses.double().sum(), device)
  dt = totals['seconds_max']   # MAX over ranks
  replay.check_status()
  assert int(totals['grad_steps']) == world * args.steps and int(totals['replicas']) == world

  if rank == 0:
    value = world * args.steps / dt
    out = {
        'metric': 'gradient-steps/sec (Rainbow, batch %d)' % args.batch,
        'value': round(value, 2), 'unit': 'steps/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(1e3 * dt / args.steps, 4),
        'host_enqueue_ms_per_step': round(1e3 * dt_host / args.steps, 4),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'replay_samples_per_sec': round(value * args.batch, 1),
        'config': {
            'workload': 'rainbow learner step: prioritized sum-tree sample + '
                        'gather + 3x noisy dueling C51 apply + double-Q loss + '
                        'backward (+ priority write-back as a side block) + clip/Adam'
                        + (' (+ the next step\'s sample + gather as side blocks)'
                           if args.mode == 'fused' else ''),
            'replay_capacity': args.capacity, 'global_batch': args.batch * world,
            'state': '84x84x4 uint8', 'num_actions': NUM_ACTIONS,
            'num_atoms': NUM_ATOMS, 'parallelism': 'replicas x%d' % world,
            'launch': 'eager' if args.no_graphs else 'hipGraph replay',
            'collective': 'rccl' if dist is not None else 'none (single process)',
            'host_cpus_rank0': len(cpus),
            'untimed_setup_steps': prime,
            'mode': args.mode, 'streams': 'one'},
    }
    if args.sustain_steps > 0:
      # the same loop over a window long enough that pipeline fill and host jitter do
      # not matter (the contract's `value` above is EXACTLY --steps steps)
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      for _ in range(args.sustain_steps):
        step()
      torch.cuda.synchronize()
      ds = time.perf_counter() - t1
      out['sustained'] = {'steps': args.sustain_steps,
                          'value': round(args.sustain_steps / ds, 2),
                          'ms_per_step': round(1e3 * ds / args.sustain_steps, 4)}
    if args.agent_form_steps > 0:
      # The form the drop-in agent enqueues per learn period (Rainbow._learn: a sample +
      # gather launch, then the learner step with the write-back inside it) -- transitions
      # are inserted between two learner steps there, so the next sample cannot ride in this
      # step's optimiser launch.  Same loop, same store, same process.
      for _ in range(32):
        seq_step()
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      for _ in range(args.agent_form_steps):
        seq_step()
      torch.cuda.synchronize()
      ds = time.perf_counter() - t1
      out['agent_form'] = {
          'what': 'sequential form of the same step, as Rainbow._learn enqueues it: '
                  'dz_prioritized_sample_gather launch + dz_rainbow_learn',
          'steps': args.agent_form_steps, 'value': round(args.agent_form_steps / ds, 2),
          'unit': 'steps/s', 'ms_per_step': round(1e3 * ds / args.agent_form_steps, 4),
          # counted from the library's event marks of this form + the sample + gather launch
          'launches': len(profile_kernels(seq_step, 2)) + 1}
    if args.prof_steps > 0:
      learner.use_graphs = False  # per-kernel events need eager launches
      prof_step = step if args.mode == 'fused' else seq_step
      out['roofline'] = measure_roofline(prof_step, args.prof_steps, args.batch)
      out['roofline']['step'] = step_roofline(
          out['roofline'], out['ms_per_step'], args.batch,
          extra_launches=0 if args.mode == 'fused' else 1)   # + the sample launch
      # the write-back timed as its own kernel (in the measured step it rides
      # inside a backward launch)
      out['replay'] = measure_replay(replay, learner, args.batch)
    a18 = None
    if world == 1 and args.other_configs:
      # the headline's step with the 18-action head of the full Atari action set (rainbow/
      # run_atari.py:145 takes num_actions from the environment; a large part of the 57 games),
      # on the SAME store, before it is released
      a18 = measure_rainbow_actions(args, device, replay, 18, steps=max(args.steps, 1000),
                                    warmup=max(args.warmup, 50), prof_steps=min(args.prof_steps, 20))
    if world == 1 and args.other_configs:
      # BASELINE configs 2 and 3 (the headline's store is released first)
      del step, seq_step, replay, learner
      torch.cuda.empty_cache()
      out['other_configs'] = measure_other_configs(
          args, device, steps=max(args.steps, 1000), warmup=max(args.warmup, 50),
          prof_steps=min(args.prof_steps, 20))
      out['other_configs']['rainbow_a18'] = a18
    if world == 1 and args.agent_loop_frames > 0:
      # the whole drop-in loop on this run's clock (the headline's store is gone by now)
      if 'step' in dir():
        del step, seq_step, replay, learner
        torch.cuda.empty_cache()
      out['agent_loop'] = {w: measure_agent_loop(w, args.agent_loop_frames)
                           for w in ('rainbow', 'dqn', 'iqn')}
    if world == 1 and args.cpu_seconds > 0:
      out['cpu_baseline'] = cpu_baseline(args, args.seed, args.cpu_seconds)
      out['speedup_vs_cpu_baseline'] = round(
          value / out['cpu_baseline']['value'], 1)
    try:  # anything a C library buffered on stdout goes out BEFORE the JSON line
      ctypes.CDLL(None).fflush(None)
    except OSError:
      pass
    print(json.dumps(out), flush=True)
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
