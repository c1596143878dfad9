"""The two-stream, software-pipelined form of the step (dqn_zoo_amd/pipeline.py:
write-back(k), sample(k+1) and the TARGET network's apply for batch k+1 on a side
stream under backward(k)/Adam(k); two online applies on the main stream) and the
hipGraph form must be BIT-IDENTICAL to the sequential single-stream three-apply
step: same sampled ids, same losses, same parameters, same tree."""

import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(pipelined, steps=12, graphs=False, sync_every_step=True, sync_target_at=(),
         fused_sample=False, capacity=2048, stored_gradient=False):
  import bench
  args = types.SimpleNamespace(capacity=capacity, batch=32)
  dev = torch.device('cuda', 0)
  replay, learner, _ = bench.build_workload(args, dev, seed=3)
  learner.use_graphs = graphs
  learner.keep_all_grads = stored_gradient
  torch.cuda.synchronize()
  prev = torch.cuda.current_stream(dev)
  torch.cuda.set_stream(torch.cuda.Stream(dev))
  step = (bench.make_step_pipelined(replay, learner, 32, dev) if pipelined
          else bench.make_step(replay, learner, 32, fused_next_sample=fused_sample))
  losses, ids = [], []
  for k in range(steps):
    if k in sync_target_at:   # target <- online between two steps
      (step.loop if pipelined else learner).sync_target()
    if pipelined:
      s = step()
      # (the loss buffer is rewritten by the next step: read it before enqueueing more)
      if sync_every_step:
        torch.cuda.synchronize()
      else:
        torch.cuda.current_stream(dev).synchronize()   # main only; side keeps running
      ids.append(s.ids.cpu().numpy().copy())
    else:
      step()
      torch.cuda.synchronize()
    losses.append(learner.losses.cpu().numpy().copy())
  replay.check_status()
  torch.cuda.synchronize()
  torch.cuda.set_stream(prev)
  return (np.stack(losses), learner.online.cpu().numpy(),
          replay.tree_storage.cpu().numpy(),
          float(replay.max_seen_priority_device.item()),
          learner.target.cpu().numpy())


def test_overlapped_steps_are_bit_identical_to_sequential():
  """(The two-stream loop splits the step into three calls -- nets | loss | backward +
  optimiser -- and a split step uses the STORED fc1 weight gradient; the one-call step
  forms it inside the optimiser (dz_fc1_onfly.h), the same sums in another float32
  order.  The two-stream comparisons therefore pin the stored form on both sides; the
  graph replays of the one-call step are compared in its default form.)"""
  ref = _run(pipelined=False)
  ref_stored = _run(pipelined=False, stored_gradient=True)
  for pipelined, graphs, sync in ((True, False, True), (False, True, True),
                                  (True, True, True), (True, True, False)):
    got = _run(pipelined, graphs=graphs, sync_every_step=sync, stored_gradient=pipelined)
    want = ref_stored if pipelined else ref
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    # the pipelined loop has prefetched one extra sample but the tree only
    # changes through write-backs, which are identical
    np.testing.assert_array_equal(got[2], want[2])
    assert got[3] == want[3]
  assert np.isfinite(ref[0]).all() and ref[0].std() > 0
  # the two forms of the fc1 gradient: same mathematics, float32 rounding apart
  np.testing.assert_allclose(ref[0], ref_stored[0], rtol=2e-5, atol=1e-6)
  np.testing.assert_allclose(ref[1], ref_stored[1], rtol=0, atol=2e-6)


@pytest.mark.parametrize('capacity', [2048, 40])
def test_fused_next_sample_is_bit_identical_to_sequential(capacity):
  """sample(k+1) + gather(k+1) as side blocks of step k's optimiser launch, with
  write-back(k) in an earlier backward launch (the two-round-trip LDS walk): same ids
  (through the losses), losses, parameters, tree and running max as the sequential
  step.  capacity 40 < batch: duplicate ids in every batch (last-wins) and shared
  tree paths from the leaves up."""
  ref = _run(pipelined=False, capacity=capacity)
  got = _run(pipelined=False, fused_sample=True, capacity=capacity)
  for a, b in zip(got, ref):
    np.testing.assert_array_equal(a, b)
  assert np.isfinite(ref[0]).all() and ref[0].std() > 0


@pytest.mark.parametrize('kind', ['dqn_uniform', 'double_q_prioritized', 'c51_adam_uniform'])
def test_dense_fused_next_sample_is_bit_identical_to_sequential(kind):
  """DenseLearner.step(next_sample=...) (the next step's sample + gather inside the
  finalize+RMSProp launch, or inside Adam for the C51 learner; the uniform replay's
  positions -> ids -> rows, or the prioritized replay after the |td| write-back) ==
  sample_device() followed by step(): ids, td errors, parameters, tree."""
  from dqn_zoo_amd import learner as ll, networks, parts
  from dqn_zoo_amd import replay as rl
  A, B, cap = 5, 32, 600
  T = rl.Transition
  dev = torch.device('cuda', 0)

  def run(fused):
    rs = np.random.RandomState(2)
    if kind == 'double_q_prioritized':
      rep = rl.PrioritizedTransitionReplay(
          cap, T(None, None, None, None, None), 0.6,
          parts.LinearSchedule(begin_t=0, end_t=1000, begin_value=0.4, end_value=1.0),
          1e-3, True, np.random.RandomState(21))
    else:
      rep = rl.TransitionReplay(cap, T(None, None, None, None, None), np.random.RandomState(21))
    fields = [torch.from_numpy(rs.randint(0, 256, (cap, 84, 84, 4)).astype(np.uint8)).to(dev),
              torch.from_numpy(rs.randint(0, A, cap).astype(np.int64)).to(dev),
              torch.from_numpy(rs.randint(-1, 2, cap).astype(np.float64)).to(dev),
              torch.from_numpy((rs.randint(0, 2, cap) * 0.99)).to(dev),
              torch.from_numpy(rs.randint(0, 256, (cap, 84, 84, 4)).astype(np.uint8)).to(dev)]
    rep.bulk_fill(fields) if kind != 'double_q_prioritized' else rep.bulk_fill(fields, priority=1.0)
    if kind == 'c51_adam_uniform':
      sup = np.linspace(-10, 10, 51).astype(np.float32)
      ln = ll.DenseLearner(networks.DenseNetwork('c51', A, support=sup), 'categorical',
                           ll.AdamConfig(learning_rate=2.5e-4, eps=0.01 / 32, max_global_grad_norm=0.0),
                           B, seed=4)
    elif kind == 'dqn_uniform':
      ln = ll.DenseLearner(networks.DenseNetwork('dqn', A), 'q', ll.RmsPropConfig(), B, seed=4)
    else:
      ln = ll.DenseLearner(networks.DenseNetwork('double_dqn', A), 'double_q',
                           ll.RmsPropConfig(), B, seed=4)
    ln.use_graphs = False
    nxt, out = None, []
    for _ in range(8):
      if kind == 'double_q_prioritized':
        sm = nxt if nxt is not None else rep.sample_device(B)
        t, ids, w = sm.transitions, sm.ids, sm.weights32
        sink = rep.priority_sink(ids)
      else:
        t, ids = nxt if nxt is not None else rep.sample_device(B)
        w, sink = None, None
      desc = None
      if fused:
        desc, nxt = rep.prepare_next_sample(B)
      ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, w, priority_sink=sink,
              next_sample=desc)
      torch.cuda.synchronize()
      out.append((ids.cpu().numpy().copy(), ln.losses.cpu().numpy().copy()))
    rep.check_status()
    tree = rep.tree_storage.cpu().numpy() if kind == 'double_q_prioritized' else np.zeros(1)
    return out, ln.online.cpu().numpy(), tree

  a, b = run(False), run(True)
  for (ia, la), (ib, lb) in zip(a[0], b[0]):
    np.testing.assert_array_equal(ia, ib)
    np.testing.assert_array_equal(la, lb)
  np.testing.assert_array_equal(a[1], b[1])
  np.testing.assert_array_equal(a[2], b[2])
  assert np.isfinite(a[1]).all() and len({tuple(x[0]) for x in a[0]}) == 8


def test_next_sample_needs_the_write_back_of_the_same_step():
  """A prioritized next_sample without this step's priority_sink would sample BEFORE the
  write-back the reference performs first (rainbow/agent.py:194-198): refused."""
  import bench
  args = types.SimpleNamespace(capacity=256, batch=32)
  replay, learner, _ = bench.build_workload(args, torch.device('cuda', 0), seed=5)
  learner.use_graphs = False
  s = replay.sample_device(32)
  t = s.transitions
  desc, _ = replay.prepare_next_sample(32)
  with pytest.raises(ValueError):
    learner.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32, next_sample=desc)
  with pytest.raises(ValueError):   # and it needs the whole step in one call
    learner.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32, phases=2,
                 priority_sink=replay.priority_sink(s.ids), next_sample=desc)
  torch.cuda.synchronize()


def test_pipelined_target_sync_matches_sequential():
  """sync_target() between two pipelined steps: the prefetched target apply used
  the OLD parameters and is redone in line -- same bits as the sequential loop."""
  ref = _run(pipelined=False, steps=9, sync_target_at=(3, 4, 7), stored_gradient=True)
  for graphs in (False, True):
    got = _run(pipelined=True, steps=9, graphs=graphs, sync_every_step=False,
               sync_target_at=(3, 4, 7), stored_gradient=True)
    for a, b in zip(got, ref):
      np.testing.assert_array_equal(a, b)
  # the sync matters: without it the losses differ from step 3 on
  plain = _run(pipelined=False, steps=9, stored_gradient=True)
  assert (plain[0][:3] == ref[0][:3]).all() and (plain[0][3:] != ref[0][3:]).any()


def test_target_pre_equals_three_apply_step():
  """One stream, no pipeline: target_forward(s_t) followed by step(target_pre=True)
  gives the bits of the three-apply step, with device-drawn noise (the target block's
  stream positions) over several optimiser steps, and with injected noise."""
  from dqn_zoo_amd import learner as ll, networks
  A, B = 5, 32
  sup = np.linspace(-10, 10, 51).astype(np.float32)
  rs = np.random.RandomState(0)
  dev = torch.device('cuda', 0)
  mk = lambda: ll.RainbowLearner(networks.RainbowNetwork(A, sup, 0.1), ll.AdamConfig(), B, seed=7)
  la, lb = mk(), mk()
  lb.target.add_(torch.from_numpy(rs.uniform(-0.05, 0.05, lb.target.numel()).astype(np.float32)).to(dev))
  la.target.copy_(lb.target)
  for ln in (la, lb):
    ln.use_graphs = False
  for it in range(4):
    s_tm1 = torch.from_numpy(rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)).to(dev)
    s_t = torch.from_numpy(rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)).to(dev)
    a = torch.from_numpy(rs.randint(0, A, B).astype(np.int64)).to(dev)
    r = torch.from_numpy(rs.randint(-1, 2, B).astype(np.float64)).to(dev)
    d = torch.from_numpy((rs.randint(0, 2, B) * 0.97).astype(np.float64)).to(dev)
    w = torch.from_numpy(rs.uniform(0.3, 1.0, B).astype(np.float32)).to(dev)
    la.step(s_tm1, a, r, d, s_t, w)
    lb.target_forward(s_t, step_from=lb.adam_count if it == 0 else None)
    lb.step(s_tm1, a, r, d, s_t, w, target_pre=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(la.losses.cpu().numpy(), lb.losses.cpu().numpy())
    np.testing.assert_array_equal(la.priorities.cpu().numpy(), lb.priorities.cpu().numpy())
    np.testing.assert_array_equal(la.online.cpu().numpy(), lb.online.cpu().numpy())
    # the target block of the three-apply step's noise == the block target_forward drew
    st = int(la.layout.noise_stride)
    np.testing.assert_array_equal(la.noise[2 * st:3 * st].cpu().numpy(),
                                  lb._tgt_noise.cpu().numpy())  # pylint: disable=protected-access
  assert np.isfinite(la.losses.cpu().numpy()).all()


def test_priority_sink_equals_separate_update():
  """RainbowLearner.step(priority_sink=...) writes the same tree, bit for bit,
  as step() followed by replay.update_priorities() (rainbow/agent.py:194-198)."""
  import torch
  from dqn_zoo_amd import learner as ll, networks, parts
  from dqn_zoo_amd import replay as rl
  A, B, cap = 4, 16, 300
  sup = np.linspace(-10, 10, 51).astype(np.float32)
  T = rl.Transition

  def build():
    rep = rl.PrioritizedTransitionReplay(
        cap, T(None, None, None, None, None), 0.5,
        parts.LinearSchedule(begin_t=0, end_t=1000, begin_value=0.4, end_value=1.0),
        1e-3, True, np.random.RandomState(11))
    rs = np.random.RandomState(5)
    for i in range(cap + 40):
      rep.add(T(rs.randint(0, 256, (84, 84, 4)).astype(np.uint8), int(rs.randint(A)),
                float(rs.randint(-1, 2)), 0.97, rs.randint(0, 256, (84, 84, 4)).astype(np.uint8)),
              1.0 + (i % 7))
    ln = ll.RainbowLearner(networks.RainbowNetwork(A, sup, 0.1), ll.AdamConfig(), B, seed=3)
    return rep, ln

  rep_a, ln_a = build()
  rep_b, ln_b = build()
  for _ in range(3):
    sa = rep_a.sample_device(B)
    ta = sa.transitions
    ln_a.step(ta.s_tm1, ta.a_tm1, ta.r_t, ta.discount_t, ta.s_t, sa.weights32)
    rep_a.update_priorities(sa.ids, ln_a.priorities)
    sb = rep_b.sample_device(B)
    tb = sb.transitions
    ln_b.step(tb.s_tm1, tb.a_tm1, tb.r_t, tb.discount_t, tb.s_t, sb.weights32,
              priority_sink=rep_b.priority_sink(sb.ids))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(sa.ids.cpu().numpy(), sb.ids.cpu().numpy())
    np.testing.assert_array_equal(ln_a.priorities.cpu().numpy(), ln_b.priorities.cpu().numpy())
    np.testing.assert_array_equal(rep_a.tree_storage.cpu().numpy(),
                                  rep_b.tree_storage.cpu().numpy())
  rep_a.check_status(); rep_b.check_status()
  assert rep_a.max_seen_priority_device.item() == rep_b.max_seen_priority_device.item()
  assert rep_b.check_valid()[0]
  with pytest.raises(ValueError):
    ln_b.step(tb.s_tm1, tb.a_tm1, tb.r_t, tb.discount_t, tb.s_t, sb.weights32,
              phases=1, priority_sink=rep_b.priority_sink(sb.ids))


def test_dense_priority_sink_equals_separate_update():
  """DenseLearner.step(priority_sink=...) for the prioritized double-Q agent:
  same tree as step() + update_priorities(|td|) (prioritized/agent.py:202-206)."""
  import torch
  from dqn_zoo_amd import learner as ll, networks, parts
  from dqn_zoo_amd import replay as rl
  A, B, cap = 5, 12, 200
  T = rl.Transition

  def build():
    rep = rl.PrioritizedTransitionReplay(
        cap, T(None, None, None, None, None), 0.6,
        parts.LinearSchedule(begin_t=0, end_t=1000, begin_value=0.4, end_value=1.0),
        1e-3, True, np.random.RandomState(21))
    rs = np.random.RandomState(6)
    for i in range(cap + 15):
      rep.add(T(rs.randint(0, 256, (84, 84, 4)).astype(np.uint8), int(rs.randint(A)),
                float(rs.randint(-1, 2)), 0.99, rs.randint(0, 256, (84, 84, 4)).astype(np.uint8)),
              1.0)
    ln = ll.DenseLearner(networks.DenseNetwork('double_dqn', A), 'double_q',
                         ll.RmsPropConfig(), B, seed=4)
    return rep, ln

  rep_a, ln_a = build()
  rep_b, ln_b = build()
  for _ in range(3):
    sa = rep_a.sample_device(B); ta = sa.transitions
    ln_a.step(ta.s_tm1, ta.a_tm1, ta.r_t, ta.discount_t, ta.s_t, sa.weights32)
    rep_a.update_priorities(sa.ids, ln_a.priorities)
    sb = rep_b.sample_device(B); tb = sb.transitions
    ln_b.step(tb.s_tm1, tb.a_tm1, tb.r_t, tb.discount_t, tb.s_t, sb.weights32,
              priority_sink=rep_b.priority_sink(sb.ids))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(sa.ids.cpu().numpy(), sb.ids.cpu().numpy())
    np.testing.assert_array_equal(rep_a.tree_storage.cpu().numpy(),
                                  rep_b.tree_storage.cpu().numpy())
  rep_b.check_status()
  assert rep_a.max_seen_priority_device.item() == rep_b.max_seen_priority_device.item()
