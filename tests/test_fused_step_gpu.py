"""Forms of the learner step that must be BIT-IDENTICAL to the sequential eager step
(sample launch, then the one-call update with the write-back inside it): the same step
replayed from hipGraphs, the loop over a static replay whose NEXT sample + gather ride
in this step's optimiser launch (`next_sample`), and the priority write-back carried by
a backward launch (`priority_sink`) instead of a separate update -- same sampled ids,
same losses, same parameters, same tree.  (The two-stream variant of the loop these
tests once also covered measured slower and was removed: EXPERIMENTS.md.)"""

import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(steps=12, graphs=False, sync_target_at=(), fused_sample=False, capacity=2048,
         stored_gradient=False):
  import bench
  args = types.SimpleNamespace(capacity=capacity, batch=32)
  dev = torch.device('cuda', 0)
  replay, learner, _ = bench.build_workload(args, dev, seed=3)
  learner.use_graphs = graphs
  learner.keep_all_grads = stored_gradient
  torch.cuda.synchronize()
  prev = torch.cuda.current_stream(dev)
  torch.cuda.set_stream(torch.cuda.Stream(dev))
  step = bench.make_step(replay, learner, 32, fused_next_sample=fused_sample)
  losses = []
  for k in range(steps):
    if k in sync_target_at:   # target <- online between two steps
      learner.sync_target()
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


def test_graph_replay_and_gradient_forms():
  """hipGraph replay of the one-call step == eager launches, bit for bit; the two forms
  of the fc1 weight gradient (formed inside the optimiser, dz_fc1_onfly.h, vs stored)
  are the same sums in another float32 order."""
  ref = _run()
  ref_stored = _run(stored_gradient=True)
  for stored, want in ((False, ref), (True, ref_stored)):
    got = _run(graphs=True, stored_gradient=stored)
    for a, b in zip(got, want):
      np.testing.assert_array_equal(a, b)
  assert np.isfinite(ref[0]).all() and ref[0].std() > 0
  np.testing.assert_allclose(ref[0], ref_stored[0], rtol=2e-5, atol=1e-6)
  np.testing.assert_allclose(ref[1], ref_stored[1], rtol=0, atol=2e-6)


def test_fused_next_sample_with_target_syncs():
  """sync_target() between steps of the fused loop: the carried sample does not depend
  on either parameter set, so the loop equals the sequential one with the same syncs --
  and differs from the loop without them."""
  ref = _run(steps=9, sync_target_at=(3, 4, 7))
  got = _run(steps=9, sync_target_at=(3, 4, 7), fused_sample=True)
  for a, b in zip(got, ref):
    np.testing.assert_array_equal(a, b)
  plain = _run(steps=9)
  assert (plain[0][:3] == ref[0][:3]).all() and (plain[0][3:] != ref[0][3:]).any()


@pytest.mark.parametrize('capacity', [2048, 40])
def test_fused_next_sample_is_bit_identical_to_sequential(capacity):
  """sample(k+1) + gather(k+1) as side blocks of step k's optimiser launch, with
  write-back(k) in an earlier backward launch (the two-round-trip LDS walk): same ids
  (through the losses), losses, parameters, tree and running max as the sequential
  step.  capacity 40 < batch: duplicate ids in every batch (last-wins) and shared
  tree paths from the leaves up."""
  ref = _run(capacity=capacity)
  got = _run(fused_sample=True, capacity=capacity)
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
    primed, out = False, []
    for _ in range(8):
      if kind == 'double_q_prioritized':
        sm = rep.take_prepared() if primed else rep.sample_device(B)
        t, ids, w = sm.transitions, sm.ids, sm.weights32
        sink = rep.priority_sink(ids)
      else:
        t, ids = rep.take_prepared() if primed else rep.sample_device(B)
        w, sink = None, None
      desc = None
      if fused:
        desc, _ = rep.prepare_next_sample(B)
        primed = True
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
