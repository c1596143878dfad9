"""Rainbow agent drop-in behaviour on the GPU (surface of rainbow/agent.py):
run_loop smoke mirroring rainbow/run_atari_test.py:31-41 (replay 1000, batch
10, learn_period 2) on a synthetic environment, the learning gates, the
inference entry point against the oracle and get_state/set_state."""

import numpy as np
import pytest
import torch

from oracle import qnet_oracle as qo

pytestmark = pytest.mark.gpu

A = 4
SUPPORT = np.linspace(-10.0, 10.0, 51).astype(np.float32)


class SyntheticEnv:
  """Preprocessed-observation environment: uint8 84x84x4 stacks, episodes of
  fixed length, rewards in {-1,0,1} (what processors.atari would emit)."""

  def __init__(self, seed, episode_length=17):
    self.rs = np.random.RandomState(seed)
    self.n = episode_length

  def _obs(self):
    return self.rs.randint(0, 256, (84, 84, 4)).astype(np.uint8)

  def reset(self):
    from dqn_zoo_amd import dm_env_shim as dm_env
    self.t = 0
    return dm_env.restart(self._obs())

  def step(self, action):
    from dqn_zoo_amd import dm_env_shim as dm_env
    assert 0 <= action < A
    self.t += 1
    r = float(self.rs.randint(-1, 2))
    if self.t == self.n:
      return dm_env.termination(r, self._obs())
    return dm_env.transition(r, self._obs(), 0.99)


def _make_agent(seed=1, capacity=1000, batch=10, learn_period=2,
                target_period=40, min_frac=0.05):
  from dqn_zoo_amd import learner, networks, parts, processors
  from dqn_zoo_amd import replay as replay_lib
  from dqn_zoo_amd.rainbow import agent as agent_lib
  net = networks.RainbowNetwork(A, SUPPORT, 0.1)
  rep = replay_lib.PrioritizedTransitionReplay(
      capacity, replay_lib.Transition(None, None, None, None, None), 0.5,
      parts.LinearSchedule(begin_t=50, end_t=500, begin_value=0.4,
                           end_value=1.0), 1e-3, True,
      np.random.RandomState(seed))
  ag = agent_lib.Rainbow(
      preprocessor=processors.Identity(),
      sample_network_input=np.zeros((84, 84, 4), np.uint8), network=net,
      support=SUPPORT, optimizer=learner.AdamConfig(),
      transition_accumulator=replay_lib.NStepTransitionAccumulator(3),
      replay=rep, batch_size=batch, min_replay_capacity_fraction=min_frac,
      learn_period=learn_period, target_network_update_period=target_period,
      rng_key=seed)
  return ag, rep


def test_run_loop_smoke_and_gates():
  import itertools
  from dqn_zoo_amd import parts
  ag, rep = _make_agent()
  env = SyntheticEnv(0)
  p0 = ag.online_params
  assert np.isnan(ag.statistics['state_value']) and ag.max_seen_priority == 1.0
  seq = itertools.islice(parts.run_loop(ag, env, max_steps_per_episode=50), 120)
  stats = parts.generate_statistics(parts.make_default_trackers(ag), seq)
  rep.check_status()
  assert stats['num_steps_since_reset'] == 120 and stats['num_episodes'] >= 5
  assert np.isfinite(stats['state_value'])
  # every step adds transitions (n-step flush at episode ends): size grew
  assert 90 <= rep.size <= 120
  # learning started once size >= 5% of capacity (= 50): parameters moved
  p1 = ag.online_params
  assert any(np.abs(p1[k] - p0[k]).max() > 0 for k in p1)
  steps_learned = int(ag.learner.adam_count.item())
  assert 25 <= steps_learned <= 40   # every 2nd frame after ~50 frames
  assert ag.max_seen_priority >= 1.0
  assert 0.4 <= ag.importance_sampling_exponent <= 1.0
  assert rep.check_valid()[0]
  # repeat-without-action error (rainbow/agent.py:142-143)
  ag2, _ = _make_agent()
  ag2._preprocessor = lambda ts: None
  with pytest.raises(RuntimeError, match='Cannot repeat if action'):
    ag2.step(env.reset())


def test_target_sync_period():
  # min replay = 4 items: with a 3-step accumulator the replay reaches 4 items
  # at frame 6; learning every frame from then on, target sync when
  # frame_t % 7 == 0 (rainbow/agent.py:151-158).
  ag, rep = _make_agent(target_period=7, min_frac=0.004, learn_period=1)
  env = SyntheticEnv(1, episode_length=100)
  ts = env.reset()
  ag.reset()
  synced = []
  for t in range(16):
    a = ag.step(ts)
    ts = env.step(a)
    tgt = ag.learner.get_params('target')
    onl = ag.online_params
    synced.append(all(np.array_equal(tgt[k], onl[k]) for k in tgt))
  learned_at = int(ag.learner.adam_count.item())
  assert learned_at == 16 - 6
  assert synced[:6] == [True] * 6          # nothing learned yet
  assert synced[6] is False                # learned at frame 6, no sync
  assert synced[7] is True and synced[14] is True
  assert not any(synced[8:14]) and synced[15] is False


def test_apply_matches_oracle_forward():
  from dqn_zoo_amd import learner, networks
  rs = np.random.RandomState(5)
  params = qo.init_params('rainbow', A, rs)
  for k in params:
    if 'sigma' in k:
      params[k] = (params[k] * 3).astype(np.float32)
  ln = learner.RainbowLearner(networks.RainbowNetwork(A, SUPPORT),
                              learner.AdamConfig(), 8, params=params)
  for b in (1, 5):
    x = rs.randint(0, 256, (b, 84, 84, 4)).astype(np.uint8)
    nz = qo.sample_noise(rs, A)
    q, greedy, vmax = ln.apply(torch.from_numpy(x).cuda(), noise=nz)
    _, q_ref, _ = qo.rainbow_fwd(params, x, nz, SUPPORT, A)
    np.testing.assert_allclose(q.cpu().numpy(), q_ref, rtol=2e-4, atol=2e-5)
    np.testing.assert_array_equal(greedy.cpu().numpy(), q_ref.argmax(axis=1))
    np.testing.assert_allclose(vmax.cpu().numpy(), q_ref.max(axis=1), rtol=2e-4,
                               atol=2e-5)
  # fresh noise per apply (networks.py:169-170): same input, different output
  xs = torch.from_numpy(x).cuda()
  q1, _, _ = ln.apply(xs)
  q2, g2, v2 = ln.apply(xs)
  assert not torch.equal(q1, q2)
  # the fused actor path (dz_rainbow_act: noise drawn inside the conv1 launch,
  # fc2 fold inside the q-value kernel) against the oracle with the SAME noise,
  # read back from the device
  nz_dev = ln.layout.unpack_noise(ln._act_noise.cpu().numpy())  # pylint: disable=protected-access
  _, q_ref, _ = qo.rainbow_fwd(params, x, nz_dev, SUPPORT, A)
  np.testing.assert_allclose(q2.cpu().numpy(), q_ref, rtol=2e-4, atol=2e-5)
  np.testing.assert_array_equal(g2.cpu().numpy(), q_ref.argmax(axis=1))
  a_t, v_t = ln.read_action(g2, v2)
  assert a_t == int(q_ref[0].argmax()) and abs(v_t - q_ref[0].max()) < 1e-4


def test_get_state_set_state_roundtrip():
  ag, rep = _make_agent(capacity=64, batch=8, min_frac=0.1)
  env = SyntheticEnv(2)
  ts = env.reset()
  ag.reset()
  for _ in range(40):
    ts = env.step(ag.step(ts)) if not ts.last() else env.reset()
  state = ag.get_state()
  assert set(state) == {'rng_key', 'frame_t', 'opt_state', 'online_params',
                        'target_params', 'replay', 'max_seen_priority'}
  ag2, rep2 = _make_agent(capacity=64, batch=8, min_frac=0.1, seed=99)
  ag2.set_state(state)
  s2 = ag2.get_state()
  assert s2['frame_t'] == state['frame_t']
  assert s2['max_seen_priority'] == state['max_seen_priority']
  for k in state['online_params']:
    np.testing.assert_array_equal(s2['online_params'][k],
                                  state['online_params'][k])
    np.testing.assert_array_equal(s2['opt_state']['nu'][k],
                                  state['opt_state']['nu'][k])
  # the replay part is the REFERENCE's dictionary (replay.py:747-754)
  assert set(state['replay']) == {'storage', 't', 'distribution'}
  assert set(state['replay']['distribution']) == {
      'sum_tree', 'id_to_index', 'index_to_id', 'inactive_indices',
      'active_indices', 'active_indices_location'}
  np.testing.assert_array_equal(
      s2['replay']['distribution']['sum_tree']['storage'],
      state['replay']['distribution']['sum_tree']['storage'])
  assert s2['replay']['distribution']['active_indices'] == \
      state['replay']['distribution']['active_indices']
  for (i, a), (j, b) in zip(s2['replay']['storage'], state['replay']['storage']):
    assert i == j and a.a_tm1 == b.a_tm1 and a.r_t == b.r_t
    np.testing.assert_array_equal(a.s_t, b.s_t)
  assert rep2.size == rep.size and list(rep2.ids()) == list(rep.ids())


@pytest.mark.parametrize('num_actions', [3, 6, 18])
def test_one_launch_decision_vs_oracle_and_the_multi_launch_apply(num_actions):
  """A batch of ONE observation with fresh noise is one launch (csrc/dz_act_one.h: torso, fc1
  stream and tail as workgroup roles, intermediates that are their own flags, two alternating
  sets).  Seven consecutive decisions -- every set used at least three times -- against the
  oracle forward with the noise the kernel drew and wrote out, against the multi-launch apply on
  the same noise, and the seam words left re-armed."""
  from dqn_zoo_amd import learner, networks
  rs = np.random.RandomState(17 + num_actions)
  params = qo.init_params('rainbow', num_actions, rs)
  for k in params:
    if 'sigma' in k:
      params[k] = (params[k] * 3).astype(np.float32)
  ln = learner.RainbowLearner(networks.RainbowNetwork(num_actions, SUPPORT),
                              learner.AdamConfig(), 8, params=params)
  ln.act_graphs = False
  seams = int(ln.network.layout(1).c.ws_act_seams)
  for i in range(7):
    x = rs.randint(0, 256, (1, 84, 84, 4)).astype(np.uint8)
    if i == 3:
      x[:] = 0          # all-zero activations are stored as -0.0f, never as "not written"
    xd = torch.from_numpy(x).cuda()
    q, greedy, vmax = ln.apply(xd)
    torch.cuda.synchronize()
    nz = ln.layout.unpack_noise(ln._act_noise.cpu().numpy())  # pylint: disable=protected-access
    _, q_ref, _ = qo.rainbow_fwd(params, x, nz, SUPPORT, num_actions)
    np.testing.assert_allclose(q.cpu().numpy(), q_ref, rtol=2e-4, atol=2e-5)
    assert int(greedy[0]) == int(q_ref[0].argmax())
    np.testing.assert_allclose(float(vmax[0]), q_ref[0].max(), rtol=2e-4, atol=2e-5)
    q5, g5, _ = ln.apply(xd, noise=nz)       # stored noise: the multi-launch kernels
    np.testing.assert_allclose(q.cpu().numpy(), q5.cpu().numpy(), rtol=0, atol=2e-6)
    assert int(g5[0]) == int(greedy[0])
    assert ln.act_step() == i + 1
    words = ln._act_ws[seams:seams + 64 * 8:64].view(torch.int32).tolist()  # pylint: disable=protected-access
    assert words[3] == i + 1 and words[4] == 0 and words[5] == 0, words


def test_async_acting_path_matches_oracle_and_replays_from_a_graph():
  """`apply_async` (the agents' acting path): launches enqueued, (action, value)
  written by the kernel into pinned host memory, replayed from a hipGraph from
  the second call on, fresh noise on every call (device-side stream counter) --
  each decision checked against the oracle forward with the noise the device
  actually drew."""
  from dqn_zoo_amd import device_obs, learner, networks, parts
  rs = np.random.RandomState(11)
  params = qo.init_params('rainbow', A, rs)
  for k in params:
    if 'sigma' in k:
      params[k] = (params[k] * 3).astype(np.float32)
  ln = learner.RainbowLearner(networks.RainbowNetwork(A, SUPPORT), learner.AdamConfig(),
                              8, params=params)
  cache = device_obs.ObservationCache(ln.device)
  prev = torch.cuda.current_stream()
  torch.cuda.set_stream(torch.cuda.Stream())   # graph capture needs a real stream
  try:
    seen = []
    for i in range(2 * cache._depth + 3):   # every ring slot twice: capture, then replay
      x = rs.randint(0, 256, (84, 84, 4)).astype(np.uint8)
      pending = parts.PendingAction(ln.apply_async(cache.upload(x)))
      a = pending.resolve()
      nz = ln.layout.unpack_noise(ln._act_noise.cpu().numpy())  # pylint: disable=protected-access
      _, q_ref, _ = qo.rainbow_fwd(params, x[None], nz, SUPPORT, A)
      assert a == int(q_ref[0].argmax())
      assert abs(pending.state_value - q_ref[0].max()) < 1e-4
      assert int(pending) == a and np.int64(pending) == a
      seen.append(nz['adv1/in'][:8].copy())
    assert ln.act_step() == len(seen)
    assert len(ln._act_graphs) == cache._depth   # pylint: disable=protected-access
    for u, v in zip(seen, seen[1:]):
      assert not np.array_equal(u, v)          # fresh noise per decision
  finally:
    torch.cuda.synchronize()
    torch.cuda.set_stream(prev)


def test_diverged_priorities_stop_the_run_within_a_few_steps():
  """ADVICE r1 / VERDICT r2: NaN / negative priorities raise ValueError in the
  reference's SumTree.set (replay.py:281-282) at the offending call.  Here they land in a
  sticky status word in pinned host memory that the agents look at (a plain load, no
  synchronisation) at every learner step, and definitively at every target-network sync:
  the run stops a few enqueued steps after the offending kernel, long before the next
  target sync (period 1000 here)."""
  ag, rep = _make_agent(target_period=1000, min_frac=0.004, learn_period=1)
  env = SyntheticEnv(2, episode_length=100)
  ts = env.reset()
  ag.reset()
  for _ in range(9):
    ts = env.step(ag.step(ts))
  rep.poll_status()   # nothing flagged so far
  # poison the priorities of the next write-back the way a diverged loss would
  ids = rep.sample_device(4).ids
  rep.update_priorities(ids, torch.full((4,), float('nan'), device='cuda'))
  with pytest.raises(ValueError, match='finite and positive'):
    for _ in range(12):
      ts = env.step(ag.step(ts))
  # the word is clear again and the definitive check agrees
  rep.check_status()


def test_scalar_dtypes_are_canonicalised_on_insert():
  """an environment emitting np.float32 rewards / np.int32 actions must
  not freeze the store to those dtypes (the learner kernels read int64/float64)."""
  from dqn_zoo_amd import dm_env_shim as dm_env
  from dqn_zoo_amd import learner
  ag, rep = _make_agent(min_frac=0.004, learn_period=1)
  rs = np.random.RandomState(0)
  ts = dm_env.restart(rs.randint(0, 256, (84, 84, 4)).astype(np.uint8))
  for _ in range(12):
    ag.step(ts)
    ts = dm_env.TimeStep(dm_env.StepType.MID, np.float32(1.0), np.float32(0.99),
                         rs.randint(0, 256, (84, 84, 4)).astype(np.uint8))
  torch.cuda.synchronize()
  f = rep._ring.fields   # pylint: disable=protected-access
  assert f[1].dtype == torch.int64 and f[2].dtype == torch.float64 and f[3].dtype == torch.float64
  assert int(ag.learner.adam_count.item()) > 0
  # and a wrong dtype handed to the learner directly is a TypeError, not an assert
  t = rep.sample_device(10)
  tr = t.transitions
  with pytest.raises(TypeError, match='r_t'):
    ag.learner.step(tr.s_tm1, tr.a_tm1, tr.r_t.float(), tr.discount_t, tr.s_t, t.weights32)


def test_set_state_restores_the_noise_streams_of_a_stepped_agent():
  """restoring into an agent that has already stepped (cached argument
  block, captured graphs) must continue exactly like the agent the state came
  from: same learner noise, same actor noise, same parameters."""
  def run(ag, env, n):
    ts = env.reset()
    ag.reset()
    acts = []
    for _ in range(n):
      a = ag.step(ts)
      acts.append(a)
      ts = env.step(a) if not ts.last() else env.reset()
    return acts

  a1, _ = _make_agent(min_frac=0.004, learn_period=1, capacity=64, batch=8)
  run(a1, SyntheticEnv(5), 25)
  state = a1.get_state()
  rs_state = a1._replay._random_state.get_state()   # pylint: disable=protected-access
  a2, _ = _make_agent(min_frac=0.004, learn_period=1, capacity=64, batch=8, seed=77)
  run(a2, SyntheticEnv(6), 13)          # a DIFFERENT history first: caches are warm
  a2.set_state(state)
  a2._replay._random_state.set_state(rs_state)      # pylint: disable=protected-access
  # identical continuations (fresh accumulators on both sides: reset() at the start)
  acts1 = run(a1, SyntheticEnv(9), 20)
  acts2 = run(a2, SyntheticEnv(9), 20)
  assert acts1 == acts2
  p1, p2 = a1.online_params, a2.online_params
  for k in p1:
    np.testing.assert_array_equal(p1[k], p2[k])
  assert a1.learner.act_step() == a2.learner.act_step()


def test_agent_loop_is_reproducible_frame_by_frame():
  """Two runs of the drop-in loop (act -> add -> learn, n-step flushes at episode ends,
  learn_period 2, target syncs) from the same seeds produce the SAME actions, losses,
  parameters, tree and running max priority, frame by frame: nothing in the enqueue order
  (decision kernel, inserts, learner step) depends on timing."""
  def run():
    ag, rep = _make_agent(seed=3, capacity=200, batch=10, learn_period=2, target_period=15,
                          min_frac=0.05)
    env = SyntheticEnv(7, episode_length=13)
    ts = env.reset()
    ag.reset()
    log = []
    for t in range(90):
      a = ag.step(ts)
      ts = env.step(a)
      if ts.last():
        ag.step(ts)          # the terminal frame (flushes the n-step window)
        ts = env.reset()
        ag.reset()
      torch.cuda.synchronize()
      log.append((int(a), int(ag.learner.adam_count.item()),
                  ag.learner.losses.cpu().numpy().copy()))
    rep.check_status()
    return (log, ag.learner.online.cpu().numpy(), ag.learner.target.cpu().numpy(),
            rep.tree_storage.cpu().numpy(), ag.max_seen_priority, rep.size)

  a, b = run(), run()
  assert a[0][-1][1] >= 30            # it learned (both loops the same number of steps)
  for (aa, ca, la), (ab, cb, lb) in zip(a[0], b[0]):
    assert aa == ab and ca == cb
    np.testing.assert_array_equal(la, lb)
  for x, y in zip(a[1:], b[1:]):
    np.testing.assert_array_equal(x, y)
