"""Pins the Q-loss / update half to the REAL reference (VERDICT r3 #6a).

    # in an environment with docker_requirements.txt:6-16 installed (jax 0.3.10 CPU is enough,
    # --jax_platform_name=cpu), from the root of THIS repository:
    python tools/gen_reference_qnet_fixtures.py --dqn_zoo /path/to/dqn_zoo_checkout

writes tests/golden/ref_qnet_<agent>.npz for the seven agents.  tests/test_qnet_reference.py
then compares the CPU oracle and the HIP learners with those files (and says UNPINNED, with
this command, while they are absent).  It cannot run in the build container -- jax, haiku,
optax and rlax are not installable there -- which is exactly why the half is "parity
unpinned" (DESIGN.md 2); `--selfcheck` exercises everything below that does not need JAX.

What one fixture is.  The inputs are NOT stored: they are tests/golden/qnet_cases.py's seeded
case (`make_inputs(name, float32)`: parameters of both networks, a batch of 4 transitions,
importance weights, Rainbow's factorised noise, IQN's taus), injected into the unmodified
reference objects:
  * parameters: written into the haiku parameter tree the agent's own `network.init`
    produced (leaf by leaf, matched by layer kind, creation order and shape -- asserted);
  * Rainbow's noise: `jax.random.truncated_normal` (networks.py:142) is replaced, for the
    duration of the one update, by a function that returns the value x with
    sign(x) sqrt|x| == the case's noise value, in the order the network draws it (per apply:
    adv1 in/out, adv2 in/out, val1 in/out, val2 in/out; applies: online(s_tm1), online(s_t),
    target(s_t), rainbow/agent.py:87-96).  The noise the network ACTUALLY used (float32
    sign * sqrt of what was returned) is stored in the fixture and is what the consumers feed;
  * IQN's taus: `iqn.agent._sample_tau` (iqn/agent.py:45-50) returns the case's taus, in the
    order tau_tm1, tau_t_selector, tau_t (iqn/agent.py:181-187).
Then the agent's own `update` closure (`agent._update`, e.g. rainbow/agent.py:112-123) runs ONCE
under `jax.disable_jit()` with the run script's optimizer (e.g. rainbow/run_atari.py:229-235)
wrapped in a transformation that records the gradient it is handed (the reference never exposes
`jax.grad`'s output).  Stored, in qnet_cases.pack()'s layout: per-sample losses / TD errors where
the update returns them, every gradient tensor (sampled entries + sum, L2, max), the updated
parameters and both optimiser moments.
"""

import argparse
import os
import re
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from tests.golden import qnet_cases as qc  # noqa: E402

OUT_DIR = os.path.join(ROOT, 'tests', 'golden')
NOISE_ORDER = ('adv1/in', 'adv1/out', 'adv2/in', 'adv2/out', 'val1/in', 'val1/out',
               'val2/in', 'val2/out')


# --------------------------------------------------------------------------- #
#  haiku parameter tree  <->  this repository's names
# --------------------------------------------------------------------------- #
def _creation_index(module_name):
  """hk names repeated modules  base, base_1, base_2 ... in creation order."""
  last = module_name.split('/')[-1]
  m = re.match(r'^(.*?)(?:_(\d+))?$', last)
  return m.group(1), int(m.group(2) or 0)


def map_leaves(tree, ours):
  """tree: {module: {leaf: array}} as `network.init` returns it; ours: ordered dict
  name -> array in this repository's naming (network creation order).  Returns
  {our name: (module, leaf)}.  A noisy layer's two hk.Linear modules are called 'mu' /
  'sigma' (networks.py:151-167); everything else is matched by leaf name and shape, the
  earliest-created unused module first."""
  leaves = []
  for mod, d in tree.items():
    base, idx = _creation_index(mod)
    for leaf, arr in d.items():
      leaves.append(dict(mod=mod, leaf=leaf, base=base, idx=idx, shape=tuple(np.shape(arr))))
  used, out = set(), {}
  for name, arr in ours.items():
    parts = name.split('/')
    leaf = parts[-1]
    hint = parts[-2] if parts[-2] in ('mu', 'sigma') else None
    cands = [l for l in leaves
             if l['leaf'] == leaf and l['shape'] == tuple(arr.shape) and
             (hint is None or l['base'] == hint) and (l['mod'], l['leaf']) not in used]
    if not cands:
      raise AssertionError('no haiku leaf for %s %s among %s' % (
          name, arr.shape, sorted((l['mod'], l['leaf'], l['shape']) for l in leaves)))
    pick = min(cands, key=lambda l: l['idx'])
    used.add((pick['mod'], pick['leaf']))
    out[name] = (pick['mod'], pick['leaf'])
  if len(used) != len(leaves):
    raise AssertionError('haiku leaves without a counterpart: %s' % sorted(
        (l['mod'], l['leaf']) for l in leaves if (l['mod'], l['leaf']) not in used))
  return out


def to_tree(mapping, ours, like, as_array=np.asarray):
  tree = {mod: dict(d) for mod, d in like.items()}
  for name, (mod, leaf) in mapping.items():
    assert tuple(np.shape(tree[mod][leaf])) == tuple(ours[name].shape), name
    tree[mod][leaf] = as_array(ours[name])
  return tree


def from_tree(mapping, tree):
  return {name: np.asarray(tree[mod][leaf]) for name, (mod, leaf) in mapping.items()}


def find_moments(state):
  """(mu, nu) of the first optax state in `state` that has them (ScaleByAdamState in the
  chain of rainbow/run_atari.py:229-235, ScaleByRStdDevState of dqn/run_atari.py:205-210)."""
  stack = [state]
  while stack:
    s = stack.pop(0)
    if hasattr(s, 'mu') and hasattr(s, 'nu'):
      return s.mu, s.nu
    if isinstance(s, (tuple, list)):
      stack.extend(s)
  raise AssertionError('no optimiser state with mu / nu in %r' % (state,))


# --------------------------------------------------------------------------- #
#  the random draws of one update, replaced by the case's values IN THE ORDER the reference
#  makes them (both feeders refuse a request whose shape does not fit the next value: a wrong
#  order cannot go unnoticed)
# --------------------------------------------------------------------------- #
def noise_feeder(noises, used, asarray, default_dtype):
  """Stand-in for `jax.random.truncated_normal` during ONE Rainbow update: per apply
  (online(s_tm1), online(s_t), target(s_t): rainbow/agent.py:87-96) the network draws adv1 in /
  out, adv2 in / out, val1 in / out, val2 in / out (networks.py:239-252, each noisy_linear:
  input noise then output noise, networks.py:169-170).  Returns x with sign(x) sqrt|x| == the
  case's value; what the network then forms from it is appended to `used`."""
  queue = [np.asarray(noises[g][k], np.float32) for g in range(3) for k in NOISE_ORDER]

  def fake_tn(key, lower, upper, shape=None, dtype=None):
    del key
    assert float(lower) == -2.0 and float(upper) == 2.0, (lower, upper)   # networks.py:143
    v = queue.pop(0)
    assert shape is not None and int(np.prod(shape)) == v.size and tuple(shape)[0] == 1, (
        'draw %d of the update asks for shape %r, the case has %d values next: the order of the '
        'noise draws differs from NOISE_ORDER' % (len(used), shape, v.size))
    x = (np.sign(v) * v.astype(np.float64) ** 2).astype(np.float32)
    used.append(np.sign(x) * np.sqrt(np.abs(x)))          # what make_noise_sqrt forms
    return asarray(x, dtype or default_dtype).reshape(shape)

  return fake_tn, queue


def tau_feeder(taus, asarray):
  """Stand-in for `iqn.agent._sample_tau` (iqn/agent.py:45-50) during ONE update: tau_tm1,
  tau_t_selector, tau_t in that order (iqn/agent.py:181-187)."""
  tq = [np.asarray(t, np.float32) for t in taus]

  def fake_sample_tau(key, shape):
    del key
    t = tq.pop(0)
    assert tuple(shape) == tuple(t.shape), (shape, t.shape)
    return asarray(t).reshape(shape)

  return fake_sample_tau, tq


# --------------------------------------------------------------------------- #
#  one agent, one update
# --------------------------------------------------------------------------- #
def run_reference(name, zoo_root):
  """Runs the reference's update for case `name`; returns qnet_cases.pack()-shaped dict."""
  sys.path.insert(0, zoo_root)
  import haiku as hk
  import jax
  import jax.numpy as jnp
  import optax
  from dqn_zoo import networks
  from dqn_zoo import replay as replay_lib

  inp = qc.make_inputs(name, np.float32)
  c = inp['case']
  A, B = qc.A, qc.B
  s_tm1, a, r, d, s_t = inp['batch']
  sink = {}

  def recording(opt):
    def update(grads, state, params=None):
      sink['grads'] = grads
      return opt.update(grads, state, params)
    return optax.GradientTransformation(opt.init, update)

  if c['opt'] == 'adam':                               # */run_atari.py (cited in qnet_cases.CASES)
    opt = optax.adam(learning_rate=c['lr'], eps=c['eps'])
    if c['max_norm'] > 0:
      opt = optax.chain(optax.clip_by_global_norm(c['max_norm']), opt)
  else:
    opt = optax.rmsprop(learning_rate=c['lr'], decay=c['decay'], eps=c['eps'], centered=True)
  opt = recording(opt)

  support = jnp.asarray(qc.SUPPORT, jnp.float32)
  quantiles = jnp.asarray(qc.QUANTILES, jnp.float32)
  sample_in = np.zeros((84, 84, 4), np.uint8)
  common = dict(preprocessor=lambda ts: ts, transition_accumulator=None,
                replay=types.SimpleNamespace(capacity=100), batch_size=B,
                min_replay_capacity_fraction=0.5, learn_period=1,
                target_network_update_period=1, rng_key=jax.random.PRNGKey(0))
  eps_fn = lambda t: 0.1
  if name == 'rainbow':
    from dqn_zoo.rainbow import agent as agent_lib
    net = hk.transform(networks.rainbow_atari_network(A, support, 0.1))
    ag = agent_lib.Rainbow(sample_network_input=sample_in, network=net, support=support,
                           optimizer=opt, **common)
  elif name in ('dqn', 'double_q', 'prioritized'):
    mod = {'dqn': 'dqn', 'double_q': 'double_q', 'prioritized': 'prioritized'}[name]
    agent_lib = __import__('dqn_zoo.%s.agent' % mod, fromlist=['agent'])
    fn = networks.dqn_atari_network if name == 'dqn' else networks.double_dqn_atari_network
    net = hk.transform(fn(A))
    cls = {'dqn': 'Dqn', 'double_q': 'DoubleDqn', 'prioritized': 'PrioritizedDqn'}[name]
    ag = getattr(agent_lib, cls)(sample_network_input=sample_in, network=net, optimizer=opt,
                                 exploration_epsilon=eps_fn, grad_error_bound=c['bound'], **common)
  elif name == 'c51':
    from dqn_zoo.c51 import agent as agent_lib
    net = hk.transform(networks.c51_atari_network(A, support))
    ag = agent_lib.C51(sample_network_input=sample_in, network=net, support=support,
                       optimizer=opt, exploration_epsilon=eps_fn, **common)
  elif name == 'qr':
    from dqn_zoo.qrdqn import agent as agent_lib
    net = hk.transform(networks.qr_atari_network(A, quantiles))
    ag = agent_lib.QrDqn(sample_network_input=sample_in, network=net, quantiles=quantiles,
                         optimizer=opt, exploration_epsilon=eps_fn, huber_param=c['kappa'],
                         **common)
  elif name == 'iqn':
    from dqn_zoo.iqn import agent as agent_lib
    net = hk.transform(networks.iqn_atari_network(A, 64))
    n1, n2, n3 = qc.IQN_TAUS
    ag = agent_lib.Iqn(
        sample_network_input=agent_lib.IqnInputs(state=sample_in, taus=np.zeros(n2, np.float32)),
        network=net, optimizer=opt, exploration_epsilon=eps_fn, huber_param=c['kappa'],
        tau_samples_policy=n2, tau_samples_s_tm1=n1, tau_samples_s_t=n3, **common)
  else:
    raise KeyError(name)

  like = hk.data_structures.to_mutable_dict(ag._online_params)   # pylint: disable=protected-access
  mapping = map_leaves(like, inp['online'])
  as_j = lambda x: jnp.asarray(np.asarray(x, np.float32))
  online = hk.data_structures.to_immutable_dict(to_tree(mapping, inp['online'], like, as_j))
  target = hk.data_structures.to_immutable_dict(to_tree(mapping, inp['target'], like, as_j))
  opt_state = opt.init(online)
  transitions = replay_lib.Transition(s_tm1=s_tm1, a_tm1=a, r_t=r, discount_t=d, s_t=s_t)
  used_noise = []
  undo = []
  if name == 'rainbow':
    fake_tn, queue = noise_feeder(inp['noises'], used_noise, lambda x, dt: jnp.asarray(x, dt),
                                  jnp.float32)
    orig_tn = jax.random.truncated_normal
    jax.random.truncated_normal = fake_tn
    undo.append(lambda: setattr(jax.random, 'truncated_normal', orig_tn))
  if name == 'iqn':
    fake_st, tq = tau_feeder(inp['taus'], jnp.asarray)
    orig_st = agent_lib._sample_tau                                # pylint: disable=protected-access
    agent_lib._sample_tau = fake_st                                # pylint: disable=protected-access
    undo.append(lambda: setattr(agent_lib, '_sample_tau', orig_st))
  try:
    with jax.disable_jit():
      args = [jax.random.PRNGKey(1), opt_state, online, target, transitions]
      if name in ('rainbow', 'prioritized'):
        args.append(jnp.asarray(inp['weights'], jnp.float32))
      res = ag._update(*args)                                      # pylint: disable=protected-access
  finally:
    for u in undo:
      u()
  if name == 'rainbow':
    assert not queue and len(used_noise) == 24, (len(queue), len(used_noise))
  if name == 'iqn':
    assert not tq
  new_opt_state, new_params = res[1], res[2]
  losses = np.asarray(res[3], np.float64) if len(res) > 3 else np.zeros(0)
  grads = from_tree(mapping, hk.data_structures.to_mutable_dict(sink['grads']))
  mu, nu = find_moments(new_opt_state)
  result = dict(
      loss=(np.mean(losses * inp['weights']) if (losses.size and inp['weights'] is not None)
            else (np.mean(losses) if losses.size else np.nan)),
      losses=losses, grads=grads,
      params=from_tree(mapping, hk.data_structures.to_mutable_dict(new_params)),
      opt=dict(m=from_tree(mapping, hk.data_structures.to_mutable_dict(mu)),
               v=from_tree(mapping, hk.data_structures.to_mutable_dict(nu))),
      gnorm=float(optax.global_norm(sink['grads'])))
  out = qc.pack(result)
  for i, v in enumerate(used_noise):
    out['noise/%d/%s' % (i // 8, NOISE_ORDER[i % 8])] = v.astype(np.float32)
  out['versions'] = np.array(['jax ' + jax.__version__, 'haiku ' + hk.__version__,
                              'optax ' + optax.__version__])
  return out


# --------------------------------------------------------------------------- #
#  what can be checked without JAX
# --------------------------------------------------------------------------- #
def selfcheck():
  """The name mapping, the tree round trip, the noise inversion and pack() on trees shaped
  like haiku's (module names per networks.py) but filled from the oracle's initialiser."""
  from oracle import qnet_oracle as qo
  hk_names = {
      'rainbow': ['conv2_d', 'conv2_d_1', 'conv2_d_2', 'mu', 'sigma', 'mu_1', 'sigma_1', 'mu_2',
                  'sigma_2', 'mu_3', 'sigma_3'],
      'dqn': ['conv2_d', 'conv2_d_1', 'conv2_d_2', 'linear', 'linear_1'],
      'iqn': ['conv2_d', 'conv2_d_1', 'conv2_d_2', 'linear', 'linear_1', 'linear_2'],
  }
  for name in ('rainbow', 'dqn', 'double_q', 'prioritized', 'c51', 'qr', 'iqn'):
    inp = qc.make_inputs(name, np.float32)
    ours = inp['online']
    mods = hk_names.get(name, hk_names['dqn'])
    # group our keys by layer, in creation order
    layers = []
    for k in ours:
      lay = k.rsplit('/', 1)[0]
      if lay not in layers:
        layers.append(lay)
    if name == 'rainbow':   # noisy layers: adv1 adv2 val1 val2, each mu then sigma
      layers = ['conv1', 'conv2', 'conv3', 'adv1/mu', 'adv1/sigma', 'adv2/mu', 'adv2/sigma',
                'val1/mu', 'val1/sigma', 'val2/mu', 'val2/sigma']
    assert len(layers) == len(mods), (name, layers, mods)
    tree = {}
    for lay, mod in zip(layers, mods):
      tree[mod] = {k.rsplit('/', 1)[1]: np.zeros_like(v) for k, v in ours.items()
                   if k.rsplit('/', 1)[0] == lay}
    # shuffled module order: the mapping must not depend on dict order
    tree = {m: tree[m] for m in sorted(tree, reverse=True)}
    mapping = map_leaves(tree, ours)
    for lay, mod in zip(layers, mods):
      for k in ours:
        if k.rsplit('/', 1)[0] == lay:
          assert mapping[k][0] == mod, (name, k, mapping[k], mod)
    back = from_tree(mapping, to_tree(mapping, ours, tree))
    assert all(np.array_equal(back[k], ours[k]) for k in ours)
    if name == 'rainbow':
      v = np.asarray(inp['noises'][0]['adv1/in'], np.float32)
      x = (np.sign(v) * v.astype(np.float64) ** 2).astype(np.float32)
      used = np.sign(x) * np.sqrt(np.abs(x))
      assert np.abs(used - v).max() <= np.spacing(np.abs(v).max()), 'noise inversion'
    if name == 'rainbow':
      # the noise feeder, driven in the reference's draw order with the shapes its network asks
      # for (networks.py:169-170, 239-252): 3 applies x 8 draws, every value used exactly once
      used = []
      fake_tn, queue = noise_feeder(inp['noises'], used, lambda x, dt: np.asarray(x, dt), np.float32)
      nak = qc.A * len(qc.SUPPORT)
      for g in range(3):
        for n_in, n_out in ((3136, 512), (512, nak), (3136, 512), (512, len(qc.SUPPORT))):
          for n in (n_in, n_out):
            got = fake_tn(None, -2.0, 2.0, shape=[1, n])
            assert got.shape == (1, n) and got.dtype == np.float32
      assert not queue and len(used) == 24
      for i, u in enumerate(used):
        ref = np.asarray(inp['noises'][i // 8][NOISE_ORDER[i % 8]], np.float32)
        assert np.allclose(u, ref, rtol=2e-7, atol=0), (i, NOISE_ORDER[i % 8])
      # a network that drew in another order (value head first) is refused at its second draw
      bad, _q = noise_feeder(inp['noises'], [], lambda x, dt: np.asarray(x, dt), np.float32)
      bad(None, -2.0, 2.0, shape=[1, 3136])
      try:
        bad(None, -2.0, 2.0, shape=[1, len(qc.SUPPORT)])
        raise SystemExit('selfcheck: a wrong draw order was accepted')
      except AssertionError:
        pass
    if name == 'iqn':
      fake_st, tq = tau_feeder(inp['taus'], np.asarray)
      for t in inp['taus']:
        assert np.array_equal(fake_st(None, np.shape(t)), np.asarray(t, np.float32))
      assert not tq
    res = qc.oracle_step(name, qc.make_inputs(name))
    packed = qc.pack(res)
    assert 'g/conv1/w' in packed and packed['losses'].shape == (qc.B,)
  print('selfcheck OK (mapping for 7 agents, tree round trip, noise inversion, noise / tau draw '
        'order and shapes, pack)')


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--dqn_zoo', help='root of a google-deepmind/dqn_zoo checkout (the directory '
                                    'that contains the dqn_zoo package)')
  ap.add_argument('--agents', default='rainbow,dqn,double_q,prioritized,c51,qr,iqn')
  ap.add_argument('--selfcheck', action='store_true')
  args = ap.parse_args()
  if args.selfcheck:
    selfcheck()
    return
  if not args.dqn_zoo:
    ap.error('--dqn_zoo is required (or --selfcheck)')
  os.environ.setdefault('JAX_PLATFORM_NAME', 'cpu')
  failed = []
  for name in args.agents.split(','):
    try:
      out = run_reference(name, os.path.abspath(args.dqn_zoo))
    except Exception:  # pylint: disable=broad-except
      import traceback
      traceback.print_exc()
      failed.append(name)
      continue
    path = os.path.join(OUT_DIR, 'ref_qnet_%s.npz' % name)
    np.savez_compressed(path, **out)
    print('wrote %s  loss=%s gnorm=%.6g' % (path, out['loss'], out['gnorm']))
  if failed:
    raise SystemExit('FAILED: %s' % ', '.join(failed))


if __name__ == '__main__':
  main()
