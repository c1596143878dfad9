"""Replay golden-trace protocol, shared by the generator and by the tests.

One trace = fill, then K learner iterations of
  sample(B) -> update_priorities(ids, p_k) -> 4 x add(priority=max_seen)
which mirrors one `_learn()` every 4 agent decisions (SURVEY.md Appendix C,
reference rainbow/agent.py:148-149,181-198).

`drive_prioritized` / `drive_uniform` run the protocol against ANY object that
exposes the reference's replay surface, so the same driver produces the golden
trace from the reference module, replays it against the oracle, and replays it
against the HIP-backed classes.
"""

import collections

import numpy as np

Item = collections.namedtuple('Item', ['a', 'b'])

# (name, capacity, fill, batch, steps, seed, priority_exponent, usp, normalize)
PRIORITIZED_CASES = [
    ('n7_rainbow', 7, 7, 5, 60, 1, 0.5, 1e-3, True),
    ('n8_wrap', 8, 8, 16, 60, 2, 0.5, 1e-3, True),
    ('n8_partial', 8, 5, 4, 40, 3, 1.0, 0.25, False),
    ('n64_exp0', 64, 64, 32, 50, 4, 0.0, 1e-3, True),
    ('n1000_rainbow', 1000, 1000, 32, 80, 1, 0.5, 1e-3, True),
    ('n1000_usp_half', 1000, 700, 32, 60, 5, 1.0, 0.5, True),
]
PRIORITIZED_BIG = ('n1m_rainbow', 1000000, 1000000, 32, 25, 1, 0.5, 1e-3, True)

# (name, capacity, fill, batch, steps, seed)
UNIFORM_CASES = [
    ('u7', 7, 7, 5, 40, 1),
    ('u8_partial', 8, 3, 4, 30, 2),
    ('u1000', 1000, 1000, 32, 60, 1),
]
UNIFORM_BIG = ('u1m', 1000000, 1000000, 32, 25, 1)


def beta_schedule(capacity):
  """IS-exponent schedule 0.4 -> 1.0 over 2*capacity adds (shape of
  rainbow/run_atari.py:180-188; exact constants are ours, fixed here)."""
  span = float(2 * capacity)

  def beta(t):
    frac = min(max(t, 0), span) / span
    return (1 - frac) * 0.4 + frac * 1.0

  return beta


def priority_stream(seed):
  rs = np.random.RandomState(seed + 1000)

  def draw(n):
    # heavy-tailed like replay_test.py:1122, clipped like rainbow/agent.py:194
    p = np.clip(np.abs(rs.standard_cauchy(n)), 0.0, 100.0)
    # sprinkle exact zeros: zero-priority leaves must never be returned by the
    # prioritized branch (replay_test.py:978-987).
    p[rs.uniform(size=n) < 0.05] = 0.0
    return p

  return draw


def drive_prioritized(replay, capacity, fill, batch, steps, seed, on_sample,
                      bulk_fill=None, snapshot_at=None, on_snapshot=None,
                      resume=None):
  """Runs the protocol.  `on_sample(k, ids, weights)` is called per step.

  `bulk_fill(replay, n)`, if given, must be equivalent to the n initial
  `add(Item(a=i, b=-i), 1.0)` calls (used for the 1M-item device fill).

  Checkpoint support: before the sample of step
  `snapshot_at`, `on_snapshot(replay, t, max_seen)` is called; if it returns an
  object, the protocol continues on THAT replay (a restored copy).  `resume =
  (k0, t, max_seen)` starts at step k0 on a replay restored from a snapshot
  taken there (no fill; the priority stream is advanced to step k0)."""
  draw = priority_stream(seed)
  max_seen = 1.0
  t = 0  # number of adds so far == id of the next item (replay.py:696).
  k0 = 0
  if resume is not None:
    k0, t, max_seen = resume
    for _ in range(k0):
      draw(batch)
  elif bulk_fill is not None:
    bulk_fill(replay, fill)
    t = fill
  else:
    for _ in range(fill):
      replay.add(Item(a=t, b=-t), 1.0)
      t += 1
  for k in range(k0, steps):
    if snapshot_at is not None and k == snapshot_at and resume is None:
      replay = on_snapshot(replay, t, max_seen) or replay
    _, ids, weights = replay.sample(batch)
    on_sample(k, np.asarray(ids), np.asarray(weights))
    p = draw(batch)
    replay.update_priorities(ids, p)
    max_seen = max(max_seen, float(p.max()))
    for _ in range(4):
      replay.add(Item(a=t, b=k), max_seen)
      t += 1


def drive_uniform(replay, capacity, fill, batch, steps, seed, on_sample,
                  bulk_fill=None, snapshot_at=None, on_snapshot=None,
                  resume=None):
  t = 0  # Item.a carries the item's id, so sampled ids can be read back.
  k0 = 0
  if resume is not None:
    k0, t = resume
  elif bulk_fill is not None:
    bulk_fill(replay, fill)
    t = fill
  else:
    for _ in range(fill):
      replay.add(Item(a=t, b=-t))
      t += 1
  for k in range(k0, steps):
    if snapshot_at is not None and k == snapshot_at and resume is None:
      replay = on_snapshot(replay, t) or replay
    s = replay.sample(batch)
    on_sample(k, s)
    for _ in range(4):
      replay.add(Item(a=t, b=k))
      t += 1


def f64_bits(x):
  return np.asarray(x, dtype=np.float64).view(np.uint64)


# ---- checkpoint-state fixtures (row f3): snapshot step per existing case ------
# (case name -> step before whose sample the reference's get_state() is frozen)
STATE_SNAPSHOTS = {'n8_wrap': 40, 'n8_partial': 20, 'n64_exp0': 30,
                   'n7_rainbow': 40}
UNIFORM_STATE_SNAPSHOTS = {'u7': 20, 'u8_partial': 10}


def pack_state(state, prioritized):
  """Reference get_state() dictionary -> flat dict of arrays (npz-friendly;
  list ORDER is kept, dicts are stored as (keys, values) in iteration order)."""
  out = {}
  out['t'] = np.int64(state['t'])
  out['storage_ids'] = np.array([k for k, _ in state['storage']], dtype=np.int64)
  out['storage_a'] = np.array([it.a for _, it in state['storage']], dtype=np.int64)
  out['storage_b'] = np.array([it.b for _, it in state['storage']], dtype=np.int64)
  d = state['distribution']

  def put_dict(name, m):
    out[name + '_keys'] = np.array(list(m.keys()), dtype=np.int64)
    out[name + '_vals'] = np.array(list(m.values()), dtype=np.int64)

  if prioritized:
    out['tree_size'] = np.int64(d['sum_tree']['size'])
    out['tree_first_leaf'] = np.int64(d['sum_tree']['first_leaf'])
    out['tree_storage_bits'] = f64_bits(d['sum_tree']['storage'])
    put_dict('id_to_index', d['id_to_index'])
    put_dict('index_to_id', d['index_to_id'])
    put_dict('active_indices_location', d['active_indices_location'])
    out['inactive_indices'] = np.array(d['inactive_indices'], dtype=np.int64)
    out['active_indices'] = np.array(d['active_indices'], dtype=np.int64)
  else:
    out['ids'] = np.array(d['ids'], dtype=np.int64)
    put_dict('id_to_index', d['id_to_index'])
  return out


def unpack_state(z, prioritized):
  """Inverse of pack_state: the dictionary the reference's set_state() takes."""
  storage = [(int(k), Item(int(a), int(b))) for k, a, b in
             zip(z['storage_ids'], z['storage_a'], z['storage_b'])]

  def get_dict(name):
    return {int(k): int(v) for k, v in zip(z[name + '_keys'], z[name + '_vals'])}

  if prioritized:
    dist = {
        'sum_tree': {'size': int(z['tree_size']),
                     'storage': z['tree_storage_bits'].view(np.float64).copy(),
                     'first_leaf': int(z['tree_first_leaf'])},
        'id_to_index': get_dict('id_to_index'),
        'index_to_id': get_dict('index_to_id'),
        'inactive_indices': [int(i) for i in z['inactive_indices']],
        'active_indices': [int(i) for i in z['active_indices']],
        'active_indices_location': get_dict('active_indices_location'),
    }
  else:
    dist = {'ids': [int(i) for i in z['ids']],
            'id_to_index': get_dict('id_to_index')}
  return {'storage': storage, 't': int(z['t']), 'distribution': dist}


def states_equal(got, want, prioritized):
  """Exact comparison of two reference-format state dictionaries (list order,
  dict contents, sum-tree bits).  Returns '' or a description of the first
  difference."""
  if int(got['t']) != int(want['t']):
    return 't: %s != %s' % (got['t'], want['t'])
  gs = [(int(k), tuple(int(x) for x in it)) for k, it in got['storage']]
  ws = [(int(k), tuple(int(x) for x in it)) for k, it in want['storage']]
  if gs != ws:
    return 'storage differs'
  g, w = got['distribution'], want['distribution']
  names = (['id_to_index', 'index_to_id', 'inactive_indices', 'active_indices',
            'active_indices_location'] if prioritized else ['ids', 'id_to_index'])
  for n in names:
    a, b = g[n], w[n]
    if isinstance(b, dict):
      if {int(k): int(v) for k, v in a.items()} != b:
        return n + ' differs'
    elif [int(x) for x in a] != [int(x) for x in b]:
      return n + ' differs'
  if prioritized:
    for n in ('size', 'first_leaf'):
      if int(g['sum_tree'][n]) != int(w['sum_tree'][n]):
        return 'sum_tree.%s differs' % n
    if not (f64_bits(g['sum_tree']['storage']) ==
            f64_bits(w['sum_tree']['storage'])).all():
      return 'sum_tree.storage bits differ'
  return ''
