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
                      bulk_fill=None):
  """Runs the protocol.  `on_sample(k, ids, weights)` is called per step.

  `bulk_fill(replay, n)`, if given, must be equivalent to the n initial
  `add(Item(a=i, b=-i), 1.0)` calls (used for the 1M-item device fill)."""
  draw = priority_stream(seed)
  max_seen = 1.0
  t = 0  # number of adds so far == id of the next item (replay.py:696).
  if bulk_fill is not None:
    bulk_fill(replay, fill)
    t = fill
  else:
    for _ in range(fill):
      replay.add(Item(a=t, b=-t), 1.0)
      t += 1
  for k in range(steps):
    _, ids, weights = replay.sample(batch)
    on_sample(k, np.asarray(ids), np.asarray(weights))
    p = draw(batch)
    replay.update_priorities(ids, p)
    max_seen = max(max_seen, float(p.max()))
    for _ in range(4):
      replay.add(Item(a=t, b=k), max_seen)
      t += 1


def drive_uniform(replay, capacity, fill, batch, steps, seed, on_sample,
                  bulk_fill=None):
  t = 0  # Item.a carries the item's id, so sampled ids can be read back.
  if bulk_fill is not None:
    bulk_fill(replay, fill)
    t = fill
  else:
    for _ in range(fill):
      replay.add(Item(a=t, b=-t))
      t += 1
  for k in range(steps):
    s = replay.sample(batch)
    on_sample(k, s)
    for _ in range(4):
      replay.add(Item(a=t, b=k))
      t += 1


def f64_bits(x):
  return np.asarray(x, dtype=np.float64).view(np.uint64)
