"""CPU oracle for the replay half of the hot path (TEST INFRASTRUCTURE ONLY).

A NumPy/Python restatement of the algorithms in the reference's
`dqn_zoo/replay.py`.  It exists so that the HIP path can be checked on the GPU
box, where /root/reference is not present.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it;
the product package `dqn_zoo_amd` never does.

Parity status: PINNED.  `tests/test_oracle_replay.py` checks this file against
(a) the reference module itself, imported unmodified via `oracle/ref_loader.py`
when /root/reference exists, (b) the golden traces in `tests/golden/` generated
from the reference by `tests/golden/gen_replay_golden.py`, and (c) the
known-answer tables of `replay_test.py:939-987`.

The data structures are deliberately the *general* ones (explicit free stack,
explicit swap-remove position list) and NOT the closed forms the HIP path
uses, so that the closed forms are tested against an independent model.

Reference sites restated here:
  SumTreeOracle            <- replay.py:246-426  (set 278-290, query 406-426)
  PrioritizedSamplerOracle <- replay.py:429-651  (add 475-507, remove 509-534,
                                                  sample 547-583)
  power_zero_safe          <- replay.py:203-208
  is_weights               <- replay.py:211-243
  UniformReplayOracle      <- replay.py:44-200
  PrioritizedReplayOracle  <- replay.py:654-768
"""

import numpy as np


def power_zero_safe(base, exponent):
  """`base ** exponent` but 0 ** 0 == 0   (replay.py:203-208).

  NOTE: with a Python-float exponent of exactly 0.5 NumPy's `**` takes its
  scalar fast path and evaluates sqrt() (correctly rounded); for other
  exponents it calls a vectorised pow whose last bit is CPU dependent.
  """
  base = np.asarray(base)
  return np.where(base == 0.0, 0.0, base**exponent)


def is_weights(probabilities, uniform_probability, exponent, normalize):
  """Importance-sampling weights (replay.py:211-243)."""
  if not 0.0 <= exponent <= 1.0:
    raise ValueError('Require 0 <= exponent <= 1.')
  if not 0.0 <= uniform_probability <= 1.0:
    raise ValueError('Expected 0 <= uniform_probability <= 1.')
  w = (uniform_probability / probabilities) ** exponent
  if normalize:
    w /= np.max(w)
  if not np.isfinite(w).all():
    raise ValueError('Weights are not finite: %s.' % w)
  return w


class SumTreeOracle:
  """Binary-heap sum tree in one float64 array; root at 1, leaves at [cap, 2cap)."""

  def __init__(self, size=0):
    self.size = 0
    self.cap = 0
    self.node = np.zeros(0, np.float64)
    if size:
      self.resize(size)

  # -- construction ---------------------------------------------------------
  def _rebuild(self, leaf_values):
    n = len(leaf_values)
    self.node[self.cap:self.cap + n] = leaf_values
    self.node[self.cap + n:] = 0.0
    for i in range(self.cap - 1, 0, -1):
      self.node[i] = self.node[2 * i] + self.node[2 * i + 1]
    self.node[0] = 0.0

  def _reshape(self, size, values):
    if size < self.size:
      keep = self.leaves()[:size].copy() if values is None else values
      self.size = size
      self._rebuild(keep)
    elif size <= self.cap:
      self.size = size
      if values is not None:
        self._rebuild(values)
    else:
      keep = self.leaves().copy() if values is None else values
      cap = 1
      while cap < size:
        cap *= 2
      self.node = np.empty(2 * cap, np.float64)
      self.cap = cap
      self.size = size
      self._rebuild(keep)

  def resize(self, size):
    self._reshape(size, None)

  def set_all(self, values):
    values = np.asarray(values, dtype=np.float64)
    if not np.isfinite(values).all() or (values < 0.0).any():
      raise ValueError('Values must be finite positive numbers.')
    self._reshape(len(values), values)

  # -- access ---------------------------------------------------------------
  def leaves(self):
    return self.node[self.cap:self.cap + self.size]

  def root(self):
    return self.node[1] if self.size > 0 else np.nan

  def get(self, indices):
    indices = np.asarray(indices)
    if not ((0 <= indices) & (indices < self.size)).all():
      raise IndexError('index out of range, expect 0 <= index < %s' % self.size)
    return self.leaves()[indices]

  def set(self, indices, values):
    values = np.asarray(values)
    if not np.isfinite(values).all() or (values < 0.0).any():
      raise ValueError('value must be finite and positive.')
    indices = np.asarray(indices)
    # Fancy assignment: for duplicate indices the LAST value wins.
    self.leaves()[indices] = values
    node = self.node
    for leaf in indices + self.cap:
      p = int(leaf) >> 1
      while p >= 1:
        node[p] = node[2 * p] + node[2 * p + 1]
        p >>= 1

  def query_one(self, target):
    if not 0.0 <= target < self.root():
      raise ValueError('Require 0 <= target < total sum.')
    node = self.node
    i = 1
    while i < self.cap:
      left = node[2 * i]
      if target < left:
        i = 2 * i
      else:
        target -= left
        i = 2 * i + 1
    return i - self.cap

  def query(self, targets):
    return [self.query_one(t) for t in targets]

  def consistent(self):
    for i in range(1, self.cap):
      if self.node[i] != self.node[2 * i] + self.node[2 * i + 1]:
        return False
    return True


class PrioritizedSamplerOracle:
  """Fixed-capacity prioritized id sampler (the only mode reachable through
  PrioritizedTransitionReplay, which pins min_capacity=max_capacity=capacity,
  replay.py:678-684)."""

  def __init__(self, capacity, priority_exponent, uniform_sample_probability,
               random_state):
    if priority_exponent < 0.0:
      raise ValueError('Require priority_exponent >= 0.')
    if not 0.0 <= uniform_sample_probability <= 1.0:
      raise ValueError('Require 0 <= uniform_sample_probability <= 1.')
    self.exponent = priority_exponent
    self.usp = uniform_sample_probability
    self.rs = random_state
    self.capacity = capacity
    self.tree = SumTreeOracle(capacity)
    self.free = list(range(capacity))   # stack; pop() takes from the END.
    self.active = []                    # swap-remove list of tree indices.
    self.where = {}                     # tree index -> position in `active`.
    self.index_of = {}                  # id -> tree index
    self.id_of = {}                     # tree index -> id

  @property
  def size(self):
    return len(self.index_of)

  def add(self, new_id, priority):
    if new_id in self.index_of:
      raise IndexError('ID %d already exists.' % new_id)
    if self.size + 1 > self.capacity:
      raise ValueError('Cannot add IDs as max capacity would be exceeded.')
    ti = self.free.pop()
    self.where[ti] = len(self.active)
    self.active.append(ti)
    self.index_of[new_id] = ti
    self.id_of[ti] = new_id
    self.tree.set([ti], power_zero_safe([priority], self.exponent))

  def remove(self, old_id):
    ti = self.index_of.pop(old_id)   # KeyError for unknown id, as the reference.
    del self.id_of[ti]
    j = self.where[ti]
    last = self.active[-1]
    self.active[j] = last
    self.where[last] = j
    self.active.pop()
    del self.where[ti]
    self.free.append(ti)
    self.tree.set([ti], np.zeros(1))

  def update(self, ids, priorities):
    tis = []
    for i in ids:
      if i not in self.index_of:
        raise IndexError('ID %d does not exist.' % i)
      tis.append(self.index_of[i])
    self.tree.set(tis, power_zero_safe(priorities, self.exponent))

  def sample(self, size):
    if self.size == 0:
      raise RuntimeError('No IDs to sample.')
    # RNG draw order is part of the contract (replay.py:551-566).
    pos = self.rs.randint(self.size, size=size)
    uni = [self.active[j] for j in pos]
    root = self.tree.root()
    if root == 0.0:
      pri = uni
    else:
      targets = self.rs.uniform(size=size) * root
      pri = np.asarray(self.tree.query(targets))
    pick_uniform = self.rs.uniform(size=size) < self.usp
    tis = np.where(pick_uniform, uni, pri)
    up = np.asarray(1.0 / self.size)
    leaf = self.tree.get(tis)
    if root == 0.0:
      pp = np.full_like(leaf, fill_value=up)
    else:
      pp = leaf / root
    probs = (1.0 - self.usp) * pp + self.usp * up
    ids = np.array([self.id_of[int(t)] for t in tis], dtype=np.int64)
    return ids, probs


class _Store:
  """FIFO item store keyed by monotonically increasing id."""

  def __init__(self, capacity):
    self.capacity = capacity
    self.items = {}
    self.oldest = 0
    self.t = 0

  @property
  def size(self):
    return len(self.items)

  def evict_if_full(self):
    if self.size == self.capacity:
      old = self.oldest
      del self.items[old]
      self.oldest += 1
      return old
    return None

  def put(self, item):
    self.items[self.t] = item
    self.t += 1
    return self.t - 1

  def stack(self, ids, structure):
    rows = [self.items[int(i)] for i in ids]
    cols = [np.stack(c, axis=0) for c in zip(*rows)]
    return type(structure)(*cols)


class UniformReplayOracle:
  """TransitionReplay restated (replay.py:120-200 with 44-117)."""

  def __init__(self, capacity, structure, random_state):
    self.structure = structure
    self.rs = random_state
    self.store = _Store(capacity)
    self.ids = []       # swap-remove list of ids
    self.where = {}

  @property
  def size(self):
    return self.store.size

  def add(self, item):
    old = self.store.evict_if_full()
    if old is not None:
      j = self.where.pop(old)
      last = self.ids.pop()
      if last != old:
        self.ids[j] = last
        self.where[last] = j
    new = self.store.put(item)
    self.where[new] = len(self.ids)
    self.ids.append(new)

  def sample_ids(self, size):
    pos = self.rs.randint(len(self.ids), size=size)
    return np.array([self.ids[j] for j in pos], dtype=np.int64)

  def sample(self, size):
    return self.store.stack(self.sample_ids(size), self.structure)


class PrioritizedReplayOracle:
  """PrioritizedTransitionReplay restated (replay.py:654-768)."""

  def __init__(self, capacity, structure, priority_exponent,
               importance_sampling_exponent, uniform_sample_probability,
               normalize_weights, random_state):
    self.structure = structure
    self.store = _Store(capacity)
    self.dist = PrioritizedSamplerOracle(
        capacity, priority_exponent, uniform_sample_probability, random_state)
    self.beta = importance_sampling_exponent
    self.normalize = normalize_weights

  @property
  def size(self):
    return self.store.size

  @property
  def t(self):
    return self.store.t

  def add(self, item, priority):
    old = self.store.evict_if_full()
    if old is not None:
      self.dist.remove(old)
    self.dist.add(self.store.t, priority)
    self.store.put(item)

  def bulk_fill(self, n, item_fn, priority=1.0):
    """State after `n` add(item_fn(i), priority) calls on an EMPTY replay,
    built with array operations (benchmark set-up for 1e6 items; checked against
    the sequential path by tests/test_oracle_replay.py)."""
    assert self.store.t == 0 and n <= self.store.capacity
    cap = self.store.capacity
    d = self.dist
    self.store.items = {i: item_fn(i) for i in range(n)}
    self.store.t = n
    d.free = list(range(cap - n))
    d.active = list(range(cap - 1, cap - 1 - n, -1))
    d.where = {ti: j for j, ti in enumerate(d.active)}
    d.index_of = {i: cap - 1 - i for i in range(n)}
    d.id_of = {cap - 1 - i: i for i in range(n)}
    tree = d.tree
    leaf = power_zero_safe(np.full(n, priority, np.float64), d.exponent)
    tree.node[tree.cap + cap - n:tree.cap + cap] = leaf
    lvl = tree.cap >> 1
    while lvl >= 1:
      i = np.arange(lvl, 2 * lvl)
      tree.node[i] = tree.node[2 * i] + tree.node[2 * i + 1]
      lvl >>= 1

  def sample_ids(self, size):
    ids, probs = self.dist.sample(size)
    w = is_weights(probs, 1.0 / self.size, self.beta(self.store.t),
                   self.normalize)
    return ids, probs, w

  def sample(self, size):
    ids, _, w = self.sample_ids(size)
    return self.store.stack(ids, self.structure), ids, w

  def update_priorities(self, ids, priorities):
    self.dist.update(ids, np.asarray(priorities))
