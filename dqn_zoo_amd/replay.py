"""HBM-resident experience replay with the surface of dqn_zoo's `replay.py`.

Drop-in classes (same names, constructor arguments, methods, return types and
error behaviour as the reference) whose storage, sampling and priority
bookkeeping live on one MI355X and run as hand-written HIP kernels behind the
C ABI of `include/dqnzoo_hip.h`:

  TransitionReplay             ref: dqn_zoo/replay.py:120-200
  PrioritizedTransitionReplay  ref: dqn_zoo/replay.py:654-768
  SumTree                      ref: dqn_zoo/replay.py:246-426
  Transition, importance_sampling_weights, TransitionAccumulator,
  NStepTransitionAccumulator   ref: replay.py:36-41, 211-243, 771-892

Design (DESIGN.md has the long form):
  * storage is one `[capacity, *field_shape]` device array per field of the
    replay structure; slot(id) = id mod capacity, so eviction of the oldest
    item is an overwrite and no id<->slot tables exist;
  * the reference's swap-remove position lists and free-index stack have
    closed forms under its only usage pattern, evaluated on the
    device: tree_index(id) = capacity-1-(id mod capacity);
  * randomness stays the injected `np.random.RandomState`: the host draws in
    the reference's exact order and uploads the raw draws, so ids are bit-exact
    for a fixed seed;
  * two call styles: the reference's (`sample()` -> NumPy, synchronous, exact
    float64 importance weights computed with NumPy on the 32 probabilities) and
    a pipelined one for the on-device learner (`sample_device()`, nothing
    leaves HBM, no host sync).

There is no CPU fallback: without libdqnzoo_hip.so and a GPU these classes
raise at construction.
"""

import collections
import ctypes
import typing
from typing import Any, Callable, Generic, Iterable, Mapping, Optional, Sequence, Tuple, TypeVar

import numpy as np
import torch

from dqn_zoo_amd import _lib

ReplayStructure = TypeVar('ReplayStructure', bound=Tuple[Any, ...])
COMPACT_FORMAT = 'dqn_zoo_amd.replay.compact.v1'


class Transition(typing.NamedTuple):
  s_tm1: Optional[np.ndarray]
  a_tm1: Optional[int]
  r_t: Optional[float]
  discount_t: Optional[float]
  s_t: Optional[np.ndarray]


def _require_gpu(device):
  if not torch.cuda.is_available():
    raise _lib.HipLibraryError(
        'dqn_zoo_amd.replay needs an AMD GPU (torch.cuda.is_available() is '
        'False); there is no CPU fallback.')
  return torch.device('cuda', torch.cuda.current_device()) if device is None \
      else torch.device(device)


def _next_pow2(n):
  c = 1
  while c < n:
    c *= 2
  return c


class ChainTimeoutError(RuntimeError):
  """A multi-role learner launch gave up on one of its in-launch seams (no reference
  counterpart).  The step that raised it was VOID: its optimiser, optax count and priority
  write-back launches read the sticky word and changed nothing, so a caller may clear the
  flag (`RainbowLearner.check_status()`), fall back to separate launches and carry on."""


def _raise_status(bits):
  """Maps sticky device status bits to the reference's exceptions."""
  if bits & _lib.ST_CHAIN_TIMEOUT:
    raise ChainTimeoutError('a multi-role learner launch timed out on an in-launch seam '
                            '(DZ_ST_CHAIN_TIMEOUT): the step that raised it was skipped')
  if bits & _lib.ST_BAD_VALUE:
    raise ValueError('value must be finite and positive.')
  if bits & _lib.ST_BAD_TARGET:
    raise ValueError('Require 0 <= target < total sum.')
  if bits & _lib.ST_BAD_INDEX:
    raise IndexError('index out of range')
  if bits & _lib.ST_NONFINITE_WEIGHT:
    raise ValueError('Weights are not finite.')
  if bits & _lib.ST_ZERO_ROOT:
    raise RuntimeError('pipelined sampling met an all-zero sum tree')


class _Status:
  """Sticky status word shared by the kernels of one object.

  The word lives in PINNED HOST memory (device-visible under the same address): kernels
  only ever touch it on the error path -- one system-scope atomic OR, no kernel reads it --
  so the host can look at it whenever it likes without enqueueing or waiting for anything.
  `poll()` is that look (a plain load of host memory; raises as soon as a flag has landed,
  i.e. at most a few enqueued steps after the offending kernel ran); `check()` first waits
  for everything enqueued on the device and is therefore definitive."""

  def __init__(self, device):
    self._device = device
    self.word = torch.zeros(1, dtype=torch.int32).pin_memory()
    self._np = self.word.numpy()

  def _take(self):
    bits = int(self._np[0])
    if bits:
      torch.cuda.synchronize(self._device)   # let every flag of the failing step land
      bits = int(self._np[0])
      self._np[0] = 0
      _raise_status(bits)

  def poll(self):
    if self._np[0]:
      self._take()

  def check(self):
    torch.cuda.synchronize(self._device)
    self._take()


# --------------------------------------------------------------------------- #
#  Host-side closed forms (mirrors of the device code; also used by ids()).
# --------------------------------------------------------------------------- #
def position_to_id(pos, t, capacity):
  """ids held at `pos` of the reference's swap-remove list (replay.py:52-82)."""
  pos = np.asarray(pos, dtype=np.int64)
  if t <= capacity:
    return pos.copy()
  if capacity == 1:
    return np.full_like(pos, t - 1)
  base = t - capacity
  out = base + np.mod(pos - base, capacity - 1)
  return np.where(pos == capacity - 1, t - 1, out)


def tree_index_of_id(ids, capacity):
  """Sum-tree index of an id (free stack popped from the end, replay.py:499)."""
  return capacity - 1 - np.mod(np.asarray(ids, dtype=np.int64), capacity)


def uniform_distribution_state(t, size, capacity):
  """`UniformDistribution.get_state()` of the reference (replay.py:95-100) for a
  replay that has seen `t` adds: {'ids': swap-remove list, 'id_to_index': dict},
  materialised from the closed form."""
  ids = position_to_id(np.arange(size, dtype=np.int64), t, capacity)
  return {'ids': [int(i) for i in ids],
          'id_to_index': {int(i): j for j, i in enumerate(ids)}}


def prioritized_distribution_state(t, size, capacity, cap_pow2, tree_storage):
  """`PrioritizedDistribution.get_state()` of the reference (replay.py:606-615)
  materialised from the closed forms: tree index of id i
  is N-1-(i mod N); the free stack `list(range(N))` is popped from the end, so
  while filling it still holds [0, N-size); `_active_indices` is the swap-remove
  list of R1 holding tree indices."""
  ids = np.arange(t - size, t, dtype=np.int64)
  idx = tree_index_of_id(ids, capacity)
  active_ids = position_to_id(np.arange(size, dtype=np.int64), t, capacity)
  active = tree_index_of_id(active_ids, capacity)
  return {
      'sum_tree': {'size': int(capacity), 'storage': tree_storage,
                   'first_leaf': int(cap_pow2)},
      'id_to_index': {int(i): int(j) for i, j in zip(ids, idx)},
      'index_to_id': {int(j): int(i) for i, j in zip(ids, idx)},
      'inactive_indices': list(range(capacity - size)),
      'active_indices': [int(j) for j in active],
      'active_indices_location': {int(j): k for k, j in enumerate(active)},
  }


def _check_storage_ids(storage, t, capacity):
  """A reference `storage` list must be the FIFO window [t-size, t) in order
  (replay.py:141-150 adds at the end and evicts from the front)."""
  size = len(storage)
  if size > capacity or size > t:
    raise ValueError('storage holds %d items: more than capacity or t' % size)
  ids = [int(k) for k, _ in storage]
  if ids != list(range(t - size, t)):
    raise ValueError(
        'storage ids are not the FIFO window [t - size, t): this replay only '
        'supports states produced by add() (one item at a time, oldest evicted)')
  return size


def _same_table(got, want, name):
  if isinstance(want, dict):
    ok = isinstance(got, dict) and {int(k): int(v) for k, v in got.items()} == want
  else:
    ok = [int(x) for x in got] == want
  if not ok:
    raise ValueError(
        "state['distribution'][%r] does not match the add-one/evict-oldest "
        'bookkeeping this replay implements in closed form (replay.py:475-534); '
        'states built through other usage patterns are not supported' % name)


# --------------------------------------------------------------------------- #
#  Device ring store
# --------------------------------------------------------------------------- #
class _FieldRing:
  """One `[capacity, *shape]` device array per structure field."""

  def __init__(self, capacity, structure, device):
    self.capacity = capacity
    self.structure = structure
    self.device = device
    self.fields = None      # list[torch.Tensor]
    self.np_dtypes = None
    self.shapes = None

  STAGE_DEPTH = 8   # in-flight host->device row uploads of insert()

  def allocate_like(self, item):
    fields, dts, shapes = [], [], []
    for x in item:
      if isinstance(x, torch.Tensor):
        a = np.zeros(tuple(x.shape), torch.empty(0, dtype=x.dtype).numpy().dtype)
      else:
        a = np.asarray(x)
      dts.append(a.dtype)
      shapes.append(a.shape)
      tdt = torch.from_numpy(np.zeros(1, a.dtype)).dtype
      fields.append(torch.empty((self.capacity,) + a.shape, dtype=tdt,
                                device=self.device))
    self.fields, self.np_dtypes, self.shapes = fields, dts, shapes
    self._stage = None

  def allocate(self, shapes, np_dtypes):
    self.allocate_like([np.zeros(s, d) for s, d in zip(shapes, np_dtypes)])

  def _staging(self):
    """Pinned host + device staging rows for array fields given as host arrays
    (a ring: slot k is reused only after its previous upload completed)."""
    if getattr(self, '_stage', None) is None:
      pins, devs = [], []
      for f, shp in zip(self.fields, self.shapes):
        if len(shp) == 0:
          pins.append(None); devs.append(None)
        else:
          pins.append(torch.empty((self.STAGE_DEPTH,) + tuple(shp), dtype=f.dtype,
                                  pin_memory=True))
          devs.append(torch.empty((self.STAGE_DEPTH,) + tuple(shp), dtype=f.dtype,
                                  device=self.device))
      self._stage = (pins, devs, [None] * self.STAGE_DEPTH)
      self._stage_pos = 0
      self._insert_arr = (_lib.InsertField * len(self.fields))()
      # destination and row size never change: filled once, not per add
      for i, f in enumerate(self.fields):
        self._insert_arr[i].dst = f.data_ptr()
        self._insert_arr[i].row_bytes = f[0].numel() * f.element_size()
      self._tshapes = [tuple(s) for s in self.shapes]
      # the steady state of an agent loop (insert_fields' fast path): scalar fields given as
      # NumPy scalars of exactly the field's dtype, tensor fields at addresses validated before
      self._insert_items = [self._insert_arr[i] for i in range(len(self.fields))]   # (views of the array's elements)
      self._scalar_type = [np.dtype(d).type if len(shp) == 0 else None
                           for d, shp in zip(self.np_dtypes, self.shapes)]
      self._imm_view = [np.dtype('u%d' % np.dtype(d).itemsize) if len(shp) == 0 else None
                        for d, shp in zip(self.np_dtypes, self.shapes)]
      self._numel = [int(np.prod(shp, dtype=np.int64)) for shp in self.shapes]
      self._ok_ptrs = set()
    return self._stage

  def insert_fields(self, item):
    """Fills the dz_insert_field_t array for one item: device tensors are used
    in place, scalars travel as immediates, host arrays go through the pinned
    staging ring (one async H2D each)."""
    if self.fields is None:
      self.allocate_like(item)
    pins, devs, events = self._staging()
    arr = self._insert_arr
    # fast path (every add of an agent loop after the first few): nothing to validate or stage
    items, stype, ok = self._insert_items, self._scalar_type, self._ok_ptrs
    i = 0
    for x in item:
      t = stype[i]
      if t is not None:
        if type(x) is not t:
          break
        items[i].imm = int(x.view(self._imm_view[i]))
      else:
        if type(x) is not torch.Tensor:
          break
        f = self.fields[i]
        ptr = x.data_ptr()
        if ((i, ptr) not in ok or x.dtype != f.dtype or x.numel() != self._numel[i] or
            not x.is_contiguous()):
          break
        items[i].src_row = ptr
        items[i].imm = 0
      i += 1
    else:
      return arr, i
    k = None
    for i, (f, x, dt, shp) in enumerate(zip(self.fields, item, self.np_dtypes,
                                            self._tshapes)):
      if isinstance(x, torch.Tensor):
        on_dev = x.device == f.device or (x.device.type == 'cpu' and x.is_pinned())
        if (not on_dev or x.dtype != f.dtype or
            tuple(x.shape) != shp or not x.is_contiguous()):
          raise ValueError('device field %d: need a contiguous %s%s tensor on %s (or pinned host memory)'
                           % (i, f.dtype, tuple(shp), f.device))
        arr[i].src_row = x.data_ptr()
        arr[i].imm = 0
        if len(self._ok_ptrs) < 4096:   # (validated: device, dtype, shape, contiguity)
          self._ok_ptrs.add((i, x.data_ptr()))
        continue
      a = np.asarray(x, dtype=dt)
      if a.shape != shp:
        raise ValueError('replay item field has shape %s, expected %s' %
                         (a.shape, shp))
      if a.ndim == 0:
        arr[i].src_row = None
        arr[i].imm = int.from_bytes(a.tobytes(), 'little')
        continue
      if k is None:
        k = self._stage_pos % self.STAGE_DEPTH
        self._stage_pos += 1
        if events[k] is not None:
          events[k].synchronize()
      pins[i][k].copy_(torch.from_numpy(np.ascontiguousarray(a)))
      devs[i][k].copy_(pins[i][k], non_blocking=True)
      arr[i].src_row = devs[i][k].data_ptr()
      arr[i].imm = 0
    if k is not None:
      if events[k] is None:
        events[k] = torch.cuda.Event()
      events[k].record(_lib.current_stream(self.device))
    return arr, len(self.fields)

  def read(self, slot):
    return type(self.structure)(*[
        (f[slot].cpu().numpy() if f.dim() > 1 else f[slot].cpu().numpy()[()])
        for f in self.fields])

  def gather(self, ids_dev, batch, stream):
    """dst[b] = field[ids[b] mod capacity] for all fields, one launch."""
    outs = [torch.empty((batch,) + tuple(f.shape[1:]), dtype=f.dtype,
                        device=self.device) for f in self.fields]
    n = len(self.fields)
    arr = (_lib.FieldDesc * n)()
    for i, (f, o) in enumerate(zip(self.fields, outs)):
      arr[i].src = f.data_ptr()
      arr[i].dst = o.data_ptr()
      arr[i].row_bytes = f[0].numel() * f.element_size()
    _lib.check(_lib.load().dz_replay_gather(arr, n, ids_dev.data_ptr(), batch,
                                            self.capacity, stream),
               'dz_replay_gather')
    return outs


class _ReplayBase(Generic[ReplayStructure]):

  def __init__(self, capacity, structure, random_state, encoder, decoder,
               device):
    if capacity <= 0:
      raise ValueError('capacity must be positive')
    _lib.load()
    self._device = _require_gpu(device)
    self._capacity = int(capacity)
    self._structure = structure
    self._random_state = random_state
    # encoder/decoder (snappy state compression in the reference,
    # rainbow/run_atari.py:190-206) are accepted for signature compatibility;
    # an HBM-resident store keeps raw rows, so they are not applied.
    self._encoder = encoder
    self._decoder = decoder
    self._ring = _FieldRing(self._capacity, structure, self._device)
    self._t = 0      # items ever added == id of the next item.
    self._size = 0
    self._ins_args = self._ins_key = None   # (dz_replay_insert_v's patched argument struct)
    self._status = _Status(self._device)

  # -- shared surface --------------------------------------------------------
  @property
  def size(self) -> int:
    return self._size

  @property
  def capacity(self) -> int:
    return self._capacity

  @property
  def insertions(self) -> int:
    """Total number of items ever added (the reference's `_t`): a prepared sample
    (`prepare_next_sample`) is valid exactly while this has not moved."""
    return self._t

  def ids(self) -> Iterable[int]:
    """IDs of stored items, oldest first (ref: replay.py:165-167)."""
    return range(self._t - self._size, self._t)

  def get(self, ids: Sequence[int]) -> Iterable[ReplayStructure]:
    """Retrieves items by id as host values (ref: replay.py:152-155)."""
    for i in ids:
      i = int(i)
      if not self._t - self._size <= i < self._t:
        raise KeyError(i)
      yield self._ring.read(i % self._capacity)

  def _stream(self):
    return _lib.stream_ptr(self._device)

  def _store(self, item, node=None, cap_pow2=0, priority_d=None, exponent=0.0,
             status=None):
    """One launch: the item's rows into slot `t mod capacity` and, if `node` is
    given, its sum-tree leaf from the device priority (dz_replay_insert)."""
    arr, n = self._ring.insert_fields(item)
    # the arguments live in a struct that is patched, not re-marshalled (dz_replay_insert_v):
    # once per frame of every agent loop
    key = (ctypes.addressof(arr), n, node, cap_pow2, priority_d, exponent, status)
    a = self._ins_args
    if key != self._ins_key:
      if a is None:
        a = self._ins_args = _lib.ReplayInsertArgs()
        self._ins_ref = ctypes.byref(a)
        self._ins_fn = _lib.load().dz_replay_insert_v
      a.fields = ctypes.cast(arr, ctypes.POINTER(_lib.InsertField))
      a.num_fields, a.capacity, a.node, a.cap_pow2 = n, self._capacity, node, cap_pow2
      a.priority_h, a.priority_d, a.exponent, a.status = 0.0, priority_d, exponent, status
      self._ins_key = key
    a.t = self._t
    rc = self._ins_fn(self._ins_ref, self._stream())
    if rc:
      _lib.check(rc, 'dz_replay_insert')
    self._t += 1
    self._size = min(self._size + 1, self._capacity)

  def _to_host(self, tensors):
    return type(self._structure)(*[x.cpu().numpy() for x in tensors])

  def _live_rows_host(self):
    """Host arrays [size, *shape] per field, oldest item first."""
    if self._ring.fields is None or self._size == 0:
      return None
    lo = (self._t - self._size) % self._capacity
    first = min(self._size, self._capacity - lo)
    out = []
    for f in self._ring.fields:
      a = f[lo:lo + first].cpu().numpy()
      if first < self._size:
        a = np.concatenate([a, f[:self._size - first].cpu().numpy()], axis=0)
      out.append(a)
    return out

  def _reference_storage(self):
    """`list(self._storage.items())` of the reference (replay.py:182,751):
    [(id, item)] oldest first, scalar fields as Python scalars."""
    rows = self._live_rows_host()
    if rows is None:
      return []
    make = type(self._structure)
    t0 = self._t - self._size
    return [(t0 + k, make(*[(a[k].item() if a.ndim == 1 else a[k]) for a in rows]))
            for k in range(self._size)]

  def _load_rows(self, rows, t, size):
    """Host arrays [size, *shape] per field (oldest first) into the ring."""
    self._t, self._size = int(t), int(size)
    if rows is None or size == 0:
      return
    self._allocate_fields([a.shape[1:] for a in rows], [a.dtype for a in rows])
    lo = (self._t - self._size) % self._capacity
    first = min(self._size, self._capacity - lo)
    for dst, a in zip(self._ring.fields, rows):
      a = np.ascontiguousarray(a)
      dst[lo:lo + first].copy_(torch.from_numpy(a[:first]))
      if first < self._size:
        dst[:self._size - first].copy_(torch.from_numpy(a[first:]))

  def _load_reference_storage(self, storage, t):
    size = _check_storage_ids(storage, int(t), self._capacity)
    rows = None
    if size:
      nf = len(storage[0][1])
      rows = [np.stack([np.asarray(item[i]) for _, item in storage])
              for i in range(nf)]
    self._load_rows(rows, t, size)

  def _compact_state(self, where):
    """Native state: only the live rows; `where` = 'host' (NumPy) or 'device'
    (clones in HBM: a 1M-transition store snapshots in tens of ms, and 288 GB
    holds the store plus a snapshot)."""
    if where == 'device':
      fields = None if self._ring.fields is None else [
          f.clone() for f in self._ring.fields]
      return {'format': COMPACT_FORMAT, 't': self._t, 'size': self._size,
              'layout': 'ring', 'fields': fields}
    return {'format': COMPACT_FORMAT, 't': self._t, 'size': self._size,
            'layout': 'oldest_first', 'fields': self._live_rows_host()}

  def _load_compact(self, state):
    if state.get('layout', 'ring') == 'ring':   # full [capacity, ...] arrays
      self._t, self._size = int(state['t']), int(state['size'])
      if state['fields'] is not None:
        fs = [f if isinstance(f, torch.Tensor) else torch.from_numpy(np.asarray(f))
              for f in state['fields']]
        self._allocate_fields(
            [tuple(f.shape[1:]) for f in fs],
            [np.dtype(str(f.dtype).replace('torch.', '')) for f in fs])
        for dst, src in zip(self._ring.fields, fs):
          dst.copy_(src)
    else:
      self._load_rows(state['fields'], state['t'], state['size'])

  def check_status(self) -> None:
    """Synchronises and raises what the reference would have raised at the
    offending call (ValueError for NaN/inf/negative priorities or non-finite
    importance weights, replay.py:233-242,281-282) if any pipelined kernel
    flagged it in the sticky status word since the last check (definitive: waits for
    everything enqueued).  The agents call this at every target-network sync and the
    non-blocking `poll_status()` at every learner step."""
    self._status.check()

  def poll_status(self) -> None:
    """The same without waiting for the device: a load of the (pinned host) status
    word.  Raises once a flag written by an already-executed kernel has landed; costs
    nothing otherwise -- the agents call it at every learner step, so a diverged run
    stops a few enqueued steps after the offending kernel instead of at the next
    target-network sync."""
    self._status.poll()

  def _allocate_fields(self, shapes, np_dtypes):
    """(Re-)allocates the field arrays; cached sample slots hold raw pointers
    into the old arrays and must go with them."""
    self._ring.allocate(shapes, np_dtypes)
    self._sample_ring = None

  def bulk_fill(self, fields: Sequence[torch.Tensor]) -> int:
    """Appends `n` items given as device tensors `[n, *shape]` per field
    (synthetic-benchmark fill; equivalent to n add() calls on the storage)."""
    n = int(fields[0].shape[0])
    if self._ring.fields is None:
      self._ring.allocate([tuple(f.shape[1:]) for f in fields],
                          [np.dtype(str(f.dtype).replace('torch.', ''))
                           for f in fields])
    if n > self._capacity:
      raise ValueError('bulk_fill of more than capacity items')
    start = self._t % self._capacity
    first = min(n, self._capacity - start)
    for dst, src in zip(self._ring.fields, fields):
      dst[start:start + first].copy_(src[:first])
      if first < n:
        dst[:n - first].copy_(src[first:])
    self._t += n
    self._size = min(self._size + n, self._capacity)
    return n


class TransitionReplay(_ReplayBase):
  """Uniform replay with FIFO eviction (ref: dqn_zoo/replay.py:120-200)."""

  def __init__(
      self,
      capacity: int,
      structure: ReplayStructure,
      random_state: np.random.RandomState,
      encoder: Optional[Callable[[ReplayStructure], Any]] = None,
      decoder: Optional[Callable[[Any], ReplayStructure]] = None,
      device=None,
  ):
    super().__init__(capacity, structure, random_state, encoder, decoder,
                     device)

  def add(self, item: ReplayStructure) -> None:
    """Adds a single item, evicting the oldest when full (replay.py:141-150)."""
    self._store(item)

  def sample_ids_device(self, size: int) -> torch.Tensor:
    """Draws `size` ids exactly as the reference (replay.py:76-82)."""
    if self._size == 0:
      raise ValueError('low >= high')  # what randint(0) raises in NumPy.
    pos = self._random_state.randint(self._size, size=size)
    pos_d = torch.from_numpy(pos.astype(np.int64)).to(self._device,
                                                      non_blocking=True)
    ids_d = torch.empty(size, dtype=torch.int64, device=self._device)
    _lib.check(_lib.load().dz_uniform_pos_to_id(
        pos_d.data_ptr(), size, self._t, self._size, self._capacity,
        ids_d.data_ptr(), self._stream()), 'dz_uniform_pos_to_id')
    return ids_d

  SAMPLE_RING_DEPTH = 4

  def sample_device(self, size: int, prepare_only: bool = False):
    """Pipelined sample: (structure of device tensors, ids tensor).  For batches
    <= 64 this is ONE launch: the `randint` draws travel in the kernel arguments,
    the kernel maps positions to ids and gathers the rows (outputs live in a ring
    of preallocated slots, valid for SAMPLE_RING_DEPTH further calls)."""
    if size > 64:
      ids_d = self.sample_ids_device(size)
      outs = self._ring.gather(ids_d, size, self._stream())
      return type(self._structure)(*outs), ids_d
    if self._size == 0:
      raise ValueError('low >= high')  # what randint(0) raises in NumPy.
    ring = getattr(self, '_sample_ring', None)
    if ring is None or ring[0][0] != size:
      ring = []
      for _ in range(self.SAMPLE_RING_DEPTH):
        outs = [torch.empty((size,) + tuple(f.shape[1:]), dtype=f.dtype,
                            device=self._device) for f in self._ring.fields]
        arr = (_lib.FieldDesc * len(outs))()
        for i, (f, o) in enumerate(zip(self._ring.fields, outs)):
          arr[i].src = f.data_ptr()
          arr[i].dst = o.data_ptr()
          arr[i].row_bytes = f[0].numel() * f.element_size()
        ids = torch.empty(size, dtype=torch.int64, device=self._device)
        ring.append((size, arr, type(self._structure)(*outs), ids))
      self._sample_ring, self._sample_ring_pos = ring, 0
    _, arr, outs, ids = ring[self._sample_ring_pos % len(ring)]
    self._sample_ring_pos += 1
    pos = np.ascontiguousarray(self._random_state.randint(self._size, size=size),
                               dtype=np.int64)
    if prepare_only:
      return arr, outs, ids, pos
    _lib.check(_lib.load().dz_replay_sample_uniform(
        arr, len(self._ring.fields), pos.ctypes.data, size, self._t, self._size,
        self._capacity, ids.data_ptr(), self._stream()), 'dz_replay_sample_uniform')
    return outs, ids

  def prepare_next_sample(self, size: int):
    """The NEXT `sample_device(size)` as a descriptor for `DenseLearner.step(
    next_sample=...)` (see PrioritizedTransitionReplay.prepare_next_sample): the
    `randint` draws are made now, the ring slot reserved; returns
    `(descriptor, (structure of device tensors, ids))`.  A uniform sample does not
    depend on the learner step at all, only on the store: valid while nothing is added
    in between."""
    if size > 64:
      raise ValueError('prepare_next_sample handles batches <= 64')
    arr, outs, ids, pos = self.sample_device(size, prepare_only=True)
    d = _lib.NextSample()
    d.args.node = None              # uniform replay: no tree
    d.args.capacity = self._capacity
    d.args.size = self._size
    d.args.t = self._t
    d.pos_h = pos.ctypes.data
    d.fields = ctypes.cast(arr, ctypes.c_void_p)
    d.num_fields = len(self._ring.fields)
    d.n = size
    d.ids_out = ids.data_ptr()
    d.status = self._status.word.data_ptr()
    d._keep = (pos, arr)            # the descriptor points into these
    self._prepared = (self._t, (outs, ids))
    return d, (outs, ids)

  def take_prepared(self):
    """The `(structure of device tensors, ids)` a learner step produced from
    `prepare_next_sample`'s descriptor; refuses it if the store changed in between."""
    t, batch = self._prepared
    if t != self._t:
      raise RuntimeError('the replay changed between prepare_next_sample and its use')
    return batch

  def sample(self, size: int) -> ReplayStructure:
    """Samples a batch uniformly with replacement (replay.py:157-163)."""
    outs, _ = self.sample_device(size)
    return self._to_host(outs)

  def get_state(self, compact=False) -> Mapping[str, Any]:
    """Replay state.  Default: the REFERENCE's dictionary (replay.py:178-185):
    {'storage': [(id, item)...], 't', 'distribution': {'ids', 'id_to_index'}},
    loadable by `dqn_zoo.replay.TransitionReplay.set_state`.  `compact=True`
    ('host') / 'device': native format with the live rows as arrays (no
    per-item Python objects; 'device' keeps them in HBM)."""
    if compact:
      return self._compact_state('device' if compact == 'device' else 'host')
    return {
        'storage': self._reference_storage(),
        't': self._t,
        'distribution': uniform_distribution_state(self._t, self._size,
                                                   self._capacity),
    }

  def set_state(self, state: Mapping[str, Any]) -> None:
    """Accepts the reference's dictionary (replay.py:187-191) or the native
    compact one.  The reference's distribution tables are validated against the
    closed forms this class implements; `t` and the FIFO window carry the rest."""
    if state.get('format') == COMPACT_FORMAT:
      self._load_compact(state)
      return
    if 'storage' not in state:  # round-1 native format
      self._load_compact({'t': state['t'], 'size': state['size'],
                          'fields': state['fields'], 'layout': 'ring'})
      return
    size = _check_storage_ids(state['storage'], int(state['t']), self._capacity)
    want = uniform_distribution_state(int(state['t']), size, self._capacity)
    _same_table(state['distribution']['ids'], want['ids'], 'ids')
    _same_table(state['distribution']['id_to_index'], want['id_to_index'],
                'id_to_index')
    self._load_reference_storage(state['storage'], state['t'])

  def check_valid(self) -> Tuple[bool, str]:
    if self._t < self._size:
      return False, 't should be >= storage size.'
    if not 0 <= self._size <= self._capacity:
      return False, 'size should be within [0, capacity].'
    return True, ''


def _power(base, exponent) -> np.ndarray:
  """`base ** exponent` except 0 ** 0 == 0 (ref: replay.py:203-208)."""
  base = np.asarray(base)
  return np.where(base == 0.0, 0.0, base**exponent)


def importance_sampling_weights(
    probabilities: np.ndarray,
    uniform_probability: float,
    exponent: float,
    normalize: bool,
) -> np.ndarray:
  """Importance-sampling weights from probabilities (ref: replay.py:211-243)."""
  if not 0.0 <= exponent <= 1.0:
    raise ValueError('Require 0 <= exponent <= 1.')
  if not 0.0 <= uniform_probability <= 1.0:
    raise ValueError('Expected 0 <= uniform_probability <= 1.')
  weights = (uniform_probability / probabilities) ** exponent
  if normalize:
    weights /= np.max(weights)
  if not np.isfinite(weights).all():
    raise ValueError('Weights are not finite: %s.' % weights)
  return weights


class SumTree:
  """Device-resident float64 sum tree (ref: dqn_zoo/replay.py:246-426).

  Same surface as the reference class; storage is a `float64[2*capacity]`
  implicit heap in HBM and set/get/query run as HIP kernels.  Host arguments
  are uploaded, results downloaded (this class is the synchronous, test-facing
  view of the kernels the prioritized replay uses in place).
  """

  def __init__(self, device=None):
    _lib.load()
    self._device = _require_gpu(device)
    self._size = 0
    self._first_leaf = 0
    self._storage = torch.zeros(0, dtype=torch.float64, device=self._device)
    self._status = _Status(self._device)

  def _stream(self):
    return _lib.stream_ptr(self._device)

  def _dev(self, a, dtype):
    return torch.from_numpy(np.array(a, dtype=dtype, order='C')).to(self._device)

  def resize(self, size: int) -> None:
    self._initialize(size, None)

  def get(self, indices: Sequence[int]) -> np.ndarray:
    indices = np.asarray(indices)
    if not ((0 <= indices) & (indices < self.size)).all():
      raise IndexError('index out of range, expect 0 <= index < %s' % self.size)
    flat = indices.reshape(-1)
    if flat.size == 0:
      return np.zeros(indices.shape, np.float64)
    idx_d = self._dev(flat, np.int64)
    out = torch.empty(flat.size, dtype=torch.float64, device=self._device)
    _lib.check(_lib.load().dz_sumtree_get(
        self._storage.data_ptr(), self._first_leaf, self._size,
        idx_d.data_ptr(), flat.size, out.data_ptr(),
        self._status.word.data_ptr(), self._stream()), 'dz_sumtree_get')
    return out.cpu().numpy().reshape(indices.shape)

  def set(self, indices: Sequence[int], values: Sequence[float]) -> None:
    values = np.asarray(values)
    if not np.isfinite(values).all() or (values < 0.0).any():
      raise ValueError('value must be finite and positive.')
    indices = np.asarray(indices, dtype=np.int64).reshape(-1)
    values = np.broadcast_to(values.astype(np.float64), indices.shape)
    if indices.size == 0:
      return
    if ((indices < -self._size) | (indices >= self._size)).any():
      raise IndexError('index out of range')
    for lo in range(0, indices.size, 1024):  # kernel handles <= 1024 per call
      idx_d = self._dev(indices[lo:lo + 1024], np.int64)
      val_d = self._dev(values[lo:lo + 1024], np.float64)
      _lib.check(_lib.load().dz_sumtree_set(
          self._storage.data_ptr(), self._first_leaf, self._size,
          idx_d.data_ptr(), val_d.data_ptr(), idx_d.numel(),
          self._status.word.data_ptr(), self._stream()), 'dz_sumtree_set')

  def set_all(self, values: Sequence[float]) -> None:
    values = np.asarray(values)
    if not np.isfinite(values).all() or (values < 0.0).any():
      raise ValueError('Values must be finite positive numbers.')
    self._initialize(len(values), values)

  def query(self, targets: Sequence[float]) -> Sequence[int]:
    targets = np.asarray(targets, dtype=np.float64).reshape(-1)
    if targets.size == 0:
      return []
    t_d = self._dev(targets, np.float64)
    out = torch.empty(targets.size, dtype=torch.int64, device=self._device)
    _lib.check(_lib.load().dz_sumtree_query(
        self._storage.data_ptr(), self._first_leaf, t_d.data_ptr(),
        targets.size, out.data_ptr(), self._status.word.data_ptr(),
        self._stream()), 'dz_sumtree_query')
    self._status.check()
    return [int(i) for i in out.cpu().numpy()]

  def root(self) -> float:
    if self._size == 0:
      return np.nan
    return float(self._storage[1].item())

  @property
  def values(self) -> np.ndarray:
    return self._storage[self._first_leaf:self._first_leaf + self._size
                         ].cpu().numpy()

  @property
  def size(self) -> int:
    return self._size

  @property
  def capacity(self) -> int:
    return self._first_leaf

  def get_state(self) -> Mapping[str, Any]:
    return {'size': self._size, 'storage': self._storage.cpu().numpy(),
            'first_leaf': self._first_leaf}

  def set_state(self, state: Mapping[str, Any]) -> None:
    self._size = state['size']
    self._storage = torch.from_numpy(np.array(state['storage'],
                                              dtype=np.float64)).to(self._device)
    self._first_leaf = state['first_leaf']

  def check_valid(self) -> Tuple[bool, str]:
    if self._storage.numel() != 2 * self._first_leaf:
      return False, 'first_leaf should be half the size of storage.'
    if not 0 <= self.size <= self.capacity:
      return False, 'Require 0 <= self.size <= self.capacity.'
    s = self._storage.cpu().numpy()
    n = self._first_leaf
    if n > 1:
      inner = np.arange(1, n)
      bad = np.nonzero(s[inner] != s[2 * inner] + s[2 * inner + 1])[0]
      if bad.size:
        return False, ('Non-leaf node %d should be sum of child nodes.' %
                       inner[bad[0]])
    return True, ''

  def _initialize(self, size, values):
    assert size >= 0
    assert values is None or len(values) == size
    if size < self._size:
      new_values = self.values[:size] if values is None else values
      self._size = size
      self._set_values(new_values)
    elif size <= self.capacity:
      self._size = size
      if values is not None:
        self._set_values(values)
    else:
      new_values = self.values if values is None else values
      cap = _next_pow2(size)
      self._storage = torch.zeros(2 * cap, dtype=torch.float64,
                                  device=self._device)
      self._first_leaf = cap
      self._size = size
      self._set_values(new_values)

  def _set_values(self, values):
    values = np.asarray(values, dtype=np.float64)
    n = len(values)
    assert n <= self.capacity
    if self.capacity == 0:
      return
    if n:
      self._storage[self._first_leaf:self._first_leaf + n] = self._dev(
          values, np.float64)
    _lib.check(_lib.load().dz_sumtree_rebuild(
        self._storage.data_ptr(), self._first_leaf, n, self._stream()),
               'dz_sumtree_rebuild')


class DeviceSample(typing.NamedTuple):
  """Result of `PrioritizedTransitionReplay.sample_device` (all in HBM)."""
  transitions: Any
  ids: torch.Tensor          # int64[B]
  probabilities: torch.Tensor  # float64[B]
  weights: torch.Tensor      # float64[B]  (device pow)
  weights32: torch.Tensor    # float32[B]  (what the learner consumes)


class PrioritizedTransitionReplay(_ReplayBase):
  """Proportional prioritized replay (ref: dqn_zoo/replay.py:654-768)."""

  def __init__(
      self,
      capacity: int,
      structure: ReplayStructure,
      priority_exponent: float,
      importance_sampling_exponent: Callable[[int], float],
      uniform_sample_probability: float,
      normalize_weights: bool,
      random_state: np.random.RandomState,
      encoder: Optional[Callable[[ReplayStructure], Any]] = None,
      decoder: Optional[Callable[[Any], ReplayStructure]] = None,
      device=None,
  ):
    # Same argument checks as PrioritizedDistribution (replay.py:440-448).
    if priority_exponent < 0.0:
      raise ValueError('Require priority_exponent >= 0.')
    if not 0.0 <= uniform_sample_probability <= 1.0:
      raise ValueError('Require 0 <= uniform_sample_probability <= 1.')
    super().__init__(capacity, structure, random_state, encoder, decoder,
                     device)
    self._priority_exponent = float(priority_exponent)
    self._usp = float(uniform_sample_probability)
    self._importance_sampling_exponent = importance_sampling_exponent
    self._normalize_weights = bool(normalize_weights)
    self._cap_pow2 = _next_pow2(self._capacity)
    # float64[2*cap_pow2]: 16 MiB at capacity 1e6 (replay.py:382-385).
    self._tree = torch.zeros(2 * self._cap_pow2, dtype=torch.float64,
                             device=self._device)
    # Running max of priorities, kept on the device for the pipelined learner
    # (ref: rainbow/agent.py:79,196-197 keep it in the agent).
    self.max_seen_priority_device = torch.ones(1, dtype=torch.float64,
                                               device=self._device)

  # -- adds ------------------------------------------------------------------
  def add(self, item: ReplayStructure, priority: float) -> None:
    """Adds one item with a host-side priority (ref: replay.py:690-699)."""
    leaf = _power(np.asarray([priority], dtype=np.float64),
                  self._priority_exponent)
    if not np.isfinite(leaf).all() or (leaf < 0.0).any():
      raise ValueError('value must be finite and positive.')
    ti = int(tree_index_of_id(self._t, self._capacity))
    self._store(item)
    self._tree_set_host(np.array([ti], np.int64), leaf)

  def add_with_device_priority(self, item, priority_d=None) -> None:
    """add() whose priority is a device scalar (default: the running max);
    no host sync.  Tree-side equivalent of replay.py:690-699."""
    p = self.max_seen_priority_device if priority_d is None else priority_d
    self._store(item, self._tree.data_ptr(), self._cap_pow2, p.data_ptr(),
                self._priority_exponent, self._status.word.data_ptr())

  def bulk_fill(self, fields, priority: float = 1.0) -> int:
    t0 = self._t
    n = super().bulk_fill(fields)
    lib = _lib.load()
    for lo in range(0, n, 1024):
      m = min(1024, n - lo)
      _lib.check(lib.dz_prioritized_add(
          self._tree.data_ptr(), self._cap_pow2, self._capacity, t0 + lo, m,
          float(priority), None, self._priority_exponent,
          self._status.word.data_ptr(), self._stream()), 'dz_prioritized_add')
    return n

  def _tree_set_host(self, tree_idx, leaf_values):
    idx_d = torch.from_numpy(np.ascontiguousarray(tree_idx, dtype=np.int64)
                             ).to(self._device)
    val_d = torch.from_numpy(np.ascontiguousarray(leaf_values,
                                                  dtype=np.float64)
                             ).to(self._device)
    _lib.check(_lib.load().dz_sumtree_set(
        self._tree.data_ptr(), self._cap_pow2, self._capacity,
        idx_d.data_ptr(), val_d.data_ptr(), idx_d.numel(),
        self._status.word.data_ptr(), self._stream()), 'dz_sumtree_set')

  # -- sampling --------------------------------------------------------------
  def _draw(self, size, zero_root):
    """Host RNG draws in the reference's order (replay.py:551-566)."""
    rs = self._random_state
    pos = rs.randint(self._size, size=size).astype(np.int64)
    if zero_root:
      u_target = np.zeros(size, np.float64)  # the reference skips this draw.
    else:
      u_target = rs.uniform(size=size)
    u_mix = rs.uniform(size=size)
    # one upload: [pos int64 | u_target f64 bits | u_mix f64 bits]
    packed = np.empty(3 * size, np.int64)
    packed[:size] = pos
    packed[size:2 * size] = u_target.view(np.int64)
    packed[2 * size:] = u_mix.view(np.int64)
    return torch.from_numpy(packed).to(self._device, non_blocking=True)

  def _launch_sample(self, size, draws_d, compute_weights, pipelined):
    dev = self._device
    ids = torch.empty(size, dtype=torch.int64, device=dev)
    probs = torch.empty(size, dtype=torch.float64, device=dev)
    w64 = torch.empty(size, dtype=torch.float64, device=dev)
    w32 = torch.empty(size, dtype=torch.float32, device=dev)
    a = _lib.PrioSampleArgs()
    a.node = self._tree.data_ptr()
    a.cap_pow2 = self._cap_pow2
    a.capacity = self._capacity
    a.size = self._size
    a.t = self._t
    base = draws_d.data_ptr()
    a.pos = base
    a.u_target = base + 8 * size
    a.u_mix = base + 16 * size
    up = 1.0 / self._size
    a.usp = self._usp
    a.one_minus_usp = 1.0 - self._usp
    a.usp_times_up = self._usp * up
    a.uniform_prob = up
    a.beta = float(self.importance_sampling_exponent)
    a.normalize = int(self._normalize_weights)
    a.compute_weights = int(compute_weights)
    a.assume_nonzero_root = int(pipelined)
    _lib.check(_lib.load().dz_prioritized_sample(
        ctypes.byref(a), size, ids.data_ptr(), None, probs.data_ptr(),
        w64.data_ptr(), w32.data_ptr(), self._status.word.data_ptr(),
        self._stream()), 'dz_prioritized_sample')
    return ids, probs, w64, w32

  def sample_device(self, size: int) -> DeviceSample:
    """Pipelined sample for the on-device learner: no host synchronisation.

    ids and probabilities are bit-identical to the reference; weights use the
    device pow (equal to NumPy's after the float32 cast the learner applies,
    see DESIGN.md).  Assumes root() != 0, which holds whenever priorities come
    from max_seen_priority >= 1 (rainbow/agent.py:79,149); a zero root raises
    through the sticky status word at the next `check_status()`.

    Outputs live in a ring of `SAMPLE_RING_DEPTH` preallocated slots (no
    allocator traffic, pinned staging for the 3xB RNG draws): a returned
    DeviceSample stays valid until that many further calls.
    """
    if self._size == 0:
      raise RuntimeError('No IDs to sample.')
    beta = float(self.importance_sampling_exponent)
    if not 0.0 <= beta <= 1.0:
      raise ValueError('Require 0 <= exponent <= 1.')
    slot = self._ring_slot(size)
    # host RNG draws in the reference's order (replay.py:551-566), written
    # into the slot's host buffer: for batches <= 64 they travel to the device
    # inside the kernel arguments, otherwise through one pinned async copy.
    via_args = size <= 64
    if not via_args and slot.copied is not None:
      slot.copied.synchronize()  # the previous upload from this slot finished
    rs = self._random_state
    h, hf = slot.host_np, slot.host_f64
    h[:size] = rs.randint(self._size, size=size)
    # (`random_sample(n)` is `uniform(size=n)` bit for bit -- 0.0 + 1.0 * the same doubles -- at a
    # third of the call's cost; tests/test_replay.py pins the equality)
    hf[size:2 * size] = rs.random_sample(size)
    hf[2 * size:] = rs.random_sample(size)
    if not via_args:
      slot.draws.copy_(slot.host, non_blocking=True)
      if slot.copied is None:
        slot.copied = torch.cuda.Event()
      slot.copied.record(_lib.current_stream(self._device))
    if via_args:  # ONE launch: sample (draws in the kernel arguments) + gather
      # the slot's own descriptor: everything but `args` is filled once, the call marshals
      # two arguments (dz_sample_gather_desc)
      d = getattr(slot, 'sg_desc', None)
      if d is None:
        d = slot.sg_desc = self._fill_desc(_lib.NextSample(), slot, size)
        d.args = slot.args      # (the constant fields: tree, capacity, normalisation)
        slot.sg_args, slot.sg_ref = d.args, ctypes.byref(d)   # (d.args: a view of the embedded struct)
        slot.sg_fn = _lib.load().dz_sample_gather_desc
      a = slot.sg_args
      a.size = self._size
      a.t = self._t
      up = 1.0 / self._size
      a.usp_times_up = self._usp * up
      a.uniform_prob = up
      a.beta = beta
      rc = slot.sg_fn(slot.sg_ref, self._stream())
      if rc:
        _lib.check(rc, 'dz_prioritized_sample_gather')
      return slot.sample
    a = slot.args
    a.size = self._size
    a.t = self._t
    up = 1.0 / self._size
    a.usp_times_up = self._usp * up
    a.uniform_prob = up
    a.beta = beta
    lib = _lib.load()
    stream = self._stream()
    _lib.check(lib.dz_prioritized_sample(
        ctypes.byref(a), size, slot.ids.data_ptr(), None, slot.probs.data_ptr(),
        slot.w64.data_ptr(), slot.w32.data_ptr(), self._status.word.data_ptr(),
        stream), 'dz_prioritized_sample')
    _lib.check(lib.dz_replay_gather(slot.fields, len(self._ring.fields),
                                    slot.ids.data_ptr(), size, self._capacity,
                                    stream), 'dz_replay_gather')
    return slot.sample

  def prepare_next_sample(self, size: int):
    """The NEXT `sample_device(size)` as a descriptor instead of a launch: makes the
    host RNG draws now (in the reference's order, replay.py:551-566), reserves the
    ring slot and returns `(descriptor, sample)`.  The descriptor goes to
    `RainbowLearner.step(next_sample=...)`, whose optimiser launch then carries the
    sample + gather as extra blocks AFTER that step's priority write-back; `sample`
    (a DeviceSample over the slot's buffers) is the next batch once that step has
    run.  Valid only if nothing is added to this replay in between (a learner over a
    static replay): `take_prepared()` checks that."""
    if self._size == 0:
      raise RuntimeError('No IDs to sample.')
    if size > 64:
      raise ValueError('prepare_next_sample handles batches <= 64 (draws travel in kernel arguments)')
    beta = float(self.importance_sampling_exponent)
    if not 0.0 <= beta <= 1.0:
      raise ValueError('Require 0 <= exponent <= 1.')
    slot = self._ring_slot(size)
    rs = self._random_state
    h, hf = slot.host_np, slot.host_f64
    h[:size] = rs.randint(self._size, size=size)
    # (`random_sample(n)` is `uniform(size=n)` bit for bit -- 0.0 + 1.0 * the same doubles -- at a
    # third of the call's cost; tests/test_replay.py pins the equality)
    hf[size:2 * size] = rs.random_sample(size)
    hf[2 * size:] = rs.random_sample(size)
    a = slot.args
    a.size = self._size
    a.t = self._t
    up = 1.0 / self._size
    a.usp_times_up = self._usp * up
    a.uniform_prob = up
    a.beta = beta
    d = getattr(slot, 'next_desc', None)
    if d is None:
      d = slot.next_desc = self._fill_desc(_lib.NextSample(), slot, size)
    d.args = a   # by-value copy of the slot's sample arguments
    self._prepared = (slot, self._t)
    return d, slot.sample

  def _fill_desc(self, d, slot, size):
    """dz_next_sample_t over a ring slot's buffers (everything but `args`)."""
    hp = slot.host.data_ptr()
    d.pos_h, d.u_target_h, d.u_mix_h = hp, hp + 8 * size, hp + 16 * size
    d.fields = ctypes.cast(slot.fields, ctypes.c_void_p)
    d.num_fields = len(self._ring.fields)
    d.n = size
    d.ids_out = slot.ids.data_ptr()
    d.probs_out = slot.probs.data_ptr()
    d.weights_out = slot.w64.data_ptr()
    d.weights32_out = slot.w32.data_ptr()
    d.status = self._status.word.data_ptr()
    return d

  def take_prepared(self) -> DeviceSample:
    """The batch a learner step produced from `prepare_next_sample`'s descriptor."""
    slot, t = self._prepared
    if t != self._t:
      raise RuntimeError('the replay changed between prepare_next_sample and its use')
    return slot.sample

  SAMPLE_RING_DEPTH = 4
  MAX_PREPARED_BATCH = 64   # draws of a prepared sample travel in kernel arguments

  def _ring_slot(self, size):
    ring = getattr(self, '_sample_ring', None)
    if ring is None or ring[0].size != size or ring[0].nfields != len(
        self._ring.fields):
      ring = [self._make_slot(size) for _ in range(self.SAMPLE_RING_DEPTH)]
      self._sample_ring = ring
      self._sample_ring_pos = 0
    slot = ring[self._sample_ring_pos % len(ring)]
    self._sample_ring_pos += 1
    return slot

  def _make_slot(self, size):
    dev = self._device

    class Slot:
      pass

    sl = Slot()
    sl.size = size
    sl.nfields = len(self._ring.fields)
    sl.host = torch.empty(3 * size, dtype=torch.int64).pin_memory()
    sl.host_np = sl.host.numpy()
    sl.host_f64 = sl.host_np.view(np.float64)   # the same pinned words
    sl.draws = torch.empty(3 * size, dtype=torch.int64, device=dev)
    sl.copied = None
    sl.ids = torch.empty(size, dtype=torch.int64, device=dev)
    sl.probs = torch.empty(size, dtype=torch.float64, device=dev)
    sl.w64 = torch.empty(size, dtype=torch.float64, device=dev)
    sl.w32 = torch.empty(size, dtype=torch.float32, device=dev)
    outs = [torch.empty((size,) + tuple(f.shape[1:]), dtype=f.dtype, device=dev)
            for f in self._ring.fields]
    arr = (_lib.FieldDesc * len(outs))()
    for i, (f, o) in enumerate(zip(self._ring.fields, outs)):
      arr[i].src = f.data_ptr()
      arr[i].dst = o.data_ptr()
      arr[i].row_bytes = f[0].numel() * f.element_size()
    sl.fields = arr
    a = _lib.PrioSampleArgs()
    a.node = self._tree.data_ptr()
    a.cap_pow2 = self._cap_pow2
    a.capacity = self._capacity
    base = sl.draws.data_ptr()
    a.pos = base
    a.u_target = base + 8 * size
    a.u_mix = base + 16 * size
    a.usp = self._usp
    a.one_minus_usp = 1.0 - self._usp
    a.normalize = int(self._normalize_weights)
    a.compute_weights = 1
    a.assume_nonzero_root = 1
    sl.args = a
    sl.sample = DeviceSample(type(self._structure)(*outs), sl.ids, sl.probs,
                             sl.w64, sl.w32)
    return sl

  def sample(self, size: int) -> Tuple[ReplayStructure, np.ndarray, np.ndarray]:
    """Samples a batch of transitions (ref: replay.py:706-723).

    Synchronous and bit-exact: ids/probabilities from the device, importance
    weights from NumPy on the host (`importance_sampling_weights`)."""
    if self._size == 0:
      raise RuntimeError('No IDs to sample.')
    root = float(self._tree[1].item())
    draws = self._draw(size, zero_root=(root == 0.0))
    ids_d, probs_d, _, _ = self._launch_sample(size, draws, False, False)
    outs = self._ring.gather(ids_d, size, self._stream())
    probs = probs_d.cpu().numpy()
    self._status.check()
    weights = importance_sampling_weights(
        probs, uniform_probability=1.0 / self._size,
        exponent=self.importance_sampling_exponent,
        normalize=self._normalize_weights)
    return self._to_host(outs), ids_d.cpu().numpy(), weights

  # -- priorities ------------------------------------------------------------
  def update_priorities(self, ids, priorities) -> None:
    """Updates ids with given priorities (ref: replay.py:725-730, 536-545).

    Host arrays: leaf values are computed with NumPy exactly as the reference
    does (dtype rules included) and only the tree update runs on the device.
    Device tensors: everything runs on the device (`dz_prioritized_update`),
    which also folds max(priorities) into `max_seen_priority_device`.
    """
    if isinstance(priorities, torch.Tensor) or isinstance(ids, torch.Tensor):
      ids_d = ids if isinstance(ids, torch.Tensor) else torch.from_numpy(
          np.asarray(ids, dtype=np.int64)).to(self._device)
      p_d = priorities if isinstance(priorities, torch.Tensor) else (
          torch.from_numpy(np.asarray(priorities)).to(self._device))
      if p_d.dtype not in (torch.float32, torch.float64):
        raise TypeError('priorities must be float32 or float64')
      _lib.check(_lib.load().dz_prioritized_update(
          self._tree.data_ptr(), self._cap_pow2, self._capacity, self._size,
          self._t, ids_d.data_ptr(), p_d.data_ptr(),
          int(p_d.dtype == torch.float32), self._priority_exponent,
          ids_d.numel(), self.max_seen_priority_device.data_ptr(),
          self._status.word.data_ptr(), self._stream()),
                 'dz_prioritized_update')
      return
    priorities = np.asarray(priorities)
    ids = np.asarray(ids, dtype=np.int64).reshape(-1)
    for i in ids:
      if not self._t - self._size <= i < self._t:
        raise IndexError('ID %d does not exist.' % i)
    leaf = _power(priorities, self._priority_exponent)
    if not np.isfinite(leaf).all() or (leaf < 0.0).any():
      raise ValueError('value must be finite and positive.')
    leaf = np.broadcast_to(np.asarray(leaf, dtype=np.float64), ids.shape)
    self._tree_set_host(tree_index_of_id(ids, self._capacity), leaf)

  def priority_sink(self, ids: torch.Tensor):
    """Descriptor that lets a learner step perform `update_priorities(ids, <its
    float32 priorities>)` inside its own launches (RainbowLearner.step
    `priority_sink=`): (tree, cap_pow2, capacity, ids, exponent, running-max
    scalar, status word).  `ids` must be the device ids of the batch just
    sampled from THIS replay; they are not re-validated."""
    if not (isinstance(ids, torch.Tensor) and ids.dtype == torch.int64 and
            ids.device == self._tree.device):
      raise ValueError('priority_sink needs the int64 device ids of sample_device()')
    return (self._tree.data_ptr(), self._cap_pow2, self._capacity, ids.data_ptr(),
            float(self._priority_exponent), self.max_seen_priority_device.data_ptr(),
            self._status.word.data_ptr())

  @property
  def importance_sampling_exponent(self):
    """Importance sampling exponent at current step (replay.py:742-745)."""
    return self._importance_sampling_exponent(self._t)

  @property
  def tree_storage(self) -> torch.Tensor:
    """The float64[2*cap_pow2] heap array in HBM (root at [1])."""
    return self._tree

  def get_state(self, compact=False) -> Mapping[str, Any]:
    """Replay state.  Default: the REFERENCE's dictionary (replay.py:747-754):
    {'storage': [(id, item)...], 't', 'distribution': {'sum_tree': {'size',
    'storage', 'first_leaf'}, 'id_to_index', 'index_to_id', 'inactive_indices',
    'active_indices', 'active_indices_location'}} with the tables materialised
    from the closed forms -- `dqn_zoo.replay.PrioritizedTransitionReplay
    .set_state` loads it.  `compact=True` ('host') / 'device': native format
    (live rows + the float64 tree + the running max priority)."""
    if compact:
      where = 'device' if compact == 'device' else 'host'
      st = dict(self._compact_state(where))
      st['sum_tree_storage'] = self._tree.clone() if where == 'device' \
          else self._tree.cpu().numpy()
      st['max_seen_priority'] = float(self.max_seen_priority_device.item())
      return st
    return {
        'storage': self._reference_storage(),
        't': self._t,
        'distribution': prioritized_distribution_state(
            self._t, self._size, self._capacity, self._cap_pow2,
            self._tree.cpu().numpy()),
    }

  def set_state(self, state: Mapping[str, Any]) -> None:
    """Accepts the reference's dictionary (replay.py:756-760) or the native
    compact one.  The reference format does not carry the agent's
    max_seen_priority (rainbow/agent.py:231,245 keeps it in the agent state):
    the agents restore `max_seen_priority_device` themselves."""
    if state.get('format') == COMPACT_FORMAT or 'storage' not in state:
      st = dict(state)
      st.setdefault('layout', 'ring')   # round-1 native dictionaries
      self._load_compact(st)
      tree = st['sum_tree_storage']
      self._tree.copy_(tree if isinstance(tree, torch.Tensor)
                       else torch.from_numpy(np.asarray(tree, np.float64)))
      self.max_seen_priority_device.fill_(st['max_seen_priority'])
      return
    t = int(state['t'])
    size = _check_storage_ids(state['storage'], t, self._capacity)
    dist = state['distribution']
    tree = dist['sum_tree']
    storage = np.asarray(tree['storage'], dtype=np.float64)
    if (int(tree['size']) != self._capacity or
        int(tree['first_leaf']) != self._cap_pow2 or
        storage.shape != (2 * self._cap_pow2,)):
      raise ValueError(
          'sum tree of size %s / first_leaf %s does not fit a replay of '
          'capacity %d' % (tree['size'], tree['first_leaf'], self._capacity))
    want = prioritized_distribution_state(t, size, self._capacity,
                                          self._cap_pow2, None)
    for name in ('id_to_index', 'index_to_id', 'inactive_indices',
                 'active_indices', 'active_indices_location'):
      _same_table(dist[name], want[name], name)
    self._load_reference_storage(state['storage'], t)
    self._tree.copy_(torch.from_numpy(storage))

  def check_valid(self) -> Tuple[bool, str]:
    if self._t < self._size:
      return False, 't should be >= storage size.'
    s = self._tree.cpu().numpy()
    n = self._cap_pow2
    if n > 1:
      inner = np.arange(1, n)
      bad = np.nonzero(s[inner] != s[2 * inner] + s[2 * inner + 1])[0]
      if bad.size:
        return False, ('Non-leaf node %d should be sum of child nodes.' %
                       inner[bad[0]])
    live = tree_index_of_id(np.arange(self._t - self._size, self._t),
                            self._capacity)
    dead = np.ones(n, bool)
    dead[live] = False
    if (s[n:][dead] != 0.0).any():
      return False, 'Inactive tree indices should have zero priority.'
    return True, ''


# --------------------------------------------------------------------------- #
#  Transition accumulators: host-side feeders (stay on host).
# --------------------------------------------------------------------------- #
class TransitionAccumulator:
  """Accumulates timesteps into 1-step transitions (ref: replay.py:771-805)."""

  window_size = 1   # frames between s_tm1 and s_t (device_obs.depth_for)

  def __init__(self):
    self.reset()

  def step(self, timestep_t, a_t) -> Iterable[Transition]:
    if timestep_t.first():
      self.reset()
    prev, prev_a = self._timestep_tm1, self._a_tm1
    if prev is None and not timestep_t.first():
      raise ValueError('Expected FIRST timestep, got %s.' % str(timestep_t))
    self._timestep_tm1, self._a_tm1 = timestep_t, a_t
    if prev is None:
      return
    yield Transition(s_tm1=prev.observation, a_tm1=prev_a,
                     r_t=timestep_t.reward, discount_t=timestep_t.discount,
                     s_t=timestep_t.observation)

  def reset(self) -> None:
    self._timestep_tm1 = None
    self._a_tm1 = None


def _fold_n_step(window):
  """One n-step transition from n consecutive 1-step ones (replay.py:808-823)."""
  reward, discount = 0.0, 1.0
  for tr in window:
    reward += discount * tr.r_t
    discount *= tr.discount_t
  return Transition(s_tm1=window[0].s_tm1, a_tm1=window[0].a_tm1, r_t=reward,
                    discount_t=discount, s_t=window[-1].s_t)


class NStepTransitionAccumulator:
  """Accumulates timesteps into n-step transitions (ref: replay.py:826-892).

  FIRST: nothing.  MID: once n 1-step transitions are buffered, one n-step
  transition per step.  LAST: every suffix of the buffer, longest first.
  """

  def __init__(self, n):
    self._window = collections.deque(maxlen=n)
    self.reset()

  @property
  def window_size(self) -> int:
    """n: the largest number of frames between s_tm1 and s_t of an emitted transition."""
    return self._window.maxlen

  def step(self, timestep_t, a_t) -> Iterable[Transition]:
    if timestep_t.first():
      self.reset()
    prev, prev_a = self._timestep_tm1, self._a_tm1
    if prev is None and not timestep_t.first():
      raise ValueError('Expected FIRST timestep, got %s.' % str(timestep_t))
    self._timestep_tm1, self._a_tm1 = timestep_t, a_t
    if prev is None:
      return
    self._window.append(Transition(
        s_tm1=prev.observation, a_tm1=prev_a, r_t=timestep_t.reward,
        discount_t=timestep_t.discount, s_t=timestep_t.observation))
    if timestep_t.last():
      while self._window:
        yield _fold_n_step(self._window)
        self._window.popleft()
    elif len(self._window) == self._window.maxlen:
      yield _fold_n_step(self._window)

  def reset(self) -> None:
    self._window.clear()
    self._timestep_tm1 = None
    self._a_tm1 = None


# The reference's general id distributions (arbitrary ids, removals, capacity
# growth) for callers that use them directly; the replays above do not.
from dqn_zoo_amd.distributions import PrioritizedDistribution, UniformDistribution  # noqa: E402,F401  pylint: disable=wrong-import-position
