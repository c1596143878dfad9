"""Network descriptors and parameter layouts for the on-device learner.

The reference builds networks as Haiku functions (dqn_zoo/networks.py:224-363)
whose parameters are a nest of arrays.  Here a network is a *descriptor* (kind +
dimensions) and its parameters are ONE flat float32 buffer in HBM whose layout
is defined by the C library (`dz_rainbow_layout`).  This module converts between
that buffer and a dict of Haiku-shaped arrays (HWIO conv kernels, (in, out)
linear weights -- networks_test.py:44,53) for get_state/set_state, tests and
the `online_params` property.

Layer names follow the reference's creation order (networks.py:239-251):
conv1..conv3, adv1, adv2, val1, val2, each noisy layer with mu/{w,b} and
sigma/{w,b} (adv2/val2 have no mu bias).
"""

import ctypes
import typing

import numpy as np

from dqn_zoo_amd import _lib

FLAT = 3136
HIDDEN = 512


class C51NetworkOutputs(typing.NamedTuple):
  q_values: typing.Any
  q_logits: typing.Any


class RainbowNetwork:
  """Descriptor of `rainbow_atari_network` (ref: networks.py:224-261)."""

  def __init__(self, num_actions: int, support, noisy_weight_init: float = 0.1):
    support = np.asarray(support, dtype=np.float32)
    if support.ndim != 1:
      raise ValueError('support must have rank 1')  # chex.assert_rank(support, 1)
    self.num_actions = int(num_actions)
    self.support = support
    self.num_atoms = int(support.shape[0])
    self.noisy_weight_init = float(noisy_weight_init)

  def layout(self, batch_size: int) -> 'RainbowParamLayout':
    return RainbowParamLayout(self.num_actions, self.num_atoms, batch_size)

  def init(self, random_state: np.random.RandomState) -> dict:
    """Haiku-equivalent initialisation (networks.py:58-79, 149-166):
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights AND biases, sigma constant
    noisy_weight_init/sqrt(fan_in).  (The JAX key stream itself cannot be
    reproduced; the distribution is the same.)"""
    a, k = self.num_actions, self.num_atoms
    p = {}

    def uni(shape, fan):
      c = np.sqrt(1.0 / fan)
      return random_state.uniform(-c, c, size=shape).astype(np.float32)

    for name, ks, ci, co in (('conv1', 8, 4, 32), ('conv2', 4, 32, 64),
                             ('conv3', 3, 64, 64)):
      p[name + '/w'] = uni((ks, ks, ci, co), ci * ks * ks)
      p[name + '/b'] = uni((co,), ci * ks * ks)
    for name, nin, nout, bias in (('adv1', FLAT, HIDDEN, True),
                                  ('adv2', HIDDEN, a * k, False),
                                  ('val1', FLAT, HIDDEN, True),
                                  ('val2', HIDDEN, k, False)):
      p[name + '/mu/w'] = uni((nin, nout), nin)
      if bias:
        p[name + '/mu/b'] = uni((nout,), nin)
      s = np.float32(self.noisy_weight_init / np.sqrt(nin))
      p[name + '/sigma/w'] = np.full((nin, nout), s, np.float32)
      p[name + '/sigma/b'] = np.full((nout,), s, np.float32)
    return p


class RainbowParamLayout:
  """Offsets of every tensor in the flat parameter / noise / workspace buffers."""

  def __init__(self, num_actions, num_atoms, batch_size):
    self.c = _lib.RainbowLayout()
    _lib.check(_lib.load().dz_rainbow_layout(num_actions, num_atoms, batch_size,
                                             ctypes.byref(self.c)),
               'dz_rainbow_layout')
    self.num_actions, self.num_atoms, self.batch = num_actions, num_atoms, batch_size
    self.na = num_actions * num_atoms
    # fc2 "column space": advantage logits at [0, A*K), value logits at
    # [val_off, val_off+K); both blocks padded to 4 floats (pads stay zero).
    self.val_off = int(self.c.adv2_ld)
    self.ld2 = int(self.c.adv2_ld) + int(self.c.val2_ld)

  @property
  def param_count(self):
    return int(self.c.param_count)

  @property
  def noise_stride(self):
    return int(self.c.noise_stride)

  @property
  def ws_count(self):
    return int(self.c.ws_count)

  # -- parameters -----------------------------------------------------------
  def _views(self, flat):
    """dict name -> writable NumPy view into `flat` (Haiku shapes)."""
    c, na, k = self.c, self.na, self.num_atoms
    v = {}
    shapes = [(8, 8, 4, 32), (4, 4, 32, 64), (3, 3, 64, 64)]
    for i, shp in enumerate(shapes):
      n = int(np.prod(shp))
      v['conv%d/w' % (i + 1)] = flat[c.conv_w[i]:c.conv_w[i] + n].reshape(shp)
      v['conv%d/b' % (i + 1)] = flat[c.conv_b[i]:c.conv_b[i] + shp[3]]
    for part, wo, bo in (('mu', c.fc1_mu_w, c.fc1_mu_b),
                         ('sigma', c.fc1_sig_w, c.fc1_sig_b)):
      l1 = int(c.fc1_ld)  # padded row pitch
      w = flat[wo:wo + FLAT * l1].reshape(FLAT, l1)
      b = flat[bo:bo + 1024]
      v['adv1/%s/w' % part], v['val1/%s/w' % part] = w[:, :512], w[:, 512:1024]
      v['adv1/%s/b' % part], v['val1/%s/b' % part] = b[:512], b[512:]
    la, lv = int(c.adv2_ld), int(c.val2_ld)  # padded leading dimensions
    v['adv2/mu/w'] = flat[c.adv2_mu_w:c.adv2_mu_w + HIDDEN * la].reshape(HIDDEN, la)[:, :na]
    v['adv2/sigma/w'] = flat[c.adv2_sig_w:c.adv2_sig_w + HIDDEN * la].reshape(HIDDEN, la)[:, :na]
    v['val2/mu/w'] = flat[c.val2_mu_w:c.val2_mu_w + HIDDEN * lv].reshape(HIDDEN, lv)[:, :k]
    v['val2/sigma/w'] = flat[c.val2_sig_w:c.val2_sig_w + HIDDEN * lv].reshape(HIDDEN, lv)[:, :k]
    v['adv2/sigma/b'] = flat[c.fc2_sig_b:c.fc2_sig_b + na]
    v['val2/sigma/b'] = flat[c.fc2_sig_b + la:c.fc2_sig_b + la + k]
    return v

  def pack(self, params: dict) -> np.ndarray:
    flat = np.zeros(self.param_count, np.float32)
    views = self._views(flat)
    if set(views) != set(params):
      raise ValueError('parameter names differ: %s' %
                       sorted(set(views) ^ set(params)))
    for name, view in views.items():
      view[...] = np.asarray(params[name], dtype=np.float32)
    return flat

  def unpack(self, flat: np.ndarray) -> dict:
    flat = np.asarray(flat, dtype=np.float32)
    return {k: np.array(v) for k, v in self._views(flat).items()}

  # -- noise ----------------------------------------------------------------
  def pack_noise(self, noise: dict) -> np.ndarray:
    """One apply's noise dict ('adv1/in', 'adv1/out', ...) -> flat block."""
    c = self.c
    out = np.zeros(self.noise_stride, np.float32)
    out[c.n_adv1_in:c.n_adv1_in + FLAT] = noise['adv1/in']
    out[c.n_val1_in:c.n_val1_in + FLAT] = noise['val1/in']
    out[c.n_fc1_out:c.n_fc1_out + 512] = noise['adv1/out']
    out[c.n_fc1_out + 512:c.n_fc1_out + 1024] = noise['val1/out']
    out[c.n_adv2_in:c.n_adv2_in + HIDDEN] = noise['adv2/in']
    out[c.n_val2_in:c.n_val2_in + HIDDEN] = noise['val2/in']
    out[c.n_fc2_out:c.n_fc2_out + self.na] = noise['adv2/out']
    vo = c.n_fc2_out + self.val_off
    out[vo:vo + self.num_atoms] = noise['val2/out']
    return out

  def unpack_noise(self, block: np.ndarray) -> dict:
    c = self.c
    return {
        'adv1/in': block[c.n_adv1_in:c.n_adv1_in + FLAT].copy(),
        'val1/in': block[c.n_val1_in:c.n_val1_in + FLAT].copy(),
        'adv1/out': block[c.n_fc1_out:c.n_fc1_out + 512].copy(),
        'val1/out': block[c.n_fc1_out + 512:c.n_fc1_out + 1024].copy(),
        'adv2/in': block[c.n_adv2_in:c.n_adv2_in + HIDDEN].copy(),
        'val2/in': block[c.n_val2_in:c.n_val2_in + HIDDEN].copy(),
        'adv2/out': block[c.n_fc2_out:c.n_fc2_out + self.na].copy(),
        'val2/out': block[c.n_fc2_out + self.val_off:
                          c.n_fc2_out + self.val_off + self.num_atoms].copy(),
    }


# --------------------------------------------------------------------------- #
#  Dense-head networks: dqn, double_dqn, c51, qr (ref: networks.py:295-363)
# --------------------------------------------------------------------------- #
class QNetworkOutputs(typing.NamedTuple):
  q_values: typing.Any


class QRNetworkOutputs(typing.NamedTuple):
  q_values: typing.Any
  q_dist: typing.Any


class DenseNetwork:
  """Descriptor of `dqn_atari_network` / `double_dqn_atari_network` /
  `c51_atari_network` / `qr_atari_network`: dqn_torso + linear(512) + ReLU +
  linear(num_outputs); `shared_bias` is the single scalar bias of the
  double-DQN head (ref: networks.py:120-134, 338-349)."""

  KINDS = ('dqn', 'double_dqn', 'c51', 'qr')

  def __init__(self, kind: str, num_actions: int, support=None, quantiles=None):
    if kind not in self.KINDS:
      raise ValueError('unknown network kind %r' % kind)
    self.kind = kind
    self.num_actions = int(num_actions)
    self.shared_bias = kind == 'double_dqn'
    self.support = None
    self.quantiles = None
    self.num_atoms = 0
    if kind == 'c51':
      self.support = np.asarray(support, dtype=np.float32)
      if self.support.ndim != 1:
        raise ValueError('support must have rank 1')
      self.num_atoms = int(self.support.shape[0])
    elif kind == 'qr':
      self.quantiles = np.asarray(quantiles, dtype=np.float32)
      if self.quantiles.ndim != 1:
        raise ValueError('quantiles must have rank 1')
      self.num_atoms = int(self.quantiles.shape[0])
    self.num_outputs = self.num_actions * max(self.num_atoms, 1)

  def layout(self, batch_size: int, groups: int = 2) -> 'DenseParamLayout':
    return DenseParamLayout(self.num_outputs, self.shared_bias, batch_size, groups)

  def init(self, random_state: np.random.RandomState) -> dict:
    """U(+-1/sqrt(fan_in)) for weights and biases (ref: networks.py:58-79)."""
    p = {}

    def uni(shape, fan):
      c = np.sqrt(1.0 / fan)
      return random_state.uniform(-c, c, size=shape).astype(np.float32)

    for name, ks, ci, co in (('conv1', 8, 4, 32), ('conv2', 4, 32, 64),
                             ('conv3', 3, 64, 64)):
      p[name + '/w'] = uni((ks, ks, ci, co), ci * ks * ks)
      p[name + '/b'] = uni((co,), ci * ks * ks)
    p['fc1/w'] = uni((FLAT, HIDDEN), FLAT)
    p['fc1/b'] = uni((HIDDEN,), FLAT)
    p['fc2/w'] = uni((HIDDEN, self.num_outputs), HIDDEN)
    p['fc2/b'] = uni((1,) if self.shared_bias else (self.num_outputs,), HIDDEN)
    return p


class DenseParamLayout:

  def __init__(self, num_outputs, shared_bias, batch_size, groups):
    self.c = _lib.DenseLayout()
    _lib.check(_lib.load().dz_dense_layout(num_outputs, int(shared_bias),
                                           batch_size, groups,
                                           ctypes.byref(self.c)),
               'dz_dense_layout')
    self.num_outputs = num_outputs
    self.shared_bias = bool(shared_bias)

  @property
  def param_count(self):
    return int(self.c.param_count)

  @property
  def ws_count(self):
    return int(self.c.ws_count)

  def _views(self, flat):
    c = self.c
    v = {}
    shapes = [(8, 8, 4, 32), (4, 4, 32, 64), (3, 3, 64, 64)]
    for i, shp in enumerate(shapes):
      n = int(np.prod(shp))
      v['conv%d/w' % (i + 1)] = flat[c.conv_w[i]:c.conv_w[i] + n].reshape(shp)
      v['conv%d/b' % (i + 1)] = flat[c.conv_b[i]:c.conv_b[i] + shp[3]]
    l1, l2, n = int(c.fc1_ld), int(c.fc2_ld), self.num_outputs
    v['fc1/w'] = flat[c.fc1_w:c.fc1_w + FLAT * l1].reshape(FLAT, l1)[:, :HIDDEN]
    v['fc1/b'] = flat[c.fc1_b:c.fc1_b + HIDDEN]
    v['fc2/w'] = flat[c.fc2_w:c.fc2_w + HIDDEN * l2].reshape(HIDDEN, l2)[:, :n]
    v['fc2/b'] = flat[c.fc2_b:c.fc2_b + (1 if self.shared_bias else n)]
    return v

  def pack(self, params: dict) -> np.ndarray:
    flat = np.zeros(self.param_count, np.float32)
    views = self._views(flat)
    if set(views) != set(params):
      raise ValueError('parameter names differ: %s' %
                       sorted(set(views) ^ set(params)))
    for name, view in views.items():
      view[...] = np.asarray(params[name], dtype=np.float32)
    return flat

  def unpack(self, flat: np.ndarray) -> dict:
    flat = np.asarray(flat, dtype=np.float32)
    return {k: np.array(v) for k, v in self._views(flat).items()}


# --------------------------------------------------------------------------- #
#  IQN (ref: networks.py:40-53, 264-292)
# --------------------------------------------------------------------------- #
class IqnInputs(typing.NamedTuple):
  state: typing.Any
  taus: typing.Any


class IqnOutputs(typing.NamedTuple):
  q_values: typing.Any
  q_dist: typing.Any


class IqnNetwork:
  """Descriptor of `iqn_atari_network(num_actions, latent_dim)`: dqn_torso, a
  cosine tau embedding mapped by linear(3136) + ReLU, multiplied into the state
  embedding, then the dqn value head applied per (batch, sample) row."""

  kind = 'iqn'

  def __init__(self, num_actions: int, latent_dim: int = 64):
    if latent_dim % 16 or latent_dim < 16:
      raise ValueError('latent_dim must be a positive multiple of 16')
    self.num_actions = int(num_actions)
    self.latent_dim = int(latent_dim)

  def layout(self, batch_size: int, samples=(64, 64, 64)) -> 'IqnParamLayout':
    return IqnParamLayout(self.num_actions, self.latent_dim, batch_size, samples)

  def init(self, random_state: np.random.RandomState) -> dict:
    """U(+-1/sqrt(fan_in)) (ref: networks.py:58-79); creation order torso,
    tau-embedding linear, value head (networks.py:272-287)."""
    p = {}

    def uni(shape, fan):
      c = np.sqrt(1.0 / fan)
      return random_state.uniform(-c, c, size=shape).astype(np.float32)

    for name, ks, ci, co in (('conv1', 8, 4, 32), ('conv2', 4, 32, 64),
                             ('conv3', 3, 64, 64)):
      p[name + '/w'] = uni((ks, ks, ci, co), ci * ks * ks)
      p[name + '/b'] = uni((co,), ci * ks * ks)
    p['emb/w'] = uni((self.latent_dim, FLAT), self.latent_dim)
    p['emb/b'] = uni((FLAT,), self.latent_dim)
    p['fc1/w'] = uni((FLAT, HIDDEN), FLAT)
    p['fc1/b'] = uni((HIDDEN,), FLAT)
    p['fc2/w'] = uni((HIDDEN, self.num_actions), HIDDEN)
    p['fc2/b'] = uni((self.num_actions,), HIDDEN)
    return p


class IqnParamLayout:

  def __init__(self, num_actions, latent_dim, batch_size, samples):
    self.c = _lib.IqnLayout()
    n0, n1, n2 = (int(x) for x in samples)
    _lib.check(_lib.load().dz_iqn_layout(num_actions, latent_dim, batch_size, n0,
                                         n1, n2, ctypes.byref(self.c)),
               'dz_iqn_layout')
    self.num_actions = int(num_actions)
    self.latent_dim = int(latent_dim)

  @property
  def param_count(self):
    return int(self.c.param_count)

  @property
  def ws_count(self):
    return int(self.c.ws_count)

  def _views(self, flat):
    c = self.c
    v = {}
    shapes = [(8, 8, 4, 32), (4, 4, 32, 64), (3, 3, 64, 64)]
    for i, shp in enumerate(shapes):
      n = int(np.prod(shp))
      v['conv%d/w' % (i + 1)] = flat[c.conv_w[i]:c.conv_w[i] + n].reshape(shp)
      v['conv%d/b' % (i + 1)] = flat[c.conv_b[i]:c.conv_b[i] + shp[3]]
    le, l1, l2 = int(c.emb_ld), int(c.fc1_ld), int(c.fc2_ld)
    a, lat = self.num_actions, self.latent_dim
    v['emb/w'] = flat[c.emb_w:c.emb_w + lat * le].reshape(lat, le)[:, :FLAT]
    v['emb/b'] = flat[c.emb_b:c.emb_b + FLAT]
    v['fc1/w'] = flat[c.fc1_w:c.fc1_w + FLAT * l1].reshape(FLAT, l1)[:, :HIDDEN]
    v['fc1/b'] = flat[c.fc1_b:c.fc1_b + HIDDEN]
    v['fc2/w'] = flat[c.fc2_w:c.fc2_w + HIDDEN * l2].reshape(HIDDEN, l2)[:, :a]
    v['fc2/b'] = flat[c.fc2_b:c.fc2_b + a]
    return v

  def pack(self, params: dict) -> np.ndarray:
    flat = np.zeros(self.param_count, np.float32)
    views = self._views(flat)
    if set(views) != set(params):
      raise ValueError('parameter names differ: %s' %
                       sorted(set(views) ^ set(params)))
    for name, view in views.items():
      view[...] = np.asarray(params[name], dtype=np.float32)
    return flat

  def unpack(self, flat: np.ndarray) -> dict:
    flat = np.asarray(flat, dtype=np.float32)
    return {k: np.array(v) for k, v in self._views(flat).items()}
