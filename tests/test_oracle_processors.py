"""Pins oracle/processors_oracle.py (the checker of row f4):
  * against the sha256 the REFERENCE's own test holds for the preprocessed
    observation (processors_test.py:405-475);
  * against PIL's resize, bit for bit, on random images and other geometries;
  * and records how the live reference differs in this container (BLAS FMA)."""

import hashlib

import numpy as np
import pytest

from oracle import processors_oracle as po
from oracle import ref_processors_loader as rpl

REF_OBS_SHA = '0d158a8f45aa09aa6fad0354d2eb1fc0e3f57add88e772f3b71f54819d8200aa'
REF_RGB_SHA = [
    '250557b2184381fc2ec541fc313127050098fce825a6e98a728c2993874db300',
    'db8054ca287971a0e1264bfbc5642233085f1b27efbca9082a29f5be8a24c552',
    '7016e737a257fcdb77e5f23daf96d94f9820bd7361766ca7b1401ec90984ef71',
    '356dfcf0c6eaa4e2b5e80f4611375c0131435cc22e6a413b573818d7d084e9b2',
    '73078bedd438422ad1c3dda6718aa1b54f6163f571d2c26ed714c515a6372159',
]


def fixed_frames():
  """The five frames of processors_test.py:418-447 (RandomState(1))."""
  rs = np.random.RandomState(seed=1)
  return [rs.randint(0, 256, size=(210, 160, 3), dtype=np.uint8) for _ in range(5)]


def test_reference_golden_observation_hash():
  """Timesteps F M M M M with frames 0..4: the processor emits max[0, f0] at the
  FIRST step and max[f3, f4] when the action-repeat buffer fills (the diagram at
  processors.py:441-445); the stack is [A, B, 0, 0]."""
  rgb = fixed_frames()
  assert [hashlib.sha256(o).hexdigest() for o in rgb] == REF_RGB_SHA
  a = po.pooled_frame([np.zeros_like(rgb[0]), rgb[0]])
  b = po.pooled_frame([rgb[3], rgb[4]])
  obs = po.stack_frames([a, b])
  assert obs.shape == (84, 84, 4) and obs.dtype == np.uint8
  assert hashlib.sha256(obs.flatten()).hexdigest() == REF_OBS_SHA


@pytest.mark.parametrize('in_shape,out_shape', [
    ((210, 160), (84, 84)), ((210, 160), (105, 80)), ((84, 84), (84, 84)),
    ((100, 37), (84, 84)), ((250, 160), (42, 64)), ((50, 60), (84, 84))])
def test_resize_equals_pillow(in_shape, out_shape):
  from PIL import Image
  rs = np.random.RandomState(in_shape[0] * 7 + out_shape[1])
  for _ in range(4):
    g = rs.randint(0, 256, size=in_shape, dtype=np.uint8)
    want = np.array(Image.fromarray(g).resize((out_shape[1], out_shape[0]),
                                              Image.Resampling.BILINEAR), dtype=np.uint8)
    np.testing.assert_array_equal(po.resize_bilinear(g, out_shape), want)
  # extreme values survive the fixed-point rounding
  for v in (0, 255):
    g = np.full(in_shape, v, np.uint8)
    assert (po.resize_bilinear(g, out_shape) == v).all()


@pytest.mark.parametrize('in_shape,out_shape', [
    ((210, 160), (84, 84)), ((100, 37), (84, 84)), ((250, 160), (42, 64))])
def test_rgb_resize_equals_pillow(in_shape, out_shape):
  """atari(grayscaling=False) (processors.py:429,495): the un-grayscaled RGB frame goes
  through PIL as mode "RGB" -- every band resampled on its own."""
  from PIL import Image
  rs = np.random.RandomState(in_shape[1] * 3 + out_shape[0])
  for _ in range(3):
    a = rs.randint(0, 256, size=in_shape + (3,), dtype=np.uint8)
    want = np.array(Image.fromarray(a).resize((out_shape[1], out_shape[0]),
                                              Image.Resampling.BILINEAR), dtype=np.uint8)
    assert want.shape == out_shape + (3,)
    np.testing.assert_array_equal(po.resize_bilinear_rgb(a, out_shape), want)
    np.testing.assert_array_equal(po.pooled_frame([a], out_shape, grayscaling=False), want)


def test_rgb2y_is_plain_left_to_right_float64():
  """All 2^24 colours: the restatement equals the un-fused evaluation; the
  known-answer corners are exact."""
  r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing='ij')
  arr = np.stack([r, g, b], -1).astype(np.uint8).reshape(4096, 4096, 3)
  y = po.rgb2y(arr)
  a = arr.astype(np.float64)
  np.testing.assert_array_equal(
      y, ((a[..., 0] * 0.299 + a[..., 1] * 0.587) + a[..., 2] * (1 - (0.299 + 0.587))
          ).astype(np.uint8))
  assert y[0, 0] == 0 and y.max() == 255 - 0  # white -> 255 or 254 is the point:
  assert int(po.rgb2y(np.full((1, 1, 3), 255, np.uint8))[0, 0]) in (254, 255)


@pytest.mark.skipif(not rpl.reference_available(),
                    reason='needs /root/reference (dev container only)')
def test_live_reference_agrees_except_for_blas_fma():
  """The reference module run in THIS container: resize identical; rgb2y
  identical except where this container's BLAS fuses multiply-adds (a property
  of the host's OpenBLAS kernel, not of dqn_zoo: with it the reference does not
  reproduce its own golden hash here)."""
  ref = rpl.load_reference_processors()
  rs = np.random.RandomState(3)
  g = rs.randint(0, 256, size=(210, 160), dtype=np.uint8)
  np.testing.assert_array_equal(ref.resize((84, 84))(g), po.resize_bilinear(g))
  rgb = rs.randint(0, 256, size=(210, 160, 3), dtype=np.uint8)
  diff = (ref.rgb2y(rgb).astype(int) - po.rgb2y(rgb).astype(int))
  assert np.abs(diff).max() <= 1 and (diff != 0).mean() < 1e-3
