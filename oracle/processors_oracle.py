"""CPU restatement of the observation arithmetic of dqn_zoo's Atari preprocessing.

TEST INFRASTRUCTURE ONLY: nothing under `dqn_zoo_amd/` may import this module.

What it restates (ref: dqn_zoo/processors.py):
  * max-pool of the last two RGB frames                       processors.py:489-490
  * `rgb2y`: tensordot with [0.299, 0.587, 1-(0.299+0.587)] in float64, then
    `.astype(uint8)` (truncation)                              processors.py:367-371
  * `resize((84, 84))`: PIL `Image.resize(..., BILINEAR)` on an 8-bit image
                                                               processors.py:374-387
  * frame stacking with trailing zero padding, channel-last    processors.py:495-504

PINNED: `atari_observation()` reproduces the sha256 that the reference's own test
holds for this path (processors_test.py:405-475, observation hash
0d158a8f45aa...): tests/test_oracle_processors.py.

Two facts the pin establishes (and that the HIP kernel relies on):
  1. rgb2y is `(r*c0 + g*c1) + b*c2`, evaluated left to right in IEEE float64
     WITHOUT fused multiply-add.  NumPy's tensordot calls BLAS; the OpenBLAS in
     this container contracts it as fma(b,c2, fma(r,c0, g*c1)), which differs
     from the reference's pinned result in 522 of the 2^24 (r,g,b) triples --
     the reference code run HERE does not reproduce its own golden hash, this
     restatement does.
  2. Pillow's BILINEAR resample of an 8-bit image is integer arithmetic on
     22-bit fixed-point coefficients (libImaging/Resample.c, algorithm unchanged
     through Pillow 12.2): restated below from its published source and checked
     bit-for-bit against `PIL.Image.resize` on random images.
"""

import math

import numpy as np

RGB2Y = (0.299, 0.587, 1 - (0.299 + 0.587))
PRECISION_BITS = 32 - 8 - 2


def rgb2y(array: np.ndarray) -> np.ndarray:
  """uint8 [H,W,3] -> uint8 [H,W] (processors.py:367-371), no FMA."""
  a = array.astype(np.float64)
  y = (a[..., 0] * RGB2Y[0] + a[..., 1] * RGB2Y[1]) + a[..., 2] * RGB2Y[2]
  return y.astype(np.uint8)


def _bilinear(x):
  x = abs(x)
  return 1.0 - x if x < 1.0 else 0.0


def resample_coeffs(in_size: int, out_size: int):
  """Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter
  over the whole axis (box = full image): (bounds int32 [out,2] = (first input
  index, tap count), coefficients int32 [out, ksize])."""
  scale = filterscale = float(in_size) / out_size
  if filterscale < 1.0:
    filterscale = 1.0
  support = 1.0 * filterscale
  ksize = int(math.ceil(support)) * 2 + 1
  bounds = np.zeros((out_size, 2), np.int32)
  kk = np.zeros((out_size, ksize), np.int32)
  ss = 1.0 / filterscale
  for xx in range(out_size):
    center = (xx + 0.5) * scale
    xmin = int(center - support + 0.5)
    if xmin < 0:
      xmin = 0
    xmax = int(center + support + 0.5)
    if xmax > in_size:
      xmax = in_size
    xmax -= xmin
    w = [_bilinear((x + xmin - center + 0.5) * ss) for x in range(xmax)]
    ww = 0.0
    for v in w:
      ww += v
    for x in range(xmax):
      k = w[x] / ww if ww != 0.0 else w[x]
      kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else \
          int(0.5 + k * (1 << PRECISION_BITS))
    bounds[xx] = (xmin, xmax)
  return bounds, kk


def _pass(img, bounds, kk):
  """One 8-bit resample pass along axis 1: out[y, xx] = clip8((2^21 + sum_x
  img[y, xmin+x] * k[x]) >> 22)."""
  out = np.empty((img.shape[0], bounds.shape[0]), np.uint8)
  src = img.astype(np.int64)
  for xx in range(bounds.shape[0]):
    xmin, n = bounds[xx]
    acc = (1 << (PRECISION_BITS - 1)) + (src[:, xmin:xmin + n] *
                                         kk[xx, :n].astype(np.int64)).sum(axis=1)
    out[:, xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
  return out


def resize_bilinear(gray: np.ndarray, shape=(84, 84)) -> np.ndarray:
  """PIL `Image.fromarray(gray).resize((W, H), BILINEAR)` for uint8 [H,W]:
  horizontal pass first, 8-bit intermediate, then the vertical pass."""
  oh, ow = shape
  img = gray
  if ow != gray.shape[1]:
    img = _pass(img, *resample_coeffs(gray.shape[1], ow))
  if oh != gray.shape[0]:
    img = _pass(np.ascontiguousarray(img.T), *resample_coeffs(gray.shape[0], oh)).T
  return np.ascontiguousarray(img)


def resize_bilinear_rgb(rgb: np.ndarray, shape=(84, 84)) -> np.ndarray:
  """PIL `Image.fromarray(rgb).resize((W, H), BILINEAR)` for uint8 [H,W,3] (mode "RGB"):
  Pillow's 8-bit resampler treats every band of a multi-band image on its own with the
  same coefficient tables (ImagingResampleHorizontal/Vertical_8bpc), i.e. three
  single-band resizes (checked against PIL in tests/test_oracle_processors.py)."""
  return np.stack([resize_bilinear(np.ascontiguousarray(rgb[:, :, c]), shape)
                   for c in range(rgb.shape[2])], axis=-1)


def pooled_frame(frames, shape=(84, 84), grayscaling=True) -> np.ndarray:
  """max over the given RGB frames -> grayscale -> resize: one stack entry
  (processors.py:488-494; grayscaling=False keeps the three bands)."""
  pooled = np.max(np.stack(frames, axis=0), axis=0)
  if not grayscaling:
    return resize_bilinear_rgb(pooled, shape) if pooled.ndim == 3 else resize_bilinear(pooled, shape)
  return resize_bilinear(rgb2y(pooled), shape)


def stack_frames(frames, num_stacked=4) -> np.ndarray:
  """Deque(max_length) + trailing_zero_pad + np.stack(axis=-1)
  (processors.py:495-504): oldest frame first, zeros after."""
  frames = list(frames)[-num_stacked:]
  frames = frames + [np.zeros_like(frames[0])] * (num_stacked - len(frames))
  return np.stack(frames, axis=-1)
