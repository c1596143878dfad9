"""Row f4 on the GPU: the HIP observation pipeline (dz_atari_observation, through
the C ABI) against
  * the sha256 the REFERENCE's own test holds (processors_test.py:405-475);
  * the CPU oracle (itself pinned to that hash) on random episode streams with
    FIRST / LAST / reset cycles, other geometries and pooling depths -- every
    emitted observation bit-exact;
  * all 2^24 colours for rgb2y."""

import hashlib

import numpy as np
import pytest
import torch

from dqn_zoo_amd import dm_env_shim as dm_env
from dqn_zoo_amd import processors
from oracle import processors_oracle as po
from tests import test_oracle_processors as top
from tests import test_processors as tp

pytestmark = pytest.mark.gpu

F, M = dm_env.StepType.FIRST, dm_env.StepType.MID


@pytest.mark.parametrize('device_obs', [False, True])
def test_reference_golden_hash_through_atari(device_obs):
  """processors_test.py:405-475, verbatim protocol, pixel path on the device."""
  rgb = top.fixed_frames()
  step_types = [F, M, M, M, M]
  rewards = [None, 0.5, 0.2, 0, 0.1]
  discounts = [None, 0.9, 0.9, 0.9, 0.9]
  proc = processors.atari(device_observations=device_obs)
  for i in range(5):
    out = proc(dm_env.TimeStep(step_types[i], rewards[i], discounts[i], (rgb[i], 3)))
  assert out is not None and out.step_type == dm_env.StepType.MID
  assert abs(out.reward - (0.5 + 0.2 + 0.0 + 0.1)) < 1e-12
  assert abs(out.discount - 0.9 ** 4 * 0.99) < 1e-12
  obs = out.observation.cpu().numpy() if device_obs else out.observation
  assert obs.shape == (84, 84, 4) and obs.dtype == np.uint8
  assert hashlib.sha256(obs.flatten()).hexdigest() == top.REF_OBS_SHA


@pytest.mark.parametrize('seed,repeats,pooled', [(0, 4, 2), (1, 4, 1), (2, 3, 2), (3, 5, 4)])
def test_episode_streams_equal_oracle(seed, repeats, pooled):
  stream = tp.random_episodes(np.random.RandomState(seed), 150)
  dev = processors.atari(num_action_repeats=repeats, num_pooled_frames=pooled)
  ref = processors.AtariPreprocessor(
      num_action_repeats=repeats, num_pooled_frames=pooled,
      observation_pipeline=tp.OraclePixels(pooled, 4))
  got, want = tp.drive(dev, stream), tp.drive(ref, stream)
  n = 0
  for g, w in zip(got, want):
    assert tp.same_timestep(g, w)
    n += g is not None
  assert n > 25


@pytest.mark.parametrize('in_shape,out_shape,stack', [
    ((210, 160), (84, 84), 4), ((250, 160), (105, 80), 2), ((100, 37), (84, 84), 1),
    ((84, 84), (84, 84), 3), ((50, 60), (84, 84), 8)])
def test_other_geometries(in_shape, out_shape, stack):
  rs = np.random.RandomState(in_shape[0] + stack)
  pipe = processors.ObservationPipeline(out_shape, 2, stack, True)
  frames, stacked = [], []
  for k in range(stack + 3):
    a = rs.randint(0, 256, in_shape + (3,), dtype=np.uint8)
    b = rs.randint(0, 256, in_shape + (3,), dtype=np.uint8)
    got = pipe([None, None, a, b])
    stacked.append(po.pooled_frame([a, b], out_shape))
    want = po.stack_frames(stacked, stack)
    np.testing.assert_array_equal(got, want)
  pipe.reset()
  a = rs.randint(0, 256, in_shape + (3,), dtype=np.uint8)
  np.testing.assert_array_equal(
      pipe([None, a]), po.stack_frames([po.pooled_frame([a], out_shape)], stack))
  # both pooled slots are padding (M L ~ ~): a black frame enters the stack
  got = pipe([a, a, None, None])
  if stack == 1:
    assert (got == 0).all()
  else:
    assert (got[..., 1] == 0).all()
    assert (got[..., 0] == po.pooled_frame([a], out_shape)).all()


@pytest.mark.parametrize('device_obs', [False, True])
def test_grayscaling_false_keeps_rgb_bands(device_obs):
  """atari(grayscaling=False) (processors.py:429,495): pooled RGB frame -> PIL mode
  "RGB" resize (each band on its own) -> stack on a new last axis: observations
  [84, 84, 3, 4], bit-exact against the oracle (itself checked against PIL) on episode
  streams with FIRST / LAST / reset cycles; already-gray [H, W] frames still work."""
  stream = tp.random_episodes(np.random.RandomState(7), 90)
  dev = processors.atari(grayscaling=False, device_observations=device_obs)
  ref = processors.AtariPreprocessor(
      grayscaling=False, observation_pipeline=tp.OraclePixels(2, 4, grayscaling=False))
  got, want = tp.drive(dev, stream), tp.drive(ref, stream)
  n = 0
  for g, w in zip(got, want):
    if g is not None and device_obs:
      g = g._replace(observation=g.observation.cpu().numpy())
    assert tp.same_timestep(g, w)
    if g is not None:
      assert g.observation.shape == (84, 84, 3, 4) and g.observation.dtype == np.uint8
      n += 1
  assert n > 15
  # one frame, by hand: bands differ, each equals the single-band resize of its band
  rs = np.random.RandomState(3)
  a = rs.randint(0, 256, (210, 160, 3), dtype=np.uint8)
  pipe = processors.ObservationPipeline((84, 84), 1, 2, False)
  out = pipe([a])
  for c in range(3):
    np.testing.assert_array_equal(out[:, :, c, 0], po.resize_bilinear(np.ascontiguousarray(a[:, :, c])))
  assert (out[..., 1] == 0).all() and (out[:, :, 0, 0] != out[:, :, 1, 0]).any()
  gpipe = processors.ObservationPipeline((84, 84), 1, 1, False)
  g2 = rs.randint(0, 256, (210, 160), dtype=np.uint8)
  np.testing.assert_array_equal(gpipe([g2])[..., 0], po.resize_bilinear(g2))
  with pytest.raises(ValueError, match='frames must be uint8'):
    processors.ObservationPipeline((84, 84), 1, 1, True)([g2])


def test_rgb2y_all_colours_and_resize_processor():
  r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing='ij')
  arr = np.stack([r, g, b], -1).astype(np.uint8).reshape(4096, 4096, 3)
  for lo in range(0, 4096, 512):   # 8 slabs of 512 x 4096 colours
    np.testing.assert_array_equal(processors.rgb2y(arr[lo:lo + 512]),
                                  po.rgb2y(arr[lo:lo + 512]))
  rs = np.random.RandomState(5)
  gimg = rs.randint(0, 256, (210, 160), dtype=np.uint8)
  np.testing.assert_array_equal(processors.resize((84, 84))(gimg), po.resize_bilinear(gimg))
  with pytest.raises(ValueError, match='2D'):
    processors.resize((84, 84, 3))


def test_environment_wrapper_and_device_observations_reach_the_agent():
  """AtariEnvironmentWrapper on a synthetic (rgb, lives) environment; with
  device_observations=True the observation is a CUDA tensor that the agents'
  ObservationCache adopts without a copy."""
  from dqn_zoo_amd import device_obs

  class Spec:
    def __init__(self, shape, minimum=0):
      self.shape, self.minimum = shape, minimum

  class Env:
    def __init__(self):
      self.rs, self.t = np.random.RandomState(0), 0
    def observation_spec(self):
      return Spec((210, 160, 3)), Spec(())
    def action_spec(self):
      return Spec((), 0)
    def _obs(self):
      return (self.rs.randint(0, 256, (210, 160, 3), dtype=np.uint8), 3)
    def reset(self):
      self.t = 0
      return dm_env.restart(self._obs())
    def step(self, action):
      self.t += 1
      if self.t == 10:
        return dm_env.termination(1.0, self._obs())
      return dm_env.transition(1.0, self._obs())

  env = processors.AtariEnvironmentWrapper(Env(), device_observations=True)
  assert env.observation_spec().shape == (84, 84, 4)
  ts = env.reset()
  assert ts.first() and isinstance(ts.observation, torch.Tensor) and ts.observation.is_cuda
  cache = device_obs.ObservationCache(ts.observation.device)
  view = cache.upload(ts.observation)
  assert view.data_ptr() == ts.observation.data_ptr() and tuple(view.shape) == (1, 84, 84, 4)
  assert cache.lookup(ts.observation) is not None
  kinds = []
  for _ in range(4):
    ts = env.step(0)
    kinds.append(ts.step_type)
    if not ts.first():
      assert ts.reward == 1.0   # raw rewards of 1.0 summed, clipped to 1
  assert kinds[2] == dm_env.StepType.LAST and kinds[3] == dm_env.StepType.FIRST
