"""Minimal stand-in for the parts of `dm_env` the hot path touches.

The reference depends on the `dm_env` package (docker_requirements.txt) for
`TimeStep`, `StepType` and the `restart/transition/termination/truncation`
constructors.  It is not installed in this image, so the same surface is
provided here; if the real package is importable it is used instead.
`StepType.FIRST == 0` is relied upon by the reference's processors
(processors.py:305).
"""

import enum
import typing
from typing import Any

try:  # pragma: no cover - real package not present in this image
  from dm_env import StepType, TimeStep, restart, transition, termination, truncation  # type: ignore  # noqa: F401
except ImportError:

  class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2

    def first(self) -> bool:
      return self is StepType.FIRST

    def mid(self) -> bool:
      return self is StepType.MID

    def last(self) -> bool:
      return self is StepType.LAST

  class TimeStep(typing.NamedTuple):
    step_type: Any
    reward: Any
    discount: Any
    observation: Any

    def first(self) -> bool:
      return self.step_type == StepType.FIRST

    def mid(self) -> bool:
      return self.step_type == StepType.MID

    def last(self) -> bool:
      return self.step_type == StepType.LAST

  def restart(observation):
    return TimeStep(StepType.FIRST, None, None, observation)

  def transition(reward, observation, discount=1.0):
    return TimeStep(StepType.MID, reward, discount, observation)

  def termination(reward, observation):
    return TimeStep(StepType.LAST, reward, 0.0, observation)

  def truncation(reward, observation, discount=1.0):
    return TimeStep(StepType.LAST, reward, discount, observation)
