"""The slice of dqn_zoo's `processors.py` the agents touch.

Atari preprocessing itself (ref: processors.py:421-508) is CPU-side, per-frame
and upstream of the replay: out of scope for the hot path.
Agents only need `reset(processor)` (ref: processors.py, used at
rainbow/agent.py:168) and the processor call protocol: a processor maps a
TimeStep to a TimeStep or to None ("repeat the previous action").
"""

from typing import Any, Callable, Optional

Processor = Callable[[Any], Optional[Any]]


def reset(processor: Processor) -> None:
  """Calls `reset()` on a processor (and on a Sequential's members) if present."""
  if hasattr(processor, 'reset'):
    processor.reset()


class Identity:
  """Processor for already-preprocessed observations (uint8 84x84x4 stacks)."""

  def __call__(self, timestep):
    return timestep

  def reset(self) -> None:
    pass
