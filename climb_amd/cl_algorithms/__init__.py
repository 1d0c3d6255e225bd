"""Continual-learning plug-ins under the names the upstream driver imports (REF/train/train_upstream_continual_learning.py:26-27):
EWC (device-resident theta*, Fisher and penalty), Experience Replay, and the adapter handler."""
from . import adapters
from .adapters import AdapterHandler
from .ewc import EWC
from .experience_replay import ExperienceReplayMemory, TaskMemoryBuffer

__all__ = ["EWC", "ExperienceReplayMemory", "TaskMemoryBuffer", "AdapterHandler", "adapters"]
