from .experience_replay import ExperienceReplayMemory
from .ewc import EWC
from .adapters import AdapterHandler
