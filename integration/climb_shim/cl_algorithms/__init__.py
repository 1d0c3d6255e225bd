from climb_amd.cl_algorithms import *  # noqa: F401,F403
from climb_amd.cl_algorithms import ExperienceReplayMemory, EWC, AdapterHandler  # noqa: F401
