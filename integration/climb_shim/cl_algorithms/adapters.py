from climb_amd.cl_algorithms.adapters import *  # noqa: F401,F403
from climb_amd.cl_algorithms.adapters import ADAPTER_MAP, SUPPORTED_ADAPTER_METHODS, AdapterHandler  # noqa: F401
