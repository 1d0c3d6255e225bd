from climb_amd.utils import set_seed  # noqa: F401
