from climb_amd.cl_algorithms.adapters import ADAPTER_MAP  # noqa: F401   (REF/configs/adapter_configs.py: imported by the low-shot driver)
