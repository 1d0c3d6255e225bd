from climb_amd.configs.task_configs import task_configs, SUPPORTED_VL_TASKS  # noqa: F401
