from climb_amd.configs.model_configs import model_configs, ALLOWED_CL_ENCODERS  # noqa: F401
