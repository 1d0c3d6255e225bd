from climb_amd.configs.wandb_config import wandb_config  # noqa: F401
