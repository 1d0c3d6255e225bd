from climb_amd.utils import wandb_logger, WandBLogger  # noqa: F401
