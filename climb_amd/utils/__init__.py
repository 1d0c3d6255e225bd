"""Small host utilities mirroring REF/utils (seed_utils.py:5-8, wandb.py:10-31)."""
import logging
import random

import numpy as np
import torch


def set_seed(args):
    """REF/utils/seed_utils.py:5-8"""
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)


class WandBLogger:
    """REF/utils/wandb.py: a no-op unless initialised; `get_log_freq()` is 100 when uninitialised."""

    def __init__(self):
        self.is_initialized = False
        self.log_freq = 100

    def initialize(self, wandb_config=None, experiment_name=None):
        try:
            import os
            import wandb
            os.environ["WANDB_API_KEY"] = wandb_config["api_key"]
            wandb.init(entity=wandb_config["entity"], project=wandb_config["project_name"], name=experiment_name)
            self.is_initialized = True
            self.log_freq = wandb_config["log_freq"]
        except Exception as e:   # wandb is absent in the offline image
            logging.getLogger(__name__).warning("wandb unavailable: %s", e)

    def log(self, log_dict):
        if self.is_initialized:
            import wandb
            wandb.log(log_dict)

    def get_log_freq(self):
        return self.log_freq


wandb_logger = WandBLogger()
