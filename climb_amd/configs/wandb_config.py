"""REF/configs/wandb_config.py: the keys the driver and the logger read (credentials are the user's to fill in)."""
wandb_config = {"entity": "", "api_key": "", "project_name": "climb-cl", "log_freq": 100}
