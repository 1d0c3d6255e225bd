"""Adapter plugin surface (REF/cl_algorithms/adapters.py:27-65).

The arithmetic behind `model.add_adapter / train_adapter / set_active_adapters` lives in GLAMOR's un-vendored
adapter-transformers fork (REF/.gitmodules:1-3, empty directory), so parity for it is UNPINNED."""
from __future__ import annotations

import logging

logger = logging.getLogger(__name__)


class AdapterConfig(dict):
    """Minimal stand-in for `transformers.adapters.AdapterConfig` (load / to_dict / from_dict)."""
    PRESETS = {
        "houlsby": dict(mh_adapter=True, output_adapter=True, reduction_factor=16, non_linearity="swish"),
        "pfeiffer": dict(mh_adapter=False, output_adapter=True, reduction_factor=16, non_linearity="relu"),
    }

    @classmethod
    def load(cls, name, **kw):
        if isinstance(name, dict):
            return cls(**name)
        if name not in cls.PRESETS:
            raise NotImplementedError(f"adapter config '{name}' (only {sorted(cls.PRESETS)} are described)")
        return cls(name=name, **cls.PRESETS[name], **kw)

    def to_dict(self):
        return dict(self)

    @classmethod
    def from_dict(cls, d):
        return cls(**d)


ADAPTER_MAP = {"pfeiffer": "pfeiffer", "houlsby": "houlsby"}
SUPPORTED_ADAPTER_METHODS = ["vanilla"]


class AdapterHandler:
    def __init__(self, adapter_method, args):
        self.args = args
        self.adapter_method = adapter_method
        config_dict = AdapterConfig.load(args.adapter_config).to_dict()
        if args.adapter_reduction_factor > 0:
            config_dict["reduction_factor"] = args.adapter_reduction_factor
        self.adapter_config = AdapterConfig.from_dict(config_dict)
        logger.info("Adding Adapter layers with configuration: %s", self.adapter_config)

    def add_adapters_to_model(self, model):
        for task_key in self.args.ordered_cl_tasks:
            model.add_adapter(task_key, config=self.adapter_config)

    def activate_adapter_for_training(self, task_key: str, model):
        model.train_adapter(task_key)
        model.set_active_adapters(task_key)

    def activate_adapter_for_eval(self, task_key: str, model):
        model.set_active_adapters(task_key)
