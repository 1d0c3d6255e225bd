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
    """Driver-facing switchboard (same constructor keywords and method names as the reference's handler): one adapter per task is
    inserted up front, then exactly one of them is trained / active at a time while the base encoder stays frozen."""

    def __init__(self, adapter_method, args):
        if adapter_method not in SUPPORTED_ADAPTER_METHODS:
            raise NotImplementedError(f"adapter method {adapter_method!r}; supported: {SUPPORTED_ADAPTER_METHODS}")
        self.adapter_method, self.args = adapter_method, args
        preset = AdapterConfig.load(args.adapter_config)
        override = getattr(args, "adapter_reduction_factor", 0)
        if override and override > 0:                      # the published runs use bottleneck = hidden / 16
            preset = AdapterConfig.from_dict({**preset.to_dict(), "reduction_factor": override})
        self.adapter_config = preset
        logger.info("adapter configuration: %s", dict(self.adapter_config))

    def add_adapters_to_model(self, model):
        """Every task of the CL sequence gets its adapter before training starts (parameter layout is fixed once)."""
        for key in self.args.ordered_cl_tasks:
            model.add_adapter(key, config=self.adapter_config)

    def activate_adapter_for_training(self, task_key: str, model):
        """Freeze everything but `task_key`'s adapter (and the task heads) and route the forward pass through it."""
        model.train_adapter(task_key)
        model.set_active_adapters(task_key)

    def activate_adapter_for_eval(self, task_key: str, model):
        """Route the forward pass through `task_key`'s adapter; trainability is left as it is."""
        model.set_active_adapters(task_key)
