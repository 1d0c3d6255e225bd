"""Abstract bases with the reference's names (REF/modeling/continual_learner.py:5-22)."""
import torch.nn as nn


class _CheckpointCompat:
    """Checkpoints written through transformers 4.x (the lineage the reference pins) carry the non-parameter buffer
    `...text_embeddings.position_ids`; 5.x dropped it from the state dict.  Accept both: such keys are ignored on load."""

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        state_dict = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}
        return super().load_state_dict(state_dict, strict=strict, **kw)


class EncoderWrapper(_CheckpointCompat, nn.Module):
    def __init__(self, **kwargs):
        super().__init__()

    def forward(self, **kwargs):
        raise NotImplementedError


class ContinualLearner(_CheckpointCompat, nn.Module):
    def __init__(self, **kwargs):
        super().__init__()

    def forward(self, **kwargs):
        raise NotImplementedError

    def get_encoder(self):
        raise NotImplementedError
