"""Abstract bases with the reference's names (REF/modeling/continual_learner.py:5-22)."""
import torch.nn as nn


class _CheckpointCompat:
    """Checkpoints written through transformers 4.x (the lineage the reference pins) carry the non-parameter buffer
    `...text_embeddings.position_ids`; 5.x dropped it from the state dict.  Accept both: such keys are ignored on load."""

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        state_dict = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}
        return super().load_state_dict(state_dict, strict=strict, **kw)


def _abstract(what: str):
    def method(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} must implement {what}")
    method.__name__ = what
    return method


class EncoderWrapper(_CheckpointCompat, nn.Module):
    """Base of a vision-language encoder: `forward(**encodings) -> pooled features [B, encoder_dim]`."""

    def __init__(self, **_unused):
        nn.Module.__init__(self)

    forward = _abstract("forward")


class ContinualLearner(_CheckpointCompat, nn.Module):
    """Base of encoder + per-task heads: `forward(task_key=, images=, texts=) -> (pooled, logits)`, `get_encoder()`."""

    def __init__(self, **_unused):
        nn.Module.__init__(self)

    forward = _abstract("forward")
    get_encoder = _abstract("get_encoder")
