"""Abstract bases with the reference's names (REF/modeling/continual_learner.py:5-22)."""
import torch.nn as nn


class EncoderWrapper(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()

    def forward(self, **kwargs):
        raise NotImplementedError


class ContinualLearner(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()

    def forward(self, **kwargs):
        raise NotImplementedError

    def get_encoder(self):
        raise NotImplementedError
