"""Registration point with the reference's names (REF/modeling/__init__.py:4-12)."""
from .vilt import load_vilt_encoder, create_vilt_continual_learner_model

load_encoder_map = {"vilt": load_vilt_encoder}
create_continual_learner_map = {"vilt": create_vilt_continual_learner_model}
