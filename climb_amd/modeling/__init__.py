"""Encoder registry under the reference's names (REF/modeling/__init__.py:4-12): the upstream driver looks the encoder up by
`args.encoder_name` in these two maps: ViLT, and ViLT-BERT."""
from .vilt import (ViltContinualLearner, ViltEncoderWrapper, convert_batch_to_vilt_input_dict, create_vilt_continual_learner_model,
                   load_vilt_encoder)

from .viltbert import (ViltBertContinualLearner, ViltBertEncoderWrapper, convert_batch_to_viltbert_input_dict,
                       create_viltbert_continual_learner_model, load_viltbert_encoder)

_ENCODERS = {
    "vilt": (load_vilt_encoder, create_vilt_continual_learner_model),
    "viltbert": (load_viltbert_encoder, create_viltbert_continual_learner_model),
}
load_encoder_map = {name: fns[0] for name, fns in _ENCODERS.items()}
create_continual_learner_map = {name: fns[1] for name, fns in _ENCODERS.items()}

__all__ = ["load_encoder_map", "create_continual_learner_map", "ViltContinualLearner", "ViltEncoderWrapper",
           "convert_batch_to_vilt_input_dict", "create_vilt_continual_learner_model", "load_vilt_encoder", "ViltBertContinualLearner",
           "ViltBertEncoderWrapper", "convert_batch_to_viltbert_input_dict", "create_viltbert_continual_learner_model", "load_viltbert_encoder"]
