"""REF/configs/model_configs.py:6-12, :36-41, :58-63 -- the `vilt` and `viltbert` entries, resolving to this package's classes."""
from ..modeling.vilt import ViltEncoderWrapper, convert_batch_to_vilt_input_dict

from ..modeling.viltbert import ViltBertEncoderWrapper, convert_batch_to_viltbert_input_dict

ALLOWED_CL_ENCODERS = ["vilt", "viltbert"]

vilt_config = {"encoder_dim": 768, "visual_input_type": "pil-image", "encoder_class": ViltEncoderWrapper,
               "batch2inputs_converter": convert_batch_to_vilt_input_dict, "encoder_name": "ViLT"}

viltbert_config = {"encoder_dim": 768, "visual_input_type": "pil-image", "encoder_class": ViltBertEncoderWrapper,
                   "batch2inputs_converter": convert_batch_to_viltbert_input_dict, "encoder_name": "ViLT-BERT"}        # REF/configs/model_configs.py:36-41

model_configs = {"vilt": vilt_config, "viltbert": viltbert_config}
