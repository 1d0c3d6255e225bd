"""REF/configs/model_configs.py:6-12, :58 -- the `vilt` entry, resolving to this package's classes."""
from ..modeling.vilt import ViltEncoderWrapper, convert_batch_to_vilt_input_dict

ALLOWED_CL_ENCODERS = ["vilt"]

vilt_config = {"encoder_dim": 768, "visual_input_type": "pil-image", "encoder_class": ViltEncoderWrapper,
               "batch2inputs_converter": convert_batch_to_vilt_input_dict, "encoder_name": "ViLT"}

model_configs = {"vilt": vilt_config}
