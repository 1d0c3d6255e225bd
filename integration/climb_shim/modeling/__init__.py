from climb_amd.modeling import *  # noqa: F401,F403
from climb_amd.modeling import load_encoder_map, create_continual_learner_map  # noqa: F401
