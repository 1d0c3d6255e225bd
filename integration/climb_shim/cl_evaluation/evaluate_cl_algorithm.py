from climb_amd.cl_evaluation.evaluate_cl_algorithm import *  # noqa: F401,F403
from climb_amd.cl_evaluation.evaluate_cl_algorithm import upstream_knowledge_transfer_eval, catastrophic_forgetting_eval  # noqa: F401
