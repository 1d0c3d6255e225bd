from .evaluate_cl_algorithm import (upstream_knowledge_transfer_eval, catastrophic_forgetting_eval, save_task_checkpoint,
                                    load_task_checkpoint, append_task_result)

__all__ = ["upstream_knowledge_transfer_eval", "catastrophic_forgetting_eval", "save_task_checkpoint", "load_task_checkpoint",
           "append_task_result"]
