from .task_trainer import (TaskTrainer, VLTaskTrainer, VQATrainer, NLVR2Trainer, SNLIVETrainer, VCRTrainer, LowShotMixin, LowShotVQATrainer,
                           LowShotNLVR2Trainer, LowShotSNLIVETrainer, LowShotVCRTrainer, polynomial_decay_schedule_with_warmup)
