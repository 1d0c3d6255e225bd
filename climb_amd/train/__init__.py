from .task_trainer import TaskTrainer, VLTaskTrainer, VQATrainer, NLVR2Trainer, SNLIVETrainer, VCRTrainer, polynomial_decay_schedule_with_warmup
