"""Per-task hyper-parameters, values from REF/configs/task_configs.py:16-116 (the arithmetic-relevant keys, the low-shot
configurations and the trainer classes of this package)."""
from ..train.task_trainer import (VQATrainer, NLVR2Trainer, SNLIVETrainer, VCRTrainer, LowShotVQATrainer, LowShotNLVR2Trainer,
                                   LowShotSNLIVETrainer, LowShotVCRTrainer)

SUPPORTED_VL_TASKS = ["vqa", "nlvr2", "snli-ve", "vcr"]

vqa_config = {"task_name": "VQAv2", "data_dir": "vqav2/", "images_source": "ms-coco", "splits": ["train", "val"], "num_labels": 3129,
              "num_images": 1, "model_type": "classification", "num_epochs": 10, "lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8,
              "warmup_ratio": 0.1, "task_trainer": VQATrainer, "random_baseline_score": 0.0,
              "low_shot_config": {"task_trainer": LowShotVQATrainer, "type": "percentage", "percentage": 0.05, "eval_epochs": [6, 8, 10]}}
nlvr_config = {"task_name": "NLVRv2", "data_dir": "nlvr2/", "splits": ["train", "val"], "num_labels": 2, "num_images": 2,
               "model_type": "classification", "num_epochs": 10, "lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8, "warmup_ratio": 0.1,
               "task_trainer": NLVR2Trainer, "random_baseline_score": 50.0,
               "low_shot_config": {"task_trainer": LowShotNLVR2Trainer, "type": "n-shot-per-class", "num_shots_per_class": 2048, "eval_epochs": [6, 8, 10]}}
snli_ve_config = {"task_name": "SNLI-VE", "data_dir": "snli-ve/", "images_source": "flickr30k", "splits": ["train", "dev", "test"],
                  "num_labels": 3, "num_images": 1, "model_type": "classification", "num_epochs": 5, "lr": 5e-5, "weight_decay": 1e-2,
                  "adam_epsilon": 1e-8, "warmup_ratio": 0.1, "task_trainer": SNLIVETrainer, "random_baseline_score": 33.33,
                  "low_shot_config": {"task_trainer": LowShotSNLIVETrainer, "type": "n-shot-per-class", "num_shots_per_class": 2048, "eval_epochs": [2, 4, 5]}}
vcr_config = {"task_name": "VCR", "data_dir": "vcr/", "splits": ["train", "dev", "test"], "num_labels": 4, "num_images": 1,
              "model_type": "multi-choice", "task_type": "qa", "num_choices": 4, "num_epochs": 10, "lr": 1e-4, "weight_decay": 1e-2,
              "adam_epsilon": 1e-8, "warmup_ratio": 0.1, "task_trainer": VCRTrainer, "random_baseline_score": 25.0,
              "low_shot_config": {"task_trainer": LowShotVCRTrainer, "type": "percentage", "percentage": 0.05, "eval_epochs": [2, 4, 6, 8, 10]}}

task_configs = {"vqa": vqa_config, "nlvr2": nlvr_config, "snli-ve": snli_ve_config, "vcr": vcr_config,
                "ms-coco": {"data_dir": "ms-coco/"}, "flickr30k": {"data_dir": "flickr30k/"}}
