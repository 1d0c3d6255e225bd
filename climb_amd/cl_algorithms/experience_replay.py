"""Experience Replay with the reference's plugin surface (REF/cl_algorithms/experience_replay.py:17-122).

Semantics kept exactly: the buffer stores dataset INDICES (random.sample, :102-108); a replay step builds a FRESH AdamW
(zero moments, the task's base lr, no scheduler, :61) and runs one full `train_step` on the replayed task's head and
loss (:63).  The reference does NOT concatenate replay samples into the current minibatch."""
from __future__ import annotations

import argparse
import logging
import random
from typing import Dict

import torch

from ..utils import wandb_logger

logger = logging.getLogger(__name__)


class ExperienceReplayMemory:
    def __init__(self):
        self.memory_buffers = {}

    def add_task_memory_buffer(self, args: argparse.Namespace, task_key: str, task_config: Dict, task_trainer, memory_percentage: float,
                               sampling_strategy: str):
        self.memory_buffers[task_key] = TaskMemoryBuffer(args, task_key, task_config, task_trainer, memory_percentage, sampling_strategy)

    def do_replay(self) -> bool:
        return True if len(self.memory_buffers) > 0 else False

    def sample_replay_task(self) -> str:
        return random.choice(list(self.memory_buffers.keys()))

    def run_replay_step(self, task_key: str, model) -> torch.Tensor:
        task_buffer = self.memory_buffers[task_key]
        task_trainer = task_buffer.task_trainer
        optimizer = model.create_optimizer(task_trainer.hparams)
        replay_batch = task_buffer.sample_replay_batch()
        replay_loss, output, _, _ = task_trainer.train_step(model, replay_batch, optimizer)
        logger.info("{} replay step: loss = {:.5f}".format(task_buffer.task_config["task_name"], replay_loss.item()))
        wandb_logger.log({task_key: {"loss": replay_loss.item()}})
        return replay_loss


class TaskMemoryBuffer:
    def __init__(self, args: argparse.Namespace, task_key: str, task_config: Dict, task_trainer, memory_percentage: float, sampling_strategy: str):
        self.task_key = task_key
        self.task_name = task_config["task_name"]
        self.task_config = task_config
        self.task_trainer = task_trainer
        self.dataset = task_trainer.get_train_dataloader().dataset
        self.batch_collate_fn = task_trainer.get_collate_fn()
        if task_key == "nlvr2":
            self.batch_size = int(args.batch_size / 2)
        elif task_key == "vcr":
            self.batch_size = int(args.batch_size / 4)
        else:
            self.batch_size = args.batch_size
        self.memory_percentage = memory_percentage
        assert self.memory_percentage < 1.0
        self.memory_size = int(memory_percentage * len(self.dataset))
        self.sampling_strategy = sampling_strategy
        assert sampling_strategy in ["random"]
        train_idxs = list(range(len(self.dataset)))
        self.memory_idxs = random.sample(train_idxs, self.memory_size)
        logger.info("Created {} replay memory buffer, with {} samples in the memory".format(self.task_name, len(self.memory_idxs)))

    def __len__(self):
        return len(self.memory_idxs)

    def sample_replay_batch(self) -> Dict:
        sampled_instances = random.sample(self.memory_idxs, self.batch_size)
        return self.batch_collate_fn([self.dataset[i] for i in sampled_instances])
