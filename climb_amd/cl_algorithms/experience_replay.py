"""Experience Replay behind the reference's plugin surface (REF/cl_algorithms/experience_replay.py:17-122).

What is preserved, because the trainers, the upstream driver and the parity fixtures depend on it:
  * the memory of a finished task is a list of dataset INDICES drawn once with `random.sample` (:102-108) -- same RNG, same call
    order, so a seeded run picks the same examples as the reference;
  * a replay step samples `batch_size` (halved for NLVR2, quartered for VCR: every task runs the same number of encoder sequences,
    :93-98) of those indices, re-collates them, builds a FRESH optimizer (zero moments, the task's base lr, no scheduler, :61) and
    runs one ordinary `train_step` on the replayed task's own head and loss (:63).
The reference does NOT mix replay samples into the current minibatch; neither does this.

Data parallel: `random` is seeded identically on every rank, so the memory indices, `sample_replay_task()` and the
replay batch drawn here agree across ranks; each rank then loads only its strided share of that (global) replay batch and weights its
d(loss) like a training shard (climb_amd/data/sharding.py); the step's gradients are all-reduced like any other step's.
"""
from __future__ import annotations

import logging
import random
from typing import Dict, List

import torch

from ..data.sharding import dp_rank_world, shard_of
from ..utils import wandb_logger

logger = logging.getLogger(__name__)

# encoder sequences per example: a loader batch of B examples costs B * this many encoder passes
_SEQUENCES_PER_EXAMPLE = {"nlvr2": 2, "vcr": 4}
_STRATEGIES = ("random",)


class TaskMemoryBuffer:
    """Indices into one task's training set + what is needed to turn a draw of them into a batch."""

    def __init__(self, args, task_key: str, task_config: Dict, task_trainer, memory_percentage: float, sampling_strategy: str):
        if not memory_percentage < 1.0:
            raise AssertionError("memory_percentage must be a fraction of the training set")
        if sampling_strategy not in _STRATEGIES:
            raise AssertionError(f"sampling strategy {sampling_strategy!r} is not one of {_STRATEGIES}")
        self.task_key, self.task_config, self.task_name = task_key, task_config, task_config["task_name"]
        self.task_trainer = task_trainer
        self.dataset = task_trainer.get_train_dataloader().dataset
        self.batch_collate_fn = task_trainer.get_collate_fn()
        self.batch_size = int(args.batch_size / _SEQUENCES_PER_EXAMPLE.get(task_key, 1))
        self.memory_percentage, self.sampling_strategy = memory_percentage, sampling_strategy
        self.memory_size = int(memory_percentage * len(self.dataset))
        self.memory_idxs: List[int] = random.sample(list(range(len(self.dataset))), self.memory_size)
        logger.info("%s replay memory: %d of %d training examples", self.task_name, len(self.memory_idxs), len(self.dataset))

    def __len__(self) -> int:
        return len(self.memory_idxs)

    def sample_replay_batch(self) -> Dict:
        picked = random.sample(self.memory_idxs, self.batch_size)          # the same draw on every rank
        rank, world = dp_rank_world()
        mine, weight = shard_of(picked, rank, world)
        batch = self.batch_collate_fn([self.dataset[idx] for idx in mine])
        if world > 1:
            batch["dp_weight"] = weight
        return batch


class ExperienceReplayMemory:
    """One `TaskMemoryBuffer` per finished task; the trainer asks `do_replay()` / `sample_replay_task()` / `run_replay_step()`."""

    def __init__(self):
        self.memory_buffers: Dict[str, TaskMemoryBuffer] = {}

    def add_task_memory_buffer(self, args, task_key: str, task_config: Dict, task_trainer, memory_percentage: float, sampling_strategy: str):
        self.memory_buffers[task_key] = TaskMemoryBuffer(args, task_key, task_config, task_trainer, memory_percentage, sampling_strategy)

    def do_replay(self) -> bool:
        return bool(self.memory_buffers)

    def sample_replay_task(self) -> str:
        return random.choice(list(self.memory_buffers))

    def run_replay_step(self, task_key: str, model) -> torch.Tensor:
        buf = self.memory_buffers[task_key]
        trainer = buf.task_trainer
        fresh_optimizer = model.create_optimizer(trainer.hparams)          # new moments every time: the reference's behaviour
        loss, _output, _ewc_task, _ewc_loss = trainer.train_step(model, buf.sample_replay_batch(), fresh_optimizer)
        value = loss.item()
        logger.info("replay step on %s: loss %.5f", buf.task_name, value)
        wandb_logger.log({task_key: {"loss": value}})
        return loss
