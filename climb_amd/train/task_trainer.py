"""Task trainers with the surface the reference's CL plugins use (SURVEY.md §8(b) "Trainer surface"):
`.hparams`, `.batch2inputs_converter`, `.get_train_dataloader()`, `.get_collate_fn()`, `.loss_criterion`, `.device`,
`.train_step(model, batch, optimizer=None, scheduler=None, ewc=None) -> (loss, output, ewc_task, ewc_loss)`, `.train`,
`.eval`, `.eval_forgetting`.

Semantics follow REF/train/visionlanguage_tasks/train_vqa.py (:121-282) and its NLVR2 / SNLI-VE / VCR copies (identical
but for names, dataloaders and the loss -- verified by diff).  `train_step` runs the fused HIP step
(forward + loss + backward [+ EWC term], no autograd graph); the optimizer is the fused AdamW.

Data parallel (the reference is single-process): when torch.distributed is initialised the loaders this class builds
are rank-sharded (`--batch_size` stays the GLOBAL batch: climb_amd/data/sharding.py), `train()` attaches the gradient all-reducer to the
model (parameters broadcast from rank 0, per-range all-reduce under the backward: climb_amd/parallel.py), `train_step` weights the shard's
d(loss) so that the averaged gradient IS the global-batch gradient, and `eval()` all-reduces the score sum.  Python's `random` (EWC's task
draw, the replay memory) is seeded identically on every rank and data order uses torch's generator, so every rank makes the same draws.

Construction matches the reference's call `Trainer(args, task_configs, model_config, device)`
(REF/train/train_upstream_continual_learning.py:240, :254, :314): the train / validation loaders are built from
`args.climb_data_dir` + the task's `data_dir` with climb_amd.data.datasets (the reference's file formats).  Tests and pipelines that
already hold loaders may inject them through the optional `train_dataloader` / `val_dataloader` keywords instead."""
from __future__ import annotations

import copy
import logging
import os
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from .. import parallel
from ..modeling.vilt import convert_batch_to_vilt_input_dict
from ..utils import wandb_logger

logger = logging.getLogger(__name__)


def polynomial_decay_schedule_with_warmup(optimizer, num_warmup_steps: int, num_training_steps: int, lr_end: float = 0.0, power: float = 1.0):
    """transformers.get_polynomial_decay_schedule_with_warmup as called at REF train_vqa.py:199-205 (restated so the
    product path has no transformers dependency)."""
    lr_init = optimizer.defaults["lr"]

    def lr_lambda(current_step: int):
        if current_step < num_warmup_steps:
            return float(current_step) / float(max(1, num_warmup_steps))
        if current_step > num_training_steps:
            return lr_end / lr_init
        lr_range = lr_init - lr_end
        decay_steps = num_training_steps - num_warmup_steps
        pct_remaining = 1 - (current_step - num_warmup_steps) / decay_steps
        return (lr_range * pct_remaining ** power + lr_end) / lr_init
    return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda)


class TaskTrainer(nn.Module):
    """REF/train/visionlanguage_tasks/task_trainer.py:5"""

    def __init__(self):
        super().__init__()


class VLTaskTrainer(TaskTrainer):
    task_key = None
    target_field = "labels"

    def __init__(self, args, task_configs: Dict, model_config: Dict, device: torch.device, train_dataloader=None, val_dataloader=None):
        super().__init__()
        self.args = args
        self.device = device
        self.task_config = task_configs[self.task_key]
        self.visual_input_type = model_config.get("visual_input_type", "pil-image")
        self.batch2inputs_converter = model_config.get("batch2inputs_converter", convert_batch_to_vilt_input_dict)
        if train_dataloader is None:
            if getattr(args, "climb_data_dir", None) is None:
                raise RuntimeError(f"{type(self).__name__}: needs args.climb_data_dir (the reference's data tree) or injected "
                                   "train_dataloader / val_dataloader")
            self.data_dir = os.path.join(args.climb_data_dir, self.task_config["data_dir"])
            train_dataloader, built_val = self.build_dataloaders(args, task_configs)
            val_dataloader = val_dataloader if val_dataloader is not None else built_val
        self.train_dataloader, self.val_dataloader = train_dataloader, val_dataloader
        self.num_epochs = self.task_config["num_epochs"]
        self.lr = self.task_config["lr"]
        self.adam_epsilon = self.task_config["adam_epsilon"]
        self.weight_decay = self.task_config["weight_decay"]
        self.hparams = {"lr": self.lr, "weight_decay": self.weight_decay, "adam_epsilon": self.adam_epsilon}
        self.loss_criterion = nn.BCEWithLogitsLoss(reduction="mean") if self.task_key == "vqa" else nn.CrossEntropyLoss()
        self.max_steps = len(self.train_dataloader) * self.num_epochs
        self.warmup_ratio = 0.1       # hard-coded in the reference too (train_vqa.py:97)

    def build_dataloaders(self, args, task_configs: Dict):
        """(train loader, validation loader) from the reference's data tree; one override per task below."""
        raise NotImplementedError

    def get_train_dataloader(self):
        return self.train_dataloader

    def get_collate_fn(self):
        return self.train_dataloader.collate_fn

    def prefetched(self, model, loader):
        """The loader with the host half of every batch (tokeniser, raw-byte staging, H2D copies, device image kernels:
        `model.prepare_batch`) moved to a worker thread and a side HIP stream, two batches ahead of the step that consumes them
        (climb_amd/data/prefetch.py; the reference does all of it on the training thread, REF/modeling/vilt.py:83-96).
        CLIMB_AMD_PREFETCH=0, a CPU device or a model without `prepare_batch` leave the loader as it is."""
        if torch.device(self.device).type != "cuda" or os.environ.get("CLIMB_AMD_PREFETCH", "1") == "0" or not hasattr(model, "prepare_batch"):
            return loader
        from ..data import DeviceImagePipeline, PrefetchLoader
        pipe = self.__dict__.get("_prefetch_pipeline")
        if pipe is None:
            pipe = self.__dict__["_prefetch_pipeline"] = DeviceImagePipeline(self.device)      # the worker's own pinned staging ring
        return PrefetchLoader(loader, lambda b: model.prepare_batch(self.task_key, b, self.batch2inputs_converter, pipe), depth=2, device=self.device)

    # ---- REF train_vqa.py:121-133
    def forward_pass(self, model, batch: Dict, do_eval: bool = False) -> Tuple:
        inputs = self.batch2inputs_converter(batch)
        with torch.no_grad() if do_eval else torch.enable_grad():
            return model(task_key=self.task_key, **inputs)

    # ---- REF train_vqa.py:135-174
    def train_step(self, model, batch: Dict, optimizer=None, scheduler=None, ewc=None):
        inputs = self.batch2inputs_converter(batch)
        target = batch[self.target_field]
        # (`optimizer`: step() below is the next reader of the gradients -- the encoder's weight-gradient launch may carry the update, engine.defer_dw)
        loss, output, ewc_task, ewc_loss = model.fused_forward_backward(self.task_key, inputs["images"], inputs["texts"], target, ewc,
                                                                        grad_weight=batch.get("dp_weight", 1.0), optimizer=optimizer, dp_rows=batch.get("dp_rows"))
        if optimizer is not None:
            optimizer.step()
            if scheduler is not None:
                scheduler.step()
            optimizer.zero_grad()
        return loss, output, ewc_task, ewc_loss

    # ---- REF train_vqa.py:176-244
    def train(self, model, replay_memory=None, ewc=None):
        model.to(self.device)
        parallel.ensure_reducer(model)        # N > 1 ranks: broadcast rank 0's parameters, all-reduce gradient ranges under the backward
        do_replay = do_ewc = False
        if self.args.cl_algorithm == "experience_replay":
            assert replay_memory is not None
            do_replay = replay_memory.do_replay()
        elif self.args.cl_algorithm == "ewc":
            assert ewc is not None
            do_ewc = ewc.do_ewc()
        optimizer = model.create_optimizer(self.hparams)
        scheduler = polynomial_decay_schedule_with_warmup(optimizer, int(self.max_steps * self.warmup_ratio), self.max_steps, 0.0, 1.0)
        best_score = 0
        best_model = {"epoch": 0, "model": copy.deepcopy(model), "optimizer_state": optimizer.state_dict()}
        model.zero_grad()
        for epoch in range(self.num_epochs):
            model.train()
            for step, batch in enumerate(self.prefetched(model, self.train_dataloader)):
                loss, output, ewc_task, ewc_loss = self.train_step(model, batch, optimizer, scheduler, ewc)
                if do_replay and (step + 1) % self.args.replay_frequency == 0:
                    sampled_replay_task = replay_memory.sample_replay_task()
                    replay_memory.run_replay_step(task_key=sampled_replay_task, model=model)
                if (step + 1) % wandb_logger.get_log_freq() == 0:
                    log_dict = {self.task_key: {"loss": loss.item()}}
                    if ewc is not None and do_ewc:
                        log_dict[ewc_task] = {"ewc_loss": ewc_loss.item()}
                    wandb_logger.log(log_dict)
            eval_score = self.eval(model)
            logger.info("Evaluation after epoch {}: {:.2f}".format(epoch + 1, eval_score))
            wandb_logger.log({self.task_key: {"val_score": eval_score}})
            if eval_score > best_score:
                best_score = eval_score
                best_model["epoch"] = epoch
                best_model["model"] = copy.deepcopy(model)
        return best_score, best_model

    # ---- REF train_nlvr2.py:225-244 (argmax accuracy); VQA overrides the per-batch score
    def batch_score(self, logits: torch.Tensor, batch: Dict) -> torch.Tensor:
        return (logits.argmax(-1) == batch["labels"].to(logits.device)).sum()

    def eval(self, model) -> float:
        model.eval()
        score = torch.zeros((), dtype=torch.float64, device=self.device)
        for step, batch in enumerate(self.prefetched(model, self.val_dataloader)):
            output = self.forward_pass(model, batch, do_eval=True)
            score += self.batch_score(output[1], batch).double()      # accumulated on device: one sync per eval, not per batch
        if parallel.rank_world()[1] > 1 and self._val_is_sharded():
            import torch.distributed as dist
            dist.all_reduce(score)            # every rank scored its own shard of the validation set (REF train_vqa.py:258-263 sums over all of it)
        model.train()
        return float(score.item()) / len(self.val_dataloader.dataset) * 100.0

    def _val_is_sharded(self) -> bool:
        from ..data.sharding import ShardedDataLoader
        return isinstance(self.val_dataloader, ShardedDataLoader)

    def eval_forgetting(self, model, model_path: str) -> float:
        parallel.barrier()                    # rank 0 wrote the checkpoint (parallel.rank0_only_io); nobody reads it before it is complete
        model.to(self.device)
        model.load_state_dict(torch.load(model_path, map_location=self.device))      # (rank 0 saved from cuda:0: without map_location every rank would stage the checkpoint there)
        return self.eval(model)


class VQATrainer(VLTaskTrainer):
    task_key = "vqa"
    target_field = "target_scores"

    def build_dataloaders(self, args, task_configs):
        """REF train_vqa.py:64-83"""
        from ..data.datasets import MSCOCOImagesDataset, build_vqa_dataloader
        coco = task_configs[self.task_config["images_source"]]
        self.images_dataset = MSCOCOImagesDataset(os.path.join(args.climb_data_dir, coco["data_dir"]), getattr(args, "visual_input_type", self.visual_input_type))
        return tuple(build_vqa_dataloader(args=args, data_dir=self.data_dir, images_dataset=self.images_dataset, split=sp,
                                          visual_input_type=self.visual_input_type) for sp in ("train", "val"))

    def compute_score_with_logits(self, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """REF train_vqa.py:99-113: VQA score of the argmax answer."""
        idx = torch.max(logits, 1)[1]
        one_hots = torch.zeros_like(labels)
        one_hots.scatter_(1, idx.view(-1, 1), 1)
        return one_hots * labels

    def batch_score(self, logits, batch):
        target = batch["target_scores"].to(logits.device)
        return self.compute_score_with_logits(logits, target).sum()


class NLVR2Trainer(VLTaskTrainer):
    task_key = "nlvr2"

    def build_dataloaders(self, args, task_configs):
        """REF train_nlvr2.py:65-73"""
        from ..data.datasets import build_nlvr2_dataloader
        return tuple(build_nlvr2_dataloader(args=args, data_dir=self.data_dir, split=sp, visual_input_type=self.visual_input_type) for sp in ("train", "val"))


class SNLIVETrainer(VLTaskTrainer):
    task_key = "snli-ve"

    def build_dataloaders(self, args, task_configs):
        """REF train_snli_ve.py:64-81 (validation split is `dev`)"""
        from ..data.datasets import Flickr30KImagesDataset, build_snli_ve_dataloader
        flickr = task_configs[self.task_config["images_source"]]
        images = Flickr30KImagesDataset(os.path.join(args.climb_data_dir, flickr["data_dir"]), visual_input_type=self.visual_input_type)
        return tuple(build_snli_ve_dataloader(args=args, data_dir=self.data_dir, images_dataset=images, split=sp,
                                              visual_input_type=self.visual_input_type) for sp in ("train", "dev"))


class VCRTrainer(VLTaskTrainer):
    task_key = "vcr"

    def build_dataloaders(self, args, task_configs):
        """REF train_vcr.py:59-76"""
        from ..data.datasets import build_vcr_dataloader
        self.task_type = self.task_config["task_type"]
        return tuple(build_vcr_dataloader(args=args, data_dir=self.data_dir, split=sp, task_type=self.task_type,
                                          visual_input_type=self.visual_input_type) for sp in ("train", "val"))


# ------------------------------------------------------------------------------------------------ low-shot transfer
class LowShotMixin:
    """The reference's LowShot*Trainer classes (REF train_vqa.py:284-360, train_nlvr2.py:261-340, train_snli_ve.py:269-350,
    train_vcr.py:263-345): the task trainer on a sub-sampled training set (`low_shot_config`: a percentage of the examples, or N shots
    per class), evaluated only after the epochs the config names, no replay / EWC.  Constructed by REF/train/train_lowshot_multimodal.py:52
    as `trainer_class(args, task_configs, model_config, device, low_shot_config=...)`."""

    def __init__(self, args, task_configs: Dict, model_config: Dict, device: torch.device, low_shot_config: Dict = None, **kw):
        super().__init__(args, task_configs, model_config, device, **kw)
        if low_shot_config is None:
            raise ValueError(f"{type(self).__name__}: low_shot_config is required")
        self.low_shot_config = low_shot_config
        self.eval_epochs = [x - 1 for x in low_shot_config["eval_epochs"]]
        dataset = self.train_dataloader.dataset
        if low_shot_config["type"] == "percentage":
            dataset.convert_to_low_shot(low_shot_percentage=low_shot_config["percentage"])
        else:
            dataset.convert_to_low_shot(num_shots_per_class=low_shot_config["num_shots_per_class"])
        self.max_steps = len(self.train_dataloader) * self.num_epochs

    def train(self, model):
        model.to(self.device)
        parallel.ensure_reducer(model)
        optimizer = model.create_optimizer(self.hparams)
        scheduler = polynomial_decay_schedule_with_warmup(optimizer, int(self.max_steps * self.warmup_ratio), self.max_steps, 0.0, 1.0)
        best_score = 0
        best_model = {"epoch": 0, "model": copy.deepcopy(model), "optimizer_state": optimizer.state_dict()}
        model.zero_grad()
        for epoch in range(self.num_epochs):
            model.train()
            for step, batch in enumerate(self.prefetched(model, self.train_dataloader)):
                self.train_step(model, batch, optimizer, scheduler)
            if epoch in self.eval_epochs:
                eval_score = self.eval(model)
                logger.info("Evaluation after epoch {}: {:.2f}".format(epoch + 1, eval_score))
                wandb_logger.log({self.task_key: {"val_score": eval_score}})
                if eval_score > best_score:
                    logger.info("New best evaluation score: {:.2f}".format(eval_score))
                    best_score = eval_score
                    best_model["epoch"] = epoch
                    best_model["model"] = copy.deepcopy(model)
        return best_score, best_model


class LowShotVQATrainer(LowShotMixin, VQATrainer):
    pass


class LowShotNLVR2Trainer(LowShotMixin, NLVR2Trainer):
    pass


class LowShotSNLIVETrainer(LowShotMixin, SNLIVETrainer):
    pass


class LowShotVCRTrainer(LowShotMixin, VCRTrainer):
    pass
