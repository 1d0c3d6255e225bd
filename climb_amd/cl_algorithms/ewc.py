"""Elastic Weight Consolidation with the reference's plugin surface (REF/cl_algorithms/ewc.py:16-89), device-resident.

The reference keeps theta* and the Fisher diagonal on the CPU and re-uploads 0.89 GB of them on EVERY training step
(ewc.py:84-85), then runs 206 x (sub, pow, mul, sum) tiny kernels.  Here both live in HBM as flat fp32 vectors laid out
exactly like the encoder range of the parameter buffer, so the penalty and its gradient are ONE streaming kernel
(20 B per encoder parameter), and the Fisher estimate never leaves the device.  `fisher_dict` / `param_dict` keep the
reference's shape (task -> {encoder-relative name `vilt.*` -> tensor}) as views of those vectors.

Data parallel: theta* and F are replicated per rank.  The Fisher estimate squares gradients that ACCUMULATE across
batches in order (:56-64), so its value depends on the batch sequence and sharding it would change the result: every rank runs the pass on
the WHOLE global batches (sharded loader switched to `replicated()`, gradient reducer suspended) -- the single-process arithmetic -- and
rank 0's F is then broadcast so that replicas stay bit-identical even where a kernel's summation order is not (split-K atomics).  The task
draw of `compute_ewc_loss` uses Python's `random`, seeded identically on every rank; the penalty gradient is added after the all-reduce.

r06, opt-in (`args.ewc_fisher_sharded` / CLIMB_AMD_FISHER_SHARDED=1): the SHARDED Fisher pass.  The replicated pass costs every rank the whole pass (N x
redundant).  What makes the estimate order-dependent is only the running sum G_k = sum_{j<=k} g_j that batch k squares -- g_j itself is a batch-mean
gradient and shards like any training step's: rank r computes its weighted share of g_k (the sharded loader's `dp_weight`), ONE fp32 all-reduce over the
encoder range makes g_k whole on every rank, every rank adds it to its own copy of G and squares that.  Same estimate up to the summation order of the
shares (tests compare the two passes); 1 / N of the forward / backward work per rank and one 0.36 GB collective per Fisher batch."""
from __future__ import annotations

import argparse
import contextlib
import logging
import random
from typing import Dict

import torch

from .. import _lib, parallel
from ..data.sharding import replicated

logger = logging.getLogger(__name__)


def _num_examples(raw_texts) -> int:
    """Examples in a collated batch: a list of strings in the reference's loaders (REF ewc.py:65 uses len()), or pre-tokenised
    tensors (dict of [B, T]) when the input pipeline already ran."""
    if isinstance(raw_texts, dict):
        return int(next(iter(raw_texts.values())).shape[0])
    return len(raw_texts)


def _views(flat: torch.Tensor, eng) -> Dict[str, torch.Tensor]:
    lay = eng.layout
    out = {}
    for n in lay.shapes:
        if n.startswith("vilt_encoder."):
            o = lay.offset[n]
            out[n[len("vilt_encoder."):]] = flat[o:o + lay.numel(n)].view(lay.shapes[n])
    return out


class _EwcLossFn(torch.autograd.Function):
    """`compute_ewc_loss` result that participates in autograd, for reference-style trainers that do
    `(loss + ewc_loss).backward()` (REF train_vqa.py:159-162)."""

    @staticmethod
    def forward(ctx, sentinel, ewc, model, task_key):
        eng = model._host.engine()
        ctx.args = (ewc, model, task_key)
        return eng.ewc_penalty(ewc.param_flat[task_key], ewc.fisher_flat[task_key], ewc.ewc_loss_weight, add_grad=False)

    @staticmethod
    def backward(ctx, gout):
        ewc, model, task_key = ctx.args
        host = model._host
        eng = host.engine()
        host.before_backward()
        eng.ewc_penalty(ewc.param_flat[task_key], ewc.fisher_flat[task_key], ewc.ewc_loss_weight, add_grad=True, gscale=float(gout))
        eng.touched.append((0, eng.layout.encoder_end))
        host.after_backward()
        return None, None, None, None


class EWC:
    def __init__(self, args: argparse.Namespace):
        self.fisher_sample_percentage = args.ewc_fisher_sample_percentage
        self.ewc_loss_weight = args.ewc_loss_weight
        import os
        self.fisher_sharded = bool(getattr(args, "ewc_fisher_sharded", False)) or os.environ.get("CLIMB_AMD_FISHER_SHARDED", "0") == "1"
        self.fisher_dict = {}
        self.param_dict = {}
        self.fisher_flat: Dict[str, torch.Tensor] = {}
        self.param_flat: Dict[str, torch.Tensor] = {}
        self.task_keys = []

    def save_task_parameters(self, task_key: str, model, task_trainer, device: torch.device):
        """REF ewc.py:28-73, including its quirk: `train_step` is called WITHOUT an optimizer, so gradients keep
        accumulating across batches and batch k contributes (sum_{j<=k} g_j)^2."""
        model.to(device) if next(model.parameters()).device != torch.device(device) else None
        host = model._host
        eng = host.engine()
        n = eng.layout.encoder_end
        self.param_flat[task_key] = eng.flat[:n].clone()
        fisher = torch.zeros(n, dtype=torch.float32, device=eng.device)
        assert task_key not in self.task_keys
        self.task_keys.append(task_key)
        dataloader = task_trainer.get_train_dataloader()
        fisher_sample_size = int(self.fisher_sample_percentage * len(dataloader.dataset))
        self.device = task_trainer.device
        optimizer = model.create_optimizer(task_trainer.hparams)
        optimizer.zero_grad()
        num_samples_completed = 0
        world = parallel.rank_world()[1]
        reducer = host.ddp if world > 1 else None
        if world > 1 and getattr(self, "fisher_sharded", False):
            num_samples_completed = self._sharded_fisher_pass(model, task_trainer, dataloader, fisher, n, fisher_sample_size, reducer)
        else:
            with replicated(dataloader), (reducer.suspended() if reducer is not None else contextlib.nullcontext()):
                for step, batch in enumerate(dataloader):
                    task_trainer.train_step(model, batch)
                    eng = host.engine()
                    eng.fisher_accumulate(fisher)
                    num_samples_completed += _num_examples(batch["raw_texts"])
                    if num_samples_completed >= fisher_sample_size:
                        break
        _lib.call("climb_scale", fisher, n, 1.0 / max(1, num_samples_completed), torch.cuda.current_stream().cuda_stream)
        if world > 1:
            import torch.distributed as dist
            pg = getattr(host.ddp, "pg", None)          # the reducer's group, not WORLD: a job may run several replicas' groups side by side
            dist.broadcast(fisher, src=dist.get_global_rank(pg, 0) if pg is not None else 0, group=pg)
        self.fisher_flat[task_key] = fisher
        self.fisher_dict[task_key] = _views(fisher, eng)
        self.param_dict[task_key] = _views(self.param_flat[task_key], eng)
        logger.info("Saved encoder parameters for {} task!".format(task_key))

    def _sharded_fisher_pass(self, model, task_trainer, dataloader, fisher: torch.Tensor, n: int, fisher_sample_size: int, reducer) -> int:
        """The Fisher pass with every batch's gradient computed on the rank's SHARD (module docstring, r06).  Returns the examples consumed (global)."""
        import torch.distributed as dist
        host = model._host
        pg = getattr(reducer, "pg", None)
        world = dist.get_world_size(pg)
        st = lambda: torch.cuda.current_stream().cuda_stream          # noqa: E731
        gsum = torch.zeros(n, dtype=torch.float32, device=fisher.device)          # G_k: the accumulating sum REF ewc.py:56-64 leaves in .grad
        done = 0
        with (reducer.suspended() if reducer is not None else contextlib.nullcontext()):
            for step, batch in enumerate(dataloader):
                eng = host.engine()
                eng.zero_grad()                                   # this batch's share alone: the accumulation lives in gsum
                task_trainer.train_step(model, batch)             # d(loss) weighted by the shard's share of the global batch (dp_weight)
                eng = host.engine()
                eng.materialize_dw()
                g = eng.grad[:n]
                dist.all_reduce(g, op=dist.ReduceOp.SUM, group=pg)
                _lib.call("climb_scale_f32", g, n, 1.0 / world, st())          # mean over ranks of the weighted shares = the global batch's gradient
                _lib.call("climb_elementwise", 5, gsum, g, gsum, n, 1.0, st())
                _lib.call("climb_fisher_accum", fisher, gsum, n, st())
                w = float(batch.get("dp_weight", 1.0)) if isinstance(batch, dict) else 1.0
                cnt = torch.tensor([float(_num_examples(batch["raw_texts"])) if w > 0 else 0.0], dtype=torch.float64, device=fisher.device)
                dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=pg)
                done += int(round(float(cnt.item())))
                if done >= fisher_sample_size:
                    break
            host.engine().zero_grad()
        return done

    def set_task_state(self, task_key: str, model, fisher_named: Dict[str, torch.Tensor], param_named: Dict[str, torch.Tensor]):
        """Install an externally computed Fisher / theta* (dicts keyed like the reference's: `vilt.*`)."""
        eng = model._host.engine()
        n = eng.layout.encoder_end
        f = torch.zeros(n, dtype=torch.float32, device=eng.device)
        s = eng.flat[:n].clone()
        fv, sv = _views(f, eng), _views(s, eng)
        for k in fv:
            fv[k].copy_(fisher_named[k])
            sv[k].copy_(param_named[k])
        self.fisher_flat[task_key], self.param_flat[task_key] = f, s
        self.fisher_dict[task_key], self.param_dict[task_key] = fv, sv
        if task_key not in self.task_keys:
            self.task_keys.append(task_key)

    def compute_ewc_loss(self, model):
        """REF ewc.py:75-87: one previous task sampled with Python's `random`; returns (task, lambda * sum F (theta-theta*)^2)."""
        ewc_task_key = random.choice(self.task_keys)
        sentinel = next(p for p in model.get_encoder().parameters())
        if torch.is_grad_enabled() and any(p.requires_grad for p in model.get_encoder().parameters()):
            return ewc_task_key, _EwcLossFn.apply(sentinel, self, model, ewc_task_key)
        eng = model._host.engine()
        return ewc_task_key, eng.ewc_penalty(self.param_flat[ewc_task_key], self.fisher_flat[ewc_task_key], self.ewc_loss_weight, add_grad=False)

    def add_penalty_gradient(self, model):
        """Fused-step form: penalty value AND its gradient 2*lambda*F*(theta-theta*) added into the flat grad buffer in one
        pass.  Under data parallelism this runs AFTER the gradient all-reduce (the term is identical on every rank)."""
        ewc_task_key = random.choice(self.task_keys)
        eng = model._host.engine()
        loss = eng.ewc_penalty(self.param_flat[ewc_task_key], self.fisher_flat[ewc_task_key], self.ewc_loss_weight, add_grad=True)
        eng.touched.append((0, eng.layout.encoder_end))
        return ewc_task_key, loss

    def park_penalty(self, model):
        """r05: the fused step's form when the caller named its optimizer (`fused_forward_backward(..., optimizer=opt)`): nothing is launched here --
        FusedAdamW.step() adds the term inside its own passes and fills the returned tensor with the penalty's value (engine.park_ewc).  Same task
        draw as `add_penalty_gradient` (one call of Python's `random`, REF ewc.py:77)."""
        ewc_task_key = random.choice(self.task_keys)
        eng = model._host.engine()
        loss = eng.park_ewc(self.param_flat[ewc_task_key], self.fisher_flat[ewc_task_key], self.ewc_loss_weight)
        eng.touched.append((0, eng.layout.encoder_end))
        return ewc_task_key, loss

    def do_ewc(self):
        return True if len(self.task_keys) > 0 else False
