"""Continual-learning bookkeeping around the hot path: the two CL metrics of
REF/cl_evaluation/evaluate_cl_algorithm.py and the checkpoint / results.json conventions of the upstream driver
(REF/train/train_upstream_continual_learning.py:215-277), so that a run of the reference can be evaluated by this build and
vice versa.  Host logic only: the evaluation passes themselves go through the task trainers (`eval_forgetting`)."""
from __future__ import annotations

import json
import logging
import os
from collections import defaultdict
from typing import Dict, List, Optional

import torch

from ..configs.model_configs import model_configs
from ..configs.task_configs import task_configs

logger = logging.getLogger(__name__)


def _sync_ranks():
    """Data-parallel runs: every rank writes results.json with identical content (the scores are all-reduced); wait until all of them have,
    before anyone reads it back.  No-op in a single process."""
    from ..parallel import barrier
    barrier()


def _relative_percent(gain: float, span: float) -> float:
    return 100.0 * gain / span


def upstream_knowledge_transfer_eval(args, results_file: str) -> Dict:
    """REF/cl_evaluation/evaluate_cl_algorithm.py:32-72.  Relative gain of the CL score of every task over direct fine-tuning
    of the pretrained encoder on that task alone, in percent of the single-task margin over the random baseline.
    Reads `<output_dir>/<encoder>-singletask_ft-task0_<task>/results.json` for the single-task scores."""
    _sync_ranks()
    with open(results_file) as f:
        cl_results = json.load(f)
    assert len(cl_results) == len(args.ordered_cl_tasks)
    out = {}
    for task_num, task_results in enumerate(cl_results):
        task_key = task_results["task_key"]
        assert task_key == args.ordered_cl_tasks[task_num]
        cl_task_score = task_results["best_score"]
        single_dir = os.path.join(args.output_dir, "{}-singletask_ft-task0_{}".format(args.encoder_name, task_key))
        with open(os.path.join(single_dir, "results.json")) as f:
            single = json.load(f)
        assert len(single) == 1 and single[0]["task_key"] == task_key
        singletask_score = single[0]["best_score"]
        random_score = task_configs[task_key]["random_baseline_score"]
        relative_gain = _relative_percent(cl_task_score - singletask_score, singletask_score - random_score)
        logger.info("Relative Gain for task #%d, %s = %.2f%%", task_num, task_configs[task_key]["task_name"], relative_gain)
        out[task_key] = {"relative_gain": relative_gain, "cl_task_score": cl_task_score, "singletask_score": singletask_score}
    return out


def catastrophic_forgetting_eval(args, results_file: str, model, task_trainers, adapter_handler=None) -> Dict:
    """REF/cl_evaluation/evaluate_cl_algorithm.py:75-140.  For the checkpoint saved after every task i >= 1, re-evaluate each
    earlier task j < i (`trainer.eval_forgetting(model, <ckpt>/model)`) and report the score drop in percent of the margin
    the task had over its random baseline when it was learned."""
    model_config = model_configs[args.encoder_name]
    _sync_ranks()
    with open(results_file) as f:
        cl_results = json.load(f)
    assert len(cl_results) == len(args.ordered_cl_tasks)
    output_dir = os.path.dirname(results_file)
    out = defaultdict(dict)
    for task_num, task_key in enumerate(args.ordered_cl_tasks):
        if task_num < 1:
            continue
        logger.info("Evaluating %s checkpoint after %s on previously-seen tasks %s", model_config["encoder_name"],
                    task_configs[task_key]["task_name"], ",".join(args.ordered_cl_tasks[:task_num]))
        model_path = os.path.join(output_dir, "checkpoints", "task{}_{}".format(task_num, task_key), "model")
        for prev_task_num in range(task_num):
            prev_task_key = args.ordered_cl_tasks[prev_task_num]
            if adapter_handler is not None:
                adapter_handler.activate_adapter_for_eval(prev_task_key, model)
            eval_score = task_trainers[prev_task_key].eval_forgetting(model, model_path)
            prev = cl_results[prev_task_num]
            assert prev["task_key"] == prev_task_key
            baseline_score = prev["best_score"]
            random_score = task_configs[prev_task_key]["random_baseline_score"]
            forgetting = _relative_percent(baseline_score - eval_score, baseline_score - random_score)
            out[task_key][prev_task_key] = {"prev_task": prev_task_key, "current_task": task_key,
                                            "transfer_tasks": "{}->{}".format(task_num, prev_task_num), "forgetting": forgetting,
                                            "absolute_transfer_score": eval_score, "original_prev_task_score": baseline_score}
    return out


# --------------------------------------------------------------------------------------------- driver conventions
def save_task_checkpoint(model, task_output_dir: str):
    """`model` (whole learner) and `encoder` (`get_encoder().state_dict()`, keys `vilt.*`) files, REF train_upstream...:262-267."""
    os.makedirs(task_output_dir, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(task_output_dir, "model"))
    torch.save(model.get_encoder().state_dict(), os.path.join(task_output_dir, "encoder"))


def load_task_checkpoint(model, model_path: str) -> List[str]:
    """Load a `model` file written by either implementation.  Returns the keys of `model` the file did not provide (the
    reference's recovery branch, train_upstream...:226-236: heads of tasks added after the checkpoint was written stay at
    their initial values).  Buffers that only old `transformers` wrote (`...position_ids`) are ignored."""
    sd = torch.load(model_path, map_location="cpu")
    own = model.state_dict()
    extra = [k for k in sd if k not in own and not k.endswith("position_ids")]
    if extra:
        raise KeyError(f"{model_path} holds tensors this model has no place for, e.g. {extra[:3]}")
    missing = [k for k in own if k not in sd]
    with torch.no_grad():
        for k, v in sd.items():
            if k in own:
                own[k].copy_(v)
    if missing:
        logger.info("Uninitialized keys: %s", ",".join(missing))
    return missing


def append_task_result(results_file: str, task_num: int, task_key: str, best_score: float, best_epoch: int) -> List[Dict]:
    """results.json as the driver writes it (train_upstream...:269-277): one record per finished task, in CL order."""
    results = []
    if os.path.exists(results_file):
        with open(results_file) as f:
            results = json.load(f)
    results.append({"task_num": task_num, "task_key": task_key, "best_score": best_score, "best_epoch": best_epoch})
    write_json_once(results_file, results)
    return results


def write_json_once(path: str, obj) -> None:
    """One writer per file under N ranks (identical content everywhere: the scores are all-reduced): rank 0 writes a temporary file and renames it over
    `path`; nobody returns before it is there."""
    from .. import parallel
    from ..parallel import barrier, rank_world
    rank, world = rank_world()
    if rank == 0:
        tmp = f"{path}.tmp.{os.getpid()}"
        with open(tmp, "w") as f:
            (parallel._real_json_dump or json.dump)(obj, f)          # (the driver-facing patch of json.dump is a collective: not from one rank)
        os.replace(tmp, path)
    if world > 1:
        barrier()
