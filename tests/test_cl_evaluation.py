"""SURVEY.md §8(f) row F3 on CPU: the CL metrics against the dictionaries the reference's own evaluate_cl_algorithm produced
(tests/golden/cl_eval.json, written by oracle/gen_golden.py), the results.json / checkpoint conventions of the upstream driver,
and state-dict interchange with a `transformers` ViltModel (both directions, including a 4.x-style `position_ids` buffer)."""
import argparse
import json
import os

import pytest
import torch

from climb_amd.cl_evaluation import (upstream_knowledge_transfer_eval, catastrophic_forgetting_eval, save_task_checkpoint,
                                     load_task_checkpoint, append_task_result)


@pytest.fixture()
def golden():
    with open(os.path.join(os.path.dirname(__file__), "golden", "cl_eval.json")) as f:
        return json.load(f)


def _lay_out_run(tmp_path, g):
    tasks = g["tasks"]
    run_dir = tmp_path / "vilt-sequential_ft-run"
    run_dir.mkdir()
    results_file = str(run_dir / "results.json")
    for r in g["results"]:                                   # written the way the driver appends them
        append_task_result(results_file, r["task_num"], r["task_key"], r["best_score"], r["best_epoch"])
    assert json.load(open(results_file)) == g["results"]
    for t in tasks:
        d = tmp_path / f"vilt-singletask_ft-task0_{t}"
        d.mkdir()
        json.dump([{"task_num": 0, "task_key": t, "best_score": g["singletask_scores"][t], "best_epoch": 5}], open(d / "results.json", "w"))
    args = argparse.Namespace(ordered_cl_tasks=tasks, output_dir=str(tmp_path), encoder_name="vilt")
    return args, results_file


def test_knowledge_transfer_matches_reference(tmp_path, golden):
    args, results_file = _lay_out_run(tmp_path, golden)
    out = upstream_knowledge_transfer_eval(args, results_file)
    assert out == golden["knowledge_transfer"]              # same IEEE doubles, same keys: bit-exact


def test_catastrophic_forgetting_matches_reference(tmp_path, golden):
    args, results_file = _lay_out_run(tmp_path, golden)
    calls = []

    class Trainer:
        def __init__(self, key):
            self.key = key

        def eval_forgetting(self, model, model_path):
            cur = os.path.basename(os.path.dirname(model_path)).split("_", 1)[1]
            assert os.path.basename(model_path) == "model" and "checkpoints" in model_path
            calls.append((cur, self.key))
            return golden["forgetting_scores"][f"{cur}|{self.key}"]

    class Handler:
        def __init__(self):
            self.activated = []

        def activate_adapter_for_eval(self, task_key, model):
            self.activated.append(task_key)

    h = Handler()
    out = catastrophic_forgetting_eval(args, results_file, model=object(), task_trainers={t: Trainer(t) for t in golden["tasks"]}, adapter_handler=h)
    assert {k: dict(v) for k, v in out.items()} == golden["catastrophic_forgetting"]
    assert calls == [("nlvr2", "vqa"), ("snli-ve", "vqa"), ("snli-ve", "nlvr2"), ("vcr", "vqa"), ("vcr", "nlvr2"), ("vcr", "snli-ve")]
    assert h.activated == [k for _, k in calls]              # the earlier task's adapter is switched in before each evaluation


def _tiny_learner(tasks):
    from climb_amd.configs.model_configs import model_configs
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.modeling import create_continual_learner_map
    return create_continual_learner_map["vilt"](model_name_or_path="random-init:3", ordered_cl_tasks=tasks, model_config=model_configs["vilt"],
                                                task_configs=task_configs, device=torch.device("cpu"))


def test_checkpoint_files_and_recovery_branch(tmp_path):
    a = _tiny_learner(["vqa", "nlvr2"])
    ckpt = tmp_path / "checkpoints" / "task0_vqa"
    save_task_checkpoint(a, str(ckpt))
    enc_sd = torch.load(ckpt / "encoder")
    assert all(k.startswith("vilt.") for k in enc_sd) and len(enc_sd) == len(a.get_encoder().state_dict())
    # a file that lacks one task head (written before that head existed): everything else loads, the head is reported
    sd = torch.load(ckpt / "model")
    torch.save({k: v for k, v in sd.items() if not k.startswith("task_layer.nlvr2.")}, ckpt / "model")
    b = _tiny_learner(["vqa", "nlvr2"])
    with torch.no_grad():
        for p in b.parameters():
            p.add_(1.0)
    before = {k: v.clone() for k, v in b.state_dict().items()}
    missing = load_task_checkpoint(b, str(ckpt / "model"))
    assert missing and all(k.startswith("task_layer.nlvr2.") for k in missing)
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert torch.equal(sb[k], before[k] if k in missing else sa[k]), k
    # a file holding a tensor the model has no place for is refused
    bad = dict(sa)
    bad["vilt_encoder.vilt.bogus"] = torch.zeros(1)
    torch.save(bad, tmp_path / "bad")
    with pytest.raises(KeyError):
        load_task_checkpoint(a, str(tmp_path / "bad"))


def test_state_dict_interchange_with_transformers_vilt():
    transformers = pytest.importorskip("transformers")
    torch.manual_seed(0)
    hf = transformers.ViltModel(transformers.ViltConfig())
    ours = _tiny_learner(["vqa"])
    enc = ours.get_encoder()
    # transformers -> ours (what load_vilt_encoder does with a pretrained checkpoint), with the buffer 4.x checkpoints carry
    sd = {"vilt." + k: v for k, v in hf.state_dict().items()}
    sd["vilt.embeddings.text_embeddings.position_ids"] = torch.arange(40)[None]
    assert set(sd) - {"vilt.embeddings.text_embeddings.position_ids"} == set(enc.state_dict())
    enc.load_state_dict(sd)
    for k, v in enc.state_dict().items():
        assert torch.equal(v, sd[k]), k
    # ours -> transformers: strict load, nothing missing, nothing unexpected
    res = hf.load_state_dict({k[len("vilt."):]: v for k, v in enc.state_dict().items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    # the whole learner accepts its own state dict plus the legacy buffer
    full = dict(ours.state_dict())
    full["vilt_encoder.vilt.embeddings.text_embeddings.position_ids"] = torch.arange(40)[None]
    ours.load_state_dict(full)
