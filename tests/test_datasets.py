"""Row F1: the four VL datasets, their collate functions and loader conventions against what the REFERENCE's own dataset classes
produce on the same synthetic data tree (tests/golden/datasets.json, written by oracle/gen_golden_datasets.py), plus the trainer
constructor the upstream driver calls.  CPU only."""
import json
import os
import shutil
import types

import pytest
import torch

from tests import synth_data as sd


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("climb_data"))
    sd.make_climb_data_tree(root, n_train=8, n_val=4, seed=0)
    shutil.copytree(os.path.join(root, "vqav2"), os.path.join(root, "vqav2-tok"))
    return root


def _dump(batch):
    out = {}
    for k, v in batch.items():
        if k == "images":
            out[k] = [[list(i.size) for i in im] if isinstance(im, list) else list(im.size) for im in v]
        elif k == "target_scores":
            out[k] = [[int(r), int(c), round(float(v[r, c]), 6)] for r, c in v.nonzero().tolist()]
            out["target_scores_shape"] = list(v.shape)
        elif hasattr(v, "tolist"):
            out[k] = v.tolist()
        else:
            out[k] = v
    return out


def test_datasets_equal_the_reference_on_the_synthetic_tree(tree, golden_dir):
    import transformers
    from climb_amd.data import datasets as D
    g = json.load(open(os.path.join(golden_dir, "datasets.json")))
    args = types.SimpleNamespace(batch_size=g["batch_size"], num_workers=0, visual_input_type="pil-image")
    tok = sd.make_tokenizer(sd.write_vocab(os.path.join(tree, "vocab.txt")))
    coco = D.MSCOCOImagesDataset(os.path.join(tree, "ms-coco"), "pil-image")
    flickr = D.Flickr30KImagesDataset(os.path.join(tree, "flickr30k"), "pil-image")
    loaders = {
        "vqa/val": D.build_vqa_dataloader(args, os.path.join(tree, "vqav2"), coco, "val", "pil-image"),
        "vqa/val/tokenized": D.build_vqa_dataloader(args, os.path.join(tree, "vqav2-tok"), coco, "val", "pil-image", tokenizer=tok),
        "nlvr2/val": D.build_nlvr2_dataloader(args, os.path.join(tree, "nlvr2"), "val", "pil-image"),
        "snli-ve/dev": D.build_snli_ve_dataloader(args, os.path.join(tree, "snli-ve"), flickr, "dev", "pil-image"),
        "vcr/val": D.build_vcr_dataloader(args, os.path.join(tree, "vcr") + "/", "val", "qa", "pil-image"),
    }
    assert set(loaders) == set(g["loaders"])
    for name, dl in loaders.items():
        want = g["loaders"][name]
        assert len(dl.dataset) == want["n_examples"] and len(dl) == want["n_batches"], name
        got = [_dump(b) for b in dl]
        assert json.loads(json.dumps(got)) == want["batches"], name          # texts, labels, soft targets, token ids, image sizes, keys
    for pl in g["process_list"]:
        assert D.process_list(pl["text"], pl["objects"]) == pl["out"]
    # loader conventions of the reference: NLVR2 halves and VCR quarters the batch size; training splits shuffle
    assert loaders["nlvr2/val"].batch_size == 2 and loaders["vcr/val"].batch_size == 1 and loaders["vqa/val"].batch_size == 4
    tr = D.build_vqa_dataloader(args, os.path.join(tree, "vqav2"), coco, "train", "pil-image")
    assert len(tr.dataset) == g["train_sizes"]["vqa"] and isinstance(tr.sampler, torch.utils.data.RandomSampler)
    # the parse caches are the reference's own pickle files: a second construction reads them
    assert os.path.exists(os.path.join(tree, "vqav2", "cached_vqa_data", "vqa_val.pkl"))
    again = D.VQADataset(os.path.join(tree, "vqav2"), coco, "val")
    assert again.data == loaders["vqa/val"].dataset.data


def test_pre_shrink_size_rule_and_vqa_soft_scores():
    from climb_amd.data import datasets as D
    assert D.resized_output_size(500, 400) == (480, 384)            # shorter edge -> 384
    assert D.resized_output_size(400, 1000) == (256, 640)           # ... unless the longer edge would pass 640
    assert D.resized_output_size(640, 480) == (512, 384)
    assert [D.get_score(k) for k in range(6)] == [0.0, 0.3, 0.6, 0.9, 1.0, 1.0]
    t = D.target_tensor(10, [3, 7], [0.3, 1.0])
    assert t[3] == pytest.approx(0.3) and t[7] == 1.0 and float(t.sum()) == pytest.approx(1.3)
    raw = D.image_collate([torch.zeros(3, 4, 5), torch.ones(3, 4, 5)], "raw")
    assert raw.shape == (2, 3, 4, 5)
    feats = D.image_collate([torch.ones(2, 8), torch.ones(5, 8)], "fast-rcnn")
    assert feats.shape == (2, 5, 8) and float(feats[0, 2:].abs().sum()) == 0.0


def test_trainers_are_constructible_the_way_the_upstream_driver_constructs_them(tree):
    """REF/train/train_upstream_continual_learning.py:254: `task_trainer_class(args, task_configs, model_config, device)`."""
    from climb_amd.configs.model_configs import model_configs
    from climb_amd.configs.task_configs import task_configs
    args = types.SimpleNamespace(batch_size=4, num_workers=0, visual_input_type="pil-image", climb_data_dir=tree, cl_algorithm="sequential_ft")
    sizes = {"vqa": (2, 1), "nlvr2": (4, 2), "snli-ve": (2, 1), "vcr": (8, 4)}        # batches per epoch: train, validation
    for task, (ntr, nva) in sizes.items():
        trainer = task_configs[task]["task_trainer"](args, task_configs, model_configs["vilt"], torch.device("cpu"))
        assert len(trainer.get_train_dataloader()) == ntr and len(trainer.val_dataloader) == nva
        assert trainer.max_steps == ntr * task_configs[task]["num_epochs"]
        batch = next(iter(trainer.val_dataloader))
        inputs = trainer.batch2inputs_converter(batch)
        assert set(inputs) == {"images", "texts"} and len(inputs["texts"]) == len(batch["raw_texts"])
        assert callable(trainer.get_collate_fn())
    with pytest.raises(RuntimeError):
        task_configs["vqa"]["task_trainer"](types.SimpleNamespace(batch_size=4), task_configs, model_configs["vilt"], torch.device("cpu"))


def test_low_shot_trainers_subsample_the_way_the_low_shot_driver_asks(tree):
    """REF/train/train_lowshot_multimodal.py:52: `trainer_class(args, task_configs, model_config, device, low_shot_config=...)`; the
    reference's LowShot*Trainer constructors (train_vqa.py:286-308 and siblings): percentage of the examples / N shots per class,
    evaluation epochs 1-based in the config and 0-based in the trainer, max_steps from the SHRUNK loader."""
    import copy
    import random
    from climb_amd.configs.model_configs import model_configs
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.train import task_trainer as tt
    args = types.SimpleNamespace(batch_size=4, num_workers=0, visual_input_type="pil-image", climb_data_dir=tree, cl_algorithm="sequential_ft")
    ref = {"vqa": ("percentage", 0.05, [6, 8, 10]), "nlvr2": ("n-shot-per-class", 2048, [6, 8, 10]), "snli-ve": ("n-shot-per-class", 2048, [2, 4, 5]),
           "vcr": ("percentage", 0.05, [2, 4, 6, 8, 10])}                                   # REF/configs/task_configs.py:31-34, 51-55, 73-77, 96-100
    for task, (kind, size, epochs) in ref.items():
        cfg = task_configs[task]["low_shot_config"]
        assert cfg["type"] == kind and cfg["eval_epochs"] == epochs and cfg.get("percentage", cfg.get("num_shots_per_class")) == size
        assert issubclass(cfg["task_trainer"], tt.LowShotMixin) and issubclass(cfg["task_trainer"], task_configs[task]["task_trainer"])
    random.seed(0)
    small = {"vqa": dict(percentage=0.5), "nlvr2": dict(num_shots_per_class=2), "snli-ve": dict(num_shots_per_class=1), "vcr": dict(percentage=0.5)}
    for task, over in small.items():
        cfg = copy.copy(task_configs[task]["low_shot_config"])
        cfg.update(over)
        full = task_configs[task]["task_trainer"](args, task_configs, model_configs["vilt"], torch.device("cpu"))
        n_full = len(full.get_train_dataloader().dataset)
        t = cfg["task_trainer"](args, task_configs, model_configs["vilt"], torch.device("cpu"), low_shot_config=cfg)
        n = len(t.get_train_dataloader().dataset)
        want = int(over["percentage"] * n_full) if "percentage" in over else over["num_shots_per_class"] * task_configs[task]["num_labels"]
        assert n == want < n_full, (task, n, want, n_full)
        assert t.eval_epochs == [e - 1 for e in cfg["eval_epochs"]] and t.max_steps == len(t.get_train_dataloader()) * task_configs[task]["num_epochs"]
        assert len(t.val_dataloader.dataset) == len(full.val_dataloader.dataset)              # validation untouched
    with pytest.raises(ValueError):
        task_configs["vqa"]["low_shot_config"]["task_trainer"](args, task_configs, model_configs["vilt"], torch.device("cpu"))


def test_recorded_calls_bind_to_this_packages_signatures(golden_dir):
    """CPU-checkable half (also run on the GPU box): every recorded call's positional count and keyword set binds to the callable of
    the same name in this package."""
    import inspect
    import climb_amd.cl_algorithms as cl
    import climb_amd.cl_evaluation.evaluate_cl_algorithm as ev
    import climb_amd.train.task_trainer as tt
    import climb_amd.utils as ut
    from climb_amd.modeling import create_continual_learner_map
    from climb_amd.modeling.vilt import ViltContinualLearner, ViltEncoderWrapper
    table = {"create_continual_learner_map[vilt]": (create_continual_learner_map["vilt"], 0), "TaskTrainer": (tt.VLTaskTrainer.__init__, 1), "EWC": (cl.EWC.__init__, 1),
             "ExperienceReplayMemory": (cl.ExperienceReplayMemory.__init__, 1), "AdapterHandler": (cl.AdapterHandler.__init__, 1),
             "upstream_knowledge_transfer_eval": (ev.upstream_knowledge_transfer_eval, 0), "catastrophic_forgetting_eval": (ev.catastrophic_forgetting_eval, 0),
             "set_seed": (ut.set_seed, 0)}
    classes = {"TaskTrainer": tt.VLTaskTrainer, "EWC": cl.EWC, "ExperienceReplayMemory": cl.ExperienceReplayMemory, "AdapterHandler": cl.AdapterHandler,
               "ViltContinualLearner": ViltContinualLearner, "ViltEncoderWrapper": ViltEncoderWrapper}
    table["LowShotTaskTrainer"] = (tt.LowShotMixin.__init__, 1)
    classes["LowShotTaskTrainer"] = tt.LowShotMixin
    g = json.load(open(os.path.join(golden_dir, "driver_calls.json")))
    seen = set()
    for scen in list(g["scenarios"].values()) + list(g["lowshot_scenarios"].values()):
        for c in scen["calls"]:
            if "." in c["name"]:
                fn, skip = getattr(classes[c["name"].split(".")[0]], c["name"].split(".")[1]), 1
            else:
                fn, skip = table[c["name"]]
            inspect.signature(fn).bind(*([None] * (c["nargs"] + skip)), **{k: None for k in c["kwargs"]})
            seen.add(c["name"])
    assert {"TaskTrainer.train", "EWC.save_task_parameters", "ExperienceReplayMemory.add_task_memory_buffer",
            "AdapterHandler.activate_adapter_for_training", "catastrophic_forgetting_eval", "LowShotTaskTrainer", "LowShotTaskTrainer.train"} <= seen
