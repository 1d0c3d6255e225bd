"""The upstream-driver scenarios recorded from the reference (oracle/record_driver_calls.py) and replayed on the GPU
(tests/test_gpu_driver.py): BASELINE.json configs[2] (adapters, VQA -> NLVR2) and configs[3] / [4] (EWC / ER over the four-task
sequence, here on one GPU), plus the two freezing algorithms."""
import types

FOUR = ["vqa", "nlvr2", "snli-ve", "vcr"]
SCENARIOS = {
    "adapter": dict(cl_algorithm="adapter", ordered_cl_tasks=["vqa", "nlvr2"], adapter_method="vanilla", adapter_config="houlsby", adapter_reduction_factor=16),
    "ewc": dict(cl_algorithm="ewc", ordered_cl_tasks=FOUR, ewc_fisher_sample_percentage=0.5, ewc_loss_weight=100.0),
    "experience_replay": dict(cl_algorithm="experience_replay", ordered_cl_tasks=FOUR, memory_percentage=0.5, memory_sampling_strategy="random",
                              replay_frequency=2),
    "freeze_bottom_k_layers": dict(cl_algorithm="freeze_bottom_k_layers", ordered_cl_tasks=["vqa", "nlvr2"], layers_to_freeze=9),
    "freeze_encoder": dict(cl_algorithm="freeze_encoder", ordered_cl_tasks=["snli-ve", "vcr"]),
}
SINGLETASK_SCORES = {"vqa": 65.0, "nlvr2": 72.0, "snli-ve": 74.0, "vcr": 58.0}
N_TRAIN, N_VAL, SEED, EASY_ANSWER = 8, 5, 3, 7       # 5 validation examples: accuracies are multiples of 20 and never equal a random baseline


def defaults():
    return dict(encoder_name="vilt", pretrained_model_name="random-init:5", do_train=True, do_eval=True, memory_percentage=0.0,
                memory_sampling_strategy=None, replay_frequency=None, adapter_method=None, adapter_config=None, adapter_reduction_factor=0,
                ewc_fisher_sample_percentage=0.0, ewc_loss_weight=0.0, layers_to_freeze=0, do_wandb_logging=False, batch_size=4, num_workers=0, seed=42)


def namespace(name, climb_data_dir, output_dir):
    d = defaults()
    d.update(SCENARIOS[name])
    d["ordered_cl_tasks"] = list(d["ordered_cl_tasks"])
    return types.SimpleNamespace(climb_data_dir=climb_data_dir, output_dir=output_dir, **d)


def argv(name, climb_data_dir, output_dir):
    d = defaults()
    d.update(SCENARIOS[name])
    out = []
    for k, v in d.items():
        if isinstance(v, bool):
            if v:
                out.append(f"--{k}")
        elif v is not None:
            out += [f"--{k}", ",".join(v) if isinstance(v, list) else str(v)]
    return out + ["--climb_data_dir", climb_data_dir, "--output_dir", output_dir]


def write_singletask_results(output_dir, tasks):
    import json
    import os
    for t in tasks:
        d = os.path.join(output_dir, "vilt-singletask_ft-task0_{}".format(t))
        os.makedirs(d, exist_ok=True)
        json.dump([{"task_num": 0, "task_key": t, "best_score": SINGLETASK_SCORES[t], "best_epoch": 1}], open(os.path.join(d, "results.json"), "w"))


# ---- the low-shot transfer driver (REF/train/train_lowshot_multimodal.py).  "after": it loads the checkpoints the
# upstream scenario `upstream` wrote into the same output directory and trains every LATER task low-shot from each of them.
LOWSHOT_SCENARIOS = {
    "lowshot_singletask": dict(cl_algorithm="singletask_ft", ordered_cl_tasks=["vqa"], upstream=None),
    "lowshot_after_freeze_bottom_k_layers": dict(cl_algorithm="freeze_bottom_k_layers", ordered_cl_tasks=["vqa", "nlvr2"], layers_to_freeze=9,
                                                 upstream="freeze_bottom_k_layers"),
}
# the reference's low-shot sizes (5 % of VQA, 2048 shots per class) do not exist in an 8-example tree: same code path, small numbers
LOWSHOT_OVERRIDES = {"vqa": dict(percentage=0.5, eval_epochs=[2, 3]), "nlvr2": dict(num_shots_per_class=2, eval_epochs=[1, 3]),
                     "snli-ve": dict(num_shots_per_class=2, eval_epochs=[2]), "vcr": dict(percentage=0.5, eval_epochs=[2])}


def apply_lowshot_overrides(task_configs):
    for k, o in LOWSHOT_OVERRIDES.items():
        task_configs[k]["low_shot_config"].update(o)


def lowshot_defaults():
    return dict(encoder_name="vilt", pretrained_model_name="random-init:5", memory_percentage=0.0, memory_sampling_strategy=None, replay_frequency=None,
                adapter_config=None, adapter_reduction_factor=0, ewc_fisher_sample_percentage=0.0, ewc_loss_weight=0.0, layers_to_freeze=0,
                batch_size=4, num_workers=0, seed=42)


def lowshot_namespace(name, climb_data_dir, output_dir):
    d = lowshot_defaults()
    d.update({k: v for k, v in LOWSHOT_SCENARIOS[name].items() if k != "upstream"})
    d["ordered_cl_tasks"] = list(d["ordered_cl_tasks"])
    return types.SimpleNamespace(climb_data_dir=climb_data_dir, output_dir=output_dir, **d)


def lowshot_argv(name, climb_data_dir, output_dir):
    d = lowshot_defaults()
    d.update({k: v for k, v in LOWSHOT_SCENARIOS[name].items() if k != "upstream"})
    out = []
    for k, v in d.items():
        if v is not None:
            out += [f"--{k}", ",".join(v) if isinstance(v, list) else str(v)]
    return out + ["--climb_data_dir", climb_data_dir, "--output_dir", output_dir]
