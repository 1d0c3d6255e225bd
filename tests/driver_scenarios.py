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
