"""REF/train/train_lowshot_multimodal.py:39-53 (`train_low_shot`) and :108-243 (`main()` after argument parsing), restated call for call
against this package, for the GPU box where the reference does not exist (compare tests/upstream_driver.py).
tests/golden/driver_calls.json["lowshot_scenarios"] holds the calls the reference driver ITSELF makes (oracle/record_driver_calls.py)."""
import copy
import json
import logging
import os

import torch

logger = logging.getLogger(__name__)


def train_low_shot(args, task_configs, low_shot_model, low_shot_task_key, model_config, device):                 # :39-55
    low_shot_config = task_configs[low_shot_task_key]["low_shot_config"]
    task_trainer_class = low_shot_config["task_trainer"]
    task_trainer = task_trainer_class(args, task_configs, model_config, device, low_shot_config=low_shot_config)
    best_eval_score, best_model = task_trainer.train(low_shot_model)
    return best_eval_score, low_shot_config


def run_lowshot(args, device, after_model_created=None):
    from climb_amd.configs.model_configs import model_configs
    from climb_amd.configs.task_configs import SUPPORTED_VL_TASKS, task_configs
    from climb_amd.modeling import create_continual_learner_map
    from climb_amd.utils import set_seed

    experiment_name = "{}-{}".format(args.encoder_name, args.cl_algorithm)                                       # :110-121
    if args.cl_algorithm == "adapter":
        experiment_name = "{}_{}".format(experiment_name, args.adapter_config)
    elif args.cl_algorithm == "freeze_bottom_k_layers":
        experiment_name = experiment_name.replace("_k_layers", "{}layers".format(args.layers_to_freeze))
    for i, task_key in enumerate(args.ordered_cl_tasks):
        experiment_name = "{}-task{}_{}".format(experiment_name, i, task_key)
    output_dir = os.path.join(args.output_dir, experiment_name)
    results_file = os.path.join(output_dir, "lowshot_results.json")
    os.makedirs(output_dir, exist_ok=True)
    set_seed(args)                                                                                               # :123
    for task_key in args.ordered_cl_tasks:                                                                       # :126-127
        assert task_key in SUPPORTED_VL_TASKS
    model_config = model_configs[args.encoder_name]                                                              # :130-137
    model = create_continual_learner_map[args.encoder_name](model_name_or_path=args.pretrained_model_name, ordered_cl_tasks=args.ordered_cl_tasks,
                                                            model_config=model_config, task_configs=task_configs, device=device)
    args.visual_input_type = model_config["visual_input_type"]
    if after_model_created is not None:
        after_model_created(model)
    total_params = sum(p.numel() for p in model.parameters())                                                    # :146-149
    trainable_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    results = json.load(open(results_file)) if os.path.exists(results_file) else []                              # :153-162
    if args.cl_algorithm == "singletask_ft":                                                                     # :168-186
        task_key = args.ordered_cl_tasks[0]
        low_shot_model = copy.deepcopy(model)
        low_shot_eval_score, low_shot_config = train_low_shot(args, task_configs, low_shot_model, task_key, model_config, device)
        config_copy = copy.deepcopy(low_shot_config)
        config_copy.pop("task_trainer", None)
        results.append({"task_key": task_key, "best_low_shot_score": low_shot_eval_score, "low_shot_config": config_copy})
        json.dump(results, open(results_file, "w"))
    else:
        for task_num, task_key in enumerate(args.ordered_cl_tasks):                                              # :190-241
            task_output_dir = os.path.join(output_dir, "checkpoints", "task{}_{}".format(task_num, task_key))
            assert os.path.exists(os.path.join(task_output_dir, "model"))
            model.load_state_dict(torch.load(os.path.join(task_output_dir, "model")))
            for low_shot_task_key in args.ordered_cl_tasks[task_num + 1:]:
                low_shot_task_num = args.ordered_cl_tasks.index(low_shot_task_key)
                low_shot_model = copy.deepcopy(model)
                low_shot_eval_score, low_shot_config = train_low_shot(args, task_configs, low_shot_model, low_shot_task_key, model_config, device)
                config_copy = copy.deepcopy(low_shot_config)
                config_copy.pop("task_trainer", None)
                results.append({"upstream_task_num": task_num, "upstream_task_key": task_key, "lowshot_task_num": low_shot_task_num,
                                "lowshot_task_key": low_shot_task_key, "best_low_shot_score": low_shot_eval_score, "low_shot_config": config_copy})
                json.dump(results, open(results_file, "w"))
    return {"results": results, "model": model, "output_dir": output_dir, "total_params": total_params, "trainable_params": trainable_params}
