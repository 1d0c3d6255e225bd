"""REF/train/train_upstream_continual_learning.py:108-327 (`main()` after argument parsing), restated call for call against this
package, for the GPU box where the reference does not exist.  tests/golden/driver_calls.json holds the calls the reference driver
ITSELF makes into the package (recorded through integration/climb_shim by oracle/record_driver_calls.py); the GPU test runs this
function under the same recorder and requires the same sequence."""
import json
import logging
import os

import torch

logger = logging.getLogger(__name__)


def run_upstream(args, device, after_model_created=None):
    # imports happen here, after tests/driver_trace.install(): several names are bound by `from x import f`, as in the driver
    from climb_amd.cl_algorithms import AdapterHandler, EWC, ExperienceReplayMemory
    from climb_amd.cl_evaluation.evaluate_cl_algorithm import catastrophic_forgetting_eval, upstream_knowledge_transfer_eval
    from climb_amd.configs.model_configs import model_configs
    from climb_amd.configs.task_configs import SUPPORTED_VL_TASKS, task_configs
    from climb_amd.modeling import create_continual_learner_map
    from climb_amd.utils import set_seed

    experiment_name = "{}-{}".format(args.encoder_name, args.cl_algorithm)                                   # :110-118
    if args.cl_algorithm == "adapter":
        experiment_name = "{}_{}_{}config".format(experiment_name, args.adapter_method, args.adapter_config)
    elif args.cl_algorithm == "freeze_bottom_k_layers":
        experiment_name = experiment_name.replace("_k_layers", "{}layers".format(args.layers_to_freeze))
    for i, task_key in enumerate(args.ordered_cl_tasks):
        experiment_name = "{}-task{}_{}".format(experiment_name, i, task_key)
    output_dir = os.path.join(args.output_dir, experiment_name)
    results_file = os.path.join(output_dir, "results.json")
    os.makedirs(output_dir, exist_ok=True)
    set_seed(args)                                                                                          # :122
    for task_key in args.ordered_cl_tasks:                                                                  # :141-142
        assert task_key in SUPPORTED_VL_TASKS
    model_config = model_configs[args.encoder_name]                                                         # :145-151
    model = create_continual_learner_map[args.encoder_name](model_name_or_path=args.pretrained_model_name, ordered_cl_tasks=args.ordered_cl_tasks,
                                                            model_config=model_config, task_configs=task_configs, device=device)
    args.visual_input_type = model_config["visual_input_type"]
    if after_model_created is not None:
        after_model_created(model)
    replay_memory = ewc = adapter_handler = None                                                            # :157-179
    if args.cl_algorithm == "experience_replay":
        replay_memory = ExperienceReplayMemory()
    elif args.cl_algorithm == "adapter":
        adapter_handler = AdapterHandler(adapter_method=args.adapter_method, args=args)
        adapter_handler.add_adapters_to_model(model)
    elif args.cl_algorithm == "ewc":
        ewc = EWC(args)
    elif args.cl_algorithm == "freeze_encoder":
        model.get_encoder().freeze_all_weights()
    elif args.cl_algorithm == "freeze_bottom_k_layers":
        model.get_encoder().freeze_bottom_k_layers(k=args.layers_to_freeze)
    total_params = sum(p.numel() for p in model.parameters())                                               # :185-188
    trainable_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    out = {"total_params": total_params, "trainable_params": trainable_params, "output_dir": output_dir}
    task_trainers = {}
    if args.do_train:
        results = json.load(open(results_file)) if os.path.exists(results_file) else []                     # :201-209
        for task_num, task_key in enumerate(args.ordered_cl_tasks):
            task_output_dir = os.path.join(output_dir, "checkpoints", "task{}_{}".format(task_num, task_key))
            if os.path.exists(os.path.join(task_output_dir, "model")):                                      # :222-240 resume branch
                model.load_state_dict(torch.load(os.path.join(task_output_dir, "model")))
                task_trainer = task_configs[task_key]["task_trainer"](args, task_configs, model_config, device)
            else:
                if args.cl_algorithm == "adapter":                                                          # :245-249
                    adapter_handler.activate_adapter_for_training(task_key=task_key, model=model)
                    trainable_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
                task_trainer = task_configs[task_key]["task_trainer"](args, task_configs, model_config, device)      # :253-254
                best_eval_score, best_model = task_trainer.train(model, replay_memory=replay_memory, ewc=ewc)
                os.makedirs(task_output_dir, exist_ok=True)                                                 # :261-267
                best_task_model = best_model["model"]
                torch.save(best_task_model.state_dict(), os.path.join(task_output_dir, "model"))
                torch.save(best_task_model.get_encoder().state_dict(), os.path.join(task_output_dir, "encoder"))
                results.append({"task_num": task_num, "task_key": task_key, "best_score": best_eval_score, "best_epoch": best_model["epoch"]})
                json.dump(results, open(results_file, "w"))                                                 # :270-278
            task_trainers[task_key] = task_trainer
            if args.cl_algorithm == "experience_replay":                                                    # :281-294
                replay_memory.add_task_memory_buffer(args=args, task_key=task_key, task_config=task_configs[task_key], task_trainer=task_trainer,
                                                     memory_percentage=args.memory_percentage, sampling_strategy=args.memory_sampling_strategy)
            elif args.cl_algorithm == "ewc" and task_num < len(args.ordered_cl_tasks) - 1:
                ewc.save_task_parameters(task_key=task_key, model=model, task_trainer=task_trainer, device=device)
        out["results"] = results
    if args.do_eval:
        upstream_knowledge_dict = upstream_knowledge_transfer_eval(args, results_file)                      # :302-304
        if not args.do_train:
            for task_num, task_key in enumerate(args.ordered_cl_tasks):                                     # :310-315
                task_trainers[task_key] = task_configs[task_key]["task_trainer"](args, task_configs, model_config, device)
        catastrophic_forgetting_dict = catastrophic_forgetting_eval(args, results_file, model, task_trainers, adapter_handler)     # :322
        eval_results = {"upstream_knowledge_transfer": upstream_knowledge_dict, "forgetting": catastrophic_forgetting_dict}
        json.dump(eval_results, open(os.path.join(output_dir, "eval_results.json"), "w"))
        out["eval_results"] = eval_results
    out["model"], out["ewc"], out["replay_memory"] = model, ewc, replay_memory       # for the tests' own assertions
    return out
