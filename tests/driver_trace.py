"""Call recorder for the drop-in boundary.  `install(caller_file)` wraps the functions and methods the reference's
drivers (REF/train/train_upstream_continual_learning.py, REF/train/train_lowshot_multimodal.py) reach in this package; every call whose CALLER is a frame of `caller_file`
is appended to `calls` as {name, nargs, kwargs, returns}.  The same recorder runs (a) under the reference driver itself, through
integration/climb_shim, in the build container (oracle/record_driver_calls.py -> tests/golden/driver_calls.json) and (b) under
tests/upstream_driver.py on the GPU, so the two traces can be compared call by call."""
import functools
import sys

calls = []
_installed = []


def _arity(v):
    if v is None:
        return "none"
    if isinstance(v, tuple):
        return f"tuple{len(v)}"
    if isinstance(v, dict):
        return "dict"
    if isinstance(v, (int, float)):
        return "number"
    return type(v).__name__


def _wrap(fn, name, caller_file, is_init=False):
    @functools.wraps(fn)
    def traced(*a, **k):
        hit = sys._getframe(1).f_code.co_filename == caller_file
        rec = None
        if hit:
            rec = {"name": name, "nargs": len(a) - (1 if (is_init or "." in name) else 0), "kwargs": sorted(k)}
            calls.append(rec)
        out = fn(*a, **k)
        if rec is not None:
            rec["returns"] = "none" if is_init else _arity(out)
        return out
    return traced


def _patch(owner, attr, name, caller_file, is_init=False):
    orig = owner.__dict__[attr] if isinstance(owner, type) and attr in owner.__dict__ else getattr(owner, attr)
    setattr(owner, attr, _wrap(orig, name, caller_file, is_init))
    _installed.append((owner, attr, orig))


def install(caller_file):
    import climb_amd.cl_algorithms as cl
    import climb_amd.cl_algorithms.adapters as ad
    import climb_amd.cl_evaluation.evaluate_cl_algorithm as ev
    import climb_amd.modeling as mo
    import climb_amd.train.task_trainer as tt
    import climb_amd.utils as ut
    from climb_amd.modeling.vilt import ViltContinualLearner, ViltEncoderWrapper
    calls.clear()
    fn = mo.create_continual_learner_map["vilt"]
    mo.create_continual_learner_map["vilt"] = _wrap(fn, "create_continual_learner_map[vilt]", caller_file)
    _installed.append((mo.create_continual_learner_map, "vilt", fn))
    for m in ("state_dict", "load_state_dict", "get_encoder", "parameters"):
        _patch(ViltContinualLearner, m, f"ViltContinualLearner.{m}", caller_file)
    for m in ("state_dict", "freeze_all_weights", "freeze_bottom_k_layers"):
        _patch(ViltEncoderWrapper, m, f"ViltEncoderWrapper.{m}", caller_file)
    _patch(tt.VLTaskTrainer, "__init__", "TaskTrainer", caller_file, is_init=True)
    for m in ("train", "eval", "eval_forgetting"):
        _patch(tt.VLTaskTrainer, m, f"TaskTrainer.{m}", caller_file)
    _patch(tt.LowShotMixin, "__init__", "LowShotTaskTrainer", caller_file, is_init=True)      # REF train_lowshot_multimodal.py:52-53
    _patch(tt.LowShotMixin, "train", "LowShotTaskTrainer.train", caller_file)
    _patch(cl.EWC, "__init__", "EWC", caller_file, is_init=True)
    _patch(cl.EWC, "save_task_parameters", "EWC.save_task_parameters", caller_file)
    _patch(cl.ExperienceReplayMemory, "__init__", "ExperienceReplayMemory", caller_file, is_init=True)
    _patch(cl.ExperienceReplayMemory, "add_task_memory_buffer", "ExperienceReplayMemory.add_task_memory_buffer", caller_file)
    _patch(ad.AdapterHandler, "__init__", "AdapterHandler", caller_file, is_init=True)
    for m in ("add_adapters_to_model", "activate_adapter_for_training", "activate_adapter_for_eval"):
        _patch(ad.AdapterHandler, m, f"AdapterHandler.{m}", caller_file)
    # functions the driver imports BY NAME: wrapped in their defining modules, so install() must run before the driver's imports
    _patch(ev, "upstream_knowledge_transfer_eval", "upstream_knowledge_transfer_eval", caller_file)
    _patch(ev, "catastrophic_forgetting_eval", "catastrophic_forgetting_eval", caller_file)
    _patch(ut, "set_seed", "set_seed", caller_file)
    return calls


def uninstall():
    while _installed:
        owner, attr, orig = _installed.pop()
        if isinstance(owner, dict):
            owner[attr] = orig
        else:
            setattr(owner, attr, orig)
