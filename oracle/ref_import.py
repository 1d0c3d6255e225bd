"""Import recipe for the CLiMB reference.  BUILD-CONTAINER ONLY, TEST INFRASTRUCTURE.

`/root/reference` does not exist on the GPU box; nothing that runs there imports this file.
It is used only by `oracle/gen_golden.py` (to write the fixtures under tests/golden/); the tests read the committed fixtures.

The reference needs three throw-away stubs (wandb, torchvision, jsonlines-free imports) and a
stub `transformers.adapters`; they are created in a temporary directory at run time and never
enter the repository.
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

REF_SRC = "/root/reference/src"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_SRC, "modeling"))


_done = {}


def import_reference():
    """Returns a namespace with the reference's classes.  Order matters:
    real transformers classes first, then the stubs."""
    if _done:
        return _done["ns"]
    os.environ.setdefault("HF_HUB_OFFLINE", "1")
    os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")
    import torch  # noqa: F401
    import transformers
    from transformers import (ViltConfig, ViltModel, ViltProcessor, ViltImageProcessor,  # noqa: F401
                              BertTokenizerFast, BertModel, BertConfig, BertTokenizer,
                              get_polynomial_decay_schedule_with_warmup)

    stub_dir = tempfile.mkdtemp(prefix="climb_ref_stubs_")
    os.makedirs(os.path.join(stub_dir, "wandb"))
    with open(os.path.join(stub_dir, "wandb", "__init__.py"), "w") as f:
        f.write("def init(*a, **k):\n    return None\n\ndef log(*a, **k):\n    return None\n")
    os.makedirs(os.path.join(stub_dir, "torchvision"))
    with open(os.path.join(stub_dir, "torchvision", "__init__.py"), "w") as f:
        f.write("from . import transforms\n")
    with open(os.path.join(stub_dir, "torchvision", "transforms.py"), "w") as f:
        f.write("class _T:\n    def __init__(self, *a, **k):\n        pass\n    def __call__(self, x):\n        return x\n"
                "Compose = Resize = ToTensor = Normalize = CenterCrop = RandomResizedCrop = RandomHorizontalFlip = _T\n")
    os.makedirs(os.path.join(stub_dir, "jsonlines"))
    with open(os.path.join(stub_dir, "jsonlines", "__init__.py"), "w") as f:
        f.write("def open(*a, **k):\n    raise RuntimeError('jsonlines stub')\n")

    # transformers.adapters lives only in the absent GLAMOR fork (REF/.gitmodules:1-3)
    adapters = types.ModuleType("transformers.adapters")

    class AdapterConfig(dict):
        @classmethod
        def load(cls, name, **kw):
            return cls(name=name, **kw)

        def to_dict(self):
            return dict(self)

        @classmethod
        def from_dict(cls, d):
            return cls(**d)

    adapters.AdapterConfig = AdapterConfig
    sys.modules["transformers.adapters"] = adapters

    sys.path.insert(0, stub_dir)
    sys.path.insert(0, REF_SRC)
    ns = types.SimpleNamespace()
    import modeling.vilt as ref_vilt
    # importing the reference re-binds sys.modules['transformers'] (lazy-module refresh), so the
    # fork-only names are patched onto whichever module objects exist *now*
    for mod in {id(transformers): transformers, id(sys.modules["transformers"]): sys.modules["transformers"]}.values():
        mod.adapters = adapters
        for n in ("PfeifferConfig", "HoulsbyConfig", "ParallelConfig", "CompacterConfig"):
            setattr(mod, n, type(n, (AdapterConfig,), {}))
    import cl_algorithms as ref_cl
    import cl_algorithms.ewc as ref_ewc
    import cl_algorithms.experience_replay as ref_er
    from train.visionlanguage_tasks.train_vqa import VQATrainer
    from train.visionlanguage_tasks.train_nlvr2 import NLVR2Trainer
    from train.visionlanguage_tasks.train_snli_ve import SNLIVETrainer
    from train.visionlanguage_tasks.train_vcr import VCRTrainer
    from configs.task_configs import task_configs
    from configs.model_configs import model_configs
    ns.vilt = ref_vilt
    ns.cl = ref_cl
    ns.ewc = ref_ewc
    ns.er = ref_er
    ns.trainers = {"vqa": VQATrainer, "nlvr2": NLVR2Trainer, "snli-ve": SNLIVETrainer, "vcr": VCRTrainer}
    ns.task_configs = task_configs
    ns.model_configs = model_configs
    ns.ViltConfig, ns.ViltModel = ViltConfig, ViltModel
    ns.ViltProcessor, ns.ViltImageProcessor = ViltProcessor, ViltImageProcessor
    ns.BertTokenizerFast = BertTokenizerFast
    ns.get_poly = get_polynomial_decay_schedule_with_warmup
    _done["ns"] = ns
    return ns


def build_reference_learner(tasks, state=None):
    """The reference's own ViltContinualLearner around a random-init HF ViltModel (no network:
    `dandelin/vilt-b32-mlm` cannot be fetched), optionally loaded with `state` (oracle-named dict)."""
    import torch
    ns = import_reference()
    proc = ns.ViltProcessor(image_processor=ns.ViltImageProcessor(),
                            tokenizer=ns.BertTokenizerFast.from_pretrained("bert-base-uncased"))
    enc = ns.vilt.ViltEncoderWrapper(proc, ns.ViltModel(ns.ViltConfig()), torch.device("cpu"))
    model = ns.vilt.ViltContinualLearner(list(tasks), enc, 768, ns.task_configs)
    if state is not None:
        missing, unexpected = model.load_state_dict({k: v.clone() for k, v in state.items()}, strict=False)
        assert not unexpected, unexpected
        assert all("position_ids" in m or "token_type_ids" in m for m in missing), missing
    return model


def make_trainer(task_key, loss=None):
    """A reference TaskTrainer without its dataset-loading __init__."""
    import torch
    from torch import nn
    ns = import_reference()
    cls = ns.trainers[task_key]
    tr = cls.__new__(cls)
    nn.Module.__init__(tr)
    tr.device = torch.device("cpu")
    tr.batch2inputs_converter = ns.vilt.convert_batch_to_vilt_input_dict
    tr.loss_criterion = nn.BCEWithLogitsLoss(reduction="mean") if task_key == "vqa" else nn.CrossEntropyLoss()
    tc = ns.task_configs[task_key]
    tr.hparams = {"lr": tc["lr"], "weight_decay": tc["weight_decay"], "adam_epsilon": tc["adam_epsilon"]}
    return tr


def bypass_processor(model, enc):
    """hand fixed tensor encodings to the model instead of PIL/tokenizer work."""
    model.vilt_encoder.process_inputs = lambda images, texts: {k: v for k, v in enc.items()}


def build_reference_viltbert_learner(tasks, state, bert_state, eager_attention=False):
    """The reference's own ViltBertContinualLearner (REF/modeling/viltbert.py:31-530) around seeded random-init ViltModel / BertModel.
    eager_attention: BERT's attention as explicit softmax -> dropout -> P V (the form of the transformers 4.x lineage the reference pins;
    5.x defaults to SDPA, whose dropout mask cannot be observed)."""
    import torch
    import transformers
    ns = import_reference()
    import modeling.viltbert as ref_vb
    proc = ns.ViltProcessor(image_processor=ns.ViltImageProcessor(), tokenizer=ns.BertTokenizerFast.from_pretrained("bert-base-uncased"))
    cfg = transformers.BertConfig()
    if eager_attention:
        cfg._attn_implementation = "eager"
    bert = transformers.BertModel(cfg)
    missing, unexpected = bert.load_state_dict({k: v.clone() for k, v in bert_state.items()}, strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
    enc = ref_vb.ViltBertEncoderWrapper(proc, ns.ViltModel(ns.ViltConfig()), bert, torch.device("cpu"))
    model = ref_vb.ViltBertContinualLearner(list(tasks), enc, 768, ns.task_configs)
    sd = {k.replace("vilt_encoder.", "viltbert_encoder."): v.clone() for k, v in state.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("position_ids" in m) or ("token_type_ids" in m) or m.startswith("viltbert_encoder.bert.") for m in missing), missing
    return model
