"""ViLT encoder wrapper + continual learner with the reference's API, backed by the HIP step engine.

Mirrors REF/modeling/vilt.py: `ViltEncoderWrapper` (:30-144), `ViltContinualLearner` (:147-367), `load_vilt_encoder`
(:481-514), `create_vilt_continual_learner_model` (:516-546), `convert_batch_to_vilt_input_dict` (:548-553) -- same
names, argument meaning, return values and `state_dict` keys, so REF/train/train_upstream_continual_learning.py and the
reference's trainers / CL plugins can drive it unchanged.  What differs is underneath: parameters are views into one
flat HBM buffer, and forward/backward are hand-written gfx950 kernels reached through a C ABI (climb_amd.engine).
"""
from __future__ import annotations

import copy
import itertools
import logging
import os
import threading
import types
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from ..engine import ViltEngine
from ..layout import ENC, FlatLayout, VILT_CFG, encoder_param_shapes, adapter_param_shapes
from ..optim import FusedAdamW
from .continual_learner import ContinualLearner, EncoderWrapper

logger = logging.getLogger(__name__)
_TOKENIZER_LOCK = threading.Lock()


def default_precision() -> str:
    """CLIMB_AMD_PRECISION, else the 16-bit throughput mode of the library build the process is pinned to (CLIMB_AMD_H16, default bf16)."""
    return os.environ.get("CLIMB_AMD_PRECISION") or ("fp16" if os.environ.get("CLIMB_AMD_H16") == "fp16" else "bf16")


# ----------------------------------------------------------------------------------------------- parameter tree
class _Node(nn.Module):
    """Container that reproduces HF ViltModel's parameter names (numeric children behave like a ModuleList)."""

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __iter__(self):
        return iter(self._modules.values())


def _build_tree(root: nn.Module, shapes: Dict[str, tuple]):
    for name, shape in shapes.items():
        parts = name.split(".")
        node = root
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        node.register_parameter(parts[-1], nn.Parameter(torch.empty(shape)))


def init_like_hf(module: nn.Module, seed: Optional[int] = None):
    """HF ViltPreTrainedModel._init_weights: N(0, 0.02) weights, zero biases, unit norm gains, zero cls / position."""
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    with torch.no_grad():
        for n, p in module.named_parameters():
            if n.endswith("cls_token") or n.endswith("embeddings.position_embeddings") or n.endswith(".bias"):
                p.zero_()
            elif n.endswith("LayerNorm.weight") or ("layernorm" in n and n.endswith(".weight")):
                p.fill_(1.0)
            else:
                p.copy_(torch.empty(p.shape).normal_(0.0, 0.02, generator=gen))


class ViltModelParams(_Node):
    """Stand-in for `transformers.ViltModel` holding exactly its parameters (names, shapes, registration order)."""

    def __init__(self, modality_rows: int = 2, cfg: dict = VILT_CFG):
        super().__init__()
        self.config = types.SimpleNamespace(max_position_embeddings=cfg["max_text"], hidden_size=cfg["hidden"],
                                            modality_type_vocab_size=modality_rows, num_hidden_layers=cfg["layers"])
        _build_tree(self, encoder_param_shapes(modality_rows, cfg))
        self.active_adapters = None
        self.adapter_dims: Dict[str, int] = {}

    # adapter-transformers style methods the reference reaches through `self.vilt_encoder.vilt.<method>`
    # (REF/modeling/vilt.py:357-367).  Arithmetic: Houlsby bottleneck, parity unpinned (see cl_algorithms/adapters.py).
    def add_adapter(self, name: str, config=None):
        if name in self.adapter_dims:
            return
        cfg = dict(config or {})
        rf = int(cfg.get("reduction_factor", 16))
        H = self.config.hidden_size
        r = max(8, (H // rf) // 8 * 8)
        dev = next(self.parameters()).device
        for i in range(self.config.num_hidden_layers):
            shapes = adapter_param_shapes(i, name, r, H)
            _build_tree(self, shapes)
        import zlib
        gen = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        with torch.no_grad():
            for n, p in self.named_parameters():
                if f".adapters.{name}." in n:
                    new = torch.zeros(p.shape) if n.endswith(".bias") else torch.empty(p.shape).normal_(0.0, 0.02, generator=gen)
                    p.data = new.to(dev)
        self.adapter_dims[name] = r

    def train_adapter(self, name: str):
        """Freeze the base model, train only adapter `name` (heads live outside this module and stay trainable)."""
        tag = f".adapters.{name}."
        for n, p in self.named_parameters():
            p.requires_grad = tag in n
        self.active_adapters = name

    def set_active_adapters(self, name):
        if name is not None and name not in self.adapter_dims:
            raise KeyError(f"no adapter named {name!r}")
        self.active_adapters = name


# ----------------------------------------------------------------------------------------------- engine host
class _EngineHost:
    """Binds module parameters to a ViltEngine (views into its flat buffer) and re-binds lazily after `.to()`,
    `copy.deepcopy` or a table resize replaced the tensors (handles must survive deepcopy and
    state_dict round trips, so only plain tensors are held and everything else is re-derived)."""

    def __init__(self, encoder: "ViltEncoderWrapper", heads: Optional[nn.ModuleDict], tasks: List[str], task_configs: Dict[str, dict],
                 precision: str):
        self.encoder, self.heads = encoder, heads
        self.tasks, self.task_configs, self.precision = list(tasks), task_configs, precision
        self._engine: Optional[ViltEngine] = None
        self._params: Dict[str, nn.Parameter] = {}
        self._sentinels: Tuple[str, str] = ("", "")
        self._param_version = None      # sum of the parameters' torch version counters at the last engine() call
        self.ddp = None                 # climb_amd.parallel.GradientAllReducer, when data-parallel

    def __deepcopy__(self, memo):
        new = _EngineHost(copy.deepcopy(self.encoder, memo), copy.deepcopy(self.heads, memo), self.tasks, self.task_configs, self.precision)
        # ViLT-BERT's encoder-level create_optimizer() forwards to the owning learner: the copy must point at the COPIED learner (already in
        # `memo` when the host is reached through it).  The data-parallel reducer is deliberately not carried: a snapshot (best_model) is saved,
        # not trained, and the reducer's engine hook belongs to the live model.
        learner = getattr(self, "learner", None)
        if learner is not None:
            new.learner = memo.get(id(learner), learner)
        return new

    def named(self) -> Dict[str, nn.Parameter]:
        out = {ENC + n: p for n, p in self.encoder.vilt.named_parameters()}
        if self.heads is not None:
            for n, p in self.heads.named_parameters():
                out["task_layer." + n] = p
        return out

    def _arith(self):
        return {t: dict(model_type=self.task_configs[t]["model_type"], num_labels=self.task_configs[t]["num_labels"],
                        num_images=self.task_configs[t].get("num_images", 1), num_choices=self.task_configs[t].get("num_choices", 1))
                for t in self.tasks}

    def _bound(self) -> bool:
        eng = self._engine
        if eng is None or eng.flat is None:
            return False
        if eng.layout.adapters != self.encoder.vilt.adapter_dims:      # adapters were added after binding
            return False
        for n in self._sentinels:
            p = self._params[n]
            if p.data_ptr() != eng.p(n):
                return False
        return True

    def engine(self) -> ViltEngine:
        if not self._bound():
            named = self.named()
            dev = next(iter(named.values())).device
            rows = named[ENC + "embeddings.token_type_embeddings.weight"].shape[0]
            layout = FlatLayout(self.tasks, self._arith(), modality_rows=rows, adapters=self.encoder.vilt.adapter_dims)
            for n, p in named.items():
                assert tuple(p.shape) == tuple(layout.shapes[n]), (n, tuple(p.shape), layout.shapes[n])
            eng = ViltEngine(layout, dev, self.precision, self._arith())
            eng.allocate()
            with torch.no_grad():
                for n, p in named.items():
                    v = eng.view(eng.flat, n)
                    v.copy_(p.data)
                    p.data = v
                    p.grad = None
            names = list(named.keys())
            self._engine, self._params, self._sentinels = eng, named, (names[0], names[-1])
            self._param_version = None
            if self.ddp is not None:
                self.ddp.attach(eng)
        eng = self._engine
        # Any torch-side write to a bound parameter (load_state_dict, p.copy_(), a stock torch optimizer, init code) bumps THAT
        # parameter's version counter, not the flat buffer's: the bf16 operand shadows must be rebuilt from the fp32 master then.
        # (The fused AdamW writes through the C ABI, bumps nothing, and refreshes the shadow itself.)
        ver = 0
        for n, p in self._params.items():
            eng.requires_grad[n] = p.requires_grad
            ver += p._version
        if ver != self._param_version:
            if self._param_version is not None:
                eng.params_updated()
            self._param_version = ver
        eng.active_adapter = self.encoder.vilt.active_adapters
        return eng

    # --- gradient views
    def before_backward(self):
        """If the caller dropped `.grad` (nn.Module.zero_grad / optimizer.zero_grad(set_to_none=True)) since our last
        backward, the flat gradient buffer is stale: clear it so that accumulation semantics match `.grad`."""
        eng = self._engine
        # a weight-gradient launch that was held back for an optimizer step that never came: it reads the saved activations of its
        # workspace through raw pointers, so it has to run BEFORE the next forward overwrites them, as the plain launch (C += dW)
        if eng._dw_deferred:
            eng.materialize_dw()
        if eng.touched:
            for n, p in self._params.items():
                if p.requires_grad and eng.is_touched(n):
                    if p.grad is None:
                        eng.zero_grad()
                    break
        if eng.touched:
            # gradients of an earlier backward were kept (accumulation, torch's semantics when zero_grad() is not called): the matrices' ranges hold
            # sums too, whichever launch wrote them (plain grouped, immediate, the non-fused problems of a fused launch) -- the optimizer-carrying
            # epilogue must add what is there instead of assuming zeros
            eng._grad_extra = True
            if eng._fused_consumed:
                import warnings
                warnings.warn("climb_amd: backward on top of gradients that FusedAdamW.step() already consumed inside the weight-gradient launch "
                              "(no zero_grad() since): those matrices' earlier gradients were never stored and are not part of the accumulated sum")
        eng._fused_consumed = False
        eng._grad_clean = False          # gradients are about to be written
        if eng._g16 is not None:         # (a backward on top of averaged gradients nobody consumed: they belong in the buffer this one accumulates into)
            eng.materialize_g16()

    def after_backward(self):
        """`.grad` of every parameter that received a gradient aliases its slice of the flat buffer; the others stay
        None, which is what torch's optimizers (and ours) use to skip them."""
        eng = self._engine
        for n, p in self._params.items():
            if p.requires_grad and eng.is_touched(n):
                if p.grad is None or p.grad.data_ptr() != eng.g(n):
                    p.grad = eng.view(eng.grad, n)

    def drop_grads(self):
        self._engine.zero_grad()
        for p in self._params.values():
            p.grad = None

    def frozen_prefix(self) -> Tuple[int, bool]:
        """(first layer that needs gradients, whether the embeddings do) -- REF/modeling/vilt.py:126-144 freezes."""
        enc = self.encoder.vilt
        emb = any(p.requires_grad for p in enc.embeddings.parameters())
        if emb:
            return 0, True
        first = 0
        layers = enc.encoder.layer
        while first < len(layers) and not any(p.requires_grad for p in layers[first].parameters()):
            first += 1
        return first, False

    def any_encoder_grad(self) -> Optional[nn.Parameter]:
        for p in self.encoder.vilt.parameters():
            if p.requires_grad:
                return p
        return None

    # --- autograd-aware entry points (used when a reference-style trainer calls model(...) then loss.backward())
    def encode(self, enc: Dict[str, torch.Tensor]) -> torch.Tensor:
        eng = self.engine()
        B = enc["token_type_ids"].shape[0]
        it = enc.get("image_token_type_idx", None)
        if it is None:
            it = 1
        img_type = it.to(torch.int32) if isinstance(it, torch.Tensor) else torch.full((B,), int(it), dtype=torch.int32, device=eng.device)
        sentinel = self.any_encoder_grad() if torch.is_grad_enabled() else None
        if sentinel is None:
            return eng.encoder_forward(enc.get("input_ids"), enc["token_type_ids"], enc["attention_mask"], enc["pixel_values"], img_type,
                                       save=False, pixel_mask=enc.get("pixel_mask"), inputs_embeds=enc.get("inputs_embeds")).clone()
        return _EncoderFn.apply(sentinel, self, enc, img_type)

    def head(self, task_key: str, pooled_in: torch.Tensor, training: bool) -> torch.Tensor:
        eng = self.engine()
        needs = torch.is_grad_enabled() and (pooled_in.requires_grad or any(p.requires_grad for p in self.heads[task_key].parameters()))
        if not needs:
            return eng.head_forward(task_key, pooled_in.contiguous(), training)[0]
        sentinel = next(self.heads[task_key].parameters())
        return _HeadFn.apply(pooled_in, sentinel, self, task_key, training)


class _EncoderFn(torch.autograd.Function):
    """Lets `loss.backward()` in the reference's own trainers (REF train_vqa.py:159-166) reach the HIP backward.
    Parameter gradients are written straight into the flat grad buffer that every `p.grad` aliases."""

    @staticmethod
    def forward(ctx, sentinel, host, enc, img_type):
        eng = host._engine
        pooled = eng.encoder_forward(enc.get("input_ids"), enc["token_type_ids"], enc["attention_mask"], enc["pixel_values"], img_type,
                                     pixel_mask=enc.get("pixel_mask"), inputs_embeds=enc.get("inputs_embeds"))
        ctx.host = host
        ctx.generation = eng.saved["generation"]
        return pooled.clone()

    @staticmethod
    def backward(ctx, dpooled):
        host = ctx.host
        if host._engine.saved is None or host._engine.saved.get("generation") != ctx.generation:
            # activations live in the engine's single workspace, not in the autograd graph: a second grad-enabled forward
            # before this backward has overwritten them
            raise RuntimeError("climb_amd: backward of a forward whose saved activations were overwritten by a later forward; "
                               "run backward before the next grad-enabled forward (or sum the losses of ONE forward)")
        host.before_backward()
        first, emb = host.frozen_prefix()
        eng = host._engine
        dpooled = dpooled.contiguous().float()
        if eng.h16 == "fp16":
            if not getattr(eng, "_scaled_by_head", False):       # no head of this package upstream of us in this backward: scale here
                dpooled = dpooled * eng.begin_scaled_backward(float(dpooled.abs().max()))
            eng._scaled_by_head = False
        host._engine.encoder_backward(dpooled, first_layer=first, embeddings=emb)
        eng.finish_scaled_backward()
        host.after_backward()
        if host.ddp is not None:
            host.ddp.finish()
        return None, None, None, None


class _HeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pooled_in, sentinel, host, task_key, training):
        logits, hs = host._engine.head_forward(task_key, pooled_in.contiguous(), training)
        ctx.host, ctx.hs = host, hs
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        host = ctx.host
        host.before_backward()
        eng = host._engine
        dlogits = dlogits.contiguous().float()
        if eng.h16 == "fp16":       # torch computed this d(logits): bring it into IEEE half's range (one host sync, compatibility path only);
            # the scaled d(pooled) flows through autograd into _EncoderFn.backward and every range is unscaled in the engine's _ready()
            dlogits = dlogits * eng.begin_scaled_backward(float(dlogits.abs().max()))
            eng._scaled_by_head = True
        dx = eng.head_backward(ctx.hs, dlogits)
        if eng.h16 == "fp16" and not ctx.needs_input_grad[0]:       # nothing upstream of the head needs a gradient: the backward ends here
            eng._scaled_by_head = False
            eng.finish_scaled_backward()
        host.after_backward()
        return dx, None, None, None, None


# ----------------------------------------------------------------------------------------------- encoder wrapper
class ViltEncoderWrapper(EncoderWrapper):
    """REF/modeling/vilt.py:30-144.  `vilt` holds the parameters (keys `vilt.*` in this module's state_dict, which is what
    the reference saves as the `encoder` checkpoint and what EWC keys its dictionaries on)."""

    def __init__(self, processor, vilt: ViltModelParams, device: torch.device, precision: Optional[str] = None):
        super().__init__()
        self.processor = processor
        self.vilt = vilt
        self.device = torch.device(device)
        self.max_text_length = self.vilt.config.max_position_embeddings
        self.encoder_dim = self.vilt.config.hidden_size
        self.precision = precision or default_precision()
        self._host: Optional[_EngineHost] = None
        self._image_pipeline = None           # climb_amd.data.DeviceImagePipeline, built on first use

    def __deepcopy__(self, memo):
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = v if k in ("processor", "_image_pipeline") else copy.deepcopy(v, memo)
        return new

    # --- inputs (REF:83-96)
    def process_inputs(self, images, texts, pipeline=None) -> Dict[str, torch.Tensor]:
        """`pipeline`: a DeviceImagePipeline of the caller's own (a prefetch worker thread must not share the training thread's pinned
        staging ring); default = this encoder's."""
        dev = self.device
        if isinstance(texts, dict):                               # pre-tokenised text + pre-processed pixel tensor(s)
            enc = dict(texts)
            if isinstance(images, dict):
                enc.update(images)
            else:
                enc["pixel_values"] = images
            if "token_type_ids" not in enc:
                enc["token_type_ids"] = torch.zeros_like(enc["input_ids"])
            if "attention_mask" not in enc:
                enc["attention_mask"] = torch.ones_like(enc["input_ids"])
            # no pixel_mask key == "every image fills the 384x384 canvas" (the engine's fixed-resolution fast path)
            return {k: v.to(dev, non_blocking=True) for k, v in enc.items()}
        if self.processor is None:
            raise RuntimeError("no ViltProcessor attached: pass tensor encodings (texts=dict(input_ids=...), images=pixel tensor) "
                               "or construct the encoder with a processor")
        if dev.type == "cuda":
            # row F1: tokenise on the host, but resize / rescale / normalise / pad the images on the device from their raw bytes
            # (bit-identical to ViltProcessor's tensors, a quarter of its host->device traffic, none of its host arithmetic)
            with _TOKENIZER_LOCK:       # the fast tokenizer mutates its padding / truncation state per call: one caller at a time
                enc = self.processor.tokenizer(texts, max_length=self.max_text_length, padding=True, truncation=True, return_tensors="pt")
            enc = {k: v.to(dev, non_blocking=True) for k, v in enc.items()}
            if pipeline is None:
                if self._image_pipeline is None:
                    from ..data import DeviceImagePipeline
                    self._image_pipeline = DeviceImagePipeline(dev)
                pipeline = self._image_pipeline
            enc.update(pipeline(images))
            return enc
        # module left on a CPU device (host-side inspection only: the engine refuses to run there): the reference's own host call
        enc = self.processor(images=images, text=texts, max_length=self.max_text_length, padding=True, truncation=True, return_tensors="pt")
        return {k: v.to(dev, non_blocking=True) for k, v in enc.items()}

    def prepare_encodings(self, enc: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """Last step before the encoder kernels; the plain ViLT wrapper has nothing to add (ViLT-BERT puts BERT's features here)."""
        return enc

    def expand_modality_type_embeddings(self, type_vocab_size=3):
        """REF:98-109: third modality row = copy of the second."""
        old = self.vilt.embeddings.token_type_embeddings.weight.data
        if old.shape[0] >= type_vocab_size:
            return
        self.vilt.config.modality_type_vocab_size = type_vocab_size
        new = torch.empty((type_vocab_size, self.encoder_dim), dtype=old.dtype, device=old.device)
        new[0], new[1], new[2] = old[0], old[1], old[1]
        self.vilt.embeddings.token_type_embeddings.weight = nn.Parameter(new)
        if self._host is not None:
            self._host._engine = None

    def host(self) -> _EngineHost:
        if self._host is None:
            self._host = _EngineHost(self, None, [], {}, self.precision)
        return self._host

    def forward(self, **encodings) -> torch.FloatTensor:
        """REF:111-124: pooler_output [B, 768].  Accepts `image_token_type_idx` (int, or int32 tensor [B])."""
        return self.host().encode(self.prepare_encodings(encodings))

    def freeze_all_weights(self):
        for p in self.vilt.parameters():
            p.requires_grad = False

    def freeze_bottom_k_layers(self, k: int):
        assert k < len(self.vilt.encoder.layer)
        for p in self.vilt.embeddings.parameters():
            p.requires_grad = False
        for i in range(k):
            for p in self.vilt.encoder.layer[i].parameters():
                p.requires_grad = False


# ----------------------------------------------------------------------------------------------- continual learner
class ViltContinualLearner(ContinualLearner):
    """REF/modeling/vilt.py:147-367."""
    encoder_attr = "vilt_encoder"          # the attribute (and state_dict prefix) the encoder wrapper is registered under

    @property
    def _enc(self) -> ViltEncoderWrapper:
        return getattr(self, self.encoder_attr)

    def __init__(self, ordered_cl_tasks: List[str], encoder: ViltEncoderWrapper, encoder_dim: int, task_configs: Dict):
        super().__init__()
        self.encoder_dim = encoder_dim
        setattr(self, self.encoder_attr, encoder)
        self.ordered_cl_tasks = ordered_cl_tasks
        self.task_configs = task_configs
        self.task_layer_dict = {}
        for task_key in ordered_cl_tasks:
            self.add_task_layer(task_key, task_configs[task_key])
        self.task_layer = nn.ModuleDict(self.task_layer_dict)
        if "nlvr2" in ordered_cl_tasks:
            self._enc.expand_modality_type_embeddings()
        self.task_layer.to(next(self._enc.vilt.parameters()).device)
        self._host = _EngineHost(self._enc, self.task_layer, list(ordered_cl_tasks), task_configs, self._enc.precision)
        self._enc._host = self._host

    def add_task_layer(self, task_key: str, task_config: Dict):
        """REF:179-203.  torch modules are used as parameter containers (names `0.weight` ... `3.bias`); the arithmetic
        runs in climb_amd.engine.head_forward."""
        num_labels = task_config["num_labels"]
        if task_config["model_type"] == "classification":
            num_images = task_config["num_images"]
            clf_layer = nn.Sequential(nn.Linear(self.encoder_dim * num_images, self.encoder_dim * 2), nn.LayerNorm(self.encoder_dim * 2),
                                      nn.GELU(), nn.Linear(self.encoder_dim * 2, num_labels))
        elif task_config["model_type"] == "multi-choice":
            clf_layer = nn.Sequential(nn.Dropout(0.1), nn.Linear(self.encoder_dim, 1))
        else:
            raise NotImplementedError(task_config["model_type"])
        self.task_layer_dict[task_key] = clf_layer

    def create_optimizer(self, hparams):
        """REF:205-215: AdamW(betas=(0.9, 0.98)), two groups by substring match on ['bias', 'LayerNorm.weight'] -- as ONE fused
        HIP kernel over the flat buffer."""
        no_decay = ["bias", "LayerNorm.weight"]
        groups = [
            {"params": [p for n, p in self.named_parameters() if not any(nd in n for nd in no_decay)], "weight_decay": hparams["weight_decay"]},
            {"params": [p for n, p in self.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
        ]
        return FusedAdamW(groups, lr=hparams["lr"], eps=hparams["adam_epsilon"], betas=(0.9, 0.98), host=self._host)

    # --- forward variants (REF:218-350).  Multi-image / multi-choice passes are batched into ONE encoder call of
    # b*num_images (b*num_choices) sequences: samples are independent, so this is arithmetically identical to the
    # reference's Python loop of encoder passes.
    def forward(self, task_key: str, images: List, texts: List[str]):
        task_config = self.task_configs[task_key]
        if task_config["model_type"] == "multi-choice":
            return self.forward_multi_choice(task_key, images, texts, task_config["num_choices"])
        if task_config["num_images"] == 1:
            return self.forward_single_image(task_key, images, texts)
        return self.forward_multi_images(task_key, images, texts, task_config["num_images"])

    def _expand(self, task_key: str, enc: Dict[str, torch.Tensor]):
        """processor output -> per-sequence encoder inputs + image_token_type_idx, for every task shape."""
        tc = self.task_configs[task_key]
        if tc["model_type"] == "multi-choice":
            nc = tc["num_choices"]
            enc = dict(enc)
            enc["pixel_values"] = enc["pixel_values"].repeat_interleave(nc, dim=0)     # choice j of example i at row nc*i+j (REF:331-334)
            if "pixel_mask" in enc:
                hint = getattr(enc["pixel_mask"], "_climb_max_patches", None)
                enc["pixel_mask"] = enc["pixel_mask"].repeat_interleave(nc, dim=0)
                if hint is not None:
                    enc["pixel_mask"]._climb_max_patches = hint
            enc["image_token_type_idx"] = 1
            return enc, ("choice", nc)
        n = tc["num_images"]
        if n == 1:
            enc = dict(enc)
            enc["image_token_type_idx"] = 1
            return enc, ("single", 1)
        enc = dict(enc)                                                                 # image j of example i at row n*i+j (REF:281-288)
        for k in ("input_ids", "token_type_ids", "attention_mask"):
            enc[k] = enc[k].repeat_interleave(n, dim=0)
        bn = enc["pixel_values"].shape[0]
        enc["image_token_type_idx"] = (torch.arange(bn, device=enc["pixel_values"].device) % n + 1).to(torch.int32)   # REF:299
        return enc, ("images", n)

    def _shape_pooled(self, pooled, kind):
        what, n = kind
        if what == "single":
            return pooled
        if what == "images":
            return pooled.view(pooled.shape[0] // n, n * pooled.shape[1])          # == torch.cat(pooler_outputs, -1), REF:304
        return pooled.view(pooled.shape[0] // n, n, pooled.shape[1])                # == stack(...).transpose(0,1), REF:347

    def _flatten_inputs(self, task_key, images, texts):
        tc = self.task_configs[task_key]
        if isinstance(texts, dict):
            return images, texts
        if tc["model_type"] == "multi-choice":
            return images, list(itertools.chain(*texts))
        if tc["num_images"] > 1:
            return list(itertools.chain(*images)), texts
        return images, texts

    def _forward_any(self, task_key, images, texts):
        images, texts = self._flatten_inputs(task_key, images, texts)
        enc = self._enc.process_inputs(images, texts)
        enc, kind = self._expand(task_key, enc)
        pooled = self._shape_pooled(self._enc(**enc), kind)
        logits = self._host.head(task_key, pooled, self.training)
        return pooled, logits

    def forward_single_image(self, task_key, images, texts):
        return self._forward_any(task_key, images, texts)

    def forward_multi_images(self, task_key, images, texts, num_images=2):
        return self._forward_any(task_key, images, texts)

    def forward_multi_choice(self, task_key, images, texts, num_choices):
        return self._forward_any(task_key, images, texts)

    def get_encoder(self):
        return self._enc

    def prepare_batch(self, task_key: str, batch: Dict, converter=None, pipeline=None) -> Dict:
        """The host half of a step, movable off the training thread (REF/modeling/vilt.py:83-96 runs it inline, every
        step): tokenise, stage the raw image bytes, launch the device image kernels and the host->device copies on the CURRENT stream.
        Returns the batch with `encodings` (text tensors) and `images` (pixel tensors) in place of strings / PIL images -- what
        `forward` / `fused_forward_backward` accept directly -- and its tensor targets already on the device.  Idempotent."""
        if "encodings" in batch and isinstance(batch.get("images"), dict):
            return batch
        inputs = (converter or convert_batch_to_vilt_input_dict)(batch)
        images, texts = self._flatten_inputs(task_key, inputs["images"], inputs["texts"])
        if isinstance(texts, dict):
            return batch
        enc = self._enc.process_inputs(images, texts, pipeline=pipeline)
        out = dict(batch)
        out["encodings"] = {k: v for k, v in enc.items() if not k.startswith("pixel_")}
        out["images"] = {k: v for k, v in enc.items() if k.startswith("pixel_")}
        dev = self._enc.device
        for k in ("target_scores", "labels"):
            if isinstance(out.get(k), torch.Tensor):
                out[k] = out[k].to(dev, non_blocking=True)
        return out

    # --- fused training step: forward + loss + backward (+ EWC term) with no autograd graph.  This is what
    # climb_amd.train.*Trainer.train_step runs; semantics = REF/train/visionlanguage_tasks/train_vqa.py:135-166.
    def fused_forward_backward(self, task_key: str, images, texts, target: torch.Tensor, ewc=None, dropout_keep=None, grad_weight: float = 1.0, optimizer=None,
                               dp_rows: Optional[float] = None):
        """`grad_weight` multiplies d(loss) (not the returned loss): a data-parallel rank's share of an uneven global batch,
        climb_amd/data/sharding.py.
        `optimizer`: the caller's promise that `optimizer.step()` (this model's FusedAdamW) is the next thing that happens to the gradients
        (REF/train/visionlanguage_tasks/train_vqa.py:160-170: backward, step, zero_grad).  The grouped weight-gradient launch of the encoder is then
        held back and run BY that step with AdamW in its epilogue (csrc/gemm_bf16_tnp.hip): `.grad` of those matrices is never written.  Without the
        promise -- or whenever something else reads the buffer first -- the launch runs as before.
        DEFERRED-VALUE CONTRACT of the promise: with an EWC plug-in the returned `ewc_loss` tensor RECEIVES its value when `optimizer.step()`
        runs (it reads 0 until then), and the gradient buffer / `.grad` never holds the 2 lam F (theta - theta*) term -- the optimizer adds it inside its
        own passes (engine.park_ewc).  REF/train/visionlanguage_tasks/train_vqa.py:160-170 reads the value after the step, as climb_amd's trainers do; a
        caller that wants loss + ewc_loss, or the penalised gradient, BEFORE the step must not name its optimizer (or call `engine.apply_parked_ewc()`)."""
        host = self._host
        eng = host.engine()
        from ..optim import FusedAdamW
        promised = isinstance(optimizer, FusedAdamW) and optimizer._host is host
        eng.defer_dw = bool(promised and host.ddp is None and os.environ.get("CLIMB_AMD_FUSED_ADAMW", "1") != "0")
        # data parallel: with the same promise the averaged 16-bit payload is not cast back into the gradient buffer; FusedAdamW.step() reads it in place
        # (not when an EWC term has to be ADDED to the averaged gradients: that needs them in fp32)
        self._defer_uncast = bool(promised and host.ddp is not None and not (ewc is not None and ewc.do_ewc()))
        try:
            return self._fused_forward_backward(task_key, images, texts, target, ewc, dropout_keep, grad_weight, dp_rows)
        finally:
            eng.defer_dw = False
            self._defer_uncast = False

    def _fused_forward_backward(self, task_key: str, images, texts, target: torch.Tensor, ewc=None, dropout_keep=None, grad_weight: float = 1.0, dp_rows=None):
        host = self._host
        eng = host.engine()
        host.before_backward()
        images, texts = self._flatten_inputs(task_key, images, texts)
        enc = self._enc.process_inputs(images, texts)
        enc, kind = self._expand(task_key, enc)
        enc = self._enc.prepare_encodings(enc)
        B = enc["token_type_ids"].shape[0]
        it = enc["image_token_type_idx"]
        img_type = it if isinstance(it, torch.Tensor) else torch.full((B,), int(it), dtype=torch.int32, device=eng.device)
        if host.ddp is not None:
            host.ddp.begin()
        pooled_seq = eng.encoder_forward(enc.get("input_ids"), enc["token_type_ids"], enc["attention_mask"], enc["pixel_values"], img_type,
                                         pixel_mask=enc.get("pixel_mask"), inputs_embeds=enc.get("inputs_embeds"))
        pooled = self._shape_pooled(pooled_seq, kind)
        logits, hs = eng.head_forward(task_key, pooled, self.training, dropout_keep, reuse=True)
        target = target.to(eng.device, non_blocking=True)
        if task_key == "vqa":
            target = target.float()
        # fp16 operands: d(logits) is produced already multiplied by the loss scale (every |d logit| of both losses is <= 1 / rows)
        # Under data parallelism the scale must be THE SAME on every rank (the half payload is reduced still scaled and every rank divides the average by
        # its own scale: r04 -- ranks with 2 and 1 examples of a 3-example batch used to pick 2^8 and 2^7 and diverged).  A rank's |d logit| is bounded by
        # weight / rows = ranks / examples of the global batch, whatever its share; `dp_rows` (the sharded loader) carries that number to ranks of weight 0 too.
        rows_l = max(1, logits.shape[0])
        bound = (1.0 / float(dp_rows)) if dp_rows else ((float(grad_weight) / rows_l) if grad_weight > 0 else 1.0 / rows_l)
        gs = eng.begin_scaled_backward(bound) * float(grad_weight)
        loss, dlogits = eng.loss_and_grad(task_key, logits, target, gscale=gs, hs=hs)
        dpool = eng.head_backward(hs, dlogits, dtanh_of=pooled_seq)
        first, emb = host.frozen_prefix()
        if host.any_encoder_grad() is not None:
            eng.encoder_backward(dpool.reshape(B, -1).contiguous(), first_layer=first, embeddings=emb, dpooled_is_dpre=bool(getattr(hs, "dx_is_dpre", False)))
        eng.saved = None           # (also when the encoder is frozen and its backward never ran)
        eng.finish_scaled_backward()
        if host.ddp is not None:
            host.ddp.finish(defer_uncast=bool(getattr(self, "_defer_uncast", False)))
        ewc_task, ewc_loss = None, None
        if ewc is not None and ewc.do_ewc():
            # (r05) with the optimizer named and a plain single-GPU bf16 step, the term rides in the optimizer's passes (EWC.park_penalty)
            fold = (eng.defer_dw and eng.loss_scale == 1.0 and not eng._grad_extra and eng.active_adapter is None and hasattr(ewc, "park_penalty")
                    and os.environ.get("CLIMB_AMD_EWC_FOLD", "2") != "0")
            ewc_task, ewc_loss = ewc.park_penalty(self) if fold else ewc.add_penalty_gradient(self)
        host.after_backward()
        # `pooled`, `logits` and `loss` live in buffers the next step overwrites: hand the caller its own copies (a few hundred KB)
        return loss.clone(), (pooled.clone(), logits.clone()), ewc_task, ewc_loss

    # --- the same step captured once into a hipGraph and replayed: the ~330 kernel launches of a step become one graph launch
    # (HIP streams and graphs instead of a tracing compiler).  Inputs are copied into static buffers; the returned tensors are
    # the graph's static outputs (overwritten by the next replay).  Falls back to the eager path under data parallelism / EWC.
    def graphed_forward_backward(self, task_key: str, images, texts, target: torch.Tensor, ewc=None, dropout_keep=None, grad_weight: float = 1.0):
        host = self._host
        if grad_weight != 1.0:
            return self.fused_forward_backward(task_key, images, texts, target, ewc, dropout_keep, grad_weight)
        if host.ddp is not None or dropout_keep is not None or not isinstance(texts, dict) or hasattr(self._enc, "bert") or \
                (self.training and self.task_configs[task_key]["model_type"] == "multi-choice"):
            return self.fused_forward_backward(task_key, images, texts, target, ewc, dropout_keep)
        eng = host.engine()
        if eng.h16 == "fp16" and eng._grad_dirty:
            # accumulating onto earlier (unscaled) sums needs the pre-scaling pass of begin_scaled_backward(), which a graph captured
            # from clean gradients does not contain
            return self.fused_forward_backward(task_key, images, texts, target, ewc, dropout_keep)
        img = images if isinstance(images, dict) else {"pixel_values": images}
        flags = tuple(sorted(n for n, p in host._params.items() if not p.requires_grad))
        key = (task_key, self.training, eng.active_adapter, eng.cls_only_last, hash(flags), tuple(target.shape), target.dtype,
               tuple((k, tuple(v.shape)) for k, v in sorted(texts.items())), tuple((k, tuple(v.shape)) for k, v in sorted(img.items())))
        graphs = self.__dict__.setdefault("_graphs", {})
        cs = graphs.get(key)
        if cs is None or cs["engine"] is not eng:
            dev = eng.device
            st_texts = {k: v.to(dev).clone() for k, v in texts.items()}
            st_img = {k: v.to(dev).clone() for k, v in img.items()}
            st_target = target.to(dev).clone()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                  # warm-up off the capture: allocates workspaces, raises LDS caps
                for _ in range(2):
                    self.fused_forward_backward(task_key, st_img, st_texts, st_target)
            torch.cuda.current_stream().wait_stream(side)
            host.drop_grads()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self.fused_forward_backward(task_key, st_img, st_texts, st_target)
            touched = list(eng.touched)
            host.drop_grads()                              # capture does not execute; start from clean gradients
            # the captured kernels hold raw pointers into this Workspace: the cache entry keeps it alive past the engine's
            # own three-shape eviction
            cs = graphs[key] = dict(graph=g, texts=st_texts, img=st_img, target=st_target, out=out, touched=touched, engine=eng,
                                    ws=eng.last_ws)
        host.before_backward()
        for k, v in texts.items():
            cs["texts"][k].copy_(v, non_blocking=True)
        for k, v in img.items():
            cs["img"][k].copy_(v, non_blocking=True)
        cs["target"].copy_(target, non_blocking=True)
        eng.refresh_shadow()
        cs["graph"].replay()
        eng._grad_dirty = True
        eng.touched.extend(t for t in cs["touched"] if t not in eng.touched)
        loss, output, _, _ = cs["out"]
        ewc_task, ewc_loss = None, None
        if ewc is not None and ewc.do_ewc():
            ewc_task, ewc_loss = ewc.add_penalty_gradient(self)
        host.after_backward()
        return loss, output, ewc_task, ewc_loss

    # --- adapters (REF:357-367); arithmetic of the absent GLAMOR fork is unpinned, see climb_amd/cl_algorithms/adapters.py
    def add_adapter(self, task_key: str, config: Dict):
        self._enc.vilt.add_adapter(task_key, config)

    def train_adapter(self, task_key: str):
        self._enc.vilt.train_adapter(task_key)

    def set_active_adapters(self, task_key: str):
        self._enc.vilt.set_active_adapters(task_key)

    def get_active_adapters(self):
        return self._enc.vilt.active_adapters


# ----------------------------------------------------------------------------------------------- factories
def _load_encoder_state(vilt_encoder: ViltEncoderWrapper, path: str):
    """Accepts a CLiMB `encoder` checkpoint (keys `vilt.*`, REF train_upstream...:266) or an HF ViLT checkpoint directory/file
    (keys `vilt.*` plus heads we ignore; 4.x checkpoints also carry `...text_embeddings.position_ids`)."""
    if os.path.isdir(path):
        cand = [os.path.join(path, f) for f in ("model.safetensors", "pytorch_model.bin") if os.path.exists(os.path.join(path, f))]
        if not cand:
            raise OSError(f"no model.safetensors / pytorch_model.bin under {path}")
        path = cand[0]
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu")
    own = vilt_encoder.state_dict()
    picked = {k: v for k, v in sd.items() if k in own}
    missing = [k for k in own if k not in picked]
    if missing:
        raise KeyError(f"checkpoint {path} lacks {len(missing)} encoder tensors, e.g. {missing[:3]}")
    vilt_encoder.load_state_dict(picked)


def _offline_processor():
    """A ViltProcessor built from local pieces when the hub is unreachable: `CLIMB_AMD_TOKENIZER_VOCAB` names a BERT word-piece
    vocabulary file; the image half is transformers' own PIL ViLT image processor (defaults = dandelin/vilt-b32-mlm's)."""
    vocab = os.environ.get("CLIMB_AMD_TOKENIZER_VOCAB")
    if not vocab:
        return None
    import transformers
    try:
        from transformers.models.vilt.image_processing_pil_vilt import ViltImageProcessorPil as ImageProcessor
    except Exception:      # noqa: BLE001  (older transformers: one image processor class)
        from transformers import ViltImageProcessor as ImageProcessor
    words = [w.strip() for w in open(vocab) if w.strip()]
    for kw in ({"vocab": vocab}, {"vocab_file": vocab}):         # transformers 5.x / 4.x spelling
        try:
            tok = transformers.BertTokenizerFast(do_lower_case=True, **kw)
        except Exception:      # noqa: BLE001
            continue
        if len(words) > 5 and tok.convert_tokens_to_ids(words[5]) == 5:
            return transformers.ViltProcessor(image_processor=ImageProcessor(), tokenizer=tok)
    raise RuntimeError(f"could not build a tokenizer from CLIMB_AMD_TOKENIZER_VOCAB={vocab}")


def _make_processor(pretrained_vilt_name: str):
    try:
        from transformers import ViltProcessor
        return ViltProcessor.from_pretrained(pretrained_vilt_name)
    except Exception as e:   # offline container: tensor encodings still work (bench / tests / pre-processed pipelines)
        proc = _offline_processor()
        if proc is None:
            logger.warning("ViltProcessor.from_pretrained(%s) unavailable (%s); only tensor encodings are accepted", pretrained_vilt_name, type(e).__name__)
        return proc


def load_vilt_encoder(checkpoint_name: str, device: torch.device, pretrained_vilt_name: str, precision: Optional[str] = None) -> ViltEncoderWrapper:
    """REF/modeling/vilt.py:481-514.  `random-init[:seed]` builds the architecture with HF's initialiser (no network here)."""
    logger.info("-" * 100)
    logger.info("Loading ViLT encoder model: {}".format(checkpoint_name))
    device = torch.device(device)
    if checkpoint_name.startswith("random-init"):
        # `random-init:empty`: the architecture with its parameters left at zero -- for callers that load a state dict next (the GPU tests build
        # ~150 models and HF's truncated-normal initialiser is 2 s of host time each)
        empty = checkpoint_name == "random-init:empty"
        seed = int(checkpoint_name.split(":")[1]) if (":" in checkpoint_name and not empty) else None
        vilt = ViltModelParams(2)
        if not empty:
            init_like_hf(vilt, seed)
        enc = ViltEncoderWrapper(_offline_processor(), vilt.to(device), device, precision)
        return enc
    processor = _make_processor(pretrained_vilt_name)
    rows = 3 if (checkpoint_name != pretrained_vilt_name and "nlvr2" in checkpoint_name) else 2      # REF:507-508
    vilt = ViltModelParams(rows)
    init_like_hf(vilt)
    enc = ViltEncoderWrapper(processor, vilt, device, precision)
    _load_encoder_state(enc, checkpoint_name)
    enc.vilt.to(device)
    logger.info("Successfully loaded pretrained ViLT encoder")
    return enc


def create_vilt_continual_learner_model(model_name_or_path: str, ordered_cl_tasks: List[str], model_config: Dict, task_configs: Dict,
                                        device: torch.device, precision: Optional[str] = None):
    """REF/modeling/vilt.py:516-546 (same keyword signature; `precision` is an optional extra)."""
    encoder = load_vilt_encoder(checkpoint_name=model_name_or_path, device=device, pretrained_vilt_name=model_name_or_path, precision=precision)
    cl_model = ViltContinualLearner(ordered_cl_tasks=ordered_cl_tasks, encoder=encoder, encoder_dim=model_config["encoder_dim"],
                                    task_configs=task_configs)
    logger.info("Successfully created and initialized ViLT Continual Learner model")
    return cl_model


def convert_batch_to_vilt_input_dict(batch: Dict):
    """REF/modeling/vilt.py:548-553.  Batches from a tensor pipeline may carry `encodings` instead of raw text."""
    return {"images": batch["images"], "texts": batch["raw_texts"] if "encodings" not in batch else batch["encodings"]}
