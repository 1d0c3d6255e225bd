"""Parameter naming and the flat HBM layout of the ViLT continual learner.

Names are the reference's own (`ViltContinualLearner.named_parameters()`: `vilt_encoder.vilt.*`, `task_layer.<task>.<i>.*`,
REF/modeling/vilt.py:147-203 over HF ViltModel) because checkpoints and the EWC dictionaries key on them (SURVEY.md
§8(b)).  Physically every tensor lives in ONE flat fp32 buffer, 64-element aligned, in FORWARD order (embeddings, layer
0..11, final norm, pooler, heads) so that

  * the encoder is one contiguous range   -> EWC penalty / Fisher accumulate / theta* snapshot are single launches
  * a layer is one contiguous range       -> its gradients form one all-reduce bucket the moment its backward finishes
  * q/k/v weights (and biases) are adjacent -> the fused [2304, 768] QKV projection is a view, not a copy
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Tuple

ENC = "vilt_encoder.vilt."
ALIGN = 64

VILT_CFG = dict(hidden=768, heads=12, head_dim=64, ffn=3072, layers=12, vocab=30522, max_text=40, type_vocab=2,
                patch=32, image=384, channels=3, ln_eps=1e-12, head_ln_eps=1e-5)

# arithmetic-relevant slice of REF/configs/task_configs.py:16-95
TASK_ARITH = {
    "vqa": dict(model_type="classification", num_labels=3129, num_images=1),
    "nlvr2": dict(model_type="classification", num_labels=2, num_images=2),
    "snli-ve": dict(model_type="classification", num_labels=3, num_images=1),
    "vcr": dict(model_type="multi-choice", num_labels=4, num_choices=4, num_images=1),
}


def encoder_param_shapes(modality_rows: int = 2, cfg: dict = VILT_CFG) -> "OrderedDict[str, tuple]":
    """HF ViltModel registration order (names relative to the ViltModel, i.e. without the `vilt_encoder.vilt.` prefix)."""
    H, Fd = cfg["hidden"], cfg["ffn"]
    npatch = (cfg["image"] // cfg["patch"]) ** 2
    s: "OrderedDict[str, tuple]" = OrderedDict()
    e = "embeddings."
    s[e + "cls_token"] = (1, 1, H)
    s[e + "position_embeddings"] = (1, npatch + 1, H)
    s[e + "text_embeddings.word_embeddings.weight"] = (cfg["vocab"], H)
    s[e + "text_embeddings.position_embeddings.weight"] = (cfg["max_text"], H)
    s[e + "text_embeddings.token_type_embeddings.weight"] = (cfg["type_vocab"], H)
    s[e + "text_embeddings.LayerNorm.weight"] = (H,)
    s[e + "text_embeddings.LayerNorm.bias"] = (H,)
    s[e + "patch_embeddings.projection.weight"] = (H, cfg["channels"], cfg["patch"], cfg["patch"])
    s[e + "patch_embeddings.projection.bias"] = (H,)
    s[e + "token_type_embeddings.weight"] = (modality_rows, H)
    for i in range(cfg["layers"]):
        l = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[l + f"attention.attention.{n}.weight"] = (H, H)
            s[l + f"attention.attention.{n}.bias"] = (H,)
        s[l + "attention.output.dense.weight"] = (H, H)
        s[l + "attention.output.dense.bias"] = (H,)
        s[l + "intermediate.dense.weight"] = (Fd, H)
        s[l + "intermediate.dense.bias"] = (Fd,)
        s[l + "output.dense.weight"] = (H, Fd)
        s[l + "output.dense.bias"] = (H,)
        s[l + "layernorm_before.weight"] = (H,)
        s[l + "layernorm_before.bias"] = (H,)
        s[l + "layernorm_after.weight"] = (H,)
        s[l + "layernorm_after.bias"] = (H,)
    s["layernorm.weight"] = (H,)
    s["layernorm.bias"] = (H,)
    s["pooler.dense.weight"] = (H, H)
    s["pooler.dense.bias"] = (H,)
    return s


def head_param_shapes(task_key: str, task_cfg: dict, H: int = 768) -> "OrderedDict[str, tuple]":
    """REF/modeling/vilt.py:179-203, names relative to `task_layer.<task>.`"""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    if task_cfg["model_type"] == "classification":
        s["0.weight"] = (2 * H, H * task_cfg["num_images"])
        s["0.bias"] = (2 * H,)
        s["1.weight"] = (2 * H,)
        s["1.bias"] = (2 * H,)
        s["3.weight"] = (task_cfg["num_labels"], 2 * H)
        s["3.bias"] = (task_cfg["num_labels"],)
    else:
        s["1.weight"] = (1, H)
        s["1.bias"] = (1,)
    return s


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


ADAPTER_SITES = ("attention.output", "output")     # Houlsby: after the attention out-projection and after the MLP down-projection


def adapter_param_shapes(layer: int, task: str, r: int, H: int = 768) -> "OrderedDict[str, tuple]":
    """adapter-transformers v3 naming (`<site>.adapters.<name>.adapter_down.0.{weight,bias}`, `.adapter_up.{weight,bias}`),
    relative to the ViltModel.  Parity UNPINNED: the GLAMOR fork is absent (REF/.gitmodules:1-3)."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    for site in ADAPTER_SITES:
        a = f"encoder.layer.{layer}.{site}.adapters.{task}."
        s[a + "adapter_down.0.weight"] = (r, H)
        s[a + "adapter_down.0.bias"] = (r,)
        s[a + "adapter_up.weight"] = (H, r)
        s[a + "adapter_up.bias"] = (H,)
    return s


def physical_encoder_order(names: List[str], layers: int) -> List[str]:
    """Forward order with q/k/v weights adjacent and q/k/v biases adjacent inside each layer; a layer's adapters follow it."""
    out = [n for n in names if n.startswith("embeddings.")]
    for i in range(layers):
        l = f"encoder.layer.{i}."
        out += [l + "layernorm_before.weight", l + "layernorm_before.bias"]
        out += [l + f"attention.attention.{n}.weight" for n in ("query", "key", "value")]
        out += [l + f"attention.attention.{n}.bias" for n in ("query", "key", "value")]
        out += [l + "attention.output.dense.weight", l + "attention.output.dense.bias",
                l + "layernorm_after.weight", l + "layernorm_after.bias",
                l + "intermediate.dense.weight", l + "intermediate.dense.bias",
                l + "output.dense.weight", l + "output.dense.bias"]
        out += [n for n in names if n.startswith(l) and ".adapters." in n]
    out += ["layernorm.weight", "layernorm.bias", "pooler.dense.weight", "pooler.dense.bias"]
    assert sorted(out) == sorted(names)
    return out


class FlatLayout:
    """offset (in elements) / shape of every learner parameter inside the flat buffer."""

    def __init__(self, tasks: List[str], task_cfgs: Dict[str, dict], cfg: dict = VILT_CFG, modality_rows: int = None,
                 adapters: Dict[str, int] = None):
        self.cfg = cfg
        self.tasks = list(tasks)
        if modality_rows is None:
            modality_rows = 3 if "nlvr2" in tasks else 2   # REF/modeling/vilt.py:176-177
        self.modality_rows = modality_rows
        self.adapters = dict(adapters or {})               # adapter name (task key) -> bottleneck width
        enc = encoder_param_shapes(modality_rows, cfg)
        for t, r in self.adapters.items():
            for i in range(cfg["layers"]):
                enc.update(adapter_param_shapes(i, t, r, cfg["hidden"]))
        self.shapes: "OrderedDict[str, tuple]" = OrderedDict((ENC + n, s) for n, s in enc.items())
        for t in tasks:
            for n, s in head_param_shapes(t, task_cfgs[t], cfg["hidden"]).items():
                self.shapes[f"task_layer.{t}.{n}"] = s
        phys = [ENC + n for n in physical_encoder_order(list(enc.keys()), cfg["layers"])]
        phys += [n for n in self.shapes if not n.startswith(ENC)]
        self.physical: List[str] = phys
        self.offset: Dict[str, int] = {}
        off = 0
        for n in phys:
            self.offset[n] = off
            off += (_numel(self.shapes[n]) + ALIGN - 1) // ALIGN * ALIGN
            if n == ENC + "pooler.dense.bias":
                self.encoder_end = off
        self.total = off
        starts = [self.offset[f"{ENC}encoder.layer.{i}.layernorm_before.weight"] for i in range(cfg["layers"])]
        top = self.offset[ENC + "layernorm.weight"]
        self.embed_range = (0, starts[0])
        self.layer_range: List[Tuple[int, int]] = [(starts[i], starts[i + 1] if i + 1 < len(starts) else top) for i in range(len(starts))]
        self.top_range = (top, self.encoder_end)                          # final norm + pooler
        self.head_range = {}
        for t in tasks:
            ns = [n for n in phys if n.startswith(f"task_layer.{t}.")]
            lo = self.offset[ns[0]]
            hi = self.offset[ns[-1]] + (_numel(self.shapes[ns[-1]]) + ALIGN - 1) // ALIGN * ALIGN
            self.head_range[t] = (lo, hi)

    def numel(self, name: str) -> int:
        return _numel(self.shapes[name])

    def segments(self):
        """(name, start, padded_len) in physical order."""
        out = []
        for i, n in enumerate(self.physical):
            nxt = self.offset[self.physical[i + 1]] if i + 1 < len(self.physical) else self.total
            out.append((n, self.offset[n], nxt - self.offset[n]))
        return out


def no_decay(name: str) -> bool:
    """REF/modeling/vilt.py:209-213 substring grouping (only `...text_embeddings.LayerNorm.weight` among the norm
    gains is exempt; cls_token / position_embeddings / all other gains ARE decayed -- a reference quirk we keep)."""
    return any(nd in name for nd in ("bias", "LayerNorm.weight"))
