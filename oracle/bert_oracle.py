"""CPU restatement of the frozen BERT text encoder ViLT-BERT puts in front of ViLT (row F4).  TEST INFRASTRUCTURE: only tests/,
__graft_entry__.smoke() and bench.py's checker legs may import this; the product path (climb_amd/bert.py) never does.

Follows transformers' `BertModel` (HFB = transformers/models/bert/modeling_bert.py, 5.15.0) as REF/modeling/viltbert.py:115-121 calls it:
`bert(input_ids, attention_mask, token_type_ids).last_hidden_state`, in EVAL mode (dropouts off).  Pinned by oracle/gen_golden.py::
case_viltbert against the reference's own ViltBertContinualLearner around a seeded `BertModel(BertConfig())`.

Reference quirk, reproduced since r03: `get_bert_outputs` runs BERT under no_grad but never puts it in eval mode, so while the learner is
in train mode BERT's 0.1 dropouts (embeddings, attention probabilities, both sub-layer outputs) perturb the "frozen" text features with
torch's RNG stream.  `bert_forward(..., masks=...)` applies given keep-masks at those 37 sites; oracle/gen_golden.py::case_viltbert_train
records the masks the reference itself drew (tests/golden/viltbert_vqa_b3_train.npz), which pins both this restatement and the HIP path."""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BERT_CFG = dict(hidden=768, heads=12, head_dim=64, ffn=3072, layers=12, vocab=30522, max_pos=512, type_vocab=2, ln_eps=1e-12)


def bert_param_shapes(cfg=BERT_CFG) -> "OrderedDict[str, tuple]":
    """`BertModel.state_dict()` names in registration order (HFB:53-75 embeddings, :354-416 layer, :566-592 pooler)."""
    H, Fd = cfg["hidden"], cfg["ffn"]
    s = OrderedDict()
    s["embeddings.word_embeddings.weight"] = (cfg["vocab"], H)
    s["embeddings.position_embeddings.weight"] = (cfg["max_pos"], H)
    s["embeddings.token_type_embeddings.weight"] = (cfg["type_vocab"], H)
    s["embeddings.LayerNorm.weight"] = (H,)
    s["embeddings.LayerNorm.bias"] = (H,)
    for i in range(cfg["layers"]):
        l = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[l + f"attention.self.{n}.weight"] = (H, H)
            s[l + f"attention.self.{n}.bias"] = (H,)
        s[l + "attention.output.dense.weight"] = (H, H)
        s[l + "attention.output.dense.bias"] = (H,)
        s[l + "attention.output.LayerNorm.weight"] = (H,)
        s[l + "attention.output.LayerNorm.bias"] = (H,)
        s[l + "intermediate.dense.weight"] = (Fd, H)
        s[l + "intermediate.dense.bias"] = (Fd,)
        s[l + "output.dense.weight"] = (H, Fd)
        s[l + "output.dense.bias"] = (H,)
        s[l + "output.LayerNorm.weight"] = (H,)
        s[l + "output.LayerNorm.bias"] = (H,)
    s["pooler.dense.weight"] = (H, H)
    s["pooler.dense.bias"] = (H,)
    return s


def init_bert_params(seed: int = 7, cfg=BERT_CFG) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic per-name values (numpy PCG64): the GPU box regenerates exactly the weights the fixture was made with."""
    out = OrderedDict()
    for n, shp in bert_param_shapes(cfg).items():
        rng = np.random.default_rng([seed, zlib.crc32(("bert." + n).encode())])
        x = rng.standard_normal(shp, dtype=np.float32)
        if n.endswith("LayerNorm.weight"):
            x = 1.0 + 0.1 * x
        elif n.endswith(".bias"):
            x = 0.02 * x
        elif ".encoder.layer." in "." + n:
            x = 0.04 * x
        else:
            x = 0.02 * x
        out[n] = torch.from_numpy(x.astype(np.float32))
    return out


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def bert_forward(PB, input_ids, token_type_ids, attention_mask, cfg=BERT_CFG, masks=None, p_drop: float = 0.1):
    """last_hidden_state [B, T, H] (HFB:594-700 BertModel.forward).  masks None = eval mode; else the train-mode dropouts with the given
    keep-masks {"emb": [B,T,H], "probs": L x [B,nh,T,T], "attn_out": L x [B,T,H], "ffn_out": L x [B,T,H]} (x * keep / (1 - p))."""
    def drop(x, m):
        return x if masks is None else x * m.to(x.dtype) / (1.0 - p_drop)
    B, T = input_ids.shape
    H, nh, dh = cfg["hidden"], cfg["heads"], cfg["head_dim"]
    e = "embeddings."
    x = PB[e + "word_embeddings.weight"][input_ids] + PB[e + "token_type_embeddings.weight"][token_type_ids] \
        + PB[e + "position_embeddings.weight"][:T].unsqueeze(0)                                   # HFB:97-116 (absolute positions 0..T-1)
    x = _ln(x, PB[e + "LayerNorm.weight"], PB[e + "LayerNorm.bias"], cfg["ln_eps"])
    if masks is not None:
        x = drop(x, masks["emb"])                                                                 # HFB:112-116
    bias = torch.zeros((B, 1, 1, T), dtype=x.dtype).masked_fill(attention_mask[:, None, None, :] == 0, torch.finfo(x.dtype).min)
    for i in range(cfg["layers"]):
        l = f"encoder.layer.{i}."
        q = F.linear(x, PB[l + "attention.self.query.weight"], PB[l + "attention.self.query.bias"]).view(B, T, nh, dh).transpose(1, 2)
        k = F.linear(x, PB[l + "attention.self.key.weight"], PB[l + "attention.self.key.bias"]).view(B, T, nh, dh).transpose(1, 2)
        v = F.linear(x, PB[l + "attention.self.value.weight"], PB[l + "attention.self.value.bias"]).view(B, T, nh, dh).transpose(1, 2)
        p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + bias, dim=-1)                 # HFB:139-220
        if masks is not None:
            p = drop(p, masks["probs"][i])                                                        # eager_attention_forward: softmax -> dropout -> P V
        ctx = (p @ v).transpose(1, 2).reshape(B, T, H)
        y = F.linear(ctx, PB[l + "attention.output.dense.weight"], PB[l + "attention.output.dense.bias"])
        if masks is not None:
            y = drop(y, masks["attn_out"][i])                                                     # HFB:289-293: dense -> dropout -> LayerNorm(. + input)
        h = _ln(y + x, PB[l + "attention.output.LayerNorm.weight"], PB[l + "attention.output.LayerNorm.bias"], cfg["ln_eps"])      # HFB:282-294 post-LN
        u = F.linear(h, PB[l + "intermediate.dense.weight"], PB[l + "intermediate.dense.bias"])
        a = 0.5 * u * (1.0 + torch.erf(u / math.sqrt(2.0)))                                       # HFB:325-338 'gelu'
        z = F.linear(a, PB[l + "output.dense.weight"], PB[l + "output.dense.bias"])
        if masks is not None:
            z = drop(z, masks["ffn_out"][i])                                                      # HFB:348-351
        x = _ln(z + h, PB[l + "output.LayerNorm.weight"], PB[l + "output.LayerNorm.bias"], cfg["ln_eps"])                          # HFB:340-352
    return x
