"""CPU oracle for the ViLT continual-fine-tuning step.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 *restatement* of the arithmetic on CLiMB's ViLT
training hot path.  It exists so that the HIP path can be checked against it; it is
never imported by the product package `climb_amd` (only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it).

Parity pin: the reference repository has no tests and no golden vectors of its own
(SURVEY.md §0 fact 5), so this oracle is pinned against outputs of the reference
itself, generated in the build container by `oracle/gen_golden.py` (which imports
`/root/reference/src` plus the installed `transformers==5.15.0` ViltModel the
reference delegates to) and committed under `tests/golden/`.  Adapter arithmetic is
the one exception: it lives in an un-vendored fork that is absent from the
reference tree, so `adapter_*` below says "parity unpinned".

Citations: REF = /root/reference/src, HF = transformers/models/vilt/modeling_vilt.py
(transformers 5.15.0, the third-party module the reference imports at
REF/modeling/vilt.py:17).

Everything is functional: parameters live in a dict keyed by the reference's own
`ViltContinualLearner.named_parameters()` names (`vilt_encoder.vilt.*`,
`task_layer.<task>.<idx>.*`).
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

ENC = "vilt_encoder.vilt."

# ViltConfig() defaults (HF/configuration_vilt.py; verified by import)
CFG = dict(hidden=768, heads=12, head_dim=64, ffn=3072, layers=12, vocab=30522,
           max_text=40, type_vocab=2, patch=32, image=384, channels=3,
           ln_eps=1e-12, head_ln_eps=1e-5)

# REF/configs/task_configs.py:16-95 (only what the arithmetic needs)
TASKS = {
    "vqa":     dict(model_type="classification", num_labels=3129, num_images=1, lr=1e-4),
    "nlvr2":   dict(model_type="classification", num_labels=2, num_images=2, lr=1e-4),
    "snli-ve": dict(model_type="classification", num_labels=3, num_images=1, lr=5e-5),
    "vcr":     dict(model_type="multi-choice", num_labels=4, num_choices=4, num_images=1, lr=1e-4),
}


# --------------------------------------------------------------------------- params
def param_shapes(tasks: List[str], cfg: dict = CFG) -> "OrderedDict[str, tuple]":
    """Names/shapes in `ViltContinualLearner.named_parameters()` order.

    HF ViltModel registration order (embeddings, encoder.layer.*, layernorm, pooler)
    then REF/modeling/vilt.py:171-174 task heads; the modality table has 3 rows when
    'nlvr2' is among the tasks (REF/modeling/vilt.py:176-177, :98-109)."""
    H, Fd = cfg["hidden"], cfg["ffn"]
    n_patch = (cfg["image"] // cfg["patch"]) ** 2
    s: "OrderedDict[str, tuple]" = OrderedDict()
    e = ENC + "embeddings."
    s[e + "cls_token"] = (1, 1, H)
    s[e + "position_embeddings"] = (1, n_patch + 1, H)
    s[e + "text_embeddings.word_embeddings.weight"] = (cfg["vocab"], H)
    s[e + "text_embeddings.position_embeddings.weight"] = (cfg["max_text"], H)
    s[e + "text_embeddings.token_type_embeddings.weight"] = (cfg["type_vocab"], H)
    s[e + "text_embeddings.LayerNorm.weight"] = (H,)
    s[e + "text_embeddings.LayerNorm.bias"] = (H,)
    s[e + "patch_embeddings.projection.weight"] = (H, cfg["channels"], cfg["patch"], cfg["patch"])
    s[e + "patch_embeddings.projection.bias"] = (H,)
    s[e + "token_type_embeddings.weight"] = (3 if "nlvr2" in tasks else 2, H)
    for i in range(cfg["layers"]):
        l = f"{ENC}encoder.layer.{i}."
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
    s[ENC + "layernorm.weight"] = (H,)
    s[ENC + "layernorm.bias"] = (H,)
    s[ENC + "pooler.dense.weight"] = (H, H)
    s[ENC + "pooler.dense.bias"] = (H,)
    for t in tasks:
        tc = TASKS[t]
        h = f"task_layer.{t}."
        if tc["model_type"] == "classification":      # REF/modeling/vilt.py:188-196
            s[h + "0.weight"] = (2 * H, H * tc["num_images"])
            s[h + "0.bias"] = (2 * H,)
            s[h + "1.weight"] = (2 * H,)
            s[h + "1.bias"] = (2 * H,)
            s[h + "3.weight"] = (tc["num_labels"], 2 * H)
            s[h + "3.bias"] = (tc["num_labels"],)
        else:                                          # REF/modeling/vilt.py:198-203
            s[h + "1.weight"] = (1, H)
            s[h + "1.bias"] = (1,)
    return s


def _is_norm_gain(name: str) -> bool:
    return (name.endswith("LayerNorm.weight") or name.endswith("layernorm.weight")
            or name.endswith("layernorm_before.weight") or name.endswith("layernorm_after.weight")
            or (name.startswith("task_layer.") and name.endswith(".1.weight") and "vcr" not in name))


def seeded_tensor(name: str, shape: tuple, seed: int) -> np.ndarray:
    """Deterministic per-name values from numpy's PCG64 (version-stable), so the GPU
    box can regenerate exactly the weights the golden fixtures were made with."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    x = rng.standard_normal(shape, dtype=np.float32)
    if _is_norm_gain(name):
        return (1.0 + 0.1 * x).astype(np.float32)
    if name.endswith(".bias"):
        return (0.02 * x).astype(np.float32)
    if ".encoder.layer." in name or "pooler" in name or name.startswith("task_layer."):
        return (0.04 * x).astype(np.float32)
    return (0.02 * x).astype(np.float32)


def init_params(tasks: List[str], seed: int = 42, cfg: dict = CFG) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((n, torch.from_numpy(seeded_tensor(n, shp, seed)))
                       for n, shp in param_shapes(tasks, cfg).items())


def synthetic_encodings(batch: int, seed: int = 1, text_len: int = 40, image: int = 384,
                        cfg: dict = CFG, ragged_text: bool = False) -> Dict[str, torch.Tensor]:
    """Synthetic processor output with the layout of REF/modeling/vilt.py:83-96
    (SURVEY.md §8(d) input spec): ids uniform in [0, vocab), N(0,1) pixels, all-ones masks."""
    rng = np.random.default_rng([seed, 7])
    ids = rng.integers(0, cfg["vocab"], size=(batch, text_len), dtype=np.int64)
    am = np.ones((batch, text_len), dtype=np.int64)
    if ragged_text:
        lens = rng.integers(3, text_len + 1, size=(batch,))
        lens[0] = text_len
        for b in range(batch):
            am[b, lens[b]:] = 0
            ids[b, lens[b]:] = 0
    px = rng.standard_normal((batch, cfg["channels"], image, image), dtype=np.float32)
    return dict(input_ids=torch.from_numpy(ids),
                token_type_ids=torch.zeros(batch, text_len, dtype=torch.long),
                attention_mask=torch.from_numpy(am),
                pixel_values=torch.from_numpy(px),
                pixel_mask=torch.ones(batch, image, image, dtype=torch.long))


def synthetic_varres_encodings(sizes, seed: int = 1, text_len: int = 40, cfg: dict = CFG) -> Dict[str, torch.Tensor]:
    """Padded variable-resolution batch as `ViltProcessor` produces it (HF image_processing: both sides multiples of 32, images
    placed top-left on a canvas of the batch maximum, zeros elsewhere, `pixel_mask` = 1 on real pixels); ragged text lengths."""
    rng = np.random.default_rng([seed, 17])
    B = len(sizes)
    Hc, Wc = max(h for h, _ in sizes), max(w for _, w in sizes)
    px = np.zeros((B, cfg["channels"], Hc, Wc), dtype=np.float32)
    pm = np.zeros((B, Hc, Wc), dtype=np.int64)
    for b, (h, w) in enumerate(sizes):
        px[b, :, :h, :w] = rng.standard_normal((cfg["channels"], h, w), dtype=np.float32)
        pm[b, :h, :w] = 1
    ids = rng.integers(0, cfg["vocab"], size=(B, text_len), dtype=np.int64)
    am = np.ones((B, text_len), dtype=np.int64)
    lens = rng.integers(3, text_len + 1, size=(B,))
    lens[0] = text_len
    for b in range(B):
        am[b, lens[b]:] = 0
        ids[b, lens[b]:] = 0
    return dict(input_ids=torch.from_numpy(ids), token_type_ids=torch.zeros(B, text_len, dtype=torch.long), attention_mask=torch.from_numpy(am),
                pixel_values=torch.from_numpy(px), pixel_mask=torch.from_numpy(pm))


def synthetic_vqa_targets(batch: int, seed: int = 1, num_labels: int = 3129) -> torch.Tensor:
    """Soft VQA scores in {0,.3,.6,.9,1} (REF/utils/vqa_utils.py:10-20,48-53): up to three answers per row."""
    rng = np.random.default_rng([seed, 11])
    t = np.zeros((batch, num_labels), dtype=np.float32)
    vals = np.array([1.0, 0.6, 0.3], dtype=np.float32)
    for b in range(batch):
        k = int(rng.integers(1, 4))
        idx = rng.choice(num_labels, size=k, replace=False)
        t[b, idx] = vals[:k]
    return torch.from_numpy(t)


# ---------------------------------------------------------------------------- model
def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def gelu(x):
    """Exact erf GELU (ACT2FN['gelu'], HF:393; nn.GELU() at REF/modeling/vilt.py:193)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def text_embed(P, input_ids, token_type_ids, cfg=CFG, inputs_embeds=None):
    """HF:237-269 TextEmbeddings.forward: LN(word[ids] + type[tt] + pos[0:T]); dropout p=0.  `inputs_embeds` [B,T,H] replaces the
    table lookup (HF:250-251; ViLT-BERT feeds BERT's last hidden state here, REF/modeling/viltbert.py:142-147)."""
    e = ENC + "embeddings.text_embeddings."
    T = token_type_ids.shape[1]
    x = P[e + "word_embeddings.weight"][input_ids] if inputs_embeds is None else inputs_embeds
    x = x + P[e + "token_type_embeddings.weight"][token_type_ids]
    x = x + P[e + "position_embeddings.weight"][:T].unsqueeze(0)
    return layer_norm(x, P[e + "LayerNorm.weight"], P[e + "LayerNorm.bias"], cfg["ln_eps"])


def visual_embed_fixed(P, pixel_values, pixel_mask, cfg=CFG):
    """HF:92-178 visual_embed restricted to full-size unmasked images (the benchmark
    shape): every sample has h = w = image/patch, the bilinear position-embedding
    resize (HF:103-118) is the identity, and the `torch.multinomial` patch selection
    (HF:153-159) is a pure permutation of all patches.  Self-attention followed by
    pooling of the text [CLS] row is invariant to the order of the patch rows, so
    this restatement keeps raster order (SURVEY.md §8(a) A5 measured the difference
    from the reference's random order at ~1e-6 relative: fp reassociation only)."""
    e = ENC + "embeddings."
    B, C, Hh, Ww = pixel_values.shape
    p = cfg["patch"]
    assert Hh == cfg["image"] and Ww == cfg["image"], "oracle covers the fixed-resolution path only (row F2 is next)"
    assert bool((pixel_mask == 1).all()), "oracle covers all-ones pixel masks only (row F2 is next)"
    # HF:292-300 Conv2d(3,768,k=32,s=32) == im2col GEMM
    x = F.conv2d(pixel_values, P[e + "patch_embeddings.projection.weight"],
                 P[e + "patch_embeddings.projection.bias"], stride=p)
    x = x.flatten(2).transpose(1, 2)                                   # [B, P, H] raster order
    pos = P[e + "position_embeddings"]                                 # [1, 1+P, H]
    x = x + pos[:, 1:, :]
    cls = P[e + "cls_token"].expand(B, -1, -1) + pos[:, :1, :]        # HF:168-173
    x = torch.cat([cls, x], dim=1)
    mask = torch.ones(B, x.shape[1], dtype=torch.long)
    return x, mask


def _adapt(P, prefix, y):
    return adapter_forward(y, P[prefix + "adapter_down.0.weight"], P[prefix + "adapter_down.0.bias"], P[prefix + "adapter_up.weight"],
                           P[prefix + "adapter_up.bias"])


def visual_embed_general(P, pixel_values, pixel_mask, cfg=CFG):
    """HF:92-178 visual_embed for padded, variable-resolution batches, restated deterministically.

    The reference down-samples `pixel_mask` to the patch grid (HF:96-97), reads each sample's valid extent (h, w) off the first
    column / row (HF:98-99), resizes the 12x12 position table to (h, w) bilinearly with align_corners=True and zero-pads it
    to the canvas (HF:103-118), then keeps `max_b(h*w)` patches per sample: all valid ones in a RANDOM order plus, for smaller
    images, randomly chosen invalid (masked) patches (HF:131-166).  Self-attention with masked keys followed by pooling of
    the text [CLS] row does not depend on the order of patch rows, nor on the content or number of masked rows, so this
    restatement keeps EVERY canvas patch in raster order, zeroes the invalid ones and masks them: same pooled output up
    to fp32 reassociation, no RNG, and no host synchronisation to size the sequence."""
    e = ENC + "embeddings."
    B, C, Hh, Ww = pixel_values.shape
    p = cfg["patch"]
    gh, gw = Hh // p, Ww // p
    g0 = cfg["image"] // p
    x = F.conv2d(pixel_values, P[e + "patch_embeddings.projection.weight"], P[e + "patch_embeddings.projection.bias"], stride=p)
    m = pixel_mask[:, ::p, ::p][:, :gh, :gw]                       # nearest down-sampling picks the top-left pixel of each patch
    hb = m[:, :, 0].sum(dim=1)
    wb = m[:, 0, :].sum(dim=1)
    pos_tab = P[e + "position_embeddings"]
    spatial = pos_tab[:, 1:, :].transpose(1, 2).reshape(1, -1, g0, g0)
    pos = torch.zeros(B, x.shape[1], gh, gw, dtype=x.dtype)
    valid = torch.zeros(B, gh, gw, dtype=torch.long)
    for b in range(B):
        h, w = int(hb[b]), int(wb[b])
        pos[b, :, :h, :w] = F.interpolate(spatial, size=(h, w), mode="bilinear", align_corners=True)[0]
        valid[b, :h, :w] = 1
    x = (x + pos) * valid[:, None].to(x.dtype)                     # invalid canvas patches: zero rows, masked below
    x = x.flatten(2).transpose(1, 2)
    cls = P[e + "cls_token"].expand(B, -1, -1) + pos_tab[:, :1, :]
    x = torch.cat([cls, x], dim=1)
    mask = torch.cat([torch.ones(B, 1, dtype=torch.long), valid.flatten(1)], dim=1)
    return x, mask


def encoder_layer(P, i, x, key_bias, cfg=CFG, adapter=None):
    """HF:430-451 ViltLayer (pre-LN): h1 = x + Wo.Attn(LN_b(x)); y = h1 + W2.GELU(W1.LN_a(h1)).
    `adapter` = name of a Houlsby adapter applied to both sub-layer outputs before their residual adds (UNPINNED)."""
    l = f"{ENC}encoder.layer.{i}."
    B, S, H = x.shape
    nh, hd = cfg["heads"], cfg["head_dim"]
    xn = layer_norm(x, P[l + "layernorm_before.weight"], P[l + "layernorm_before.bias"], cfg["ln_eps"])
    # HF:322-351 ViltSelfAttention
    q = F.linear(xn, P[l + "attention.attention.query.weight"], P[l + "attention.attention.query.bias"])
    k = F.linear(xn, P[l + "attention.attention.key.weight"], P[l + "attention.attention.key.bias"])
    v = F.linear(xn, P[l + "attention.attention.value.weight"], P[l + "attention.attention.value.bias"])
    q = q.view(B, S, nh, hd).transpose(1, 2)
    k = k.view(B, S, nh, hd).transpose(1, 2)
    v = v.view(B, S, nh, hd).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
    scores = scores + key_bias[:, None, None, :]
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, S, H)
    attn = F.linear(ctx, P[l + "attention.output.dense.weight"], P[l + "attention.output.dense.bias"])
    if adapter is not None:
        attn = _adapt(P, f"{l}attention.output.adapters.{adapter}.", attn)
    h1 = attn + x                                                      # HF:440
    hn = layer_norm(h1, P[l + "layernorm_after.weight"], P[l + "layernorm_after.bias"], cfg["ln_eps"])
    a = gelu(F.linear(hn, P[l + "intermediate.dense.weight"], P[l + "intermediate.dense.bias"]))
    out = F.linear(a, P[l + "output.dense.weight"], P[l + "output.dense.bias"])
    if adapter is not None:
        out = _adapt(P, f"{l}output.adapters.{adapter}.", out)
    return out + h1                                                    # HF:410-414


def encoder_forward(P, enc: Dict[str, torch.Tensor], image_token_type_idx: int = 1, cfg=CFG,
                    return_sequence: bool = False, adapter=None):
    """REF/modeling/vilt.py:111-124 -> HF:536-647 ViltModel.forward -> pooler_output [B,768]."""
    e = ENC + "embeddings."
    text = text_embed(P, enc.get("input_ids"), enc["token_type_ids"], cfg, enc.get("inputs_embeds"))
    pv, pm = enc["pixel_values"], enc["pixel_mask"]
    if pv.shape[-2:] == (cfg["image"], cfg["image"]) and bool((pm == 1).all()):
        img, img_mask = visual_embed_fixed(P, pv, pm, cfg)
        tt = P[e + "token_type_embeddings.weight"]
        img = img + tt[image_token_type_idx]                           # HF:211-213
    else:
        img, img_mask = visual_embed_general(P, pv, pm, cfg)
        tt = P[e + "token_type_embeddings.weight"]
        img = img + tt[image_token_type_idx] * img_mask[..., None].to(img.dtype)      # masked filler rows stay zero
    text = text + tt[0]                                                # HF:208-210
    x = torch.cat([text, img], dim=1)                                  # HF:216
    mask = torch.cat([enc["attention_mask"], img_mask], dim=1)         # HF:217
    # HF:623-627 create_bidirectional_mask: additive, finfo.min where masked
    key_bias = torch.zeros(mask.shape, dtype=x.dtype)
    key_bias = key_bias.masked_fill(mask == 0, torch.finfo(x.dtype).min)
    for i in range(cfg["layers"]):
        x = encoder_layer(P, i, x, key_bias, cfg, adapter)
    seq = layer_norm(x, P[ENC + "layernorm.weight"], P[ENC + "layernorm.bias"], cfg["ln_eps"])  # HF:637
    pooled = torch.tanh(F.linear(seq[:, 0], P[ENC + "pooler.dense.weight"], P[ENC + "pooler.dense.bias"]))  # HF:657-663
    return (pooled, seq) if return_sequence else pooled


def head_forward(P, task_key: str, pooled, training: bool = False, dropout_keep: Optional[torch.Tensor] = None,
                 cfg=CFG):
    """REF/modeling/vilt.py:179-203 task heads."""
    h = f"task_layer.{task_key}."
    tc = TASKS[task_key]
    if tc["model_type"] == "classification":
        x = F.linear(pooled, P[h + "0.weight"], P[h + "0.bias"])
        x = layer_norm(x, P[h + "1.weight"], P[h + "1.bias"], cfg["head_ln_eps"])
        x = gelu(x)
        return F.linear(x, P[h + "3.weight"], P[h + "3.bias"])
    # multi-choice: Dropout(0.1) -> Linear(768,1) -> squeeze (REF/modeling/vilt.py:198-203, :349)
    x = pooled
    if training:
        assert dropout_keep is not None, "pass the keep mask explicitly (the only stochastic op on the path)"
        x = x * dropout_keep / 0.9
    return F.linear(x, P[h + "1.weight"], P[h + "1.bias"]).squeeze(-1)


def learner_forward(P, task_key: str, enc: Dict[str, torch.Tensor], training: bool = False,
                    dropout_keep: Optional[torch.Tensor] = None, cfg=CFG, adapter=None):
    """REF/modeling/vilt.py:218-350: single image / multi-image (NLVR2) / multi-choice (VCR).

    `enc` is what `process_inputs` returns for the flattened lists:
      nlvr2: texts [b], images flattened [b*2] (image j of example i at row 2*i+j, REF:281,:288);
      vcr:   texts flattened [b*4] (choice j of example i at row 4*i+j, REF:331,:334), images [b]."""
    tc = TASKS[task_key]
    if tc["model_type"] == "classification" and tc["num_images"] == 1:
        pooled = encoder_forward(P, enc, 1, cfg, adapter=adapter)
        return pooled, head_forward(P, task_key, pooled, cfg=cfg)
    if tc["model_type"] == "classification":
        n = tc["num_images"]
        bs = enc["input_ids"].shape[0]
        pv = enc["pixel_values"].view(bs, n, *enc["pixel_values"].shape[-3:])
        pm = enc["pixel_mask"].view(bs, n, *enc["pixel_mask"].shape[-2:])
        outs = []
        for i in range(n):                                             # REF:292-303
            e = dict(input_ids=enc["input_ids"], token_type_ids=enc["token_type_ids"],
                     attention_mask=enc["attention_mask"], pixel_values=pv[:, i], pixel_mask=pm[:, i])
            outs.append(encoder_forward(P, e, i + 1, cfg, adapter=adapter))
        pooled = torch.cat(outs, dim=-1)                               # REF:304
        return pooled, head_forward(P, task_key, pooled, cfg=cfg)
    nc = tc["num_choices"]
    bs = enc["pixel_values"].shape[0]
    ids = enc["input_ids"].view(bs, nc, -1)
    am = enc["attention_mask"].view(bs, nc, -1)
    tt = enc["token_type_ids"].view(bs, nc, -1)
    outs = []
    for i in range(nc):                                                # REF:335-345
        e = dict(input_ids=ids[:, i], token_type_ids=tt[:, i], attention_mask=am[:, i],
                 pixel_values=enc["pixel_values"], pixel_mask=enc["pixel_mask"])
        outs.append(encoder_forward(P, e, 1, cfg, adapter=adapter))
    pooled = torch.stack(outs, dim=0).transpose(0, 1)                  # REF:347  [b, nc, H]
    return pooled, head_forward(P, task_key, pooled, training, dropout_keep, cfg)


# --------------------------------------------------------------------------- losses
def vqa_loss(logits, target):
    """REF/train/visionlanguage_tasks/train_vqa.py:95,:157: BCEWithLogits(mean) * num_labels."""
    return F.binary_cross_entropy_with_logits(logits, target, reduction="mean") * target.shape[1]


def ce_loss(logits, labels):
    """REF/train/visionlanguage_tasks/train_nlvr2.py:80 (SNLI-VE, VCR identical)."""
    return F.cross_entropy(logits, labels)


def task_loss(task_key, logits, target):
    return vqa_loss(logits, target) if task_key == "vqa" else ce_loss(logits, target)


def vqa_score(logits, target):
    """REF/train/visionlanguage_tasks/train_vqa.py:99-113, :258-263."""
    idx = logits.argmax(dim=1)
    return target.gather(1, idx[:, None]).squeeze(1)


# ------------------------------------------------------------------------ optimiser
def no_decay(name: str) -> bool:
    """REF/modeling/vilt.py:209-213: substring match on ['bias', 'LayerNorm.weight']."""
    return any(nd in name for nd in ("bias", "LayerNorm.weight"))


def poly_lr(step: int, base_lr: float, warmup: int, total: int) -> float:
    """transformers.get_polynomial_decay_schedule_with_warmup(lr_end=0, power=1) as called at
    REF/train/visionlanguage_tasks/train_vqa.py:199-205; `step` = number of scheduler.step() calls so far."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    if step > total:
        return 0.0
    return base_lr * (1.0 - (step - warmup) / (total - warmup))


def adamw_step(P, G, state, lr, wd=1e-2, eps=1e-8, betas=(0.9, 0.98), names=None):
    """torch.optim.AdamW (decoupled decay) with the reference's grouping/betas
    (REF/modeling/vilt.py:205-215).  In-place on P; `state` maps name -> (m, v, t)."""
    b1, b2 = betas
    for n in (names if names is not None else P.keys()):
        g = G.get(n)
        if g is None:
            continue
        m, v, t = state.get(n, (torch.zeros_like(P[n]), torch.zeros_like(P[n]), 0))
        t += 1
        w = 0.0 if no_decay(n) else wd
        P[n].mul_(1.0 - lr * w)
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        bc1 = 1.0 - b1 ** t
        bc2 = 1.0 - b2 ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        P[n].addcdiv_(m, denom, value=-lr / bc1)
        state[n] = (m, v, t)


# ------------------------------------------------------------------------------ EWC
def encoder_names(P) -> List[str]:
    return [n for n in P if n.startswith(ENC)]


def ewc_loss(P, fisher: Dict[str, torch.Tensor], theta_star: Dict[str, torch.Tensor], lam: float):
    """REF/cl_algorithms/ewc.py:75-87: lam * sum_n sum_i F_i (theta_i - theta*_i)^2, encoder params only
    (dict keys are the encoder-relative names `vilt.*`), no 1/2 factor."""
    tot = 0.0
    for n in encoder_names(P):
        k = n[len("vilt_encoder."):]
        if k in fisher:
            tot = tot + (fisher[k] * (P[n] - theta_star[k]).pow(2)).sum()
    return lam * tot


def fisher_from_batch_grads(batch_grads: List[Dict[str, torch.Tensor]], samples_per_batch: List[int]):
    """REF/cl_algorithms/ewc.py:56-71 including its quirk: `.grad` is never zeroed between
    batches, so batch k contributes (sum_{j<=k} g_j)^2; divided by #samples at the end."""
    acc: Dict[str, torch.Tensor] = {}
    fisher: Dict[str, torch.Tensor] = {}
    for g in batch_grads:
        for n, t in g.items():
            acc[n] = acc[n] + t if n in acc else t.clone()
            fisher[n] = fisher.get(n, 0.0) + acc[n].pow(2)
    tot = float(sum(samples_per_batch))
    return {n: f / tot for n, f in fisher.items()}


# -------------------------------------------------------------------- adapters (unpinned)
def adapter_forward(x, down_w, down_b, up_w, up_b):
    """Houlsby bottleneck, parity UNPINNED: the GLAMOR adapter-transformers fork that defines it is
    absent (REF/.gitmodules:1-3).  Public adapter-transformers v3 semantics:
    out = x + W_up . swish(W_down . x + b_down) + b_up."""
    return x + F.linear(F.silu(F.linear(x, down_w, down_b)), up_w, up_b)


# ----------------------------------------------------------------------- train step
def train_step(P, task_key, enc, target, opt_state=None, lr=None, ewc=None, wd=1e-2, eps=1e-8,
               dropout_keep=None, trainable=None, adapter=None):
    """REF/train/visionlanguage_tasks/train_vqa.py:135-174 (train_nlvr2.py:110-150 identical but for the loss).

    Returns (loss, (pooled, logits), ewc_loss, grads).  If `opt_state` is given an AdamW step follows
    (optimizer.step(); zero_grad()); the caller owns the lr schedule.
    `ewc` = (fisher, theta_star, lam) or None.  `trainable` = optional set of names with requires_grad."""
    names = list(P.keys()) if trainable is None else [n for n in P if n in trainable]
    leaves = {n: (P[n].detach().clone().requires_grad_(True) if n in names else P[n].detach()) for n in P}
    pooled, logits = learner_forward(leaves, task_key, enc, training=True, dropout_keep=dropout_keep, adapter=adapter)
    loss = task_loss(task_key, logits, target)
    el = None
    total = loss
    if ewc is not None:
        el = ewc_loss(leaves, *ewc)
        total = loss + el
    total.backward()
    G = {n: leaves[n].grad for n in names if leaves[n].grad is not None}
    if opt_state is not None:
        with torch.no_grad():
            adamw_step(P, G, opt_state, lr, wd=wd, eps=eps, names=names)
    return loss.detach(), (pooled.detach(), logits.detach()), (None if el is None else el.detach()), G
