"""The CPU oracle (oracle/vilt_oracle.py) against the golden vectors the reference itself produced
(tests/golden/*.npz, written by oracle/gen_golden.py in the build container).  No GPU, no reference tree."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vilt_oracle as vo


def _meta(z):
    return dict(kv.split("=", 1) for kv in str(z["meta"][0]).split(";"))


def _summary(named, names, k=8):
    norms = np.array([float(named[n].double().norm()) for n in names])
    heads = np.zeros((len(names), k), dtype=np.float32)
    for i, n in enumerate(names):
        f = named[n].detach().reshape(-1)[:k].float().numpy()
        heads[i, :f.size] = f
    return norms, heads


def _close(a, b, rtol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b).max()
    assert err <= rtol * scale, f"{what}: max|d|={err:.3e} scale={scale:.3e}"


@pytest.mark.parametrize("fname", ["vqa_b2.npz", "vqa_b3_ragged.npz", "snlive_b2.npz"])
def test_single_image_forward_backward(golden_dir, fname):
    z = np.load(os.path.join(golden_dir, fname))
    m = _meta(z)
    tasks = m["tasks"].split(",")
    B = int(m["B"])
    P = vo.init_params(tasks, int(m["wseed"]))
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]), ragged_text=bool(int(m["ragged"])))
    if m["task"] == "vqa":
        target = vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))
    else:
        rng = np.random.default_rng([int(m["dseed"]), 13])
        target = torch.from_numpy(rng.integers(0, vo.TASKS[m["task"]]["num_labels"], size=(B,), dtype=np.int64))
    loss, (pooled, logits), _, G = vo.train_step(P, m["task"], enc, target)
    _close(pooled, z["pooled"], 2e-5, "pooled")
    _close(logits, z["logits"], 2e-5, "logits")
    _close(loss, z["loss"], 2e-5, "loss")
    assert np.array_equal(logits.argmax(-1).numpy(), z["logits"].argmax(-1))
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    _close(norms, z["grad_norms"], 1e-4, "grad norms")
    _close(heads, z["grad_heads"], 1e-4, "grad heads")


def test_nlvr2_two_images(golden_dir):
    z = np.load(os.path.join(golden_dir, "nlvr2_b2.npz"))
    m = _meta(z)
    b = int(m["b"])
    P = vo.init_params(m["tasks"].split(","), int(m["wseed"]))
    e1 = vo.synthetic_encodings(2 * b, seed=int(m["dseed"]))
    enc = dict(input_ids=e1["input_ids"][:b], token_type_ids=e1["token_type_ids"][:b],
               attention_mask=e1["attention_mask"][:b], pixel_values=e1["pixel_values"], pixel_mask=e1["pixel_mask"])
    loss, (pooled, logits), _, G = vo.train_step(P, "nlvr2", enc, torch.from_numpy(z["labels"]))
    _close(pooled, z["pooled"], 2e-5, "pooled")
    _close(logits, z["logits"], 2e-5, "logits")
    _close(loss, z["loss"], 2e-5, "loss")
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    _close(norms, z["grad_norms"], 1e-4, "grad norms")
    _close(heads, z["grad_heads"], 1e-4, "grad heads")


def test_vcr_four_choices_eval(golden_dir):
    z = np.load(os.path.join(golden_dir, "vcr_b2.npz"))
    m = _meta(z)
    b = int(m["b"])
    P = vo.init_params(m["tasks"].split(","), int(m["wseed"]))
    e1 = vo.synthetic_encodings(4 * b, seed=int(m["dseed"]), ragged_text=True)
    enc = dict(input_ids=e1["input_ids"], token_type_ids=e1["token_type_ids"], attention_mask=e1["attention_mask"],
               pixel_values=e1["pixel_values"][:b], pixel_mask=e1["pixel_mask"][:b])
    with torch.no_grad():
        pooled, logits = vo.learner_forward(P, "vcr", enc, training=False)
    _close(pooled, z["pooled"], 2e-5, "pooled")
    _close(logits, z["logits"], 2e-5, "logits")
    _close(vo.ce_loss(logits, torch.from_numpy(z["labels"])), z["loss"], 2e-5, "loss")


def test_vcr_four_choices_train_mode_with_the_references_dropout_mask(golden_dir):
    """The one stochastic op of the path (REF/modeling/vilt.py:199-202 Dropout(0.1) in the VCR head): the fixture carries the keep-mask the
    reference drew; fed the same mask the oracle reproduces logits, loss and every gradient (and the eval-mode result is far away)."""
    z = np.load(os.path.join(golden_dir, "vcr_b2_train.npz"))
    m = _meta(z)
    b = int(m["b"])
    P = vo.init_params(m["tasks"].split(","), int(m["wseed"]))
    e1 = vo.synthetic_encodings(4 * b, seed=int(m["dseed"]), ragged_text=True)
    enc = dict(input_ids=e1["input_ids"], token_type_ids=e1["token_type_ids"], attention_mask=e1["attention_mask"],
               pixel_values=e1["pixel_values"][:b], pixel_mask=e1["pixel_mask"][:b])
    keep = torch.from_numpy(np.unpackbits(z["keep"])[:b * 4 * 768].reshape(b, 4, 768).astype(np.float32))
    assert 0.85 < float(keep.mean()) < 0.95
    leaves = {n: P[n].clone().requires_grad_(True) for n in P}
    pooled, logits = vo.learner_forward(leaves, "vcr", enc, training=True, dropout_keep=keep)
    loss = vo.ce_loss(logits, torch.from_numpy(z["labels"]))
    loss.backward()
    _close(pooled.detach(), z["pooled"], 2e-5, "pooled")
    _close(logits.detach(), z["logits"], 2e-5, "logits")
    _close(loss.detach(), z["loss"], 2e-5, "loss")
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary({n: leaves[n].grad for n in names}, names)
    _close(norms, z["grad_norms"], 1e-4, "grad norms")
    _close(heads, z["grad_heads"], 1e-4, "grad heads")
    with torch.no_grad():
        _, ev = vo.learner_forward(P, "vcr", enc, training=False)
    assert float((ev - torch.from_numpy(z["logits"])).abs().max()) > 1e-2 * float(np.abs(z["logits"]).max())


def _ewc_state(P, seed=5):
    import zlib
    fisher, star = {}, {}
    for n in vo.encoder_names(P):
        k = n[len("vilt_encoder."):]
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])
        fisher[k] = torch.from_numpy((rng.random(P[n].shape, dtype=np.float32) * 1e-4).astype(np.float32))
        star[k] = P[n] + torch.from_numpy((0.01 * rng.standard_normal(P[n].shape, dtype=np.float32)).astype(np.float32))
    return fisher, star


def test_ewc_penalty(golden_dir):
    z = np.load(os.path.join(golden_dir, "ewc_b2.npz"))
    m = _meta(z)
    B = int(m["B"])
    P = vo.init_params(m["tasks"].split(","), int(m["wseed"]))
    fisher, star = _ewc_state(P, int(m["ewc_seed"]))
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]))
    target = vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))
    loss, _, el, G = vo.train_step(P, "vqa", enc, target, ewc=(fisher, star, float(m["lam"])))
    _close(el, z["ewc_loss"], 1e-5, "ewc loss")
    _close(loss, z["loss"], 2e-5, "loss")
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    _close(norms, z["grad_norms"], 1e-4, "grad norms")
    _close(heads, z["grad_heads"], 1e-4, "grad heads")


def test_fisher_accumulating_quirk(golden_dir):
    z = np.load(os.path.join(golden_dir, "fisher_3x2.npz"))
    m = _meta(z)
    B, nb = int(m["B"]), int(m["batches"])
    P = vo.init_params(m["tasks"].split(","), int(m["wseed"]))
    grads = []
    for i in range(nb):
        _, _, _, G = vo.train_step(P, "vqa", vo.synthetic_encodings(B, seed=200 + i), vo.synthetic_vqa_targets(B, seed=200 + i))
        grads.append({n[len("vilt_encoder."):]: g for n, g in G.items() if n.startswith(vo.ENC)})
    fisher = vo.fisher_from_batch_grads(grads, [B] * nb)
    names = [str(n) for n in z["names"]]
    norms, heads = _summary(fisher, names)
    _close(norms, z["fisher_norms"], 5e-4, "fisher norms")
    _close(heads, z["fisher_heads"], 5e-4, "fisher heads")


def test_replay_fresh_adamw(golden_dir):
    z = np.load(os.path.join(golden_dir, "replay_b2.npz"))
    m = _meta(z)
    B = int(m["B"])
    P = vo.init_params(m["tasks"].split(","), int(m["wseed"]))
    P0 = {n: t.clone() for n, t in P.items()}
    loss, _, _, _ = vo.train_step(P, "vqa", vo.synthetic_encodings(B, seed=int(m["dseed"])),
                                  vo.synthetic_vqa_targets(B, seed=int(m["dseed"])), opt_state={}, lr=float(m["lr"]))
    _close(loss, z["loss"], 2e-5, "loss")
    names = [str(n) for n in z["names"]]
    norms, _ = _summary({n: P[n] - P0[n] for n in names}, names)
    _close(norms, z["delta_norms"], 2e-3, "param delta norms")


def test_ten_steps_config1(golden_dir):
    """BASELINE.json configs[0]: ViLT sequential-FT on VQAv2, batch=2, 10 steps, CPU."""
    z = np.load(os.path.join(golden_dir, "vqa_b2_10steps.npz"))
    m = _meta(z)
    B, steps = int(m["B"]), int(m["steps"])
    P = vo.init_params(m["tasks"].split(","), int(m["wseed"]))
    P0 = {n: t.clone() for n, t in P.items()}
    state, losses = {}, []
    for s in range(steps):
        lr = vo.poly_lr(s, float(m["lr"]), int(steps * 0.1), steps)
        loss, _, _, _ = vo.train_step(P, "vqa", vo.synthetic_encodings(B, seed=100 + s),
                                      vo.synthetic_vqa_targets(B, seed=100 + s), opt_state=state, lr=lr)
        losses.append(loss.item())
    _close(losses, z["losses"], 1e-4, "loss curve")
    names = [str(n) for n in z["names"]]
    norms, _ = _summary({n: P[n] - P0[n] for n in names}, names)
    _close(norms, z["delta_norms"], 3e-3, "param delta norms")


def test_decay_grouping_quirk():
    """REF/modeling/vilt.py:209-213 substring grouping: only text_embeddings.LayerNorm.weight is exempt among gains."""
    names = list(vo.param_shapes(["vqa", "nlvr2"]).keys())
    nd = [n for n in names if vo.no_decay(n)]
    gains_exempt = [n for n in nd if not n.endswith("bias")]
    assert gains_exempt == [vo.ENC + "embeddings.text_embeddings.LayerNorm.weight"]
    assert not vo.no_decay(vo.ENC + "encoder.layer.0.layernorm_before.weight")
    assert not vo.no_decay(vo.ENC + "embeddings.cls_token")


def test_poly_schedule_matches_transformers():
    from transformers import get_polynomial_decay_schedule_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-4)
    sch = get_polynomial_decay_schedule_with_warmup(opt, num_warmup_steps=3, num_training_steps=30, lr_end=0, power=1)
    for s in range(32):
        assert abs(opt.param_groups[0]["lr"] - vo.poly_lr(s, 1e-4, 3, 30)) < 1e-12
        opt.step()
        sch.step()


def test_variable_resolution_general_visual_embed(golden_dir):
    """the deterministic restatement (all canvas patches in raster order, invalid ones zeroed + masked) against the
    reference's random masked patch selection with per-sample bilinear position resize."""
    z = np.load(os.path.join(golden_dir, "vqa_b4_varres.npz"))
    m = _meta(z)
    sizes = [tuple(int(v) for v in r) for r in z["sizes"]]
    P = vo.init_params(m["tasks"].split(","), int(m["wseed"]))
    enc = vo.synthetic_varres_encodings(sizes, seed=int(m["dseed"]))
    target = vo.synthetic_vqa_targets(len(sizes), seed=int(m["dseed"]))
    loss, (pooled, logits), _, G = vo.train_step(P, "vqa", enc, target)
    _close(pooled, z["pooled"], 2e-5, "pooled")
    _close(logits, z["logits"], 2e-5, "logits")
    _close(loss, z["loss"], 2e-5, "loss")
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    _close(norms, z["grad_norms"], 1e-4, "grad norms")
    _close(heads, z["grad_heads"], 1e-4, "grad heads")


def test_bert_oracle_equals_transformers_bert_and_fixture(golden_dir):
    """Row F4: oracle/bert_oracle.py against transformers' own BertModel (a dependency of the reference installed here and on the GPU box)
    on the seeded weights, and against the features the reference's ViltBertEncoderWrapper.get_bert_outputs produced (fixture)."""
    import numpy as np
    import torch
    transformers = pytest.importorskip("transformers")
    from oracle import bert_oracle as bo
    from oracle import vilt_oracle as vo
    z = np.load(os.path.join(golden_dir, "viltbert_vqa_b3.npz"))
    m = dict(kv.split("=", 1) for kv in str(z["meta"][0]).split(";"))
    PB = bo.init_bert_params(int(m["bseed"]))
    enc = vo.synthetic_encodings(int(m["B"]), seed=int(m["dseed"]), ragged_text=True)
    with torch.no_grad():
        feats = bo.bert_forward(PB, enc["input_ids"], enc["token_type_ids"], enc["attention_mask"])
        hf = transformers.BertModel(transformers.BertConfig()).eval()
        hf.load_state_dict({k: v for k, v in PB.items()}, strict=False)
        ref = hf(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"], token_type_ids=enc["token_type_ids"]).last_hidden_state
    assert float((feats - ref).abs().max() / ref.abs().max()) < 2e-5
    assert float((feats[:, :, :8] - torch.from_numpy(z["bert_feats_head"])).abs().max()) < 2e-5 * float(ref.abs().max())
    assert abs(float(feats.double().norm()) - float(z["bert_feats_norm"])) < 1e-4 * float(z["bert_feats_norm"])
    # the ViLT half on top of those features reproduces the reference's pooled output / logits / loss
    P = vo.init_params(m["tasks"].split(","), int(m["wseed"]))
    oenc = {k: v for k, v in enc.items() if k != "input_ids"}
    oenc["inputs_embeds"] = feats
    with torch.no_grad():
        pooled, logits = vo.learner_forward(P, "vqa", oenc, training=False)
    assert float((pooled - torch.from_numpy(z["pooled"])).abs().max()) < 2e-5
    assert float((logits - torch.from_numpy(z["logits"])).abs().max()) < 2e-5 * float(np.abs(z["logits"]).max())


def unpack_bert_masks(z, B, T, H=768, nh=12, L=12):
    """tests/golden/viltbert_vqa_b3_train.npz["masks"]: the 37 keep-masks the reference drew, bit-packed in the order
    emb | probs x L | attn_out x L | ffn_out x L (oracle/gen_golden.py::case_viltbert_train)."""
    bits = np.unpackbits(z["masks"])
    sizes = [B * T * H] + [B * nh * T * T] * L + [B * T * H] * (2 * L)
    out, at = [], 0
    for n in sizes:
        nb = (n + 7) // 8 * 8
        out.append(bits[at:at + n].astype(bool))
        at += nb
    assert at == bits.size
    t = lambda a, *shape: torch.from_numpy(a.reshape(shape))
    return {"emb": t(out[0], B, T, H), "probs": [t(out[1 + i], B, nh, T, T) for i in range(L)],
            "attn_out": [t(out[1 + L + i], B, T, H) for i in range(L)], "ffn_out": [t(out[1 + 2 * L + i], B, T, H) for i in range(L)]}


def test_viltbert_train_mode_with_the_references_bert_dropout_masks(golden_dir):
    """REF/modeling/viltbert.py:115-120 leaves its frozen BERT in train mode: 37 dropouts perturb the text features.  The fixture is the
    reference's own train-mode step with the masks it drew; given those masks the BERT oracle reproduces the features and the ViLT oracle
    the step -- and the eval-mode features are far away (this is not a small effect: tests/golden/viltbert_train_dropout.json)."""
    from oracle import bert_oracle as bo
    z = np.load(os.path.join(golden_dir, "viltbert_vqa_b3_train.npz"))
    m = _meta(z)
    tasks, B, T = m["tasks"].split(","), int(m["B"]), int(m["T"])
    P, PB = vo.init_params(tasks, int(m["wseed"])), bo.init_bert_params(int(m["bseed"]))
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]), ragged_text=True)
    masks = unpack_bert_masks(z, B, T)
    with torch.no_grad():
        feats = bo.bert_forward(PB, enc["input_ids"], enc["token_type_ids"], enc["attention_mask"], masks=masks)
        ev = bo.bert_forward(PB, enc["input_ids"], enc["token_type_ids"], enc["attention_mask"])
    valid = enc["attention_mask"].bool()
    _close(feats[valid][:, :8], torch.from_numpy(z["bert_feats_head"])[valid], 2e-5, "BERT features (train mode, reference masks)")
    assert float((feats[valid] - ev[valid]).norm() / ev[valid].norm()) > 0.2
    oenc = dict(enc, inputs_embeds=feats)
    oenc.pop("input_ids")
    with torch.no_grad():
        pooled, logits = vo.learner_forward(P, "vqa", oenc, training=True)
    _close(pooled, z["pooled"], 2e-5, "pooled")
    _close(logits, z["logits"], 2e-5, "logits")
    _close(vo.vqa_loss(logits, vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))), z["loss"], 2e-5, "loss")
    d = json.load(open(os.path.join(golden_dir, "viltbert_train_dropout.json")))
    assert 0.3 < d["mean_feature_rel_rms"] < 0.9 and d["mean_grad_rel_l2"] > 0.3 and len(d["rows"]) >= 8


@pytest.mark.parametrize("fixture", ["nlvr2_b4_varres.npz", "snlive_b4_640.npz", "vcr_b3_varres.npz"])
def test_oracle_at_the_loaders_real_input_shapes(golden_dir, fixture):
    """r03 fixtures (the reference's own steps on variable-resolution / 384 x 640 inputs for NLVR2, SNLI-VE, VCR): the CPU oracle reproduces
    pooled, logits and loss (its gradients were checked against the reference's when the fixtures were generated, oracle/gen_golden.py)."""
    z = np.load(os.path.join(golden_dir, fixture))
    m = _meta(z)
    task = m["task"]
    P = vo.init_params(m["tasks"].split(","), int(m["wseed"]))
    if task == "snli-ve":
        enc = vo.synthetic_varres_encodings([(384, 640)] * int(m["B"]), seed=int(m["dseed"]))
    else:
        sizes = [tuple(int(v) for v in r) for r in z["sizes"]]
        ei = vo.synthetic_varres_encodings(sizes, seed=int(m["dseed"]))
        if task == "nlvr2":
            b = int(m["b"])
            enc = dict(input_ids=ei["input_ids"][:b], token_type_ids=ei["token_type_ids"][:b], attention_mask=ei["attention_mask"][:b],
                       pixel_values=ei["pixel_values"], pixel_mask=ei["pixel_mask"])
        else:
            et = vo.synthetic_encodings(4 * len(sizes), seed=int(m["dseed"]), ragged_text=True)
            enc = dict(input_ids=et["input_ids"], token_type_ids=et["token_type_ids"], attention_mask=et["attention_mask"],
                       pixel_values=ei["pixel_values"], pixel_mask=ei["pixel_mask"])
    with torch.no_grad():
        pooled, logits = vo.learner_forward(P, task, enc, training=False)
    _close(pooled, z["pooled"], 2e-5, "pooled")
    _close(logits, z["logits"], 2e-5, "logits")
    _close(vo.ce_loss(logits, torch.from_numpy(z["labels"])), z["loss"], 2e-5, "loss")
