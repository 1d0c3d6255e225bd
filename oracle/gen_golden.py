"""Generate tests/golden/*.npz by running the CLiMB reference itself.  BUILD-CONTAINER ONLY.

Usage:  python oracle/gen_golden.py [fullsize|varres|cl_eval]   (needs /root/reference; ~5 min on 8 cores)

Every fixture is DATA: outputs of the reference's own `ViltContinualLearner`, `*Trainer.train_step`,
`EWC.compute_ewc_loss` and `EWC.save_task_parameters` on seeded synthetic inputs.  Weights and inputs
are regenerated on the consumer side from `oracle/vilt_oracle.py` (`init_params`, `synthetic_*`),
so only outputs are stored.  While generating, the script also asserts that the oracle restatement
agrees with the reference (tolerances below), which is what pins the oracle.
"""
from __future__ import annotations

import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import as ri          # noqa: E402
import vilt_oracle as vo         # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SLICE = 8


def tensor_summary(named):
    """per-tensor L2 norm + first SLICE elements, in dict order."""
    names = list(named.keys())
    norms = np.array([float(named[n].double().norm()) for n in names], dtype=np.float64)
    heads = np.zeros((len(names), SLICE), dtype=np.float32)
    for i, n in enumerate(names):
        f = named[n].detach().reshape(-1)[:SLICE].float().numpy()
        heads[i, :f.size] = f
    return norms, heads


def check(name, a, b, rtol):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    err = float((a - b).abs().max())
    scale = float(b.abs().max()) + 1e-30
    print(f"  oracle-vs-reference {name:34s} max|d|={err:.3e} scale={scale:.3e} rel={err / scale:.2e}")
    assert err <= rtol * scale, (name, err, scale)


def ref_grads(model):
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def run_ref_step(model, trainer, enc, batch, optimizer=None, scheduler=None, ewc=None):
    ri.bypass_processor(model, enc)
    torch.manual_seed(0)      # only affects the reference's patch shuffle (HF:153-159)
    return trainer.train_step(model, batch, optimizer, scheduler, ewc)


def case_single_image(task, tasks, B, fname, ragged=False, wseed=42, dseed=1):
    print(f"[{fname}] task={task} tasks={tasks} B={B} ragged={ragged}")
    P = vo.init_params(tasks, wseed)
    enc = vo.synthetic_encodings(B, seed=dseed, ragged_text=ragged)
    model = ri.build_reference_learner(tasks, P)
    model.train()
    trainer = ri.make_trainer(task)
    if task == "vqa":
        target = vo.synthetic_vqa_targets(B, seed=dseed)
        batch = {"raw_texts": [""] * B, "images": None, "target_scores": target}
    else:
        rng = np.random.default_rng([dseed, 13])
        target = torch.from_numpy(rng.integers(0, vo.TASKS[task]["num_labels"], size=(B,), dtype=np.int64))
        batch = {"raw_texts": [""] * B, "images": None, "labels": target}
    model.zero_grad()
    loss, (pooled, logits), _, _ = run_ref_step(model, trainer, enc, batch)
    G = ref_grads(model)
    # oracle agreement
    o_loss, (o_pooled, o_logits), _, o_G = vo.train_step(P, task, enc, target)
    check("pooled", o_pooled, pooled, 2e-5)
    check("logits", o_logits, logits, 2e-5)
    check("loss", o_loss, loss, 2e-5)
    assert torch.equal(o_logits.argmax(-1), logits.argmax(-1))
    for n in G:
        check("grad " + n[-28:], o_G[n], G[n], 2e-4) if n.endswith(("query.weight", "cls_token", "3.weight", "word_embeddings.weight", "projection.weight")) else None
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary({n: o_G[n] for n in G})
    check("grad norms (all tensors)", on, gn, 1e-4)
    np.savez_compressed(os.path.join(OUT, fname), pooled=pooled.detach().numpy(), logits=logits.detach().numpy(),
                        loss=np.float64(loss.item()), grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        meta=np.array([f"task={task};tasks={','.join(tasks)};B={B};wseed={wseed};dseed={dseed};ragged={int(ragged)}"]))
    return model, P


VARRES_SIZES = [(384, 512), (288, 384), (384, 384), (352, 640)]      # (H, W) multiples of 32, shortest edge <= 384, longest <= 640


# 16 COCO-like images of BOTH orientations (what every real VQA batch looks like, and what bench.py's real_input leg trains on): the padded
# canvas is 640 x 640 = 400 patches, no image has more than 240 valid ones -- the reference keeps max_b(h w) rows (HF:131-159)
MIXED_SIZES = [(384, 512), (512, 384), (384, 384), (352, 640), (640, 352), (384, 576), (576, 384), (320, 384)] * 2


def case_varres(fname, tasks=("vqa", "nlvr2"), wseed=42, dseed=9, sizes=None):
    """Row F2: padded variable-resolution batch through the reference's own masked visual_embed (random patch selection)."""
    VARRES_SIZES = sizes or globals()["VARRES_SIZES"]
    print(f"[{fname}] variable-resolution batch {VARRES_SIZES}")
    tasks = list(tasks)
    B = len(VARRES_SIZES)
    P = vo.init_params(tasks, wseed)
    enc = vo.synthetic_varres_encodings(VARRES_SIZES, seed=dseed)
    target = vo.synthetic_vqa_targets(B, seed=dseed)
    model = ri.build_reference_learner(tasks, P)
    model.train()
    trainer = ri.make_trainer("vqa")
    batch = {"raw_texts": [""] * B, "images": None, "target_scores": target}
    model.zero_grad()
    loss, (pooled, logits), _, _ = run_ref_step(model, trainer, enc, batch)
    G = ref_grads(model)
    o_loss, (o_pooled, o_logits), _, o_G = vo.train_step(P, "vqa", enc, target)
    check("pooled", o_pooled, pooled, 2e-5)
    check("logits", o_logits, logits, 2e-5)
    check("loss", o_loss, loss, 2e-5)
    assert torch.equal(o_logits.argmax(-1), logits.argmax(-1))
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary({n: o_G[n] for n in G})
    check("grad norms (all tensors)", on, gn, 1e-4)
    pe = vo.ENC + "embeddings.position_embeddings"
    check("grad position_embeddings (bilinear taps)", o_G[pe], G[pe], 2e-4)
    np.savez_compressed(os.path.join(OUT, fname), pooled=pooled.detach().numpy(), logits=logits.detach().numpy(),
                        loss=np.float64(loss.item()), grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        sizes=np.array(VARRES_SIZES),
                        meta=np.array([f"task=vqa;tasks={','.join(tasks)};B={B};wseed={wseed};dseed={dseed};varres=1"]))


def case_steps(fname, steps=10, B=2, tasks=("vqa", "nlvr2"), wseed=42):
    """BASELINE config 1: ViLT sequential-FT on VQAv2, batch=2, 10 steps, with the reference's optimizer + schedule
    (REF/train/visionlanguage_tasks/train_vqa.py:197-205; max_steps=steps so warm-up = 1 step)."""
    print(f"[{fname}] {steps} reference AdamW steps, B={B}")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    P0 = {n: t.clone() for n, t in P.items()}
    model = ri.build_reference_learner(tasks, P)
    model.train()
    trainer = ri.make_trainer("vqa")
    ns = ri.import_reference()
    opt = model.create_optimizer(trainer.hparams)
    sched = ns.get_poly(opt, num_warmup_steps=int(steps * 0.1), num_training_steps=steps, lr_end=0, power=1)
    model.zero_grad()
    losses, o_losses = [], []
    state = {}
    warm = int(steps * 0.1)
    for s in range(steps):
        enc = vo.synthetic_encodings(B, seed=100 + s)
        target = vo.synthetic_vqa_targets(B, seed=100 + s)
        batch = {"raw_texts": [""] * B, "images": None, "target_scores": target}
        loss, _, _, _ = run_ref_step(model, trainer, enc, batch, opt, sched)
        losses.append(loss.item())
        lr = vo.poly_lr(s, 1e-4, warm, steps)
        o_loss, _, _, _ = vo.train_step(P, "vqa", enc, target, opt_state=state, lr=lr)
        o_losses.append(o_loss.item())
    check("loss curve", np.array(o_losses), np.array(losses), 1e-4)
    after = {n: p.detach().clone() for n, p in model.named_parameters()}
    delta = {n: after[n] - P0[n] for n in after}
    o_delta = {n: P[n] - P0[n] for n in after}
    dn, dh = tensor_summary(delta)
    odn, _ = tensor_summary(o_delta)
    check("param-delta norms", odn, dn, 2e-3)
    _, ah = tensor_summary(after)
    np.savez_compressed(os.path.join(OUT, fname), losses=np.array(losses), names=np.array(list(after.keys())),
                        delta_norms=dn, delta_heads=dh, after_heads=ah,
                        meta=np.array([f"task=vqa;tasks={','.join(tasks)};B={B};steps={steps};wseed={wseed};dseed=100+s;lr=1e-4"]))


def synthetic_ewc_state(P, seed=5):
    """theta* = weights + small seeded noise, F ~ U(0,1)*1e-4, encoder-relative names."""
    fisher, star = {}, {}
    for n in vo.encoder_names(P):
        k = n[len("vilt_encoder."):]
        rng = np.random.default_rng([seed, vo.zlib.crc32(k.encode())])
        fisher[k] = torch.from_numpy((rng.random(P[n].shape, dtype=np.float32) * 1e-4).astype(np.float32))
        star[k] = P[n] + torch.from_numpy((0.01 * rng.standard_normal(P[n].shape, dtype=np.float32)).astype(np.float32))
    return fisher, star


def case_ewc(fname, B=2, tasks=("vqa", "nlvr2"), wseed=42, dseed=1):
    print(f"[{fname}] EWC penalty + train_step with ewc")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    fisher, star = synthetic_ewc_state(P)
    model = ri.build_reference_learner(tasks, P)
    model.train()
    ns = ri.import_reference()
    args = types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=100.0)
    ewc = ns.ewc.EWC(args)
    ewc.task_keys = ["nlvr2"]
    ewc.fisher_dict = {"nlvr2": fisher}
    ewc.param_dict = {"nlvr2": star}
    ewc.device = torch.device("cpu")
    trainer = ri.make_trainer("vqa")
    enc = vo.synthetic_encodings(B, seed=dseed)
    target = vo.synthetic_vqa_targets(B, seed=dseed)
    batch = {"raw_texts": [""] * B, "images": None, "target_scores": target}
    model.zero_grad()
    random.seed(0)
    loss, _, ewc_task, ewc_loss = run_ref_step(model, trainer, enc, batch, ewc=ewc)
    G = ref_grads(model)
    o_loss, _, o_el, o_G = vo.train_step(P, "vqa", enc, target, ewc=(fisher, star, 100.0))
    check("ewc_loss", o_el, ewc_loss, 1e-5)
    check("loss", o_loss, loss, 2e-5)
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary({n: o_G[n] for n in G})
    check("grad norms with EWC", on, gn, 1e-4)
    np.savez_compressed(os.path.join(OUT, fname), loss=np.float64(loss.item()), ewc_loss=np.float64(ewc_loss.item()),
                        grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        meta=np.array([f"task=vqa;tasks={','.join(tasks)};B={B};wseed={wseed};dseed={dseed};ewc_seed=5;lam=100"]))


def case_fisher(fname, B=2, nb=3, tasks=("vqa", "nlvr2"), wseed=42):
    """EWC.save_task_parameters (REF/cl_algorithms/ewc.py:28-73) over 3 accumulating batches."""
    print(f"[{fname}] Fisher from {nb} accumulating batches")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    model = ri.build_reference_learner(tasks, P)
    model.train()
    ns = ri.import_reference()
    trainer = ri.make_trainer("vqa")
    encs = [vo.synthetic_encodings(B, seed=200 + i) for i in range(nb)]
    tgts = [vo.synthetic_vqa_targets(B, seed=200 + i) for i in range(nb)]

    class _Loader(list):
        pass
    loader = _Loader()
    for i in range(nb):
        loader.append({"raw_texts": [""] * B, "images": None, "target_scores": tgts[i], "_i": i})
    loader.dataset = list(range(int(nb * B / 0.01)))          # 1 % of it == nb*B samples
    trainer.get_train_dataloader = lambda: loader
    orig = trainer.train_step

    def step(model, batch, *a, **k):
        ri.bypass_processor(model, encs[batch["_i"]])
        torch.manual_seed(0)
        return orig(model, batch, *a, **k)
    trainer.train_step = step
    args = types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=100.0)
    ewc = ns.ewc.EWC(args)
    ewc.save_task_parameters(task_key="vqa", model=model, task_trainer=trainer, device=torch.device("cpu"))
    fisher = {k: v for k, v in ewc.fisher_dict["vqa"].items()}
    # oracle
    grads = []
    for i in range(nb):
        _, _, _, G = vo.train_step(P, "vqa", encs[i], tgts[i])
        grads.append({n[len("vilt_encoder."):]: g for n, g in G.items() if n.startswith(vo.ENC)})
    o_fisher = vo.fisher_from_batch_grads(grads, [B] * nb)
    fn, fh = tensor_summary(fisher)
    ofn, _ = tensor_summary({k: o_fisher[k] for k in fisher})
    check("fisher norms", ofn, fn, 5e-4)
    np.savez_compressed(os.path.join(OUT, fname), names=np.array(list(fisher.keys())), fisher_norms=fn, fisher_heads=fh,
                        meta=np.array([f"task=vqa;tasks={','.join(tasks)};B={B};batches={nb};wseed={wseed};dseed=200+i"]))


def case_nlvr2(fname, b=2, tasks=("vqa", "nlvr2"), wseed=42, dseed=3):
    print(f"[{fname}] NLVR2 two-image forward/backward")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    e1 = vo.synthetic_encodings(2 * b, seed=dseed)
    enc = dict(input_ids=e1["input_ids"][:b], token_type_ids=e1["token_type_ids"][:b],
               attention_mask=e1["attention_mask"][:b], pixel_values=e1["pixel_values"], pixel_mask=e1["pixel_mask"])
    labels = torch.tensor([1, 0][:b]) if b <= 2 else torch.from_numpy(np.random.default_rng([dseed, 13]).integers(0, 2, size=(b,), dtype=np.int64))
    model = ri.build_reference_learner(tasks, P)
    model.train()
    trainer = ri.make_trainer("nlvr2")
    batch = {"raw_texts": [""] * b, "images": [[None, None]] * b, "labels": labels}
    model.zero_grad()
    loss, (pooled, logits), _, _ = run_ref_step(model, trainer, enc, batch)
    G = ref_grads(model)
    o_loss, (o_pooled, o_logits), _, o_G = vo.train_step(P, "nlvr2", enc, labels)
    check("pooled", o_pooled, pooled, 2e-5)
    check("logits", o_logits, logits, 2e-5)
    check("loss", o_loss, loss, 2e-5)
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary({n: o_G[n] for n in G})
    check("grad norms", on, gn, 1e-4)
    np.savez_compressed(os.path.join(OUT, fname), pooled=pooled.detach().numpy(), logits=logits.detach().numpy(),
                        loss=np.float64(loss.item()), labels=labels.numpy(),
                        grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        meta=np.array([f"task=nlvr2;tasks={','.join(tasks)};b={b};wseed={wseed};dseed={dseed}"]))


NLVR2_VARRES_SIZES = [(384, 512), (512, 384), (384, 384), (352, 640), (640, 352), (320, 384), (384, 576), (576, 384)]      # image j of example i = row 2 i + j


def case_nlvr2_varres(fname, tasks=("vqa", "nlvr2"), wseed=42, dseed=14):
    """NLVR2 on real-world-shaped inputs: two images per example, every image its own resolution and orientation on one padded canvas, ragged
    text.  The reference flattens the images through ONE processor call and runs two encoder passes (image 0 of every example with
    image_token_type_idx 1, then image 1 with 2: REF/modeling/vilt.py:281-304), each with its own max_b(h w) patch count."""
    sizes = NLVR2_VARRES_SIZES
    b = len(sizes) // 2
    print(f"[{fname}] NLVR2, {b} examples x 2 variable-resolution images {sizes}")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    e1 = vo.synthetic_varres_encodings(sizes, seed=dseed)
    enc = dict(input_ids=e1["input_ids"][:b], token_type_ids=e1["token_type_ids"][:b], attention_mask=e1["attention_mask"][:b],
               pixel_values=e1["pixel_values"], pixel_mask=e1["pixel_mask"])
    labels = torch.from_numpy(np.random.default_rng([dseed, 13]).integers(0, 2, size=(b,), dtype=np.int64))
    model = ri.build_reference_learner(tasks, P)
    model.train()
    trainer = ri.make_trainer("nlvr2")
    batch = {"raw_texts": [""] * b, "images": [[None, None]] * b, "labels": labels}
    model.zero_grad()
    loss, (pooled, logits), _, _ = run_ref_step(model, trainer, enc, batch)
    G = ref_grads(model)
    o_loss, (o_pooled, o_logits), _, o_G = vo.train_step(P, "nlvr2", enc, labels)
    check("pooled", o_pooled, pooled, 2e-5)
    check("logits", o_logits, logits, 2e-5)
    check("loss", o_loss, loss, 2e-5)
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary({n: o_G[n] for n in G})
    check("grad norms", on, gn, 1e-4)
    np.savez_compressed(os.path.join(OUT, fname), pooled=pooled.detach().numpy(), logits=logits.detach().numpy(),
                        loss=np.float64(loss.item()), labels=labels.numpy(), sizes=np.array(sizes),
                        grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        meta=np.array([f"task=nlvr2;tasks={','.join(tasks)};b={b};wseed={wseed};dseed={dseed};varres=1"]))


def case_snlive_640(fname="snlive_b4_640.npz", tasks=("snli-ve", "vcr"), B=4, wseed=42, dseed=15):
    """SNLI-VE as CLiMB feeds it: Flickr30K images are resized to exactly 384 x 640 (REF/data/image_datasets/flickr30kimages_dataset.py:51), so
    every sequence has the maximum 12 x 20 = 240 patches: 40 + 1 + 240 = 281 tokens, ragged text."""
    print(f"[{fname}] SNLI-VE, {B} images of 384 x 640 (281-token sequences)")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    enc = vo.synthetic_varres_encodings([(384, 640)] * B, seed=dseed)
    target = torch.from_numpy(np.random.default_rng([dseed, 13]).integers(0, 3, size=(B,), dtype=np.int64))
    model = ri.build_reference_learner(tasks, P)
    model.train()
    trainer = ri.make_trainer("snli-ve")
    batch = {"raw_texts": [""] * B, "images": None, "labels": target}
    model.zero_grad()
    loss, (pooled, logits), _, _ = run_ref_step(model, trainer, enc, batch)
    G = ref_grads(model)
    o_loss, (o_pooled, o_logits), _, o_G = vo.train_step(P, "snli-ve", enc, target)
    check("pooled", o_pooled, pooled, 2e-5)
    check("logits", o_logits, logits, 2e-5)
    check("loss", o_loss, loss, 2e-5)
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary({n: o_G[n] for n in G})
    check("grad norms", on, gn, 1e-4)
    np.savez_compressed(os.path.join(OUT, fname), pooled=pooled.detach().numpy(), logits=logits.detach().numpy(), loss=np.float64(loss.item()),
                        labels=target.numpy(), grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        meta=np.array([f"task=snli-ve;tasks={','.join(tasks)};B={B};wseed={wseed};dseed={dseed};size=384x640"]))


VCR_VARRES_SIZES = [(384, 512), (512, 384), (352, 640)]


def case_vcr_varres(fname="vcr_b3_varres.npz", tasks=("snli-ve", "vcr"), wseed=42, dseed=16):
    """VCR on variable-resolution images: four answer choices per question over the SAME image (REF/modeling/vilt.py:331-347), eval mode."""
    sizes = VCR_VARRES_SIZES
    b = len(sizes)
    print(f"[{fname}] VCR, {b} variable-resolution images x 4 choices {sizes}")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    ei = vo.synthetic_varres_encodings(sizes, seed=dseed)
    et = vo.synthetic_encodings(4 * b, seed=dseed, ragged_text=True)
    enc = dict(input_ids=et["input_ids"], token_type_ids=et["token_type_ids"], attention_mask=et["attention_mask"],
               pixel_values=ei["pixel_values"], pixel_mask=ei["pixel_mask"])
    labels = torch.from_numpy(np.random.default_rng([dseed, 13]).integers(0, 4, size=(b,), dtype=np.int64))
    model = ri.build_reference_learner(tasks, P)
    model.eval()
    trainer = ri.make_trainer("vcr")
    batch = {"raw_texts": [[""] * 4] * b, "images": [None] * b, "labels": labels}
    model.zero_grad()
    loss, (pooled, logits), _, _ = run_ref_step(model, trainer, enc, batch)
    G = ref_grads(model)
    leaves = {n: P[n].clone().requires_grad_(True) for n in P}
    o_pooled, o_logits = vo.learner_forward(leaves, "vcr", enc, training=False)
    o_loss = vo.ce_loss(o_logits, labels)
    o_loss.backward()
    o_G = {n: leaves[n].grad for n in G}
    check("pooled", o_pooled, pooled, 2e-5)
    check("logits", o_logits, logits, 2e-5)
    check("loss", o_loss, loss, 2e-5)
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary(o_G)
    check("grad norms", on, gn, 1e-4)
    np.savez_compressed(os.path.join(OUT, fname), pooled=pooled.detach().numpy(), logits=logits.detach().numpy(), loss=np.float64(loss.item()),
                        labels=labels.numpy(), sizes=np.array(sizes), grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        meta=np.array([f"task=vcr;tasks={','.join(tasks)};b={b};wseed={wseed};dseed={dseed};varres=1;eval=1"]))


def case_vcr(fname, b=2, tasks=("snli-ve", "vcr"), wseed=42, dseed=4):
    """VCR multi-choice, eval mode (the head's Dropout(0.1) is the only stochastic op on the path)."""
    print(f"[{fname}] VCR four-choice forward/backward (eval mode)")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    e1 = vo.synthetic_encodings(4 * b, seed=dseed, ragged_text=True)
    enc = dict(input_ids=e1["input_ids"], token_type_ids=e1["token_type_ids"], attention_mask=e1["attention_mask"],
               pixel_values=e1["pixel_values"][:b], pixel_mask=e1["pixel_mask"][:b])
    labels = torch.tensor([2, 0][:b]) if b <= 2 else torch.from_numpy(np.random.default_rng([dseed, 13]).integers(0, 4, size=(b,), dtype=np.int64))
    model = ri.build_reference_learner(tasks, P)
    model.eval()
    trainer = ri.make_trainer("vcr")
    batch = {"raw_texts": [[""] * 4] * b, "images": [None] * b, "labels": labels}
    model.zero_grad()
    loss, (pooled, logits), _, _ = run_ref_step(model, trainer, enc, batch)
    G = ref_grads(model)
    leaves = {n: P[n].clone().requires_grad_(True) for n in P}
    o_pooled, o_logits = vo.learner_forward(leaves, "vcr", enc, training=False)
    o_loss = vo.ce_loss(o_logits, labels)
    o_loss.backward()
    o_G = {n: leaves[n].grad for n in G}
    check("pooled", o_pooled, pooled, 2e-5)
    check("logits", o_logits, logits, 2e-5)
    check("loss", o_loss, loss, 2e-5)
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary(o_G)
    check("grad norms", on, gn, 1e-4)
    np.savez_compressed(os.path.join(OUT, fname), pooled=pooled.detach().numpy(), logits=logits.detach().numpy(),
                        loss=np.float64(loss.item()), labels=labels.numpy(),
                        grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        meta=np.array([f"task=vcr;tasks={','.join(tasks)};b={b};wseed={wseed};dseed={dseed};ragged=1;eval=1"]))


def case_vcr_train(fname, b=2, tasks=("snli-ve", "vcr"), wseed=42, dseed=4, drop_seed=11):
    """VCR in TRAIN mode: the head's Dropout(0.1) (REF/modeling/vilt.py:199-202) is live.  The reference draws its mask from torch's CPU
    generator; a forward hook on that Dropout module records which elements it kept (output != 0 where the input was not), and the fixture
    carries that keep-mask so the oracle and the HIP path can be fed the very same one (`dropout_keep`)."""
    print(f"[{fname}] VCR four-choice forward/backward, TRAIN mode, keep-mask of the head dropout recorded")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    e1 = vo.synthetic_encodings(4 * b, seed=dseed, ragged_text=True)
    enc = dict(input_ids=e1["input_ids"], token_type_ids=e1["token_type_ids"], attention_mask=e1["attention_mask"],
               pixel_values=e1["pixel_values"][:b], pixel_mask=e1["pixel_mask"][:b])
    labels = torch.tensor([2, 0][:b]) if b <= 2 else torch.from_numpy(np.random.default_rng([dseed, 13]).integers(0, 4, size=(b,), dtype=np.int64))
    model = ri.build_reference_learner(tasks, P)
    model.train()
    drop = model.task_layer["vcr"][0]
    assert isinstance(drop, torch.nn.Dropout) and drop.p == 0.1
    seen = {}

    def hook(mod, inp, out):
        x = inp[0].detach()
        assert bool((x != 0).all())                       # tanh outputs: never exactly zero, so `out != 0` IS the keep-mask
        seen["keep"] = (out.detach() != 0)
        seen["ratio"] = float((out.detach()[seen["keep"]] / x[seen["keep"]]).mean())
    h = drop.register_forward_hook(hook)
    trainer = ri.make_trainer("vcr")
    batch = {"raw_texts": [[""] * 4] * b, "images": [None] * b, "labels": labels}
    model.zero_grad()
    ri.bypass_processor(model, enc)
    torch.manual_seed(drop_seed)
    loss, (pooled, logits), _, _ = trainer.train_step(model, batch)
    h.remove()
    keep = seen["keep"]
    assert keep.shape == (b, 4, 768) and abs(seen["ratio"] - 1.0 / 0.9) < 1e-5 and 0.85 < float(keep.float().mean()) < 0.95
    G = ref_grads(model)
    leaves = {n: P[n].clone().requires_grad_(True) for n in P}
    o_pooled, o_logits = vo.learner_forward(leaves, "vcr", enc, training=True, dropout_keep=keep.float())
    o_loss = vo.ce_loss(o_logits, labels)
    o_loss.backward()
    o_G = {n: leaves[n].grad for n in G}
    check("pooled", o_pooled, pooled, 2e-5)
    check("logits", o_logits, logits, 2e-5)
    check("loss", o_loss, loss, 2e-5)
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary(o_G)
    check("grad norms", on, gn, 1e-4)
    # the mask matters: the eval-mode logits differ from these by far more than any tolerance used downstream
    with torch.no_grad():
        _, e_logits = vo.learner_forward(P, "vcr", enc, training=False)
    assert float((e_logits - logits).abs().max()) > 1e-2 * float(logits.abs().max())
    np.savez_compressed(os.path.join(OUT, fname), pooled=pooled.detach().numpy(), logits=logits.detach().numpy(),
                        loss=np.float64(loss.item()), labels=labels.numpy(), keep=np.packbits(keep.numpy().reshape(-1)),
                        grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        meta=np.array([f"task=vcr;tasks={','.join(tasks)};b={b};wseed={wseed};dseed={dseed};ragged=1;eval=0;drop_seed={drop_seed}"]))


def case_replay(fname, B=2, tasks=("vqa", "nlvr2"), wseed=42, dseed=6):
    """ExperienceReplayMemory.run_replay_step (REF/cl_algorithms/experience_replay.py:53-67): a FRESH AdamW
    (zero moments, base lr, no scheduler) and one train_step on the replayed task."""
    print(f"[{fname}] ER replay step with a fresh optimizer")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    P0 = {n: t.clone() for n, t in P.items()}
    model = ri.build_reference_learner(tasks, P)
    model.train()
    ns = ri.import_reference()
    trainer = ri.make_trainer("vqa")
    enc = vo.synthetic_encodings(B, seed=dseed)
    target = vo.synthetic_vqa_targets(B, seed=dseed)
    batch = {"raw_texts": [""] * B, "images": None, "target_scores": target}
    orig = trainer.train_step

    def step(model, b, *a, **k):
        ri.bypass_processor(model, enc)
        torch.manual_seed(0)
        return orig(model, b, *a, **k)
    trainer.train_step = step
    mem = ns.er.ExperienceReplayMemory()
    buf = types.SimpleNamespace(task_config=ns.task_configs["vqa"], task_trainer=trainer,
                                sample_replay_batch=lambda: batch)
    mem.memory_buffers["vqa"] = buf
    model.zero_grad()
    loss = mem.run_replay_step(task_key="vqa", model=model)
    after = {n: p.detach().clone() for n, p in model.named_parameters()}
    state = {}
    o_loss, _, _, _ = vo.train_step(P, "vqa", enc, target, opt_state=state, lr=1e-4)
    check("loss", o_loss, loss, 2e-5)
    delta = {n: after[n] - P0[n] for n in after}
    dn, dh = tensor_summary(delta)
    odn, _ = tensor_summary({n: P[n] - P0[n] for n in after})
    check("param-delta norms", odn, dn, 1e-3)
    np.savez_compressed(os.path.join(OUT, fname), loss=np.float64(loss.item()), names=np.array(list(after.keys())),
                        delta_norms=dn, delta_heads=dh,
                        meta=np.array([f"task=vqa;tasks={','.join(tasks)};B={B};wseed={wseed};dseed={dseed};lr=1e-4;fresh_adamw=1"]))


def case_cl_eval(fname="cl_eval.json"):
    """Row F3: the reference's own upstream_knowledge_transfer_eval / catastrophic_forgetting_eval
    (REF/cl_evaluation/evaluate_cl_algorithm.py:32-140) on a synthetic 4-task run: the inputs (results.json contents, single-task
    scores, the scores `eval_forgetting` returns) and the dictionaries the reference computes from them."""
    import argparse
    import importlib
    import json
    import shutil
    import tempfile
    print(f"[{fname}] CL metrics from the reference's evaluate_cl_algorithm")
    ri.import_reference()
    ref_eval = importlib.import_module("cl_evaluation.evaluate_cl_algorithm")
    tasks = ["vqa", "nlvr2", "snli-ve", "vcr"]
    cl_scores = {"vqa": 67.31, "nlvr2": 73.07, "snli-ve": 76.28, "vcr": 61.02}
    single = {"vqa": 67.70, "nlvr2": 73.07 + 0.44, "snli-ve": 76.31, "vcr": 61.31}
    forget = {("nlvr2", "vqa"): 41.2, ("snli-ve", "vqa"): 30.5, ("snli-ve", "nlvr2"): 51.9, ("vcr", "vqa"): 12.25, ("vcr", "nlvr2"): 50.0,
              ("vcr", "snli-ve"): 70.125}
    root = tempfile.mkdtemp(prefix="climb_cl_eval_")
    try:
        args = argparse.Namespace(ordered_cl_tasks=tasks, output_dir=root, encoder_name="vilt")
        run_dir = os.path.join(root, "vilt-sequential_ft-task0_vqa-task1_nlvr2-task2_snli-ve-task3_vcr")
        os.makedirs(run_dir)
        results = [{"task_num": i, "task_key": t, "best_score": cl_scores[t], "best_epoch": 3 + i} for i, t in enumerate(tasks)]
        results_file = os.path.join(run_dir, "results.json")
        json.dump(results, open(results_file, "w"))
        for t in tasks:
            d = os.path.join(root, "vilt-singletask_ft-task0_{}".format(t))
            os.makedirs(d)
            json.dump([{"task_num": 0, "task_key": t, "best_score": single[t], "best_epoch": 5}], open(os.path.join(d, "results.json"), "w"))

        class _Trainer:
            def __init__(self, key):
                self.key = key

            def eval_forgetting(self, model, model_path):
                cur = os.path.basename(os.path.dirname(model_path)).split("_", 1)[1]
                return forget[(cur, self.key)]

        kt = ref_eval.upstream_knowledge_transfer_eval(args, results_file)
        cf = ref_eval.catastrophic_forgetting_eval(args, results_file, model=None, task_trainers={t: _Trainer(t) for t in tasks}, adapter_handler=None)
    finally:
        shutil.rmtree(root)
    out = {"tasks": tasks, "cl_scores": cl_scores, "singletask_scores": single, "forgetting_scores": {f"{a}|{b}": v for (a, b), v in forget.items()},
           "results": results, "knowledge_transfer": kt, "catastrophic_forgetting": {k: dict(v) for k, v in cf.items()}}
    json.dump(out, open(os.path.join(OUT, fname), "w"), indent=1)


def case_viltbert(fname="viltbert_vqa_b3.npz", tasks=("vqa", "nlvr2"), B=3, wseed=42, bseed=7, dseed=21):
    """Row F4: the reference's ViltBertContinualLearner (REF/modeling/viltbert.py): frozen BERT last hidden state as `inputs_embeds` of
    ViLT.  EVAL mode (the reference leaves BERT's dropouts live in train mode -- torch-RNG dependent; see oracle/bert_oracle.py)."""
    import bert_oracle as bo
    print(f"[{fname}] ViLT-BERT forward/backward (eval mode), B={B}, ragged text")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    PB = bo.init_bert_params(bseed)
    enc = vo.synthetic_encodings(B, seed=dseed, ragged_text=True)
    target = vo.synthetic_vqa_targets(B, seed=dseed)
    model = ri.build_reference_viltbert_learner(tasks, P, PB)
    model.eval()
    trainer = ri.make_trainer("vqa")
    import modeling.viltbert as ref_vb
    trainer.batch2inputs_converter = ref_vb.convert_batch_to_viltbert_input_dict
    model.viltbert_encoder.process_inputs = lambda images, texts: dict(enc)
    batch = {"raw_texts": [""] * B, "images": None, "target_scores": target}
    model.zero_grad()
    torch.manual_seed(0)
    loss, (pooled, logits), _, _ = trainer.train_step(model, batch, None, None, None)
    G = {n.replace("viltbert_encoder.", "vilt_encoder."): p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    assert not any(".bert." in n for n in G), "BERT is frozen (no_grad): no gradient reaches it"
    word = "vilt_encoder.vilt.embeddings.text_embeddings.word_embeddings.weight"
    assert word not in G, "the word-embedding table is bypassed by inputs_embeds"
    # oracle agreement
    with torch.no_grad():
        feats = bo.bert_forward(PB, enc["input_ids"], enc["token_type_ids"], enc["attention_mask"])
    ref_feats = model.viltbert_encoder.get_bert_outputs(**enc)
    check("bert last_hidden_state", feats, ref_feats, 2e-5)
    oenc = dict(enc, inputs_embeds=feats)
    oenc.pop("input_ids")
    leaves = {n: P[n].clone().requires_grad_(True) for n in P}
    o_pooled, o_logits = vo.learner_forward(leaves, "vqa", oenc, training=False)
    o_loss = vo.vqa_loss(o_logits, target)
    o_loss.backward()
    o_G = {n: leaves[n].grad for n in G}
    check("pooled", o_pooled, pooled, 2e-5)
    check("logits", o_logits, logits, 2e-5)
    check("loss", o_loss, loss, 2e-5)
    assert leaves[word].grad is None
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary(o_G)
    check("grad norms", on, gn, 1e-4)
    np.savez_compressed(os.path.join(OUT, fname), pooled=pooled.detach().numpy(), logits=logits.detach().numpy(), loss=np.float64(loss.item()),
                        bert_feats_head=ref_feats[:, :, :8].detach().numpy(), bert_feats_norm=np.float64(ref_feats.double().norm().item()),
                        grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        meta=np.array([f"task=vqa;tasks={','.join(tasks)};B={B};wseed={wseed};bseed={bseed};dseed={dseed};ragged=1;eval=1"]))


def case_viltbert_train(fname="viltbert_vqa_b3_train.npz", tasks=("vqa", "nlvr2"), B=3, wseed=42, bseed=7, dseed=21, drop_seed=3):
    """Row F4 in TRAIN mode: the reference never puts its frozen BERT in eval mode (REF/modeling/viltbert.py:115-120), so BERT's 37
    dropouts are live while the learner trains.  The reference's own train-mode step is run with every dropout call observed
    (torch.nn.functional.dropout wrapped for the duration: the keep-mask is `output != 0` wherever the input is not 0); the fixture carries
    those masks, so the oracle and the HIP path can be given the very masks the reference drew."""
    import bert_oracle as bo
    import torch.nn.functional as F
    print(f"[{fname}] ViLT-BERT forward/backward, TRAIN mode (BERT dropouts live, masks recorded), B={B}, ragged text")
    tasks = list(tasks)
    P = vo.init_params(tasks, wseed)
    PB = bo.init_bert_params(bseed)
    enc = vo.synthetic_encodings(B, seed=dseed, ragged_text=True)
    T = enc["input_ids"].shape[1]
    target = vo.synthetic_vqa_targets(B, seed=dseed)
    model = ri.build_reference_viltbert_learner(tasks, P, PB, eager_attention=True)
    model.train()
    assert model.viltbert_encoder.bert.training
    trainer = ri.make_trainer("vqa")
    import modeling.viltbert as ref_vb
    trainer.batch2inputs_converter = ref_vb.convert_batch_to_viltbert_input_dict
    model.viltbert_encoder.process_inputs = lambda images, texts: dict(enc)
    batch = {"raw_texts": [""] * B, "images": None, "target_scores": target}
    model.zero_grad()
    seen = []
    real = F.dropout

    def observed(input, p=0.5, training=True, inplace=False):
        out = real(input, p, training, False)
        if training and p > 0.0:
            assert p == 0.1
            seen.append(((out != 0) | (input == 0)).detach().clone())       # where the input is 0 (masked keys) the draw is unobservable AND irrelevant
        return out
    F.dropout = observed
    try:
        torch.manual_seed(drop_seed)
        loss, (pooled, logits), _, _ = trainer.train_step(model, batch, None, None, None)
    finally:
        F.dropout = real
    L, nh = 12, 12
    assert len(seen) == 1 + 3 * L, len(seen)
    masks = {"emb": seen[0], "probs": [seen[1 + 3 * i] for i in range(L)], "attn_out": [seen[2 + 3 * i] for i in range(L)],
             "ffn_out": [seen[3 + 3 * i] for i in range(L)]}
    assert masks["emb"].shape == (B, T, 768) and all(m.shape == (B, nh, T, T) for m in masks["probs"])
    assert all(m.shape == (B, T, 768) for m in masks["attn_out"] + masks["ffn_out"])
    assert 0.88 < float(masks["emb"].float().mean()) < 0.92
    G = {n.replace("viltbert_encoder.", "vilt_encoder."): p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    with torch.no_grad():
        feats = bo.bert_forward(PB, enc["input_ids"], enc["token_type_ids"], enc["attention_mask"], masks=masks)
        feats_eval = bo.bert_forward(PB, enc["input_ids"], enc["token_type_ids"], enc["attention_mask"])
    oenc = dict(enc, inputs_embeds=feats)
    oenc.pop("input_ids")
    leaves = {n: P[n].clone().requires_grad_(True) for n in P}
    o_pooled, o_logits = vo.learner_forward(leaves, "vqa", oenc, training=True)
    o_loss = vo.vqa_loss(o_logits, target)
    o_loss.backward()
    o_G = {n: leaves[n].grad for n in G}
    check("pooled", o_pooled, pooled, 2e-5)
    check("logits", o_logits, logits, 2e-5)
    check("loss", o_loss, loss, 2e-5)
    gn, gh = tensor_summary(G)
    on, _ = tensor_summary(o_G)
    check("grad norms", on, gn, 1e-4)
    valid = enc["attention_mask"].bool()
    dev = float((feats[valid] - feats_eval[valid]).norm() / feats_eval[valid].norm())
    print(f"  train-mode BERT features differ from the eval-mode ones by {dev:.2f} relative rms")
    assert dev > 0.2
    packed = np.concatenate([np.packbits(m.numpy().reshape(-1)) for m in [masks["emb"]] + masks["probs"] + masks["attn_out"] + masks["ffn_out"]])
    np.savez_compressed(os.path.join(OUT, fname), pooled=pooled.detach().numpy(), logits=logits.detach().numpy(), loss=np.float64(loss.item()),
                        masks=packed, bert_feats_head=feats[:, :, :8].numpy(), bert_feats_norm=np.float64(feats.double().norm().item()),
                        grad_names=np.array(list(G.keys())), grad_norms=gn, grad_heads=gh,
                        meta=np.array([f"task=vqa;tasks={','.join(tasks)};B={B};T={T};wseed={wseed};bseed={bseed};dseed={dseed};ragged=1;eval=0;drop_seed={drop_seed}"]))


def case_fullsize():
    """BASELINE configs[1] at its own size (64 sequences of 40 tokens + 384x384 per GPU) and the equal-sized NLVR2 / VCR batches
    (32 pairs, 16 x 4 choices): the reference's own `*Trainer.train_step` (REF/train/visionlanguage_tasks/train_vqa.py:135-174)."""
    case_single_image("vqa", ["vqa", "nlvr2"], 64, "vqa_b64.npz", dseed=64)
    case_nlvr2("nlvr2_b32.npz", b=32, dseed=32)
    case_vcr("vcr_b16.npz", b=16, dseed=16)


def main():
    assert ri.reference_available(), "needs /root/reference (build container only)"
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 8)
    if len(sys.argv) > 1 and sys.argv[1] == "varres":
        case_varres("vqa_b4_varres.npz")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "realshapes":
        case_snlive_640()
        case_vcr_varres()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "nlvr2_varres":
        case_nlvr2_varres("nlvr2_b4_varres.npz")
        return
    if len(sys.argv) > 1 and sys.argv[1] == "mixed":
        case_varres("vqa_b16_mixed.npz", dseed=10, sizes=MIXED_SIZES)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cl_eval":
        case_cl_eval()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":
        case_fullsize()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "viltbert":
        case_viltbert()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "viltbert_train":
        case_viltbert_train()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "vcr_train":
        case_vcr_train("vcr_b2_train.npz")
        return
    case_single_image("vqa", ["vqa", "nlvr2"], 2, "vqa_b2.npz")
    case_single_image("vqa", ["vqa", "nlvr2"], 3, "vqa_b3_ragged.npz", ragged=True, dseed=2)
    case_single_image("snli-ve", ["snli-ve", "vcr"], 2, "snlive_b2.npz", dseed=8)
    case_nlvr2("nlvr2_b2.npz")
    case_vcr("vcr_b2.npz")
    case_vcr_train("vcr_b2_train.npz")
    case_ewc("ewc_b2.npz")
    case_fisher("fisher_3x2.npz")
    case_replay("replay_b2.npz")
    case_steps("vqa_b2_10steps.npz")
    case_varres("vqa_b4_varres.npz")
    case_cl_eval()
    case_fullsize()
    case_viltbert()
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
