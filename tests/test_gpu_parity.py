"""Step-level parity on the MI355X, through the C ABI: the HIP path (climb_amd, fp32 mode) against the CPU oracle on the
same seeded inputs and against the golden vectors the reference produced (tests/golden/).

Tolerance (BASELINE.json north_star): <= 1e-3 relative in fp32, argmax task predictions bit-exact.
"relative" = max|a-b| / max|b| over the tensor."""
import json
import os
import random
import types

import numpy as np
import pytest
import torch

from oracle import vilt_oracle as vo

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _close(a, b, rtol, what=""):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if hasattr(a, "detach") else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if hasattr(b, "detach") else b)).double()
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max())
    assert err <= rtol * scale, f"{what}: max|d|={err:.3e} scale={scale:.3e} rel={err / scale:.2e}"
    return err / scale


def _meta(z):
    return dict(kv.split("=", 1) for kv in str(z["meta"][0]).split(";"))


# the 16-bit throughput mode under test: bf16, or IEEE half when the suite runs as CLIMB_AMD_H16=fp16 (tests/test_gpu_fp16_build.py
# runs part of this file that way in a subprocess: one process holds one build of the library).  Tolerances below are the bf16 ones.
H16 = os.environ.get("CLIMB_AMD_H16", "bf16")


_PARAMS = {}


def _seeded_params(tasks, wseed):
    """vo.init_params, generated once per (tasks, seed) in this process and handed out as fresh clones (the oracle's optimizer updates its
    copy in place): 3 s of numpy generators per call otherwise, in ~80 tests."""
    key = (tuple(tasks), wseed)
    if key not in _PARAMS:
        _PARAMS[key] = vo.init_params(list(tasks), wseed)
    return type(_PARAMS[key])((k, v.clone()) for k, v in _PARAMS[key].items())


def make_model(tasks, wseed=42, precision="fp32"):
    from climb_amd.modeling import create_continual_learner_map
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.configs.model_configs import model_configs
    dev = _dev()
    model = create_continual_learner_map["vilt"](model_name_or_path="random-init:empty", ordered_cl_tasks=list(tasks),      # every parameter is loaded below (strict)
                                                 model_config=model_configs["vilt"], task_configs=task_configs, device=dev, precision=precision)
    P = _seeded_params(tasks, wseed)
    missing, unexpected = model.load_state_dict({k: v for k, v in P.items()}, strict=True)
    model.to(dev)
    return model, P


def enc_to_inputs(enc):
    texts = dict(input_ids=enc["input_ids"], token_type_ids=enc["token_type_ids"], attention_mask=enc["attention_mask"])
    return enc["pixel_values"], texts


def grads_of(model):
    return {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}


def _summary(named, names, k=8):
    norms = np.array([float(named[n].double().norm()) for n in names])
    heads = np.zeros((len(names), k), dtype=np.float32)
    for i, n in enumerate(names):
        f = named[n].reshape(-1)[:k].float().numpy()
        heads[i, :f.size] = f
    return norms, heads


def test_state_dict_names_and_layout():
    model, P = make_model(["vqa", "nlvr2"])
    sd = model.state_dict()
    assert list(sd.keys()) == list(P.keys())
    enc_sd = model.get_encoder().state_dict()
    assert all(k.startswith("vilt.") for k in enc_sd) and len(enc_sd) == 206
    for n, p in model.named_parameters():
        assert torch.equal(p.detach().cpu(), P[n]), n


_ORACLE_STEPS = {}


# the two modes that promise north_star's bar: exact-fp32 MFMA, and (r06) split (hi, lo) bf16 operands with three MFMA products per k-step
PARITY_MODES = ["fp32"] + (["bf16x3"] if os.environ.get("CLIMB_AMD_H16", "bf16") == "bf16" else [])


@pytest.mark.parametrize("precision", PARITY_MODES)
@pytest.mark.parametrize("fname", ["vqa_b2.npz", "vqa_b3_ragged.npz", "snlive_b2.npz"])
def test_single_image_step_vs_oracle_and_golden(golden_dir, fname, precision):
    z = np.load(os.path.join(golden_dir, fname))
    m = _meta(z)
    tasks, B, task = m["tasks"].split(","), int(m["B"]), m["task"]
    model, P = make_model(tasks, int(m["wseed"]), precision=precision)
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]), ragged_text=bool(int(m["ragged"])))
    if task == "vqa":
        target = vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))
    else:
        rng = np.random.default_rng([int(m["dseed"]), 13])
        target = torch.from_numpy(rng.integers(0, vo.TASKS[task]["num_labels"], size=(B,), dtype=np.int64))
    images, texts = enc_to_inputs(enc)
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward(task, images, texts, target)
    # vs golden (reference outputs)
    _close(pooled, z["pooled"], TOL, "pooled vs reference")
    _close(logits, z["logits"], TOL, "logits vs reference")
    _close(loss, z["loss"], TOL, "loss vs reference")
    assert np.array_equal(logits.argmax(-1).cpu().numpy(), z["logits"].argmax(-1)), "argmax must be bit-exact"
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    assert set(names) <= set(G.keys())
    norms, heads = _summary(G, names)
    _close(norms, z["grad_norms"], TOL, "grad norms vs reference")
    _close(heads, z["grad_heads"], TOL, "grad heads vs reference")
    # vs oracle, every element of every gradient (the oracle's CPU step is the same for every mode: computed once per fixture)
    if fname not in _ORACLE_STEPS:
        _ORACLE_STEPS[fname] = vo.train_step(P, task, enc, target)[3]
    oG = _ORACLE_STEPS[fname]
    worst = 0.0
    for n in names:
        if n.endswith("attention.key.bias"):
            # d loss / d key.bias is identically zero (a per-query constant added to every score cancels in the softmax);
            # both sides hold rounding noise only, so compare it against the query-bias gradient scale instead
            assert float(G[n].abs().max()) <= 1e-3 * float(oG[n.replace("key.bias", "query.bias")].abs().max() + 1e-30), n
            continue
        worst = max(worst, _close(G[n], oG[n], TOL, f"grad {n}"))
    print(f"{fname}: worst per-tensor gradient error vs oracle = {worst:.2e}")


def test_nlvr2_two_images(golden_dir):
    z = np.load(os.path.join(golden_dir, "nlvr2_b2.npz"))
    m = _meta(z)
    b = int(m["b"])
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]))
    e1 = vo.synthetic_encodings(2 * b, seed=int(m["dseed"]))
    texts = dict(input_ids=e1["input_ids"][:b], token_type_ids=e1["token_type_ids"][:b], attention_mask=e1["attention_mask"][:b])
    loss, (pooled, logits), _, _ = model.fused_forward_backward("nlvr2", e1["pixel_values"], texts, torch.from_numpy(z["labels"]))
    _close(pooled, z["pooled"], TOL, "pooled")
    _close(logits, z["logits"], TOL, "logits")
    _close(loss, z["loss"], TOL, "loss")
    assert np.array_equal(logits.argmax(-1).cpu().numpy(), z["logits"].argmax(-1))
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    _close(norms, z["grad_norms"], TOL, "grad norms")
    _close(heads, z["grad_heads"], TOL, "grad heads")


@pytest.mark.parametrize("precision,tol", [(p_, TOL) for p_ in PARITY_MODES] + [(H16, 3e-2)])
def test_nlvr2_two_variable_resolution_images_vs_reference(golden_dir, precision, tol):
    """NLVR2 as it arrives in practice (r03 fixture from the reference): two images per example, every image its own resolution and
    orientation on one padded canvas, ragged text.  The reference runs two encoder passes with their own patch counts
    (REF/modeling/vilt.py:281-304); here all 2b sequences are ONE packed batch with per-row image_token_type_idx."""
    z = np.load(os.path.join(golden_dir, "nlvr2_b4_varres.npz"))
    m = _meta(z)
    b = int(m["b"])
    sizes = [tuple(int(v) for v in r) for r in z["sizes"]]
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]), precision=precision)
    e1 = vo.synthetic_varres_encodings(sizes, seed=int(m["dseed"]))
    texts = dict(input_ids=e1["input_ids"][:b], token_type_ids=e1["token_type_ids"][:b], attention_mask=e1["attention_mask"][:b])
    images = dict(pixel_values=e1["pixel_values"], pixel_mask=e1["pixel_mask"])
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("nlvr2", images, texts, torch.from_numpy(z["labels"]))
    _close(pooled, z["pooled"], tol, "pooled")
    _close(logits, z["logits"], tol, "logits")
    _close(loss, z["loss"], tol, "loss")
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    if precision in PARITY_MODES:
        assert np.array_equal(logits.argmax(-1).cpu().numpy(), z["logits"].argmax(-1))
        _close(norms, z["grad_norms"], tol, "grad norms")
        _close(heads, z["grad_heads"], tol, "grad heads")
    else:
        big = z["grad_norms"] > 1e-3 * z["grad_norms"].max()
        assert (np.abs(norms - z["grad_norms"])[big] / z["grad_norms"][big]).max() < 6e-2


@pytest.mark.parametrize("precision,tol", [("fp32", TOL), (H16, 3e-2)])
@pytest.mark.parametrize("fixture", ["snlive_b4_640.npz", "vcr_b3_varres.npz"])
def test_tasks_at_their_real_input_shapes_vs_reference(golden_dir, fixture, precision, tol):
    """r03 fixtures from the reference at the shapes CLiMB's loaders actually produce: SNLI-VE's Flickr30K images are resized to exactly
    384 x 640 (REF/data/image_datasets/flickr30kimages_dataset.py:51): every sequence has the maximum 240 patches, 281 tokens; VCR's four
    answer choices run over the SAME variable-resolution image (REF/modeling/vilt.py:331-347; eval mode, the train-mode dropout has its own
    fixture)."""
    z = np.load(os.path.join(golden_dir, fixture))
    m = _meta(z)
    task = m["task"]
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]), precision=precision)
    if task == "snli-ve":
        B = int(m["B"])
        enc = vo.synthetic_varres_encodings([(384, 640)] * B, seed=int(m["dseed"]))
        model.train()
    else:
        sizes = [tuple(int(v) for v in r) for r in z["sizes"]]
        ei = vo.synthetic_varres_encodings(sizes, seed=int(m["dseed"]))
        et = vo.synthetic_encodings(4 * len(sizes), seed=int(m["dseed"]), ragged_text=True)
        enc = dict(input_ids=et["input_ids"], token_type_ids=et["token_type_ids"], attention_mask=et["attention_mask"],
                   pixel_values=ei["pixel_values"], pixel_mask=ei["pixel_mask"])
        model.eval()
    texts = dict(input_ids=enc["input_ids"], token_type_ids=enc["token_type_ids"], attention_mask=enc["attention_mask"])
    images = dict(pixel_values=enc["pixel_values"], pixel_mask=enc["pixel_mask"])
    loss, (pooled, logits), _, _ = model.fused_forward_backward(task, images, texts, torch.from_numpy(z["labels"]))
    _close(pooled, z["pooled"], tol, "pooled")
    _close(logits, z["logits"], tol, "logits")
    _close(loss, z["loss"], tol, "loss")
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    if precision in PARITY_MODES:
        assert np.array_equal(logits.argmax(-1).cpu().numpy(), z["logits"].argmax(-1))
        _close(norms, z["grad_norms"], tol, "grad norms")
        _close(heads, z["grad_heads"], tol, "grad heads")
    else:
        big = z["grad_norms"] > 1e-3 * z["grad_norms"].max()
        assert (np.abs(norms - z["grad_norms"])[big] / z["grad_norms"][big]).max() < 6e-2


def test_vcr_four_choices_eval(golden_dir):
    z = np.load(os.path.join(golden_dir, "vcr_b2.npz"))
    m = _meta(z)
    b = int(m["b"])
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]))
    e1 = vo.synthetic_encodings(4 * b, seed=int(m["dseed"]), ragged_text=True)
    texts = dict(input_ids=e1["input_ids"], token_type_ids=e1["token_type_ids"], attention_mask=e1["attention_mask"])
    model.eval()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vcr", e1["pixel_values"][:b], texts, torch.from_numpy(z["labels"]))
    _close(pooled, z["pooled"], TOL, "pooled")
    _close(logits, z["logits"], TOL, "logits")
    _close(loss, z["loss"], TOL, "loss")
    assert np.array_equal(logits.argmax(-1).cpu().numpy(), z["logits"].argmax(-1))
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    _close(norms, z["grad_norms"], TOL, "grad norms")
    _close(heads, z["grad_heads"], TOL, "grad heads")


def test_vcr_four_choices_train_mode_with_the_references_dropout_mask(golden_dir):
    """VERDICT r2 missing #4: the VCR head's train-mode Dropout(0.1) (REF/modeling/vilt.py:199-202).  `vcr_b2_train.npz` is the reference's
    own train-mode step with the keep-mask it drew recorded; the HIP path is given that mask (`dropout_keep`) and must reproduce the
    reference's logits, loss, argmax and gradients at the same bar as every other fixture."""
    z = np.load(os.path.join(golden_dir, "vcr_b2_train.npz"))
    m = _meta(z)
    b = int(m["b"])
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]))
    e1 = vo.synthetic_encodings(4 * b, seed=int(m["dseed"]), ragged_text=True)
    texts = dict(input_ids=e1["input_ids"], token_type_ids=e1["token_type_ids"], attention_mask=e1["attention_mask"])
    keep = torch.from_numpy(np.unpackbits(z["keep"])[:b * 4 * 768].reshape(b * 4, 768).astype(np.float32)).to(_dev())
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vcr", e1["pixel_values"][:b], texts, torch.from_numpy(z["labels"]), dropout_keep=keep)
    _close(pooled, z["pooled"], TOL, "pooled")
    _close(logits, z["logits"], TOL, "logits")
    _close(loss, z["loss"], TOL, "loss")
    assert np.array_equal(logits.argmax(-1).cpu().numpy(), z["logits"].argmax(-1))
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    _close(norms, z["grad_norms"], TOL, "grad norms")
    _close(heads, z["grad_heads"], TOL, "grad heads")
    # without the mask the train-mode step draws its own: a different result (the mask is live), same statistics
    model._host.drop_grads()
    _, (_, logits2), _, _ = model.fused_forward_backward("vcr", e1["pixel_values"][:b], texts, torch.from_numpy(z["labels"]))
    assert float((logits2.cpu() - torch.from_numpy(z["logits"])).abs().max()) > 1e-3


def _ewc_state(P, seed=5):
    import zlib
    fisher, star = {}, {}
    for n in vo.encoder_names(P):
        k = n[len("vilt_encoder."):]
        rng = np.random.default_rng([seed, zlib.crc32(k.encode())])
        fisher[k] = torch.from_numpy((rng.random(P[n].shape, dtype=np.float32) * 1e-4).astype(np.float32))
        star[k] = P[n] + torch.from_numpy((0.01 * rng.standard_normal(P[n].shape, dtype=np.float32)).astype(np.float32))
    return fisher, star


def test_ewc_penalty_and_gradient(golden_dir):
    from climb_amd.cl_algorithms import EWC
    z = np.load(os.path.join(golden_dir, "ewc_b2.npz"))
    m = _meta(z)
    B = int(m["B"])
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]))
    fisher, star = _ewc_state(P, int(m["ewc_seed"]))
    ewc = EWC(types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=float(m["lam"])))
    ewc.set_task_state("nlvr2", model, fisher, star)
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]))
    target = vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))
    images, texts = enc_to_inputs(enc)
    loss, _, ewc_task, ewc_loss = model.fused_forward_backward("vqa", images, texts, target, ewc=ewc)
    assert ewc_task == "nlvr2"
    _close(ewc_loss, z["ewc_loss"], 1e-5, "ewc loss")          # pure fp32 streaming reduction: much tighter than 1e-3
    _close(loss, z["loss"], TOL, "loss")
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    _close(norms, z["grad_norms"], TOL, "grad norms with EWC")
    _close(heads, z["grad_heads"], TOL, "grad heads with EWC")
    # reference-style call: compute_ewc_loss participates in autograd
    model._host.drop_grads()
    task, el = ewc.compute_ewc_loss(model)
    _close(el, z["ewc_loss"], 1e-5, "compute_ewc_loss")
    el.backward()
    g = model.get_encoder().vilt.pooler.dense.weight.grad
    k = "vilt.pooler.dense.weight"
    _close(g, 200.0 * fisher[k] * (P["vilt_encoder." + k] - star[k]), 1e-5, "d ewc / d theta")


def test_fisher_accumulating_quirk(golden_dir):
    from climb_amd.cl_algorithms import EWC
    from climb_amd.train import VQATrainer
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.configs.model_configs import model_configs
    z = np.load(os.path.join(golden_dir, "fisher_3x2.npz"))
    m = _meta(z)
    B, nb = int(m["B"]), int(m["batches"])
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]))

    class Loader(list):
        collate_fn = None
    loader = Loader()
    for i in range(nb):
        e = vo.synthetic_encodings(B, seed=200 + i)
        images, texts = enc_to_inputs(e)
        loader.append({"raw_texts": [""] * B, "encodings": texts, "images": images, "target_scores": vo.synthetic_vqa_targets(B, seed=200 + i)})
    loader.dataset = list(range(int(nb * B / 0.01)))
    args = types.SimpleNamespace(cl_algorithm="ewc")
    trainer = VQATrainer(args, task_configs, model_configs["vilt"], _dev(), train_dataloader=loader, val_dataloader=loader)
    ewc = EWC(types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=100.0))
    ewc.save_task_parameters(task_key="vqa", model=model, task_trainer=trainer, device=_dev())
    names = [str(n) for n in z["names"]]
    fisher = {k: v.detach().cpu() for k, v in ewc.fisher_dict["vqa"].items()}
    norms, heads = _summary(fisher, names)
    _close(norms, z["fisher_norms"], 2e-3, "fisher norms")      # squares of accumulated grads: 2x the relative error
    _close(heads, z["fisher_heads"], 2e-3, "fisher heads")
    for k in ("vilt.pooler.dense.weight", "vilt.embeddings.cls_token"):
        assert torch.equal(ewc.param_dict["vqa"][k].cpu(), P["vilt_encoder." + k])


@pytest.mark.parametrize("precision", PARITY_MODES)
def test_ten_steps_config1(golden_dir, precision):
    """BASELINE.json configs[0] on the GPU path: 10 AdamW steps at B=2 with the reference's schedule."""
    from climb_amd.train import polynomial_decay_schedule_with_warmup
    z = np.load(os.path.join(golden_dir, "vqa_b2_10steps.npz"))
    m = _meta(z)
    B, steps = int(m["B"]), int(m["steps"])
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]), precision=precision)
    P0 = {n: t.clone() for n, t in P.items()}
    opt = model.create_optimizer({"lr": float(m["lr"]), "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    sched = polynomial_decay_schedule_with_warmup(opt, int(steps * 0.1), steps, 0.0, 1.0)
    model.train()
    opt.zero_grad()
    losses = []
    for s in range(steps):
        enc = vo.synthetic_encodings(B, seed=100 + s)
        images, texts = enc_to_inputs(enc)
        loss, _, _, _ = model.fused_forward_backward("vqa", images, texts, vo.synthetic_vqa_targets(B, seed=100 + s), optimizer=opt if precision != "fp32" else None)
        opt.step()
        sched.step()
        opt.zero_grad()
        losses.append(loss.item())
    _close(np.array(losses), z["losses"], TOL, "loss curve")
    names = [str(n) for n in z["names"]]
    after = {n: p.detach().cpu() for n, p in model.named_parameters()}
    norms, _ = _summary({n: after[n] - P0[n] for n in names}, names)
    # d loss / d key.bias is identically zero (a per-query constant cancels in the softmax): both sides hold rounding noise only, and Adam's
    # g / (sqrt(v) + eps) turns noise of ANY size into updates of ~lr -- the reference's own delta of these tensors (2.3e-3 .. 3.5e-3) is that noise, and
    # every mode differs from it by about as much (measured r06: fp32 3.0e-3, bf16x3 3.5e-3 -- the latter straddled the old common 5e-3 bar from run to
    # run).  They are bounded by the noise's scale; everything else is compared as before.
    kb = np.array([n.endswith("attention.key.bias") for n in names])
    _close(norms[~kb], z["delta_norms"][~kb], 5e-3, "param delta norms")   # Adam's g/sqrt(v) amplifies rounding of tiny gradients
    assert kb.any() and float(norms[kb].max()) <= 3.0 * float(z["delta_norms"][kb].max()), "key.bias moved by more than its gradient noise allows"
    untouched = "task_layer.nlvr2.0.weight"
    assert torch.equal(after[untouched], P0[untouched]), "a head that received no gradient must not be decayed (torch skips grad=None)"


def test_replay_step_fresh_optimizer(golden_dir):
    from climb_amd.cl_algorithms import ExperienceReplayMemory
    from climb_amd.train import VQATrainer
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.configs.model_configs import model_configs
    z = np.load(os.path.join(golden_dir, "replay_b2.npz"))
    m = _meta(z)
    B = int(m["B"])
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]))
    P0 = {n: t.clone() for n, t in P.items()}
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]))
    images, texts = enc_to_inputs(enc)
    batch = {"raw_texts": [""] * B, "encodings": texts, "images": images, "target_scores": vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))}

    class Loader(list):
        collate_fn = staticmethod(lambda items: batch)
    loader = Loader([batch])
    loader.dataset = list(range(1000))
    trainer = VQATrainer(types.SimpleNamespace(cl_algorithm="experience_replay", batch_size=B), task_configs, model_configs["vilt"], _dev(),
                         train_dataloader=loader, val_dataloader=loader)
    random.seed(0)
    mem = ExperienceReplayMemory()
    mem.add_task_memory_buffer(args=types.SimpleNamespace(batch_size=B), task_key="vqa", task_config=task_configs["vqa"], task_trainer=trainer,
                               memory_percentage=0.01, sampling_strategy="random")
    model.train()
    loss = mem.run_replay_step(task_key="vqa", model=model)
    _close(loss, z["loss"], TOL, "replay loss")
    names = [str(n) for n in z["names"]]
    after = {n: p.detach().cpu() for n, p in model.named_parameters()}
    norms, _ = _summary({n: after[n] - P0[n] for n in names}, names)
    _close(norms, z["delta_norms"], 5e-3, "param delta norms")


def test_reference_style_autograd_path(golden_dir):
    """A reference-shaped train_step (REF train_vqa.py:135-174): model(...) -> torch loss -> loss.backward() -> optimizer.step()."""
    z = np.load(os.path.join(golden_dir, "vqa_b2.npz"))
    m = _meta(z)
    B = int(m["B"])
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]))
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]))
    images, texts = enc_to_inputs(enc)
    target = vo.synthetic_vqa_targets(B, seed=int(m["dseed"])).to(_dev())
    model.train()
    model.zero_grad()
    output = model(task_key="vqa", images=images, texts=texts)
    loss = torch.nn.BCEWithLogitsLoss(reduction="mean")(output[1], target) * target.shape[1]
    loss.backward()
    _close(loss, z["loss"], TOL, "loss")
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    _close(norms, z["grad_norms"], TOL, "grad norms")
    # accumulate a second backward (no zero_grad): grads double, like torch's .grad
    output = model(task_key="vqa", images=images, texts=texts)
    (torch.nn.BCEWithLogitsLoss(reduction="mean")(output[1], target) * target.shape[1]).backward()
    norms2, _ = _summary(grads_of(model), names)
    _close(norms2, 2 * z["grad_norms"], TOL, "accumulated grad norms")
    # deepcopy survives and still runs
    import copy
    m2 = copy.deepcopy(model)
    with torch.no_grad():
        o2 = m2(task_key="vqa", images=images, texts=texts)
    _close(o2[1], z["logits"], TOL, "deepcopy logits")


def test_freeze_bottom_k_skips_frozen_gradients():
    model, P = make_model(["vqa"], 42)
    model.get_encoder().freeze_bottom_k_layers(9)
    enc = vo.synthetic_encodings(2, seed=1)
    images, texts = enc_to_inputs(enc)
    target = vo.synthetic_vqa_targets(2, seed=1)
    model.fused_forward_backward("vqa", images, texts, target)
    G = grads_of(model)
    assert all(".layer.8." not in n and "embeddings" not in n for n in G)
    trainable = {n for n in P if not (n.startswith(vo.ENC + "embeddings") or any(f".layer.{i}." in n for i in range(9)))}
    _, _, _, oG = vo.train_step(P, "vqa", enc, target, trainable=trainable)
    for n in (vo.ENC + "encoder.layer.9.attention.attention.query.weight", vo.ENC + "encoder.layer.11.output.dense.bias", "task_layer.vqa.0.weight"):
        _close(G[n], oG[n], TOL, n)


# ------------------------------------------------------------------------------------------------ bf16 throughput mode
BF16_TOL = 3e-2     # bf16 operands (2^-8 relative rounding per GEMM input) through 12 layers; fp32 accumulate/statistics


def test_bf16_mode_step_vs_oracle(golden_dir):
    """BASELINE.json configs[1] arithmetic (bf16 MFMA operands) on the config-0 inputs: same step as the fp32 test,
    looser tolerance.  The 1e-3 / argmax-exact bar of north_star is the fp32 mode's; this documents what bf16 costs."""
    z = np.load(os.path.join(golden_dir, "vqa_b2.npz"))
    m = _meta(z)
    B = int(m["B"])
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]), precision=H16)
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]))
    target = vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))
    images, texts = enc_to_inputs(enc)
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target)
    e = dict(pooled=_close(pooled, z["pooled"], BF16_TOL, "pooled"), logits=_close(logits, z["logits"], BF16_TOL, "logits"),
             loss=_close(loss, z["loss"], BF16_TOL, "loss"))
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    norms, _ = _summary(G, names)
    big = z["grad_norms"] > 1e-3 * z["grad_norms"].max()
    rel = np.abs(norms - z["grad_norms"])[big] / z["grad_norms"][big]
    print("bf16 mode errors:", {k: f"{v:.2e}" for k, v in e.items()}, f"grad-norm rel err: median {np.median(rel):.2e} max {rel.max():.2e}")
    assert rel.max() < 6e-2
    # a second step after the fused AdamW must see refreshed bf16 shadows (straight and transposed)
    opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    opt.step()
    opt.zero_grad()
    loss2, _, _, _ = model.fused_forward_backward("vqa", images, texts, target)
    P2 = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    o_loss2, _, _, _ = vo.train_step(P2, "vqa", enc, target)
    _close(loss2, o_loss2, BF16_TOL, "loss after one AdamW step")
    assert abs(float(loss2) - float(loss)) > 1e-3 * abs(float(loss)), "the update must change the loss"


def test_bf16_shadows_follow_torch_side_parameter_writes(tmp_path):
    """nn.Module.load_state_dict / p.copy_() on a BOUND bf16 model write the fp32 master through the parameter
    views; the bf16 operand shadows (straight and transposed) must be rebuilt, or every encoder GEMM keeps the old weights.
    This is what `eval_forgetting` does (load a checkpoint into an already-used model, then eval)."""
    dev = _dev()
    B = 4
    pixels, texts, target = _rand_batch(B, 5, dev)
    model, _ = make_model(["vqa"], 42, precision=H16)
    model.train()
    opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    model.fused_forward_backward("vqa", pixels, texts, target)          # shadows built, one optimizer step taken
    opt.step()
    opt.zero_grad()
    other = vo.init_params(["vqa"], 43)                                   # a different model's weights
    path = str(tmp_path / "ckpt.pt")
    torch.save(other, path)
    model.load_state_dict(torch.load(path))
    model.eval()
    with torch.no_grad():
        got = model(task_key="vqa", images=pixels, texts=texts)[1].clone()
    fresh, _ = make_model(["vqa"], 43, precision=H16)
    fresh.eval()
    with torch.no_grad():
        want = fresh(task_key="vqa", images=pixels, texts=texts)[1].clone()
    _close(got, want, 1e-5, "logits after load_state_dict vs a freshly built model")
    # an in-place write to ONE weight (what a stock torch optimizer does) must be seen too
    with torch.no_grad():
        dict(model.named_parameters())["vilt_encoder.vilt.encoder.layer.3.output.dense.weight"].mul_(0.5)
        changed = model(task_key="vqa", images=pixels, texts=texts)[1].clone()
    assert float((changed - got).abs().max()) > 1e-4 * float(got.abs().max())
    # optimizer state survives a state_dict round trip (moments + per-parameter step counts)
    model.train()
    opt2 = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    model.fused_forward_backward("vqa", pixels, texts, target)
    opt2.step()
    opt2.zero_grad()
    sd = opt2.state_dict()
    assert "climb_amd_flat" in sd and float(sd["climb_amd_flat"]["v"].sum()) > 0
    opt3 = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    opt3.load_state_dict(sd)
    assert opt3._steps == opt2._steps and torch.equal(opt3._m, opt2._m) and torch.equal(opt3._v, opt2._v)


# ------------------------------------------------------------------------------------------------ Houlsby adapters (unpinned)
@pytest.mark.parametrize("precision,tol", [(p_, TOL) for p_ in PARITY_MODES] + [(H16, 4e-2)])
def test_houlsby_adapter_nlvr2_step_vs_oracle_restatement(precision, tol):
    """The NLVR2 half of BASELINE.json configs[2]: two images per example (image_token_type_idx 1 / 2, REF/modeling/vilt.py:292-303)
    under an ACTIVE adapter, base frozen -- the step the VQA -> NLVR2 adapter sequence runs for its second task."""
    from climb_amd.cl_algorithms import AdapterHandler
    tasks = ["vqa", "nlvr2"]
    model, _ = make_model(tasks, 42, precision=precision)
    args = types.SimpleNamespace(adapter_config="houlsby", adapter_reduction_factor=16, ordered_cl_tasks=tasks)
    handler = AdapterHandler("vanilla", args)
    handler.add_adapters_to_model(model)
    handler.activate_adapter_for_training(task_key="nlvr2", model=model)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".adapters." in n:
                p.copy_((torch.randn(p.shape, generator=g) * (0.05 if n.endswith("weight") else 0.02)).to(p.device))
    P = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    assert all((".adapters.nlvr2." in n) or n.startswith("task_layer.") for n in trainable)
    b = 3
    e1 = vo.synthetic_encodings(2 * b, seed=12)
    enc = dict(input_ids=e1["input_ids"][:b], token_type_ids=e1["token_type_ids"][:b], attention_mask=e1["attention_mask"][:b],
               pixel_values=e1["pixel_values"], pixel_mask=e1["pixel_mask"])
    texts = dict(input_ids=enc["input_ids"], token_type_ids=enc["token_type_ids"], attention_mask=enc["attention_mask"])
    labels = torch.tensor([1, 0, 1])
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("nlvr2", e1["pixel_values"], texts, labels)
    o_loss, (o_pooled, o_logits), _, oG = vo.train_step(P, "nlvr2", enc, labels, trainable=trainable, adapter="nlvr2")
    _close(pooled, o_pooled, tol, "pooled")
    _close(logits, o_logits, tol, "logits")
    _close(loss, o_loss, tol, "loss")
    G = grads_of(model)
    assert set(G) == set(oG), set(G) ^ set(oG)
    worst = 0.0
    for n in oG:
        if precision in PARITY_MODES:
            worst = max(worst, _close(G[n], oG[n], tol, n))
        else:
            worst = max(worst, abs(float(G[n].double().norm()) - float(oG[n].double().norm())) / (float(oG[n].double().norm()) + 1e-30))
    print(f"adapters nlvr2 [{precision}]: worst gradient error {worst:.2e}")
    assert worst < (tol if precision in PARITY_MODES else 6e-2)


@pytest.mark.parametrize("precision,tol", [(p_, TOL) for p_ in PARITY_MODES] + [(H16, 4e-2)])
def test_houlsby_adapters_vs_oracle_restatement(precision, tol):
    """BASELINE.json configs[2] arithmetic.  The GLAMOR adapter fork is absent, so this pins the HIP path to the oracle's
    restatement of public adapter-transformers semantics (out = y + up(swish(down(y)))), not to the reference."""
    from climb_amd.cl_algorithms import AdapterHandler
    tasks = ["vqa", "nlvr2"]
    model, _ = make_model(tasks, 42, precision=precision)
    args = types.SimpleNamespace(adapter_config="houlsby", adapter_reduction_factor=16, ordered_cl_tasks=tasks)
    handler = AdapterHandler("vanilla", args)
    handler.add_adapters_to_model(model)
    handler.activate_adapter_for_training(task_key="vqa", model=model)
    assert model.get_active_adapters() == "vqa"
    n_ad = sum(p.numel() for n, p in model.named_parameters() if ".adapters.vqa." in n)
    assert n_ad == 12 * 2 * (768 * 48 * 2 + 48 + 768)                      # ~1.8 M trainable per task
    # make the adapters non-trivial and deterministic, then mirror the weights into the oracle
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if ".adapters." in n:
                p.copy_((torch.randn(p.shape, generator=g) * (0.05 if n.endswith("weight") else 0.02)).to(p.device))
    P = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    assert all((".adapters.vqa." in n) or n.startswith("task_layer.") for n in trainable)
    B = 2
    enc = vo.synthetic_encodings(B, seed=1)
    target = vo.synthetic_vqa_targets(B, seed=1)
    images, texts = enc_to_inputs(enc)
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target)
    o_loss, (o_pooled, o_logits), _, oG = vo.train_step(P, "vqa", enc, target, trainable=trainable, adapter="vqa")
    _close(pooled, o_pooled, tol, "pooled")
    _close(logits, o_logits, tol, "logits")
    _close(loss, o_loss, tol, "loss")
    G = grads_of(model)
    assert set(G) == set(oG), set(G) ^ set(oG)                              # frozen base: no gradients at all
    worst = 0.0
    for n in oG:
        if precision in PARITY_MODES:
            worst = max(worst, _close(G[n], oG[n], tol, n))
        else:
            worst = max(worst, abs(float(G[n].double().norm()) - float(oG[n].double().norm())) / (float(oG[n].double().norm()) + 1e-30))
    print(f"adapters[{precision}]: worst gradient error {worst:.2e}")
    assert worst < (tol if precision in PARITY_MODES else 6e-2)
    # optimizer step touches adapters + the vqa head only
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    opt.step()
    opt.zero_grad()
    for n, p in model.named_parameters():
        changed = not torch.equal(p.detach(), before[n])
        assert changed == ((".adapters.vqa." in n) or n.startswith("task_layer.vqa.")), n
    # switching the active adapter changes the function; switching back restores it (no interference between tasks)
    model.eval()
    with torch.no_grad():
        p_vqa, l_vqa = (t.clone() for t in model(task_key="vqa", images=images, texts=texts))
        handler.activate_adapter_for_eval("nlvr2", model)
        l_other = model(task_key="vqa", images=images, texts=texts)[1].clone()
        handler.activate_adapter_for_eval("vqa", model)
        p_back, l_back = (t.clone() for t in model(task_key="vqa", images=images, texts=texts))
    # fp32 mode: bit-reproducible; bf16 mode: the pooler / head GEMMs use split-K atomics (sum order varies in the last bits)
    if precision in PARITY_MODES:
        assert torch.equal(p_vqa, p_back) and torch.equal(l_vqa, l_back)
    assert torch.allclose(p_vqa, p_back, rtol=1e-5, atol=1e-6) and torch.allclose(l_vqa, l_back, rtol=1e-5, atol=1e-5)
    assert not torch.allclose(l_vqa, l_other)


# ------------------------------------------------------------------------------------------------ ViLT-BERT (row F4)
@pytest.mark.parametrize("precision,tol", [(p_, TOL) for p_ in PARITY_MODES] + [(H16, 6e-2)])      # bf16: 24 layers of bf16 operands instead of 12 (measured: pooled 4.9e-2)
def test_viltbert_vs_reference(golden_dir, precision, tol):
    """REF/modeling/viltbert.py: a frozen BERT-base's last hidden state replaces ViLT's word-embedding lookup.  Golden = the reference's
    own ViltBertContinualLearner (eval mode) on seeded weights; the BERT features are also checked against the CPU oracle."""
    from oracle import bert_oracle as bo
    from climb_amd.modeling import create_continual_learner_map
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.configs.model_configs import model_configs
    z = np.load(os.path.join(golden_dir, "viltbert_vqa_b3.npz"))
    m = _meta(z)
    tasks, B = m["tasks"].split(","), int(m["B"])
    dev = _dev()
    model = create_continual_learner_map["viltbert"](model_name_or_path="random-init:0", ordered_cl_tasks=tasks, model_config=model_configs["viltbert"],
                                                     task_configs=task_configs, device=dev, precision=precision)
    P, PB = vo.init_params(tasks, int(m["wseed"])), bo.init_bert_params(int(m["bseed"]))
    sd = {k.replace("vilt_encoder.", "viltbert_encoder."): v for k, v in P.items()}
    sd.update({"viltbert_encoder.bert." + k: v for k, v in PB.items()})
    model.load_state_dict(sd, strict=True)
    model.to(dev)
    assert list(model.state_dict().keys())[0].startswith("viltbert_encoder.vilt.") and model.get_encoder() is model.viltbert_encoder
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]), ragged_text=True)
    target = vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))
    images, texts = enc_to_inputs(enc)
    # the frozen text features
    model.eval()                                           # this fixture is eval mode; train mode (BERT's dropouts live) is the next test
    dtexts = {k: v.to(dev) for k, v in texts.items()}
    feats = model.get_encoder().get_bert_outputs(**dtexts)[:, :enc["input_ids"].shape[1]].float().cpu()
    with torch.no_grad():
        ofeats = bo.bert_forward(PB, enc["input_ids"], enc["token_type_ids"], enc["attention_mask"])
    valid = enc["attention_mask"].bool()
    _close(feats[valid], ofeats[valid], tol, "BERT last_hidden_state (valid tokens) vs oracle")
    _close(feats[valid][:, :8], torch.from_numpy(z["bert_feats_head"])[valid], tol, "BERT features vs reference")
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target)
    _close(pooled, z["pooled"], tol, "pooled vs reference")
    _close(logits, z["logits"], tol, "logits vs reference")
    _close(loss, z["loss"], tol, "loss vs reference")
    G = {n.replace("viltbert_encoder.", "vilt_encoder."): g for n, g in grads_of(model).items()}
    names = [str(n) for n in z["grad_names"]]
    assert set(G) == set(names), set(G) ^ set(names)       # no gradient for BERT, none for the bypassed word-embedding table
    norms, heads = _summary(G, names)
    if precision in PARITY_MODES:
        assert np.array_equal(logits.argmax(-1).cpu().numpy(), z["logits"].argmax(-1))
        _close(norms, z["grad_norms"], tol, "grad norms vs reference")
        _close(heads, z["grad_heads"], tol, "grad heads vs reference")
    else:
        big = z["grad_norms"] > 1e-3 * z["grad_norms"].max()
        assert (np.abs(norms - z["grad_norms"])[big] / z["grad_norms"][big]).max() < 6e-2
    # an optimizer step moves ViLT and the head, not BERT and not the bypassed table
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    opt.step()
    opt.zero_grad()
    word = "viltbert_encoder.vilt.embeddings.text_embeddings.word_embeddings.weight"
    for n, p in model.named_parameters():
        changed = not torch.equal(p.detach(), before[n])
        expect = not (".bert." in n or n == word or n.startswith("task_layer.nlvr2."))
        assert changed == expect, n
    # reference-style call path: model(task_key=..., images=..., texts=...) -> (pooled, logits)
    with torch.no_grad():
        out = model(task_key="vqa", images=images, texts=texts)
    assert out[1].shape == (B, 3129) and bool(torch.isfinite(out[1]).all())


@pytest.mark.parametrize("precision,tol", [("fp32", TOL), (H16, 8e-2)])
def test_viltbert_train_mode_with_the_references_bert_dropout_masks(golden_dir, precision, tol):
    """VERDICT r2 missing #4, second half.  REF/modeling/viltbert.py:115-120 never puts the frozen BERT in eval mode: while the learner
    trains, BERT's 37 dropouts are live.  `viltbert_vqa_b3_train.npz` is the reference's own train-mode step with every mask it drew
    recorded; the HIP BERT (probability dropout inside `climb_attn_fwd_dropout`, row dropouts between the GEMMs and their residual adds)
    is given those masks and must reproduce the reference's features, logits, loss, argmax and gradients.  Without masks, train mode draws
    its own (different features every call, eval mode deterministic)."""
    from oracle import bert_oracle as bo
    from tests.test_oracle_golden import unpack_bert_masks
    from climb_amd.modeling import create_continual_learner_map
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.configs.model_configs import model_configs
    z = np.load(os.path.join(golden_dir, "viltbert_vqa_b3_train.npz"))
    m = _meta(z)
    tasks, B, T = m["tasks"].split(","), int(m["B"]), int(m["T"])
    dev = _dev()
    model = create_continual_learner_map["viltbert"](model_name_or_path="random-init:0", ordered_cl_tasks=tasks, model_config=model_configs["viltbert"],
                                                     task_configs=task_configs, device=dev, precision=precision)
    P, PB = vo.init_params(tasks, int(m["wseed"])), bo.init_bert_params(int(m["bseed"]))
    sd = {k.replace("vilt_encoder.", "viltbert_encoder."): v for k, v in P.items()}
    sd.update({"viltbert_encoder.bert." + k: v for k, v in PB.items()})
    model.load_state_dict(sd, strict=True)
    model.to(dev)
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]), ragged_text=True)
    target = vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))
    images, texts = enc_to_inputs(enc)
    masks = unpack_bert_masks(z, B, T)
    valid = enc["attention_mask"].bool()
    encw = model.get_encoder()
    dtexts = {k: v.to(dev) for k, v in texts.items()}
    model.train()
    assert encw.bert.training
    encw.bert_dropout_masks = masks
    feats = encw.get_bert_outputs(**dtexts)[:, :T].float().cpu()
    _close(feats[valid][:, :8], torch.from_numpy(z["bert_feats_head"])[valid], tol, "BERT features (train mode, the reference's masks) vs reference")
    with torch.no_grad():
        ofeats = bo.bert_forward(PB, enc["input_ids"], enc["token_type_ids"], enc["attention_mask"], masks=masks)
    _close(feats[valid], ofeats[valid], tol, "BERT features vs oracle")
    encw.bert_dropout_masks = masks
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target)
    assert encw.bert_dropout_masks is None                 # consumed by that step
    _close(pooled, z["pooled"], tol, "pooled vs reference")
    _close(logits, z["logits"], tol, "logits vs reference")
    _close(loss, z["loss"], tol, "loss vs reference")
    G = {n.replace("viltbert_encoder.", "vilt_encoder."): g for n, g in grads_of(model).items()}
    names = [str(n) for n in z["grad_names"]]
    assert set(G) == set(names)
    norms, heads = _summary(G, names)
    if precision in PARITY_MODES:
        assert np.array_equal(logits.argmax(-1).cpu().numpy(), z["logits"].argmax(-1))
        _close(norms, z["grad_norms"], tol, "grad norms vs reference")
        _close(heads, z["grad_heads"], tol, "grad heads vs reference")
    else:
        big = z["grad_norms"] > 1e-3 * z["grad_norms"].max()
        assert (np.abs(norms - z["grad_norms"])[big] / z["grad_norms"][big]).max() < 8e-2
    # own masks: train mode is stochastic with the reference's statistics, eval mode deterministic
    a = encw.get_bert_outputs(**dtexts)[:, :T].float().cpu().clone()
    b = encw.get_bert_outputs(**dtexts)[:, :T].float().cpu().clone()
    model.eval()
    e1 = encw.get_bert_outputs(**dtexts)[:, :T].float().cpu().clone()
    e2 = encw.get_bert_outputs(**dtexts)[:, :T].float().cpu().clone()
    assert torch.equal(e1, e2) and not torch.equal(a, b)
    dev_rel = float((a[valid] - e1[valid]).norm() / e1[valid].norm())
    ref = json.load(open(os.path.join(golden_dir, "viltbert_train_dropout.json")))["mean_feature_rel_rms"]
    print(f"train-mode features vs eval-mode: {dev_rel:.3f} relative rms (reference, mean of 8 seeds: {ref:.3f})")
    assert 0.7 * ref < dev_rel < 1.4 * ref


# ------------------------------------------------------------------------------------------------ full size vs the REFERENCE
def _full_size_inputs(z):
    """Rebuild the seeded inputs of a tests/golden/*_b{64,32,16}.npz fixture (oracle/gen_golden.py::case_fullsize)."""
    m = _meta(z)
    task = m["task"]
    if task == "vqa":
        B = int(m["B"])
        enc = vo.synthetic_encodings(B, seed=int(m["dseed"]))
        images, texts = enc_to_inputs(enc)
        return task, images, texts, vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))
    b = int(m["b"])
    if task == "nlvr2":
        e1 = vo.synthetic_encodings(2 * b, seed=int(m["dseed"]))
        texts = dict(input_ids=e1["input_ids"][:b], token_type_ids=e1["token_type_ids"][:b], attention_mask=e1["attention_mask"][:b])
        return task, e1["pixel_values"], texts, torch.from_numpy(z["labels"])
    e1 = vo.synthetic_encodings(4 * b, seed=int(m["dseed"]), ragged_text=True)
    texts = dict(input_ids=e1["input_ids"], token_type_ids=e1["token_type_ids"], attention_mask=e1["attention_mask"])
    return task, e1["pixel_values"][:b], texts, torch.from_numpy(z["labels"])


def full_size_errors(z, precision):
    """One step of the HIP path on a full-size reference fixture; relative errors (max|d| / max|ref|) against the reference's own
    outputs and the fraction of rows whose argmax prediction equals the reference's.  Also used by bench.py (`bf16_vs_ref`)."""
    m = _meta(z)
    task, images, texts, target = _full_size_inputs(z)
    model, _ = make_model(m["tasks"].split(","), int(m["wseed"]), precision=precision)
    model.train() if task != "vcr" else model.eval()          # the VCR fixture is eval mode (head dropout is the only RNG on the path)
    loss, (pooled, logits), _, _ = model.fused_forward_backward(task, images, texts, target)

    def rel(a, b):
        a = torch.as_tensor(np.asarray(a.detach().float().cpu())).double()
        b = torch.as_tensor(np.asarray(b)).double()
        return float((a - b).abs().max() / (b.abs().max() + 1e-30))
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    big = z["grad_norms"] > 1e-3 * z["grad_norms"].max()
    gn = np.abs(norms - z["grad_norms"])[big] / z["grad_norms"][big]
    out = dict(pooled=rel(pooled, z["pooled"]), logits=rel(logits, z["logits"]), loss=rel(loss, z["loss"]),
               argmax_agreement=float((logits.argmax(-1).cpu().numpy() == z["logits"].argmax(-1)).mean()), rows=int(logits.shape[0]),
               grad_norm_median=float(np.median(gn)), grad_norm_max=float(gn.max()),
               grad_heads=rel(torch.from_numpy(heads), z["grad_heads"]))
    del model
    torch.cuda.empty_cache()
    return out


FULL_SIZE = ["vqa_b64.npz", "nlvr2_b32.npz", "vcr_b16.npz"]


@pytest.mark.parametrize("precision", PARITY_MODES)
@pytest.mark.parametrize("fname", FULL_SIZE)
def test_full_size_fp32_vs_reference(golden_dir, fname, precision):
    """BASELINE.json configs[1] at ITS OWN size (64 sequences x (40 tokens + 384x384)) and the equal-sized NLVR2 (32 pairs) / VCR
    (16 x 4 choices) batches: the HIP path in the parity mode against what the reference's `*Trainer.train_step` computed
    (REF train_vqa.py:135-174).  These sizes select the 192x192 three-stage, 256x256 and one-round split kernels."""
    e = full_size_errors(np.load(os.path.join(golden_dir, fname)), precision)
    print(f"{fname} {precision} vs reference: {e}")
    assert e["pooled"] <= TOL and e["logits"] <= TOL and e["loss"] <= TOL
    assert e["argmax_agreement"] == 1.0, "argmax task predictions must be bit-exact on every row"
    assert e["grad_norm_max"] <= TOL and e["grad_heads"] <= TOL


# error budget of the throughput mode at the benchmark's size, measured against the REFERENCE (DESIGN.md section 3 tabulates where it comes from)
# r06: measured x 1.3 per fixture (a 2x regression used to pass silently).  Measured on the final r06 tree (gpurun_out/r06_gputest1.log):
#   vqa_b64  pooled 2.24e-2 logits 5.96e-3 loss 4.0e-5 grad norms 5.3e-3 argmax 63 / 64;  nlvr2_b32  2.43e-2 7.07e-3 7.8e-4 3.8e-3 32 / 32;  vcr_b16  2.33e-2 7.45e-3 6.0e-4 3.3e-3 16 / 16
BF16_FULL = {"vqa_b64.npz": dict(pooled=2.9e-2, logits=7.8e-3, loss=1e-4, grad_norm_max=7e-3),
             "nlvr2_b32.npz": dict(pooled=3.2e-2, logits=9.2e-3, loss=1.1e-3, grad_norm_max=5e-3),
             # (vcr: the one fixture whose 16-bit step is not bit-reproducible -- its M = 64-row head GEMMs take the split-K path, whose fp32 atomics order the 16-bit
             # roundings downstream differently from run to run (tools/probe/vcr_determinism.py; the fp32 mode forbids itself split-K and repeats bit for bit):
             # the worst gradient-norm error moved between 3.2e-3 and 4.9e-3 over 12 runs (tools/probe/vcr_full_repeat.py) -- limit = the largest seen x 1.3)
             "vcr_b16.npz": dict(pooled=3.1e-2, logits=9.7e-3, loss=8e-4, grad_norm_max=6.4e-3)}


@pytest.mark.parametrize("fname", FULL_SIZE)
def test_full_size_bf16_vs_reference(golden_dir, fname):
    e = full_size_errors(np.load(os.path.join(golden_dir, fname)), H16)
    print(f"{fname} bf16 vs reference: {e}")
    for k, lim in BF16_FULL[fname].items():
        assert e[k] <= lim, (k, e[k], lim)
    # random-init heads put many top-2 logits closer than bf16's error (the smallest margin of vqa_b64 is 4e-4 of the logit scale):
    # the agreement is REPORTED (bench.py prints it) and bounded from below, it cannot be 100 % in this mode
    assert e["argmax_agreement"] >= 0.95          # (measured: 63 / 64 = 0.984 on vqa_b64, 1.0 on the other two)


# ------------------------------------------------------------------------------------------------ full size (bs = 64) properties
def _rand_batch(B, seed, dev):
    g = torch.Generator().manual_seed(seed)
    texts = dict(input_ids=torch.randint(0, 30522, (B, 40), generator=g), token_type_ids=torch.zeros(B, 40, dtype=torch.long),
                 attention_mask=torch.ones(B, 40, dtype=torch.long))
    lens = torch.randint(5, 41, (B,), generator=g)
    for b in range(B):
        texts["attention_mask"][b, lens[b]:] = 0
    pixels = torch.randn(B, 3, 384, 384, generator=g)
    target = torch.zeros(B, 3129)
    target[torch.arange(B), torch.randint(0, 3129, (B,), generator=g)] = 1.0
    return pixels.to(dev), {k: v.to(dev) for k, v in texts.items()}, target.to(dev)


@pytest.mark.parametrize("config", ["ewc", "adapters_nlvr2", "replay_vcr"])
def test_continual_learning_configs_at_the_benchmarks_64_sequences(config):
    """BASELINE.json configs[2] / [3] / [4] at the benchmark's size on one GPU (only B = 2 fixtures and batch-4 driver
    runs exercised them): 64 encoder sequences per step through the throughput (16-bit) mode and the fp32 parity mode (itself pinned to the
    reference by the B = 2 fixtures of the same code paths) -- the EWC-penalised VQA step, the NLVR2 step under an active Houlsby adapter
    (32 pairs), and an experience-replay step of VCR (16 x 4 choices, fresh optimizer, eval-mode head so that no dropout mask differs)."""
    from climb_amd.cl_algorithms import AdapterHandler, EWC
    dev = _dev()
    tasks = ["vqa", "nlvr2", "vcr"]
    out = {}
    for precision in ("fp32", H16):
        torch.manual_seed(0)
        random.seed(0)
        model, P = make_model(tasks, 42, precision=precision)
        model.train()
        if config == "ewc":
            pixels, texts, target = _rand_batch(64, 501, dev)
            ewc = EWC(types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=100.0))
            fisher, star = _ewc_state(P, 5)
            ewc.set_task_state("nlvr2", model, fisher, star)
            loss, (_, logits), ewc_task, ewc_loss = model.fused_forward_backward("vqa", pixels, texts, target, ewc)
            assert ewc_task == "nlvr2"
            extra = float(ewc_loss)
        elif config == "adapters_nlvr2":
            args = types.SimpleNamespace(adapter_config="houlsby", adapter_reduction_factor=16, ordered_cl_tasks=tasks)
            handler = AdapterHandler("vanilla", args)
            handler.add_adapters_to_model(model)
            handler.activate_adapter_for_training(task_key="nlvr2", model=model)
            g = torch.Generator().manual_seed(9)
            with torch.no_grad():
                for n, p in model.named_parameters():
                    if ".adapters." in n:
                        p.copy_((torch.randn(p.shape, generator=g) * (0.05 if n.endswith("weight") else 0.02)).to(p.device))
            pixels, texts, _ = _rand_batch(64, 502, dev)
            texts = {k: v[:32] for k, v in texts.items()}
            labels = torch.randint(0, 2, (32,), generator=torch.Generator().manual_seed(3))
            loss, (_, logits), _, _ = model.fused_forward_backward("nlvr2", pixels, texts, labels)
            extra = 0.0
            assert all((".adapters.nlvr2." in n) or n.startswith("task_layer.") for n in grads_of(model))
        else:
            pixels, texts, _ = _rand_batch(64, 503, dev)
            labels = torch.randint(0, 4, (16,), generator=torch.Generator().manual_seed(4))
            model.eval()
            opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})        # replay: a fresh AdamW (REF experience_replay.py:61)
            loss, (_, logits), _, _ = model.fused_forward_backward("vcr", pixels[:16], texts, labels)
            extra = 0.0
        G = {n: float(g.double().norm()) for n, g in grads_of(model).items()}
        if config == "replay_vcr":
            before = model.get_encoder().vilt.encoder.layer[5].output.dense.weight.detach().clone()
            opt.step()
            assert not torch.equal(before, model.get_encoder().vilt.encoder.layer[5].output.dense.weight.detach())
        out[precision] = (float(loss), logits.detach().float().cpu(), G, extra)
        assert bool(torch.isfinite(logits).all())
        del model
        torch.cuda.empty_cache()
    l32, lg32, g32, x32 = out["fp32"]
    l16, lg16, g16, x16 = out[H16]
    assert abs(l16 - l32) <= 3e-3 * abs(l32) and abs(x16 - x32) <= 1e-4 * max(1.0, abs(x32))       # the EWC term is fp32 in both modes
    _close(lg16, lg32, BF16_TOL, f"logits {H16} vs fp32 ({config})")
    assert set(g16) == set(g32)
    top = max(g32.values())
    worst = max(abs(g16[n] - v) / v for n, v in g32.items() if v > 1e-3 * top)
    print(f"{config} at 64 sequences: loss {l32:.4f} / {l16:.4f}, worst gradient-norm difference {worst:.2e}")
    assert worst <= 6e-2


@pytest.mark.parametrize("config", ["plain", "ewc", "frozen9", "accumulate"])
def test_optimizer_in_the_weight_gradient_epilogue_is_the_same_training_step(config):
    """r04: when the caller names its optimizer -- `fused_forward_backward(..., optimizer=opt)`, what the trainers' train_step does;
    REF/train/visionlanguage_tasks/train_vqa.py:160-170 calls step() right after backward() -- the encoder's grouped weight-gradient launch is held back
    and run by FusedAdamW.step() with AdamW in its epilogue: p, m, v, the 16-bit shadow and the transposed shadow written from the tile sums, the
    gradient never stored, the flat optimizer pass and the shadow transposes skipping those tensors.  The kernel itself is pinned bit for bit in
    tests/test_gpu_kernels.py; a whole 16-bit step is not bit-reproducible from run to run (atomics in the heads' split-K, in the bias gradients and in
    the stream-K tail), so here: three steps at the benchmark's 64 sequences with and without, also with the EWC term in the gradient buffer
    (REF/cl_algorithms/ewc.py:75-87), with the bottom 9 layers frozen (other plans / tails) and with a second backward before the step (the held
    launch must run as the plain one) -- the fused launch really ran (or did not), the loss curves agree, and after the last step BOTH shadows are
    exactly the 16-bit image / transpose of the fp32 parameters (a tensor the epilogue updated but whose shadows it missed would show)."""
    if H16 != "bf16":
        pytest.skip("the fused update is the bf16 build's (the half build scales its gradients)")
    from climb_amd.cl_algorithms import EWC
    dev = _dev()
    res = {}
    for fused in (False, True):
        torch.manual_seed(0)
        random.seed(0)
        model, P = make_model(["vqa", "nlvr2"], 42, precision=H16)
        model.train()
        if config == "frozen9":
            model.get_encoder().freeze_bottom_k_layers(9)
        ewc = None
        if config == "ewc":
            ewc = EWC(types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=100.0))
            fisher, star = _ewc_state(P, 5)
            ewc.set_task_state("nlvr2", model, fisher, star)
        opt = model.create_optimizer({"lr": 2e-5, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
        opt.zero_grad()
        eng = model._host.engine()
        w0 = model.get_encoder().vilt.encoder.layer[10].intermediate.dense.weight.detach().clone()
        launches = []
        orig = eng._timed_call

        def spy(kernel, flops, name, *a, _o=orig, _l=launches):
            _l.append(name)
            return _o(kernel, flops, name, *a)
        eng._timed_call = spy
        losses = []
        for step in range(3):
            pixels, texts, target = _rand_batch(64, 900 + step, dev)
            if config == "accumulate" and step == 1:          # two backwards, one step: the first one's held launch must become a plain one
                model.fused_forward_backward("vqa", pixels, texts, target, ewc, optimizer=opt if fused else None)
                pixels, texts, target = _rand_batch(64, 950, dev)
            loss, _, _, _ = model.fused_forward_backward("vqa", pixels, texts, target, ewc, optimizer=opt if fused else None)
            opt.step()
            # the flat pass clears what it consumes when that leaves the whole buffer zero (no EWC term parked in it): zero_grad() then skips its fill
            # (r05: the EWC term no longer parks anything in the buffer either -- FusedAdamW.step() adds it inside its passes)
            expect_clean = fused and not (config == "accumulate" and step == 1)
            assert eng._grad_clean == expect_clean, (config, fused, step)
            if expect_clean:
                assert not bool(eng.grad.any()), "the step claimed a zero gradient buffer"
            opt.zero_grad()
            assert not bool(eng.grad.any()) and not eng._grad_clean
            losses.append(float(loss))
        eng.refresh_shadow()
        torch.cuda.synchronize()
        n_fused = sum(1 for n in launches if n in ("climb_gemm_bf16_tn_grouped_adamw", "climb_gemm_bf16_tn_grouped_adamw_ewc"))
        n_plain = sum(1 for n in launches if n == "climb_gemm_bf16_tn_grouped")
        if config == "ewc" and fused and os.environ.get("CLIMB_AMD_EWC_FOLD", "2") == "2":
            # (r05) the EWC term rides in the flat optimizer pass and the weight gradients are written by the plain launch: measured faster than
            # carrying two more operands through the (exposed) optimizer epilogue of the weight-gradient launch, tools/ewc_ab.py
            assert (n_fused, n_plain) == (0, 3), launches
        else:
            assert (n_fused, n_plain) == ((3, 1 if config == "accumulate" else 0) if fused else (0, 4 if config == "accumulate" else 3)), launches
        assert not torch.equal(w0, model.get_encoder().vilt.encoder.layer[10].intermediate.dense.weight.detach())
        assert bool(torch.isfinite(eng.flat).all()) and bool(torch.isfinite(opt._m).all()) and bool(torch.isfinite(opt._v).all())
        assert bool((opt._v >= 0).all()), "negative second moment (r06: the store-data hazard of tools/check_store_hazard.py would show here first)"
        assert torch.equal(eng._shadow, eng.flat.to(torch.bfloat16)), "16-bit shadow is not the image of the fp32 parameters"
        for name, N, K in eng._linear_weight_names():
            t = eng._shadow_t[eng._t_off[name]:eng._t_off[name] + N * K].view(K, N)
            o = eng.layout.offset[name]
            assert torch.equal(t, eng._shadow[o:o + N * K].view(N, K).t()), f"transposed shadow of {name}"
        res[fused] = losses
        del model, opt
        torch.cuda.empty_cache()
    for a, b in zip(res[False], res[True]):
        assert abs(a - b) <= 5e-3 * abs(a), (res[False], res[True])


def test_ewc_term_folded_into_the_optimizer_passes_is_the_same_update():
    """r05: with the optimizer named, the EWC term (REF/cl_algorithms/ewc.py:75-87) is not written by a pass of its own: the
    weight-gradient epilogue and the flat AdamW pass add 2 lam F (theta - theta*) to the gradient they consume and accumulate lam sum F (theta - theta*)^2.
    Checked on one step from zero moments, where AdamW's first moment IS the gradient (m = (1 - beta1) g): (a) the penalty's value equals the float64
    sum over the PRE-step parameters; (b) m of the folded step minus m of the same step without EWC equals the analytic term, tensor by tensor --
    matrices (updated in the weight-gradient epilogue), vectors and embeddings (flat pass) alike; (c) the unfolded path (CLIMB_AMD_EWC_FOLD=0) gives the
    same value and moments; (d) the step leaves the gradient buffer clean (nothing was parked in it)."""
    if H16 != "bf16":
        pytest.skip("the fused update is the bf16 build's")
    from climb_amd.cl_algorithms import EWC
    dev = _dev()
    lam = 100.0
    runs = {}
    for mode in ("none", "fold", "fold_flat", "unfold"):
        model, P = make_model(["vqa", "nlvr2"], 42, precision=H16)
        model.train()
        ewc = None
        if mode != "none":
            ewc = EWC(types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=lam))
            fisher, star = _ewc_state(P, 5)
            ewc.set_task_state("nlvr2", model, fisher, star)
        opt = model.create_optimizer({"lr": 2e-5, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
        opt.zero_grad()
        eng = model._host.engine()
        pixels, texts, target = _rand_batch(8, 900, dev)
        os.environ["CLIMB_AMD_EWC_FOLD"] = {"unfold": "0", "fold": "1", "fold_flat": "2", "none": "2"}[mode]
        try:
            random.seed(0)
            _, _, _, ewc_loss = model.fused_forward_backward("vqa", pixels, texts, target, ewc, optimizer=opt)
            if mode in ("fold", "fold_flat"):
                assert eng._ewc_fold is not None, "the term was not parked for the optimizer"
            opt.step()
        finally:
            os.environ.pop("CLIMB_AMD_EWC_FOLD", None)
        assert eng._ewc_fold is None
        if mode == "fold":
            assert eng._grad_clean and not bool(eng.grad.any()), "the folded step left something in the gradient buffer"
        if mode == "fold_flat":          # (the flat pass clears what it consumes: the weight gradients the plain launch wrote included)
            opt.zero_grad()
            assert not bool(eng.grad.any())
        torch.cuda.synchronize()
        n_enc = eng.layout.encoder_end
        runs[mode] = dict(m=opt._m[:n_enc].detach().double().cpu(), loss=None if ewc_loss is None else float(ewc_loss))
        if mode == "fold":
            F = ewc.fisher_flat["nlvr2"].double().cpu()
            S = ewc.param_flat["nlvr2"].double().cpu()
            runs["F"], runs["S"] = F, S
            runs["segs"] = [(n, eng.layout.offset[n], eng.layout.numel(n)) for n in eng.layout.shapes if eng.layout.offset[n] < n_enc]
            # the parameters BEFORE the step, in the engine's flat order (64-element aligned tensors: gaps are zero in theta, theta* and F alike)
            flat0 = torch.zeros(n_enc, dtype=torch.float64)
            for n, o, k in runs["segs"]:
                flat0[o:o + k] = P[n].reshape(-1).double()          # (layout names are the state dict's)
            runs["theta0"] = flat0
        del model, opt
        torch.cuda.empty_cache()
    F, S, th = runs["F"], runs["S"], runs["theta0"]
    value = lam * float((F * (th - S) ** 2).sum())
    assert abs(runs["fold"]["loss"] - value) <= 2e-5 * value, (runs["fold"]["loss"], value)
    assert abs(runs["unfold"]["loss"] - value) <= 2e-5 * value, (runs["unfold"]["loss"], value)
    term = 2.0 * lam * F * (th - S)
    got = (runs["fold"]["m"] - runs["none"]["m"]) / (1.0 - 0.9)
    worst = 0.0
    for n, o, k in runs["segs"]:
        t, g = term[o:o + k], got[o:o + k]
        scale = float(t.abs().max())
        if scale == 0:
            continue
        # the two runs' task gradients differ by the 16-bit step's run-to-run noise (atomics); the term itself is exact fp32 arithmetic
        tol = 2e-2 * float(runs["none"]["m"][o:o + k].abs().max()) / 0.1 + 1e-5 * scale
        err = float((g - t).abs().max())
        assert err <= tol + 1e-12, (n, err, tol, scale)
        worst = max(worst, err / scale)
    assert float((runs["fold"]["m"] - runs["unfold"]["m"]).abs().max()) <= 2e-2 * float(runs["unfold"]["m"].abs().max())
    assert float((runs["fold_flat"]["m"] - runs["unfold"]["m"]).abs().max()) <= 2e-2 * float(runs["unfold"]["m"].abs().max())
    assert abs(runs["fold_flat"]["loss"] - value) <= 2e-5 * value, (runs["fold_flat"]["loss"], value)
    print(f"EWC fold: value {runs['fold']['loss']:.6f} (float64 {value:.6f}); worst term error {worst:.2e} of its tensor's largest element")


def test_held_back_weight_gradients_of_a_step_that_never_came_use_their_own_activations():
    """a weight-gradient launch held back for `optimizer.step()` points at the saved activations of its workspace; when a second
    backward (accumulation) or a no-grad forward of the same shape comes first, it has to run BEFORE that forward overwrites them.  Compared here
    as GRADIENTS (a loss curve at lr 2e-5 cannot see it): two backwards on different batches with the promise, no step -> the accumulated `.grad`
    of every tensor equals the same two backwards without the promise; and a no-grad forward between backward and step changes nothing either."""
    if H16 != "bf16":
        pytest.skip("the held-back launch is the bf16 build's")
    dev = _dev()
    G = {}
    for promise in (False, True):
        model, _ = make_model(["vqa"], 42, precision=H16)
        model.train()
        opt = model.create_optimizer({"lr": 2e-5, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
        opt.zero_grad()
        eng = model._host.engine()
        for seed in (900, 950):
            pixels, texts, target = _rand_batch(8, seed, dev)
            model.fused_forward_backward("vqa", pixels, texts, target, optimizer=opt if promise else None)
        if promise:
            assert eng._dw_deferred, "nothing was held back: the test does not exercise the path"
            with torch.no_grad():          # an evaluation pass of the same shape between backward and step
                model(task_key="vqa", images=_rand_batch(8, 970, dev)[0], texts=texts)
            assert not eng._dw_deferred, "the held-back launch survived a forward that overwrote its operands"
        eng.materialize_dw()
        torch.cuda.synchronize()
        G[promise] = {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        del model, opt
    assert set(G[False]) == set(G[True])
    for n, g in G[False].items():
        scale = float(g.abs().max())
        if scale > 0:
            assert float((G[True][n] - g).abs().max()) <= 2e-2 * scale, n          # run-to-run 16-bit noise (atomics) only; stale operands give O(1)


def test_full_size_batch_permutation_and_mode_agreement():
    """BASELINE.json configs[1] size (64 sequences x 185 tokens), where the CPU oracle would take minutes: size-independent
    properties instead.  (1) samples are independent: permuting the batch permutes pooled/logits and leaves the loss and every
    gradient unchanged (up to fp32 summation order); (2) the bf16 throughput mode agrees with the fp32 parity mode."""
    dev = _dev()
    B = 64
    pixels, texts, target = _rand_batch(B, 11, dev)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).to(dev)
    model, _ = make_model(["vqa"], 42, precision="fp32")
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", pixels, texts, target)
    pooled, logits = pooled.clone(), logits.clone()
    G = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    model._host.drop_grads()
    loss_p, (pooled_p, logits_p), _, _ = model.fused_forward_backward("vqa", pixels[perm], {k: v[perm] for k, v in texts.items()}, target[perm])
    assert torch.equal(pooled_p, pooled[perm]) and torch.equal(logits_p, logits[perm]), "per-sample outputs must not depend on batch position"
    _close(loss_p, loss, 1e-5, "loss under permutation")
    for n, p in model.named_parameters():
        if p.grad is not None and not n.endswith("attention.key.bias"):
            _close(p.grad, G[n], 1e-4, f"grad {n} under permutation")
    assert bool(torch.isfinite(logits).all()) and float(loss) > 0
    # bf16 mode on the same weights / batch
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    del model
    torch.cuda.empty_cache()
    m16, _ = make_model(["vqa"], 42, precision=H16)
    m16.load_state_dict(sd)
    m16.train()
    loss16, (pooled16, logits16), _, _ = m16.fused_forward_backward("vqa", pixels, texts, target)
    _close(loss16, loss, BF16_TOL, "bf16 vs fp32 loss")
    _close(logits16, logits, BF16_TOL, "bf16 vs fp32 logits")
    _close(pooled16, pooled, 5e-2, "bf16 vs fp32 pooled")
    agree = float((logits16.argmax(-1) == logits.argmax(-1)).float().mean())
    print(f"bs=64: bf16/fp32 argmax agreement {agree:.3f}")
    gq = "vilt_encoder.vilt.encoder.layer.0.attention.attention.query.weight"
    g16 = dict(m16.named_parameters())[gq].grad
    rel = float((g16.double().cpu().norm() - G[gq].double().cpu().norm()).abs() / G[gq].double().cpu().norm())
    assert rel < 3e-2, rel


# ------------------------------------------------------------------------------------------------ last layer on its [CLS] rows only
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), (H16, 2e-2)])
@pytest.mark.parametrize("B,frozen", [(64, 0), (3, 0), (5, 11)])
def test_last_layer_on_cls_rows_only_equals_every_row(precision, tol, B, frozen):
    """`ViltEngine.cls_only_last` (opt-in, CLIMB_AMD_CLS_ONLY_LAST=1): after the last layer's attention only the B [CLS] rows go through the out-projection, the MLP
    and the final LayerNorm, forward and backward -- no other row of x_L has a reader (REF/modeling/vilt.py:123-124 returns pooler_output
    alone).  The step must be the one that computes every row like the reference does: same pooled / logits / loss and the same gradient of
    EVERY parameter (fp32: summation order only; 16-bit: the B-row GEMMs run on another kernel, so operand roundings land differently).
    `frozen` = layers frozen from the bottom (11: the pruned layer is the only one with gradients)."""
    dev = _dev()
    pixels, texts, target = _rand_batch(B, 17 + B, dev)
    model, _ = make_model(["vqa"], 42, precision=precision)
    if frozen:
        model.get_encoder().freeze_bottom_k_layers(frozen)
    model.train()
    eng = model._host.engine()
    assert not eng.cls_only_last, "the default step computes every row"
    out = {}
    for mode in (True, False):
        eng.cls_only_last = mode
        model._host.drop_grads()
        loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", pixels, texts, target)
        out[mode] = (float(loss), pooled.clone(), logits.clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    eng.cls_only_last = False
    (l1, p1, g1, G1), (l0, p0, g0, G0) = out[True], out[False]
    assert abs(l1 - l0) <= tol * abs(l0), (l1, l0)
    _close(p1, p0, tol, "pooled")
    _close(g1, g0, tol, "logits")
    assert set(G1) == set(G0) and len(G0) > 4
    # (the key bias shifts every score of a row alike, softmax cancels it: its gradient is rounding noise around zero in both runs)
    worst = max((_close(G1[n], G0[n], tol * (5 if precision != "fp32" else 1), f"grad {n}"), n) for n in G0 if not n.endswith("attention.key.bias"))
    print(f"{precision} B={B} frozen={frozen}: {len(G0)} gradients, worst relative difference {worst[0]:.2e} ({worst[1]})")


# ------------------------------------------------------------------------------------------------ variable resolution (row F2)
@pytest.mark.parametrize("fixture", ["vqa_b4_varres.npz", "vqa_b16_mixed.npz"])
@pytest.mark.parametrize("precision,tol", [(p_, TOL) for p_ in PARITY_MODES] + [(H16, BF16_TOL)])
def test_variable_resolution_batch_vs_reference(golden_dir, precision, tol, fixture):
    """Padded variable-resolution batch (HF:92-178 masked visual_embed with per-sample bilinear position resize): the HIP path keeps
    every canvas patch in raster order and masks the invalid ones; the reference shuffles and pads randomly.  Pooled output,
    logits, loss and all gradients must agree (golden = the reference's own run).  `vqa_b16_mixed` (r03): 16 COCO-like images of BOTH
    orientations -- a 640 x 640 canvas of 400 patches of which no image fills more than 240, so the engine PACKS each sample's valid patches
    (261-token sequences, S_pad = 288: the shape bench.py's real_input leg trains on), against the reference's own run of that batch."""
    z = np.load(os.path.join(golden_dir, fixture))
    m = _meta(z)
    sizes = [tuple(int(v) for v in r) for r in z["sizes"]]
    B = len(sizes)
    model, P = make_model(m["tasks"].split(","), int(m["wseed"]), precision=precision)
    enc = vo.synthetic_varres_encodings(sizes, seed=int(m["dseed"]))
    target = vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))
    texts = dict(input_ids=enc["input_ids"], token_type_ids=enc["token_type_ids"], attention_mask=enc["attention_mask"])
    images = dict(pixel_values=enc["pixel_values"], pixel_mask=enc["pixel_mask"])
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target)
    _close(pooled, z["pooled"], tol, "pooled vs reference")
    _close(logits, z["logits"], tol, "logits vs reference")
    _close(loss, z["loss"], tol, "loss vs reference")
    G = grads_of(model)
    names = [str(n) for n in z["grad_names"]]
    norms, heads = _summary(G, names)
    if precision in PARITY_MODES:
        assert np.array_equal(logits.argmax(-1).cpu().numpy(), z["logits"].argmax(-1))
        _close(norms, z["grad_norms"], tol, "grad norms vs reference")
        _close(heads, z["grad_heads"], tol, "grad heads vs reference")
        if B <= 4:          # (the oracle's per-element gradients: seconds at B = 4, most of a minute at 16)
            _, _, _, oG = vo.train_step(P, "vqa", enc, target)
            for n in (vo.ENC + "embeddings.position_embeddings", vo.ENC + "embeddings.patch_embeddings.projection.weight", vo.ENC + "embeddings.cls_token",
                      vo.ENC + "embeddings.token_type_embeddings.weight", vo.ENC + "encoder.layer.0.attention.attention.value.weight"):
                _close(G[n], oG[n], tol, n)          # incl. the transpose of the bilinear position resize
        else:
            ws = model._host._engine.last_ws
            assert ws.compact and ws.NP == 400 and ws.NS == 220 and ws.S_pad == 288          # 11 x 20 patches at most, packed
    else:
        big = z["grad_norms"] > 1e-3 * z["grad_norms"].max()
        assert (np.abs(norms - z["grad_norms"])[big] / z["grad_norms"][big]).max() < 6e-2
    # the autograd-facing path accepts the same encodings
    model.eval()
    with torch.no_grad():
        out = model(task_key="vqa", images=images, texts=texts)
    _close(out[1], z["logits"], tol, "eval logits")


def test_maximum_sequence_384x640_vs_oracle():
    """The largest input the reference's processor can emit (shortest edge 384, longest capped at 640: 12 x 20 patches, 40 + 1 + 240
    = 281 tokens, S_pad 288) next to a small 32-pixel-high strip and 3-token texts: step output and gradients against the oracle."""
    sizes = [(384, 640), (32, 608)]
    model, P = make_model(["vqa"], 7, precision="fp32")
    enc = vo.synthetic_varres_encodings(sizes, seed=5)
    target = vo.synthetic_vqa_targets(len(sizes), seed=5)
    texts = dict(input_ids=enc["input_ids"], token_type_ids=enc["token_type_ids"], attention_mask=enc["attention_mask"])
    images = dict(pixel_values=enc["pixel_values"], pixel_mask=enc["pixel_mask"])
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target)
    oloss, (opooled, ologits), _, oG = vo.train_step(P, "vqa", enc, target)
    _close(pooled, opooled, TOL, "pooled")
    _close(logits, ologits, TOL, "logits")
    _close(loss, oloss, TOL, "loss")
    assert torch.equal(logits.argmax(-1).cpu(), ologits.argmax(-1))
    G = grads_of(model)
    for n, g in oG.items():
        if n in G and not n.endswith("attention.key.bias"):
            _close(G[n], g, TOL, n)
    # one patch more than the attention tiles are sized for is refused loudly, not truncated
    with pytest.raises(NotImplementedError):
        big = vo.synthetic_varres_encodings([(384, 672)], seed=5)
        model.fused_forward_backward("vqa", dict(pixel_values=big["pixel_values"], pixel_mask=big["pixel_mask"]),
                                     dict(input_ids=big["input_ids"], token_type_ids=big["token_type_ids"], attention_mask=big["attention_mask"]),
                                     vo.synthetic_vqa_targets(1, seed=5))


@pytest.mark.parametrize("precision,tol", [("fp32", TOL), (H16, BF16_TOL)])
def test_mixed_orientation_batch_packs_valid_patches(precision, tol):
    """A portrait image next to a landscape one pads to a 640 x 640 canvas = 400 patches, more than the 288-row sequences the
    attention tiles are sized for; no image has more than 240 valid patches, so the engine packs each sample's valid patches
    (the reference keeps max_b(h*w) patch rows: HF:136-159).  Real VQA / NLVR2 batches are of this kind.  Output and every
    gradient against the oracle, which pads and masks the whole canvas."""
    sizes = [(384, 640), (640, 384), (352, 480), (608, 384)]
    model, P = make_model(["vqa"], 11, precision=precision)
    enc = vo.synthetic_varres_encodings(sizes, seed=6)
    assert tuple(enc["pixel_values"].shape[-2:]) == (640, 640)
    target = vo.synthetic_vqa_targets(len(sizes), seed=6)
    texts = dict(input_ids=enc["input_ids"], token_type_ids=enc["token_type_ids"], attention_mask=enc["attention_mask"])
    images = dict(pixel_values=enc["pixel_values"], pixel_mask=enc["pixel_mask"])
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", images, texts, target)
    ws = model._host._engine.last_ws
    assert ws.compact and ws.NP == 400 and ws.NS == 240 and ws.S_pad == 288
    oloss, (opooled, ologits), _, oG = vo.train_step(P, "vqa", enc, target)
    _close(pooled, opooled, tol, "pooled")
    _close(logits, ologits, tol, "logits")
    _close(loss, oloss, tol, "loss")
    G = grads_of(model)
    if precision == "fp32":
        assert torch.equal(logits.argmax(-1).cpu(), ologits.argmax(-1))
        for n, g in oG.items():
            if n in G and not n.endswith("attention.key.bias"):
                _close(G[n], g, tol, n)          # incl. position_embeddings (transpose of the per-sample bilinear resize) and the patch projection
    else:
        for n in (vo.ENC + "embeddings.patch_embeddings.projection.weight", vo.ENC + "embeddings.position_embeddings"):
            assert abs(float(G[n].double().norm()) - float(oG[n].double().norm())) <= 6e-2 * float(oG[n].double().norm()), n


def test_hipgraph_replay_matches_eager():
    """The captured step (one hipGraph launch) must reproduce the eager launches bit for bit, across optimizer steps and new inputs."""
    dev = _dev()
    B = 4
    res = {}
    for mode in ("eager", "graph"):
        model, _ = make_model(["vqa"], 42, precision=H16)
        model.train()
        opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
        opt.zero_grad()
        losses = []
        for s in range(4):
            pixels, texts, target = _rand_batch(B, 100 + s, dev)
            fn = model.graphed_forward_backward if mode == "graph" else model.fused_forward_backward
            loss, (pooled, logits), _, _ = fn("vqa", pixels, texts, target)
            losses.append(loss.clone())
            opt.step()
            opt.zero_grad()
        res[mode] = (torch.stack(losses).cpu(), {n: p.detach().cpu().clone() for n, p in model.named_parameters()}, logits.detach().cpu().clone())
        del model, opt
    # the dW kernels accumulate split partial sums with fp32 atomics, so runs agree to rounding, not bitwise
    _close(res["graph"][0], res["eager"][0], 1e-3, "loss curve graph vs eager")   # atomics: run-to-run sum order
    # (final logits after 4 Adam steps at 10x the reference's learning rate: the +-lr steps of near-zero gradient entries described below
    # reach the logits at the 1e-3 level -- measured 0.3 - 1.01e-3 over ten runs of the SAME mode pair)
    _close(res["graph"][2], res["eager"][2], 3e-3, "final logits")
    # weights: Adam's g/sqrt(v) turns rounding-level gradient differences of near-zero entries into +-lr steps, so compare the
    # update as a whole (norm of the 4-step delta) rather than element-wise
    P0 = vo.init_params(["vqa"], 42)
    for n in ("vilt_encoder.vilt.encoder.layer.5.intermediate.dense.weight", "task_layer.vqa.3.bias", "vilt_encoder.vilt.embeddings.cls_token"):
        dg = (res["graph"][1][n] - P0[n]).double().norm()
        de = (res["eager"][1][n] - P0[n]).double().norm()
        assert abs(float(dg - de)) <= 2e-2 * float(de), (n, float(dg), float(de))


# ------------------------------------------------------------------------------------------------ data parallel, end to end
def dp_join(rank, world, port, backend):
    """Join a `world`-rank job from a spawned test process.  "gloo": every rank on the test box's one GPU (collectives on device tensors through the
    host).  "nccl" (= RCCL; needs >= `world` GPUs): rank r sees ONLY GPU r -- as cuda:0, which is what every helper of this file uses -- the
    one-process-per-GPU layout of the product (climb_amd/parallel.py::init_data_parallel).  Must run before the process touches the device."""
    import os as _os
    _os.environ["MASTER_ADDR"], _os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    if backend == "nccl":
        _os.environ["HIP_VISIBLE_DEVICES"] = str(rank)
        _os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def need_backend(backend, world=2):
    if backend == "nccl" and torch.cuda.device_count() < world:
        pytest.skip(f"RCCL with {world} ranks needs {world} GPUs (this box has {torch.cuda.device_count()}); the gloo variant covers the host logic")


BACKENDS = ["gloo", "nccl"]


def _dp_worker(rank, world, port, q, precision, gbatch=4, promise=False, backend="gloo"):
    dist = dp_join(rank, world, port, backend)
    if promise and precision == "fp16":
        os.environ["CLIMB_AMD_DP_COMPRESS"] = "fp16"          # (the half payload is opt-in since r05; the deferred un-cast exists for 16-bit payloads)
    try:
        from climb_amd.parallel import GradientAllReducer
        torch.manual_seed(1000 + rank)                                     # replicas start DIFFERENT: the broadcast must fix that
        model, _ = make_model(["vqa"], 42 + rank, precision=precision)
        ddp = GradientAllReducer(model)          # payload: fp32 in the fp32 mode, bf16 staging buffer in the bf16 mode
        from climb_amd.data.sharding import shard_of
        enc = vo.synthetic_encodings(gbatch, seed=21)
        tgt = vo.synthetic_vqa_targets(gbatch, seed=21)
        # this rank's strided share of the global batch and the weight its d(loss) carries (4: halves, weight 1; 3: shares of 2 and 1 examples,
        # weights 4/3 and 2/3 -- the ranks then run DIFFERENT weight-gradient paths (384 and 192 token rows) and must still issue the same collectives)
        sl, weight = shard_of(list(range(gbatch)), rank, world)
        images, texts = enc_to_inputs({k: v[sl] for k, v in enc.items()})
        opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
        model.train()
        losses, grads = [], None
        for it in range(2):
            # promise: the trainers' call -- FusedAdamW.step() is the next reader of the gradients, so the averaged 16-bit payload stays in the
            # reducer's buffer and the optimizer reads it there (no un-cast pass); `.grad` is then NOT the averaged gradient
            loss, _, _, _ = model.fused_forward_backward("vqa", images, texts, tgt[sl], grad_weight=weight, optimizer=opt if promise else None)
            if promise and it == 0 and rank == 0:
                grads = bool(model._host.engine()._g16)          # something really was deferred
            if not promise and it == 0 and rank == 0:          # the averaged gradients the optimizer is about to consume (numpy: pickled by value)
                grads = {n: p.grad.detach().cpu().numpy() for n, p in model.named_parameters() if p.grad is not None}
            opt.step()
            opt.zero_grad()
            losses.append(float(loss))
        ok = ddp.replicas_in_sync()
        flat = model._host.engine().flat.double()
        q.put((rank, ok, losses, grads, ddp.bytes_reduced, weight / world, [float(flat.sum()), float(flat.abs().sum()), float((flat * flat).sum())]))
    finally:
        dist.destroy_process_group()


def _run_two_ranks(precision, gbatch, promise, backend="gloo"):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q, precision, gbatch, promise, backend)) for r in range(2)]
    return collect_ranks(procs, q)


def collect_ranks(procs, q, timeout=600):
    """start the rank processes, gather one queue item per rank; a rank that dies fails the test at once (not after the queue's timeout)"""
    import queue as _queue
    import time as _time
    for p in procs:
        p.start()
    res, t0 = [], _time.time()
    while len(res) < len(procs):
        try:
            res.append(q.get(timeout=2))
        except _queue.Empty:
            dead = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode not in (None, 0)]
            if dead or _time.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError(f"rank processes died (rank, exit code) {dead}" if dead else f"no result after {timeout} s")
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    return res


@pytest.mark.parametrize("backend", BACKENDS)
def test_data_parallel_optimizer_reads_the_averaged_payload_in_place(backend):
    """r04: with the trainers' promise (`fused_forward_backward(..., optimizer=opt)`) the reducer leaves the averaged 16-bit payload in its staging
    buffer and FusedAdamW's flat pass reads it there with the averaging factor folded in (`climb_adamw_spans`, source 1) instead of an un-cast pass
    into the gradient buffer followed by a re-read.  Same arithmetic: after two steps on two ranks the parameters are what the run WITHOUT the promise
    produces (16-bit GEMM order noise aside: the two runs are separate processes), both replicas in sync."""
    if H16 == "fp32":
        pytest.skip("16-bit payload only")
    need_backend(backend)
    a = _run_two_ranks(H16, 4, False, backend)
    b = _run_two_ranks(H16, 4, True, backend)
    assert all(r[1] for r in a) and all(r[1] for r in b), "replicas diverged"
    assert b[0][3] is True, "nothing was deferred: the optimizer did not read the payload buffer"
    # (sum, sum |.|, sum of squares) of all 116 M parameters after the two steps.  The two runs are separate processes and a 16-bit step is not
    # bit-reproducible (atomics in the bias gradients and the stream-K tail), so: tolerances a missed or doubled update of any tensor would break by
    # orders of magnitude (lr = 1e-3 per element and step), measured against sum |.|
    sa, sb = a[0][6], b[0][6]
    assert abs(sa[0] - sb[0]) <= 1e-6 * sa[1] and abs(sa[1] - sb[1]) <= 1e-6 * sa[1] and abs(sa[2] - sb[2]) <= 1e-5 * sa[2], (sa, sb)
    for la, lb in zip(a[0][2], b[0][2]):
        assert abs(la - lb) <= 2e-3 * abs(la)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("precision,tol,gbatch", [("fp32", 1e-4, 3), (H16, 4e-2, 4), (H16, 4e-2, 3)] + ([("bf16x3", 3e-4, 3)] if H16 == "bf16" else []))      # bf16: payload rounding 2^-9 + bf16 GEMM order noise; 3 = uneven shards; bf16x3: fp32 payload, split-operand noise
def test_data_parallel_two_ranks_equals_single_process_on_the_global_batch(precision, tol, gbatch, backend):
    """Two processes (gloo collectives on device tensors, both on the test box's single GPU) train on halves of a batch of 4 through the
    real engine hooks: weights broadcast from rank 0, per-range gradient all-reduce during the backward, finish() before AdamW.
    The averaged gradients equal (fp32 summation order aside) those of ONE process on the whole batch -- the loss is a batch mean,
    so the average of the shard gradients is the global gradient -- and after two optimizer steps the replicas are bit-identical."""
    need_backend(backend)
    res = _run_two_ranks(precision, gbatch, False, backend)
    assert all(r[1] for r in res), "replicas diverged"
    assert res[0][4] > 0 and res[0][4] == res[1][4]
    # single process, global batch: its gradient is what the ranks' averaged gradient must be
    model, _ = make_model(["vqa"], 42, precision=precision)
    enc = vo.synthetic_encodings(gbatch, seed=21)
    tgt = vo.synthetic_vqa_targets(gbatch, seed=21)
    images, texts = enc_to_inputs(enc)
    model.train()
    loss, _, _, _ = model.fused_forward_backward("vqa", images, texts, tgt)
    dp = res[0][3]
    G = grads_of(model)
    assert set(G) == set(dp)
    worst = 0.0
    for n, g in G.items():
        if n.endswith("attention.key.bias"):
            continue             # mathematically zero (softmax shift invariance): only rounding noise on both sides
        worst = max(worst, _close(torch.from_numpy(dp[n]), g.cpu(), tol, f"averaged gradient {n}"))
    print(f"data parallel[{precision}] (2 ranks) vs single process on the global batch: worst gradient error {worst:.2e}")
    # the shard losses, weighted by their share of the global batch, are the global loss (r[5] = examples of the shard / examples of the batch)
    assert abs(sum(r[5] * r[2][0] for r in res) - float(loss)) < tol * abs(float(loss))


@pytest.mark.parametrize("B", [1, 7, 33, 48])
def test_odd_batch_sizes_select_other_kernel_variants(B):
    """The GEMM launchers pick tiles / split counts from the problem size (64x128, 96x192, 128x128, 192x192, ragged last tiles,
    split targets).  Batch sizes other than the benchmark's walk those branches: the bf16 mode must agree with the fp32 mode
    (itself pinned to the oracle) on logits, loss and every gradient norm."""
    dev = _dev()
    pixels, texts, target = _rand_batch(B, 100 + B, dev)
    out = {}
    for precision in ("fp32", H16):
        model, _ = make_model(["vqa"], 42, precision=precision)
        model.train()
        loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", pixels, texts, target)
        out[precision] = (float(loss), logits.detach().float().cpu(), {n: float(g.double().norm()) for n, g in grads_of(model).items()})
        assert bool(torch.isfinite(logits).all())
    l32, lg32, g32 = out["fp32"]
    l16, lg16, g16 = out[H16]
    assert abs(l16 - l32) <= 2e-3 * abs(l32)
    _close(lg16, lg32, BF16_TOL, f"logits bf16 vs fp32 at B={B}")
    top = max(g32.values())
    for n, v in g32.items():
        if v > 1e-3 * top:
            assert abs(g16[n] - v) <= 6e-2 * v, (n, v, g16[n])


def test_bf16_training_curve_tracks_fp32():
    """Accuracy-parity proxy: 30 optimizer steps (AdamW + the reference's warm-up / linear-decay schedule) on a fixed batch in both
    modes from the same initialisation.  The bf16 loss curve must stay within 0.2 % of the fp32 one at every step and end lower than
    it started (the model is actually fitting the batch)."""
    from climb_amd.train import polynomial_decay_schedule_with_warmup
    dev = _dev()
    B, steps = 8, 30
    pixels, texts, target = _rand_batch(B, 77, dev)
    curves = {}
    for precision in ("fp32", H16):
        model, _ = make_model(["vqa"], 42, precision=precision)
        model.train()
        opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
        sched = polynomial_decay_schedule_with_warmup(opt, 3, steps, 0.0, 1.0)
        opt.zero_grad()
        losses = []
        for _ in range(steps):
            loss, _, _, _ = model.fused_forward_backward("vqa", pixels, texts, target)
            opt.step()
            sched.step()
            opt.zero_grad()
            losses.append(float(loss))
        curves[precision] = np.array(losses)
    rel = np.abs(curves[H16] - curves["fp32"]) / curves["fp32"]
    print(f"loss fp32 {curves['fp32'][0]:.3f} -> {curves['fp32'][-1]:.3f}, {H16} {curves[H16][0]:.3f} -> {curves[H16][-1]:.3f}, max rel diff {rel.max():.2e}")
    assert rel.max() < 2e-3          # measured 1.1e-4
    assert curves["fp32"][-1] < 0.8 * curves["fp32"][0] and curves[H16][-1] < 0.8 * curves[H16][0]


# ------------------------------------------------------------------------------------------------ the upstream driver in miniature
class _SynthTask(torch.utils.data.Dataset):
    """A few pre-processed examples of one task (what the input pipeline hands to the trainers)."""

    def __init__(self, task, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.task, self.n = task, n
        self.ids = torch.randint(0, 30522, (n, 40), generator=g)
        nimg = 2 if task == "nlvr2" else 1
        self.px = torch.randn(n, nimg, 3, 384, 384, generator=g)
        if task == "vqa":
            self.tgt = torch.zeros(n, 3129)
            self.tgt[torch.arange(n), torch.randint(0, 3129, (n,), generator=g)] = 1.0
        else:
            self.tgt = torch.randint(0, 2, (n,), generator=g)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return dict(ids=self.ids[i], px=self.px[i], tgt=self.tgt[i])


def _collate(task):
    def fn(items):
        ids = torch.stack([it["ids"] for it in items])
        texts = dict(input_ids=ids, token_type_ids=torch.zeros_like(ids), attention_mask=torch.ones_like(ids))
        px = torch.cat([it["px"] for it in items])                       # NLVR2: image j of example i at row 2i+j (flattened list)
        batch = dict(images=px, raw_texts=texts)
        batch["target_scores" if task == "vqa" else "labels"] = torch.stack([it["tgt"] for it in items])
        return batch
    return fn


@pytest.mark.parametrize("cl_algorithm", ["ewc", "experience_replay"])
def test_upstream_continual_learning_driver_in_miniature(tmp_path, cl_algorithm):
    """REF/train/train_upstream_continual_learning.py:215-300 end to end on two tiny synthetic tasks (VQA -> NLVR2) with the
    package's trainers and plug-ins: train (scheduler, per-epoch eval, best-model deepcopy), checkpoint + results.json, then
    either Fisher estimation + EWC-penalised training of the next task or replay-buffer creation + replay steps, and finally the
    forgetting evaluation that reloads the saved checkpoint.  Checks plumbing and finiteness, not scores."""
    import copy as _copy
    import json
    from torch.utils.data import DataLoader
    from climb_amd.cl_algorithms import EWC, ExperienceReplayMemory
    from climb_amd.cl_evaluation import append_task_result, catastrophic_forgetting_eval, save_task_checkpoint
    from climb_amd.configs.model_configs import model_configs
    from climb_amd.configs.task_configs import task_configs as base_cfgs
    from climb_amd.modeling import create_continual_learner_map
    dev = _dev()
    random.seed(0)
    tasks = ["vqa", "nlvr2"]
    cfgs = {k: dict(v) for k, v in base_cfgs.items()}
    for t in tasks:
        cfgs[t]["num_epochs"] = 1
    args = types.SimpleNamespace(cl_algorithm=cl_algorithm, batch_size=4, replay_frequency=2, memory_percentage=0.5, memory_sampling_strategy="random",
                                 ewc_fisher_sample_percentage=0.5, ewc_loss_weight=100.0, ordered_cl_tasks=tasks, encoder_name="vilt",
                                 output_dir=str(tmp_path))
    model = create_continual_learner_map["vilt"](model_name_or_path="random-init:5", ordered_cl_tasks=tasks, model_config=model_configs["vilt"],
                                                 task_configs=cfgs, device=dev, precision=H16)
    replay = ExperienceReplayMemory() if cl_algorithm == "experience_replay" else None
    ewc = EWC(args) if cl_algorithm == "ewc" else None
    run_dir = tmp_path / "vilt-run"
    run_dir.mkdir()
    results_file = str(run_dir / "results.json")
    trainers = {}
    for task_num, task in enumerate(tasks):
        bs = args.batch_size // (2 if task == "nlvr2" else 1)
        train = DataLoader(_SynthTask(task, 16 // (2 if task == "nlvr2" else 1), 10 + task_num), batch_size=bs, shuffle=False, collate_fn=_collate(task))
        val = DataLoader(_SynthTask(task, 4, 20 + task_num), batch_size=bs, shuffle=False, collate_fn=_collate(task))
        trainer = cfgs[task]["task_trainer"](args, cfgs, model_configs["vilt"], dev, train, val)
        best_score, best = trainer.train(model, replay_memory=replay, ewc=ewc)
        assert 0.0 <= best_score <= 100.0 and isinstance(best["model"], type(model)) and best["model"] is not model
        ckpt_dir = run_dir / "checkpoints" / f"task{task_num}_{task}"
        save_task_checkpoint(best["model"], str(ckpt_dir))
        # random labels: the real score can equal the task's random baseline, where the reference's (and our) forgetting formula divides by
        # zero; record a margin of 10 points so the metric's plumbing can be exercised
        append_task_result(results_file, task_num, task, cfgs[task]["random_baseline_score"] + 10.0 + best_score, best["epoch"])
        trainers[task] = trainer
        if replay is not None:
            replay.add_task_memory_buffer(args=args, task_key=task, task_config=cfgs[task], task_trainer=trainer,
                                          memory_percentage=args.memory_percentage, sampling_strategy=args.memory_sampling_strategy)
            assert len(replay.memory_buffers[task]) == int(0.5 * len(train.dataset))
        elif task_num < len(tasks) - 1:
            ewc.save_task_parameters(task_key=task, model=model, task_trainer=trainer, device=dev)
            assert bool(torch.isfinite(ewc.fisher_flat[task]).all()) and float(ewc.fisher_flat[task].sum()) > 0
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    assert [r["task_key"] for r in json.load(open(results_file))] == tasks
    out = catastrophic_forgetting_eval(args, results_file, _copy.deepcopy(model), trainers)
    rec = out["nlvr2"]["vqa"]
    assert rec["transfer_tasks"] == "1->0" and 0.0 <= rec["absolute_transfer_score"] <= 100.0
