"""Where the bf16 throughput mode's error comes from.  Runs the SAME batch (the inputs of tests/golden/vqa_b64.npz:
64 seeded sequences, the weights of the fixture) through the fp32 parity mode and the bf16 mode and tabulates, per encoder layer, the
relative error (max|d| / max|ref| and rms(d) / rms(ref)) of every saved intermediate: LN outputs, qkv, attention context, the two
residual sums, the pre-GELU and GELU outputs.  GPU box:  python tools/bf16_error_budget.py > gpurun_out/bf16_budget.txt"""
import os
import sys

os.environ["CLIMB_AMD_CLS_ONLY_LAST"] = "0"      # this table reads EVERY row of every saved intermediate (the default; pinned here)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import vilt_oracle as vo          # seeded inputs only (this is a measurement tool, not the product path)
from climb_amd.configs.model_configs import model_configs
from climb_amd.configs.task_configs import task_configs
from climb_amd.modeling import create_continual_learner_map

dev = torch.device("cuda:0")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vqa_b64.npz"))
m = dict(kv.split("=", 1) for kv in str(z["meta"][0]).split(";"))
tasks, B = m["tasks"].split(","), int(m["B"])
P = vo.init_params(tasks, int(m["wseed"]))
enc = vo.synthetic_encodings(B, seed=int(m["dseed"]))
texts = dict(input_ids=enc["input_ids"], token_type_ids=enc["token_type_ids"], attention_mask=enc["attention_mask"])
target = vo.synthetic_vqa_targets(B, seed=int(m["dseed"]))


def run(precision):
    model = create_continual_learner_map["vilt"](model_name_or_path="random-init:0", ordered_cl_tasks=tasks, model_config=model_configs["vilt"],
                                                 task_configs=task_configs, device=dev, precision=precision)
    model.load_state_dict(P, strict=True)
    model.to(dev)
    model.train()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", enc["pixel_values"], texts, target)
    ws = model._host._engine.last_ws
    S = ws.S
    keep = {"pooled": pooled.float().cpu(), "logits": logits.float().cpu()}
    for name in ("x", "xn", "qkv", "ctx", "h1", "hn", "u", "a"):
        for i, t in enumerate(getattr(ws, name)):
            keep[f"{name}[{i}]"] = t.view(B, ws.S_pad, -1)[:, :S].float().cpu()      # valid rows only
    keep["clsn"] = ws.clsn.float().cpu().clone()            # final LayerNorm of row 0 (the pooler's input)
    keep["grads"] = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if p.grad is not None}
    del model
    torch.cuda.empty_cache()
    return keep


H16 = os.environ.get("CLIMB_AMD_H16", "bf16")          # CLIMB_AMD_H16=fp16: the same table for the IEEE-half build
ref, low = run("fp32"), run(H16)


def err(a, b):
    d = (a.double() - b.double())
    return float(d.abs().max() / (b.double().abs().max() + 1e-30)), float(d.pow(2).mean().sqrt() / (b.double().pow(2).mean().sqrt() + 1e-30))


print(f"{H16} mode vs fp32 mode, B = 64, valid token rows only.  columns: max|d|/max|ref|   rms(d)/rms(ref)")
print(f"{'layer':>5s} | " + " | ".join(f"{n:^19s}" for n in ("x (residual in)", "LN1(x)", "qkv", "ctx", "h1 = x+attn", "LN2(h1)", "u (pre-GELU)", "a = GELU(u)")))
for i in range(12):
    cells = []
    for n in ("x", "xn", "qkv", "ctx", "h1", "hn", "u", "a"):
        e = err(low[f"{n}[{i}]"], ref[f"{n}[{i}]"])
        cells.append(f"{e[0]:.2e} {e[1]:.2e}")
    print(f"{i:5d} | " + " | ".join(f"{c:^19s}" for c in cells))
e = err(low["x[12]"], ref["x[12]"])
print(f"x[12] (encoder output, all rows): {e[0]:.2e} {e[1]:.2e};  CLS row only: %.2e %.2e" % err(low["x[12]"][:, 0], ref["x[12]"][:, 0]))
print("pooled: %.2e %.2e   logits: %.2e %.2e" % (err(low["pooled"], ref["pooled"]) + err(low["logits"], ref["logits"])))
# where does the max-norm of `pooled` come from?  Stage by stage from the encoder output's CLS row to pooled, with the
# element count of each tensor: for noise of one scale the max over N elements sits sqrt(2 ln N) sigma out, so max / rms of ~4 at N = 49 152
# is what a uniform error gives, not a localised amplification
import math
for name, a, b in (("x[12] CLS row", low["x[12]"][:, 0], ref["x[12]"][:, 0]), ("final LN(CLS row)", low["clsn"], ref["clsn"]), ("pooled = tanh(Wp LN)", low["pooled"], ref["pooled"])):
    d = (a.double() - b.double())
    n = d.numel()
    rms_abs, max_abs = float(d.pow(2).mean().sqrt()), float(d.abs().max())
    print(f"  {name:22s} N = {n:6d}: max|d|/max|ref| {max_abs / float(b.abs().max()):.2e}  rms(d)/rms(ref) {rms_abs / float(b.double().pow(2).mean().sqrt()):.2e}  "
          f"max|d| / rms(d) = {max_abs / rms_abs:.2f} (Gaussian expectation sqrt(2 ln N) = {math.sqrt(2 * math.log(n)):.2f});  max|ref| / rms(ref) = "
          f"{float(b.abs().max()) / float(b.double().pow(2).mean().sqrt()):.2f}")
agree = float((low["logits"].argmax(-1) == ref["logits"].argmax(-1)).float().mean())
srt = ref["logits"].sort(-1).values
margin = (srt[:, -1] - srt[:, -2])
flipped = (low["logits"].argmax(-1) != ref["logits"].argmax(-1))
print(f"argmax agreement {agree:.4f}; top-2 logit margin of the fp32 mode: min {float(margin.min()):.2e} median {float(margin.median()):.2e}; "
      f"margins of the flipped rows: {[round(float(v), 5) for v in margin[flipped]]}; max |logit error| {float((low['logits'] - ref['logits']).abs().max()):.2e}")
gn = []
for n, g in ref["grads"].items():
    if float(g.norm()) > 1e-3 * max(float(v.norm()) for v in ref["grads"].values()):
        gn.append((float((low["grads"][n].double() - g.double()).norm() / g.double().norm()), n))
gn.sort(reverse=True)
print("gradients, ||d|| / ||ref|| per tensor: median %.2e, worst: %s" % (float(np.median([v for v, _ in gn])), [(round(v, 4), n.split("vilt.")[-1]) for v, n in gn[:4]]))
