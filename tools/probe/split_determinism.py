"""run-to-run stability of the bf16x3 step: the same batch N times; every saved activation and gradient against the first run's"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vilt_oracle as vo
import test_gpu_parity as tp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
model, P = tp.make_model(["vqa"], 42, precision="bf16x3")
model.train()
enc = vo.synthetic_encodings(B, seed=100)
images, texts = tp.enc_to_inputs(enc)
tgt = vo.synthetic_vqa_targets(B, seed=100)
eng = model._host.engine()
ref = None
worst = {}
for it in range(N):
    model._host.drop_grads() if hasattr(model._host, "drop_grads") else None
    eng.zero_grad()
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", images, texts, tgt)
    torch.cuda.synchronize()
    ws = eng.last_ws
    snap = {"loss": loss.clone(), "pooled": pooled.clone(), "grad": eng.grad.clone()}
    for i in range(12):
        for nm in ("x", "qkv", "ctx", "h1", "u"):
            snap[f"{nm}{i}"] = getattr(ws, nm)[i].clone()
        for nm in ("xn", "hn", "a", "ctx_s"):
            snap[f"{nm}{i}"] = getattr(ws, nm)[i].float().clone()
        snap[f"lse{i}"] = ws.lse[i].clone()
    if ref is None:
        ref = snap
        continue
    for k, v in snap.items():
        d = float((v.float() - ref[k].float()).abs().max())
        sc = float(ref[k].float().abs().max()) + 1e-30
        if d / sc > worst.get(k, 0.0):
            worst[k] = d / sc
bad = {k: v for k, v in worst.items() if v > 1e-5}
print("B", B, "iters", N, "tensors deviating > 1e-5:", sorted(bad.items(), key=lambda kv: -kv[1])[:20])
print("grad dev", worst.get("grad"), "loss dev", worst.get("loss"), "pooled dev", worst.get("pooled"))
order = [f"{nm}{i}" for i in range(12) for nm in ("xn", "qkv", "lse", "ctx", "ctx_s", "h1", "hn", "u", "a", "x")]
first = next((k for k in order if worst.get(k, 0) > 1e-6), None)
print("first deviating tensor in forward order:", first, worst.get(first) if first else None)
