import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vilt_oracle as vo
import test_gpu_parity as tp
from climb_amd.train import polynomial_decay_schedule_with_warmup
z = np.load(os.path.join(ROOT, "tests/golden/vqa_b2_10steps.npz"))
m = tp._meta(z)
B, steps = int(m["B"]), int(m["steps"])
for precision in ("fp32", "bf16x3"):
    model, P = tp.make_model(m["tasks"].split(","), int(m["wseed"]), precision=precision)
    P0 = {n: t.clone() for n, t in P.items()}
    opt = model.create_optimizer({"lr": float(m["lr"]), "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    sched = polynomial_decay_schedule_with_warmup(opt, int(steps * 0.1), steps, 0.0, 1.0)
    model.train(); opt.zero_grad()
    for s in range(steps):
        enc = vo.synthetic_encodings(B, seed=100 + s)
        images, texts = tp.enc_to_inputs(enc)
        model.fused_forward_backward("vqa", images, texts, vo.synthetic_vqa_targets(B, seed=100 + s))
        opt.step(); sched.step(); opt.zero_grad()
    names = [str(n) for n in z["names"]]
    after = {n: p.detach().cpu() for n, p in model.named_parameters()}
    norms, _ = tp._summary({n: after[n] - P0[n] for n in names}, names)
    d = np.abs(norms - z["delta_norms"])
    order = np.argsort(-d)[:6]
    print(precision, "scale", float(z["delta_norms"].max()), [(names[i].replace("vilt_encoder.vilt.", ""), f"{d[i]:.2e}", f"ref {z['delta_norms'][i]:.3e}") for i in order])
