"""r05: what the EWC term costs a step (BASELINE.json configs[3], one GPU's share): the plain step, the step with the term folded into the optimizer's
passes (default), and with the separate penalty pass of r01 - r04 (CLIMB_AMD_EWC_FOLD=0), interleaved.  GPU box:  python tools/ewc_ab.py"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from climb_amd.modeling import create_continual_learner_map
from climb_amd.configs.task_configs import task_configs
from climb_amd.configs.model_configs import model_configs
from climb_amd.cl_algorithms import EWC

dev = torch.device("cuda:0")
B, T = 64, 40
g = torch.Generator().manual_seed(1)
m = create_continual_learner_map["vilt"](model_name_or_path="random-init:42", ordered_cl_tasks=["vqa", "nlvr2"], model_config=model_configs["vilt"],
                                         task_configs=task_configs, device=dev, precision="bf16")
m.train()
tx = dict(input_ids=torch.randint(0, 30522, (B, T), generator=g).to(dev), token_type_ids=torch.zeros(B, T, dtype=torch.long, device=dev),
          attention_mask=torch.ones(B, T, dtype=torch.long, device=dev))
pix = torch.randn(B, 3, 384, 384, generator=g).to(dev)
tgt = torch.zeros(B, 3129)
tgt[torch.arange(B), torch.randint(0, 3129, (B,), generator=g)] = 1.0
tgt = tgt.to(dev)
ewc = EWC(types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=100.0))
enc = m.get_encoder()
ewc.set_task_state("nlvr2", m, {n: torch.rand_like(p) * 1e-4 for n, p in enc.named_parameters()}, {n: p.detach().clone() + 0.01 for n, p in enc.named_parameters()})
opt = m.create_optimizer({"lr": 1e-5, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
opt.zero_grad()


def run(e, fold, steps=15, warm=3):
    os.environ["CLIMB_AMD_EWC_FOLD"] = str(int(fold))
    for i in range(steps + warm):
        if i == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = m.fused_forward_backward("vqa", pix, tx, tgt, ewc=e, optimizer=opt)
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, (float(out[3]) if out[3] is not None else None)


res = {"plain": [], "ewc in both optimizer passes (1)": [], "ewc in the flat pass, dW unfused (2)": [], "ewc separate pass (0)": []}
for rep in range(3):
    res["plain"].append(run(None, 1)[0])
    a = run(ewc, 1)
    res["ewc in both optimizer passes (1)"].append(a[0])
    c = run(ewc, 2)
    res["ewc in the flat pass, dW unfused (2)"].append(c[0])
    b = run(ewc, 0)
    res["ewc separate pass (0)"].append(b[0])
    print(f"pass {rep}: penalty value (1) {a[1]:.4f} | (2) {c[1]:.4f} | (0) {b[1]:.4f}", flush=True)
med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
for k, v in res.items():
    print(f"{k:40s} " + " ".join(f"{x:7.3f}" for x in v) + f"   median {med[k]:7.3f} ms/step" + ("" if k == "plain" else f"   (+{med[k] - med['plain']:.3f} over the plain step)"))
