"""fp16-operand mode soak: 300 optimizer steps at the benchmark's shape with a learning rate 10x the reference's (weights move fast), fresh
random batches every step; nothing may become inf / nan and the loss must fall.  CLIMB_AMD_H16=fp16 python tools/probe/fp16_soak.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd.modeling import create_continual_learner_map
from climb_amd.configs.task_configs import task_configs
from climb_amd.configs.model_configs import model_configs
dev = torch.device("cuda:0")
B, T = 64, 40
prec = "fp16" if os.environ.get("CLIMB_AMD_H16") == "fp16" else "bf16"
m = create_continual_learner_map["vilt"](model_name_or_path="random-init:42", ordered_cl_tasks=["vqa"], model_config=model_configs["vilt"], task_configs=task_configs, device=dev, precision=prec)
m.train()
opt = m.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8}); opt.zero_grad()
g = torch.Generator(device=dev).manual_seed(0)
losses = []
for i in range(300):
    tx = dict(input_ids=torch.randint(0, 30522, (B, T), generator=g, device=dev), token_type_ids=torch.zeros(B, T, dtype=torch.long, device=dev), attention_mask=torch.ones(B, T, dtype=torch.long, device=dev))
    pix = torch.randn(B, 3, 384, 384, generator=g, device=dev) * (1.0 + 3.0 * (i % 7 == 0))       # every 7th batch: 4x larger pixels
    t = torch.zeros(B, 3129, device=dev); t[torch.arange(B, device=dev), torch.randint(0, 3129, (B,), generator=g, device=dev)] = 1.0
    loss, _, _, _ = m.fused_forward_backward("vqa", pix, tx, t)
    opt.step(); opt.zero_grad()
    losses.append(float(loss))
ok = all(bool(torch.isfinite(p).all()) for p in m.parameters()) and all(l == l and abs(l) < 1e30 for l in losses)
print(prec, "finite", ok, "loss", [round(l, 2) for l in losses[:3]], "->", [round(l, 2) for l in losses[-3:]], "max |param|", max(float(p.abs().max()) for p in m.parameters()))
