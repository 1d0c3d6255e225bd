"""the configs[2] step (Houlsby adapters r = 16 on a frozen base, bs = 64) for a kernel trace: python tools/probe/adapter_step.py [plain|adapter] [steps]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd.modeling import create_continual_learner_map
from climb_amd.configs.task_configs import task_configs
from climb_amd.configs.model_configs import model_configs
from climb_amd.cl_algorithms import AdapterHandler
mode = sys.argv[1] if len(sys.argv) > 1 else "adapter"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
B, T = 64, 40
g = torch.Generator().manual_seed(1)
m = create_continual_learner_map["vilt"](model_name_or_path="random-init:42", ordered_cl_tasks=["vqa", "nlvr2"], model_config=model_configs["vilt"],
                                         task_configs=task_configs, device=dev, precision=os.environ.get("CLIMB_AMD_PRECISION", "bf16"))
m.train()
if mode == "adapter":
    h = AdapterHandler("vanilla", types.SimpleNamespace(adapter_config="houlsby", adapter_reduction_factor=16, ordered_cl_tasks=["vqa", "nlvr2"]))
    h.add_adapters_to_model(m)
    h.activate_adapter_for_training(task_key="vqa", model=m)
tx = dict(input_ids=torch.randint(0, 30522, (B, T), generator=g).to(dev), token_type_ids=torch.zeros(B, T, dtype=torch.long, device=dev),
          attention_mask=torch.ones(B, T, dtype=torch.long, device=dev))
pix = torch.randn(B, 3, 384, 384, generator=g).to(dev)
tgt = torch.zeros(B, 3129)
tgt[torch.arange(B), torch.randint(0, 3129, (B,), generator=g)] = 1.0
tgt = tgt.to(dev)
opt = m.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
opt.zero_grad()
import time
for i in range(steps + 3):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    m.fused_forward_backward("vqa", pix, tx, tgt, optimizer=opt)
    opt.step()
    opt.zero_grad()
torch.cuda.synchronize()
print(f"{mode}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
