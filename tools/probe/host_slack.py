"""Is the bs = 64 step ever waiting for the host?  Adds a busy-wait of D microseconds before optimizer.step() and before the forward:
if the step time does not move, the host runs that far ahead of the GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climb_amd.modeling import create_continual_learner_map
from climb_amd.configs.task_configs import task_configs
from climb_amd.configs.model_configs import model_configs

dev = torch.device("cuda:0")
B, T = 64, 40
model = create_continual_learner_map["vilt"](model_name_or_path="random-init:42", ordered_cl_tasks=["vqa"], model_config=model_configs["vilt"],
                                             task_configs=task_configs, device=dev, precision="bf16")
model.train()
g = torch.Generator().manual_seed(1)
texts = dict(input_ids=torch.randint(0, 30522, (B, T), generator=g).to(dev), token_type_ids=torch.zeros(B, T, dtype=torch.long, device=dev),
             attention_mask=torch.ones(B, T, dtype=torch.long, device=dev))
pixels = torch.randn(B, 3, 384, 384, generator=g).to(dev)
target = torch.zeros(B, 3129); target[torch.arange(B), torch.randint(0, 3129, (B,), generator=g)] = 1.0; target = target.to(dev)
opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
opt.zero_grad()


def spin(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us:
        pass


def run(d_fwd, d_opt, steps=20):
    for i in range(steps + 5):
        if i == 5:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        spin(d_fwd)
        model.fused_forward_backward("vqa", pixels, texts, target, optimizer=opt)
        spin(d_opt)
        opt.step(); opt.zero_grad()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for d_fwd, d_opt in [(0, 0), (0, 200), (0, 1000), (200, 0), (1000, 0), (2000, 2000)]:
    print(f"host delay before forward {d_fwd:5d} us, before optimizer {d_opt:5d} us: {run(d_fwd, d_opt):7.3f} ms/step")
# host time of one step's launches with the GPU idle-proof: enqueue without waiting
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    model.fused_forward_backward("vqa", pixels, texts, target, optimizer=opt); opt.step(); opt.zero_grad()
t_host = (time.perf_counter() - t0) / 10 * 1e3
torch.cuda.synchronize()
print(f"host enqueue time per step (includes back-pressure if the queue fills): {t_host:.3f} ms")
