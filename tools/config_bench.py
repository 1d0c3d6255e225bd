"""Step time of the BASELINE.json configurations next to the headline one, bs = 64 sequences per GPU, bf16 mode, synthetic inputs
resident in HBM: sequential fine-tuning, + EWC penalty, Houlsby adapters (base frozen), bottom-9 layers frozen, NLVR2 (32 pairs =
64 sequences), VCR (16 questions x 4 choices).  Run on the GPU box:  python tools/config_bench.py"""
import os
import sys
import time
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from climb_amd.modeling import create_continual_learner_map
from climb_amd.configs.task_configs import task_configs
from climb_amd.configs.model_configs import model_configs
from climb_amd.cl_algorithms import EWC, AdapterHandler

dev = torch.device("cuda:0")
B, T = 64, 40
g = torch.Generator().manual_seed(1)


def make(tasks):
    m = create_continual_learner_map["vilt"](model_name_or_path="random-init:42", ordered_cl_tasks=tasks, model_config=model_configs["vilt"],
                                             task_configs=task_configs, device=dev, precision="bf16")
    m.train()
    return m


def texts(n):
    return dict(input_ids=torch.randint(0, 30522, (n, T), generator=g).to(dev), token_type_ids=torch.zeros(n, T, dtype=torch.long, device=dev),
                attention_mask=torch.ones(n, T, dtype=torch.long, device=dev))


def run(name, model, task, images, tx, target, ewc=None, steps=12, warm=4):
    opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    opt.zero_grad()
    for i in range(steps + warm):
        if i == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        model.fused_forward_backward(task, images, tx, target, ewc=ewc, optimizer=opt)
        opt.step()
        opt.zero_grad()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    nseq = int(tx["input_ids"].shape[0]) if task != "nlvr2" else int(images.shape[0])
    print(f"{name:46s} {ms:7.2f} ms/step  {nseq / ms * 1e3:7.0f} encoder sequences/s")


pix = torch.randn(B, 3, 384, 384, generator=g).to(dev)
vqa_t = torch.zeros(B, 3129)
vqa_t[torch.arange(B), torch.randint(0, 3129, (B,), generator=g)] = 1.0
vqa_t = vqa_t.to(dev)

m = make(["vqa", "nlvr2"])
run("sequential fine-tuning, VQA (headline)", m, "vqa", pix, texts(B), vqa_t)

ewc = EWC(types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=100.0))
enc = m.get_encoder()
star = {n: p.detach().clone() for n, p in enc.named_parameters()}
fisher = {n: torch.rand_like(p) * 1e-4 for n, p in enc.named_parameters()}
ewc.set_task_state("nlvr2", m, fisher, star)
run("+ EWC penalty (1 previous task)", m, "vqa", pix, texts(B), vqa_t, ewc=ewc)

m2 = make(["vqa", "nlvr2"])
m2.get_encoder().freeze_bottom_k_layers(9)
run("bottom 9 layers frozen", m2, "vqa", pix, texts(B), vqa_t)

m3 = make(["vqa", "nlvr2"])
h = AdapterHandler("vanilla", types.SimpleNamespace(adapter_config="houlsby", adapter_reduction_factor=16, ordered_cl_tasks=["vqa", "nlvr2"]))
h.add_adapters_to_model(m3)
h.activate_adapter_for_training(task_key="vqa", model=m3)
run("Houlsby adapters r=16, base frozen", m3, "vqa", pix, texts(B), vqa_t)

m4 = make(["vqa", "nlvr2"])
pairs = torch.randn(B, 3, 384, 384, generator=g).to(dev)          # the processor sees the flattened image list (REF/modeling/vilt.py:281-288)
run("NLVR2: 32 pairs = 64 sequences", m4, "nlvr2", pairs, texts(B // 2), torch.randint(0, 2, (B // 2,), generator=g).to(dev))

m5 = make(["snli-ve", "vcr"])
m5.eval()       # the VCR head's Dropout(0.1) is host-seeded; timing is the same
run("VCR: 16 questions x 4 choices = 64 sequences", m5, "vcr", torch.randn(B // 4, 3, 384, 384, generator=g).to(dev), texts(B),
    torch.randint(0, 4, (B // 4,), generator=g).to(dev))

# r04: the per-rank shares of `--batch_size 64` (the trajectory-preserving data-parallel launch of INTEGRATION.md keeps the GLOBAL batch: on 2 / 4 / 8
# GPUs a rank steps 32 / 16 / 8 sequences) -- what one GPU does with them, so that the scaling that launch can reach is known before hardware sees it
m6 = make(["vqa", "nlvr2"])
for b in (32, 16, 8):
    tb = torch.zeros(b, 3129)
    tb[torch.arange(b), torch.randint(0, 3129, (b,), generator=g)] = 1.0
    run(f"sequential FT, {b} sequences per GPU (= 64 / {64 // b} GPUs)", m6, "vqa", pix[:b].contiguous(), texts(b), tb.to(dev))
