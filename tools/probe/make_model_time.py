import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_parity as tp
from climb_amd.modeling import create_continual_learner_map
from climb_amd.configs.task_configs import task_configs
from climb_amd.configs.model_configs import model_configs
dev = torch.device("cuda:0")
torch.zeros(1, device=dev); torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    model = create_continual_learner_map["vilt"](model_name_or_path="random-init:empty", ordered_cl_tasks=["vqa"], model_config=model_configs["vilt"], task_configs=task_configs, device=dev, precision="fp32")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    P = tp._seeded_params(["vqa"], 42)
    t2 = time.perf_counter()
    model.load_state_dict({k: v for k, v in P.items()}, strict=True)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    model.to(dev); torch.cuda.synchronize(); t4 = time.perf_counter()
    from oracle import vilt_oracle as vo
    enc = vo.synthetic_encodings(2, seed=1); tgt = vo.synthetic_vqa_targets(2, seed=1)
    images, texts = tp.enc_to_inputs(enc)
    t5 = time.perf_counter()
    model.train(); model.fused_forward_backward("vqa", images, texts, tgt); torch.cuda.synchronize(); t6 = time.perf_counter()
    G = tp.grads_of(model); t7 = time.perf_counter()
    print(f"iter {it}: create {t1-t0:.2f}  params {t2-t1:.2f}  load_state_dict {t3-t2:.2f}  to {t4-t3:.2f}  inputs {t5-t4:.2f}  step {t6-t5:.2f}  grads_of {t7-t6:.2f}")
    del model
