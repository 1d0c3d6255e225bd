import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import vilt_oracle as vo
import test_gpu_parity as tp
for fused in ("1", "0"):
    os.environ["CLIMB_AMD_FUSED_ADAMW"] = fused
    model, _ = tp.make_model(["vqa"], 42, precision="bf16x3")
    model.train()
    opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    opt.zero_grad()
    eng = model._host.engine()
    for s in range(4):
        enc = vo.synthetic_encodings(2, seed=300 + s)
        images, texts = tp.enc_to_inputs(enc)
        loss, _, _, _ = model.fused_forward_backward("vqa", images, texts, vo.synthetic_vqa_targets(2, seed=300 + s), optimizer=opt)
        opt.step(); opt.zero_grad()
        torch.cuda.synchronize()
        bad = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
        gb = int((~torch.isfinite(eng.grad)).sum())
        mb = int((~torch.isfinite(opt._m)).sum()), int((~torch.isfinite(opt._v)).sum())
        print(f"fused={fused} step {s}: loss {float(loss):.4f} non-finite params: {len(bad)} {bad[:4]}  grad non-finite {gb}  m/v non-finite {mb}")
